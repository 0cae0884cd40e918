// Warp + z-buffered splat for gfx950 (MI355X) — binned, LDS-resident z-buffer (no global atomics).
//
// Replaces reference panoptic_forecasting/models/pc_transform/pc_transform_model.py:41-150:
// unproject (:41-59), camera->vehicle (:63), ego warp (:68), vehicle->camera + projection (:71-78),
// validity (:83-89), sentinel (:105), 4-corner bins (:106-117), torch_scatter.scatter_min (:118-119)
// and the winner gather (:120-139).
//
// The first version scattered with 64-bit global atomicMin: 3.0 ms per 1024x2048 frame triple, because
// agent-scope atomics are executed memory-side across the 8 non-coherent XCD L2s
// (profiles/r01_a_first_path_kernel_stats.txt).  This version is a two-pass binning rasteriser:
//
//   bin_kernel     one workgroup per 16x64 SOURCE tile: exact-order fp32 projection (this file is built with
//                  -ffp-contract=off; every product/sum rounded separately, IEEE divides), stored as 8 B per
//                  point {bits(z), packed bins}; block max(z) partial; the bounding box of the destination bins
//                  its VALID points reach; a byte mark per bin reached by an INVALID point (plain stores of the
//                  constant 1 - no atomics needed); optional result2d.
//   raster_kernel  one workgroup per 32x128 DESTINATION tile: scans the bounding boxes, re-reads the stored
//                  projections (8 B/point) of only the source tiles that can reach it, and resolves "min depth,
//                  ties -> lowest element index" with 64-bit ds_min on a packed key in a 32 KB LDS z-buffer it
//                  alone owns; then writes seg/depth for its pixels.  The z-buffer never exists in HBM.
//
// Packed key (valid points have z > 0, so raw fp32 bits are monotone as unsigned):
//     [ bits(z) : 32 | e : 32 ],  e = r*P + t*N + n  (corner replica r, P = T*N)   — pc_transform_model.py:112
// Invalid points all carry the same depth (max+1, :105) and a zeroed payload (:133), so which of them wins a
// bin is unobservable: a per-bin "touched by an invalid point" byte reproduces the reference output exactly
// (seg 0, depth max+1) wherever no valid point lands; untouched bins give seg 0, depth -1 (:136-138).
#include "pf_common.h"
#include "pf_prof.h"
#include <cstdlib>

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // 100 MHz wall-clock stamps of one raster workgroup + whole-kernel workgroup-time sum
#define RPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 100 && blockIdx.y == 1 && blockIdx.z == 0 && a.probe) a.probe[i] = wall_clock64(); } while (0)
#else
#define RPROBE(i) do { } while (0)
#endif

namespace pf {
long long *probe_buffer();

constexpr unsigned long long kEmpty = ~0ull;
constexpr int kThreads = 256;
constexpr int kSrcTH = 16, kSrcTW = 64;   // source tile (pixels); 256 threads x 4 consecutive pixels
constexpr int kDstTH = 32, kDstTW = 128;  // destination tile owned by one raster workgroup (32 KB LDS)
constexpr int kScan = 256;                // bounding boxes tested per thread-pass
constexpr int kScanBatches = 8;           // passes per list fill (list holds kScan * kScanBatches tile ids)
constexpr int kZSlots = 64;               // atomicMax slots per z-buffer group (spreads the memory-side atomics)

struct SplatArgs {
    const float *depth;
    const uint8_t *mask;
    const uint8_t *seg;
    const float *Kinv, *E, *Tt, *Einv, *K;
    int4 *bbox;           // [B][T][src tiles]  (x0min, y0min, x1max, y1max) of valid points' bins
    unsigned *zmax_part;  // [G][kZSlots]       order-preserving u32 of max(z) per z-buffer group (atomicMax, zeroed per call)
    uint8_t *inv_mark;    // [B*G][N]           1 where an invalid point lands
    uint2 *proj;          // [B][T][N]          {bits(z), x0 | y0<<13 | (x1!=x0)<<26 | (y1!=y0)<<27 | valid<<28}
    uint8_t *out_seg;
    float *out_depth;
    long long *out_r2d;
    int B, T_total, t_first, T, H, W, C, per_frame;
    int zgroups_per_sample;   // 0: one sentinel per z-buffer group over the whole call (:105); G: one per (sample, group)
    int stx, sty;         // source tiles per row / column
    int dtx, dty;         // destination tiles per row / column
    long long *probe;     // PF_PROBE builds only
};

__device__ __forceinline__ unsigned float_to_ordered(float f) {
    unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// acc = 0; acc = acc + M[k]*v[k] (k ascending), every op rounded on its own.
__device__ __forceinline__ float dot3(const float *m, float a, float b, float c) {
    float acc = 0.0f;
    acc = __fadd_rn(acc, __fmul_rn(m[0], a));
    acc = __fadd_rn(acc, __fmul_rn(m[1], b));
    acc = __fadd_rn(acc, __fmul_rn(m[2], c));
    return acc;
}
__device__ __forceinline__ float dot4(const float *m, float a, float b, float c, float d) {
    float acc = 0.0f;
    acc = __fadd_rn(acc, __fmul_rn(m[0], a));
    acc = __fadd_rn(acc, __fmul_rn(m[1], b));
    acc = __fadd_rn(acc, __fmul_rn(m[2], c));
    acc = __fadd_rn(acc, __fmul_rn(m[3], d));
    return acc;
}

struct Camera {
    float Kinv[9], E[16], Tm[16], Einv[16], K[9];
};

__device__ __forceinline__ void load_camera(const SplatArgs &a, int b, int t, Camera &c) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        c.Kinv[i] = a.Kinv[b * 9 + i];
        c.K[i] = a.K[b * 9 + i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        c.E[i] = a.E[b * 16 + i];
        c.Einv[i] = a.Einv[b * 16 + i];
        c.Tm[i] = a.Tt[((long long)b * a.T_total + t) * 16 + i];
    }
}

struct Proj {
    float z;
    int x0, y0, x1, y1;  // clamped floor/ceil bins
    bool valid;
};

// pc_transform_model.py:54-114 for one pixel.  Used by BOTH kernels so they agree bit for bit.
__device__ __forceinline__ Proj project(const Camera &c, int x, int y, float d, bool m, float Wf, float Hf) {
    const float u = (float)x, v = (float)y;
    const float r0 = dot3(c.Kinv + 0, u, v, 1.0f), r1 = dot3(c.Kinv + 3, u, v, 1.0f), r2 = dot3(c.Kinv + 6, u, v, 1.0f);
    const float c0 = __fmul_rn(r0, d), c1 = __fmul_rn(r1, d), c2 = __fmul_rn(r2, d);                    // :55-59
    const float v0 = dot4(c.E + 0, c0, c1, c2, 1.0f), v1 = dot4(c.E + 4, c0, c1, c2, 1.0f),
                v2 = dot4(c.E + 8, c0, c1, c2, 1.0f), v3 = dot4(c.E + 12, c0, c1, c2, 1.0f);           // :63
    const float w0 = dot4(c.Tm + 0, v0, v1, v2, v3), w1 = dot4(c.Tm + 4, v0, v1, v2, v3),
                w2 = dot4(c.Tm + 8, v0, v1, v2, v3), w3 = dot4(c.Tm + 12, v0, v1, v2, v3);             // :68
    const float e0 = dot4(c.Einv + 0, w0, w1, w2, w3), e1 = dot4(c.Einv + 4, w0, w1, w2, w3),
                e2 = dot4(c.Einv + 8, w0, w1, w2, w3), e3 = dot4(c.Einv + 12, w0, w1, w2, w3);         // :71
    const float px = __fdiv_rn(e0, e3), py = __fdiv_rn(e1, e3), z = __fdiv_rn(e2, e3);                // :72-73
    const float q0 = dot3(c.K + 0, px, py, z), q1 = dot3(c.K + 3, px, py, z), q2 = dot3(c.K + 6, px, py, z);  // :74
    const float uu = __fdiv_rn(q0, q2), vv = __fdiv_rn(q1, q2);                                       // :75-78
    Proj p;
    p.z = z;
    const bool inb = (uu >= 0.0f) && (uu < Wf) && (vv >= 0.0f) && (vv < Hf);                          // :83-86
    p.valid = m && (z > 0.0f) && inb;                                                                 // :87-89
    // :106-114 floor/ceil then clamp (clamping in float first keeps the int conversion in range)
    p.x0 = (int)fminf(fmaxf(floorf(uu), 0.0f), Wf - 1.0f);
    p.x1 = (int)fminf(fmaxf(ceilf(uu), 0.0f), Wf - 1.0f);
    p.y0 = (int)fminf(fmaxf(floorf(vv), 0.0f), Hf - 1.0f);
    p.y1 = (int)fminf(fmaxf(ceilf(vv), 0.0f), Hf - 1.0f);
    return p;
}

// 4 consecutive pixels of a row (x multiple of 4): vector loads when the row allows it
__device__ __forceinline__ void load4(const SplatArgs &a, long long base, int x, int y, float d[4], bool m[4]) {
    const long long i = base + (long long)y * a.W + x;
    if ((a.W & 3) == 0 && x + 3 < a.W) {
        const float4 dv = *reinterpret_cast<const float4 *>(a.depth + i);
        const uchar4 mv = *reinterpret_cast<const uchar4 *>(a.mask + i);
        d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
        m[0] = mv.x != 0; m[1] = mv.y != 0; m[2] = mv.z != 0; m[3] = mv.w != 0;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = x + k < a.W;
            d[k] = in ? a.depth[i + k] : 1.0f;
            m[k] = in ? a.mask[i + k] != 0 : false;
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void bin_kernel(SplatArgs a) {
    const int tile = blockIdx.x, tl = blockIdx.y, b = blockIdx.z;
    const int t = a.t_first + tl;
    const int ty0 = (tile / a.stx) * kSrcTH, tx0 = (tile % a.stx) * kSrcTW;
    const long long N = (long long)a.H * a.W;
    Camera cam;
    load_camera(a, b, t, cam);
    const long long in_base = ((long long)b * a.T_total + t) * N;
    const int g = a.per_frame ? tl : 0, G = a.per_frame ? a.T : 1;
    uint8_t *mark = a.inv_mark + ((long long)b * G + g) * N;
    long long *r2d = a.out_r2d ? a.out_r2d + ((long long)b * a.T + tl) * N * 2 : nullptr;
    const float Wf = (float)a.W, Hf = (float)a.H;

    const int y = ty0 + (threadIdx.x >> 4), x = tx0 + (threadIdx.x & 15) * 4;
    float zmax = -INFINITY;
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
    if (y < a.H && x < a.W) {
        float d[4];
        bool m[4];
        load4(a, in_base, x, y, d, m);
        uint2 *pj = a.proj + ((long long)b * a.T + tl) * N + (long long)y * a.W + x;
        unsigned pk[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x + k >= a.W) break;
            const Proj p = project(cam, x + k, y, d[k], m[k], Wf, Hf);
            zmax = fmaxf(zmax, p.z);   // :105 max runs over valid and invalid points alike
            pk[2 * k] = __float_as_uint(p.z);
            pk[2 * k + 1] = (unsigned)p.x0 | ((unsigned)p.y0 << 13) | ((unsigned)(p.x1 != p.x0) << 26) |
                            ((unsigned)(p.y1 != p.y0) << 27) | ((unsigned)p.valid << 28);
            if ((a.W & 3) != 0) pj[k] = make_uint2(pk[2 * k], pk[2 * k + 1]);
            if (r2d) {
                const long long n = (long long)y * a.W + x + k;
                r2d[n * 2] = p.x0;      // :147 floor/floor corner after the clamp
                r2d[n * 2 + 1] = p.y0;
            }
            if (p.valid) {
                bx0 = min(bx0, p.x0); by0 = min(by0, p.y0);
                bx1 = max(bx1, p.x1); by1 = max(by1, p.y1);
            } else {
                // every invalid point carries depth max+1 and payload 0: marking its bins is enough
                mark[(long long)p.y0 * a.W + p.x0] = 1;
                mark[(long long)p.y1 * a.W + p.x0] = 1;
                mark[(long long)p.y0 * a.W + p.x1] = 1;
                mark[(long long)p.y1 * a.W + p.x1] = 1;
            }
        }
        if ((a.W & 3) == 0) {   // the raster pass re-reads these instead of re-projecting (coalesced 32 B per thread)
            reinterpret_cast<uint4 *>(pj)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            reinterpret_cast<uint4 *>(pj)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
    }
    // block reductions: max z, bounding box
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        zmax = fmaxf(zmax, __shfl_xor(zmax, o));
        bx0 = min(bx0, __shfl_xor(bx0, o)); by0 = min(by0, __shfl_xor(by0, o));
        bx1 = max(bx1, __shfl_xor(bx1, o)); by1 = max(by1, __shfl_xor(by1, o));
    }
    __shared__ float zred[kThreads / 64];
    __shared__ int bred[kThreads / 64][4];
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        zred[w] = zmax;
        bred[w][0] = bx0; bred[w][1] = by0; bred[w][2] = bx1; bred[w][3] = by1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kThreads / 64; ++w) {
            zmax = fmaxf(zmax, zred[w]);
            bx0 = min(bx0, bred[w][0]); by0 = min(by0, bred[w][1]);
            bx1 = max(bx1, bred[w][2]); by1 = max(by1, bred[w][3]);
        }
        const long long ntile = (long long)a.stx * a.sty;
        // :105 the sentinel is max(z)+1 over the whole predict call (one frame's points in per_frame mode): one
        // order-independent atomicMax per source tile instead of every raster workgroup re-reducing all partials
        atomicMax(&a.zmax_part[(b * a.zgroups_per_sample + (a.per_frame ? tl : 0)) * kZSlots + ((tile + b) & (kZSlots - 1))], float_to_ordered(zmax));
        a.bbox[((long long)b * a.T + tl) * ntile + tile] = make_int4(bx0, by0, bx1, by1);
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void raster_kernel(SplatArgs a) {
    __shared__ unsigned long long zb[kDstTH * kDstTW];   // 16 KB
    __shared__ unsigned short list[kScan * kScanBatches];
    __shared__ int list_n;
    __shared__ float sentinel_s;

    const int dtile = blockIdx.x, g = blockIdx.y, b = blockIdx.z;
    const int G = a.per_frame ? a.T : 1, Tg = a.per_frame ? 1 : a.T;
    const int dy0 = (dtile / a.dtx) * kDstTH, dx0 = (dtile % a.dtx) * kDstTW;
    const int dy1 = min(dy0 + kDstTH, a.H) - 1, dx1 = min(dx0 + kDstTW, a.W) - 1;
    const long long N = (long long)a.H * a.W;
    const long long P = (long long)Tg * N;
    const int ntile = a.stx * a.sty;

    RPROBE(0);
    int n_hits_total = 0;
    for (int i = threadIdx.x; i < kDstTH * kDstTW; i += kThreads) zb[i] = kEmpty;

    // sentinel = max(z over the whole predict call) + 1 (:105); one frame's points in per_frame mode
    if (threadIdx.x < 64) {
        unsigned m = a.zmax_part[(b * a.zgroups_per_sample + (a.per_frame ? g : 0)) * kZSlots + threadIdx.x];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) sentinel_s = __fadd_rn(ordered_to_float(m), 1.0f);
    }

    // ---- rasterise: every source tile whose valid-point bounding box touches this destination tile.  Hits are
    //      collected first (one barrier pair per 256 boxes) and then consumed four at a time: their 32-B projection
    //      records are all in flight before the first goes through the LDS atomics - the loop was
    //      a chain of ~2 us load latencies, one per hit (tools/probe_splat.py).
    auto splat4 = [&](const unsigned (&pk)[8], int x, int y, unsigned long long ebase) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned f = pk[2 * k + 1];
            if (!((f >> 28) & 1u)) continue;   // invalid points were handled by the byte marks
            const int px0 = (int)(f & 8191u), py0 = (int)((f >> 13) & 8191u);
            const int px1 = px0 + (int)((f >> 26) & 1u), py1 = py0 + (int)((f >> 27) & 1u);
            // replicas r = 0:(x0,y0) 1:(x0,y1) 2:(x1,y0) 3:(x1,y1); e = r*P + t*N + n  (:112).  A replica
            // on the bin of a lower replica of the same point can never win the tie-break: skip it.
            const unsigned long long e0 = ebase + (unsigned long long)((long long)y * a.W + x + k);
            const unsigned long long khi = (unsigned long long)pk[2 * k] << 32;
            const bool in_x0 = px0 >= dx0 && px0 <= dx1, in_x1 = px1 >= dx0 && px1 <= dx1 && px1 != px0;
            const bool in_y0 = py0 >= dy0 && py0 <= dy1, in_y1 = py1 >= dy0 && py1 <= dy1 && py1 != py0;
            if (in_x0 && in_y0) atomicMin(&zb[(py0 - dy0) * kDstTW + (px0 - dx0)], khi | e0);
            if (in_x0 && in_y1) atomicMin(&zb[(py1 - dy0) * kDstTW + (px0 - dx0)], khi | (e0 + (unsigned long long)P));
            if (in_x1 && in_y0) atomicMin(&zb[(py0 - dy0) * kDstTW + (px1 - dx0)], khi | (e0 + 2ull * P));
            if (in_x1 && in_y1) atomicMin(&zb[(py1 - dy0) * kDstTW + (px1 - dx0)], khi | (e0 + 3ull * P));
        }
    };
    auto load4 = [&](const uint2 *pbase, int st, unsigned (&pk)[8], int &x, int &y) {
        y = (st / a.stx) * kSrcTH + (threadIdx.x >> 4);
        x = (st % a.stx) * kSrcTW + (threadIdx.x & 15) * 4;
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = 0u;      // valid bit clear: nothing to splat
        if (st < 0 || y >= a.H || x >= a.W) return;
        const uint2 *pj = pbase + (long long)y * a.W + x;
        if ((a.W & 3) == 0) {
            const uint4 q0 = reinterpret_cast<const uint4 *>(pj)[0], q1 = reinterpret_cast<const uint4 *>(pj)[1];
            pk[0] = q0.x; pk[1] = q0.y; pk[2] = q0.z; pk[3] = q0.w;
            pk[4] = q1.x; pk[5] = q1.y; pk[6] = q1.z; pk[7] = q1.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint2 q = x + k < a.W ? pj[k] : make_uint2(0u, 0u);
                pk[2 * k] = q.x; pk[2 * k + 1] = q.y;
            }
        }
    };
    for (int tt = 0; tt < Tg; ++tt) {
        const int tl = a.per_frame ? g : tt;     // local frame index
        const int4 *boxes = a.bbox + ((long long)b * a.T + tl) * ntile;
        const uint2 *pbase = a.proj + ((long long)b * a.T + tl) * N;
        const unsigned long long ebase = (unsigned long long)tt * N;
        for (int s0 = 0; s0 < ntile; s0 += kScan * kScanBatches) {
            __syncthreads();
            if (threadIdx.x == 0) list_n = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kScanBatches; ++j) {
                const int s = s0 + j * kScan + threadIdx.x;
                if (s < ntile) {
                    const int4 bb = boxes[s];
                    if (bb.x <= dx1 && bb.z >= dx0 && bb.y <= dy1 && bb.w >= dy0) list[atomicAdd(&list_n, 1)] = (unsigned short)(s - s0);
                }
            }
            __syncthreads();
            const int n_hit = list_n;
            n_hits_total += n_hit;
            RPROBE(4);
            for (int li = 0; li < n_hit; li += 4) {
                unsigned pk[4][8];
                int xs[4], ys[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) load4(pbase, li + j < n_hit ? s0 + list[li + j] : -1, pk[j], xs[j], ys[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) splat4(pk[j], xs[j], ys[j], ebase);
            }
        }
    }
    __syncthreads();
    RPROBE(1);
#if PF_PROBE
    if (threadIdx.x == 0 && blockIdx.x == 100 && blockIdx.y == 1 && blockIdx.z == 0 && a.probe) a.probe[3] = n_hits_total;
#endif

    // ---- resolve this tile's pixels (:120-139): 4 consecutive pixels per lane, all loads issued before any is used
    const float sentinel = sentinel_s;
    const uint8_t *mark = a.inv_mark + ((long long)b * G + g) * N;
    const long long out_base = ((long long)b * G + g) * N;
    const long long seg_base = ((long long)b * a.T_total + a.t_first + (a.per_frame ? g : 0)) * N;
    const int C = a.C;
    const unsigned Pu = (unsigned)P;
#pragma unroll
    for (int it = 0; it < kDstTH * kDstTW / 4 / kThreads; ++it) {
        const int i4 = it * kThreads + threadIdx.x;
        const int y = dy0 + i4 / (kDstTW / 4), x = dx0 + (i4 % (kDstTW / 4)) * 4;
        if (y >= a.H || x >= a.W) continue;
        const long long n0 = (long long)y * a.W + x;
        unsigned long long key[4];
        long long src[4];
        uint8_t mk[4], sg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            key[k] = zb[i4 * 4 + k];
            unsigned e = (unsigned)key[k];               // e = r*P + t*N + n with r < 4: strip the corner replica
            e -= e >= 2u * Pu ? 2u * Pu : 0u;
            e -= e >= Pu ? Pu : 0u;
            src[k] = key[k] != kEmpty ? seg_base + (long long)e : seg_base;     // always a readable address
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = x + k < a.W;
            mk[k] = in ? mark[n0 + k] : (uint8_t)0;
            sg[k] = C == 1 ? a.seg[src[k]] : (uint8_t)0;
        }
        float dep[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // empty bin: won by an invalid point (:105,:133) -> max+1, or never touched (:136-138) -> -1
            dep[k] = key[k] != kEmpty ? __uint_as_float((unsigned)(key[k] >> 32)) : (mk[k] ? sentinel : -1.0f);
            if (key[k] == kEmpty) sg[k] = 0;
        }
        if ((a.W & 3) == 0) {
            *reinterpret_cast<float4 *>(a.out_depth + out_base + n0) = make_float4(dep[0], dep[1], dep[2], dep[3]);
            if (C == 1) *reinterpret_cast<uchar4 *>(a.out_seg + out_base + n0) = make_uchar4(sg[0], sg[1], sg[2], sg[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < a.W) {
                    a.out_depth[out_base + n0 + k] = dep[k];
                    if (C == 1) a.out_seg[out_base + n0 + k] = sg[k];
                }
        }
        if (C != 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < a.W)
                    for (int c = 0; c < C; ++c)
                        a.out_seg[(out_base + n0 + k) * C + c] = key[k] != kEmpty ? a.seg[src[k] * C + c] : (uint8_t)0;
        }
    }
    RPROBE(2);
}

struct SplatLayout {
    size_t bbox_off, zmax_off, mark_off, mark_bytes, proj_off, total;
    int stx, sty, dtx, dty;
};

static SplatLayout splat_layout(int B, int T, int H, int W, int per_frame) {
    SplatLayout L;
    L.stx = (W + kSrcTW - 1) / kSrcTW; L.sty = (H + kSrcTH - 1) / kSrcTH;
    L.dtx = (W + kDstTW - 1) / kDstTW; L.dty = (H + kDstTH - 1) / kDstTH;
    const size_t ntile = (size_t)L.stx * L.sty, N = (size_t)H * W;
    L.bbox_off = 0;
    L.zmax_off = align_up(L.bbox_off + (size_t)B * T * ntile * sizeof(int4), 256);
    L.mark_off = align_up(L.zmax_off + (size_t)B * T * kZSlots * sizeof(unsigned), 256);   // zmax slots sit right before the marks: one memset
    L.mark_bytes = (size_t)B * (per_frame ? T : 1) * N;
    L.proj_off = align_up(L.mark_off + L.mark_bytes, 256);
    L.total = align_up(L.proj_off + (size_t)B * T * N * sizeof(uint2), 256);
    return L;
}

}  // namespace pf

extern "C" int pf_warp_splat_workspace(int B, int T, int H, int W, int per_frame, size_t *bytes) {
    if (!bytes || B <= 0 || T <= 0 || H <= 0 || W <= 0)
        return pf::fail(PF_EINVAL, "pf_warp_splat_workspace: bad dims B=%d T=%d H=%d W=%d", B, T, H, W);
    if (4ll * T * H * W >= (1ll << 32))
        return pf::fail(PF_EUNSUPPORTED, "pf_warp_splat: 4*T*H*W must be < 2^32 (element index packs in 32 bits)");
    if (H > 8192 || W > 8192)
        return pf::fail(PF_EUNSUPPORTED, "pf_warp_splat: H and W must be <= 8192 (13-bit bin coordinates)");
    *bytes = pf::splat_layout(B, T, H, W, per_frame & PF_SPLAT_PER_FRAME).total;
    return PF_OK;
}

extern "C" int pf_warp_splat(const float *depth, const uint8_t *depth_mask, const uint8_t *seg,
                             int seg_channels, const float *Kinv, const float *E, const float *T_tgt,
                             const float *Einv, const float *K, int B, int T_total, int t_first, int T,
                             int H, int W, int per_frame, uint8_t *out_seg, float *out_depth,
                             int64_t *out_result2d, void *ws, size_t ws_bytes, void *stream) {
    if (!depth || !depth_mask || !seg || !Kinv || !E || !T_tgt || !Einv || !K || !out_seg || !out_depth || !ws)
        return pf::fail(PF_EINVAL, "pf_warp_splat: null pointer argument");
    if (seg_channels != 1 && seg_channels != 3)
        return pf::fail(PF_EINVAL, "pf_warp_splat: seg_channels must be 1 or 3, got %d", seg_channels);
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || t_first < 0 || t_first + T > T_total)
        return pf::fail(PF_EINVAL, "pf_warp_splat: bad dims B=%d T_total=%d t_first=%d T=%d H=%d W=%d", B,
                        T_total, t_first, T, H, W);
    size_t need = 0;
    int rc = pf_warp_splat_workspace(B, T, H, W, per_frame, &need);
    if (rc) return rc;
    if (ws_bytes < need)
        return pf::fail(PF_EWORKSPACE, "pf_warp_splat: workspace %zu B < required %zu B", ws_bytes, need);

    const pf::SplatLayout L = pf::splat_layout(B, T, H, W, per_frame & PF_SPLAT_PER_FRAME);
    pf::SplatArgs a;
    a.depth = depth; a.mask = depth_mask; a.seg = seg;
    a.Kinv = Kinv; a.E = E; a.Tt = T_tgt; a.Einv = Einv; a.K = K;
    a.bbox = (int4 *)((char *)ws + L.bbox_off);
    a.zmax_part = (unsigned *)((char *)ws + L.zmax_off);
    a.inv_mark = (uint8_t *)ws + L.mark_off;
    a.proj = (uint2 *)((char *)ws + L.proj_off);
    a.out_seg = out_seg; a.out_depth = out_depth; a.out_r2d = (long long *)out_result2d;
    a.B = B; a.T_total = T_total; a.t_first = t_first; a.T = T; a.H = H; a.W = W; a.C = seg_channels;
    a.per_frame = (per_frame & PF_SPLAT_PER_FRAME) ? 1 : 0;
    a.zgroups_per_sample = (per_frame & PF_SPLAT_PER_SAMPLE_SENTINEL) ? (a.per_frame ? T : 1) : 0;
    a.stx = L.stx; a.sty = L.sty; a.dtx = L.dtx; a.dty = L.dty;
    a.probe = nullptr;
#if PF_PROBE
    a.probe = getenv("PF_PROBE") ? pf::probe_buffer() : nullptr;
#endif
    hipStream_t s = (hipStream_t)stream;
    const int G = a.per_frame ? T : 1;

    // algorithmic bytes: source side depth 4 + mask 1 B/px; destination side seg 1 in, seg 1 + depth 4 out
    const double src_px = (double)B * T * H * W, dst_px = (double)B * G * H * W;
    {
        pf::ProfScope ps(s, "inv_mark_memset", 0, (double)L.mark_bytes);
        // zmax slots (ordered-u32 encoding: 0 is below every float) + invalid-point marks, contiguous
        PF_HIP_CHECK(hipMemsetAsync(a.zmax_part, 0, (L.mark_off - L.zmax_off) + L.mark_bytes, s));
    }
    {
        pf::ProfScope ps(s, "pf::bin_kernel(pf::SplatArgs)", 0, src_px * (5.0 + (out_result2d ? 16.0 : 0.0)));
        hipLaunchKernelGGL(pf::bin_kernel, dim3(L.stx * L.sty, T, B), dim3(pf::kThreads), 0, s, a);
        PF_LAUNCH_CHECK("bin_kernel");
    }
    {
        pf::ProfScope ps(s, "pf::raster_kernel(pf::SplatArgs)", 0, dst_px * (2.0 * seg_channels + 4.0));
        hipLaunchKernelGGL(pf::raster_kernel, dim3(L.dtx * L.dty, G, B), dim3(pf::kThreads), 0, s, a);
        PF_LAUNCH_CHECK("raster_kernel");
    }
    return PF_OK;
}
