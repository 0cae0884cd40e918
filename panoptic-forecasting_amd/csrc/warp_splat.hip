// Warp + z-buffered splat for gfx950 (MI355X).
//
// Replaces reference panoptic_forecasting/models/pc_transform/pc_transform_model.py:41-150:
// unproject (:41-59), camera->vehicle (:63), ego warp (:68), vehicle->camera + projection (:71-78),
// validity (:83-89), sentinel (:105), 4-corner bins (:106-117), torch_scatter.scatter_min (:118-119)
// and the winner gather (:120-139).
//
// Two kernels, both HBM/L2-bound integer+fp32 work (no MFMA):
//   project_scatter : 1 thread per source pixel.  Exact-order fp32 chain (this file is compiled with
//                     -ffp-contract=off; every product and sum is rounded separately, divides are IEEE)
//                     -> (u',v',z), validity, up to 4 de-duplicated bins, one 64-bit atomicMin per bin
//                     on a packed key  [ z bits | element index e ]  so "min depth, ties -> lowest e"
//                     is a single unsigned compare.  Block-level max(z) is written as a partial
//                     (no same-address atomics).
//   resolve         : 1 thread per destination pixel decodes the winning key into (seg, depth).
//
// Key encoding (valid points have z > 0 so the raw fp32 bits are monotone as unsigned):
//   valid   : hi = bits(z)           lo = e
//   invalid : hi = 0xFFFFFFF0        lo = e     (sorts after every valid z, before EMPTY; the reference
//                                                 gives these depth max+1 which is > every valid z)
//   EMPTY   : 0xFFFFFFFFFFFFFFFF                 (memset 0xFF)
#include "pf_common.h"
#include "pf_prof.h"

namespace pf {

constexpr unsigned kInvalidHi = 0xFFFFFFF0u;
constexpr unsigned long long kEmpty = ~0ull;
constexpr int kSplatThreads = 256;

struct SplatArgs {
    const float *depth;
    const uint8_t *mask;
    const uint8_t *seg;
    const float *Kinv, *E, *Tt, *Einv, *K;
    unsigned long long *zbuf;  // [B*G][N]
    unsigned *zmax_part;       // [T][B][chunks] order-preserving u32 of float
    uint8_t *out_seg;
    float *out_depth;
    long long *out_r2d;
    int B, T_total, t_first, T, H, W, C, per_frame, chunks;
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

__global__ __launch_bounds__(kSplatThreads) void project_scatter_kernel(SplatArgs a) {
    const int chunk = blockIdx.x, tl = blockIdx.y, b = blockIdx.z;
    const int t = a.t_first + tl;
    const long long N = (long long)a.H * a.W;
    const long long P = a.per_frame ? N : (long long)a.T * N;
    const long long ebase = a.per_frame ? 0 : (long long)tl * N;

    // wave-uniform matrices -> scalar registers
    float Kinv[9], E[16], Tm[16], Einv[16], K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { Kinv[i] = a.Kinv[b * 9 + i]; K[i] = a.K[b * 9 + i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        E[i] = a.E[b * 16 + i];
        Einv[i] = a.Einv[b * 16 + i];
        Tm[i] = a.Tt[((long long)b * a.T_total + t) * 16 + i];
    }
    const long long in_base = ((long long)b * a.T_total + t) * N;
    unsigned long long *zb = a.zbuf + ((long long)b * (a.per_frame ? a.T : 1) + (a.per_frame ? tl : 0)) * N;
    long long *r2d = a.out_r2d ? a.out_r2d + ((long long)b * a.T + tl) * N * 2 : nullptr;

    const long long per_chunk = (N + a.chunks - 1) / a.chunks;
    const long long n0 = (long long)chunk * per_chunk;
    const long long n1 = n0 + per_chunk < N ? n0 + per_chunk : N;
    const float Wf = (float)a.W, Hf = (float)a.H;
    float zmax = -INFINITY;

    for (long long n = n0 + threadIdx.x; n < n1; n += kSplatThreads) {
        const int y = (int)(n / a.W), x = (int)(n - (long long)y * a.W);
        const float d = a.depth[in_base + n];
        const bool m = a.mask[in_base + n] != 0;
        const float u = (float)x, v = (float)y;
        // :54  ray = Kinv · (u, v, 1)
        const float r0 = dot3(Kinv + 0, u, v, 1.0f), r1 = dot3(Kinv + 3, u, v, 1.0f),
                    r2 = dot3(Kinv + 6, u, v, 1.0f);
        // :55-59  camera point (homogeneous)
        const float c0 = __fmul_rn(r0, d), c1 = __fmul_rn(r1, d), c2 = __fmul_rn(r2, d);
        // :63  vehicle frame
        const float v0 = dot4(E + 0, c0, c1, c2, 1.0f), v1 = dot4(E + 4, c0, c1, c2, 1.0f),
                    v2 = dot4(E + 8, c0, c1, c2, 1.0f), v3 = dot4(E + 12, c0, c1, c2, 1.0f);
        // :68  target vehicle frame
        const float w0 = dot4(Tm + 0, v0, v1, v2, v3), w1 = dot4(Tm + 4, v0, v1, v2, v3),
                    w2 = dot4(Tm + 8, v0, v1, v2, v3), w3 = dot4(Tm + 12, v0, v1, v2, v3);
        // :71-72  back to the camera, homogeneous divide
        const float e0 = dot4(Einv + 0, w0, w1, w2, w3), e1 = dot4(Einv + 4, w0, w1, w2, w3),
                    e2 = dot4(Einv + 8, w0, w1, w2, w3), e3 = dot4(Einv + 12, w0, w1, w2, w3);
        const float px = __fdiv_rn(e0, e3), py = __fdiv_rn(e1, e3), z = __fdiv_rn(e2, e3);
        // :74-78  projection
        const float q0 = dot3(K + 0, px, py, z), q1 = dot3(K + 3, px, py, z), q2 = dot3(K + 6, px, py, z);
        const float uu = __fdiv_rn(q0, q2), vv = __fdiv_rn(q1, q2);

        zmax = fmaxf(zmax, z);  // :105 max runs over valid and invalid points alike
        const bool inb = (uu >= 0.0f) && (uu < Wf) && (vv >= 0.0f) && (vv < Hf);  // :83-86
        const bool valid = m && (z > 0.0f) && inb;                                  // :87-89
        const unsigned hi = valid ? __float_as_uint(z) : kInvalidHi;

        // :106-114 floor/ceil then clamp (clamping in float first keeps the int conversion in range)
        const float fu = floorf(uu), cu = ceilf(uu), fv = floorf(vv), cv = ceilf(vv);
        const int x0 = (int)fminf(fmaxf(fu, 0.0f), Wf - 1.0f), x1 = (int)fminf(fmaxf(cu, 0.0f), Wf - 1.0f);
        const int y0 = (int)fminf(fmaxf(fv, 0.0f), Hf - 1.0f), y1 = (int)fminf(fmaxf(cv, 0.0f), Hf - 1.0f);
        if (r2d) {
            r2d[n * 2] = x0;
            r2d[n * 2 + 1] = y0;
        }
        // replicas r = 0:(x0,y0) 1:(x0,y1) 2:(x1,y0) 3:(x1,y1); e = r*P + t*N + n  (:112)
        // a replica landing on the bin of a lower replica of the same point can never win: skip it
        const unsigned long long e0k = (unsigned long long)(ebase + n);
        const unsigned long long khi = (unsigned long long)hi << 32;
        const long long b00 = (long long)y0 * a.W + x0;
        atomicMin(&zb[b00], khi | e0k);
        if (y1 != y0) atomicMin(&zb[(long long)y1 * a.W + x0], khi | (e0k + (unsigned long long)P));
        if (x1 != x0) {
            atomicMin(&zb[(long long)y0 * a.W + x1], khi | (e0k + 2ull * P));
            if (y1 != y0) atomicMin(&zb[(long long)y1 * a.W + x1], khi | (e0k + 3ull * P));
        }
    }

    // block max(z) -> one partial per block (plain store, no same-address atomics)
    __shared__ float red[kSplatThreads / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = zmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
#pragma unroll
        for (int i = 1; i < kSplatThreads / 64; ++i) m = fmaxf(m, red[i]);
        a.zmax_part[((long long)tl * a.B + b) * a.chunks + chunk] = float_to_ordered(m);
    }
}

__global__ __launch_bounds__(kSplatThreads) void resolve_kernel(SplatArgs a) {
    const int tl = blockIdx.y, b = blockIdx.z;  // tl = 0 unless per_frame
    const long long N = (long long)a.H * a.W;
    const long long P = a.per_frame ? N : (long long)a.T * N;
    const int G = a.per_frame ? a.T : 1;

    // sentinel = max(z over the whole predict call) + 1  (:105); per frame in per_frame mode
    __shared__ unsigned red[kSplatThreads / 64];
    __shared__ float sentinel_s;
    {
        const unsigned *part = a.zmax_part + (a.per_frame ? (long long)tl * a.B * a.chunks : 0);
        const int cnt = (a.per_frame ? 1 : a.T) * a.B * a.chunks;
        unsigned m = 0;
        for (int i = threadIdx.x; i < cnt; i += kSplatThreads) m = max(m, part[i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned mm = red[0];
#pragma unroll
            for (int i = 1; i < kSplatThreads / 64; ++i) mm = max(mm, red[i]);
            sentinel_s = __fadd_rn(ordered_to_float(mm), 1.0f);
        }
        __syncthreads();
    }
    const float sentinel = sentinel_s;

    const unsigned long long *zb = a.zbuf + ((long long)b * G + tl) * N;
    const long long out_base = ((long long)b * G + tl) * N;
    const long long seg_base = ((long long)b * a.T_total + a.t_first + (a.per_frame ? tl : 0)) * N;
    const int C = a.C;
    for (long long n = (long long)blockIdx.x * kSplatThreads + threadIdx.x; n < N;
         n += (long long)gridDim.x * kSplatThreads) {
        const unsigned long long key = zb[n];
        float dep = -1.0f;  // :136-138
        long long src = -1;
        if (key != kEmpty) {
            const unsigned hi = (unsigned)(key >> 32);
            const long long p = (long long)(key & 0xFFFFFFFFull) % P;  // e -> point index t*N + n
            if (hi == kInvalidHi) {
                dep = sentinel;  // won by an invalid point: seg 0 (:133), depth max+1 (:105)
            } else {
                dep = __uint_as_float(hi);
                src = seg_base + p;  // frames are contiguous: t*N + n indexes [t_first.., H, W]
            }
        }
        a.out_depth[out_base + n] = dep;
        if (C == 1) {
            a.out_seg[out_base + n] = src >= 0 ? a.seg[src] : (uint8_t)0;
        } else {
            for (int c = 0; c < C; ++c)
                a.out_seg[(out_base + n) * C + c] = src >= 0 ? a.seg[src * C + c] : (uint8_t)0;
        }
    }
}

static int splat_chunks(long long N) {
    long long c = (N + 1023) / 1024;
    return (int)(c < 1 ? 1 : (c > 512 ? 512 : c));
}

}  // namespace pf

extern "C" int pf_warp_splat_workspace(int B, int T, int H, int W, int per_frame, size_t *bytes) {
    if (!bytes || B <= 0 || T <= 0 || H <= 0 || W <= 0)
        return pf::fail(PF_EINVAL, "pf_warp_splat_workspace: bad dims B=%d T=%d H=%d W=%d", B, T, H, W);
    const long long N = (long long)H * W;
    if (4ll * T * N >= (1ll << 32))
        return pf::fail(PF_EUNSUPPORTED, "pf_warp_splat: 4*T*H*W must be < 2^32 (element index packs in 32 bits)");
    const size_t zbuf = (size_t)B * (per_frame ? T : 1) * N * sizeof(unsigned long long);
    const size_t part = (size_t)T * B * pf::splat_chunks(N) * sizeof(unsigned);
    *bytes = pf::align_up(zbuf, 256) + pf::align_up(part, 256);
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

    const long long N = (long long)H * W;
    pf::SplatArgs a;
    a.depth = depth; a.mask = depth_mask; a.seg = seg;
    a.Kinv = Kinv; a.E = E; a.Tt = T_tgt; a.Einv = Einv; a.K = K;
    const size_t zbuf_bytes = (size_t)B * (per_frame ? T : 1) * N * sizeof(unsigned long long);
    a.zbuf = (unsigned long long *)ws;
    a.zmax_part = (unsigned *)((char *)ws + pf::align_up(zbuf_bytes, 256));
    a.out_seg = out_seg; a.out_depth = out_depth; a.out_r2d = (long long *)out_result2d;
    a.B = B; a.T_total = T_total; a.t_first = t_first; a.T = T; a.H = H; a.W = W; a.C = seg_channels;
    a.per_frame = per_frame ? 1 : 0;
    a.chunks = pf::splat_chunks(N);
    hipStream_t s = (hipStream_t)stream;

    // algorithmic bytes: read depth 4 + mask 1 per source pixel; resolve: seg 1 in, seg 1 + depth 4 out
    const double src_px = (double)B * T * N, dst_px = (double)B * (per_frame ? T : 1) * N;
    {
        pf::ProfScope ps(s, "zbuf_memset", 0, (double)zbuf_bytes);
        PF_HIP_CHECK(hipMemsetAsync(a.zbuf, 0xFF, zbuf_bytes, s));
    }
    {
        pf::ProfScope ps(s, "pf::project_scatter_kernel(pf::SplatArgs)", 0,
                         src_px * (5.0 + (out_result2d ? 16.0 : 0.0)));
        hipLaunchKernelGGL(pf::project_scatter_kernel, dim3(a.chunks, T, B), dim3(pf::kSplatThreads), 0, s, a);
        PF_LAUNCH_CHECK("project_scatter_kernel");
    }
    long long rb = (N + pf::kSplatThreads - 1) / pf::kSplatThreads;
    if (rb > 1024) rb = 1024;
    {
        pf::ProfScope ps(s, "pf::resolve_kernel(pf::SplatArgs)", 0, dst_px * (2.0 * seg_channels + 4.0));
        hipLaunchKernelGGL(pf::resolve_kernel, dim3((unsigned)rb, per_frame ? T : 1, B), dim3(pf::kSplatThreads),
                           0, s, a);
        PF_LAUNCH_CHECK("resolve_kernel");
    }
    return PF_OK;
}
