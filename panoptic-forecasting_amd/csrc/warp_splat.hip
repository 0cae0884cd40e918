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
//                  -ffp-contract=off; every product/sum rounded separately, IEEE divides) — four pixels per lane in
//                  lockstep on the packed fp32 pipe with exact shortcuts for affine cameras (project4_fast), the full (project4_full)
//                  scalar chain otherwise — stored as 8 B per point {bits(z), packed bins}; block max(z) partial; the
//                  bounding box of the destination bins its VALID points reach; a byte mark per bin reached by an
//                  INVALID point (plain stores of the constant 1 - no atomics needed); optional result2d.  The tile then
//                  appends its id to the list of every destination tile its box touches (one global atomicAdd per pair).
//   raster_kernel  one workgroup (512 threads = two groups of 256, each taking its own source tiles) per 32x128 DESTINATION
//                  tile: reads its list (kFlight source tiles' 32-B-per-lane records in flight per group at a time; a list that overflowed kListCap falls back to testing every box of the frame), and
//                  resolves "min depth, ties -> lowest element index" with 64-bit ds_min on a packed key in a 32 KB LDS
//                  z-buffer it alone owns; then writes seg/depth for its pixels.  The z-buffer never exists in HBM.
//                  List order is arbitrary (atomic appends); min() does not care: the output is deterministic.
//
// Measured bounds and the variants that were tried and dropped (re-projecting in the raster pass instead of the 8 B/point
// round trip, software-pipelined multi-tile workgroups): profiles/r02_experiments.md.
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
// raster workgroup shape (A/B builds: -DPF_RASTER_THREADS=256 -DPF_RASTER_FLIGHT=8 -DPF_RASTER_MINWAVES=1 is the round-2 start).
// The kernel waits on dependent memory round trips (list -> records -> LDS -> gather -> store) at a workgroup count per CU
// that its 32 KB z-buffer fixes at 4: waves hide that better than loads in flight per wave.  Same box, 16 frames:
// 256 threads x 8 tiles in flight (4 waves per SIMD, 109 registers) 557 us; 512 x 4 at 4 waves 652; 512 x 4 at 5 waves (66
// registers) 517; 512 x 2 at 8 waves (64 registers) 460-478; 512 x 3 at 8 waves 470; 512 x 1 at 8 waves 485.
// invalid-point mark stores of bin_kernel (same-box A/B, 32 frames, profiles/r06_experiments.md): 0 = four unconditional byte stores
// (rounds 1-5): 1138 / 1119 / 1138 us; 1 = only the distinct bins: 1088 / 1093 / 1095; 2 (shipped) = distinct bins, the two bins of a
// row as one two-byte store: 1064 / 1071 / 1067 (-6 %).  Bit-exact either way (tests/test_gpu_warp_splat.py, test_gpu_pipeline.py)
#ifndef PF_MARK_VARIANT
#define PF_MARK_VARIANT 2
#endif
#ifndef PF_RASTER_THREADS
#define PF_RASTER_THREADS 512
#endif
#ifndef PF_RASTER_FLIGHT
#define PF_RASTER_FLIGHT 2
#endif
#ifndef PF_RASTER_MINWAVES
#define PF_RASTER_MINWAVES 8   // __launch_bounds__' second argument on HIP: waves per SIMD -> 64 registers
#endif
constexpr int kRThreads = PF_RASTER_THREADS;   // raster workgroup: kRHalves groups of 256 threads, each group takes its own source tile
constexpr int kRHalves = kRThreads / 256;
constexpr int kSrcTH = 16, kSrcTW = 64;   // source tile (pixels); 256 threads x 4 consecutive pixels
constexpr int kDstTH = 32, kDstTW = 128;  // destination tile owned by one raster workgroup (32 KB LDS)
constexpr int kScan = 256;                // bounding boxes tested per thread-pass (overflow path only)
constexpr int kScanBatches = 8;           // (the list holds kScan * kScanBatches = 2048 tile ids)
constexpr int kListCap = 64;              // source-tile ids per destination-tile list written by bin_kernel (typical fill 6-12)
constexpr int kFlight = PF_RASTER_FLIGHT;  // source tiles whose projections a group of 256 raster threads keeps in flight
constexpr int kIdFrameShift = 20;         // list entry = (frame inside the z-buffer group) << 20 | source tile  (in the raster workgroup's LDS
                                          // copy: | source tile row << 10 | column; both < 1024 as bins have 13 bits)
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
    unsigned *count;      // [B][G][dst tiles]  fill of the destination tile's list (zeroed per call; > kListCap = overflowed)
    unsigned *lists;      // [B][G][dst tiles][kListCap]  source tiles whose valid points can reach the destination tile
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

// the same (uniform) pointer as a value the optimiser cannot trace back: loads through it are not merged with earlier
// loads of the same addresses, and stay scalar loads
typedef const float __attribute__((address_space(1))) *GlobalF;
__device__ __forceinline__ GlobalF relaunder(const float *p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (GlobalF)(((unsigned long long)hi << 32) | lo);
}

struct Proj {
    float z;
    int x0, y0, x1, y1;  // clamped floor/ceil bins
    bool valid;
};

// :106-114 `.floor().long()` / `.ceil().long()` then clamp(0, hi).  Clamping in float first keeps the conversion in range;
// a value the reference's float -> int64 cast cannot represent (NaN, +-inf, |f| >= 2^63: x86 returns INT64_MIN, which the
// clamp turns into 0) gives 0 here too, so that bins of points with non-finite projections (invalid, but they still mark
// bins and appear in result2d) agree with the reference run on x86.  (v_med3_f32 with a NaN operand returns the minimum of
// the other two: 0.)
__device__ __forceinline__ int to_bin(float f, float hi) {
    const float c = __builtin_amdgcn_fmed3f(f, 0.0f, hi);
    return (f < 9223372036854775808.0f) ? (int)c : 0;
}
// both bins of one coordinate.  The ceil bin is the floor bin or the next one: clamp(ceil(u)) != clamp(floor(u)) exactly
// when 0 < u < hi and u is not an integer (u <= 0: both clamp to 0; u >= hi: both clamp to hi, or to 0 beyond the int64
// range; NaN: both 0) - one floor and three compares instead of floor, ceil and two clamped conversions.
__device__ __forceinline__ void bins(float u, float hi, int &b0, int &b1) {
    const float fl = floorf(u);
    b0 = to_bin(fl, hi);
    b1 = b0 + (int)((u > 0.0f) & (u < hi) & (u != fl));
}

// ---- the same chain for four consecutive pixels of a row on the packed fp32 pipe, for `affine` cameras.
// v_pk_mul_f32 / v_pk_add_f32 perform two separately rounded IEEE operations per lane per issue: same results as the
// scalar chain at half the issues (bin_kernel is bound by the vector ALU: 256 instructions per point in the scalar form,
// profiles/r02_a_pmc.json).  Every product and sum below is one IEEE operation per element in the order of dot3/dot4
// (this file is built with -ffp-contract=off).  What is left out is exact:  m*1.0f == m;  with a last row (0,0,1) /
// (0,0,0,1) the homogeneous component is (((0 + 0*a) + 0*b) + 0*c) + 1*1 == 1 whenever a, b, c are finite (0*finite =
// +-0, +-0 + 1 = 1), and x / 1.0f == x.  If a, b or c is NOT finite the reference gets NaN there; then every component
// downstream is non-finite too (inf or NaN times any matrix entry, zero included, is inf or NaN, and sums keep it), so
// "e0, e1, e2 all finite" proves the shortcut was exact; anything else takes the full chain.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Round 5: the leading `0 +` of every dot product is gone, and so are the products with matrix entries that are exactly zero
// (`Z0` / `Z1`: entry 0 / 1 of the row is 0.0f, checked per launch on the scalar unit - the skew-free K and K^-1 of a pinhole
// camera: K = [[fx,0,cx],[0,fy,cy]]).  Both change the SIGN OF A ZERO at most: 0 + x == x unless x is -0 (then +0), and
// acc + 0*b == acc unless acc is a zero of the other sign (b is finite here: pixel coordinates, or checked by the caller).  A zero of
// either sign cannot reach an output: every sum of this chain that feeds a bin, a validity test or the stored depth either adds a
// non-zero translation next, or is compared / floored / divided where +0 and -0 behave alike (u >= 0, floor, 0/0 = NaN either
// way, z > 0), and stored depths of valid points are > 0.  Bit-exactness at full size and on every fixture: tests/test_gpu_warp_splat.py.
template <bool Z0 = false, bool Z1 = false>
__device__ __forceinline__ f32x2 pk_dot2c(const float *m, f32x2 a, f32x2 b) {       // (m0*a + m1*b) + m2*1
    if (Z0) return f32x2{m[1], m[1]} * b + f32x2{m[2], m[2]};
    f32x2 acc = f32x2{m[0], m[0]} * a;
    if (!Z1) acc = acc + f32x2{m[1], m[1]} * b;
    return acc + f32x2{m[2], m[2]};
}
__device__ __forceinline__ f32x2 pk_dot3c(const float *m, f32x2 a, f32x2 b, f32x2 c) {   // dot4 with a trailing 1
    f32x2 acc = f32x2{m[0], m[0]} * a;
    acc = acc + f32x2{m[1], m[1]} * b;
    acc = acc + f32x2{m[2], m[2]} * c;
    return acc + f32x2{m[3], m[3]};
}
template <bool Z0 = false, bool Z1 = false>
__device__ __forceinline__ f32x2 pk_dot3(const float *m, f32x2 a, f32x2 b, f32x2 c) {
    if (Z0) return f32x2{m[1], m[1]} * b + f32x2{m[2], m[2]} * c;
    f32x2 acc = f32x2{m[0], m[0]} * a;
    if (!Z1) acc = acc + f32x2{m[1], m[1]} * b;
    return acc + f32x2{m[2], m[2]} * c;
}
__device__ __forceinline__ bool finite2(f32x2 a) { return __builtin_isfinite(a.x) && __builtin_isfinite(a.y); }

// :83-89 validity and :106-114 floor/ceil then clamp
__device__ __forceinline__ Proj finish(float uu, float vv, float z, bool m, float Wf, float Hf) {
    Proj p;
    p.z = z;
    const bool inb = (uu >= 0.0f) && (uu < Wf) && (vv >= 0.0f) && (vv < Hf);
    p.valid = m && (z > 0.0f) && inb;
    bins(uu, Wf - 1.0f, p.x0, p.x1);
    bins(vv, Hf - 1.0f, p.y0, p.y1);
    return p;
}

// Scalar registers are what this kernel runs out of (102 per wave; the camera is 66 scalars, every packed operand a scalar
// PAIR, and the arguments stay live to the end): each spilled scalar costs a v_writelane / v_readlane pair on the vector
// pipe.  So the fast chain keeps only the rows it uses (48 scalars; the last rows are read once for the `affine` flag), and
// the full chain - the rare branch, but the allocator sizes the whole kernel for it - fetches its own copy matrix by matrix
// (`after`: the address of a matrix is picked between two equal pointers by a value of the stage before, so that the
// scalar loads cannot be hoisted into one 66-register block).  62 -> 31 spilled scalars, 141 -> 70 lane moves per wave,
// bin_kernel 586 -> 570 us per 16 frames (same-box A/B).  Staging the FAST chain the same way (no spills left in it) measured
// no gain: the three dependent scalar-load round trips cost what the lane moves did (profiles/r02_experiments.md).
__device__ __forceinline__ GlobalF after(GlobalF p, GlobalF p_same, float v) {
    // p_same == p (relaunder): whichever is picked, the address is the same - but it is not known before v is
    const int bits = __builtin_amdgcn_readfirstlane(__float_as_int(v));
    return bits == 0x7fc12345 ? p_same : p;
}
template <int N>
__device__ __forceinline__ void fetch_rows(GlobalF p, float (&m)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = p[i];
}
// uniform: K^-1 and K end in (0,0,1); E, T, E^-1 end in (0,0,0,1)
__device__ __forceinline__ bool camera_affine(GlobalF pKinv, GlobalF pE, GlobalF pT, GlobalF pEinv, GlobalF pK) {
    int ok = 1;   // combined without short-circuit: an && chain over loaded values compiles to dependent scalar-load round trips
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float want = i == 2 ? 1.0f : 0.0f;
        ok &= (int)(pKinv[6 + i] == want) & (int)(pK[6 + i] == want);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float want = i == 3 ? 1.0f : 0.0f;
        ok &= (int)(pE[12 + i] == want) & (int)(pT[12 + i] == want) & (int)(pEinv[12 + i] == want);
    }
    return ok != 0;
}
// uniform: the four entries the SKEWFREE chain skips are exactly zero
__device__ __forceinline__ bool camera_skewfree(GlobalF pKinv, GlobalF pK) {
    return ((int)(pKinv[1] == 0.0f) & (int)(pKinv[3] == 0.0f) & (int)(pK[1] == 0.0f) & (int)(pK[3] == 0.0f)) != 0;
}
// SKEWFREE: K[0][1] == K[1][0] == 0 and the same for K^-1 (uniform, camera_skewfree): those products are skipped
template <bool SKEWFREE>
__device__ __forceinline__ bool project4_fast(GlobalF pKinv, GlobalF pE, GlobalF pT, GlobalF pEinv, GlobalF pK, int x, int y,
                                                const float (&d)[4], const bool (&m)[4], float Wf, float Hf, Proj (&p)[4]) {
    float Kinv[6], E[12];
    fetch_rows(pKinv, Kinv);
    fetch_rows(pE, E);
    const f32x2 v = f32x2{(float)y, (float)y};
    const f32x2 ua = f32x2{(float)x, (float)(x + 1)}, ub = f32x2{(float)(x + 2), (float)(x + 3)};
    const f32x2 da = f32x2{d[0], d[1]}, db = f32x2{d[2], d[3]};
    const f32x2 r0a = pk_dot2c<false, SKEWFREE>(Kinv + 0, ua, v), r0b = pk_dot2c<false, SKEWFREE>(Kinv + 0, ub, v);
    const f32x2 r1a = pk_dot2c<SKEWFREE, false>(Kinv + 3, ua, v), r1b = pk_dot2c<SKEWFREE, false>(Kinv + 3, ub, v);
    const f32x2 c0a = r0a * da, c1a = r1a * da, c0b = r0b * db, c1b = r1b * db;                   // r2 == 1: c2 = d
    float Tm[12];
    fetch_rows(pT, Tm);
    f32x2 va[3], vb[3], wa[3], wb[3], ea[3], eb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { va[i] = pk_dot3c(E + 4 * i, c0a, c1a, da); vb[i] = pk_dot3c(E + 4 * i, c0b, c1b, db); }
    float Einv[12];
    fetch_rows(pEinv, Einv);
#pragma unroll
    for (int i = 0; i < 3; ++i) { wa[i] = pk_dot3c(Tm + 4 * i, va[0], va[1], va[2]); wb[i] = pk_dot3c(Tm + 4 * i, vb[0], vb[1], vb[2]); }
    float K[6];
    fetch_rows(pK, K);
#pragma unroll
    for (int i = 0; i < 3; ++i) { ea[i] = pk_dot3c(Einv + 4 * i, wa[0], wa[1], wa[2]); eb[i] = pk_dot3c(Einv + 4 * i, wb[0], wb[1], wb[2]); }
    if (!(finite2(ea[0]) && finite2(ea[1]) && finite2(ea[2]) && finite2(eb[0]) && finite2(eb[1]) && finite2(eb[2]))) return false;
    // e3 == 1: px = e0, py = e1, z = e2;  q2 == z
    const f32x2 q0a = pk_dot3<false, SKEWFREE>(K + 0, ea[0], ea[1], ea[2]), q0b = pk_dot3<false, SKEWFREE>(K + 0, eb[0], eb[1], eb[2]);
    const f32x2 q1a = pk_dot3<SKEWFREE, false>(K + 3, ea[0], ea[1], ea[2]), q1b = pk_dot3<SKEWFREE, false>(K + 3, eb[0], eb[1], eb[2]);
    p[0] = finish(__fdiv_rn(q0a.x, ea[2].x), __fdiv_rn(q1a.x, ea[2].x), ea[2].x, m[0], Wf, Hf);
    p[1] = finish(__fdiv_rn(q0a.y, ea[2].y), __fdiv_rn(q1a.y, ea[2].y), ea[2].y, m[1], Wf, Hf);
    p[2] = finish(__fdiv_rn(q0b.x, eb[2].x), __fdiv_rn(q1b.x, eb[2].x), eb[2].x, m[2], Wf, Hf);
    p[3] = finish(__fdiv_rn(q0b.y, eb[2].y), __fdiv_rn(q1b.y, eb[2].y), eb[2].y, m[3], Wf, Hf);
    return true;
}

// ---- pc_transform_model.py:54-114, the full chain, for the four pixels of a lane: non-affine cameras and lanes with
// non-finite intermediates.  Matrix by matrix (see above): it must not need more scalar registers than the fast chain
__device__ __forceinline__ void project4_full(GlobalF pKinv, GlobalF pE, GlobalF pT, GlobalF pEinv, GlobalF pK, int x, int y,
                                              const float (&d)[4], const bool (&m)[4], float Wf, float Hf, Proj (&p)[4]) {
    float c[4][3], vv[4][4], w[4][4], e[4][4];
    {
        float Kinv[9];
        fetch_rows(relaunder((const float *)pKinv), Kinv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float u = (float)(x + k), v = (float)y;
#pragma unroll
            for (int i = 0; i < 3; ++i) c[k][i] = __fmul_rn(dot3(Kinv + 3 * i, u, v, 1.0f), d[k]);                  // :55-59
        }
    }
    {
        float E[16];
        fetch_rows(after(relaunder((const float *)pE), relaunder((const float *)pE), c[3][2]), E);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) vv[k][i] = dot4(E + 4 * i, c[k][0], c[k][1], c[k][2], 1.0f);                 // :63
    }
    {
        float Tm[16];
        fetch_rows(after(relaunder((const float *)pT), relaunder((const float *)pT), vv[3][3]), Tm);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) w[k][i] = dot4(Tm + 4 * i, vv[k][0], vv[k][1], vv[k][2], vv[k][3]);          // :68
    }
    {
        float Einv[16];
        fetch_rows(after(relaunder((const float *)pEinv), relaunder((const float *)pEinv), w[3][3]), Einv);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i) e[k][i] = dot4(Einv + 4 * i, w[k][0], w[k][1], w[k][2], w[k][3]);            // :71
    }
    float K[9];
    fetch_rows(after(relaunder((const float *)pK), relaunder((const float *)pK), e[3][3]), K);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float px = __fdiv_rn(e[k][0], e[k][3]), py = __fdiv_rn(e[k][1], e[k][3]), z = __fdiv_rn(e[k][2], e[k][3]);   // :72-73
        const float q0 = dot3(K + 0, px, py, z), q1 = dot3(K + 3, px, py, z), q2 = dot3(K + 6, px, py, z);           // :74
        p[k] = finish(__fdiv_rn(q0, q2), __fdiv_rn(q1, q2), z, m[k], Wf, Hf);                                        // :75-114
    }
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
    const long long in_base = ((long long)b * a.T_total + t) * N;
    const int y = ty0 + (threadIdx.x >> 4), x = tx0 + (threadIdx.x & 15) * 4;
    // the pixel loads go out BEFORE the 66 camera scalars are fetched: two independent round trips in flight together (a
    // one-tile workgroup lives ~9 us, most of it dependent latency: profiles/r02_experiments.md)
    float d[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    bool m[4] = {false, false, false, false};
    if (y < a.H && x < a.W) load4(a, in_base, x, y, d, m);
    const GlobalF pKinv = (GlobalF)a.Kinv + b * 9, pK = (GlobalF)a.K + b * 9, pE = (GlobalF)a.E + b * 16, pEinv = (GlobalF)a.Einv + b * 16,
                  pT = (GlobalF)a.Tt + ((long long)b * a.T_total + t) * 16;
    const bool affine = camera_affine(pKinv, pE, pT, pEinv, pK);
    const bool skewfree = camera_skewfree(pKinv, pK);
    const int g = a.per_frame ? tl : 0, G = a.per_frame ? a.T : 1;
    uint8_t *mark = a.inv_mark + ((long long)b * G + g) * N;
    long long *r2d = a.out_r2d ? a.out_r2d + ((long long)b * a.T + tl) * N * 2 : nullptr;
    const float Wf = (float)a.W, Hf = (float)a.H;

    float zmax = -INFINITY;
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -1, by1 = -1;
    if (y < a.H && x < a.W) {
        uint2 *pj = a.proj + ((long long)b * a.T + tl) * N + (long long)y * a.W + x;
        unsigned pk[8];
        Proj p4[4];
        // (one instantiation of the fast chain: a second one for cameras with skew doubled the spilled scalars; such cameras take
        //  the full chain below - exact, ~2.5x the instructions)
        const bool fast_done = affine && skewfree && project4_fast<true>(pKinv, pE, pT, pEinv, pK, x, y, d, m, Wf, Hf, p4);
        if (!fast_done) {
            // the full chain fetches its own copy of the camera, through pointers the optimiser cannot match with the staged
            // loads of the fast chain (nothing of the camera stays live for this rare branch)
            project4_full(pKinv, pE, pT, pEinv, pK, x, y, d, m, Wf, Hf, p4);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (x + k >= a.W) break;
            const Proj p = p4[k];
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
                // every invalid point carries depth max+1 and payload 0: marking its bins is enough (32-bit offsets from
                // the uniform base: N < 2^30 is checked at launch)
                const unsigned o00 = (unsigned)p.y0 * (unsigned)a.W + (unsigned)p.x0, dxo = (unsigned)(p.x1 - p.x0);
                const unsigned o10 = (unsigned)p.y1 * (unsigned)a.W + (unsigned)p.x0;
#if PF_MARK_VARIANT == 1
                // A/B (profiles/r06_experiments.md): only the DISTINCT bins are marked - a point clamped to the image edge has
                // x1 == x0 and / or y1 == y0 and stores once or twice instead of four times
                mark[o00] = 1;
                if (dxo) mark[o00 + 1] = 1;
                if (o10 != o00) {
                    mark[o10] = 1;
                    if (dxo) mark[o10 + 1] = 1;
                }
#elif PF_MARK_VARIANT == 2
                // the two bins of a row as ONE two-byte store where they differ (x1 = x0 + 1 <= W - 1: inside the row; the store may be
                // unaligned - global memory takes that), a byte store where they do not; the second row only if it is another row
                if (dxo) {
                    __builtin_memcpy(mark + o00, "\1\1", 2);
                    if (o10 != o00) __builtin_memcpy(mark + o10, "\1\1", 2);
                } else {
                    mark[o00] = 1;
                    if (o10 != o00) mark[o10] = 1;
                }
#else
                mark[o00] = 1;
                mark[o10] = 1;
                mark[o00 + dxo] = 1;
                mark[o10 + dxo] = 1;
#endif
            }
        }
        if ((a.W & 3) == 0) {   // the raster pass re-reads these instead of re-projecting (coalesced 32 B per thread)
            reinterpret_cast<uint4 *>(pj)[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            reinterpret_cast<uint4 *>(pj)[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
        }
    }
    // block reductions: max z, bounding box.  In the wave by DPP butterflies (xor 1, xor 2, half-row mirror, row mirror, then
    // lane 15 / 31 of a row broadcast into the following rows: lane 63 ends up with the reduction) on three values - the
    // box corners are < 2^13, so the two minima travel as one packed pair of u16 and the two maxima as one of i16;
    // the shuffles this replaces were 30 LDS round trips + ~150 vector instructions per lane of a kernel that is bound by
    // vector issue (profiles/r02_experiments.md)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    u16x2 bmin = {(unsigned short)min(bx0, 0xFFFF), (unsigned short)min(by0, 0xFFFF)};
    i16x2 bmax = {(short)bx1, (short)by1};
#define PF_DPP_STEP(ctrl, rmask)                                                                                              \
    {                                                                                                                         \
        zmax = fmaxf(zmax, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(zmax), __float_as_int(zmax), ctrl, rmask, 0xF, false))); \
        const int mn = __builtin_bit_cast(int, bmin), mx = __builtin_bit_cast(int, bmax);                                     \
        bmin = __builtin_elementwise_min(bmin, __builtin_bit_cast(u16x2, __builtin_amdgcn_update_dpp(mn, mn, ctrl, rmask, 0xF, false))); \
        bmax = __builtin_elementwise_max(bmax, __builtin_bit_cast(i16x2, __builtin_amdgcn_update_dpp(mx, mx, ctrl, rmask, 0xF, false))); \
    }
    PF_DPP_STEP(0xB1, 0xF)    // quad_perm [1,0,3,2]
    PF_DPP_STEP(0x4E, 0xF)    // quad_perm [2,3,0,1]
    PF_DPP_STEP(0x141, 0xF)   // row_half_mirror
    PF_DPP_STEP(0x140, 0xF)   // row_mirror: every lane of a row holds the row's reduction
    PF_DPP_STEP(0x142, 0xA)   // row_bcast:15 into rows 1 and 3
    PF_DPP_STEP(0x143, 0xC)   // row_bcast:31 into rows 2 and 3
#undef PF_DPP_STEP
    __shared__ float zred[kThreads / 64];
    __shared__ int bred[kThreads / 64][2];
    if ((threadIdx.x & 63) == 63) {
        const int w = threadIdx.x >> 6;
        zred[w] = zmax;
        bred[w][0] = __builtin_bit_cast(int, bmin);
        bred[w][1] = __builtin_bit_cast(int, bmax);
    }
    __syncthreads();
    zmax = zred[0];
    bmin = __builtin_bit_cast(u16x2, bred[0][0]);
    bmax = __builtin_bit_cast(i16x2, bred[0][1]);
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) {
        zmax = fmaxf(zmax, zred[w]);
        bmin = __builtin_elementwise_min(bmin, __builtin_bit_cast(u16x2, bred[w][0]));
        bmax = __builtin_elementwise_max(bmax, __builtin_bit_cast(i16x2, bred[w][1]));
    }
    bx0 = bmin[0] == 0xFFFF ? 0x7fffffff : (int)bmin[0];
    by0 = bmin[1] == 0xFFFF ? 0x7fffffff : (int)bmin[1];
    bx1 = bmax[0];
    by1 = bmax[1];
    if (threadIdx.x == 0) {
        const long long ntile = (long long)a.stx * a.sty;
        // :105 the sentinel is max(z)+1 over the whole predict call (one frame's points in per_frame mode): one
        // order-independent atomicMax per source tile instead of every raster workgroup re-reducing all partials
        atomicMax(&a.zmax_part[(b * a.zgroups_per_sample + (a.per_frame ? tl : 0)) * kZSlots + ((tile + b) & (kZSlots - 1))], float_to_ordered(zmax));
        a.bbox[((long long)b * a.T + tl) * ntile + tile] = make_int4(bx0, by0, bx1, by1);
    }
    // register this tile with every destination tile its box touches (one global atomicAdd per pair, ~1.5 per tile): the
    // raster workgroups then read their list instead of testing every box of the frame
    if (bx1 >= 0) {
        const int ndst = a.dtx * a.dty;
        const long long dbase = ((long long)b * G + g) * ndst;
        const unsigned id = ((unsigned)(a.per_frame ? 0 : tl) << kIdFrameShift) | (unsigned)tile;
        const int dtx0 = bx0 / kDstTW, dty0 = by0 / kDstTH;
        const int ntx = bx1 / kDstTW - dtx0 + 1, nt = ntx * (by1 / kDstTH - dty0 + 1);
        for (int j = threadIdx.x; j < nt; j += kThreads) {
            const long long dst = dbase + (dty0 + j / ntx) * a.dtx + dtx0 + j % ntx;
            const unsigned slot = atomicAdd(&a.count[dst], 1u);
            if (slot < (unsigned)kListCap) a.lists[dst * kListCap + slot] = id;
        }
    }
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRThreads, PF_RASTER_MINWAVES) void raster_kernel(SplatArgs a) {
    __shared__ unsigned long long zb[kDstTH * kDstTW];   // 16 KB
    __shared__ unsigned short list[kScan * kScanBatches];
    __shared__ unsigned ent[kListCap];
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
    [[maybe_unused]] int n_hits_total = 0;   // PF_PROBE builds report it
    for (int i = threadIdx.x; i < kDstTH * kDstTW; i += kRThreads) zb[i] = kEmpty;
    const int lt = threadIdx.x & 255, half = threadIdx.x >> 8;   // thread inside its group of 256, group

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
    // (the kernel is bound by vector-instruction issue - 87 % busy at 8 waves per SIMD, profiles/r02_k_pmc.json - and this is
    //  its inner loop: everything in 32 bits - e < 4*T*N < 2^32 is checked at launch - and one unsigned compare per range:
    //  bins are clamped to the image, so "inside the destination tile" is  bin - tile origin < tile size)
    auto splat4 = [&](const unsigned (&pk)[8], int x, int y, unsigned ebase) {
        const unsigned e_first = ebase + (unsigned)y * (unsigned)a.W + (unsigned)x;
        const unsigned Pu1 = (unsigned)P;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned f = pk[2 * k + 1];
            if (!((f >> 28) & 1u)) continue;   // invalid points were handled by the byte marks
            const unsigned rx0 = (f & 8191u) - (unsigned)dx0, ry0 = ((f >> 13) & 8191u) - (unsigned)dy0;
            const unsigned fx = (f >> 26) & 1u, fy = (f >> 27) & 1u;
            const unsigned rx1 = rx0 + fx, ry1 = ry0 + fy;
            // replicas r = 0:(x0,y0) 1:(x0,y1) 2:(x1,y0) 3:(x1,y1); e = r*P + t*N + n  (:112).  A replica
            // on the bin of a lower replica of the same point can never win the tie-break: skip it.
            const bool in_x0 = rx0 < (unsigned)kDstTW, in_x1 = rx1 < (unsigned)kDstTW && fx != 0u;
            const bool in_y0 = ry0 < (unsigned)kDstTH, in_y1 = ry1 < (unsigned)kDstTH && fy != 0u;
            const unsigned e0 = e_first + (unsigned)k;
            const unsigned long long khi = (unsigned long long)pk[2 * k] << 32;
            if (in_x0 && in_y0) atomicMin(&zb[ry0 * kDstTW + rx0], khi | e0);
            if (in_x0 && in_y1) atomicMin(&zb[ry1 * kDstTW + rx0], khi | (e0 + Pu1));
            if (in_x1 && in_y0) atomicMin(&zb[ry0 * kDstTW + rx1], khi | (e0 + 2u * Pu1));
            if (in_x1 && in_y1) atomicMin(&zb[ry1 * kDstTW + rx1], khi | (e0 + 3u * Pu1));
        }
    };
    auto load4 = [&](const uint2 *pbase, int stx, int sty, unsigned (&pk)[8], int &x, int &y) {   // source tile column, row (< 0: none)
        y = sty * kSrcTH + (lt >> 4);
        x = stx * kSrcTW + (lt & 15) * 4;
#pragma unroll
        for (int k = 0; k < 8; ++k) pk[k] = 0u;      // valid bit clear: nothing to splat
        if (stx < 0 || y >= a.H || x >= a.W) return;
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
    // ---- the destination tile's list (written by bin_kernel): kFlight source tiles' records in flight at a time
    const int ndst = a.dtx * a.dty;
    const long long dslot = ((long long)b * G + g) * ndst + dtile;
    const unsigned cnt = a.count[dslot];
    if (cnt <= (unsigned)kListCap) {
        if (threadIdx.x < cnt) {
            // source tile -> (row, column) once per entry here: a division by the tiles per row inside the loop below costs
            // every lane ~25 vector instructions per listed tile, in a kernel bound by vector issue
            const unsigned e = a.lists[dslot * kListCap + threadIdx.x], st = e & ((1u << kIdFrameShift) - 1u), row = st / (unsigned)a.stx;
            ent[threadIdx.x] = (e & ~((1u << kIdFrameShift) - 1u)) | (row << 10) | (st - row * (unsigned)a.stx);
        }
        __syncthreads();
        for (int li = 0; li < (int)cnt; li += kFlight * kRHalves) {
            unsigned pk[kFlight][8];
            int xs[kFlight], ys[kFlight];
            unsigned eb[kFlight];
#pragma unroll
            for (int j = 0; j < kFlight; ++j) {
                const int le = li + j * kRHalves + half;
                const unsigned e = le < (int)cnt ? ent[le] : 0u;
                const int tt = (int)(e >> kIdFrameShift), sx = le < (int)cnt ? (int)(e & 1023u) : -1, sy = (int)((e >> 10) & 1023u);
                const int tl = a.per_frame ? g : tt;
                eb[j] = (unsigned)tt * (unsigned)N;
                load4(a.proj + ((long long)b * a.T + tl) * N, sx, sy, pk[j], xs[j], ys[j]);
            }
#pragma unroll
            for (int j = 0; j < kFlight; ++j) splat4(pk[j], xs[j], ys[j], eb[j]);
        }
    } else {
    // the list overflowed: test every box of the group's frames (the path every call took before the lists existed)
    for (int tt = 0; tt < Tg; ++tt) {
        const int tl = a.per_frame ? g : tt;     // local frame index
        const int4 *boxes = a.bbox + ((long long)b * a.T + tl) * ntile;
        const uint2 *pbase = a.proj + ((long long)b * a.T + tl) * N;
        const unsigned ebase = (unsigned)tt * (unsigned)N;
        for (int s0 = 0; s0 < ntile; s0 += kScan * kScanBatches) {
            __syncthreads();
            if (threadIdx.x == 0) list_n = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kScanBatches; ++j) {
                const int s = s0 + j * kScan + lt;
                if (s < ntile && half == 0) {
                    const int4 bb = boxes[s];
                    if (bb.x <= dx1 && bb.z >= dx0 && bb.y <= dy1 && bb.w >= dy0) list[atomicAdd(&list_n, 1)] = (unsigned short)(s - s0);
                }
            }
            __syncthreads();
            const int n_hit = list_n;
            n_hits_total += n_hit;
            RPROBE(4);
            for (int li = 0; li < n_hit; li += 4 * kRHalves) {
                unsigned pk[4][8];
                int xs[4], ys[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int le = li + j * kRHalves + half;
                    const int st = le < n_hit ? s0 + list[le] : -1;
                    load4(pbase, st < 0 ? -1 : st % a.stx, st < 0 ? 0 : st / a.stx, pk[j], xs[j], ys[j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) splat4(pk[j], xs[j], ys[j], ebase);
            }
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
    // uniform bases + 32-bit lane offsets (N < 2^30): one address register per access instead of a 64-bit sum each
    const uint8_t *segp = a.seg + seg_base * C;
    float *outd = a.out_depth + out_base;
    uint8_t *outs = a.out_seg + out_base * C;
#pragma unroll
    for (int it = 0; it < kDstTH * kDstTW / 4 / kRThreads; ++it) {
        const int i4 = it * kRThreads + threadIdx.x;
        const int y = dy0 + i4 / (kDstTW / 4), x = dx0 + (i4 % (kDstTW / 4)) * 4;
        if (y >= a.H || x >= a.W) continue;
        const unsigned n0 = (unsigned)y * (unsigned)a.W + (unsigned)x;
        unsigned zbits[4], src[4];
        bool empty[4];
        uint8_t mk[4], sg[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long key = zb[i4 * 4 + k];
            zbits[k] = (unsigned)(key >> 32);
            empty[k] = zbits[k] == 0xFFFFFFFFu;          // kEmpty; a valid point's z > 0 is never the all-ones pattern
            unsigned e = (unsigned)key;                  // e = r*P + t*N + n with r < 4: strip the corner replica
            e -= e >= 2u * Pu ? 2u * Pu : 0u;
            e -= e >= Pu ? Pu : 0u;
            src[k] = empty[k] ? 0u : e;                  // always a readable address
        }
        if ((a.W & 3) == 0) {
            const uchar4 mv = *reinterpret_cast<const uchar4 *>(mark + n0);
            mk[0] = mv.x; mk[1] = mv.y; mk[2] = mv.z; mk[3] = mv.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) mk[k] = x + k < a.W ? mark[n0 + k] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) sg[k] = C == 1 ? segp[src[k]] : (uint8_t)0;
        float dep[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // empty bin: won by an invalid point (:105,:133) -> max+1, or never touched (:136-138) -> -1
            dep[k] = !empty[k] ? __uint_as_float(zbits[k]) : (mk[k] ? sentinel : -1.0f);
            if (empty[k]) sg[k] = 0;
        }
        if ((a.W & 3) == 0) {
            *reinterpret_cast<float4 *>(outd + n0) = make_float4(dep[0], dep[1], dep[2], dep[3]);
            if (C == 1) *reinterpret_cast<uchar4 *>(outs + n0) = make_uchar4(sg[0], sg[1], sg[2], sg[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < a.W) {
                    outd[n0 + k] = dep[k];
                    if (C == 1) outs[n0 + k] = sg[k];
                }
        }
        if (C != 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (x + k < a.W)
                    for (int c = 0; c < C; ++c)
                        outs[(size_t)(n0 + k) * C + c] = !empty[k] ? segp[(size_t)src[k] * C + c] : (uint8_t)0;
        }
    }
    RPROBE(2);
}

struct SplatLayout {
    size_t bbox_off, zmax_off, count_off, mark_off, mark_bytes, proj_off, lists_off, total;
    int stx, sty, dtx, dty;
};

static SplatLayout splat_layout(int B, int T, int H, int W, int per_frame) {
    SplatLayout L;
    L.stx = (W + kSrcTW - 1) / kSrcTW; L.sty = (H + kSrcTH - 1) / kSrcTH;
    L.dtx = (W + kDstTW - 1) / kDstTW; L.dty = (H + kDstTH - 1) / kDstTH;
    const size_t ntile = (size_t)L.stx * L.sty, N = (size_t)H * W;
    L.bbox_off = 0;
    L.zmax_off = align_up(L.bbox_off + (size_t)B * T * ntile * sizeof(int4), 256);
    // zmax slots, list counters and marks are contiguous: one memset per call
    const size_t ndst = (size_t)L.dtx * L.dty, G = per_frame ? T : 1;
    L.count_off = align_up(L.zmax_off + (size_t)B * T * kZSlots * sizeof(unsigned), 256);
    L.mark_off = align_up(L.count_off + B * G * ndst * sizeof(unsigned), 256);
    L.mark_bytes = (size_t)B * G * N;
    L.proj_off = align_up(L.mark_off + L.mark_bytes, 256);
    L.lists_off = align_up(L.proj_off + (size_t)B * T * N * sizeof(uint2), 256);
    L.total = align_up(L.lists_off + B * G * ndst * kListCap * sizeof(unsigned), 256);
    return L;
}

}  // namespace pf

extern "C" int pf_warp_splat_workspace(int B, int T, int H, int W, int per_frame, size_t *bytes) {
    if (!bytes || B <= 0 || T <= 0 || H <= 0 || W <= 0)
        return pf::fail(PF_EINVAL, "pf_warp_splat_workspace: bad dims B=%d T=%d H=%d W=%d", B, T, H, W);
    if (4ll * T * H * W >= (1ll << 32))
        return pf::fail(PF_EUNSUPPORTED, "pf_warp_splat: 4*T*H*W must be < 2^32 (element index packs in 32 bits)");
    if (H > 8192 || W > 8192 || T >= 4096)
        return pf::fail(PF_EUNSUPPORTED, "pf_warp_splat: H and W must be <= 8192 (13-bit bin coordinates) and T < 4096");
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
    a.count = (unsigned *)((char *)ws + L.count_off);
    a.lists = (unsigned *)((char *)ws + L.lists_off);
    a.out_seg = out_seg; a.out_depth = out_depth; a.out_r2d = (long long *)out_result2d;
    a.B = B; a.T_total = T_total; a.t_first = t_first; a.T = T; a.H = H; a.W = W; a.C = seg_channels;
    a.per_frame = (per_frame & PF_SPLAT_PER_FRAME) ? 1 : 0;
    a.zgroups_per_sample = (per_frame & PF_SPLAT_PER_SAMPLE_SENTINEL) ? (a.per_frame ? T : 1) : 0;
    a.stx = L.stx; a.sty = L.sty; a.dtx = L.dtx; a.dty = L.dty;
    a.probe = nullptr;
#if PF_PROBE
    static const bool splat_probe = ab_env("PF_PROBE") != nullptr;   // read once, not per call
    a.probe = splat_probe ? pf::probe_buffer() : nullptr;
#endif
    hipStream_t s = (hipStream_t)stream;
    const int G = a.per_frame ? T : 1;

    // algorithmic bytes: source side depth 4 + mask 1 B/px; destination side seg 1 in, seg 1 + depth 4 out
    const double src_px = (double)B * T * H * W, dst_px = (double)B * G * H * W;
    {
        pf::ProfScope ps(s, "inv_mark_memset", 0, (double)L.mark_bytes);
        // zmax slots (ordered-u32 encoding: 0 is below every float) + invalid-point marks, contiguous
        if (int rc = pf::launch_zero_fill(a.zmax_part, (L.mark_off - L.zmax_off) + L.mark_bytes, s)) return rc;
    }
    {
        pf::ProfScope ps(s, "pf::bin_kernel(pf::SplatArgs)", 0, src_px * (5.0 + (out_result2d ? 16.0 : 0.0)));
        hipLaunchKernelGGL(pf::bin_kernel, dim3(L.stx * L.sty, T, B), dim3(pf::kThreads), 0, s, a);
        PF_LAUNCH_CHECK("bin_kernel");
    }
    {
        pf::ProfScope ps(s, "pf::raster_kernel(pf::SplatArgs)", 0, dst_px * (2.0 * seg_channels + 4.0));
        hipLaunchKernelGGL(pf::raster_kernel, dim3(L.dtx * L.dty, G, B), dim3(pf::kRThreads), 0, s, a);
        PF_LAUNCH_CHECK("raster_kernel");
    }
    return PF_OK;
}
