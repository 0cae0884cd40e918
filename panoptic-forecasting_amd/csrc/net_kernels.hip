// The non-GEMM stages of the bg network (all HBM/L2-bound, VALU only):
//   stem_onehot_kernel : bg_model.py:53-69 (one-hot, depth normalise, concat) fused with the stem conv
//                        hardnet.py base.0 (3x3 s2, BN folded, ReLU).  The [B,36,H,W] tensor is never built:
//                        a one-hot channel contributes exactly one weight column per tap, so the conv is
//                        a gather-sum of 16-wide weight rows from LDS.  Optionally emulates the reference's
//                        on-disk hop (export_cityscapes_segmentation_results.py:34-38,119-124 and
//                        bg_dataset.py:224-230,166-170) on the fly.
//   avgpool2_kernel    : nn.AvgPool2d(2,2)                     hardnet.py:296
//   upsample_kernel    : F.interpolate(bilinear, align_corners) hardnet.py:248-253
//   head_kernel        : final bilinear upsample + argmax      hardnet.py:372-384, bg_model.py:98
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "net_kernels.h"
#include "conv_epilogue.h"
#include "pf_prof.h"

namespace pf {

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_onehot_kernel(StemArgs a) {
    // weights in LDS as [tap][ch][16]: row = one input channel's 16 output weights for that tap
    extern __shared__ __attribute__((aligned(16))) float wl[];
    __shared__ uint8_t lut[256];
    const int in_ch = a.T * (a.n_cls + 1);
    for (int e = threadIdx.x; e < 9 * in_ch * 16; e += 256) {
        const int co = e & 15, ch = (e >> 4) % in_ch, tap = (e >> 4) / in_ch;
        wl[e] = a.w[((size_t)co * in_ch + ch) * 9 + tap];
    }
    lut[threadIdx.x] = (a.hop & PF_HOP_TRAINID_LUT) ? a.lut[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();

    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= a.Wout || oy >= a.Hout) return;
    const size_t N = (size_t)a.H * a.W;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = a.bias[i];
    const float mean_ = a.depth_mean, std_ = a.depth_std;

    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if (iy < 0 || iy >= a.H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if (ix < 0 || ix >= a.W) continue;
            const int tap = ky * 3 + kx;
            const size_t pix = (size_t)iy * a.W + ix;
            for (int t = 0; t < a.T; ++t) {
                const size_t idx = ((size_t)b * a.T + t) * N + pix;
                // label -> one-hot column (labels >= n_cls contribute nothing, bg_model.py:54-57)
                int cls = a.seg_is_i64 ? (int)reinterpret_cast<const long long *>(a.seg)[idx]
                                       : (int)reinterpret_cast<const uint8_t *>(a.seg)[idx];
                if (a.hop & PF_HOP_TRAINID_LUT) cls = lut[cls & 255];
                if (cls >= 0 && cls < a.n_cls) {
                    const f32x4v *row = reinterpret_cast<const f32x4v *>(wl + (tap * in_ch + t * a.n_cls + cls) * 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v r = row[q];
                        acc[q * 4 + 0] += r[0]; acc[q * 4 + 1] += r[1]; acc[q * 4 + 2] += r[2]; acc[q * 4 + 3] += r[3];
                    }
                }
                // depth channel: ((d - mean)/std) * mask   (bg_model.py:50-51,66-67)
                float d = a.depth[idx];
                float m;
                if (a.hop & PF_HOP_DEPTH_U16) {
                    const float q = rintf(fminf(fmaxf(d + 1.f, 0.f), 255.f) * 256.f);  // export :119-124
                    d = q / 256.f - 1.f;                                                // load bg_dataset.py:225
                    const bool mk = d > 0.f;
                    d = mk ? fminf(fmaxf(d, a.min_depth), a.max_depth) : -1.f;           // :227-228,:166-170
                    m = mk ? 1.f : 0.f;
                } else {
                    m = a.mask[idx] ? 1.f : 0.f;
                }
                const float dn = ((d - mean_) / std_) * m;
                const f32x4v *row = reinterpret_cast<const f32x4v *>(wl + (tap * in_ch + a.T * a.n_cls + t) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4v r = row[q];
                    acc[q * 4 + 0] += r[0] * dn; acc[q * 4 + 1] += r[1] * dn;
                    acc[q * 4 + 2] += r[2] * dn; acc[q * 4 + 3] += r[3] * dn;
                }
            }
        }
    }
    const size_t op = (size_t)a.Hout * a.Wout;
    float *o = a.dst + (size_t)b * 16 * op + (size_t)oy * a.Wout + ox;
    float vmax = 0.f;   // the output is >= 0 after the ReLU: range guard of the operand split (conv_mfma.h)
    bool nan = false;   // a NaN input propagates like in the reference's fp32 conv, but max() would drop it: flag it
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o[i * op] = fmaxf(acc[i], 0.f);
        vmax = fmaxf(vmax, acc[i]);
        nan = nan || acc[i] != acc[i];
    }
    range_commit(a.status, a.range_slot, nan ? __builtin_inff() : vmax);
}

// ------------------------------------------------------------------------------------------------
// Stem, latency-organised (T compile-time): the generic kernel above walks its 9*T taps one dependent
// load -> LUT -> weight-row chain at a time (54 serial HBM round trips per wave; profiles/r01_e: 193 us at B=4 for
// 65 us of traffic).  Here every lane first issues ALL its 9*T label and depth loads (clamped addresses, no branches
// in between, so they are in flight together), and only then consumes them; the depth-channel weights are uniform
// per (tap, t) and come through the scalar cache (a.wdep, [tap][t][16]) instead of LDS, which halves the LDS reads.
template <int T, bool SEG64>
__global__ __launch_bounds__(256) void stem_onehot_batched_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [tap][ch][16]
    __shared__ uint8_t lut[256];
    const int in_ch = T * (a.n_cls + 1);
    for (int e = threadIdx.x; e < 9 * in_ch * 16; e += 256) {
        const int co = e & 15, ch = (e >> 4) % in_ch, tap = (e >> 4) / in_ch;
        wl[e] = a.w[((size_t)co * in_ch + ch) * 9 + tap];
    }
    lut[threadIdx.x] = (a.hop & PF_HOP_TRAINID_LUT) ? a.lut[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();

    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= a.Wout || oy >= a.Hout) return;
    const size_t N = (size_t)a.H * a.W;
    const bool hop_d = (a.hop & PF_HOP_DEPTH_U16) != 0;

    // ---- phase 1: all loads
    int lab[9 * T];
    float dep[9 * T];
    uint8_t msk[9 * T];
    unsigned okbits = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy * 2 - 1 + tap / 3, ix = ox * 2 - 1 + tap % 3;
        const bool ok = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        okbits |= ok ? (1u << tap) : 0u;
        const size_t pix = (size_t)(ok ? iy : 0) * a.W + (ok ? ix : 0);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const size_t idx = ((size_t)b * T + t) * N + pix;
            lab[tap * T + t] = SEG64 ? (int)reinterpret_cast<const long long *>(a.seg)[idx]
                                     : (int)reinterpret_cast<const uint8_t *>(a.seg)[idx];
            dep[tap * T + t] = a.depth[idx];
            msk[tap * T + t] = hop_d ? (uint8_t)0 : a.mask[idx];
        }
    }

    // ---- phase 2
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = a.bias[i];
    const float mean_ = a.depth_mean, std_ = a.depth_std;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const bool ok = (okbits >> tap) & 1u;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int cls = lab[tap * T + t];
            if (a.hop & PF_HOP_TRAINID_LUT) cls = lut[cls & 255];
            if (ok && cls >= 0 && cls < a.n_cls) {     // labels >= n_cls contribute nothing (bg_model.py:54-57)
                const f32x4v *row = reinterpret_cast<const f32x4v *>(wl + (tap * in_ch + t * a.n_cls + cls) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4v r = row[q];
                    acc[q * 4 + 0] += r[0]; acc[q * 4 + 1] += r[1]; acc[q * 4 + 2] += r[2]; acc[q * 4 + 3] += r[3];
                }
            }
            float d = dep[tap * T + t], m;
            if (hop_d) {
                const float q = rintf(fminf(fmaxf(d + 1.f, 0.f), 255.f) * 256.f);  // export :119-124
                d = q / 256.f - 1.f;                                                // load bg_dataset.py:225
                const bool mk = d > 0.f;
                d = mk ? fminf(fmaxf(d, a.min_depth), a.max_depth) : -1.f;           // :227-228,:166-170
                m = mk ? 1.f : 0.f;
            } else {
                m = msk[tap * T + t] ? 1.f : 0.f;
            }
            const float dn = ok ? ((d - mean_) / std_) * m : 0.f;                    // (bg_model.py:50-51,66-67)
            const float *wd = a.wdep + (tap * T + t) * 16;                           // uniform: scalar loads
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] += wd[i] * dn;
        }
    }
    const size_t op = (size_t)a.Hout * a.Wout;
    float *o = a.dst + (size_t)b * 16 * op + (size_t)oy * a.Wout + ox;
    float vmax = 0.f;   // the output is >= 0 after the ReLU: range guard of the operand split (conv_mfma.h)
    bool nan = false;   // a NaN input propagates like in the reference's fp32 conv, but max() would drop it: flag it
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        o[i * op] = fmaxf(acc[i], 0.f);
        vmax = fmaxf(vmax, acc[i]);
        nan = nan || acc[i] != acc[i];
    }
    range_commit(a.status, a.range_slot, nan ? __builtin_inff() : vmax);
}

// ------------------------------------------------------------------------------------------------
// Stem, third form.  The batched kernel above was bound by the vector ALU (SQ counters, profiles/r02_a_pmc.json: 2270
// instructions per output pixel, 300 M wave-instructions per 16 frames = 485 us of issue time in a 573 us kernel), and a
// third of those instructions were not arithmetic: 64-bit address arithmetic for each of the 54 loads of a lane, a divergent
// branch per (tap, frame) around the one-hot row, and every workgroup re-gathering its 20 KB of weights from the OIHW array
// 4 bytes at a time.  Here
//   * the one-hot rows arrive pre-packed [tap][t][n_cls + 1][16] (plan creation; the last row of each group is zero) and are
//     copied to LDS with 16-B loads; a label outside 0..n_cls-1 (or a tap outside the image) selects the zero row: no branch;
//   * loads use one uniform base per tensor + a 32-bit lane offset;
//   * accumulation is on register pairs (v_pk_add_f32 for the one-hot rows, v_pk_fma_f32 with scalar weight pairs for depth).
// Same operations on the same values in the same order as the kernels above: outputs are bit-identical.
typedef float f32x2v __attribute__((ext_vector_type(2)));
constexpr int kStemRow = 16;   // floats per one-hot weight row in LDS (20 = conflict-free for any label mix; measured no faster: warped label maps are coherent)
template <int T, bool SEG64, bool HOP_D, bool HOP_LUT>
__global__ __launch_bounds__(256) void stem_onehot_v3_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [tap][t][n_cls + 1][kStemRow]
    __shared__ uint8_t lut[256];
    const int rows_per = a.n_cls + 1;
    {
        const f32x4v *src = reinterpret_cast<const f32x4v *>(a.woh);
        f32x4v *dst = reinterpret_cast<f32x4v *>(wl);
        // (kStemRow = 20 would pad the rows so that lanes picking different rows never share a bank group)
        for (int e = threadIdx.x; e < 9 * T * rows_per * 4; e += 256) dst[(e >> 2) * (kStemRow / 4) + (e & 3)] = src[e];
    }
    lut[threadIdx.x] = HOP_LUT ? a.lut[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();

    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= a.Wout || oy >= a.Hout) return;
    const unsigned N = (unsigned)a.H * (unsigned)a.W;
    constexpr bool hop_d = HOP_D;
    // uniform bases of this sample; lane offsets in elements (T*N < 2^32 is checked at launch)
    const uint8_t *seg8 = reinterpret_cast<const uint8_t *>(a.seg) + (size_t)b * T * N;
    const long long *seg64 = reinterpret_cast<const long long *>(a.seg) + (size_t)b * T * N;
    const float *depth = a.depth + (size_t)b * T * N;
    const uint8_t *mask = a.mask ? a.mask + (size_t)b * T * N : nullptr;
    // One (tap) per iteration of a ROLLED loop, the loads of the next three taps in flight (registers rotate).  Fully
    // unrolled, hipcc issues all 216 LDS row reads of the straight-line code first and spills them (512 registers +
    // scratch; scheduling fences do not stop it); the rolled loop holds 24 reads.
    struct Tap { int lab[T]; float dep[T]; uint8_t msk[T]; bool ok; };
    auto issue = [&](int tap, Tap &p) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
        p.ok = tap < 9 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        const unsigned pix = p.ok ? (unsigned)iy * (unsigned)a.W + (unsigned)ix : 0u;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const unsigned idx = (unsigned)t * N + pix;
            p.lab[t] = SEG64 ? (int)seg64[idx] : (int)seg8[idx];
            p.dep[t] = depth[idx];
            p.msk[t] = hop_d ? (uint8_t)0 : mask[idx];
        }
    };
    f32x2v acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x2v{a.bias[2 * i], a.bias[2 * i + 1]};
    const float mean_ = a.depth_mean, std_ = a.depth_std;
    Tap c0, c1, c2, c3;
    issue(0, c0);
    issue(1, c1);
    issue(2, c2);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
        issue(tap + 3 < 9 ? tap + 3 : 8, c3);
        const float *wrow = wl + tap * T * rows_per * kStemRow;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int cls = c0.lab[t];
            if (HOP_LUT) cls = lut[cls & 255];
            // labels >= n_cls contribute nothing (bg_model.py:54-57): they, and taps outside the image, read the zero row
            const int r = (c0.ok && (unsigned)cls < (unsigned)a.n_cls) ? cls : a.n_cls;
            const f32x4v *row = reinterpret_cast<const f32x4v *>(wrow + (t * rows_per + r) * kStemRow);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4v w4 = row[q];
                acc[2 * q] += f32x2v{w4[0], w4[1]};
                acc[2 * q + 1] += f32x2v{w4[2], w4[3]};
            }
            float d = c0.dep[t], m;
            if (hop_d) {
                const float q = rintf(fminf(fmaxf(d + 1.f, 0.f), 255.f) * 256.f);  // export :119-124
                d = q / 256.f - 1.f;                                                // load bg_dataset.py:225
                const bool mk = d > 0.f;
                d = mk ? fminf(fmaxf(d, a.min_depth), a.max_depth) : -1.f;           // :227-228,:166-170
                m = mk ? 1.f : 0.f;
            } else {
                m = c0.msk[t] ? 1.f : 0.f;
            }
            const float dn = c0.ok ? ((d - mean_) / std_) * m : 0.f;                 // (bg_model.py:50-51,66-67)
            const float *wd = a.wdep + (tap * T + t) * 16;                           // uniform: scalar loads, SGPR-pair operands
            const f32x2v dn2 = f32x2v{dn, dn};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(f32x2v{wd[2 * i], wd[2 * i + 1]}, dn2, acc[i]);
        }
        // the LDS row reads stay inside their iteration (hipcc otherwise rotates the loop and carries 12 rows = 48 registers
        // across the back edge: one wave per SIMD less); their ~100 cycles are hidden by the other waves
        __builtin_amdgcn_sched_barrier(0);
        c0 = c1;
        c1 = c2;
        c2 = c3;
    }
    const size_t op = (size_t)a.Hout * a.Wout;
    float *o = a.dst + (size_t)b * 16 * op + (size_t)oy * a.Wout + ox;
    float vmax = 0.f;   // range guard of the operand split (conv_mfma.h)
    bool nan = false;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        o[(2 * i) * op] = fmaxf(acc[i].x, 0.f);
        o[(2 * i + 1) * op] = fmaxf(acc[i].y, 0.f);
        vmax = fmaxf(vmax, fmaxf(acc[i].x, acc[i].y));
        nan = nan || acc[i].x != acc[i].x || acc[i].y != acc[i].y;
    }
    range_commit(a.status, a.range_slot, nan ? __builtin_inff() : vmax);
}

// ------------------------------------------------------------------------------------------------
// Stem, 2 x 2 outputs per lane (u8 labels; HOP_D: in-register depth hop - the fused forecast model - else depth + mask planes).
// stem_onehot_v3_kernel is 82 % vector-ALU-busy (PMC) and most of its instructions are not arithmetic on the accumulators:
// a lane evaluates the label -> row and depth -> hop -> normalise chain for all 9 x T taps of ITS output pixel, although a
// stride-2 3x3 window shares 5 of its 9 columns/rows with the neighbouring outputs - every input pixel goes through the
// chain 2.25 times - and it issues 2 loads per tap and frame (54 per output pixel).  Here a lane owns the outputs
// (oy0..oy0+1) x (ox0..ox0+1): their windows cover 5 x 5 input pixels, so
//   * the chain runs once per input pixel and frame: 75 instead of 108 evaluations per 4 outputs;
//   * the five pixels of a window row are ONE aligned 4-byte (labels) / 16-byte (depth) load plus the left neighbour:
//     4 load instructions per row and frame, 15 per output pixel instead of 54;
//   * an input pixel is accumulated into the 1-4 (output, tap) pairs it belongs to; which output ROWS an input row feeds
//     (row 2 feeds both) is wave-uniform: scalar branches in a rolled loop over the 5 input rows.
// For every output the operations and their order are those of v3 (ky, kx, t ascending; one-hot row, then depth): the
// results are bit-identical.  FAST_DIV: (d - mean) / std as q = x*r, q += fma(-q, std, x) * r with r = 1/std - correctly
// rounded for every value the hop chain can produce, which launch_stem() proves by trying all 65 281 of them on the host
// (stem_fast_div_exact) before it selects this variant.
// (134 registers = 3 waves per SIMD; capped at 128 for 4 waves it measures the same 342 us with 5 spills)
template <int T, bool HOP_D, bool HOP_LUT, bool FAST_DIV>
__global__ __launch_bounds__(256) void stem_onehot_v4_kernel(StemArgs a, float inv_std) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [tap][t][n_cls + 1][16]
    __shared__ uint8_t lut[256];
    const int rows_per = a.n_cls + 1;
    {
        const f32x4v *src = reinterpret_cast<const f32x4v *>(a.woh);
        f32x4v *dst = reinterpret_cast<f32x4v *>(wl);
        for (int e = threadIdx.x; e < 9 * T * rows_per * 4; e += 256) dst[e] = src[e];
    }
    lut[threadIdx.x] = HOP_LUT ? a.lut[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();

    const int ox0 = blockIdx.x * 128 + 2 * (threadIdx.x & 63);
    const int oy0 = blockIdx.y * 8 + 2 * (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox0 >= a.Wout || oy0 >= a.Hout) return;          // Wout, Hout even and W = 2 Wout, H = 2 Hout (checked at launch)
    const unsigned N = (unsigned)a.H * (unsigned)a.W;
    const uint8_t *seg8 = reinterpret_cast<const uint8_t *>(a.seg) + (size_t)b * T * N;
    const float *depth = a.depth + (size_t)b * T * N;
    const uint8_t *mask = HOP_D ? nullptr : a.mask + (size_t)b * T * N;
    const int ix0 = 2 * ox0;                               // multiple of 4: window columns ix0 - 1 .. ix0 + 3
    const bool left_ok = ix0 > 0;
    struct Row { unsigned lab4[T], labl[T], msk4[T], mskl[T]; f32x4v dep4[T]; float depl[T]; bool ok; };
    auto issue = [&](int r, Row &p) {
        const int iy = 2 * oy0 - 1 + r;                    // <= H - 1 for r <= 4
        p.ok = iy >= 0;
        const unsigned row = (unsigned)(p.ok ? iy : 0) * (unsigned)a.W + (unsigned)ix0;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const unsigned idx = (unsigned)t * N + row;
            p.lab4[t] = *reinterpret_cast<const unsigned *>(seg8 + idx);
            p.dep4[t] = *reinterpret_cast<const f32x4v *>(depth + idx);
            const unsigned il = left_ok ? idx - 1u : idx;
            p.labl[t] = seg8[il];
            p.depl[t] = depth[il];
            p.msk4[t] = HOP_D ? 0u : *reinterpret_cast<const unsigned *>(mask + idx);
            p.mskl[t] = HOP_D ? 0u : (unsigned)mask[il];
        }
    };
    f32x2v acc[2][2][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[o >> 1][o & 1][i] = f32x2v{a.bias[2 * i], a.bias[2 * i + 1]};
    const float mean_ = a.depth_mean, std_ = a.depth_std;
    // one (output, tap) pair: the one-hot row of the label, then the depth channel - as in v3
    auto accum = [&](f32x2v (&ac)[8], const float *wrow, const float *wd, float dn) {
        const f32x4v *row = reinterpret_cast<const f32x4v *>(wrow);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4v w4 = row[q];
            ac[2 * q] += f32x2v{w4[0], w4[1]};
            ac[2 * q + 1] += f32x2v{w4[2], w4[3]};
        }
        const f32x2v dn2 = f32x2v{dn, dn};
#pragma unroll
        for (int i = 0; i < 8; ++i) ac[i] = __builtin_elementwise_fma(f32x2v{wd[2 * i], wd[2 * i + 1]}, dn2, ac[i]);
    };
    Row cur, nxt;
    issue(0, cur);
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
        issue(r < 4 ? r + 1 : 4, nxt);
        const bool f0 = r <= 2, f1 = r >= 2;               // input row r = tap row r of output row 0, r - 2 of output row 1
        const int tap0 = r * 3, tap1 = (r - 2) * 3;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const bool okc = cur.ok && (c > 0 || left_ok);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                int cls = c == 0 ? (int)cur.labl[t] : (int)((cur.lab4[t] >> (8 * (c == 0 ? 0 : c - 1))) & 255u);
                float d = c == 0 ? cur.depl[t] : cur.dep4[t][c == 0 ? 0 : c - 1];
                if (HOP_LUT) cls = lut[cls & 255];
                // labels >= n_cls contribute nothing (bg_model.py:54-57): they, and taps outside the image, read the zero row
                const int rowi = (okc && (unsigned)cls < (unsigned)a.n_cls) ? cls : a.n_cls;
                float m;
                if (HOP_D) {
                    const float q = rintf(fminf(fmaxf(d + 1.f, 0.f), 255.f) * 256.f);  // export :119-124
                    d = q / 256.f - 1.f;                                                // load bg_dataset.py:225
                    const bool mk = d > 0.f;
                    d = mk ? fminf(fmaxf(d, a.min_depth), a.max_depth) : -1.f;           // :227-228,:166-170
                    m = mk ? 1.f : 0.f;
                } else {
                    const unsigned mb = c == 0 ? cur.mskl[t] : (cur.msk4[t] >> (8 * (c == 0 ? 0 : c - 1))) & 255u;
                    m = mb ? 1.f : 0.f;
                }
                float qn;
                if (FAST_DIV) {
                    const float x = d - mean_;
                    const float q0 = x * inv_std;
                    qn = __builtin_fmaf(__builtin_fmaf(-q0, std_, x), inv_std, q0);
                } else {
                    qn = (d - mean_) / std_;
                }
                const float dn = okc ? qn * m : 0.f;                                 // (bg_model.py:50-51,66-67)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int kx = c - 2 * dx;
                    if (kx < 0 || kx > 2) continue;
                    if (f0) accum(acc[0][dx], wl + (((tap0 + kx) * T + t) * rows_per + rowi) * 16, a.wdep + ((tap0 + kx) * T + t) * 16, dn);
                    if (f1) accum(acc[1][dx], wl + (((tap1 + kx) * T + t) * rows_per + rowi) * 16, a.wdep + ((tap1 + kx) * T + t) * 16, dn);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
    const size_t op = (size_t)a.Hout * a.Wout;
    float vmax = 0.f;   // range guard of the operand split (conv_mfma.h): base.1 reads this tensor through conv_split
    if (!HOP_D) {
        // a NaN depth (the hop chain cannot produce one: its clamp absorbs it) propagates through the reference's fp32 conv; here
        // the ReLU's max would turn it into 0: the sum of all accumulators is NaN iff one of them is (or two infinities cancel)
        f32x2v sum = f32x2v{0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < 8; ++i) sum += acc[o >> 1][o & 1][i];
        const float s1 = sum.x + sum.y;
        if (s1 != s1) vmax = __builtin_inff();
    }
    if (a.dst_fmt) {
        // packed pairs for the fused front kernel (conv_front.hip): per term and channel group one 16-B unit [2 px][4 ch]
        char *base = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * 4 * op * 8;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const size_t pix = (size_t)(oy0 + dy) * a.Wout + ox0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4v v0, v1;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v0[2 * h] = fmaxf(acc[dy][0][2 * g + h].x, 0.f); v0[2 * h + 1] = fmaxf(acc[dy][0][2 * g + h].y, 0.f);
                    v1[2 * h] = fmaxf(acc[dy][1][2 * g + h].x, 0.f); v1[2 * h + 1] = fmaxf(acc[dy][1][2 * g + h].y, 0.f);
                }
                vmax = range_acc(range_acc(vmax, v0[0], v0[1], v0[2], v0[3]), v1[0], v1[1], v1[2], v1[3]);
                split_x4 h0, m0, h1, m1;
                split_terms4(v0, h0, m0);
                split_terms4(v1, h1, m1);
                char *p = base + ((size_t)g * op + pix) * 8;
                *reinterpret_cast<split_x8 *>(p) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<split_x8 *>(p + (size_t)4 * op * 8) = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
        range_commit(a.status, a.range_slot, vmax);
        return;
    }
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        float *o = a.dst + (size_t)b * 16 * op + (size_t)(oy0 + dy) * a.Wout + ox0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<f32x2v *>(o + (2 * i) * op) = f32x2v{fmaxf(acc[dy][0][i].x, 0.f), fmaxf(acc[dy][1][i].x, 0.f)};
            *reinterpret_cast<f32x2v *>(o + (2 * i + 1) * op) = f32x2v{fmaxf(acc[dy][0][i].y, 0.f), fmaxf(acc[dy][1][i].y, 0.f)};
            vmax = range_acc(vmax, acc[dy][0][i].x, acc[dy][1][i].x, acc[dy][0][i].y, acc[dy][1][i].y);
        }
    }
    range_commit(a.status, a.range_slot, vmax);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       int planes, int Hin, int Win, int Hout, int Wout, unsigned *status, unsigned *slot) {
    float vmax = 0.f;
    const size_t total = (size_t)planes * Hout * Wout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wout);
        const size_t r = i / Wout;
        const int y = (int)(r % Hout);
        const size_t p = r / Hout;
        const float *s = src + (p * Hin + 2 * y) * (size_t)Win + 2 * x;
        const float v = (((s[0] + s[1]) + s[Win]) + s[Win + 1]) * 0.25f;
        dst[i] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    range_commit(status, slot, vmax);
}

__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       int planes, int Hin, int Win, int Hout, int Wout, unsigned *status, unsigned *slot) {
    float vmax = 0.f;
    const float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
    const float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    const size_t total = (size_t)planes * Hout * Wout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wout);
        const size_t r = i / Wout;
        const int y = (int)(r % Hout);
        const size_t p = r / Hout;
        int y0, y1, x0, x1;
        float hy0, hy1, lx0, lx1;
        lin_coord(y, sh, Hin, y0, y1, hy0, hy1);
        lin_coord(x, sw, Win, x0, x1, lx0, lx1);
        const float *s = src + p * (size_t)Hin * Win;
        const float t0 = lx0 * s[(size_t)y0 * Win + x0] + lx1 * s[(size_t)y0 * Win + x1];
        const float t1 = lx0 * s[(size_t)y1 * Win + x0] + lx1 * s[(size_t)y1 * Win + x1];
        const float v = hy0 * t0 + hy1 * t1;
        dst[i] = v;
        vmax = fmaxf(vmax, fabsf(v));
    }
    range_commit(status, slot, vmax);
}

// final upsample + argmax; logits [B,C,Hin,Win] -> seg [B,Hout,Wout] (+ optional full logits)
__global__ __launch_bounds__(256) void head_kernel(HeadArgs a) {
    const float sh = a.Hout > 1 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f;
    const float sw = a.Wout > 1 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const size_t opl = (size_t)a.Hout * a.Wout, ipl = (size_t)a.Hin * a.Win;
    const size_t total = (size_t)a.B * opl;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % a.Wout);
        const size_t r = i / a.Wout;
        const int y = (int)(r % a.Hout);
        const int b = (int)(r / a.Hout);
        int y0, y1, x0, x1;
        float hy0, hy1, lx0, lx1;
        lin_coord(y, sh, a.Hin, y0, y1, hy0, hy1);
        lin_coord(x, sw, a.Win, x0, x1, lx0, lx1);
        const float *s = a.logits + (size_t)b * a.C * ipl;
        const size_t o00 = (size_t)y0 * a.Win + x0, o01 = (size_t)y0 * a.Win + x1;
        const size_t o10 = (size_t)y1 * a.Win + x0, o11 = (size_t)y1 * a.Win + x1;
        float best = -INFINITY;
        int arg = 0;
        for (int c = 0; c < a.C; ++c) {
            const float *sc = s + (size_t)c * ipl;
            const float t0 = lx0 * sc[o00] + lx1 * sc[o01];
            const float t1 = lx0 * sc[o10] + lx1 * sc[o11];
            const float v = hy0 * t0 + hy1 * t1;
            if (a.out_logits) a.out_logits[((size_t)b * a.C + c) * opl + (size_t)y * a.Wout + x] = v;
            if (v > best) { best = v; arg = c; }  // first maximum wins, like torch.argmax on CPU
        }
        if (a.out_is_i64) reinterpret_cast<long long *>(a.out_seg)[i] = arg;
        else reinterpret_cast<uint8_t *>(a.out_seg)[i] = (uint8_t)arg;
    }
}

// Fast path of the head for >= 3x upsampling (the bg net: logits at 1/4 resolution): one lane = 4 consecutive output
// pixels of a row.  Their taps fall on at most 3 consecutive source columns, so each channel costs 6 loads for 4
// outputs (the generic kernel: 16) and the labels leave as one 32-bit store.
__global__ __launch_bounds__(256) void head4_kernel(HeadArgs a) {
    const float sh = a.Hout > 1 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f;
    const float sw = a.Wout > 1 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const size_t opl = (size_t)a.Hout * a.Wout, ipl = (size_t)a.Hin * a.Win;
    const int W4 = a.Wout >> 2;
    const size_t total = (size_t)a.B * a.Hout * W4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xq = (int)(i % W4), x = xq * 4;
        const size_t r = i / W4;
        const int y = (int)(r % a.Hout), b = (int)(r / a.Hout);
        int y0, y1;
        float hy0, hy1;
        lin_coord(y, sh, a.Hin, y0, y1, hy0, hy1);
        int i0[4], i1[4];
        float l0[4], l1[4];
        int xa = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int x0, x1;
            lin_coord(x + k, sw, a.Win, x0, x1, l0[k], l1[k]);
            if (k == 0) xa = x0;
            i0[k] = x0 - xa;   // 0 or 1
            i1[k] = x1 - xa;   // 0, 1 or 2
        }
        const int xb = min(xa + 1, a.Win - 1), xc = min(xa + 2, a.Win - 1);
        const float *s = a.logits + (size_t)b * a.C * ipl;
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int arg[4] = {0, 0, 0, 0};
        for (int c = 0; c < a.C; ++c) {
            const float *r0 = s + (size_t)c * ipl + (size_t)y0 * a.Win, *r1 = s + (size_t)c * ipl + (size_t)y1 * a.Win;
            const float t0[3] = {r0[xa], r0[xb], r0[xc]}, t1[3] = {r1[xa], r1[xb], r1[xc]};
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float a0 = i0[k] ? t0[1] : t0[0], a1 = i1[k] == 2 ? t0[2] : (i1[k] ? t0[1] : t0[0]);
                const float b0 = i0[k] ? t1[1] : t1[0], b1 = i1[k] == 2 ? t1[2] : (i1[k] ? t1[1] : t1[0]);
                const float u0 = l0[k] * a0 + l1[k] * a1;
                const float u1 = l0[k] * b0 + l1[k] * b1;
                v[k] = hy0 * u0 + hy1 * u1;
                if (v[k] > best[k]) { best[k] = v[k]; arg[k] = c; }   // first maximum wins, like torch.argmax on CPU
            }
            if (a.out_logits)
                *reinterpret_cast<f32x4v *>(a.out_logits + ((size_t)b * a.C + c) * opl + (size_t)y * a.Wout + x) = f32x4v{v[0], v[1], v[2], v[3]};
        }
        const size_t o = (size_t)b * opl + (size_t)y * a.Wout + x;
        if (a.out_is_i64) {
            long long *d = reinterpret_cast<long long *>(a.out_seg) + o;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = arg[k];
        } else {
            *reinterpret_cast<unsigned *>(reinterpret_cast<uint8_t *>(a.out_seg) + o) =
                (unsigned)arg[0] | ((unsigned)arg[1] << 8) | ((unsigned)arg[2] << 16) | ((unsigned)arg[3] << 24);
        }
    }
}

// Column head.  head4_kernel issues 6 scalar loads per channel per lane - 66 load instructions for 4 output pixels - and is
// bound by the texture addresser (188 us for 16 frames at 0.7 TB/s); a first tiled version (round 1: source window of a
// 16 x 256 output tile staged in LDS, a lane = 4 consecutive pixels of a row) removed the loads but left ~165 vector
// instructions per output pixel - per pixel and channel six selects of the 3 source columns its 4 pixels straddle, two
// horizontal and one vertical interpolation, the argmax update: 165 us, vector-ALU-bound (profiles/README.md).
// Here a lane owns ONE output column of a 32 x 256 tile and walks down its 32 rows:
//   * its two source columns and their weights are fixed: no selects;
//   * the horizontal interpolation of a source row (per channel: 2 LDS reads, mul, fma) is done once and serves the ~4
//     output rows between two source rows - the rows a lane needs are wave-uniform, so "next source row" is a scalar branch;
//   * per pixel and channel there remain the vertical interpolation (mul, fma) and the argmax update (compare, 2 selects);
//   * the per-row coordinates (lin_coord) are computed once per workgroup by 32 lanes and read back as LDS broadcasts.
// Same operations on the same values as head_kernel: u = lx0*a + lx1*b per source row, v = hy0*u0 + hy1*u1.
constexpr int kHeadTH = 16, kHeadTW = 256;   // tile of seg_loss_tile_kernel below
constexpr int kHeadCH = 32, kHeadCW = 256;   // tile of head_col_kernel
template <int CC>
__global__ __launch_bounds__(256) void head_col_kernel(HeadArgs a) {
    extern __shared__ float hl[];            // [CC][rows][cols] source window
    __shared__ int row_r0[kHeadCH], row_r1[kHeadCH];
    __shared__ float row_h1[kHeadCH];
    const float sh = a.Hout > 1 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f;
    const float sw = a.Wout > 1 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const size_t opl = (size_t)a.Hout * a.Wout, ipl = (size_t)a.Hin * a.Win;
    const int ty0 = blockIdx.y * kHeadCH, tx0 = blockIdx.x * kHeadCW, b = blockIdx.z;
    const int ylast = min(ty0 + kHeadCH, a.Hout) - 1, xlast = min(tx0 + kHeadCW, a.Wout) - 1;
    const int sy0 = min((int)(sh * (float)ty0), a.Hin - 1), sx0 = min((int)(sw * (float)tx0), a.Win - 1);
    const int rows = min(min((int)(sh * (float)ylast), a.Hin - 1) + 1, a.Hin - 1) - sy0 + 1;
    const int cols = min(min((int)(sw * (float)xlast), a.Win - 1) + 1, a.Win - 1) - sx0 + 1;
    const int per = rows * cols;
    const float *src = a.logits + (size_t)b * CC * ipl;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int r = e / cols, x = e - r * cols;
        const size_t off = (size_t)(sy0 + r) * a.Win + sx0 + x;
#pragma unroll
        for (int c = 0; c < CC; ++c) hl[c * per + e] = src[(size_t)c * ipl + off];
    }
    if (threadIdx.x < kHeadCH) {
        int y0, y1;
        float hy0, hy1;
        lin_coord(min(ty0 + (int)threadIdx.x, a.Hout - 1), sh, a.Hin, y0, y1, hy0, hy1);
        row_r0[threadIdx.x] = y0 - sy0;
        row_r1[threadIdx.x] = y1 - sy0;
        row_h1[threadIdx.x] = hy1;
    }
    __syncthreads();
    const int x = tx0 + threadIdx.x;
    if (x >= a.Wout) return;
    int x0, x1;
    float lx0, lx1;
    lin_coord(x, sw, a.Win, x0, x1, lx0, lx1);
    const float *p0 = hl + (x0 - sx0), *p1 = hl + (x1 - sx0);
    float tA[CC], tB[CC];                    // horizontally interpolated source rows curA (= y0) and curB (= y1)
    int curA = -1, curB = -1;                // wave-uniform
    auto hrow = [&](int r, float (&t)[CC]) {
#pragma unroll
        for (int c = 0; c < CC; ++c) t[c] = lx0 * p0[c * per + r * cols] + lx1 * p1[c * per + r * cols];
    };
    for (int ry = 0; ry < kHeadCH; ++ry) {
        const int y = ty0 + ry;
        if (y >= a.Hout) break;
        const int r0 = __builtin_amdgcn_readfirstlane(row_r0[ry]), r1 = __builtin_amdgcn_readfirstlane(row_r1[ry]);
        const float hy1 = row_h1[ry], hy0 = 1.f - hy1;
        if (r0 != curA) {
            if (r0 == curB) {
#pragma unroll
                for (int c = 0; c < CC; ++c) tA[c] = tB[c];
            } else {
                hrow(r0, tA);
            }
            curA = r0;
        }
        if (r1 != curB) {
            if (r1 == curA) {
#pragma unroll
                for (int c = 0; c < CC; ++c) tB[c] = tA[c];
            } else {
                hrow(r1, tB);
            }
            curB = r1;
        }
        float best = -INFINITY;
        int arg = 0;
        const size_t o = (size_t)b * opl + (size_t)y * a.Wout + x;
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float v = hy0 * tA[c] + hy1 * tB[c];
            if (a.out_logits) a.out_logits[((size_t)b * CC + c) * opl + (size_t)y * a.Wout + x] = v;
            if (v > best) { best = v; arg = c; }   // first maximum wins, like torch.argmax on CPU
        }
        if (a.out_is_i64) reinterpret_cast<long long *>(a.out_seg)[o] = arg;
        else reinterpret_cast<uint8_t *>(a.out_seg)[o] = (uint8_t)arg;
    }
}

// ------------------------------------------------------------------------------------------------
// Validation loss of the bg model (scope row f4, forward part): BGModel.loss (bg_model.py:73-89) =
// nn.CrossEntropyLoss(ignore_index=255)(logits, labels) + accuracy, with logits = the bilinearly upsampled network output
// (hardnet.py:372-384).  Fused like the head: a workgroup stages the low-resolution logits of its 16 x 256 output tile in
// LDS, every lane interpolates the C logits of its pixels, forms log-sum-exp, the negative log-likelihood of the label
// and "argmax == label" - the full-resolution logits (92 MB per frame) are never written.  Per-workgroup partial sums
// (fp64) go to the workspace and a second one-block kernel adds them in index order: deterministic.
struct LossArgs {
    const float *logits;   // [B,C,Hin,Win]
    const void *labels;    // [B,Hout,Wout] i64 or u8
    double *partial;       // [blocks][3]: sum of nll, valid pixels, correct pixels
    int labels_i64, B, Hin, Win, Hout, Wout, ignore;
};

template <int CC>
__global__ __launch_bounds__(256) void seg_loss_tile_kernel(LossArgs a) {
    extern __shared__ float hl[];
    __shared__ double red[4][3];
    const float sh = a.Hout > 1 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f;
    const float sw = a.Wout > 1 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const size_t opl = (size_t)a.Hout * a.Wout, ipl = (size_t)a.Hin * a.Win;
    const int ty0 = blockIdx.y * kHeadTH, tx0 = blockIdx.x * kHeadTW, b = blockIdx.z;
    const int ylast = min(ty0 + kHeadTH, a.Hout) - 1, xlast = min(tx0 + kHeadTW, a.Wout) - 1;
    const int sy0 = min((int)(sh * (float)ty0), a.Hin - 1), sx0 = min((int)(sw * (float)tx0), a.Win - 1);
    const int rows = min(min((int)(sh * (float)ylast), a.Hin - 1) + 1, a.Hin - 1) - sy0 + 1;
    const int cols = min(min((int)(sw * (float)xlast), a.Win - 1) + 1, a.Win - 1) - sx0 + 1;
    const int per = rows * cols;
    const float *src = a.logits + (size_t)b * CC * ipl;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int r = e / cols, x = e - r * cols;
        const size_t off = (size_t)(sy0 + r) * a.Win + sx0 + x;
#pragma unroll
        for (int c = 0; c < CC; ++c) hl[c * per + e] = src[(size_t)c * ipl + off];
    }
    __syncthreads();
    double nll = 0.0, valid = 0.0, correct = 0.0;
    const int x = tx0 + threadIdx.x;
    if (x < a.Wout) {
        int x0, x1;
        float lx0, lx1;
        lin_coord(x, sw, a.Win, x0, x1, lx0, lx1);
        for (int ry = 0; ry < kHeadTH; ++ry) {
            const int y = ty0 + ry;
            if (y >= a.Hout) break;
            const size_t o = (size_t)b * opl + (size_t)y * a.Wout + x;
            const int lab = a.labels_i64 ? (int)reinterpret_cast<const long long *>(a.labels)[o] : (int)reinterpret_cast<const uint8_t *>(a.labels)[o];
            int y0, y1;
            float hy0, hy1;
            lin_coord(y, sh, a.Hin, y0, y1, hy0, hy1);
            const float *r0 = hl + (y0 - sy0) * cols - sx0, *r1 = hl + (y1 - sy0) * cols - sx0;
            float v[CC], best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float t0 = lx0 * r0[c * per + x0] + lx1 * r0[c * per + x1];
                const float t1 = lx0 * r1[c * per + x0] + lx1 * r1[c * per + x1];
                v[c] = hy0 * t0 + hy1 * t1;
                if (v[c] > best) { best = v[c]; arg = c; }
            }
            if (lab == a.ignore || lab < 0 || lab >= CC) {
                // ignored pixel: no loss term, not in the accuracy denominator; a prediction never equals 255
                continue;
            }
            float se = 0.f, vl = 0.f;
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                se += expf(v[c] - best);
                vl = c == lab ? v[c] : vl;
            }
            nll += (double)((best + logf(se)) - vl);
            valid += 1.0;
            correct += arg == lab ? 1.0 : 0.0;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        nll += __shfl_xor(nll, o);
        valid += __shfl_xor(valid, o);
        correct += __shfl_xor(correct, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = nll; red[threadIdx.x >> 6][1] = valid; red[threadIdx.x >> 6][2] = correct;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double *p = a.partial + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 3;
        p[0] = ((red[0][0] + red[1][0]) + red[2][0]) + red[3][0];
        p[1] = ((red[0][1] + red[1][1]) + red[2][1]) + red[3][1];
        p[2] = ((red[0][2] + red[1][2]) + red[2][2]) + red[3][2];
    }
}

__global__ __launch_bounds__(256) void seg_loss_finish_kernel(const double *partial, int n, double *out3) {
    __shared__ double red[256][3];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { s0 += partial[i * 3]; s1 += partial[i * 3 + 1]; s2 += partial[i * 3 + 2]; }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 256; ++i) { s0 += red[i][0]; s1 += red[i][1]; s2 += red[i][2]; }
        out3[0] = s0; out3[1] = s1; out3[2] = s2;
    }
}

// ------------------------------------------------------------------------------------------------
static unsigned grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// Is  q = x * r;  q += fma(-q, std, x) * r  (r = 1 / std rounded to fp32) the correctly rounded x / std for EVERY x the
// in-register depth hop can produce?  The hop quantises depth to a u16 code c in 0..65280 (round(clamp(d + 1, 0, 255) * 256)),
// so x = clamp(c / 256 - 1) - mean takes at most 65 281 values: all of them are tried here (fmaf of the host libm is exact).
// Cached for the last parameter set.
static bool stem_fast_div_exact(float mean, float stdv, float dmin, float dmax) {
    static float key[4] = {0.f, 0.f, 0.f, 0.f};
    static int cached = -1;
    if (cached >= 0 && key[0] == mean && key[1] == stdv && key[2] == dmin && key[3] == dmax) return cached != 0;
    bool ok = std::isfinite(stdv) && stdv != 0.f;
    const volatile float r = 1.0f / stdv;
    for (int c = 0; ok && c <= 65280; ++c) {
        float d = (float)c / 256.f - 1.f;
        d = d > 0.f ? fminf(fmaxf(d, dmin), dmax) : -1.f;
        const volatile float x = d - mean;
        const volatile float q0 = x * r;
        const volatile float e = fmaf(-q0, stdv, x);
        const volatile float q1 = fmaf(e, r, q0);
        const volatile float want = x / stdv;
        const float got = q1, w = want;
        ok = memcmp(&got, &w, 4) == 0;
    }
    key[0] = mean; key[1] = stdv; key[2] = dmin; key[3] = dmax;
    cached = ok ? 1 : 0;
    return ok;
}

// the conditions of the 2 x 2-outputs-per-lane kernel (the only one that writes the packed-pair layout)
bool stem_writes_s4(const StemArgs &a) {
    return a.T == 3 && a.wdep && a.woh && !a.seg_is_i64 && (a.W & 3) == 0 && (a.H & 3) == 0 && a.Wout * 2 == a.W && a.Hout * 2 == a.H &&
           (unsigned long long)a.T * a.H * a.W < (1ull << 32) && !ab_env("PF_STEM_GENERIC") && !ab_env("PF_STEM_BATCHED") && !ab_env("PF_STEM_V3");
}

int launch_stem(const StemArgs &a, hipStream_t s) {
    if (a.T * (a.n_cls + 1) * 9 * 16 * sizeof(float) > 60000)
        return fail(PF_EUNSUPPORTED, "stem: T=%d n_cls=%d weights exceed LDS budget", a.T, a.n_cls);
    if (a.dst_fmt && !stem_writes_s4(a)) return fail(PF_EUNSUPPORTED, "stem: these arguments select a kernel that writes fp32 only");
    const size_t lds = (size_t)a.T * (a.n_cls + 1) * 9 * 16 * sizeof(float);
    const double ipx = (double)a.B * a.T * a.H * a.W, opx = (double)a.B * a.Hout * a.Wout;
    static const bool generic = ab_env("PF_STEM_GENERIC") != nullptr;   // A/B switch for profiling
    const bool batched = a.T == 3 && a.wdep && !generic;
    static const bool no_v3 = ab_env("PF_STEM_BATCHED") != nullptr;       // A/B switch: the previous form
    const bool v3 = batched && a.woh && !no_v3 && (unsigned long long)a.T * a.H * a.W < (1ull << 32);
    static const bool no_v4 = ab_env("PF_STEM_V3") != nullptr;            // A/B switch: one output per lane
    const bool v4 = v3 && !no_v4 && !a.seg_is_i64 && (a.W & 3) == 0 && (a.H & 3) == 0 && a.Wout * 2 == a.W && a.Hout * 2 == a.H;
    const char *label = !batched ? "pf::stem_onehot_kernel(pf::StemArgs)"
                        : v4 ? "pf::stem_onehot_v4_kernel(pf::StemArgs, float)"
                        : v3 ? "pf::stem_onehot_v3_kernel(pf::StemArgs)"
                        : a.seg_is_i64 ? "void pf::stem_onehot_batched_kernel<3, true>(pf::StemArgs)"
                                       : "void pf::stem_onehot_batched_kernel<3, false>(pf::StemArgs)";
    ProfScope ps(s, label, 2.0 * opx * 16 * a.T * (a.n_cls + 1) * 9,
                 ipx * ((a.seg_is_i64 ? 8 : 1) + 4 + ((a.hop & PF_HOP_DEPTH_U16) ? 0 : 1)) + opx * 16 * 4);
    const dim3 grid((a.Wout + 63) / 64, (a.Hout + 3) / 4, a.B);
    if (v4) {
        const size_t lds4 = (size_t)9 * a.T * (a.n_cls + 1) * 16 * sizeof(float);
        const bool hd = (a.hop & PF_HOP_DEPTH_U16) != 0, hl = (a.hop & PF_HOP_TRAINID_LUT) != 0;
        // the reciprocal form of the division is exact only on the value set of the hop (proven per parameter set)
        const bool fast = hd && stem_fast_div_exact(a.depth_mean, a.depth_std, a.min_depth, a.max_depth);
        const float inv_std = 1.0f / a.depth_std;
        const dim3 grid4((a.Wout + 127) / 128, (a.Hout + 7) / 8, a.B);
        const int variant = (hd ? 4 : 0) | (hl ? 2 : 0) | (fast ? 1 : 0);
#define PF_STEM4(V, HD, HL, FD) \
        if (variant == V) hipLaunchKernelGGL((stem_onehot_v4_kernel<3, HD, HL, FD>), grid4, dim3(256), lds4, s, a, inv_std);
        PF_STEM4(0, false, false, false) PF_STEM4(2, false, true, false)
        PF_STEM4(4, true, false, false) PF_STEM4(5, true, false, true) PF_STEM4(6, true, true, false) PF_STEM4(7, true, true, true)
#undef PF_STEM4
    } else if (v3) {
        const size_t lds3 = (size_t)9 * a.T * (a.n_cls + 1) * kStemRow * sizeof(float);
        const int variant = (a.seg_is_i64 ? 4 : 0) | ((a.hop & PF_HOP_DEPTH_U16) ? 2 : 0) | ((a.hop & PF_HOP_TRAINID_LUT) ? 1 : 0);
#define PF_STEM3(V, S64, HD, HL) \
        if (variant == V) hipLaunchKernelGGL((stem_onehot_v3_kernel<3, S64, HD, HL>), grid, dim3(256), lds3, s, a);
        PF_STEM3(0, false, false, false) PF_STEM3(1, false, false, true) PF_STEM3(2, false, true, false) PF_STEM3(3, false, true, true)
        PF_STEM3(4, true, false, false) PF_STEM3(5, true, false, true) PF_STEM3(6, true, true, false) PF_STEM3(7, true, true, true)
#undef PF_STEM3
    } else if (batched) {
        if (a.seg_is_i64) hipLaunchKernelGGL((stem_onehot_batched_kernel<3, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((stem_onehot_batched_kernel<3, false>), grid, dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL(stem_onehot_kernel, grid, dim3(256), lds, s, a);
    }
    PF_LAUNCH_CHECK("stem_onehot_kernel");
    return PF_OK;
}

int launch_avgpool2(const float *src, float *dst, int planes, int Hin, int Win, unsigned *status, unsigned *slot, hipStream_t s) {
    const int Ho = Hin / 2, Wo = Win / 2;
    ProfScope ps(s, "pf::avgpool2_kernel", 0, 4.0 * planes * ((double)Hin * Win + (double)Ho * Wo));
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((size_t)planes * Ho * Wo)), dim3(256), 0, s, src, dst, planes,
                       Hin, Win, Ho, Wo, status, slot);
    PF_LAUNCH_CHECK("avgpool2_kernel");
    return PF_OK;
}

int launch_upsample(const float *src, float *dst, int planes, int Hin, int Win, int Hout, int Wout, unsigned *status, unsigned *slot,
                    hipStream_t s) {
    ProfScope ps(s, "pf::upsample_kernel", 0, 4.0 * planes * ((double)Hin * Win + (double)Hout * Wout));
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for((size_t)planes * Hout * Wout)), dim3(256), 0, s, src, dst,
                       planes, Hin, Win, Hout, Wout, status, slot);
    PF_LAUNCH_CHECK("upsample_kernel");
    return PF_OK;
}

int launch_head(const HeadArgs &a, hipStream_t s) {
    const double opx = (double)a.B * a.Hout * a.Wout;
    ProfScope ps(s, "pf::head_col_kernel(pf::HeadArgs)", 0, 4.0 * a.B * a.C * a.Hin * a.Win + opx * (a.out_is_i64 ? 8 : 1) +
                                                          (a.out_logits ? opx * a.C * 4 : 0));
    const float sw = a.Wout > 1 ? (float)(a.Win - 1) / (float)(a.Wout - 1) : 0.f;
    const float shh = a.Hout > 1 ? (float)(a.Hin - 1) / (float)(a.Hout - 1) : 0.f;
    const size_t win = ((size_t)(shh * (kHeadCH - 1)) + 3) * ((size_t)(sw * (kHeadCW - 1)) + 3) * 4;   // window bound of one tile, per channel
    static const bool no_tile = ab_env("PF_HEAD_UNTILED") != nullptr;   // A/B switch
    if ((a.C == 11 || a.C == 19) && a.Hin >= 2 && a.Win >= 2 && win * a.C <= 60 * 1024 && !no_tile) {
        const dim3 grid((a.Wout + kHeadCW - 1) / kHeadCW, (a.Hout + kHeadCH - 1) / kHeadCH, a.B);
        static bool attr = false;
        if (!attr) {
            PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&head_col_kernel<11>), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024));
            PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&head_col_kernel<19>), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024));
            attr = true;
        }
        if (a.C == 11) hipLaunchKernelGGL(head_col_kernel<11>, grid, dim3(256), win * 11, s, a);
        else hipLaunchKernelGGL(head_col_kernel<19>, grid, dim3(256), win * 19, s, a);
    } else if ((a.Wout & 3) == 0 && 3.f * sw < 1.f && a.Win >= 3)   // 4 consecutive outputs span <= 2 source columns
        hipLaunchKernelGGL(head4_kernel, dim3(grid_for((size_t)a.B * a.Hout * (a.Wout >> 2))), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(head_kernel, dim3(grid_for((size_t)a.B * a.Hout * a.Wout)), dim3(256), 0, s, a);
    PF_LAUNCH_CHECK("head_kernel");
    return PF_OK;
}

}  // namespace pf

extern "C" int pf_seg_loss_workspace(int B, int out_h, int out_w, size_t *bytes) {
    if (!bytes || B <= 0 || out_h <= 0 || out_w <= 0) return pf::fail(PF_EINVAL, "pf_seg_loss_workspace: bad arguments");
    const size_t blocks = (size_t)B * ((out_h + pf::kHeadTH - 1) / pf::kHeadTH) * ((out_w + pf::kHeadTW - 1) / pf::kHeadTW);
    *bytes = pf::align_up(blocks * 3 * sizeof(double), 256);
    return PF_OK;
}

extern "C" int pf_seg_loss(const float *logits, int B, int C, int Hin, int Win, const void *labels, int labels_i64, int out_h,
                           int out_w, int ignore_index, double *out3, void *ws, size_t ws_bytes, void *stream) {
    if (!logits || !labels || !out3 || !ws || B <= 0 || Hin <= 0 || Win <= 0 || out_h <= 0 || out_w <= 0)
        return pf::fail(PF_EINVAL, "pf_seg_loss: bad arguments");
    if (C != 11 && C != 19) return pf::fail(PF_EUNSUPPORTED, "pf_seg_loss: built for 11 or 19 classes, got %d", C);
    size_t need = 0;
    pf_seg_loss_workspace(B, out_h, out_w, &need);
    if (ws_bytes < need) return pf::fail(PF_EWORKSPACE, "pf_seg_loss: workspace %zu B < required %zu B", ws_bytes, need);
    const float sh = out_h > 1 ? (float)(Hin - 1) / (float)(out_h - 1) : 0.f, sw = out_w > 1 ? (float)(Win - 1) / (float)(out_w - 1) : 0.f;
    const int wr = (int)(sh * (pf::kHeadTH - 1)) + 3, wc = (int)(sw * (pf::kHeadTW - 1)) + 3;
    const size_t lds = (size_t)wr * wc * C * sizeof(float);
    if (lds > 60 * 1024) return pf::fail(PF_EUNSUPPORTED, "pf_seg_loss: source window of %zu B per tile does not fit LDS (downsampling heads are not built)", lds);
    pf::LossArgs a{logits, labels, (double *)ws, labels_i64 ? 1 : 0, B, Hin, Win, out_h, out_w, ignore_index};
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((out_w + pf::kHeadTW - 1) / pf::kHeadTW, (out_h + pf::kHeadTH - 1) / pf::kHeadTH, B);
    {
        pf::ProfScope ps(s, C == 11 ? "void pf::seg_loss_tile_kernel<11>(pf::LossArgs)" : "void pf::seg_loss_tile_kernel<19>(pf::LossArgs)", 0.0,
                         4.0 * B * C * Hin * Win + (double)B * out_h * out_w * (labels_i64 ? 8 : 1));
        if (C == 11) {
            static bool set11 = false;
            if (!set11) { PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pf::seg_loss_tile_kernel<11>), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024)); set11 = true; }
            hipLaunchKernelGGL(pf::seg_loss_tile_kernel<11>, grid, dim3(256), lds, s, a);
        } else {
            static bool set19 = false;
            if (!set19) { PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&pf::seg_loss_tile_kernel<19>), hipFuncAttributeMaxDynamicSharedMemorySize, 60 * 1024)); set19 = true; }
            hipLaunchKernelGGL(pf::seg_loss_tile_kernel<19>, grid, dim3(256), lds, s, a);
        }
        PF_LAUNCH_CHECK("seg_loss_tile_kernel");
    }
    hipLaunchKernelGGL(pf::seg_loss_finish_kernel, dim3(1), dim3(256), 0, s, (const double *)ws, (int)(grid.x * grid.y * grid.z), out3);
    PF_LAUNCH_CHECK("seg_loss_finish_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ dense network input
// bg_model.py:61-69 for the configurations the fused stem does not cover (convert2onehot = False: frames that are images or
// already one-hot; or no depth channels): x[b, t * C + c] = frame channel c (labels: (label == c), labels >= n_cls -> zero
// vector, :53-59), then - with depth - x[b, T * C + t] = (depth - mean) / std * mask (IEEE division, like ATen).  Round 6: was ATen glue.
namespace pf {
template <int KIND>   // 0: u8 labels, 1: i64 labels, 2: f32 frames [B][T][C][H][W]
__global__ __launch_bounds__(256) void dense_input_kernel(const void *frames, int C, const float *depth, const uint8_t *mask, float mean,
                                                          float stdv, int T, long long HW, float *x) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    if (i >= HW) return;
    const int Cx = T * C + (depth ? T : 0);
    float *xb = x + (long long)b * Cx * HW + i;
    if (KIND == 2) {
        const float *f = reinterpret_cast<const float *>(frames) + (long long)bt * C * HW + i;
        for (int c = 0; c < C; ++c) xb[(long long)(t * C + c) * HW] = f[(long long)c * HW];
    } else {
        const long long in = (long long)bt * HW + i;
        const long long lab = KIND == 1 ? reinterpret_cast<const long long *>(frames)[in] : (long long)reinterpret_cast<const uint8_t *>(frames)[in];
        for (int c = 0; c < C; ++c) xb[(long long)(t * C + c) * HW] = (lab == c) ? 1.f : 0.f;
    }
    if (depth) {
        const long long in = (long long)bt * HW + i;
        xb[(long long)(T * C + t) * HW] = ((depth[in] - mean) / stdv) * (mask[in] ? 1.f : 0.f);
    }
}
}  // namespace pf

extern "C" int pf_bg_dense_input(const void *frames, int kind, int channels, const float *depth, const uint8_t *depth_mask, float depth_mean,
                                 float depth_std, int B, int T, int H, int W, float *x, void *stream) {
    if (!frames || !x || B <= 0 || T <= 0 || H <= 0 || W <= 0 || channels <= 0 || kind < 0 || kind > 2 || (depth && !depth_mask))
        return pf::fail(PF_EINVAL, "pf_bg_dense_input: bad argument (kind %d, channels %d, B %d T %d %dx%d)", kind, channels, B, T, H, W);
    const long long HW = (long long)H * W;
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(B * T));
    hipStream_t s = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(pf::dense_input_kernel<0>, grid, dim3(256), 0, s, frames, channels, depth, depth_mask, depth_mean, depth_std, T, HW, x);
    else if (kind == 1) hipLaunchKernelGGL(pf::dense_input_kernel<1>, grid, dim3(256), 0, s, frames, channels, depth, depth_mask, depth_mean, depth_std, T, HW, x);
    else hipLaunchKernelGGL(pf::dense_input_kernel<2>, grid, dim3(256), 0, s, frames, channels, depth, depth_mask, depth_mean, depth_std, T, HW, x);
    PF_LAUNCH_CHECK("dense_input_kernel");
    return PF_OK;
}

