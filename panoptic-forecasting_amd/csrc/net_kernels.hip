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
#include "net_kernels.h"
#include "conv_epilogue.h"
#include "pf_prof.h"

namespace pf {

// ------------------------------------------------------------------------------------------------
// Stem: one workgroup = 4 x 64 output pixels x 16 channels.
//   stage 1  the 9 x 129 input window of each of the T frames is read ONCE, coalesced (16 B of depth, 4 B of labels
//            per lane), turned into {class index or -1, normalised masked depth} (the hop emulation runs here, once
//            per input pixel) and stored in LDS de-interleaved by column parity, so that stage 2's stride-2 window
//            reads are consecutive across lanes (no bank conflicts);
//   stage 2  one lane = one output pixel: per tap and frame one class byte + one depth float from LDS; the one-hot
//            part is a 16-float weight row gathered from LDS (4 x ds_read_b128), the depth part is 16 FMAs whose
//            weights come through the scalar cache (uniform address -> s_load, SGPR operands).
// HBM traffic is the algorithmic minimum (inputs once, output once); profiles/r01_c had it at 5x that time.
constexpr int kStemTH = 8, kStemTW = 64;   // 2 output rows per lane
constexpr int kStemRows = 2 * kStemTH + 1;        // input rows per tile
constexpr int kStemCols = kStemTW + 4;            // padded columns per parity (even: 64 used, odd: 65 used)
constexpr int kStemRowF = 20;                     // floats between one-hot weight rows in LDS

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE
#define SPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == a.dbg_plane_pad % 100 && blockIdx.y == a.dbg_plane_pad / 100 && blockIdx.z == 0 && a.probe) { a.probe[i] = clock64(); a.probe[i + 8] = wall_clock64(); } } while (0)
#else
#define SPROBE(i) do { } while (0)
#endif

template <int T>
__global__ __launch_bounds__(256) void stem_onehot_kernel(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    SPROBE(0);
#if PF_PROBE
    const long long wg_t0 = wall_clock64();
#endif
    const int in_ch = T * (a.n_cls + 1);
    // one-hot weight rows [tap][t][class 0..n_cls][kStemRowF]: row n_cls is zeros (labels >= n_cls / padding select
    // it: no branch in the tap loop); rows are 20 floats apart so that 16 lanes reading 16 different classes with
    // ds_read_b128 touch 16 different 4-bank slots (5 is coprime with 16)
    const int nrow = a.n_cls + 1;
    float *wl = smem;
    float *dn_s = smem + 9 * T * nrow * kStemRowF;                      // [T][rows][parity][cols]
    int8_t *cls_s = reinterpret_cast<int8_t *>(dn_s + T * kStemRows * 2 * kStemCols);
    __shared__ uint8_t lut[256];

    const int tid = threadIdx.x;
    for (int e = tid; e < 9 * T * nrow * 16; e += 256) {
        const int co = e & 15, row = e >> 4;                       // row = (tap*T + t)*nrow + cls
        const int cls = row % nrow, t = (row / nrow) % T, tap = row / (nrow * T);
        wl[row * kStemRowF + co] = cls < a.n_cls ? a.w[((size_t)co * in_ch + t * a.n_cls + cls) * 9 + tap] : 0.f;
    }
    lut[tid] = (a.hop & PF_HOP_TRAINID_LUT) ? a.lut[tid] : (uint8_t)tid;
    __syncthreads();
    SPROBE(1);

    const int ox0 = blockIdx.x * kStemTW, oy0 = blockIdx.y * kStemTH, b = blockIdx.z;
    const size_t N = (size_t)a.H * a.W;
    const int ix0 = 2 * ox0 - 4, iy0 = 2 * oy0 - 1;      // window origin (ix0 is a multiple of 4)
    constexpr int NV = (2 * kStemTW + 8) / 4;            // float4 pieces per window row (34)
    const bool vec = (a.W & 3) == 0;

    // ---- stage 1
    for (int e = tid; e < T * kStemRows * NV; e += 256) {
        const int t = e / (kStemRows * NV), rem = e - t * (kStemRows * NV);
        const int r = rem / NV, i = rem - r * NV;
        const int gy = iy0 + r, gx = ix0 + 4 * i;
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        int lab[4] = {-1, -1, -1, -1};
        bool m[4] = {false, false, false, false};
        if (gy >= 0 && gy < a.H && gx + 3 >= 0 && gx < a.W) {
            const size_t idx = ((size_t)b * T + t) * N + (size_t)gy * a.W + gx;
            if (vec && gx >= 0 && gx + 3 < a.W) {
                const float4 dv = *reinterpret_cast<const float4 *>(a.depth + idx);
                d[0] = dv.x; d[1] = dv.y; d[2] = dv.z; d[3] = dv.w;
                if (a.seg_is_i64) {
                    const longlong2 s0 = reinterpret_cast<const longlong2 *>(reinterpret_cast<const long long *>(a.seg) + idx)[0];
                    const longlong2 s1 = reinterpret_cast<const longlong2 *>(reinterpret_cast<const long long *>(a.seg) + idx)[1];
                    lab[0] = (int)s0.x; lab[1] = (int)s0.y; lab[2] = (int)s1.x; lab[3] = (int)s1.y;
                } else {
                    const uchar4 sv = *reinterpret_cast<const uchar4 *>(reinterpret_cast<const uint8_t *>(a.seg) + idx);
                    lab[0] = sv.x; lab[1] = sv.y; lab[2] = sv.z; lab[3] = sv.w;
                }
                if (!(a.hop & PF_HOP_DEPTH_U16)) {
                    const uchar4 mv = *reinterpret_cast<const uchar4 *>(a.mask + idx);
                    m[0] = mv.x != 0; m[1] = mv.y != 0; m[2] = mv.z != 0; m[3] = mv.w != 0;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (gx + k < 0 || gx + k >= a.W) continue;
                    d[k] = a.depth[idx + k];
                    lab[k] = a.seg_is_i64 ? (int)reinterpret_cast<const long long *>(a.seg)[idx + k]
                                          : (int)reinterpret_cast<const uint8_t *>(a.seg)[idx + k];
                    if (!(a.hop & PF_HOP_DEPTH_U16)) m[k] = a.mask[idx + k] != 0;
                }
            }
        }
        float *drow = dn_s + ((t * kStemRows + r) * 2) * kStemCols;
        int8_t *crow = cls_s + ((t * kStemRows + r) * 2) * kStemCols;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool inside = gy >= 0 && gy < a.H && gx + k >= 0 && gx + k < a.W;
            int cls = lab[k];
            if (a.hop & PF_HOP_TRAINID_LUT) cls = lut[cls & 255];
            if (!inside || cls < 0 || cls >= a.n_cls) cls = a.n_cls;  // labels >= n_cls: zero vector (bg_model.py:54-57)
            float dd = d[k], mm;
            if (a.hop & PF_HOP_DEPTH_U16) {
                const float q = rintf(fminf(fmaxf(dd + 1.f, 0.f), 255.f) * 256.f);   // export :119-124
                dd = q / 256.f - 1.f;                                                  // load bg_dataset.py:225
                const bool mk = dd > 0.f;
                dd = mk ? fminf(fmaxf(dd, a.min_depth), a.max_depth) : -1.f;            // :227-228,:166-170
                mm = mk ? 1.f : 0.f;
            } else {
                mm = m[k] ? 1.f : 0.f;
            }
            const float dn = inside ? ((dd - a.depth_mean) / a.depth_std) * mm : 0.f;   // bg_model.py:50-51,66-67
            // window column c = 4*i + k (absolute column ix0 + c): even columns -> j = (c-4)/2, odd -> j = (c-3)/2
            const int c = 4 * i + k, par = c & 1, j = (c - 3 - (1 - par)) >> 1;        // valid j: even 0..63, odd 0..64
            if (j >= 0 && j < kStemCols) {
                drow[par * kStemCols + j] = dn;
                crow[par * kStemCols + j] = (int8_t)cls;
            }
        }
    }
    __syncthreads();
    SPROBE(2);

    // ---- stage 2: lane = output column lx, rows 2*lyp and 2*lyp+1 of the tile
    const int lx = tid & 63, lyp = tid >> 6;
    float acc[2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[p][i] = a.bias[i];
    // (t, ky) loops stay rolled: fully unrolled and branch-free, hipcc hoists all 54 gathers and spills 200 VGPRs
#pragma unroll 1
    for (int t = 0; t < T; ++t) {
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tap = ky * 3 + kx;
                // depth column of this (tap, frame): 16 contiguous floats at a uniform address -> one s_load_dwordx16
                const float *wd = a.wdep + (tap * T + t) * 16;
                // input column 2*ox - 1 + kx: kx = 1 is even (j = lx), kx = 0 / 2 are odd (j = lx, lx + 1)
                const int par = kx == 1 ? 0 : 1, j = lx + (kx == 2 ? 1 : 0);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int r = 2 * (2 * lyp + p) + ky;
                    const int o = ((t * kStemRows + r) * 2 + par) * kStemCols + j;
                    const int cls = cls_s[o];
                    const float dn = dn_s[o];
                    const f32x4v *row = reinterpret_cast<const f32x4v *>(wl + ((tap * T + t) * nrow + cls) * kStemRowF);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4v w4 = row[q];
                        acc[p][q * 4 + 0] += w4[0]; acc[p][q * 4 + 1] += w4[1];
                        acc[p][q * 4 + 2] += w4[2]; acc[p][q * 4 + 3] += w4[3];
                    }
#pragma unroll
                    for (int co = 0; co < 16; ++co) acc[p][co] += wd[co] * dn;
                }
            }
        }
    }
    SPROBE(3);
    const int ox = ox0 + lx;
    const size_t op = (size_t)a.Hout * a.Wout;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int oy = oy0 + 2 * lyp + p;
        if (ox < a.Wout && oy < a.Hout) {
            float *o = a.dst + (size_t)b * 16 * op + (size_t)oy * a.Wout + ox;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[i * op] = fmaxf(acc[p][i], 0.f);
        }
    }
    SPROBE(4);
#if PF_PROBE
    if (threadIdx.x == 0 && a.probe) {   // whole-kernel view: first start, last end, summed workgroup time (100 MHz ticks)
        const long long t1 = wall_clock64();
        atomicMin((unsigned long long *)&a.probe[60], (unsigned long long)wg_t0);
        atomicMax((unsigned long long *)&a.probe[61], (unsigned long long)t1);
        atomicAdd((unsigned long long *)&a.probe[62], (unsigned long long)(t1 - wg_t0));
        atomicAdd((unsigned long long *)&a.probe[63], 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool2_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       int planes, int Hin, int Win, int Hout, int Wout) {
    const size_t total = (size_t)planes * Hout * Wout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % Wout);
        const size_t r = i / Wout;
        const int y = (int)(r % Hout);
        const size_t p = r / Hout;
        const float *s = src + (p * Hin + 2 * y) * (size_t)Win + 2 * x;
        dst[i] = (((s[0] + s[1]) + s[Win]) + s[Win + 1]) * 0.25f;
    }
}

__global__ __launch_bounds__(256) void upsample_kernel(const float *__restrict__ src, float *__restrict__ dst,
                                                       int planes, int Hin, int Win, int Hout, int Wout) {
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
        dst[i] = hy0 * t0 + hy1 * t1;
    }
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

// ------------------------------------------------------------------------------------------------
static unsigned grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

int launch_stem(const StemArgs &a, hipStream_t s) {
    if (a.T < 1 || a.T > 4 || a.n_cls > 126)
        return fail(PF_EUNSUPPORTED, "stem: T=%d n_cls=%d (supported: 1..4 frames, <= 126 classes)", a.T, a.n_cls);
    const size_t lds = (size_t)9 * a.T * (a.n_cls + 1) * kStemRowF * sizeof(float) +
                       (size_t)a.T * kStemRows * 2 * kStemCols * (sizeof(float) + 1);
    if (lds > 64 * 1024) return fail(PF_EUNSUPPORTED, "stem: T=%d n_cls=%d needs %zu B of LDS", a.T, a.n_cls, lds);
    const double ipx = (double)a.B * a.T * a.H * a.W, opx = (double)a.B * a.Hout * a.Wout;
    ProfScope ps(s, "pf::stem_onehot_kernel(pf::StemArgs)", 2.0 * opx * 16 * a.T * (a.n_cls + 1) * 9,
                 ipx * ((a.seg_is_i64 ? 8 : 1) + 4 + ((a.hop & PF_HOP_DEPTH_U16) ? 0 : 1)) + opx * 16 * 4);
    const dim3 grid((a.Wout + kStemTW - 1) / kStemTW, (a.Hout + kStemTH - 1) / kStemTH, a.B);
    switch (a.T) {
        case 1: hipLaunchKernelGGL(stem_onehot_kernel<1>, grid, dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL(stem_onehot_kernel<2>, grid, dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL(stem_onehot_kernel<3>, grid, dim3(256), lds, s, a); break;
        default: hipLaunchKernelGGL(stem_onehot_kernel<4>, grid, dim3(256), lds, s, a); break;
    }
    PF_LAUNCH_CHECK("stem_onehot_kernel");
    return PF_OK;
}

int launch_avgpool2(const float *src, float *dst, int planes, int Hin, int Win, hipStream_t s) {
    const int Ho = Hin / 2, Wo = Win / 2;
    ProfScope ps(s, "pf::avgpool2_kernel", 0, 4.0 * planes * ((double)Hin * Win + (double)Ho * Wo));
    hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((size_t)planes * Ho * Wo)), dim3(256), 0, s, src, dst, planes,
                       Hin, Win, Ho, Wo);
    PF_LAUNCH_CHECK("avgpool2_kernel");
    return PF_OK;
}

int launch_upsample(const float *src, float *dst, int planes, int Hin, int Win, int Hout, int Wout, hipStream_t s) {
    ProfScope ps(s, "pf::upsample_kernel", 0, 4.0 * planes * ((double)Hin * Win + (double)Hout * Wout));
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for((size_t)planes * Hout * Wout)), dim3(256), 0, s, src, dst,
                       planes, Hin, Win, Hout, Wout);
    PF_LAUNCH_CHECK("upsample_kernel");
    return PF_OK;
}

int launch_head(const HeadArgs &a, hipStream_t s) {
    const double opx = (double)a.B * a.Hout * a.Wout;
    ProfScope ps(s, "pf::head_kernel(pf::HeadArgs)", 0, 4.0 * a.B * a.C * a.Hin * a.Win + opx * (a.out_is_i64 ? 8 : 1) +
                                                          (a.out_logits ? opx * a.C * 4 : 0));
    hipLaunchKernelGGL(head_kernel, dim3(grid_for((size_t)a.B * a.Hout * a.Wout)), dim3(256), 0, s, a);
    PF_LAUNCH_CHECK("head_kernel");
    return PF_OK;
}

}  // namespace pf
