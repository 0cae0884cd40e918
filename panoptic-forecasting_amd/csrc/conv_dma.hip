// Fast-path fp32 MFMA convolution (input width % 4 == 0): double-buffered LDS-DMA staging + in-workgroup K split.
//
// Same GEMM view as conv_mfma.hip (M = 16-pixel row segments, N = 16-cout tiles, K = 4 input channels per
// v_mfma_f32_16x16x4_f32), re-organised around what the profiles showed (profiles/r01_*): the first kernel
// spent its time in register-staged 4-byte gathers and starved the chip on the low-resolution layers.
//   * staging is asynchronous global->LDS DMA in 16-byte pieces (global_load_lds_dwordx4): the halo tile is
//     stored with a 4-float left apron so every row piece is 16-B aligned in HBM and in LDS; out-of-image
//     pieces and padding read a zero page.  No VGPR round trip, no ds_write.
//   * two LDS buffers: the DMA of round r+1 is in flight while round r is multiplied; one barrier per round.
//   * a workgroup is WM x WK waves: WM waves tile 64 pixels each, WK waves split the input-channel chunks (K)
//     among themselves and are summed through LDS in the epilogue, so a 16x32 or 32x64 layer still fills the
//     machine ((WM,WK) = (4,1) | (2,2) | (1,4), picked per layer at launch by a small cost model).
//   * weights are packed per 16-cout tile so the number of cout tiles per workgroup (NT) is a launch-time
//     choice too (fat workgroups at high resolution, many at low resolution).
#include "conv_mfma.h"
#include "pf_prof.h"

#ifndef PF_ABLATE
#define PF_ABLATE 0
#endif

namespace pf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS, int STRIDE, int WM, int WK, int NT>
struct DmaCfg {
    static constexpr int KC = dma_kc_ct(KS, STRIDE);  // input channels per stage
    static constexpr int MP = 4;                      // M-tiles (16 px) per wave
    static constexpr int TWT = WM == 1 ? 1 : 2;       // tile width in M-tiles
    static constexpr int TW = 16 * TWT;
    static constexpr int TH = WM * MP / TWT;          // WM=4: 8x32, WM=2: 4x32, WM=1: 4x16 output pixels
    static constexpr int APRON = KS == 3 ? 4 : 0;     // left apron (floats) keeping rows 16-B aligned
    static constexpr int IW = TW * STRIDE + 2 * APRON;
    static constexpr int IH = (TH - 1) * STRIDE + KS;
    static constexpr int RAW = IH * IW;
    static constexpr int PLANE = (RAW + 15) / 32 * 32 + 16;  // == 16 (mod 32): conflict-free stride-1 A reads
    static constexpr int PP = PLANE / 4, RP = RAW / 4, RW = IW / 4;  // 16-B pieces per plane / real / per row
    static constexpr int KS2 = KS * KS;
    static constexpr int WFRAG = (KC / 4) * KS2 * 64;  // floats per (cout tile, chunk)
    static constexpr int SLOT = KC * PLANE + NT * WFRAG;  // floats per K-split slot
    static constexpr int BUF = WK * SLOT;
    static constexpr int RED = (WK - 1) * WM * MP * NT * 256;
    static constexpr int LDS_FLOATS = 2 * BUF > RED ? 2 * BUF : RED;
    static_assert(PLANE % 4 == 0 && PLANE >= RAW && PLANE % 32 == 16, "plane stride");
};

__device__ __forceinline__ void dma16(const float *g, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int KS, int STRIDE, int WM, int WK, int NT>
__global__ __launch_bounds__(64 * WM * WK) void conv_dma_kernel(ConvArgs a) {
    using C = DmaCfg<KS, STRIDE, WM, WK, NT>;
    constexpr int NTHR = 64 * WM * WK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wk = wave / WM;
    const int tileY = blockIdx.x / a.tilesX, tileX = blockIdx.x - tileY * a.tilesX;
    const int tile0 = blockIdx.y * NT, b = blockIdx.z;   // first cout tile of this workgroup
    const int iy0 = tileY * C::TH * STRIDE - KS / 2, ix0 = tileX * C::TW * STRIDE - C::APRON;

    f32x4 acc[C::MP][NT];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int abase[C::MP];
#pragma unroll
    for (int m = 0; m < C::MP; ++m) {
        const int mt = wm * C::MP + m;
        const int ty = mt / C::TWT, tx0 = (mt % C::TWT) * 16;
        abase[m] = (lane >> 4) * C::PLANE + ty * STRIDE * C::IW + (tx0 + (lane & 15)) * STRIDE + (C::APRON - KS / 2);
    }

    const size_t in_plane = (size_t)a.Hin * a.Win;
    const float *zero = a.zero_page;
    const int nrounds = (a.nchunks + WK - 1) / WK;

    // issue the DMA of one round (WK chunks: inputs + weights) into buffer `buf`; piece p lands at p*16 B
    auto stage = [&](int round, float *buf) {
#pragma unroll
        for (int s = 0; s < WK; ++s) {
            const int chunk = round * WK + s;
            if (chunk >= a.nchunks) break;
            float *slot = buf + s * C::SLOT;
            constexpr int NPI = C::KC * C::PP;
#pragma unroll
            for (int it = 0; it < (NPI + NTHR - 1) / NTHR; ++it) {
                const int p = it * NTHR + tid;
                if (p < NPI) {
                    const int cl = p / C::PP, q = p - cl * C::PP;
                    const int row = q / C::RW, j = q - row * C::RW;
                    const int c = chunk * C::KC + cl, gy = iy0 + row, gx = ix0 + j * 4;
                    const float *g = zero;
                    if (q < C::RP && c < a.Cin && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) {
                        // source-range lookup as a select chain over kernel-argument scalars: a dynamically
                        // indexed a.src[sidx] becomes a vector load whose vmcnt wait serialises the DMA queue
                        const float *sp = a.src[0];
                        int ctot = a.src_ctotal[0], coff = a.src_choff[0];
#pragma unroll
                        for (int k = 1; k < kConvMaxSrc; ++k) {
                            const bool take = k < a.n_src && c >= a.src_cstart[k];
                            sp = take ? a.src[k] : sp;
                            ctot = take ? a.src_ctotal[k] : ctot;
                            coff = take ? a.src_choff[k] - a.src_cstart[k] : coff;
                        }
                        const size_t ch = (size_t)b * ctot + coff + c;
                        g = sp + ch * in_plane + (size_t)gy * a.Win + gx;
                    }
                    dma16(g, slot + (it * NTHR + wave * 64) * 4);
                }
            }
            constexpr int NPW = NT * C::WFRAG / 4;
            float *wslot = slot + C::KC * C::PLANE;
#pragma unroll
            for (int it = 0; it < (NPW + NTHR - 1) / NTHR; ++it) {
                const int p = it * NTHR + tid;
                if (p < NPW) {
                    const int n = p / (C::WFRAG / 4), q = p - n * (C::WFRAG / 4);
                    const float *g = zero;
                    if (tile0 + n < a.ntiles)
                        g = a.wpk + ((size_t)(tile0 + n) * a.nchunks + chunk) * C::WFRAG + q * 4;
                    dma16(g, wslot + (it * NTHR + wave * 64) * 4);
                }
            }
        }
    };

    stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int round = 0; round < nrounds; ++round) {
        float *cur = smem + (round & 1) * C::BUF;
#if PF_ABLATE == 1   /* no staging after the prologue: MFMA + LDS-read time only */
#else
        if (round + 1 < nrounds) stage(round + 1, smem + ((round + 1) & 1) * C::BUF);   // in flight during the MFMAs
#endif
#if PF_ABLATE == 2   /* staging only */
        if (round * WK + wk < a.nchunks && a.relu == 12345) {
#else
        if (round * WK + wk < a.nchunks) {
#endif
            const float *in_s = cur + wk * C::SLOT;
            const float *w_s = in_s + C::KC * C::PLANE;
#pragma unroll
            for (int kg = 0; kg < C::KC / 4; ++kg) {
#pragma unroll
                for (int tap = 0; tap < C::KS2; ++tap) {
                    const int ky = tap / KS, kx = tap - ky * KS;
                    float bf[NT], af[C::MP];
#pragma unroll
                    for (int n = 0; n < NT; ++n) bf[n] = w_s[n * C::WFRAG + (kg * C::KS2 + tap) * 64 + lane];
#pragma unroll
                    for (int m = 0; m < C::MP; ++m) af[m] = in_s[abase[m] + kg * 4 * C::PLANE + ky * C::IW + kx];
#pragma unroll
                    for (int m = 0; m < C::MP; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next round landed
        __syncthreads();                                    // ... and everyone is done with `cur`
    }

    // ---- K-split reduction through LDS: waves wk>0 publish, wave wk==0 sums
    if (WK > 1) {
        f32x4 *red = reinterpret_cast<f32x4 *>(smem);
        if (wk > 0) {
#pragma unroll
            for (int m = 0; m < C::MP; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    red[((((wk - 1) * WM + wm) * C::MP + m) * NT + n) * 64 + lane] = acc[m][n];
        }
        __syncthreads();
        if (wk > 0) return;
#pragma unroll
        for (int k = 1; k < WK; ++k)
#pragma unroll
            for (int m = 0; m < C::MP; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] += red[((((k - 1) * WM + wm) * C::MP + m) * NT + n) * 64 + lane];
    }

    // ---- epilogue (same fragment map as conv_mfma.hip): bias + ReLU, NCHW float4 stores
    const size_t out_plane = (size_t)a.Hout * a.Wout;
    const bool vec = (a.Wout & 3) == 0;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = (tile0 + n) * 16 + (lane & 15);
        if (co >= a.Cout) continue;
        const float bias = a.bias[co];
        float *dplane = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co) * out_plane;
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int mt = wm * C::MP + m;
            const int oy = tileY * C::TH + mt / C::TWT;
            const int ox = tileX * C::TW + (mt % C::TWT) * 16 + (lane >> 4) * 4;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            f32x4 v = acc[m][n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += bias;
                if (a.relu) v[r] = fmaxf(v[r], 0.f);
            }
            float *p = dplane + (size_t)oy * a.Wout + ox;
            if (vec && ox + 3 < a.Wout) {
                *reinterpret_cast<f32x4 *>(p) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ox + r < a.Wout) p[r] = v[r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int WM, int WK, int NT>
static int launch_dma_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = DmaCfg<KS, STRIDE, WM, WK, NT>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    const size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_dma_kernel<KS, STRIDE, WM, WK, NT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_dma_kernel<%d, %d, %d, %d, %d>(pf::ConvArgs)", KS, STRIDE, WM, WK, NT);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * KS * KS,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * KS * KS));
    hipLaunchKernelGGL((conv_dma_kernel<KS, STRIDE, WM, WK, NT>),
                       dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(64 * WM * WK), lds, s, a);
    PF_LAUNCH_CHECK("conv_dma_kernel");
    return PF_OK;
}

// Cost model (shader cycles) for one (WM, WK, NT) shape: the larger of the chip-wide MFMA time and the serial
// time of one workgroup times the number of workgroup waves; calibrated against profiles/r01_b_*.
static double shape_cost(const ConvArgs &a, int ks, int stride, int B, int wm, int wk, int nt) {
    const int kc = dma_kc(ks, stride), ks2 = ks * ks;
    const int tw = wm == 1 ? 16 : 32, th = wm * 4 / (tw / 16);
    const double px_tiles = (double)B * ((a.Hout + th - 1) / th) * ((a.Wout + tw - 1) / tw);
    const int cblocks = (a.ntiles + nt - 1) / nt;
    const double wgs = px_tiles * cblocks;
    const int nchunks = (a.Cin + kc - 1) / kc, rounds = (nchunks + wk - 1) / wk;
    const double mfma_round = (kc / 4) * ks2 * 4.0 * nt * 32.0;            // cycles of MFMA issue per wave per round
    const int iw = tw * stride + (ks == 3 ? 8 : 0), ih = (th - 1) * stride + ks;
    const double lds_bytes = 2.0 * wk * (kc * (ih * iw + 32) + nt * (kc / 4) * ks2 * 64) * 4.0;
    int occ = (int)(160.0 * 1024 / lds_bytes);
    occ = occ < 1 ? 1 : (occ > 4 ? 4 : occ);
    const double stage_round = 1800.0;                                      // exposed DMA latency per round, cycles
    const double wg_serial = 3000.0 + rounds * (mfma_round + stage_round / (occ > 1 ? 2 : 1)) + (wk > 1 ? 1500.0 : 0.0);
    const double waves = (double)(long)((wgs + 256.0 * occ - 1) / (256.0 * occ));
    const double serial = waves * wg_serial;
    // all MFMA work (padded cout tiles included; each chunk is multiplied by the WM pixel waves once)
    const double chip = wgs * wm * nchunks * mfma_round / (1024.0 * 0.85);
    // LDS-DMA volume: measured ~5.4 TB/s chip-wide from L2/MALL (ablation run, 2.1 GHz => ~2500 B/cycle)
    const double dma = wgs * nchunks * (kc * (ih * iw + 32.0) + nt * (kc / 4) * ks2 * 64.0) * 4.0 / 2500.0;
    double c = serial > chip ? serial : chip;
    return c > dma ? c : dma;
}

static void pick_shape(const ConvArgs &a, int ks, int stride, int B, int &wm, int &wk, int &nt) {
    double best = 1e300;
    const int wms[3] = {4, 2, 1};
    for (int i = 0; i < 3; ++i) {
        const int wki = 4 / wms[i];
        const int ntmax = wki == 4 ? 2 : 4;
        for (int n = ntmax < a.ntiles ? ntmax : a.ntiles; n >= 1; --n) {   // ties go to the fatter workgroup
            const double c = shape_cost(a, ks, stride, B, wms[i], wki, n);
            if (c < best * 0.999) {
                best = c;
                wm = wms[i];
                wk = wki;
                nt = n;
            }
        }
    }
}

int launch_conv_dma(const ConvArgs &a, int ks, int stride, int B, hipStream_t s, int force_wm, int force_nt) {
    int wm = 4, wk = 1, nt = 1;
    pick_shape(a, ks, stride, B, wm, wk, nt);
    if (force_wm > 0) {
        wm = force_wm;
        wk = 4 / wm;
        nt = force_nt < a.ntiles ? force_nt : a.ntiles;
    }
#define PF_CASE(KS_, ST_, WM_, WK_, NT_) \
    if (ks == KS_ && stride == ST_ && wm == WM_ && nt == NT_) return launch_dma_cfg<KS_, ST_, WM_, WK_, NT_>(a, B, s);
    PF_CASE(3, 1, 4, 1, 1) PF_CASE(3, 1, 4, 1, 2) PF_CASE(3, 1, 4, 1, 3) PF_CASE(3, 1, 4, 1, 4)
    PF_CASE(3, 1, 2, 2, 1) PF_CASE(3, 1, 2, 2, 2) PF_CASE(3, 1, 2, 2, 3) PF_CASE(3, 1, 2, 2, 4)
    PF_CASE(3, 1, 1, 4, 1) PF_CASE(3, 1, 1, 4, 2)
    PF_CASE(3, 2, 4, 1, 1) PF_CASE(3, 2, 4, 1, 2) PF_CASE(3, 2, 4, 1, 3) PF_CASE(3, 2, 4, 1, 4)
    PF_CASE(3, 2, 2, 2, 1) PF_CASE(3, 2, 2, 2, 2) PF_CASE(3, 2, 2, 2, 3) PF_CASE(3, 2, 2, 2, 4)
    PF_CASE(3, 2, 1, 4, 1) PF_CASE(3, 2, 1, 4, 2)
    PF_CASE(1, 1, 4, 1, 1) PF_CASE(1, 1, 4, 1, 2) PF_CASE(1, 1, 4, 1, 3) PF_CASE(1, 1, 4, 1, 4)
    PF_CASE(1, 1, 2, 2, 1) PF_CASE(1, 1, 2, 2, 2) PF_CASE(1, 1, 2, 2, 3) PF_CASE(1, 1, 2, 2, 4)
    PF_CASE(1, 1, 1, 4, 1) PF_CASE(1, 1, 1, 4, 2)
#undef PF_CASE
    return fail(PF_EUNSUPPORTED, "conv_dma: no kernel for ks=%d stride=%d wm=%d nt=%d", ks, stride, wm, nt);
}

// per-tile packing: [cout tile][chunk][kgroup][tap][64 lanes]
void pack_conv_weights_tiled(const float *w, int cin, int cout, int ks, int kc, float *out) {
    const int ks2 = ks * ks, ntiles = (cout + 15) / 16, nchunks = (cin + kc - 1) / kc;
    size_t o = 0;
    for (int t = 0; t < ntiles; ++t)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int kg = 0; kg < kc / 4; ++kg)
                for (int tap = 0; tap < ks2; ++tap)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int co = t * 16 + (lane & 15), ci = ch * kc + kg * 4 + (lane >> 4);
                        out[o++] = (co < cout && ci < cin) ? w[((size_t)co * cin + ci) * ks2 + tap] : 0.f;
                    }
}

}  // namespace pf
