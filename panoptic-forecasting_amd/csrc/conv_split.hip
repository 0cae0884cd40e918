// 3x3/s1 convolution with fp32 operands SPLIT into two fp16 terms on the 16-bit matrix pipe of gfx950 (the default for the layers
// the tuned table gives it; plan option split_f16 = 0 keeps everything on the fp32 matrix instructions).
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 vector rate and shares its budget with the VALU: ~135 TF/s is
// the ceiling of conv_dma.hip.  v_mfma_f32_16x16x32_f16 is 16x faster.  Every fp32 value x is split into
//     x = hi + mid + lo,   hi = fp16(x),  mid = fp16(x - hi)            (x - hi is exact in fp32; conv_mfma.h: split_terms2)
// and a product a*b is evaluated as  a_hi*b_hi + a_hi*b_mid + a_mid*b_hi  in fp32 accumulators: three matrix
// instructions instead of one at 3/16 of the time.  The dropped terms (a_mid*b_mid, a*b_lo, a_lo*b) are <= 2^-21 of
// the product each - below the rounding of the fp32 accumulation itself: the tests hold these kernels to the tolerance of
// the fp32 kernels (tests/test_gpu_conv.py: 2e-5*(1+max|ref|)) and whole-network logits move by 4e-5 either way.  (Round 1
// and most of round 2 used bf16 terms, 8 + 8 bits: 2^-16 per product, 5e-4 on the logits, same speed.)  fp16's range is
// handled by saturating conversions for activations and an exact power-of-two scaling of the weights (conv_mfma.h).
// Inputs and outputs stay fp32 NCHW: the split happens inside the kernel, so the path is a drop-in for single layers.
//
// GEMM view: M = 16 pixels of a row, N = 16 output channels, K = 32 per instruction = 8 input channels x 4 TAPS: lane l
// supplies 8 consecutive-K values = the 8 channels of one pixel for tap 4*step + (l >> 4).  9 taps = 3 steps (the last
// with 3 zero-weight slots).  Per round of 8 input channels:
//   * every thread loads "its" group of 4 consecutive halo pixels (8 channels, one 16-B load each) into registers one
//     round ahead, splits them and writes 16-B vectors [pixel][8 ch] (hi, mid) to LDS: an A fragment is then ONE
//     ds_read_b128 per term; workgroup tiles are 8x32 or 8x64 pixels (halo over-read 1.56x / 1.41x);
//   * weights are pre-split and pre-tiled on the host in fragment order ([tile][chunk][step][term][lane][8]) and arrive by
//     LDS-DMA; a B fragment is one conflict-free ds_read_b128;
//   * 4 waves x 4 M-tiles x NT cout tiles x 3 steps x 3 products MFMAs per round.
#include <cstring>

#include "conv_epilogue.h"
#include "pf_prof.h"

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // shader-clock stamps of one workgroup in the middle of the grid (tools/probe_split.py)
#define SPLIT_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && a.probe && (i) < 60) a.probe[i] = clock64(); } while (0)
#else
#define SPLIT_PROBE(i) do { } while (0)
#endif

namespace pf {

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef split_x8 sp_h8;   // 8 fp16 terms (conv_mfma.h: split_terms2)
typedef __attribute__((address_space(3))) void *sp_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kSplitOob = 0x80000000u;

template <int NT, int TW_>
struct SplitCfg {
    static constexpr int KC = 8, TW = TW_, TH = 8, MTR = TW / 16, MP = 2 * MTR;   // 4 waves x MP M-tiles = 8 rows x TW pixels
    static constexpr int IW = TW + 8, IH = TH + 2, NPIX = IH * IW;       // halo tile with a 4-pixel apron left/right
    static constexpr int GW = IW / 4, NGRP = IH * GW;                    // 4-pixel groups: one per thread (<= 256)
    static constexpr int ABUF = 2 * NPIX * 16;                           // bytes: [term][pixel][8 fp16]
    static constexpr int WBUF = NT * 3 * 2 * 64 * 16;                    // bytes: [nt][step][term][lane][8 fp16]
    // ONE activation buffer (an extra barrier per round before it is overwritten) and two weight buffers: 47-49 KB for the
    // common shapes = 3 workgroups per CU; double-buffering the activations as well (70 KB, 2 per CU) measured ~10 % slower
    static constexpr size_t LDS_BYTES = (size_t)ABUF + 2 * (size_t)WBUF;
    static constexpr int WPIECES = WBUF / 16;
    static_assert(NGRP <= 256, "one halo pixel group per thread");
};

template <int NT, int TW_>
__global__ __launch_bounds__(256) void conv_split_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = SplitCfg<NT, TW_>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware order of tiles and cout groups (xcd_tile_order, conv_mfma.h): workgroup ids are dealt round-robin to the 8
    // XCDs (id % 8), each with its own L2; every XCD gets a contiguous band of tiles, the cout groups of a tile back to back
    int tid_lin, cgroup;
    xcd_tile_order(a.tilesX * a.tilesY, tid_lin, cgroup);
    const int tileY = tid_lin / a.tilesX, tileX = tid_lin - tileY * a.tilesX;
    const int tile0 = cgroup * NT, b = blockIdx.z;
    const int iy0 = tileY * C::TH - 1, ix0 = tileX * C::TW - 4;
    auto abuf = [&](int) { return smem_raw; };
    auto wbuf = [&](int i) { return smem_raw + C::ABUF + i * C::WBUF; };

    sp_f32x4 acc[C::MP][NT];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = sp_f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's halo pixel GROUP (4 consecutive pixels of a row: one 16-B load per channel; W % 4 == 0 and the 4-pixel
    // apron keep a group entirely inside or entirely outside the image): offset inside a channel plane, or -1
    const unsigned in_plane = (unsigned)a.Hin * a.Win;
    int poff;
    {
        const int row = tid / C::GW, col = (tid - row * C::GW) * 4;
        const int gy = iy0 + row, gx = ix0 + col;
        poff = (tid < C::NGRP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? gy * a.Win + gx : -1;
    }
    // A-fragment byte offsets (inside one term's plane) for step s, M-tile m: tap = min(4s + (lane>>4), 8)
    int aoff[3][C::MP];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = min(4 * s + (lane >> 4), 8), ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int mt = wave * C::MP + m, ty = mt / C::MTR, tx0 = (mt % C::MTR) * 16;
            aoff[s][m] = ((ty + ky) * C::IW + tx0 + (lane & 15) + kx + 3) * 16;
        }
    }
    unsigned woff[(C::WPIECES + 255) / 256];
#pragma unroll
    for (int it = 0; it < (C::WPIECES + 255) / 256; ++it) {
        const int p = it * 256 + tid, n = p / (3 * 2 * 64);   // pieces of cout tile n are contiguous
        woff[it] = (p < C::WPIECES && tile0 + n < a.ntiles) ? (unsigned)((tile0 + n) * a.nchunks * (3 * 2 * 64) + (p - n * (3 * 2 * 64))) * 16u : kSplitOob;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wpk, 0, 0x7FFFFFFF, 0x00020000);

    float biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = epi_bias(a, (tile0 + n) * 16 + (lane & 15));

    // source tensor / first channel / valid channels of a chunk (workgroup-uniform select chains)
    auto chunk_src = [&](int chunk, const float *&base, int &nvalid) {
        const float *sp = a.src[0];
        int ctot = a.src_ctotal[0], coff = a.src_choff[0], ch0 = 0, cend = a.src_cstart[1], c0 = 0;
#pragma unroll
        for (int k = 1; k < kConvMaxSrc; ++k) {
            const bool take = k < a.n_src && chunk >= a.src_chunk0[k];
            sp = take ? a.src[k] : sp;
            ctot = take ? a.src_ctotal[k] : ctot;
            coff = take ? a.src_choff[k] : coff;
            ch0 = take ? a.src_chunk0[k] : ch0;
            c0 = take ? a.src_cstart[k] : c0;
            cend = take ? a.src_cstart[k + 1] : cend;
        }
        const int lc = chunk - ch0;
        nvalid = (cend - c0) - lc * C::KC;
        base = sp + ((size_t)b * ctot + coff + lc * C::KC) * in_plane;
    };

    sp_f32x4 xr[C::KC];   // the round in flight: 8 channels x this thread's 4 halo pixels
    auto load_round = [&](int chunk, unsigned char *wdst) {
        const float *base;
        int nv;
        chunk_src(chunk, base, nv);
#pragma unroll
        for (int c = 0; c < C::KC; ++c)
            xr[c] = (poff >= 0 && c < nv) ? *reinterpret_cast<const sp_f32x4 *>(base + (size_t)c * in_plane + poff)
                                          : sp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < (C::WPIECES + 255) / 256; ++it)
            if (it * 256 + tid < C::WPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (sp_lds_ptr_t)(wdst + (it * 256 + wave * 64) * 16), 16, woff[it],
                                                         (unsigned)chunk * (3 * 2 * 64 * 16), 0, 0);
    };
    auto split_store = [&](unsigned char *adst) {
        if (tid >= C::NGRP) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sp_h8 hi, mid;
#pragma unroll
            for (int c = 0; c < C::KC; c += 2) {
                split_x2 h, m;
                split_terms2(xr[c][k], xr[c + 1][k], h, m);
                hi[c] = h[0]; hi[c + 1] = h[1];
                mid[c] = m[0]; mid[c + 1] = m[1];
            }
            *reinterpret_cast<sp_h8 *>(adst + (tid * 4 + k) * 16) = hi;
            *reinterpret_cast<sp_h8 *>(adst + C::NPIX * 16 + (tid * 4 + k) * 16) = mid;
        }
    };

    const int cb = a.chunk_begin, nrounds = a.chunk_end - cb;
    if (nrounds > 0) {
        load_round(cb, wbuf(0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        split_store(abuf(0));
    }
    __syncthreads();

    for (int round = 0; round < nrounds; ++round) {
        const unsigned char *ab = abuf(round & 1), *wb = wbuf(round & 1);
        SPLIT_PROBE(round * 6 + 0);
        if (round + 1 < nrounds) load_round(cb + round + 1, wbuf((round + 1) & 1));   // in flight during the MFMAs
        SPLIT_PROBE(round * 6 + 1);
        // Units of work: (tap step s, group of 4 M-tiles), written as fetch(u+1) before the MFMAs of unit u.  hipcc re-sinks
        // most of the reads towards their uses (fewer live registers, one more wave per SIMD); pinning the prefetch with
        // sched_barriers was measured 5-20 % SLOWER (188 registers -> 2 waves per SIMD).
        // Within a unit the order is product-major: consecutive MFMAs hit different accumulators (a 16x16x32 MFMA has 8
        // passes; back-to-back updates of ONE accumulator would serialise on its result); smallest terms first.
        constexpr int HALVES = C::MP / 4, UNITS = 3 * HALVES;
        sp_h8 fa_h[2][4], fa_m[2][4], fb_h[2][NT], fb_m[2][NT];
        auto fetch = [&](int u, int set) {
            const int s = u / HALVES, m0 = (u % HALVES) * 4;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                fb_h[set][n] = *reinterpret_cast<const sp_h8 *>(wb + (((n * 3 + s) * 2 + 0) * 64 + lane) * 16);
                fb_m[set][n] = *reinterpret_cast<const sp_h8 *>(wb + (((n * 3 + s) * 2 + 1) * 64 + lane) * 16);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                fa_h[set][m] = *reinterpret_cast<const sp_h8 *>(ab + aoff[s][m0 + m]);
                fa_m[set][m] = *reinterpret_cast<const sp_h8 *>(ab + C::NPIX * 16 + aoff[s][m0 + m]);
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int set = u & 1, m0 = (u % HALVES) * 4;
            if (u + 1 < UNITS) fetch(u + 1, set ^ 1);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m0 + m][n] = PF_MFMA_SPLIT(fa_m[set][m], fb_h[set][n], acc[m0 + m][n]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m0 + m][n] = PF_MFMA_SPLIT(fa_h[set][m], fb_m[set][n], acc[m0 + m][n]);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m0 + m][n] = PF_MFMA_SPLIT(fa_h[set][m], fb_h[set][n], acc[m0 + m][n]);
        }
        SPLIT_PROBE(round * 6 + 2);   // all MFMAs of the round issued
        if (round + 1 < nrounds) {
            __syncthreads();   // everyone is done reading the activation buffer
            SPLIT_PROBE(round * 6 + 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SPLIT_PROBE(round * 6 + 4);
            split_store(abuf((round + 1) & 1));
        }
        SPLIT_PROBE(round * 6 + 5);
        __syncthreads();
    }

    // ---- epilogue: bias + ReLU, NCHW float4 stores (same D fragment as the fp32 kernels)
    float vmax = 0.f;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int co = (tile0 + n) * 16 + (lane & 15);
        if (epi_skip(a, co)) continue;
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int mt = wave * C::MP + m;
            const int oy = tileY * C::TH + mt / C::MTR;
            const int ox = tileX * C::TW + (mt % C::MTR) * 16 + (lane >> 4) * 4;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            sp_f32x4 v = acc[m][n];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = v[r] * a.acc_scale + biasv[n];   // acc_scale = 2^-k of the weight scaling: exact
                if (a.relu) v[r] = fmaxf(v[r], 0.f);
            }
            epi_store(a, b, co, oy, ox, v, vmax);
        }
    }
    range_commit(a.status, a.range_slot, vmax);
#endif
}

template <int NT, int TW_>
static int launch_split_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = SplitCfg<NT, TW_>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_split_kernel<NT, TW_>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_split_kernel<%d, %d>(pf::ConvArgs)", NT, TW_);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * 9,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * 9));
    hipLaunchKernelGGL((conv_split_kernel<NT, TW_>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(256), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_split_kernel");
    return PF_OK;
}

// a.wpk = pack_conv_weights_split() output; chunks of 8 channels (a.src_chunk0 / chunk_begin / chunk_end set for 8).
// nt = cout tiles per workgroup (1..3), wide = 8x64-pixel workgroup tiles instead of 8x32.
int launch_conv_split(const ConvArgs &a, int nt, int wide, int B, hipStream_t s) {
    if (a.pool || a.res || a.no_bias) return fail(PF_EUNSUPPORTED, "conv_split: no fused epilogue stages");
    if ((a.Wout & 3) != 0 || a.Hin != a.Hout || a.Win != a.Wout) return fail(PF_EUNSUPPORTED, "conv_split: 3x3/s1, width % 4 == 0 only");
    nt = nt < 1 ? 1 : (nt > a.ntiles ? a.ntiles : nt);
    if (wide) {
        if (nt == 1) return launch_split_cfg<1, 64>(a, B, s);
        if (nt == 2) return launch_split_cfg<2, 64>(a, B, s);
        return launch_split_cfg<3, 64>(a, B, s);
    }
    if (nt == 1) return launch_split_cfg<1, 32>(a, B, s);
    if (nt == 2) return launch_split_cfg<2, 32>(a, B, s);
    return launch_split_cfg<3, 32>(a, B, s);
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolutions on the same scheme: K = 32 per instruction = 4 groups of 8 input channels (no taps, nothing padded but
// the channel ranges, to 32); one round = 32 channels of an 8x32-pixel tile; thread (g, q) loads the 4 pixels of group g
// for the 8 channels of k-group q, splits them and writes [term][k-group][pixel][8 ch].  The fused epilogue stages of
// conv_epilogue.h (2x2 average pool, bilinearly upsampled residual, bias-free low-resolution half) are supported: the
// D fragment is the one the fp32 kernels produce.  These layers were fp32-MFMA-bound at ~50 TF/s (K = Cin only) while
// moving < 2.5 TB/s; here they are bound by their bytes.
template <int NT>
struct Split1Cfg {
    static constexpr int KC = 32, TW = 32, TH = 8, MTR = 2, MP = 4, NPIX = TH * TW;
    static constexpr int ABUF = 2 * 4 * NPIX * 16;                  // bytes: [term][k-group][pixel][8 fp16]
    static constexpr int WBUF = NT * 2 * 64 * 16;                   // bytes: [nt][term][lane][8 fp16]
    static constexpr int MAIN = ABUF + 2 * WBUF;                    // one activation buffer, two weight buffers (see SplitCfg)
    static constexpr int WPIECES = WBUF / 16;
};

template <int NT, int EPI>
__global__ __launch_bounds__(256) void conv_split1_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Split1Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tid_lin, cgroup;
    xcd_tile_order(a.tilesX * a.tilesY, tid_lin, cgroup);
    const int tileY = tid_lin / a.tilesX, tileX = tid_lin - tileY * a.tilesX;
    const int tile0 = cgroup * NT, b = blockIdx.z;
    auto abuf = [&](int) { return smem_raw; };
    auto wbuf = [&](int i) { return smem_raw + C::ABUF + i * C::WBUF; };

    sp_f32x4 acc[C::MP][NT];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = sp_f32x4{0.f, 0.f, 0.f, 0.f};

    // thread = (pixel group g of 4 consecutive pixels, k-group q of 8 channels)
    const int g = tid & 63, q = tid >> 6;
    const unsigned in_plane = (unsigned)a.Hin * a.Win;
    int poff;
    {
        const int row = g >> 3, col = (g & 7) * 4;
        const int gy = tileY * C::TH + row, gx = tileX * C::TW + col;
        poff = (gy < a.Hin && gx < a.Win) ? gy * a.Win + gx : -1;
    }
    int aoff[C::MP];
#pragma unroll
    for (int m = 0; m < C::MP; ++m) {
        const int mt = wave * C::MP + m, ty = mt / C::MTR, tx0 = (mt % C::MTR) * 16;
        aoff[m] = ((lane >> 4) * C::NPIX + ty * C::TW + tx0 + (lane & 15)) * 16;
    }
    constexpr int NITW = (C::WPIECES + 255) / 256;
    unsigned woff[NITW];
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
        const int p = it * 256 + tid, n = p / (2 * 64);
        woff[it] = (p < C::WPIECES && tile0 + n < a.ntiles) ? (unsigned)((tile0 + n) * a.nchunks * (2 * 64) + (p - n * (2 * 64))) * 16u : kSplitOob;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wpk, 0, 0x7FFFFFFF, 0x00020000);

    float biasv[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = epi_bias(a, (tile0 + n) * 16 + (lane & 15));

    auto chunk_src = [&](int chunk, const float *&base, int &nvalid) {
        const float *sp = a.src[0];
        int ctot = a.src_ctotal[0], coff = a.src_choff[0], ch0 = 0, cend = a.src_cstart[1], c0 = 0;
#pragma unroll
        for (int k = 1; k < kConvMaxSrc; ++k) {
            const bool take = k < a.n_src && chunk >= a.src_chunk0[k];
            sp = take ? a.src[k] : sp;
            ctot = take ? a.src_ctotal[k] : ctot;
            coff = take ? a.src_choff[k] : coff;
            ch0 = take ? a.src_chunk0[k] : ch0;
            c0 = take ? a.src_cstart[k] : c0;
            cend = take ? a.src_cstart[k + 1] : cend;
        }
        const int lc = chunk - ch0;
        nvalid = (cend - c0) - lc * C::KC;
        base = sp + ((size_t)b * ctot + coff + lc * C::KC) * in_plane;
    };

    sp_f32x4 xr[8];
    auto load_round = [&](int chunk, unsigned char *wdst) {
        const float *base;
        int nv;
        chunk_src(chunk, base, nv);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            xr[c] = (poff >= 0 && q * 8 + c < nv) ? *reinterpret_cast<const sp_f32x4 *>(base + (size_t)(q * 8 + c) * in_plane + poff)
                                                  : sp_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int it = 0; it < NITW; ++it)
            if (it * 256 + tid < C::WPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (sp_lds_ptr_t)(wdst + (it * 256 + wave * 64) * 16), 16, woff[it],
                                                         (unsigned)chunk * (2 * 64 * 16), 0, 0);
    };
    auto split_store = [&](unsigned char *adst) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sp_h8 hi, mid;
#pragma unroll
            for (int c = 0; c < 8; c += 2) {
                split_x2 h, m;
                split_terms2(xr[c][k], xr[c + 1][k], h, m);
                hi[c] = h[0]; hi[c + 1] = h[1];
                mid[c] = m[0]; mid[c + 1] = m[1];
            }
            *reinterpret_cast<sp_h8 *>(adst + (q * C::NPIX + g * 4 + k) * 16) = hi;
            *reinterpret_cast<sp_h8 *>(adst + (4 * C::NPIX + q * C::NPIX + g * 4 + k) * 16) = mid;
        }
    };

    const int cb = a.chunk_begin, nrounds = a.chunk_end - cb;
    if (nrounds > 0) {
        load_round(cb, wbuf(0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        split_store(abuf(0));
    }
    __syncthreads();

    for (int round = 0; round < nrounds; ++round) {
        const unsigned char *ab = abuf(round & 1), *wb = wbuf(round & 1);
        if (round + 1 < nrounds) load_round(cb + round + 1, wbuf((round + 1) & 1));
        sp_h8 bh[NT], bm[NT], ah[C::MP], am[C::MP];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[n] = *reinterpret_cast<const sp_h8 *>(wb + ((n * 2 + 0) * 64 + lane) * 16);
            bm[n] = *reinterpret_cast<const sp_h8 *>(wb + ((n * 2 + 1) * 64 + lane) * 16);
        }
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            ah[m] = *reinterpret_cast<const sp_h8 *>(ab + aoff[m]);
            am[m] = *reinterpret_cast<const sp_h8 *>(ab + 4 * C::NPIX * 16 + aoff[m]);
        }
#pragma unroll
        for (int m = 0; m < C::MP; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(am[m], bh[n], acc[m][n]);
#pragma unroll
        for (int m = 0; m < C::MP; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(ah[m], bm[n], acc[m][n]);
#pragma unroll
        for (int m = 0; m < C::MP; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(ah[m], bh[n], acc[m][n]);
        if (round + 1 < nrounds) {
            __syncthreads();   // everyone is done reading the activation buffer
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            split_store(abuf((round + 1) & 1));
        }
        __syncthreads();
    }

    float vmax = 0.f;
    if (EPI == 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = (tile0 + n) * 16 + (lane & 15);
            if (epi_skip(a, co)) continue;
#pragma unroll
            for (int m = 0; m < C::MP; ++m) {
                const int mt = wave * C::MP + m;
                const int oy = tileY * C::TH + mt / C::MTR;
                const int ox = tileX * C::TW + (mt % C::MTR) * 16 + (lane >> 4) * 4;
                if (oy >= a.Hout || ox >= a.Wout) continue;
                sp_f32x4 v = acc[m][n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = v[r] * a.acc_scale + biasv[n];   // acc_scale = 2^-k of the weight scaling: exact
                    if (a.relu) v[r] = fmaxf(v[r], 0.f);
                }
                epi_store(a, b, co, oy, ox, v, vmax);
            }
        }
    } else {
        // fused stages (conv_epilogue.h), as in conv_dma.hip: residual window of this tile -> LDS (the main-loop buffers
        // are free after the last barrier), then pixel-group outer / channel inner
        ResWin rw = ResWin();
        const lds_float *res_lds = nullptr;
        const bool has_res = a.res && a.res_lds_off >= 0;
        if (has_res) {
            rw = res_window(a, tileY * C::TH, C::TH, tileX * C::TW, C::TW);
            res_stage(a, rw, b, tile0 * 16, NT * 16, (lds_float *)(reinterpret_cast<float *>(smem_raw) + a.res_lds_off), tid, 256);
            res_lds = (const lds_float *)(reinterpret_cast<float *>(smem_raw) + a.res_lds_off);
            __syncthreads();
        }
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int mt = wave * C::MP + m;
            const int oy = tileY * C::TH + mt / C::MTR;
            const int ox = tileX * C::TW + (mt % C::MTR) * 16 + (lane_e >> 4) * 4;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            const bool pool_top = a.pool && ((m / C::MTR) & 1) == 0 && m + C::MTR < C::MP && oy + 1 < a.Hout;
            if (a.pool && !pool_top) continue;
            ResTaps t0, t1;
            if (has_res) {
                t0 = res_taps(a, rw, oy, ox);
                if (pool_top) t1 = res_taps(a, rw, oy + 1, ox);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = (tile0 + n) * 16 + (lane_e & 15);
                if (epi_skip(a, co)) continue;
                const lds_float *chan = res_lds + (n * 16 + (lane_e & 15)) * rw.cs;
                const sp_f32x4 top = epi_finish(a, b, co, oy, ox, acc[m][n] * a.acc_scale, biasv[n], has_res, chan, &t0);
                if (!a.pool) epi_store(a, b, co, oy, ox, top, vmax);
                else epi_store_pooled(a, b, co, oy, ox, top, epi_finish(a, b, co, oy + 1, ox, acc[(m + C::MTR) % C::MP][n] * a.acc_scale, biasv[n], has_res, chan, &t1), vmax);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    range_commit(a.status, a.range_slot, vmax);
#endif
}

template <int NT, int EPI>
static int launch_split1_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = Split1Cfg<NT>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    size_t lds = C::MAIN;
    a.res_lds_off = -1;
    if (a.res) {
        const size_t need = (size_t)NT * 16 * res_chan_stride(res_extent(C::TH, a.res_sh), res_extent(C::TW, a.res_sw)) * sizeof(float);
        if (need > 64 * 1024) return fail(PF_EUNSUPPORTED, "conv_split1: residual window of %zu B does not fit LDS", need);
        a.res_lds_off = 0;
        if (need > lds) lds = need;
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_split1_kernel<NT, EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    char label[112];
    snprintf(label, sizeof(label), "void pf::conv_split1_kernel<%d, %d>(pf::ConvArgs)", NT, EPI);
    if (a.res) strncat(label, " +res", sizeof(label) - strlen(label) - 1);
    if (a.pool) strncat(label, " +pool", sizeof(label) - strlen(label) - 1);
    if (a.no_bias) strncat(label, " lowres-half", sizeof(label) - strlen(label) - 1);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin, 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin));
    hipLaunchKernelGGL((conv_split1_kernel<NT, EPI>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(256), lds, s, a);
    PF_LAUNCH_CHECK("conv_split1_kernel");
    return PF_OK;
}

// 1x1/s1: a.wpk = pack_conv_weights_split1() output; chunks of 32 channels.  nt = cout tiles per workgroup (1..4).
int launch_conv_split1(const ConvArgs &a, int nt, int B, hipStream_t s) {
    if ((a.Wout & 3) != 0 || a.Hin != a.Hout || a.Win != a.Wout) return fail(PF_EUNSUPPORTED, "conv_split1: 1x1/s1, width % 4 == 0 only");
    nt = nt < 1 ? 1 : (nt > a.ntiles ? a.ntiles : nt);
    nt = nt > 4 ? 4 : nt;
    const bool fused = a.pool || a.res || a.no_bias;
#define PF_S1(NT_) \
    if (nt == NT_) return fused ? launch_split1_cfg<NT_, 1>(a, B, s) : launch_split1_cfg<NT_, 0>(a, B, s);
    PF_S1(1) PF_S1(2) PF_S1(3) PF_S1(4)
#undef PF_S1
    return PF_EUNSUPPORTED;
}

int split1_chunks(const int *src_ch, int n_src) {
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += (src_ch[j] + 31) / 32;
    return n;
}

// bytes as floats: [tile][chunk of 32 ch][term 2][lane 64][8 fp16]; lane = (cout n = lane & 15, k-group lane >> 4)
size_t split1_packed_floats(const int *src_ch, int n_src, int cout) {
    return (size_t)((cout + 15) / 16) * split1_chunks(src_ch, n_src) * 2 * 64 * 4;
}

void pack_conv_weights_split1(const float *w, int cin, int cout, const int *src_ch, int n_src, float *out_f) {
    unsigned short *out = reinterpret_cast<unsigned short *>(out_f);
    const int ntiles = (cout + 15) / 16;
    size_t o = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c0 = 0;
        for (int j = 0; j < n_src; ++j) {
            for (int lc = 0; lc * 32 < src_ch[j]; ++lc)
                for (int term = 0; term < 2; ++term)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = t * 16 + (lane & 15), cl = lc * 32 + (lane >> 4) * 8 + e;
                            const float v = (co < cout && cl < src_ch[j]) ? w[(size_t)co * cin + c0 + cl] : 0.f;
                            const unsigned short hi = split_host_f16(v);
                            out[o++] = term == 0 ? hi : split_host_f16(v - split_host_f32(hi));
                        }
            c0 += src_ch[j];
        }
    }
}

// ---- host halves of the operand split (conv_mfma.h)
unsigned short split_host_f16(float x) {
    unsigned u;
    memcpy(&u, &x, 4);
    const unsigned short sign = (unsigned short)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (unsigned short)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));   // nan / inf
    if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);                                  // >= 65520 rounds to inf
    if (u < 0x38800000u) {   // below 2^-14: a multiple of the subnormal step 2^-24 (1024 steps = the smallest normal: same bits)
        float f;
        memcpy(&f, &u, 4);
        return (unsigned short)(sign | (unsigned)nearbyintf(f * 16777216.0f));   // default rounding mode: nearest even
    }
    u -= 0x38000000u;                        // exponent bias 127 -> 15
    u += 0xfffu + ((u >> 13) & 1u);          // nearest even on the 13 dropped bits (a carry moves into the exponent)
    return (unsigned short)(sign | (u >> 13));
}
float split_host_f32(unsigned short h) {
    const unsigned sign = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    float f;
    if (e == 0) {
        f = (float)m * (1.0f / 16777216.0f);
        unsigned u;
        memcpy(&u, &f, 4);
        u |= sign;
        memcpy(&f, &u, 4);
        return f;
    }
    const unsigned u = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112u) << 23) | (m << 13));
    memcpy(&f, &u, 4);
    return f;
}
float split_weight_scale(const float *w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.0f;
    int e;
    frexpf(mx, &e);                          // mx = f * 2^e, f in [0.5, 1)
    return ldexpf(1.0f, 15 - e);             // mx * 2^(15 - e) in [2^14, 2^15)
}

int split_chunks(const int *src_ch, int n_src) {
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += (src_ch[j] + 7) / 8;
    return n;
}

// bytes as floats (the weight arena is a float array): [tile][chunk][step 3][term 2][lane 64][8 fp16] = 16 B per lane
size_t split_packed_floats(const int *src_ch, int n_src, int cout) {
    return (size_t)((cout + 15) / 16) * split_chunks(src_ch, n_src) * 3 * 2 * 64 * 4;
}

void pack_conv_weights_split(const float *w, int cin, int cout, const int *src_ch, int n_src, float *out_f) {
    unsigned short *out = reinterpret_cast<unsigned short *>(out_f);
    const int ntiles = (cout + 15) / 16;
    size_t o = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c0 = 0;
        for (int j = 0; j < n_src; ++j) {
            for (int lc = 0; lc * 8 < src_ch[j]; ++lc)
                for (int s = 0; s < 3; ++s)
                    for (int term = 0; term < 2; ++term)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int co = t * 16 + (lane & 15), tap = 4 * s + (lane >> 4), cl = lc * 8 + e;
                                float v = 0.f;
                                if (co < cout && tap < 9 && cl < src_ch[j]) v = w[((size_t)co * cin + c0 + cl) * 9 + tap];
                                const unsigned short hi = split_host_f16(v);
                                out[o++] = term == 0 ? hi : split_host_f16(v - split_host_f32(hi));
                            }
            c0 += src_ch[j];
        }
    }
}

}  // namespace pf
