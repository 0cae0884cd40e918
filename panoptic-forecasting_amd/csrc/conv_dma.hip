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
#include <cstring>

#include "conv_epilogue.h"
#include "pf_prof.h"

#ifndef PF_ABLATE
#define PF_ABLATE 0
#endif

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // 100 MHz wall-clock stamps of workgroup (5, 0, 0), thread 0 (tools/probe_conv.py)
#define DPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 5 && blockIdx.y == 0 && blockIdx.z == 0 && a.probe) a.probe[i] = wall_clock64(); } while (0)
#else
#define DPROBE(i) do { } while (0)
#endif

namespace pf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS, int STRIDE, int WM, int WK, int NT, int RV = 0>
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
    static constexpr int RFRAG = (KC / 4) * KS2 * RV * 4;  // vector-ALU cout weights per chunk: [kgroup][tap][RV/4][4 ch][4 couts]
    static constexpr int SLOT = KC * PLANE + NT * WFRAG + RFRAG;  // floats per K-split slot
    static constexpr int BUF = WK * SLOT;
    static constexpr int RED = (WK - 1) * WM * MP * NT * 256;
    static constexpr int LDS_FLOATS = 2 * BUF > RED ? 2 * BUF : RED;
    static_assert(PLANE % 4 == 0 && PLANE >= RAW && PLANE % 32 == 16, "plane stride");
};

typedef __attribute__((address_space(3))) void *dma_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kDmaOob = 0x80000000u;   // voffset >= num_records: the DMA writes zeros (conv zero padding)

// EPI: 0 = bias + ReLU (the common case keeps its register budget), 1 = fused stages of conv_epilogue.h (2x2 pool,
// upsampled residual) - separate instantiations because the fused epilogue needs ~70 more VGPRs.
// RV > 0 (a multiple of 4): the last `a.rem` <= RV output channels do not go through the matrix pipe at all.  On gfx950 the
// fp32 vector ALU has the same peak as the fp32 MFMA (157 TF/s) and the two pipes run concurrently, so these channels
// are accumulated on the vector ALU in the issue shadow of the MFMAs of the full tiles: a 28-channel HarDBlock layer
// runs as one 16-wide MFMA tile + 12 VALU channels instead of two MFMA tiles (12.5 % of them zeros).  Per k-group/tap:
//   * the A operands the MFMAs just used (lane = pixel l&15 of M-tile m, channel l>>4) are turned into the VALU layout
//     (lane = pixel l&15 of M-tile l>>4, registers = the 4 channels) by a 4x4 block transpose in registers:
//     2 x v_permlane32_swap + 2 x v_permlane16_swap (gfx950) - no second LDS read of the input tile;
//   * the weights of 4 output channels x 4 input channels arrive with ONE ds_read_b128 per wave: lane l reads the 16-B
//     piece of input channel l&3 (4 distinct addresses: a broadcast read), and each FMA picks "its" channel's weight
//     out of the quad with a DPP quad_perm on the multiplier (v_fmac_f32_dpp): no broadcast instruction, no SGPRs.
// (hipcc's __builtin_amdgcn_permlane*_swap returns the same register for both results on ROCm 7.2, hence inline asm;
// the s_nop's are the VALU-write -> permlane-swap-read wait states the compiler would insert.)
#define PF_FMAC_QUAD(acc, w, x, C) \
    asm("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #C "," #C "," #C "," #C "] row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(x))
// KACC (training forward, ConvArgs::kacc): the sums of every round (8 input channels x the taps: 72 terms for a 3x3 conv) are
// added into a second set of accumulators instead of running ONE fp32 chain over all 9 * Cin terms of an output - the blocked
// summation ATen's CPU convolution does; round-off grows with sqrt(terms per block + blocks) instead of sqrt(terms)
// (at 800x800 the unsplit chain was 1.2-1.5x as far from float64 as torch-CPU fp32; tools/train_fwd_error.py)
template <int KS, int STRIDE, int WM, int WK, int NT, int EPI, int RV, int KACC = 0>
__global__ __launch_bounds__(64 * WM * WK) void conv_dma_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // device-only builtin types in the body; the host pass only needs the stub
    using C = DmaCfg<KS, STRIDE, WM, WK, NT, RV>;
    static_assert(RV == 0 || (WK == 1 && EPI == 0 && C::MP == 4 && (RV == 2 || RV % 4 == 0)), "the vector-ALU path is built for WK = 1, plain epilogue");
    constexpr int NTHR = 64 * WM * WK;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = wave / WM;
    // XCD-aware order of tiles and cout groups (conv_mfma.h).  Stride-2 base.2: a 72-float row segment of a 32-pixel-wide
    // tile touches 3-4 128-B lines for 2.25 lines of data; neighbours on the same XCD re-read them from its L2
    int tid_lin, cgroup;
    xcd_tile_order(a.tilesX * a.tilesY, tid_lin, cgroup);
    const int tileY = tid_lin / a.tilesX, tileX = tid_lin - tileY * a.tilesX;
    const int tile0 = cgroup * NT, b = blockIdx.z;   // first cout tile of this workgroup
    const int iy0 = tileY * C::TH * STRIDE - KS / 2, ix0 = tileX * C::TW * STRIDE - C::APRON;

    f32x4 acc[C::MP][NT];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 tot[KACC ? C::MP : 1][KACC ? NT : 1];
#pragma unroll
    for (int m = 0; m < (KACC ? C::MP : 1); ++m)
#pragma unroll
        for (int n = 0; n < (KACC ? NT : 1); ++n) tot[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    int abase[C::MP];
#pragma unroll
    for (int m = 0; m < C::MP; ++m) {
        const int mt = wm * C::MP + m;
        const int ty = mt / C::TWT, tx0 = (mt % C::TWT) * 16;
        abase[m] = (lane >> 4) * C::PLANE + ty * STRIDE * C::IW + (tx0 + (lane & 15)) * STRIDE + (C::APRON - KS / 2);
    }

    // vector-ALU remainder: this lane's pixel inside the tile and its accumulators
    const bool do_rem = RV > 0 && cgroup == (int)gridDim.y - 1;
    const int v_mt = wm * C::MP + (lane >> 4), v_ty = v_mt / C::TWT, v_tx = (v_mt % C::TWT) * 16 + (lane & 15);
    float accv[RV > 0 ? RV : 1];
#pragma unroll
    for (int r = 0; r < (RV > 0 ? RV : 1); ++r) accv[r] = 0.f;

    // ---- per-thread constants of the staging pattern (16-B piece p = it*NTHR + tid of a slot): byte offset of the
    //      piece from the chunk's first channel plane (inputs) / from the chunk's weight block (weights); pieces that
    //      fall outside the image, the channel range or the cout range get an out-of-range offset and read as zeros
    constexpr int NPI = C::KC * C::PP, NITI = (NPI + NTHR - 1) / NTHR;
    constexpr int NPW = NT * C::WFRAG / 4, NITW = (NPW + NTHR - 1) / NTHR;
    const unsigned in_plane = (unsigned)a.Hin * a.Win;
    unsigned voff[NITI], woff[NITW];
    int pcl[NITI];
#pragma unroll
    for (int it = 0; it < NITI; ++it) {
        const int p = it * NTHR + tid;
        const int cl = p / C::PP, q = p - cl * C::PP;
        const int row = q / C::RW, j = q - row * C::RW;
        const int gy = iy0 + row, gx = ix0 + j * 4;
        const bool ok = p < NPI && q < C::RP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        voff[it] = ok ? (cl * in_plane + (unsigned)(gy * a.Win + gx)) * 4u : kDmaOob;
        pcl[it] = cl;
    }
#pragma unroll
    for (int it = 0; it < NITW; ++it) {
        const int p = it * NTHR + tid;
        const int n = p / (C::WFRAG / 4), q = p - n * (C::WFRAG / 4);
        woff[it] = (p < NPW && tile0 + n < a.ntiles) ? ((unsigned)(tile0 + n) * a.nchunks * C::WFRAG + q * 4) * 4u : kDmaOob;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wpk, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(RV > 0 ? a.wrem : a.wpk), 0, 0x7FFFFFFF, 0x00020000);

    float biasv[NT];   // fetched now, used in the epilogue: the latency hides behind the main loop
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = epi_bias(a, (tile0 + n) * 16 + (lane & 15));

    const int cb = a.chunk_begin, nrounds = (a.chunk_end - cb + WK - 1) / WK;
    DPROBE(0);

    // channels of `chunk` that exist (its range may end inside it): k-groups past them are all-zero and are skipped
    auto chunk_valid = [&](int chunk) {
        int ch0 = 0, cend = a.src_cstart[1], c0 = 0;
#pragma unroll
        for (int k = 1; k < kConvMaxSrc; ++k) {
            const bool take = k < a.n_src && chunk >= a.src_chunk0[k];
            ch0 = take ? a.src_chunk0[k] : ch0;
            c0 = take ? a.src_cstart[k] : c0;
            cend = take ? a.src_cstart[k + 1] : cend;
        }
        return (cend - c0) - (chunk - ch0) * C::KC;
    };

    // issue the DMA of one round (WK chunks: inputs + weights) into buffer `buf`.  Which tensor a chunk reads, its
    // first channel and how many of its KC channels exist are workgroup-uniform: select chains on the scalar ALU.
    auto stage = [&](int round, float *buf) {
#pragma unroll
        for (int s = 0; s < WK; ++s) {
            const int chunk = cb + round * WK + s;
            if (chunk >= a.chunk_end) break;
            float *slot = buf + s * C::SLOT;
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
            const int nvalid = (cend - c0) - lc * C::KC;
            const unsigned soff = (unsigned)(coff + lc * C::KC) * in_plane * 4u;
            const __amdgpu_buffer_rsrc_t r =
                __builtin_amdgcn_make_buffer_rsrc((void *)(sp + (size_t)b * ctot * in_plane), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
            for (int it = 0; it < NITI; ++it)
                if (it * NTHR + tid < NPI)   // lanes past the slot's input region are masked off (they would overwrite the weights)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (dma_lds_ptr_t)(slot + (it * NTHR + wave * 64) * 4), 16,
                                                             pcl[it] < nvalid ? voff[it] : kDmaOob, soff, 0, 0);
            float *wslot = slot + C::KC * C::PLANE;
            const unsigned wsoff = (unsigned)chunk * C::WFRAG * 4u;
#pragma unroll
            for (int it = 0; it < NITW; ++it)
                if (it * NTHR + tid < NPW)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (dma_lds_ptr_t)(wslot + (it * NTHR + wave * 64) * 4), 16,
                                                             woff[it], wsoff, 0, 0);
            if (RV > 0 && do_rem) {   // vector-ALU weights of the chunk
                constexpr int NPR = C::RFRAG / 4, NITR = (NPR + NTHR - 1) / NTHR;
#pragma unroll
                for (int it = 0; it < NITR; ++it)
                    if (it * NTHR + tid < NPR)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rrsrc, (dma_lds_ptr_t)(wslot + NT * C::WFRAG + (it * NTHR + wave * 64) * 4), 16,
                                                                 (it * NTHR + tid) * 16u, (unsigned)chunk * C::RFRAG * 4u, 0, 0);
            }
        }
    };

    if (nrounds > 0) stage(0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int round = 0; round < nrounds; ++round) {
        float *cur = smem + (round & 1) * C::BUF;
        if (round + 1 < nrounds) stage(round + 1, smem + ((round + 1) & 1) * C::BUF);   // in flight during the MFMAs
        if (cb + round * WK + wk < a.chunk_end) {
            const float *in_s = cur + wk * C::SLOT;
            const float *w_s = in_s + C::KC * C::PLANE;
            const int nv = chunk_valid(cb + round * WK + wk);
#pragma unroll
            for (int kg = 0; kg < C::KC / 4; ++kg) {
                if (kg > 0 && kg * 4 >= nv) break;   // wave-uniform: the tail of a range shorter than the chunk
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
                    if (RV > 0 && do_rem) {
                        float x0 = af[0], x1 = af[1], x2 = af[2], x3 = af[3];   // [M-tile][channel row] -> [channel][M-tile row]
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
                                     "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
                        if (RV == 2) {   // two channels: two uniform (broadcast) 16-B weight reads and plain FMAs beat the DPP form
                            const f32x4 *wu = reinterpret_cast<const f32x4 *>(w_s + NT * C::WFRAG + (kg * C::KS2 + tap) * RV * 4);
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                const f32x4 w4 = wu[r];
                                accv[r] = fmaf(x0, w4[0], accv[r]);
                                accv[r] = fmaf(x1, w4[1], accv[r]);
                                accv[r] = fmaf(x2, w4[2], accv[r]);
                                accv[r] = fmaf(x3, w4[3], accv[r]);
                            }
                        }
                        const f32x4 *wr = reinterpret_cast<const f32x4 *>(w_s + NT * C::WFRAG + (kg * C::KS2 + tap) * RV * 4) + (lane & 3);
#pragma unroll
                        for (int g = 0; g < RV / 4; ++g) {
                            const f32x4 w4 = wr[g * 4];   // lane l: weights of couts 4g..4g+3 for input channel l&3
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float w = w4[j];
                                PF_FMAC_QUAD(accv[g * 4 + j], w, x0, 0);
                                PF_FMAC_QUAD(accv[g * 4 + j], w, x1, 1);
                                PF_FMAC_QUAD(accv[g * 4 + j], w, x2, 2);
                                PF_FMAC_QUAD(accv[g * 4 + j], w, x3, 3);
                            }
                        }
                    }
                }
            }
        }
        if (KACC) {
#pragma unroll
            for (int m = 0; m < C::MP; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    tot[KACC ? m : 0][KACC ? n : 0] += acc[m][n];
                    acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next round landed
        __syncthreads();                                    // ... and everyone is done with `cur`
    }
    if (KACC) {
#pragma unroll
        for (int m = 0; m < C::MP; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = tot[KACC ? m : 0][KACC ? n : 0];
    }

    DPROBE(1);
    // ---- residual window of this tile -> LDS (all waves, before the K-split partners leave)
    ResWin rw = ResWin();
    const lds_float *res_lds = nullptr;
    const bool has_res = EPI == 1 && a.res && a.res_lds_off >= 0;
    if (has_res) {
        rw = res_window(a, tileY * C::TH, C::TH, tileX * C::TW, C::TW);
        res_stage(a, rw, b, tile0 * 16, NT * 16, (lds_float *)(smem + a.res_lds_off), tid, NTHR);
        res_lds = (const lds_float *)(smem + a.res_lds_off);
        if (WK == 1) __syncthreads();   // (WK > 1: the reduction barrier below orders it)
    }

    DPROBE(2);
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

    DPROBE(3);
    float vmax = 0.f;
    if (EPI == 0) {
        // ---- plain epilogue: bias + ReLU, NCHW float4 stores
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int co = (tile0 + n) * 16 + (lane & 15);
            if (epi_skip(a, co)) continue;
#pragma unroll
            for (int m = 0; m < C::MP; ++m) {
                const int mt = wm * C::MP + m;
                const int oy = tileY * C::TH + mt / C::TWT;
                const int ox = tileX * C::TW + (mt % C::TWT) * 16 + (lane >> 4) * 4;
                if (oy >= a.Hout || ox >= a.Wout) continue;
                f32x4 v = acc[m][n];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += biasv[n];
                    if (a.relu) v[r] = fmaxf(v[r], 0.f);
                }
                epi_store(a, b, co, oy, ox, v, vmax);
            }
        }
        if (RV > 0 && do_rem) {   // remainder channels: lane = pixel, stores coalesced along the row
            const int oy = tileY * C::TH + v_ty, ox = tileX * C::TW + v_tx;
            if (oy < a.Hout && ox < a.Wout) {
#pragma unroll
                for (int r = 0; r < RV; ++r) {
                    const int co = a.ntiles * 16 + r;
                    if (co >= a.Cout) break;
                    float v = accv[r] + a.bias[co];
                    if (a.relu) v = fmaxf(v, 0.f);
                    vmax = range_acc(vmax, v, 0.f, 0.f, 0.f);
                    a.dst[((size_t)b * a.dst_ctotal + a.dst_choff + co) * ((size_t)a.Hout * a.Wout) + (size_t)oy * a.Wout + ox] = v;
                }
            }
        }
    } else {
        // The pixel coordinates are laundered through an empty asm so that hipcc cannot hoist the (loop-invariant)
        // interpolation taps of all MP pixel groups above the main loop, where they cost ~100 live VGPRs (occupancy).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        // ---- fused epilogue (conv_epilogue.h): bias, residual, ReLU, optional 2x2 pool (rows ty, ty+1 = M-tiles
        //      m, m+TWT).  Pixel-group outer, channels inner: the residual taps are computed once for all NT tiles.
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int mt = wm * C::MP + m;
            const int oy = tileY * C::TH + mt / C::TWT;
            const int ox = tileX * C::TW + (mt % C::TWT) * 16 + (lane_e >> 4) * 4;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            const bool pool_top = a.pool && ((m / C::TWT) & 1) == 0 && m + C::TWT < C::MP && oy + 1 < a.Hout;
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
                const f32x4 top = epi_finish(a, b, co, oy, ox, acc[m][n], biasv[n], has_res, chan, &t0);
                if (!a.pool) epi_store(a, b, co, oy, ox, top, vmax);
                else epi_store_pooled(a, b, co, oy, ox, top, epi_finish(a, b, co, oy + 1, ox, acc[(m + C::TWT) % C::MP][n], biasv[n], has_res, chan, &t1), vmax);
                __builtin_amdgcn_sched_barrier(0);   // one fragment at a time: interleaving them costs ~100 VGPRs (occupancy)
            }
            DPROBE(5 + m);
        }
    }
    range_commit(a.status, a.range_slot, vmax);
    DPROBE(4);
#endif
}

// ------------------------------------------------------------------------------------------------
template <int KS, int STRIDE, int WM, int WK, int NT, int EPI, int RV = 0, int KACC = 0>
static int launch_dma_epi(const ConvArgs &a0, int B, hipStream_t s) {
    if constexpr (KACC == 0 && EPI == 0 && RV == 0 && KS == 3) {
        if (a0.kacc) return launch_dma_epi<KS, STRIDE, WM, WK, NT, EPI, RV, 1>(a0, B, s);
    }
    using C = DmaCfg<KS, STRIDE, WM, WK, NT, RV>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    a.res_lds_off = -1;
    if (a.res) {   // staged residual window lives behind the K-split partial sums
        const size_t need = (size_t)C::RED + (size_t)NT * 16 * res_chan_stride(res_extent(C::TH, a.res_sh), res_extent(C::TW, a.res_sw));
        if (need * sizeof(float) > 64 * 1024) return fail(PF_EUNSUPPORTED, "residual window of %zu B does not fit LDS", need * sizeof(float));
        a.res_lds_off = C::RED;
        if (need * sizeof(float) > lds) lds = need * sizeof(float);
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_dma_kernel<KS, STRIDE, WM, WK, NT, EPI, RV, KACC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    char label[96];
    snprintf(label, sizeof(label), KACC ? "void pf::conv_dma_kernel<%d, %d, %d, %d, %d, %d, %d, 1>(pf::ConvArgs)" : "void pf::conv_dma_kernel<%d, %d, %d, %d, %d, %d, %d>(pf::ConvArgs)", KS, STRIDE, WM, WK, NT, EPI, RV);
    if (a.res) strncat(label, " +res", sizeof(label) - strlen(label) - 1);
    if (a.pool) strncat(label, " +pool", sizeof(label) - strlen(label) - 1);
    if (a.no_bias) strncat(label, " lowres-half", sizeof(label) - strlen(label) - 1);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * KS * KS,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * KS * KS));
    hipLaunchKernelGGL((conv_dma_kernel<KS, STRIDE, WM, WK, NT, EPI, RV, KACC>),
                       dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(64 * WM * WK), lds, s, a);
    PF_LAUNCH_CHECK("conv_dma_kernel");
    return PF_OK;
}

template <int KS, int STRIDE, int WM, int WK, int NT>
static int launch_dma_cfg(const ConvArgs &a, int B, hipStream_t s) {
    if (a.pool || a.res || a.no_bias) {
        if (KS == 1 && STRIDE == 1) return launch_dma_epi<KS, STRIDE, WM, WK, NT, (KS == 1 && STRIDE == 1) ? 1 : 0>(a, B, s);
        return fail(PF_EUNSUPPORTED, "conv_dma: fused epilogue stages are built for 1x1 convs only");
    }
    if (a.rem > 0) {   // the last a.rem couts on the vector ALU (a.ntiles = tiles that go through the matrix pipe)
        if (KS == 3 && STRIDE == 1 && WM == 4 && WK == 1 && NT <= 3) {
            constexpr bool ok = KS == 3 && STRIDE == 1 && WM == 4 && WK == 1 && NT <= 3;
            if (a.rem <= 2) return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0, ok ? 2 : 0>(a, B, s);
            if (a.rem <= 4) return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0, ok ? 4 : 0>(a, B, s);
            if (a.rem <= 8) return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0, ok ? 8 : 0>(a, B, s);
            if (a.rem <= 12) return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0, ok ? 12 : 0>(a, B, s);
            if (a.rem <= 16) return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0, ok ? 16 : 0>(a, B, s);
        }
        return fail(PF_EUNSUPPORTED, "conv_dma: the vector-ALU cout path is built for 3x3/s1, WM=4, NT<=3, <=16 channels");
    }
    return launch_dma_epi<KS, STRIDE, WM, WK, NT, 0>(a, B, s);
}

// Cost model (shader cycles) for one (WM, WK, NT) shape: the larger of the chip-wide MFMA time and the serial
// time of one workgroup times the number of workgroup waves; calibrated against profiles/r01_b_*.
static double shape_cost(const ConvArgs &a, int ks, int stride, int B, int wm, int wk, int nt) {
    const int kc = dma_kc(ks, stride), ks2 = ks * ks;
    const int tw = wm == 1 ? 16 : 32, th = wm * 4 / (tw / 16);
    const double px_tiles = (double)B * ((a.Hout + th - 1) / th) * ((a.Wout + tw - 1) / tw);
    const int cblocks = (a.ntiles + nt - 1) / nt;
    const double wgs = px_tiles * cblocks;
    const int nchunks = a.chunk_end - a.chunk_begin, rounds = (nchunks + wk - 1) / wk;
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

int launch_conv_dma(const ConvArgs &a, int ks, int stride, int B, hipStream_t s, int force_wm, int force_nt, int model_B) {
    int wm = 4, wk = 1, nt = 1;
    pick_shape(a, ks, stride, model_B > 0 ? model_B : B, wm, wk, nt);   // the K split (WK) changes the summation order
    if (force_wm > 0) {
        const int model_nt = wm == force_wm ? nt : 2;
        wm = force_wm;
        wk = 4 / wm;
        nt = force_nt > 0 ? force_nt : model_nt;
        nt = nt < a.ntiles ? nt : a.ntiles;
        if (wm == 1 && nt > 2) nt = 2;
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

// [chunk][kgroup][tap][rv/4][4 input channels][4 couts] (rv = 2: [chunk][kgroup][tap][2 couts][4 input channels]):
// the last `rem` output channels, zero padded to rv = dma_rem_rv(rem)
void pack_conv_weights_rem(const float *w, int cin, int cout, int rem, int ks, int kc, const int *src_ch, int n_src, float *out) {
    const int ks2 = ks * ks, rv = dma_rem_rv(rem), co0 = cout - rem;
    size_t o = 0;
    int c0 = 0;
    for (int j = 0; j < n_src; ++j) {
        for (int lc = 0; lc * kc < src_ch[j]; ++lc)
            for (int kg = 0; kg < kc / 4; ++kg)
                for (int tap = 0; tap < ks2; ++tap) {
                    if (rv == 2) {   // [cout][4 input channels]: read as two uniform 16-B rows
                        for (int r = 0; r < 2; ++r)
                            for (int c = 0; c < 4; ++c) {
                                const int cl = lc * kc + kg * 4 + c;
                                out[o++] = (r < rem && cl < src_ch[j]) ? w[((size_t)(co0 + r) * cin + c0 + cl) * ks2 + tap] : 0.f;
                            }
                        continue;
                    }
                    for (int g = 0; g < rv / 4; ++g)
                        for (int c = 0; c < 4; ++c)
                            for (int q = 0; q < 4; ++q) {
                                const int cl = lc * kc + kg * 4 + c, r = g * 4 + q;
                                out[o++] = (r < rem && cl < src_ch[j]) ? w[((size_t)(co0 + r) * cin + c0 + cl) * ks2 + tap] : 0.f;
                            }
                }
        c0 += src_ch[j];
    }
}

int dma_chunks(const int *src_ch, int n_src, int ks, int stride) {
    const int kc = dma_kc(ks, stride);
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += (src_ch[j] + kc - 1) / kc;
    return n;
}

// per-tile packing: [cout tile][chunk][kgroup][tap][64 lanes]; K order = the input ranges in order, each padded to
// whole chunks (zero weights), so a chunk never straddles two source tensors
void pack_conv_weights_tiled(const float *w, int cin, int cout, int ks, int kc, const int *src_ch, int n_src, float *out) {
    const int ks2 = ks * ks, ntiles = (cout + 15) / 16;
    size_t o = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c0 = 0;
        for (int j = 0; j < n_src; ++j) {
            for (int lc = 0; lc * kc < src_ch[j]; ++lc)
                for (int kg = 0; kg < kc / 4; ++kg)
                    for (int tap = 0; tap < ks2; ++tap)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = t * 16 + (lane & 15), cl = lc * kc + kg * 4 + (lane >> 4);
                            out[o++] = (co < cout && cl < src_ch[j]) ? w[((size_t)co * cin + c0 + cl) * ks2 + tap] : 0.f;
                        }
            c0 += src_ch[j];
        }
    }
}

}  // namespace pf
