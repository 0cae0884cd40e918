// Convolutions on activations stored PRE-SPLIT ("S4" layout, conv_mfma.h): the operand-split scheme of conv_split.hip without its
// memory side.
//
// conv_split.hip reads fp32 NCHW, and every consumer of a tensor splits it again: per round of 8 channels a thread holds 32
// staging registers, spends 15-20 % of the round on conversions + 8 ds_write_b128 and needs two barriers because the
// activation buffer cannot be overwritten while it is read (profiles/r02_experiments.md: the memory side and the matrix
// side of a 91->28 layer need 285 and 318 us alone and 406 us together).  Here the PRODUCER's epilogue writes
//     [B][2 terms: hi, mid][C4 = ceil(C/4)][H][W][4 channels] fp16       hi = fp16(x), mid = fp16(x - hi)
// - the same 4 B per element as fp32 and exactly what these consumers feed the matrix pipe with, so nothing is lost for
// them - and a consumer's halo tile arrives by LDS-DMA (buffer_load_dwordx4 ... lds, 16 B = 2 pixels x 4 channels of one
// term; out-of-image pieces carry an out-of-range offset and land as zeros = the convolution's padding): no staging
// registers, no conversion, no LDS stores, double-buffered stages with ONE barrier per round.
//
// Channel groups of 4 (not 8): FC-HarDNet's tensors have 10/18/28/46 channels; a range is read as the groups it touches
// (weights of channels it does not own inside its first/last group are zero), so 10 channels cost 12, not 16.  A block's
// concatenated output tensor is one buffer: layers write their slice (4-B granularity: all offsets are even) and whoever
// reads the whole block reads no padding at all.
//
// 3x3: GEMM view as in conv_split.hip (M = 16 pixels of a row, N = 16 couts, K = 32 = 4 lane groups x 8 values), a lane's 8
// K-values = the 4 channels of TWO group entries for one tap: two ds_read_b64 per term.  A round = 2 entries = 8 channels.
// Nine taps do not fill whole instructions of 4 tap slots (conv_split.hip and the first version of this kernel spent 12
// slots on them: a quarter of the matrix instructions multiplied zeros, and with the memory side out of the way these
// kernels are bound by the matrix pipe - profiles/r02_experiments.md).  Here a round issues TWO full instructions
//     instr 0: (0,0) (0,1) | (1,0) (1,1)        instr 1: (2,0) (2,1) | (0,2) (1,2)
// (the two lane groups one ds_read_b64 pass serves read the same tile row - conflict-free for any row pitch - except the
// last pair, a 2-way conflict on one pass) and the ninth tap (2,2) is COLLECTED: in round r only lane group r & 3 reads
// its (2,2) fragments, into registers that survive the round; after four rounds the four lane groups hold the K-slices
// of four different rounds and one more instruction (weights packed to match) retires them: 9 instructions per 4 rounds
// instead of 12, no zero slots.  Wave w DMAs plane (term = w >> 1, entry = w & 1) of a stage.
// 1x1: K = 32 = 8 group entries per instruction and round; 8x32-pixel tiles, the fused epilogue stages of conv_epilogue.h.
#include <cstring>
#include <vector>

#include "conv_s4.h"
#include "pf_prof.h"

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // shader-clock stamps of one workgroup in the middle of the grid (tools/probe_s4.py)
#define S4_PROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && blockIdx.z == gridDim.z / 2 && a.probe && (i) < 60) a.probe[i] = clock64(); } while (0)
#else
#define S4_PROBE(i) do { } while (0)
#endif

// A/B knobs of profiles/r05_experiments.md (defaults = the shipped kernels): cache-policy bits of the activation LDS-DMA (2 = nt),
// non-temporal epilogue stores of the 3x3 kernel's common path, raised wave priority across the matrix phase of a round
#ifndef S4_ACT_AUX
#define S4_ACT_AUX 0
#endif
#ifndef S4_STORE_NT
#define S4_STORE_NT 0
#endif
#ifndef S4_SETPRIO
#define S4_SETPRIO 0
#endif
// Pixel fragments of the 3x3 kernel: a B operand is 8 fp16 = the 4 channels of TWO group entries of one pixel, which live in two
// planes of a stage, 2880 B apart - out of reach of one ds_read2_b64 (255 x 8 B).  Left alone, hipcc pairs the reads of two
// DIFFERENT M-tiles of one plane into a ds_read2_b64 and then moves the halves into place: 72 v_mov_b32 per round next to 54
// matrix instructions (and a ds_read2_b64 occupies the LDS for 8 cycles where two ds_read_b64 take 4).  1 = volatile 8-B
// reads: no pairing, no moves
#ifndef S4_PREFETCH
#define S4_PREFETCH 0
#endif
#ifndef S4_COL_EARLY
#define S4_COL_EARLY 0
#endif
#ifndef S4_FRAG_B64
#define S4_FRAG_B64 1
#endif
// the DMA parts of the next stage go out after MFMA groups 1, 3, .. of a round (every placement measured within 1 %)
[[maybe_unused]] constexpr int kS4IssueFirst = 1, kS4IssueStep = 2;

namespace pf {

// KS_ = 2 / 4 (the small levels, 64x128 and below at B = 16: a few hundred workgroups of 20-50 dependent rounds on 256 CUs):
// KS_ wave groups per workgroup on the same pixel tile, each with its own stage ring; group k runs the k-th part of the rounds
// (parts of a multiple of 4 rounds: the collected tap spans groups of 4); groups 1.. hand their sums over through LDS and
// group 0 runs the epilogue.  KS_ times the waves per tile and 1 / KS_ of the dependent chain per wave.
template <int NT, int TW_, int TH_ = 8, int KS_ = 1>
struct S4Cfg {
    static constexpr int TW = TW_, TH = TH_, MTR = TW / 16, MP = 2 * MTR;   // TH / 2 waves x MP M-tiles = TH rows x TW pixels
    static constexpr int KS = KS_;
    static constexpr int NW = TH / 2, NTHR = 64 * NW;                       // per wave group.  TH = 16: 8 waves share the stage (half the weight bytes per MAC, halo 1.27 instead of 1.41)
    static constexpr int IW = TW + 4, IH = TH + 2;                        // halo tile, 2-pixel apron left/right (16-B pieces)
    static constexpr int ROWP = IW / 2, PIECES = IH * ROWP;               // 16-B pieces per (term, entry) plane
    static constexpr int NDMA_ALL = (PIECES + 63) / 64;                   // DMA instructions per plane ...
    static constexpr int NDMA = (NDMA_ALL + NW / 4 - 1) / (NW / 4);       // ... of which a wave issues this many (NW / 4 waves per plane)
    // COLREG (the <2, 32> and <3, 32> shapes): the collected-tap block (a third of a flush round's weights, used once in four
    // rounds) does NOT pass through LDS - its fragments are loaded straight into registers right before the flush products -
    // and the planes hold exactly their pieces: 38.5 instead of 48 KB (NT = 2) / 46.5 instead of 60 KB (NT = 3) per workgroup
    // = 4 instead of 3 / 3 instead of 2 workgroups per CU, with the register count capped by the launch bounds (128: 6 values,
    // 168: 8 values parked in scratch across the main loop).  Resident waves are what these kernels are short of
    // (profiles/r02_experiments.md: taking one workgroup per CU away costs 16-52 %; giving one: -3.6 % / -14 %).  The other
    // shapes cannot reach their next workgroup this way (<1, 32> would need 102 registers: 16 spills in the main loop, +35 %;
    // <4, 32> and the 8x64 shapes stay at 2) and keep the third block in LDS, where it costs no exposed load
    static constexpr bool COLREG = (NT == 2 || NT == 3) && TW_ == 32;
    static constexpr int PLANE = COLREG ? PIECES * 16 : NDMA_ALL * 64 * 16;   // bytes (lanes past the last piece are masked off)
    static constexpr int ABUF = 4 * PLANE;                                // [term][entry] per stage
    static constexpr int WBLK = 2 * 64 * 16;                              // one instruction's weights of one cout tile: [term][lane][8 fp16]
    static constexpr int BPT = COLREG ? 2 : 3;                            // blocks per cout tile in LDS: instr 0, instr 1[, collected tap]
    static constexpr int WBUF = NT * BPT * WBLK;                          // [nt][block][term][lane]
    static constexpr int WPIECES = WBUF / 16, NITW = (WPIECES + NTHR - 1) / NTHR;
    static constexpr size_t STAGES_BYTES = 2 * (size_t)ABUF + 2 * (size_t)WBUF;
    static constexpr size_t LDS_BYTES = KS * STAGES_BYTES + NT * 16 * sizeof(float);   // + the bias values of the workgroup's couts
    static_assert(KS == 1 || (size_t)MP * NT * NW * 64 * 16 <= STAGES_BYTES, "the K-split hand-over uses group 1's stage ring");
};

// KACC (training, train_s4.hip): every round's products (72 terms per output: 8 channels x 9 taps) are summed on their own and added
// to a second accumulator set - blocked summation, like the fp32 step's conv_dma (KACC there) and like ATen - instead of one fp32
// chain over all 9 Cin terms: the forward pass of a training step is then as close to float64 as the fp32 step's
#define S4_KERNEL_NAME conv_s4_kernel
#define S4_KERNEL_KACC 0
#define S4_KERNEL_WAVES (KS_ == 4 ? 1 : (TH_ == 16 || KS_ == 2) ? 2 : (TW_ == 32 && NT <= 2) ? 4 : (TW_ == 32 && NT == 3) ? 3 : 2)
#include "conv_s4_kernel.inc"
#undef S4_KERNEL_NAME
#undef S4_KERNEL_KACC
#undef S4_KERNEL_WAVES
// the blocked-sum form (instantiated for 8 x 32 tiles without a K split): one workgroup per CU fewer than the plain form where the
// second accumulator set needs the registers
#define S4_KERNEL_NAME conv_s4_blocked_kernel
#define S4_KERNEL_KACC 1
#define S4_KERNEL_WAVES (NT == 1 ? 4 : (NT == 2 ? 3 : 2))
#include "conv_s4_kernel.inc"
#undef S4_KERNEL_NAME
#undef S4_KERNEL_KACC
#undef S4_KERNEL_WAVES

template <int NT, int TW_, int TH_ = 8, int KS_ = 1>
static int launch_s4_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = S4Cfg<NT, TW_, TH_, KS_>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s4_kernel<NT, TW_, TH_, KS_>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_s4_kernel<%d, %d, %d, %d>(pf::ConvArgs)", NT, TW_, TH_, KS_);   // = the symbol rocprofv3 reports (bench.py looks its PMC bytes up by it)
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * 9,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * 9));
    hipLaunchKernelGGL((conv_s4_kernel<NT, TW_, TH_, KS_>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(C::NTHR * C::KS), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_s4_kernel");
    return PF_OK;
}

template <int NT>
static int launch_s4_blocked_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = S4Cfg<NT, 32, 8, 1>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s4_blocked_kernel<NT, 32, 8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_s4_blocked_kernel<%d, 32, 8, 1>(pf::ConvArgs)", NT);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * 9, 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * 9));
    hipLaunchKernelGGL((conv_s4_blocked_kernel<NT, 32, 8, 1>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(C::NTHR), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_s4_blocked_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------
// 1x1: a streaming kernel (these layers are bound by their bytes: 2*Cin*Cout flops per pixel against 4*(Cin+Cout) bytes).
// One round = 8 group entries (32 channels); lane group g of an instruction supplies entries 2g and 2g+1.  The activation
// fragments come STRAIGHT from memory into registers - in the packed layout a 16-B load is [2 pixels][4 channels] of one
// term, which is the K-slice two M-tiles need (M-tile A = the even pixels of a 32-pixel row segment, M-tile B = the odd
// ones; 16 lanes x 16 B = 256 B contiguous per plane) - so there is no activation staging, and the only barrier per round
// is the one of the weight stages (LDS-DMA, double-buffered).  A wave owns rows 2w, 2w+1 of an 8x32-pixel tile and all NT
// cout tiles.  The D fragments of tiles A and B interleave to 8 consecutive pixels per lane, i.e. two of the 4-pixel
// fragments conv_epilogue.h works on (bias, upsampled residual, ReLU, 2x2 pool, fp32 or packed stores).
// The residual window of the tile (commuted upsample, conv_epilogue.h) is prefetched by 4-B LDS-DMA before the main loop.
template <int NT>
struct S41Cfg {
    static constexpr int TW = 32, TH = 8, MP = 4;
    static constexpr int WBUF = NT * 2 * 64 * 16;                   // [nt][term][lane][8 fp16]
    static constexpr int BIAS_OFF = 2 * WBUF;                       // the workgroup's bias values (DMA, read in the epilogue)
    static constexpr int MAIN = 2 * WBUF + 256;
    static constexpr int WPIECES = WBUF / 16, NITW = (WPIECES + 255) / 256;
};

typedef unsigned s4_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s4_u32x2 __attribute__((ext_vector_type(2)));

template <int NT, int EPI>
__global__ __launch_bounds__(256, NT <= 2 ? 4 : (NT == 3 ? 3 : 2)) void conv_s4_1x1_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = S41Cfg<NT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tid_lin, cgroup;   // XCD-aware order: the cout groups of a tile run back to back on one XCD and share its L2 (conv_mfma.h)
    xcd_tile_order(a.tilesX * a.tilesY, tid_lin, cgroup);
    const int tileY = tid_lin / a.tilesX, tileX = tid_lin - tileY * a.tilesX;
    const int tile0 = cgroup * NT, b = blockIdx.z;
    const size_t plane_bytes = (size_t)a.Hin * a.Win * 8;
    auto wbuf = [&](int i) { return smem_raw + i * C::WBUF; };
    const int g = lane >> 4;

    s4_f32x4 acc[C::MP][NT];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = s4_f32x4{0.f, 0.f, 0.f, 0.f};
    const bool has_res = EPI == 1 && a.res && a.res_lds_off >= 0;
    ResWin rw = ResWin();

    // this lane's pixel pair in rows 2w, 2w+1: byte offset inside a group plane, or -1
    int poff[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int gy = tileY * C::TH + 2 * wave + rr, gx = tileX * C::TW + 2 * (lane & 15);
        poff[rr] = (gy < a.Hin && gx < a.Win) ? (gy * a.Win + gx) * 8 : -1;
    }
    unsigned woff[C::NITW];
#pragma unroll
    for (int it = 0; it < C::NITW; ++it) {
        const int p = it * 256 + tid, n = p / (2 * 64);
        woff[it] = (p < C::WPIECES && tile0 + n < a.ntiles) ? (unsigned)((tile0 + n) * a.nchunks * (2 * 64) + (p - n * (2 * 64))) * 16u : kS4Oob;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)a.wpk, 0, 0x7FFFFFFF, 0x00020000);

    // operand roles swapped (weights = A, pixels = B): lane (g, i) ends up with couts 4g..4g+3 of the pixel pair (2i, 2i+1).
    // Bias values through LDS (see conv_s4_kernel); zero for the low-resolution half of a commuted upsample (no_bias)
    float *bias_lds = reinterpret_cast<float *>(smem_raw + C::BIAS_OFF);
    if (wave == 0 && lane < NT * 16) {
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias, 0, 0x7FFFFFFF, 0x00020000);
        const int co = tile0 * 16 + lane;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (s4_lds_ptr_t)bias_lds, 4, (!a.no_bias && co < a.ntiles * 16) ? (unsigned)co * 4u : kS4Oob, 0,
                                                 0, 0);
    }

    const int cb = a.chunk_begin, nrounds = a.chunk_end - cb;
    // fragments of one round: [row][entry of the pair][term], each 16 B = 2 pixels x 4 channels
    auto load_round = [&](int r, s4_u32x4 (&x)[2][2][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = 8 * (cb + r) + 2 * g + h;   // per lane group
            const float *sp = a.src[0];
            int c4 = a.src_c4[0], g0 = a.src_g0[0], gn = a.src_gn[0], e0 = 0;
#pragma unroll
            for (int k = 1; k < kConvMaxSrc; ++k) {
                const bool take = k < a.n_src && e >= a.src_ent0[k];
                sp = take ? a.src[k] : sp;
                c4 = take ? a.src_c4[k] : c4;
                g0 = take ? a.src_g0[k] : g0;
                gn = take ? a.src_gn[k] : gn;
                e0 = take ? a.src_ent0[k] : e0;
            }
            const bool real = e - e0 < gn;   // padding entries (zero weights) read the zero page
            const char *hi = reinterpret_cast<const char *>(sp) + ((size_t)b * 2 * c4 + (g0 + min(e - e0, gn - 1))) * plane_bytes;
            const size_t tstride = (size_t)c4 * plane_bytes;
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const bool ok = real && poff[rr] >= 0;
                const char *p = ok ? hi + poff[rr] : reinterpret_cast<const char *>(a.zero_page);
                x[rr][h][0] = *reinterpret_cast<const s4_u32x4 *>(p);
                x[rr][h][1] = *reinterpret_cast<const s4_u32x4 *>(ok ? p + tstride : p);
            }
        }
    };
    auto load_weights = [&](int r, unsigned char *wdst) {
#pragma unroll
        for (int it = 0; it < C::NITW; ++it)
            if (it * 256 + tid < C::WPIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (s4_lds_ptr_t)(wdst + (it * 256 + wave * 64) * 16), 16, woff[it],
                                                         (unsigned)(cb + r) * (2 * 64 * 16), 0, 0);
    };

    s4_u32x4 cur[2][2][2], nxt[2][2][2];
    if (nrounds > 0) {
        load_weights(0, wbuf(0));
        load_round(0, cur);
    }
    // residual window -> LDS, asynchronously, behind the first round's loads (fixed-size window of res_rows x res_cols per
    // channel, edge-clamped)
    if (has_res) {
        rw = res_window(a, tileY * C::TH, C::TH, tileX * C::TW, C::TW);
        rw.rows = a.res_rows;
        rw.cols = a.res_cols;
        rw.cs = res_chan_stride(a.res_rows, a.res_cols);
        const unsigned per = (unsigned)(rw.rows * rw.cols), total = (unsigned)(NT * 16 * rw.cs);
        const size_t rplane = (size_t)a.Hres * a.Wres;
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.res + ((size_t)b * a.res_ctotal + a.res_choff + tile0 * 16) * rplane), 0, 0x7FFFFFFF, 0x00020000);
        unsigned char *rdst = smem_raw + a.res_lds_off * 4;
        for (unsigned e0 = (unsigned)wave * 64; e0 < total; e0 += 256) {
            const unsigned e = e0 + lane;
            const unsigned c = __umulhi(e, a.res_magic_cs), rem = e - c * (unsigned)rw.cs;
            const unsigned r = __umulhi(rem, a.res_magic_cols), x = rem - r * (unsigned)rw.cols;
            const int row = min(rw.sy0 + (int)r, a.Hres - 1), col = min(rw.sx0 + (int)x, a.Wres - 1);
            const bool ok = e < total && rem < per && tile0 * 16 + (int)c < a.Cout;
            const unsigned off = ok ? (unsigned)((c * rplane + (size_t)row * a.Wres + col) * 4) : kS4Oob;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rrs, (s4_lds_ptr_t)(rdst + e0 * 4), 4, off, 0, 0, 0);
        }
    }

    for (int round = 0; round < nrounds; ++round) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // weights of this round (and everything older) have landed
        __syncthreads();
        const bool more = round + 1 < nrounds;
        if (more) {
            load_weights(round + 1, wbuf((round + 1) & 1));
            load_round(round + 1, nxt);
        }
        const unsigned char *wb = wbuf(round & 1);
        s4_h8 bh[NT], bm[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[n] = *reinterpret_cast<const s4_h8 *>(wb + ((n * 2 + 0) * 64 + lane) * 16);
            bm[n] = *reinterpret_cast<const s4_h8 *>(wb + ((n * 2 + 1) * 64 + lane) * 16);
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // q = pixel parity: low / high 8 B of the loads
                const s4_u32x4 h0 = cur[rr][0][0], h1 = cur[rr][1][0], m0 = cur[rr][0][1], m1 = cur[rr][1][1];
                const s4_u32x4 fh = q ? s4_u32x4{h0[2], h0[3], h1[2], h1[3]} : s4_u32x4{h0[0], h0[1], h1[0], h1[1]};
                const s4_u32x4 fm = q ? s4_u32x4{m0[2], m0[3], m1[2], m1[3]} : s4_u32x4{m0[0], m0[1], m1[0], m1[1]};
                const s4_h8 ah = __builtin_bit_cast(s4_h8, fh), am = __builtin_bit_cast(s4_h8, fm);
                const int m = rr * 2 + q;
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(bh[n], am, acc[m][n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(bm[n], ah, acc[m][n]);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = PF_MFMA_SPLIT(bh[n], ah, acc[m][n]);
            }
        if (more) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < 2; ++t) cur[rr][h][t] = nxt[rr][h][t];
        }
    }

    // ---- epilogue, lane-local: acc[rr*2 + q][n][r] = cout (tile0+n)*16 + 4g + r at pixel (row 2w + rr, column x0 + 2i + q).
    //      bias -> (+ bilinearly upsampled residual) -> ReLU -> (2x2 average pool: the pixel pair of a lane and its two rows)
    //      -> one [2 px][4 ch] unit per term (16 lanes = 256 B contiguous), [1 px][4 ch] when pooled, or fp32 NCHW pairs
    if (has_res) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // every wave's pieces of the residual window have landed
    }
    {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) the compiler can see: no waits for loads inside the store loop (conv_s4_kernel)
        const lds_float *res_lds = (const lds_float *)(reinterpret_cast<float *>(smem_raw) + (has_res ? a.res_lds_off : 0));
        const int i2 = 2 * (lane & 15);
        const int ox = tileX * C::TW + i2;
        const bool pooled = EPI == 1 && a.pool;
        const int Ho = pooled ? a.Hout >> 1 : a.Hout, Wo = pooled ? a.Wout >> 1 : a.Wout;
        const size_t hw = (size_t)Ho * Wo, term = (size_t)a.dst_c4 * hw * 8;
        const bool mis = (a.dst_choff & 2) != 0;
        typedef split_x2 h2;
        float vmax = 0.f;   // range guard of the operand split (conv_mfma.h)
        const float relu_lo1 = a.relu ? 0.f : -__builtin_inff();
        // finished values of pixel (rr, q), cout tile n
        auto finish = [&](int rr, int q, int n, const int (&o0)[2], const int (&o1)[2], const float (&lx1)[2], float hy1) {
            s4_f32x4 v = acc[rr * 2 + q][n];
            const s4_f32x4 b4 = *reinterpret_cast<const s4_f32x4 *>(bias_lds + n * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = v[r] * a.acc_scale + b4[r];   // acc_scale = 2^-k of the weight scaling: exact
                if (has_res) {   // same arithmetic as res_apply (conv_epilogue.h)
                    const lds_float *chan = res_lds + (n * 16 + 4 * g + r) * rw.cs;
                    const float lx0 = 1.f - lx1[q], hy0 = 1.f - hy1;
                    const float t0 = lx0 * chan[o0[q]] + lx1[q] * chan[o0[q] + 1];
                    const float t1 = lx0 * chan[o1[q]] + lx1[q] * chan[o1[q] + 1];
                    v[r] += hy0 * t0 + hy1 * t1;
                }
                v[r] = fmaxf(v[r], relu_lo1);
            }
            vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
            return v;
        };
        // store 4 channels (couts co..co+3) of ONE output pixel at element offset pix
        auto store_px = [&](int co, size_t pix, s4_f32x4 v) {
            if (a.dst_fmt) {
                s4_h4 hi, mid;
                split_terms4(v, hi, mid);
                const int chb = a.dst_choff + co;
                const bool ok0 = chb < a.dst_limit, ok1 = chb + 2 < a.dst_limit;
                char *p = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * term + pix * 8 + (size_t)(chb >> 2) * hw * 8;
                if (!mis) {
                    if (ok1) {
                        *reinterpret_cast<s4_h4 *>(p) = hi;
                        *reinterpret_cast<s4_h4 *>(p + term) = mid;
                    } else if (ok0) {
                        *reinterpret_cast<h2 *>(p) = h2{hi[0], hi[1]};
                        *reinterpret_cast<h2 *>(p + term) = h2{mid[0], mid[1]};
                    }
                } else {
                    if (ok0) {
                        *reinterpret_cast<h2 *>(p + 4) = h2{hi[0], hi[1]};
                        *reinterpret_cast<h2 *>(p + 4 + term) = h2{mid[0], mid[1]};
                    }
                    if (ok1) {
                        *reinterpret_cast<h2 *>(p + hw * 8) = h2{hi[2], hi[3]};
                        *reinterpret_cast<h2 *>(p + hw * 8 + term) = h2{mid[2], mid[3]};
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < a.Cout) a.dst[((size_t)b * a.dst_ctotal + a.dst_choff + co + r) * hw + pix] = v[r];
            }
        };
        // ... and of the PAIR of pixels (pix, pix + 1): one 16-B unit per term when the group is whole
        auto store_pair = [&](int co, size_t pix, s4_f32x4 v0, s4_f32x4 v1) {
            const int chb = a.dst_choff + co;
            if (a.dst_fmt && !mis && chb + 2 < a.dst_limit) {
                s4_h4 h0, m0, h1, m1;
                split_terms4(v0, h0, m0);
                split_terms4(v1, h1, m1);
                const s4_h8 hi = s4_join(h0, h1), mid = s4_join(m0, m1);
                char *p = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * term + pix * 8 + (size_t)(chb >> 2) * hw * 8;
                *reinterpret_cast<s4_h8 *>(p) = hi;
                *reinterpret_cast<s4_h8 *>(p + term) = mid;
            } else if (!a.dst_fmt) {
                typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co + r < a.Cout)
                        *reinterpret_cast<f2 *>(a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co + r) * hw + pix) = f2{v0[r], v1[r]};
            } else {
                store_px(co, pix, v0);
                store_px(co, pix + 1, v1);
            }
        };
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            if (pooled && rr == 1) continue;   // row 2w+1 is consumed with row 2w
            const int oy = tileY * C::TH + 2 * wave + rr;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            if (pooled && oy + 1 >= a.Hout) continue;
            // interpolation taps of the two pixels (rows oy and, pooled, oy + 1)
            int o0[2] = {0, 0}, o1[2] = {0, 0}, p0[2] = {0, 0}, p1[2] = {0, 0};
            float lx1[2] = {0.f, 0.f}, hy1 = 0.f, hy1b = 0.f;
            if (has_res) {
                int y0, y1;
                float hy0;
                lin_coord(oy, a.res_sh, a.Hres, y0, y1, hy0, hy1);
                int z0 = 0, z1 = 0;
                if (pooled) lin_coord(oy + 1, a.res_sh, a.Hres, z0, z1, hy0, hy1b);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    int x0, x1;
                    float lx0;
                    lin_coord(min(ox + q, a.Wout - 1), a.res_sw, a.Wres, x0, x1, lx0, lx1[q]);
                    o0[q] = (y0 - rw.sy0) * rw.cols + (x0 - rw.sx0);
                    o1[q] = (y1 - rw.sy0) * rw.cols + (x0 - rw.sx0);
                    p0[q] = (z0 - rw.sy0) * rw.cols + (x0 - rw.sx0);
                    p1[q] = (z1 - rw.sy0) * rw.cols + (x0 - rw.sx0);
                }
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = (tile0 + n) * 16 + 4 * g;
                if (co >= a.Cout + 2) continue;   // nothing of this unit is stored (dst_limit <= Cout + 2)
                const s4_f32x4 va = finish(rr, 0, n, o0, o1, lx1, hy1), vb = finish(rr, 1, n, o0, o1, lx1, hy1);
                if (!pooled) {
                    store_pair(co, (size_t)oy * a.Wout + ox, va, vb);
                } else {
                    const s4_f32x4 vc = finish(1, 0, n, p0, p1, lx1, hy1b), vd = finish(1, 1, n, p0, p1, lx1, hy1b);
                    s4_f32x4 pv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = (((va[r] + vb[r]) + vc[r]) + vd[r]) * 0.25f;   // order of avgpool2_kernel
                    if ((oy >> 1) < Ho && (ox >> 1) < Wo) store_px(co, (size_t)(oy >> 1) * Wo + (ox >> 1), pv);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        range_commit(a.status, a.range_slot, vmax);
    }
#endif
}

template <int NT, int EPI>
static int launch_s41_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = S41Cfg<NT>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    size_t lds = C::MAIN;
    a.res_lds_off = -1;
    if (a.res) {
        a.res_rows = res_extent(C::TH, a.res_sh);
        a.res_cols = res_extent(C::TW, a.res_sw);
        const int cs = res_chan_stride(a.res_rows, a.res_cols);
        const size_t need = (size_t)NT * 16 * cs * sizeof(float);
        if (need > 64 * 1024 || (size_t)NT * 16 * cs >= 65536) return fail(PF_EUNSUPPORTED, "conv_s4 1x1: residual window of %zu B does not fit LDS", need);
        a.res_magic_cs = (unsigned)(0x100000000ull / (unsigned)cs) + 1u;          // exact quotients for dividends < 2^16
        a.res_magic_cols = (unsigned)(0x100000000ull / (unsigned)a.res_cols) + 1u;
        a.res_lds_off = C::MAIN / 4;
        lds = C::MAIN + align_up(need, 256);
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_s4_1x1_kernel<NT, EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    char label[112];
    snprintf(label, sizeof(label), "void pf::conv_s4_1x1_kernel<%d, %d>(pf::ConvArgs)", NT, EPI);
    if (a.res) strncat(label, " +res", sizeof(label) - strlen(label) - 1);
    if (a.pool) strncat(label, " +pool", sizeof(label) - strlen(label) - 1);
    if (a.no_bias) strncat(label, " lowres-half", sizeof(label) - strlen(label) - 1);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin, 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin));
    hipLaunchKernelGGL((conv_s4_1x1_kernel<NT, EPI>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B), dim3(256), lds, s, a);
    PF_LAUNCH_CHECK("conv_s4_1x1_kernel");
    return PF_OK;
}

// a.wpk = pack_conv_weights_s4() output, a.nchunks = s4_rounds(), a.src_fmt = 1 (sources in the S4 layout, a.src_c4 /
// src_g0 / src_ent0 filled), a.chunk_begin/chunk_end = the rounds to run.
// ks = 3: nt = cout tiles per workgroup (1..3), wide = 8x64-pixel tiles.  ks = 1: nt = 1..4.
int launch_conv_s4(const ConvArgs &a, int ks, int nt, int wide, int B, hipStream_t s) {
    if (!a.src_fmt) return fail(PF_EINVAL, "conv_s4: sources are not in the S4 layout");
    if ((a.Wout & 3) != 0 || a.Hin != a.Hout || a.Win != a.Wout) return fail(PF_EUNSUPPORTED, "conv_s4: stride 1, width % 4 == 0 only");
    nt = nt < 1 ? 1 : (nt > a.ntiles ? a.ntiles : nt);
    if (ks == 3) {
        if (a.pool || a.res || a.no_bias) return fail(PF_EUNSUPPORTED, "conv_s4 3x3: no fused epilogue stages");
        if (a.chunk_begin != 0 || a.chunk_end != a.nchunks) return fail(PF_EUNSUPPORTED, "conv_s4 3x3: whole K range only");
        if (a.kacc) {      // blocked summation (training): 8 x 32 tiles, 1 - 3 cout tiles
            if (nt == 1) return launch_s4_blocked_cfg<1>(a, B, s);
            if (nt == 2) return launch_s4_blocked_cfg<2>(a, B, s);
            return launch_s4_blocked_cfg<3>(a, B, s);
        }
        if (wide == 2) {   // 16x32-pixel tiles, 8 waves
            if (nt == 1) return launch_s4_cfg<1, 32, 16>(a, B, s);
            return launch_s4_cfg<2, 32, 16>(a, B, s);
        }
        if (wide == 4 && a.nchunks >= 16) {   // 8x32-pixel tiles, FOUR wave groups split the rounds (16 waves on one tile)
            if (nt == 1) return launch_s4_cfg<1, 32, 8, 4>(a, B, s);
            return launch_s4_cfg<2, 32, 8, 4>(a, B, s);
        }
        if (wide >= 3 && a.nchunks >= 8) {   // 8x32-pixel tiles, two wave groups split the rounds (small images, long K)
            if (nt == 1) return launch_s4_cfg<1, 32, 8, 2>(a, B, s);
            return launch_s4_cfg<2, 32, 8, 2>(a, B, s);
        }
        if (wide >= 3) wide = 0;
        if (wide) {
            if (nt == 1) return launch_s4_cfg<1, 64>(a, B, s);
            if (nt == 2) return launch_s4_cfg<2, 64>(a, B, s);
            return launch_s4_cfg<3, 32>(a, B, s);   // 3 cout tiles: the 8x64 stages would leave one workgroup per CU
        }
        if (nt == 1) return launch_s4_cfg<1, 32>(a, B, s);
        if (nt == 2) return launch_s4_cfg<2, 32>(a, B, s);
        if (nt == 3) return launch_s4_cfg<3, 32>(a, B, s);
        return launch_s4_cfg<4, 32>(a, B, s);   // 4 cout tiles per pixel fragment read from LDS (2 workgroups per CU)
    }
    nt = nt > 4 ? 4 : nt;
    const bool fused = a.pool || a.res || a.no_bias;
#define PF_S41(NT_) \
    if (nt == NT_) return fused ? launch_s41_cfg<NT_, 1>(a, B, s) : launch_s41_cfg<NT_, 0>(a, B, s);
    PF_S41(1) PF_S41(2) PF_S41(3) PF_S41(4)
#undef PF_S41
    return PF_EUNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------ host side: weights
// group entries of range j; with pad_sources every range is padded to whole rounds (convs that may run one range at a
// time: the two halves of a commuted upsample + 1x1, hardnet_plan.hip)
static int s4_range_groups(const S4Range &r) { return (r.choff + r.ch + 3) / 4 - r.choff / 4; }
int s4_entries(const S4Range *r, int n_src, int ks, int pad_sources) {
    const int per = ks == 3 ? 2 : 8;
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += pad_sources ? (s4_range_groups(r[j]) + per - 1) / per * per : s4_range_groups(r[j]);
    return n;
}
int s4_rounds(const S4Range *r, int n_src, int ks, int pad_sources) {
    const int per = ks == 3 ? 2 : 8;
    return (s4_entries(r, n_src, ks, pad_sources) + per - 1) / per;
}
size_t s4_packed_floats(const S4Range *r, int n_src, int cout, int ks, int pad_sources) {
    const int rounds = s4_rounds(r, n_src, ks, pad_sources);
    return (size_t)((cout + 15) / 16) * (ks == 3 ? s4_blocks_total(rounds) : rounds) * 2 * 64 * 4;
}

// Blocks of [term 2][lane 64][8 fp16]; lane = (cout n = lane & 15, lane group g = lane >> 4), the lane's 8 values = 4 channels
// of two group entries.  1x1: [tile][round], g = entry pair of the round.  3x3: [tile][per round: instr 0, instr 1, and after
// every 4th (and the last) round the collected-tap block]; instr blocks: g = tap, entries of the round; collected block:
// g = round 4q + g of its group of four, tap (2,2)
void pack_conv_weights_s4(const float *w, int cin, int cout, int ks, const S4Range *r, int n_src, int pad_sources, float *out_f) {
    pack_conv_weights_s4_ex(w, cin, cout, ks, r, nullptr, n_src, pad_sources, out_f);
}
void pack_conv_weights_s4_ex(const float *w, int cin, int cout, int ks, const S4Range *r, const int *cstart, int n_src, int pad_sources, float *out_f) {
    unsigned short *out = reinterpret_cast<unsigned short *>(out_f);
    // entry -> (first conv input channel of the group's channel 0, may be negative; valid channel window)
    std::vector<int> ent_c0, ent_lo, ent_hi;
    const int per = ks == 3 ? 2 : 8;
    int c0 = 0;
    for (int j = 0; j < n_src; ++j) {
        if (cstart) c0 = cstart[j];
        const int g0 = r[j].choff / 4, g1 = (r[j].choff + r[j].ch + 3) / 4;
        for (int g = g0; g < g1; ++g) {
            ent_c0.push_back(c0 + 4 * g - r[j].choff);
            ent_lo.push_back(c0);
            ent_hi.push_back(c0 + r[j].ch);
        }
        while (pad_sources && ent_c0.size() % per != 0) {   // padding entry: no valid channel
            ent_c0.push_back(0);
            ent_lo.push_back(0);
            ent_hi.push_back(0);
        }
        c0 += r[j].ch;
    }
    const int n_ent = (int)ent_c0.size(), rounds = (n_ent + per - 1) / per, ntiles = (cout + 15) / 16;
    size_t o = 0;
    // one block: value of (lane, e) = weight of cout (t, lane & 15), entry ent_of(lane >> 4, e / 4), channel e & 3, tap tap_of(lane >> 4)
    auto block = [&](int t, auto ent_of, auto tap_of) {
        for (int term = 0; term < 2; ++term)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = t * 16 + (lane & 15), g = lane >> 4;
                    const int ent = ent_of(g, e / 4), tap = tap_of(g);
                    float v = 0.f;
                    if (co < cout && tap >= 0 && ent >= 0 && ent < n_ent) {
                        const int ci = ent_c0[ent] + (e & 3);
                        if (ci >= ent_lo[ent] && ci < ent_hi[ent]) v = w[((size_t)co * cin + ci) * ks * ks + tap];
                    }
                    const unsigned short hi = split_host_f16(v);
                    out[o++] = term == 0 ? hi : split_host_f16(v - split_host_f32(hi));
                }
    };
    static const int tap3[2][4] = {{0, 1, 3, 4}, {6, 7, 2, 5}};   // ky * 3 + kx of (instr, lane group): see the kernel's aoff
    for (int t = 0; t < ntiles; ++t)
        for (int rd = 0; rd < rounds; ++rd) {
            if (ks == 1) {
                block(t, [&](int g, int h) { return rd * 8 + g * 2 + h; }, [](int) { return 0; });
                continue;
            }
            for (int s = 0; s < 2; ++s) block(t, [&](int, int h) { return rd * 2 + h; }, [&](int g) { return tap3[s][g]; });
            if ((rd & 3) == 3 || rd == rounds - 1) {
                const int q = rd / 4;
                block(t, [&](int g, int h) { return 4 * q + g < rounds ? (4 * q + g) * 2 + h : -1; }, [](int) { return 8; });
            }
        }
}

// ------------------------------------------------------------------------------------------------ layout conversion
__global__ void s4_pack_kernel(const float *src, unsigned short *dst, int B, int C, int H, int W, unsigned *status) {
    const size_t hw = (size_t)H * W, c4 = (C + 3) / 4, n = (size_t)B * c4 * hw;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i % hw, g = (i / hw) % c4, b = i / (hw * c4);
        for (int r = 0; r < 4; ++r) {
            const int c = (int)g * 4 + r;
            const float x = c < C ? src[((size_t)b * C + c) * hw + px] : 0.f;
            bad = bad || !(fabsf(x) <= kSplitMaxAbs);
            split_x2 h, m;
            split_terms2(x, 0.f, h, m);
            dst[(((b * 2 + 0) * c4 + g) * hw + px) * 4 + r] = __builtin_bit_cast(unsigned short, h[0]);
            dst[(((b * 2 + 1) * c4 + g) * hw + px) * 4 + r] = __builtin_bit_cast(unsigned short, m[0]);
        }
    }
    if (bad && status) atomicOr(status, 1u);   // PF_STATUS_RANGE
}
// the range guard for tensors no kernel of this library produced (dense network inputs): overflow / NaN -> PF_STATUS_RANGE,
// max |x| -> the launch's slot (low side, conv_mfma.h)
__global__ void range_check_kernel(const float *x, size_t n, unsigned *status, unsigned *slot) {
    bool bad = false;
    float m = 0.f;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const s4_f32x4 v = reinterpret_cast<const s4_f32x4 *>(x)[i];
        bad = bad || !(fabsf(v[0]) <= kSplitMaxAbs) || !(fabsf(v[1]) <= kSplitMaxAbs) || !(fabsf(v[2]) <= kSplitMaxAbs) ||
              !(fabsf(v[3]) <= kSplitMaxAbs);
        m = range_acc(m, v[0], v[1], v[2], v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < n - n4 * 4) {
        const float t = x[n4 * 4 + threadIdx.x];
        bad = bad || !(fabsf(t) <= kSplitMaxAbs);
        m = range_acc(m, t, 0.f, 0.f, 0.f);
    }
    if (bad) atomicOr(status, 1u);   // PF_STATUS_RANGE
    range_commit(status, slot, m);
}
int launch_range_check(const float *x, size_t n, unsigned *status, unsigned *slot, hipStream_t s) {
    if (!status || n == 0) return PF_OK;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return fail(PF_EINVAL, "range check: input is not 16-B aligned");
    const size_t blocks = (n / 4 + 255) / 256;
    hipLaunchKernelGGL(range_check_kernel, dim3((unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048)), dim3(256), 0, s, x, n, status, slot);
    PF_LAUNCH_CHECK("range_check_kernel");
    return PF_OK;
}
// end of a forward: a launch whose reported maximum is below kRangeLowMax raises PF_STATUS_RANGE_LOW; the forward's status
// (the live word the kernels ORed into) is published in word 0 and ORed into the sticky word (only the host clears that one);
// the reported maxima move to the "last forward" half of the slot area, and the live word and slots are left cleared for the
// next forward (hardnet_plan.hip: no memset at the start of a forward)
__global__ void range_finalize_kernel(unsigned *st, int live_word, int first_slot, int n_slots, unsigned low_bits, int sticky_word) {
    __shared__ unsigned low;
    if (threadIdx.x == 0) low = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n_slots; i += blockDim.x) {
        const unsigned v = st[first_slot + i];
        // v = bits(max |stored value|) | 1; 0 = no report, 1 = every sampled value was exactly zero: zeros lose nothing in the
        // fp16 pair (a dead-ReLU layer, blank frames, an all-zero dense input), so only a NON-ZERO maximum below 2^-6 is low
        if ((v & ~1u) != 0 && v < low_bits) low = 1;
        st[first_slot + n_slots + i] = v;
        st[first_slot + i] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned w = st[live_word] | (low ? 2u : 0u);   // PF_STATUS_RANGE_LOW
        st[0] = w;
        st[sticky_word] |= w;
        st[live_word] = 0;
    }
}
int launch_range_finalize(unsigned *st, int live_word, int first_slot, int n_slots, int sticky_word, hipStream_t s) {
    hipLaunchKernelGGL(range_finalize_kernel, dim3(1), dim3(256), 0, s, st, live_word, first_slot, n_slots,
                       __builtin_bit_cast(unsigned, kRangeLowMax), sticky_word);
    PF_LAUNCH_CHECK("range_finalize_kernel");
    return PF_OK;
}
__global__ void s4_unpack_kernel(const unsigned short *src, float *dst, int B, int C, int H, int W) {
    const size_t hw = (size_t)H * W, c4 = (C + 3) / 4, n = (size_t)B * C * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t px = i % hw, c = (i / hw) % C, b = i / (hw * C);
        const size_t at = ((c / 4) * hw + px) * 4 + (c & 3);
        const unsigned short h = src[(b * 2 + 0) * c4 * hw * 4 + at], m = src[(b * 2 + 1) * c4 * hw * 4 + at];
        dst[i] = (float)__builtin_bit_cast(split_t, h) + (float)__builtin_bit_cast(split_t, m);
    }
}
int launch_s4_pack(const float *src, void *dst, int B, int C, int H, int W, unsigned *status, hipStream_t s) {
    hipLaunchKernelGGL(s4_pack_kernel, dim3(1024), dim3(256), 0, s, src, (unsigned short *)dst, B, C, H, W, status);
    PF_LAUNCH_CHECK("s4_pack_kernel");
    return PF_OK;
}
int launch_s4_unpack(const void *src, float *dst, int B, int C, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(s4_unpack_kernel, dim3(1024), dim3(256), 0, s, (const unsigned short *)src, dst, B, C, H, W);
    PF_LAUNCH_CHECK("s4_unpack_kernel");
    return PF_OK;
}

}  // namespace pf
