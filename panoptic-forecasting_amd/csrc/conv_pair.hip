// An odd HarDBlock layer computed INSIDE its consumer (round 6; VERDICT r5 item 1: "odd-layer-in-consumer pairs").
//
// reference hardnet.py:177-194 (get_link) / :220-240 (HarDBlock.forward): layer k of a block reads the concatenation of layers
// k - 1, k - 2, k - 4, ... .  For odd k the link list is [k - 1] alone; for the even layer behind it the list starts with that odd
// layer and contains its input too:
//      P = layer 2m+1 = conv3x3(S)                 S = layer 2m (or the block input)
//      C = layer 2m+2 = conv3x3(P ++ S ++ others)
// Today P is a launch of its own (one cout tile per workgroup: the shape with the most activation bytes per MAC), is written to
// the block's output tensor and read back by C, and S streams from memory twice.  Here ONE workgroup owns an 8x32-pixel tile of C:
//   phase 1  the rounds of S.  S is staged with a halo of TWO pixels (12 x 36 instead of 10 x 36 per plane); the same staged
//            planes feed C's accumulators (rows 1..10) and P's accumulators over the tile plus ONE halo pixel (10 x 34 values).
//            P's positions are LINEAR in the staged pitch: position q = i * 36 + j reads S at q + ky * 36 + kx - 1, so a 16-position
//            M-tile may run from one row into the next and 360 positions are 22.5 M-tiles instead of 30 row-aligned ones (the two
//            columns per row nobody needs are computed on garbage and never read: a pixel's K values are all its own);
//   P's epilogue  bias, ReLU, ZERO outside the image (C's zero padding applies to P, not to conv(S outside)), operand split ->
//            LDS planes in the layout of a staged (term, entry) plane, and the tile's own 8 x 32 values -> P's slice of the block
//            output in memory (it is part of the block's output: hardnet.py:233-239);
//   phase 2  C's other ranges by LDS-DMA (10 x 36 planes, written one row down so that C's fragment offsets do not change),
//            then C's rounds over P straight from the LDS planes: no DMA, no memory traffic at all.
// Per pair: one launch instead of two, S read once instead of twice, P never read from memory.  The price: P is computed on
// 360 positions per 256 pixels (1.41x its matrix work - a small layer: 10..18 couts), 79 KB of LDS (2 workgroups per CU).
// Arithmetic = conv_s4.hip's (three fp16 products per fp32 MAC, fp32 accumulate, collected ninth tap); C accumulates its ranges
// in the order [S, others, P] instead of [P, S, others]: same terms, another fp32 summation order.
#include <cstring>
#include <type_traits>
#include <vector>

#include "conv_s4.h"
#include "pf_prof.h"

#ifndef PAIR_DBG
#define PAIR_DBG 0
#endif

namespace pf {

template <int NT, int NTP, int MERGED = 0>
struct PairCfg {
    static constexpr int TW = 32, TH = 8, NW = 8, NTHR = 512, MP = 2;   // wave w: row w of the tile (2 M-tiles) + 3 of P's 23 M-tiles
    static constexpr int IW = TW + 4;                               // pixels per plane row (2-pixel apron: 16-B pieces)
    static constexpr int IH_S = TH + 4, IH_C = TH + 2;              // rows staged for S / for C's other ranges
    static constexpr int ROWP = IW / 2;
    static constexpr int PIECES_S = IH_S * ROWP, PIECES_C = IH_C * ROWP;
    static constexpr int NDMA_S = ((PIECES_S + 63) / 64 + 1) / 2, NDMA_C = ((PIECES_C + 63) / 64 + 1) / 2;   // per wave: two waves share a plane
    static constexpr int PLANE = PIECES_S * 16;                     // bytes of a staged (term, entry) plane
    static constexpr int ABUF = 4 * PLANE;
    static constexpr int WBLK = 2 * 64 * 16;
    static constexpr int WC = NT * 2 * WBLK, WP = NTP * 2 * WBLK;   // instr 0 / instr 1 blocks; the collected-tap blocks go global -> registers
    static constexpr int WBUF = WC + WP;
    static constexpr int DPLANE = IH_C * IW * 8;                    // a derived (term, entry) plane: 10 x 36 positions
    static constexpr int NDENT = 4 * NTP;                           // derived entries (every one is written: couts past P's are zeros)
    static constexpr int DBUF = 2 * NDENT * DPLANE;                 // [term][entry]
    static constexpr int NPOS = IH_C * IW;                          // P positions of a tile
    // P's M-tiles per wave.  Plain: 23 linear M-tiles over all 360 positions, 3 per wave.  MERGED: the tile's OWN 256 positions come
    // out of C's matrix instructions (P's couts ride in the zero-padded rows of C's last cout tile: see the kernel), only the halo
    // ring - row 0, row 9 and columns 1 / 34 of rows 1..8: 84 positions = 6 M-tiles with per-lane positions - is computed here
    static constexpr int PM = MERGED ? 1 : ((NPOS + 15) / 16 + NW - 1) / NW;
    static constexpr int PWAVES = MERGED ? 6 : NW, NHALO = 84;
    static constexpr int PADF = 64, TAIL = 512;                     // the linear taps reach 8 B in front of / ~200 B behind the ring
    static constexpr int MAXR = 64;                                 // rounds that read memory (S + other ranges): the round table below
    static constexpr int OFF_A = PADF, OFF_W = OFF_A + 2 * ABUF + TAIL, OFF_D = OFF_W + 2 * WBUF, OFF_B = OFF_D + DBUF;
    static constexpr int OFF_T = OFF_B + (NT + NTP) * 16 * (int)sizeof(float);   // [round][plane 4] {address of the plane's group in memory, real?}
    static constexpr size_t LDS_BYTES = OFF_T + MAXR * 4 * 16;
    static constexpr int NPARTS = 2;
    static constexpr int NWC = (NT + 1) / 2;                        // weight DMA instructions per thread and round (512 threads: two cout tiles each)
};

// acc[m][n] += w[n] x f[m] as three fp16 products (conv_mfma.h); `tick(slot)` runs after each of the three groups with the
// COMPILE-TIME slot numbers SLOT0 .. SLOT0 + 2 (DMA issue points; SLOT0 < 0: none.  A run-time slot counter cost ~1000 scalar
// instructions and ~290 branches per round: profiles/r06_experiments.md)
template <int NTn, int FTn, int SLOT0, typename Tick>
__device__ __forceinline__ void pair_products(s4_f32x4 (*acc)[NTn], const s4_h8 (&wh)[NTn], const s4_h8 (&wm)[NTn], const s4_h8 (&fh)[FTn],
                                              const s4_h8 (&fm)[FTn], Tick &&tick) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int m = 0; m < FTn; ++m)
#pragma unroll
        for (int n = 0; n < NTn; ++n) acc[m][n] = PF_MFMA_SPLIT(wh[n], fm[m], acc[m][n]);
    if constexpr (SLOT0 >= 0) tick(std::integral_constant<int, SLOT0>());
#pragma unroll
    for (int m = 0; m < FTn; ++m)
#pragma unroll
        for (int n = 0; n < NTn; ++n) acc[m][n] = PF_MFMA_SPLIT(wm[n], fh[m], acc[m][n]);
    if constexpr (SLOT0 >= 0) tick(std::integral_constant<int, SLOT0 + 1>());
#pragma unroll
    for (int m = 0; m < FTn; ++m)
#pragma unroll
        for (int n = 0; n < NTn; ++n) acc[m][n] = PF_MFMA_SPLIT(wh[n], fh[m], acc[m][n]);
    if constexpr (SLOT0 >= 0) tick(std::integral_constant<int, SLOT0 + 2>());
#endif
}

// one pixel's 4 output channels into the packed-pair layout (group tails, ranges that start in the middle of a group): conv_s4.hip
struct PairDst {
    char *base;      // frame base of the destination tensor (hi term)
    size_t term, hw; // bytes between the terms, pixels per plane
    int choff, limit;
};
__device__ __forceinline__ void pair_store_px(const PairDst &d, int co, size_t pix, const s4_f32x4 &v) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef split_x2 h2;
    s4_h4 hi, mid;
    split_terms4(v, hi, mid);
    const int chb = d.choff + co;
    const bool ok0 = chb < d.limit, ok1 = chb + 2 < d.limit;
    char *p = d.base + pix * 8 + (size_t)(chb >> 2) * d.hw * 8;
    if ((d.choff & 2) == 0) {
        if (ok1) {
            *reinterpret_cast<s4_h4 *>(p) = hi;
            *reinterpret_cast<s4_h4 *>(p + d.term) = mid;
        } else if (ok0) {
            *reinterpret_cast<h2 *>(p) = h2{hi[0], hi[1]};
            *reinterpret_cast<h2 *>(p + d.term) = h2{mid[0], mid[1]};
        }
    } else {   // upper half of one group, lower half of the next
        if (ok0) {
            *reinterpret_cast<h2 *>(p + 4) = h2{hi[0], hi[1]};
            *reinterpret_cast<h2 *>(p + 4 + d.term) = h2{mid[0], mid[1]};
        }
        if (ok1) {
            *reinterpret_cast<h2 *>(p + d.hw * 8) = h2{hi[2], hi[3]};
            *reinterpret_cast<h2 *>(p + d.hw * 8 + d.term) = h2{mid[2], mid[3]};
        }
    }
#endif
}

template <int NT, int NTP, int MERGED>
__global__ __launch_bounds__(512, 4) void conv_pair_kernel(PairArgs pa) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = PairCfg<NT, NTP, MERGED>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const ConvArgs &a = pa.c;
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int tid_lin, cgroup;
    xcd_tile_order(a.tilesX * a.tilesY, tid_lin, cgroup);   // gridDim.y == 1: the workgroup owns every cout of C
    const int tileY = tid_lin / a.tilesX, tileX = tid_lin - tileY * a.tilesX;
    const int b = blockIdx.z;
    const int Y0 = tileY * C::TH, X0 = tileX * C::TW;
    const size_t plane_bytes = (size_t)a.Hin * a.Win * 8;
    auto abuf = [&](int i) { return smem + C::OFF_A + i * C::ABUF; };
    auto wbuf = [&](int i) { return smem + C::OFF_W + i * C::WBUF; };
    unsigned char *const dbuf = smem + C::OFF_D;
    float *const bias_lds = reinterpret_cast<float *>(smem + C::OFF_B);
    const int RS = pa.rounds_s, RD = pa.round_d, nrounds = a.nchunks;

    s4_f32x4 acc[C::MP][NT], pacc[C::PM][NTP], pacc9[C::PM][NTP];
#pragma unroll
    for (int m = 0; m < C::MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = s4_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < C::PM; ++m)
#pragma unroll
        for (int n = 0; n < NTP; ++n) pacc[m][n] = pacc9[m][n] = s4_f32x4{0.f, 0.f, 0.f, 0.f};

    // this lane's pieces of the plane its wave helps to fetch (plane w & 3: term (w & 3) >> 1, entry w & 1 of the round; waves w and
    // w + 4 take alternate 64-piece instructions of it), both geometries
    const int w4 = wave & 3, half = wave >> 2;
    unsigned poffS[C::NDMA_S], poffC[C::NDMA_C];
#pragma unroll
    for (int j = 0; j < C::NDMA_S; ++j) {
        const int p = (2 * j + half) * 64 + lane, row = p / C::ROWP, cp = p - row * C::ROWP;
        const int gy = Y0 - 2 + row, gx = X0 - 2 + 2 * cp;
        poffS[j] = (p < C::PIECES_S && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? (unsigned)(gy * a.Win + gx) * 8u : kS4Oob;
    }
#pragma unroll
    for (int j = 0; j < C::NDMA_C; ++j) {
        const int p = (2 * j + half) * 64 + lane, row = p / C::ROWP, cp = p - row * C::ROWP;
        const int gy = Y0 - 1 + row, gx = X0 - 2 + 2 * cp;
        poffC[j] = (p < C::PIECES_C && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? (unsigned)(gy * a.Win + gx) * 8u : kS4Oob;
    }
    const int nblocksC = s4_blocks_total(nrounds), nblocksP = 2 * RS;   // P's stream: instr 0 / instr 1 of every round (pack_conv_weights_s4_two); its ninth tap: p_w9
    // weight pieces: a cout tile's two instruction blocks of a round are 256 16-B pieces; threads 0..255 fetch tile 2i, threads
    // 256..511 tile 2i + 1 (uniform per wave)
    unsigned woffC[C::NWC], woffP;
#pragma unroll
    for (int i = 0; i < C::NWC; ++i) {
        const int n = 2 * i + half;
        woffC[i] = (n < NT && n < a.ntiles) ? ((unsigned)n * (unsigned)nblocksC * (2 * 64) + (unsigned)(tid & 255)) * 16u : kS4Oob;
    }
    woffP = (half < NTP && half < pa.p_ntiles) ? ((unsigned)half * (unsigned)nblocksP * (2 * 64) + (unsigned)(tid & 255)) * 16u : kS4Oob;
    const __amdgpu_buffer_rsrc_t wrsC = __builtin_amdgcn_make_buffer_rsrc((void *)a.wpk, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsP = __builtin_amdgcn_make_buffer_rsrc((void *)pa.p_wpk, 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs9 = __builtin_amdgcn_make_buffer_rsrc((void *)pa.p_w9, 0, 0x7FFFFFFF, 0x00020000);

    // fragment byte offsets.  C: as conv_s4.hip (lane group g = tap of instr 0 / 1), in a plane whose row 0 is image row Y0 - 1;
    // the staged planes start one row higher (+ IW * 8).  P: linear positions, wave w owns positions [48 w, 48 w + 48)
    const int g = lane >> 4;
    int aoff[2], aoff_col, paoff[2], paoff9;
    // P position of this lane's pixel in M-tile 0 of its wave (M-tile m of the plain form: + 16 m)
    int q0 = wave * C::PM * 16 + (lane & 15);
    bool phalo = true;          // MERGED: the lane's halo position exists (84 of 6 x 16)
    if (MERGED) {
        const int k = wave * 16 + (lane & 15);
        phalo = k < C::NHALO;
        const int s8 = k - 68;
        q0 = k < 34 ? 1 + k : (k < 68 ? 9 * C::IW + 1 + (k - 34) : (phalo ? C::IW * (1 + (s8 >> 1)) + ((s8 & 1) ? C::TW + 2 : 1) : 0));
    }
    {
        const int ky[2] = {g >> 1, g < 2 ? 2 : g - 2};
        const int kx[2] = {g & 1, g < 2 ? g : 2};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            aoff[s] = ((wave + ky[s]) * C::IW + (lane & 15) + kx[s] + 1) * 8;
            paoff[s] = (q0 + ky[s] * C::IW + kx[s] - 1) * 8;
        }
        aoff_col = ((wave + 2) * C::IW + (lane & 15) + 2 + 1) * 8;
        paoff9 = (q0 + 2 * C::IW + 2 - 1) * 8 + (g == 1 ? C::PLANE : 0);   // (groups 2, 3: any finite values, their weights are zero)
    }

    if (wave == 0) {   // bias values of both convs -> LDS (conv_s4.hip: kept in registers they cost spills and a wait on the stores)
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)pa.p_bias, 0, 0x7FFFFFFF, 0x00020000);
        if (lane < NT * 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(brs, (s4_lds_ptr_t)bias_lds, 4, lane < a.ntiles * 16 ? (unsigned)lane * 4u : kS4Oob, 0, 0, 0);
        if (lane < NTP * 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(prs, (s4_lds_ptr_t)(bias_lds + NT * 16), 4, lane < pa.p_ntiles * 16 ? (unsigned)lane * 4u : kS4Oob, 0, 0, 0);
    }

    // Round table: which tensor / group / term plane (r, w4) of a stage comes from is SCALAR bookkeeping over the consumer's
    // ranges (s4_entry: ~40 scalar instructions per round and ~25 SGPRs of range descriptors live across the loop - the
    // kernel spilled 60-110 SGPRs).  It is computed once, one (round, plane) per thread, into LDS; a round then reads its 16 B
    typedef unsigned tab_u32x4 __attribute__((ext_vector_type(4)));
    tab_u32x4 *const rtab = reinterpret_cast<tab_u32x4 *>(smem + C::OFF_T);
    if (tid < 4 * RD) {
        const int r = tid >> 2, w = tid & 3;
        const char *base;
        unsigned goff, tstride;
        const bool real = s4_entry(a, 2 * r + (w & 1), b, plane_bytes, base, goff, tstride);
        const unsigned long long addr = reinterpret_cast<unsigned long long>(base) + goff + (unsigned long long)(w >> 1) * tstride;
        rtab[tid] = tab_u32x4{(unsigned)addr, (unsigned)(addr >> 32), real ? 1u : 0u, 0u};
    }
    __syncthreads();
    tab_u32x4 ent = {0, 0, 0, 0};
    auto fetch_round = [&](int r) {      // (the LDS read: issued at the top of a round, consumed at the first DMA slot)
        if (r < RD) ent = *reinterpret_cast<const volatile __attribute__((address_space(3))) tab_u32x4 *>(
                              (const __attribute__((address_space(3))) unsigned char *)(smem + C::OFF_T + (r * 4 + w4) * 16));
    };
    __amdgpu_buffer_rsrc_t ars;
    bool areal = true;
    auto prepare_round = [&](int r) {
        if (r >= RD) return;
        const unsigned lo = __builtin_amdgcn_readfirstlane(ent[0]), hi = __builtin_amdgcn_readfirstlane(ent[1]);
        areal = __builtin_amdgcn_readfirstlane(ent[2]) != 0;
        ars = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
    };
    // part j of round r's stage: one activation piece per lane (none in derived rounds) + a share of the weight pieces
    auto issue_part = [&](int r, auto part_tag) {
        constexpr int j = decltype(part_tag)::value;
        unsigned char *const ad = abuf(r & 1) + w4 * C::PLANE;
        if (r < RS) {
#pragma unroll
            for (int q = 0; q < C::NDMA_S; ++q)
                if (q == j && (2 * q + half) * 64 + lane < C::PIECES_S)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (s4_lds_ptr_t)(ad + (2 * q + half) * 1024), 16, areal ? poffS[q] : kS4Oob, 0, 0, 0);
        } else if (r < RD) {
#pragma unroll
            for (int q = 0; q < C::NDMA_C; ++q)
                if (q == j && (2 * q + half) * 64 + lane < C::PIECES_C)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (s4_lds_ptr_t)(ad + C::IW * 8 + (2 * q + half) * 1024), 16, areal ? poffC[q] : kS4Oob, 0, 0, 0);
        }
        unsigned char *const wd = wbuf(r & 1);
        const unsigned wsoff = (unsigned)s4_blocks_before(r) * C::WBLK, wsoffP = (unsigned)(2 * r) * C::WBLK;
#pragma unroll
        for (int k = 0; k < C::NWC + 1; ++k) {
            if (k % C::NPARTS != j) continue;
            if (k < C::NWC) {
                if (2 * k + half < NT)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsC, (s4_lds_ptr_t)(wd + (2 * k + half) * 2 * C::WBLK + w4 * 1024), 16, woffC[k], wsoff, 0, 0);
            } else if (r < RS && half < NTP) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsP, (s4_lds_ptr_t)(wd + C::WC + half * 2 * C::WBLK + w4 * 1024), 16, woffP, wsoffP, 0, 0);
            }
        }
    };

    const s4_h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    s4_h8 col_h[C::MP], col_m[C::MP];
#pragma unroll
    for (int m = 0; m < C::MP; ++m) col_h[m] = col_m[m] = zero8;

    typedef const volatile __attribute__((address_space(3))) s4_h4 *lds_h4;
    // a pixel fragment: the 4 channels of two entries x two terms, four plain 8-B reads (conv_s4.hip: S4_FRAG_B64)
    auto frag = [&](const unsigned char *p, int ent_stride8, int term_stride8, s4_h8 &h, s4_h8 &md) {
        const lds_h4 q = (lds_h4)(const __attribute__((address_space(3))) unsigned char *)p;
        h = s4_join(q[0], q[ent_stride8]);
        md = s4_join(q[term_stride8], q[term_stride8 + ent_stride8]);
    };
    auto mtile_off = [&](int mm) { return mm * 16 * 8; };
    constexpr int FT = C::MP;               // one batch of (volatile) fragment reads per instruction: 16 fragment registers

    fetch_round(0);
    prepare_round(0);
    issue_part(0, std::integral_constant<int, 0>());
    issue_part(0, std::integral_constant<int, 1>());
    static_assert(C::NPARTS == 2, "two DMA parts per round: after product groups 1 and 3 of C's first instruction pair");

    // One round.  PH = 0: a round of S (C from the stage + P), 1: a round of C's other ranges, 2: a round over P's planes in LDS.
    // Three loops of one kind each (P's epilogue between the first two): the epilogue's addresses are not loop invariants of
    // a loop they are used in once (hoisted, they were 25 spilled registers), and P's accumulators are dead after phase 1
    auto round_body = [&](int R, auto phase_tag) {
        constexpr int PH = decltype(phase_tag)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the stage have landed
        __syncthreads();                                    // everyone's have; everyone is done with the other stage; P's planes are visible
        const bool more = R + 1 < nrounds;
        if (more) fetch_round(R + 1);
        const unsigned char *const ab = abuf(R & 1), *const wb = wbuf(R & 1);
        // the two DMA parts of the next stage go out after product groups 1 and 3 of C (C has 6 groups in every round)
        auto tick = [&](auto slot_tag) {
            constexpr int slot = decltype(slot_tag)::value;
            if constexpr (slot == 1) {
                if (more) {
                    prepare_round(R + 1);
                    issue_part(R + 1, std::integral_constant<int, 0>());
                }
            }
            if constexpr (slot == 3) { if (more) issue_part(R + 1, std::integral_constant<int, 1>()); }
        };
        auto no_tick = [](auto) {};
        // C's collected ninth tap is retired every fourth round and at the end - and, MERGED, after the last round of S: P's own
        // pixels are read out of C's accumulators then, and need the (2,2) products of rounds 4q .. RS - 1 too.  The block used is the
        // one of the group's regular flush round rf (its weights for rounds that have not been collected yet meet zeroed fragments)
        const bool flushC = (R & 3) == 3 || R == nrounds - 1 || (MERGED && R == RS - 1);
        const int rf = min(R | 3, nrounds - 1);

        // ---- C over this round's two entries: from the stage, or from P's planes in LDS
        {
            constexpr bool DER = PH == 2;
            constexpr int ES = (DER ? C::DPLANE : C::PLANE) / 8, TS = (DER ? C::NDENT * C::DPLANE : 2 * C::PLANE) / 8;
            const unsigned char *const cb = DER ? dbuf + (R - RD) * 2 * C::DPLANE : ab + C::IW * 8;
            auto instr = [&](auto s_tag) {
                constexpr int s = decltype(s_tag)::value;
                s4_h8 wh[NT], wm[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    wh[n] = *reinterpret_cast<const s4_h8 *>(wb + (((n * 2 + s) * 2 + 0) * 64 + lane) * 16);
                    wm[n] = *reinterpret_cast<const s4_h8 *>(wb + (((n * 2 + s) * 2 + 1) * 64 + lane) * 16);
                }
                s4_h8 fh[FT], fm[FT];
#pragma unroll
                for (int m = 0; m < FT; ++m) frag(cb + aoff[s] + mtile_off(m), ES, TS, fh[m], fm[m]);
                pair_products<NT, FT, 3 * s>(&acc[0], wh, wm, fh, fm, tick);
                __builtin_amdgcn_sched_barrier(0);
            };
            instr(std::integral_constant<int, 0>());
            instr(std::integral_constant<int, 1>());
            if (g == (R & 3)) {   // the ninth tap of this round's entries: K-slice R & 3 of the collected fragments
#pragma unroll
                for (int m = 0; m < C::MP; ++m) frag(cb + aoff_col + mtile_off(m), ES, TS, col_h[m], col_m[m]);
            }
        }
        if (flushC) {
            s4_h8 cwh[NT], cwm[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                // (buffer loads: scalar base + one lane offset; 64-bit lane pointers per block were spilled)
                const bool real = n < a.ntiles;   // uniform
                const unsigned so = (unsigned)((real ? n : 0) * nblocksC + s4_blocks_before(rf) + 2) * C::WBLK;
                cwh[n] = real ? __builtin_bit_cast(s4_h8, __builtin_amdgcn_raw_buffer_load_b128(wrsC, lane * 16, so, 0)) : zero8;
                cwm[n] = real ? __builtin_bit_cast(s4_h8, __builtin_amdgcn_raw_buffer_load_b128(wrsC, lane * 16, so + 64 * 16, 0)) : zero8;
            }
            pair_products<NT, C::MP, -1>(&acc[0], cwh, cwm, col_h, col_m, no_tick);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < C::MP; ++m) col_h[m] = col_m[m] = zero8;
        }

        // ---- P over the same staged entries, on the tile plus one halo pixel (linear positions).  P's ninth tap is NOT collected over
        //      four rounds like C's (that costs 8 registers per M-tile across the whole phase; at 16 waves per CU there are 128): it is
        //      one K = 16 instruction per product and round - lane group 0 / 1 = the (2,2) tap of the round's first / second entry,
        //      groups 2 and 3 zero weights - with its weights straight from memory (L2) into 4 registers per cout tile
        if (PH == 0 && wave < C::PWAVES) {      // (MERGED: six halo M-tiles, waves 6 and 7 have none)
            constexpr int ES = C::PLANE / 8, TS = 2 * C::PLANE / 8;
            s4_h4 w9h[NTP], w9m[NTP];
#pragma unroll
            for (int n = 0; n < NTP; ++n) {
                const bool real = n < pa.p_ntiles;   // uniform
                const unsigned so = (unsigned)((real ? n : 0) * RS + R) * (2 * 64 * 8);
                w9h[n] = __builtin_bit_cast(s4_h4, __builtin_amdgcn_raw_buffer_load_b64(wrs9, lane * 8, so, 0));
                w9m[n] = __builtin_bit_cast(s4_h4, __builtin_amdgcn_raw_buffer_load_b64(wrs9, lane * 8, so + 64 * 8, 0));
                if (!real) w9h[n] = w9m[n] = s4_h4{0, 0, 0, 0};
            }
            // every product group runs over the wave's three M-tiles (independent accumulators: a chain of dependent matrix
            // instructions issues at half rate)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                s4_h8 wh[NTP], wm[NTP];
#pragma unroll
                for (int n = 0; n < NTP; ++n) {
                    wh[n] = *reinterpret_cast<const s4_h8 *>(wb + C::WC + (((n * 2 + s) * 2 + 0) * 64 + lane) * 16);
                    wm[n] = *reinterpret_cast<const s4_h8 *>(wb + C::WC + (((n * 2 + s) * 2 + 1) * 64 + lane) * 16);
                }
                s4_h8 fh[C::PM], fm[C::PM];
#pragma unroll
                for (int m = 0; m < C::PM; ++m) frag(ab + paoff[s] + m * 128, ES, TS, fh[m], fm[m]);
                pair_products<NTP, C::PM, -1>(&pacc[0], wh, wm, fh, fm, no_tick);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                // The K = 16 products go to accumulators of their OWN (added in the epilogue).  Round 6 measured why: a K = 16 matrix
                // instruction issued right behind a K = 32 one on the SAME accumulator read the accumulator before the K = 32 result
                // had landed - intermittently, one M-tile of one wave in a few tiles per launch (hipcc pads that dependency for equal
                // opcodes only; profiles/r06_experiments.md)
                const lds_h4 q9 = (lds_h4)(const __attribute__((address_space(3))) unsigned char *)(ab + paoff9);
                s4_h4 f9h[C::PM], f9m[C::PM];
#pragma unroll
                for (int m = 0; m < C::PM; ++m) {
                    f9h[m] = q9[m * 16];
                    f9m[m] = q9[m * 16 + TS];
                }
#pragma unroll
                for (int m = 0; m < C::PM; ++m)
#pragma unroll
                    for (int n = 0; n < NTP; ++n) pacc9[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(w9h[n], f9m[m], pacc9[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < C::PM; ++m)
#pragma unroll
                    for (int n = 0; n < NTP; ++n) pacc9[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(w9m[n], f9h[m], pacc9[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < C::PM; ++m)
#pragma unroll
                    for (int n = 0; n < NTP; ++n) pacc9[m][n] = __builtin_amdgcn_mfma_f32_16x16x16f16(w9h[n], f9h[m], pacc9[m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if PAIR_DBG & 1
        __syncthreads();
#endif
    };
    int R = 0;
    for (; R < RS; ++R) round_body(R, std::integral_constant<int, 0>());
#if PAIR_DBG & 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#endif
    {
        // ---- P's epilogue: bias, ReLU, zero outside the image, split -> LDS planes (+ the tile's own pixels -> memory)
        const int px = lane & 15;
        const size_t hw = (size_t)a.Hout * a.Wout;
        PairDst pd;
        pd.term = (size_t)pa.p_dst_c4 * hw * 8;
        pd.hw = hw;
        pd.base = reinterpret_cast<char *>(pa.p_dst) + (size_t)b * 2 * pd.term;
        pd.choff = pa.p_dst_choff;
        pd.limit = pa.p_dst_limit;
        const float relu_lo = pa.p_relu ? 0.f : -__builtin_inff();
        float vmax = 0.f;
        if (wave < C::PWAVES) {
#pragma unroll
        for (int m = 0; m < C::PM; ++m) {
            const int q = q0 + m * 16;
            const int i = (q * 1821) >> 16, j = q - i * C::IW;       // q / 36 for q < 1024
            const int y = Y0 - 1 + i, x = X0 - 2 + j;
            // (columns 0 and 35 of a row were computed on the neighbouring rows' pixels: never read by C, kept out of the planes'
            //  numbers and of the range guard)
            const bool inplane = MERGED ? phalo : q < C::NPOS;
            const bool inimg = inplane && j >= 1 && j <= C::TW + 2 && y >= 0 && y < a.Hout && x >= 0 && x < a.Wout;
            const bool own = !MERGED && inimg && i >= 1 && i <= C::TH && j >= 2 && j < 2 + C::TW;
#pragma unroll
            for (int n = 0; n < NTP; ++n) {
                s4_f32x4 v = pacc[m][n] + pacc9[m][n];
                const s4_f32x4 b4 = *reinterpret_cast<const s4_f32x4 *>(bias_lds + (NT + n) * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inimg ? fmaxf(v[r] * pa.p_acc_scale + b4[r], relu_lo) : 0.f;
                vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
                if (inplane) {
                    s4_h4 hi, mid;
                    split_terms4(v, hi, mid);
                    unsigned char *dp = dbuf + (n * 4 + g) * C::DPLANE + q * 8;
                    *reinterpret_cast<s4_h4 *>(dp) = hi;
                    *reinterpret_cast<s4_h4 *>(dp + C::NDENT * C::DPLANE) = mid;
                }
                const int co = n * 16 + 4 * g;
                if (own && co < pa.p_cout + 2) pair_store_px(pd, co, (size_t)y * a.Wout + x, v);
            }
        }
        }
        if (MERGED) {
            // the tile's own pixels: P's couts were rows 4 .. 4 + p_cout - 1 of C's LAST cout tile during the rounds of S (their
            // weights sit in the rows C pads with zeros, conv_mfma.h: PairArgs::merged), so lane group g >= 1 of that tile's
            // accumulators holds P's couts 4 (g - 1) .. of the wave's two M-tiles; lane group 0 (C's own couts) zero-fills the
            // plane entry no cout of P reaches (its consumer weights are zero, but 0 x stale LDS bits may be NaN)
#pragma unroll
            for (int m = 0; m < C::MP; ++m) {
                const int oy = Y0 + wave, ox = X0 + m * 16 + px;
                const int q = (wave + 1) * C::IW + m * 16 + px + 2;
                const bool inimg = oy < a.Hout && ox < a.Wout;
                s4_f32x4 v = acc[m][NT - 1];
                const int pg = g > 0 ? g - 1 : 3;             // plane entry this lane writes
                const s4_f32x4 b4 = *reinterpret_cast<const s4_f32x4 *>(bias_lds + NT * 16 + 4 * pg);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (inimg && g > 0) ? fmaxf(v[r] * pa.p_acc_scale + b4[r], relu_lo) : 0.f;
                vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
                s4_h4 hi, mid;
                split_terms4(v, hi, mid);
                unsigned char *dp = dbuf + pg * C::DPLANE + q * 8;
                *reinterpret_cast<s4_h4 *>(dp) = hi;
                *reinterpret_cast<s4_h4 *>(dp + C::NDENT * C::DPLANE) = mid;
                if (inimg && g > 0 && 4 * pg < pa.p_cout + 2) pair_store_px(pd, 4 * pg, (size_t)oy * a.Wout + ox, v);
            }
        }
#if !(PAIR_DBG & 4)
        range_commit(a.status, pa.p_range_slot, vmax);
#endif
    }
    for (; R < RD; ++R) round_body(R, std::integral_constant<int, 1>());
    for (; R < nrounds; ++R) round_body(R, std::integral_constant<int, 2>());

    // ---- C's epilogue (conv_s4.hip): lane (g, i) holds couts 4g..4g+3 of pixel i of every M-tile
    {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        const int px = lane & 15;
        const size_t hw = (size_t)a.Hout * a.Wout;
        PairDst cd;
        cd.term = (size_t)a.dst_c4 * hw * 8;
        cd.hw = hw;
        cd.base = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * cd.term;
        cd.choff = a.dst_choff;
        cd.limit = a.dst_limit;
        const float relu_lo = a.relu ? 0.f : -__builtin_inff();
        float vmax = 0.f;
#pragma unroll
        for (int m = 0; m < C::MP; ++m) {
            const int oy = Y0 + wave, ox = X0 + m * 16 + px;
            if (oy >= a.Hout || ox >= a.Wout) continue;
            const size_t pix = (size_t)oy * a.Wout + ox;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int co = n * 16 + 4 * g;
                if (co >= a.Cout + 2) continue;
                s4_f32x4 v = acc[m][n];
                const s4_f32x4 b4 = *reinterpret_cast<const s4_f32x4 *>(bias_lds + n * 16 + 4 * g);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] * a.acc_scale + b4[r], relu_lo);
                vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
                pair_store_px(cd, co, pix, v);
            }
        }
        range_commit(a.status, a.range_slot, vmax);
    }
#endif
}

template <int NT, int NTP, int MERGED = 0>
static int launch_pair_cfg(const PairArgs &pa0, int B, hipStream_t s) {
    using C = PairCfg<NT, NTP, MERGED>;
    PairArgs pa = pa0;
    ConvArgs &a = pa.c;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pair_kernel<NT, NTP, MERGED>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_pair_kernel<%d, %d, %d>(pf::PairArgs)", NT, NTP, MERGED);
    const double px = (double)B * a.Hout * a.Wout;
    const int cin_mem = a.Cin - pa.p_cout;                          // C's input channels that come from memory
    ProfScope ps(s, label, 2.0 * px * 9 * ((double)a.Cout * a.Cin + (double)pa.p_cout * pa.p_cin),
                 4.0 * (px * (cin_mem + a.Cout + pa.p_cout) + 9.0 * ((double)a.Cout * a.Cin + (double)pa.p_cout * pa.p_cin)));
    hipLaunchKernelGGL((conv_pair_kernel<NT, NTP, MERGED>), dim3(a.tilesX * a.tilesY, 1, B), dim3(C::NTHR), C::LDS_BYTES, s, pa);
    PF_LAUNCH_CHECK("conv_pair_kernel");
    return PF_OK;
}

bool conv_pair_supports(int c_cout, int p_cout) { return c_cout <= 48 && p_cout <= 32; }
// merged: built for two cout tiles of C, P in at most three lane groups of the second (the fourth plane entry is zero-filled by C's lanes)
bool conv_pair_merged_supports(int c_cout, int p_cout) {
    return c_cout > 16 && c_cout <= 32 && p_cout <= 12 && (c_cout + 3) / 4 * 4 - 16 == 4 && p_cout + 20 <= 32;
}

// ---- host side: P's weights.  `two`: blocks of [term 2][lane 64][8 fp16] as in pack_conv_weights_s4 (lane = cout n = lane & 15,
// lane group g = tap of instr 0 / 1, the lane's 8 values = 4 channels of the round's two entries), two per round and tile, no
// collected-tap blocks.  `nine`: per round and tile [term 2][lane 64][4 fp16]: lane group 0 / 1 = tap (2,2) of the round's first /
// second entry, groups 2 and 3 zero (K = 16 instruction)
static int pair_p_rounds(const S4Range &r) { return (((r.choff + r.ch + 3) / 4 - r.choff / 4) + 1) / 2; }
size_t pair_p_two_floats(const S4Range &r, int cout) { return (size_t)((cout + 15) / 16) * pair_p_rounds(r) * 2 * (2 * 64 * 4); }
size_t pair_p_nine_floats(const S4Range &r, int cout) { return (size_t)((cout + 15) / 16) * pair_p_rounds(r) * (2 * 64 * 2); }
void pack_conv_weights_pair_p(const float *w, int cin, int cout, const S4Range &r, float *two_f, float *nine_f) {
    unsigned short *two = reinterpret_cast<unsigned short *>(two_f), *nine = reinterpret_cast<unsigned short *>(nine_f);
    const int g0 = r.choff / 4, n_ent = (r.choff + r.ch + 3) / 4 - g0, rounds = (n_ent + 1) / 2, ntiles = (cout + 15) / 16;
    // weight of (cout co, entry ent, channel slot e of the group, tap): zero outside the range / the layer
    auto wv = [&](int co, int ent, int e, int tap) -> float {
        const int ci = 4 * (g0 + ent) + e - r.choff;
        if (co >= cout || ent >= n_ent || ci < 0 || ci >= r.ch) return 0.f;
        return w[((size_t)co * cin + ci) * 9 + tap];
    };
    auto put = [&](unsigned short *&o, float v, int term) {
        const unsigned short hi = split_host_f16(v);
        *o++ = term == 0 ? hi : split_host_f16(v - split_host_f32(hi));
    };
    static const int tap3[2][4] = {{0, 1, 3, 4}, {6, 7, 2, 5}};   // conv_s4.hip
    for (int t = 0; t < ntiles; ++t)
        for (int rd = 0; rd < rounds; ++rd) {
            for (int s = 0; s < 2; ++s)
                for (int term = 0; term < 2; ++term)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) put(two, wv(t * 16 + (lane & 15), rd * 2 + e / 4, e & 3, tap3[s][lane >> 4]), term);
            for (int term = 0; term < 2; ++term)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) put(nine, (lane >> 4) < 2 ? wv(t * 16 + (lane & 15), rd * 2 + (lane >> 4), e, 8) : 0.f, term);
        }
}

// pa.c = the consumer with its sources in the K order [S, other ranges.., P's output] (pack_conv_weights_s4 with pad_sources = 1 in
// that order; P's range declared as {0, p_cout}: the LDS planes start at P's own channel 0), S4 sources and destinations
int launch_conv_pair(const PairArgs &pa, int B, hipStream_t s) {
    const ConvArgs &a = pa.c;
    if (!a.src_fmt || !a.dst_fmt) return fail(PF_EINVAL, "conv_pair: packed-pair sources and destinations only");
    if ((a.Wout & 3) != 0 || a.Hin != a.Hout || a.Win != a.Wout) return fail(PF_EUNSUPPORTED, "conv_pair: stride 1, width % 4 == 0 only");
    if (!conv_pair_supports(a.Cout, pa.p_cout)) return fail(PF_EUNSUPPORTED, "conv_pair: %d / %d output channels", a.Cout, pa.p_cout);
    if (pa.rounds_s < 1 || pa.round_d < pa.rounds_s || pa.round_d >= a.nchunks) return fail(PF_EINVAL, "conv_pair: round layout");
    const int nt = a.ntiles, ntp = pa.p_ntiles;
    if (pa.merged) {
        if (conv_pair_merged_supports(a.Cout, pa.p_cout)) return launch_pair_cfg<2, 1, 1>(pa, B, s);
        return fail(PF_EUNSUPPORTED, "conv_pair (merged): built for C of 17..20 and P of <= 12 output channels");
    }
#define PF_PAIR(NT_, NTP_) \
    if (nt == NT_ && ntp == NTP_) return launch_pair_cfg<NT_, NTP_>(pa, B, s);
    PF_PAIR(1, 1) PF_PAIR(2, 1) PF_PAIR(3, 1) PF_PAIR(2, 2) PF_PAIR(3, 2)
#undef PF_PAIR
    return fail(PF_EUNSUPPORTED, "conv_pair: no kernel for %d + %d cout tiles", nt, ntp);
}

}  // namespace pf
