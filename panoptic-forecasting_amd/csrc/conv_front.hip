// The 512x1024 front end in one pass: base.1 (3x3, stride 1, 16 -> 24) and base.2 (3x3, stride 2, 24 -> 32) of FC-HarDNet
// (hardnet.py:274-283) as ONE kernel - the tensor between them (24 channels at half resolution: 100 MB per 1024x2048 frame,
// written and read back, 39 % of the bytes the three front-end kernels moved) never leaves LDS.  Plan option fuse_front (on).
//
// The stem writes the packed-pair layout of conv_mfma.h (two fp16 terms per value, [B][2][C/4][H][W][4]).  A workgroup (4 waves)
// owns a strip of 31 output columns of one frame and MARCHES DOWN it, two output rows per step:
//   * stem rows live in a ring of three 4-row blocks in LDS; the block of step s + 1 is fetched by LDS-DMA while step s
//     computes (no exposed window latency; every stem row is read once per strip: 66/62 of the tensor);
//   * the first conv computes only the 4 NEW intermediate rows of a step (the fifth is the last row of the step before and stays
//     in LDS): 4 x 63 pixels = 252 = 15.75 M-tiles of 16 LINEARISED pixels (a tile may wrap from one row into the next) = 4 per
//     wave - strips of 31 columns exist for this: 32 would need 4 x 65 = 16.25 tiles.  Dense-tap scheme of conv_s4.hip (two
//     full matrix instructions per 8 channels + the collected ninth tap);
//   * its weights (5 blocks x 2 cout tiles, 80 registers) stay in registers for the whole strip - two waves per SIMD, 216
//     registers; the second conv's (7 blocks, one cout tile per wave) are loaded from L2 between the M-tiles of the split
//     phase and have arrived by the time the barrier behind it is passed;
//   * bias, ReLU, zero outside the image (= the second conv's padding), round-to-nearest split, and into LDS with odd and even
//     columns in separate halves of a row, so that a lane's stride-2 pixel (2 ox + kx) is a stride-1 slot for the second conv;
//   * a step's results are stored at the START of the next step, so that the wait for the next window block (vmcnt counts
//     stores on gfx950) never waits for stores that were just issued.
// A strip is cut into vertical segments to fill the chip; a segment starts with one masked step (peeled) that only produces
// the intermediate row above its first output row.  Arithmetic per layer is that of conv_s4.hip: three fp16 products per fp32
// multiply, fp32 accumulation, round-to-nearest split (operand bound 2^-23), weights pre-scaled by a power of two, range guard.
//
// History (profiles/r03_experiments.md): the first form computed one 2 x 32 output tile per workgroup (5 waves, 7 x 68 window,
// weights from L2 or LDS): correct, 1.00 / 1.19 ms per 16 frames against 0.46 + 0.36 ms for conv_split + conv_dma - every tile
// one dependent chain at 2.5 waves per SIMD (tools/probe_front.py: 22 000 clocks per tile, the matrix pipe busy for 7 000).
// This form: 0.42 ms.
#include <algorithm>

#include "conv_epilogue.h"
#include "pf_prof.h"
#include <type_traits>

namespace pf {

typedef float fm_f32x4 __attribute__((ext_vector_type(4)));
typedef split_x8 fm_h8;
typedef split_x4 fm_h4;
typedef __attribute__((address_space(3))) void *fm_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kFmOob = 0x80000000u;

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // shader-clock stamps of one workgroup in the middle of the grid, steps 2.. (tools/probe_front.py): slot 16 wave + i
#define FM_PROBE(i) do { if (lane == 0 && step == 5 && blockIdx.x == gridDim.x / 2 && blockIdx.z == gridDim.z / 2 && a.probe) a.probe[16 * wave + (i)] = clock64(); } while (0)
#else
#define FM_PROBE(i) do { } while (0)
#endif

struct FrontCfg {
    static constexpr int TW2 = 31;                      // output columns of a strip
    static constexpr int YC = 2 * TW2 + 1;              // intermediate columns (63)
    static constexpr int NPX1 = 4 * YC;                 // new intermediate pixels per step (252)
    static constexpr int NPX2 = 2 * TW2;                // outputs per step (62)
    static constexpr int XC = YC + 3;                   // stem window columns (66, starts at an even column)
    static constexpr int XROWP = XC / 2;                // 16-B pieces (2 pixels x 4 channels of one term) per row
    static constexpr int XROW = XROWP * 16;             // bytes per row of one plane (528)
    static constexpr int XBPLANE = 4 * XROW;            // one (term, group) plane of a 4-row block (2112)
    static constexpr int XBLK = 8 * XBPLANE;            // [term][group] planes of a block (16896)
    static constexpr int XPIECES = XBLK / 16;           // 1056
    static constexpr int XRING = 3 * XBLK;
    static constexpr int YSA = TW2 + 1, YSB = TW2;      // slots of the even-column / odd-column half of an intermediate row
    static constexpr int YROW = (YSA + YSB) * 8;        // 504
    static constexpr int YPLANE = 5 * YROW;             // rows 0..2 of the step, carried row of even / odd steps
    static constexpr int YBYTES = 12 * YPLANE;          // [term][6 groups]
    static constexpr int X_OFF = YBYTES;                // LDS: [intermediate | window ring | bias]: the intermediate planes at offset 0 keep every
    static constexpr int BIAS_OFF = XRING + YBYTES;     // ds_write / ds_read offset of the split phase and the second conv inside the 16-bit immediate
    static constexpr int LDS_BYTES = BIAS_OFF + 256;
    static constexpr int NB1 = 5, NB2 = 7;              // weight blocks per cout tile (conv_s4.hip: s4_blocks_total(2), (3))
    static constexpr int NIT = (XPIECES + 255) / 256;   // DMA instructions per thread and block
};

__device__ __forceinline__ fm_h8 fm_join(fm_h4 lo, fm_h4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }


__global__ __launch_bounds__(256, 2) void conv_front_kernel(FrontArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = FrontCfg;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem + C::X_OFF, *ys = smem;
    static_assert(C::X_OFF % 16 == 0, "DMA destinations are 16-B aligned");
    float *bias_lds = reinterpret_cast<float *>(smem + C::BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, li = lane & 15;
    const int strip = (int)blockIdx.x % a.tilesX, seg = (int)blockIdx.x / a.tilesX, b = blockIdx.z;
    const int ox0 = strip * C::TW2, oyA = seg * a.seg_steps * 2;
    const int nsteps = min(a.seg_steps, (a.H2 - oyA + 1) / 2);   // steps of this segment (2 output rows each)
    const int yc0 = 2 * ox0 - 1, xc0 = yc0 - 1;                  // first intermediate / stem column of the strip (xc0 even)

    // ---- window blocks: block k (k = 0 is the block of the masked first step) = stem rows 2 oyA - 3 + 4k .. + 3
    const size_t plane_bytes = (size_t)a.H1 * a.W1 * 8;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(reinterpret_cast<const char *>(a.x) + (size_t)b * 8 * plane_bytes), 0, 0x7FFFFFFF, 0x00020000);
    // this thread's pieces of a block (piece p = it * 256 + tid -> plane, row, piece of the row): byte offset in row 0 of the image
    // (a multiple of 16) | row, or out of range when the column is outside the image
    unsigned dma_pk[C::NIT];
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int p = it * 256 + tid, pl = p / (4 * C::XROWP), rem = p - pl * (4 * C::XROWP);
        const int row = rem / C::XROWP, cp = rem - row * C::XROWP, gx = xc0 + 2 * cp;
        dma_pk[it] = (p < C::XPIECES && gx >= 0 && gx < a.W1) ? ((unsigned)(pl * plane_bytes) + (unsigned)gx * 8u) | (unsigned)row : kFmOob;
    }
    const unsigned row_bytes = (unsigned)a.W1 * 8u;
    // pass `it` of block k (a DMA instruction per wave)
    auto issue_part = [&](int k, int slot, int it) {
        const int r0 = 2 * oyA - 3 + 4 * k;
        const bool whole = (it + 1) * 256 <= C::XPIECES;            // every thread has a piece in this pass
        if (!whole && it * 256 + wave * 64 >= C::XPIECES) return;   // uniform
        const int gy = r0 + (int)(dma_pk[it] & 3u);
        const bool ok = (dma_pk[it] != kFmOob) & ((unsigned)gy < (unsigned)a.H1);
        const unsigned off = ok ? (dma_pk[it] & ~3u) + (unsigned)gy * row_bytes : kFmOob;
        if (whole || it * 256 + tid < C::XPIECES)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (fm_lds_ptr_t)(xs + slot * C::XBLK + (it * 256 + wave * 64) * 16), 16, off, 0, 0, 0);
    };
    auto issue_block = [&](int k, int slot) {
#pragma unroll
        for (int it = 0; it < C::NIT; ++it) issue_part(k, slot, it);
    };
    issue_block(0, 0);
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t b1 = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias1, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t b2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias2, 0, 0x7FFFFFFF, 0x00020000);
        if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(b1, (fm_lds_ptr_t)bias_lds, 4, (unsigned)lane * 4u, 0, 0, 0);
        if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(b2, (fm_lds_ptr_t)(bias_lds + 32), 4, (unsigned)lane * 4u, 0, 0, 0);
    }

    // ---- first conv's weights: resident.  [tile][block][term][lane][8 fp16]
    fm_h8 w1h[C::NB1][2], w1m[C::NB1][2];
    {
        const char *wp = reinterpret_cast<const char *>(a.w1) + lane * 16;
#pragma unroll
        for (int k = 0; k < C::NB1; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                w1h[k][n] = *reinterpret_cast<const fm_h8 *>(wp + ((size_t)(n * C::NB1 + k) * 2 + 0) * 1024);
                w1m[k][n] = *reinterpret_cast<const fm_h8 *>(wp + ((size_t)(n * C::NB1 + k) * 2 + 1) * 1024);
            }
    }
    const int n2 = wave & 1;   // the second conv: this wave's cout tile, M-tiles 2 (wave >> 1), + 1
    // (buffer loads: uniform resource + the lane's 32-bit offset + a scalar block offset - with plain pointers hipcc keeps one
    // 64-bit per-lane address per block alive across the loop: 22 registers)
    const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(reinterpret_cast<const char *>(a.w2) + (size_t)n2 * C::NB2 * 2048), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned w2lane = (unsigned)lane * 16u;

    // ---- per-lane geometry, PACKED: the loop derives everything it needs from these few registers in every step (a few dozen
    //      vector instructions) behind an opaque copy - left to itself the compiler hoists every derived offset and mask out of
    //      the loop and the kernel no longer fits 256 registers beside its resident weights.
    //      first conv: M-tile m of this wave = linearised pixels 64 wave + 16 m + li of the 4 x 63 new rows: q | column << 2 | valid << 8 | ... << 9
    int qc_pack[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int p = 64 * wave + 16 * m + li, pv = min(p, C::NPX1 - 1), q = pv / C::YC;
        const int c = pv - q * C::YC;
        qc_pack[m] = q | (c << 2) | ((p < C::NPX1) << 8) | ((((c & 1) ? C::YSA * 8 : 0) + (c >> 1) * 8) << 9);   // + byte offset of the column in an intermediate row
    }
    //      second conv: pixels 16 mt + li of the 2 x 31 outputs of a step: row | column << 1 | valid << 6
    int rc_pack[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = 16 * (2 * (wave >> 1) + j) + li, pv = min(p, C::NPX2 - 1), ry = pv / C::TW2;
        rc_pack[j] = ry | ((pv - ry * C::TW2) << 1) | ((p < C::NPX2) << 6);
    }

    // results of the previous step, stored at the start of the next one
    fm_f32x4 pend[2];
    int pend_oy = -1;
    pend[0] = pend[1] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
    // stores go through a buffer resource of the frame with 32-bit offsets computed in the step (no 64-bit per-lane pointers
    // kept alive across the loop)
    const unsigned hw2 = (unsigned)(a.H2 * a.W2), term2 = (unsigned)a.dst_c4 * hw2 * 8u;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(reinterpret_cast<char *>(a.dst) + (size_t)b * (a.dst_fmt ? 2 * (size_t)term2 : (size_t)a.dst_ctotal * hw2 * 4)), 0, 0x7FFFFFFF, 0x00020000);
    typedef unsigned fm_u2 __attribute__((ext_vector_type(2)));
    // every lane of this wave stores whole 4-channel units of a packed tensor (uniform): no per-lane channel tests
    const bool whole_units = a.dst_fmt && (a.dst_choff & 3) == 0 && a.dst_choff + n2 * 16 + 16 <= a.dst_limit && n2 * 16 + 16 <= a.C2 + 2;
    auto store_pending = [&](const int (&rc)[2], int gq, int j0, int j1) {   // M-tiles j0 .. j1 - 1 of the previous step
        if (pend_oy < 0) return;   // uniform
        const int co = n2 * 16 + 4 * gq;
        if (whole_units) {
#pragma unroll
            for (int j = j0; j < j1; ++j) {
                const int oy = pend_oy + (rc[j] & 1), ox = ox0 + ((rc[j] >> 1) & 31);
                fm_h4 hi, mid;
                split_terms4(pend[j], hi, mid);
                const unsigned off = (unsigned)(oy * a.W2 + ox) * 8u + (unsigned)((a.dst_choff + co) >> 2) * hw2 * 8u;
                if (((rc[j] >> 6) != 0) & (oy < a.H2) & (ox < a.W2)) {
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fm_u2, hi), drs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fm_u2, mid), drs, off, term2, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const int oy = pend_oy + (rc[j] & 1), ox = ox0 + ((rc[j] >> 1) & 31);
            if (!(rc[j] >> 6) || oy >= a.H2 || ox >= a.W2 || co >= a.C2 + 2) continue;
            const unsigned pix = (unsigned)(oy * a.W2 + ox);
            const fm_f32x4 v = pend[j];
            if (a.dst_fmt) {
                fm_h4 hi, mid;
                split_terms4(v, hi, mid);
                const fm_u2 hu = __builtin_bit_cast(fm_u2, hi), mu = __builtin_bit_cast(fm_u2, mid);
                const int chb = a.dst_choff + co;
                const bool k0 = chb < a.dst_limit, k1 = chb + 2 < a.dst_limit;
                const unsigned off = pix * 8u + (unsigned)(chb >> 2) * hw2 * 8u;
                if (k1) {            // (the plan only fuses a destination range that starts on a channel group: dst_choff % 4 == 0)
                    __builtin_amdgcn_raw_buffer_store_b64(hu, drs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(mu, drs, off, term2, 0);
                } else if (k0) {
                    __builtin_amdgcn_raw_buffer_store_b32(hu[0], drs, off, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(mu[0], drs, off, term2, 0);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float vr = v[r];
                    if (co + r < a.C2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vr), drs, ((unsigned)(a.dst_choff + co + r) * hw2 + pix) * 4u, 0, 0);
                }
            }
        }
    };

    // one wait the compiler's wait-count pass can see, for the resident weights: the loop's own waits are inline asm, and a load the
    // pass believes pending would be protected by s_waitcnt vmcnt(n) in front of its first use in EVERY step - a wait that, in
    // the loop, falls on the window block just requested
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float vmax = 0.f, vmax_mid = 0.f;       // range guard (conv_mfma.h): what is stored / what the second conv reads from LDS
    int cur = 0, prev = 2, next = 1;        // ring slots of the window blocks of this step, the one before, the one after
    // step -1 = the masked first step (intermediate row 2 oyA - 1 only); step s >= 0 = output rows oyA + 2 s, + 1
    auto run_step = [&](auto first_c, const int step) {
        constexpr bool FIRST = decltype(first_c)::value;   // the masked first step of the segment (peeled: the steady-state body carries none of its tests)
        FM_PROBE(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the block (and, first time, the bias values) have landed
        __syncthreads();                                    // everyone's have; everyone is done with the previous step's LDS reads
        FM_PROBE(1);
        if (step + 1 < nsteps) issue_block(step + 2, next);   // as early as possible: a block takes ~4 000 clocks to land under load
        FM_PROBE(2);
        int gq = g, qc[4], rc[2];
        asm volatile("" : "+v"(gq));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            qc[m] = qc_pack[m];
            asm volatile("" : "+v"(qc[m]));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            rc[j] = rc_pack[j];
            asm volatile("" : "+v"(rc[j]));
        }
        FM_PROBE(3);
        // taps of lane group g in the two full instructions and the collected one (conv_s4.hip)
        const int kyt[3] = {gq >> 1, gq < 2 ? 2 : gq - 2, 2}, kxt[3] = {gq & 1, gq < 2 ? gq : 2, 2};
        const int sp = step & 1;                            // slot 3 + sp holds this step's last row, 3 + (sp ^ 1) the carried one
        const int yr0 = 2 * oyA + 4 * step;                 // first new intermediate row

        // ---- first conv: 4 M-tiles x 2 cout tiles per wave
        fm_f32x4 acc[4][2];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][0] = acc[m][1] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
        // byte offset of the lane's fragment of (M-tile, tap): stem row d = q - 1 + ky relative to the step's first new row is in the
        // previous block (d < 1) or in this one
        int xoff[4][3];
        const int xprev = __builtin_amdgcn_readfirstlane(prev * C::XBLK + 3 * C::XROW), xcur = __builtin_amdgcn_readfirstlane(cur * C::XBLK - C::XROW);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int d = (qc[m] & 3) - 1 + kyt[t];
                xoff[m][t] = (d < 1 ? xprev : xcur) + __mul24(d, C::XROW) + (((qc[m] >> 2) & 63) + kxt[t]) * 8 +
                             (t == 2 ? (gq & 1) * 2 * C::XBPLANE : 0);
            }
        auto xfrag = [&](int m, int t, int rd, fm_h8 &h, fm_h8 &md) {
            const unsigned char *p = xs + xoff[m][t] + rd * 2 * C::XBPLANE;
            h = fm_join(*reinterpret_cast<const fm_h4 *>(p), *reinterpret_cast<const fm_h4 *>(p + C::XBPLANE));
            md = fm_join(*reinterpret_cast<const fm_h4 *>(p + 4 * C::XBPLANE), *reinterpret_cast<const fm_h4 *>(p + 5 * C::XBPLANE));
        };
        // 10 units of (weight block, M-tile pair); the fragments of unit u + 1 are read while the matrix instructions of unit u
        // issue (two waves per SIMD do not hide an LDS round trip per unit: tools/probe_front.py)
        auto mfmas1 = [&](int k, int m0, const fm_h8 (&fh)[2], const fm_h8 (&fmd)[2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m0 + m][n] = PF_MFMA_SPLIT(w1h[k][n], fmd[m], acc[m0 + m][n]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m0 + m][n] = PF_MFMA_SPLIT(w1m[k][n], fh[m], acc[m0 + m][n]);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m0 + m][n] = PF_MFMA_SPLIT(w1h[k][n], fh[m], acc[m0 + m][n]);
        };
        // ---- bias, ReLU, zero outside the image (= the second conv's padding), split, -> LDS by column parity: one (M-tile, cout tile)
        const float lo1 = a.relu1 ? 0.f : -__builtin_inff();
        fm_f32x4 b1v[2];
        b1v[0] = *reinterpret_cast<const fm_f32x4 *>(bias_lds + 4 * gq);
        b1v[1] = *reinterpret_cast<const fm_f32x4 *>(bias_lds + 16 + 4 * gq);
        auto split_unit = [&](int m, int n) {
            const int q = qc[m] & 3, c = (qc[m] >> 2) & 63;
            const bool live = ((qc[m] >> 8) & 1) && !(FIRST && q != 3);
            const int slot = q == 3 ? 3 + sp : q;
            const int gy = yr0 + q, gx = yc0 + c;
            // outside the image the value is the second conv's zero padding: median(v, lo, hi) with lo = hi = 0 there, (ReLU floor, +inf) inside
            const bool in = ((unsigned)gy < (unsigned)a.H1) & ((unsigned)gx < (unsigned)a.W1);
            const float lom = in ? lo1 : 0.f, him = in ? __builtin_inff() : 0.f;
            unsigned char *row = ys + __mul24(slot, C::YROW) + (qc[m] >> 9) + __mul24(gq, C::YPLANE);
            fm_f32x4 v = acc[m][n];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(__builtin_fmaf(v[r], a.scale1, b1v[n][r]), lom, him);
            if (n * 4 + gq >= 6 || !live) return;
            vmax_mid = range_acc(vmax_mid, v[0], v[1], v[2], v[3]);
            fm_h4 hi, mid;
            split_terms4(v, hi, mid);
            *reinterpret_cast<fm_h4 *>(row + n * 4 * C::YPLANE) = hi;
            *reinterpret_cast<fm_h4 *>(row + (6 + n * 4) * C::YPLANE) = mid;
        };
        // the second conv's weights for this step: loaded from L2 between the last split units (a block costs 8 registers and every
        // finished unit frees 4 accumulators: all seven blocks at once do not fit beside the resident first-conv weights), used
        // after the next barrier
        fm_h8 w2h[C::NB2], w2m[C::NB2];
        auto load_w2 = [&](int k) {
            if (FIRST) return;
            w2h[k] = __builtin_bit_cast(fm_h8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, w2lane, k * 2048, 0));
            w2m[k] = __builtin_bit_cast(fm_h8, __builtin_amdgcn_raw_buffer_load_b128(w2rs, w2lane, k * 2048 + 1024, 0));
        };
        FM_PROBE(4);
        {
            // 10 units: M-tile pair 0 through its 5 weight blocks, then pair 1 - and while pair 1's matrix instructions issue, the
            // finished accumulators of pair 0 go through bias / split / LDS in their shadow (the split phase of all four M-tiles
            // behind the last matrix instruction was 2 400 of a step's 10 400 clocks)
            fm_h8 fh[2][2], fmd[2][2];
            auto load_unit = [&](int u, int buf) {
                const int k = u % C::NB1, m0 = (u / C::NB1) * 2;
#pragma unroll
                for (int m = 0; m < 2; ++m) xfrag(m0 + m, k < 4 ? (k & 1) : 2, k < 4 ? (k >> 1) : 0, fh[buf][m], fmd[buf][m]);
            };
            load_unit(0, 0);
#pragma unroll
            for (int u = 0; u < 2 * C::NB1; ++u) {
                if (u + 1 < 2 * C::NB1) load_unit(u + 1, (u + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);   // the reads of the next unit go out BEFORE this unit's matrix instructions (hipcc sinks them to the last two otherwise)
                mfmas1(u % C::NB1, (u / C::NB1) * 2, fh[u & 1], fmd[u & 1]);
                // the previous step's results go out between the units: right behind the DMA instructions above, the stores of all eight
                // waves of a CU queue up in front of the one address unit (1 700 clocks per wave with no matrix instruction issued)
                if (u == 1) store_pending(rc, gq, 0, 1);
                if (u == 3) store_pending(rc, gq, 1, 2);
                if (u >= 5 && u <= 8) split_unit((u - 5) >> 1, (u - 5) & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            pend_oy = -1;
        }
        FM_PROBE(5);
        load_w2(0);
        load_w2(1);
        split_unit(2, 0);
        split_unit(2, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_w2(2);
        load_w2(3);
        load_w2(4);
        __builtin_amdgcn_sched_barrier(0);
        split_unit(3, 0);
        split_unit(3, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_w2(5);
        load_w2(6);
        FM_PROBE(6);
        __syncthreads();
        FM_PROBE(7);
        if (FIRST) {
            prev = cur; cur = next; next = next == 2 ? 0 : next + 1;
            return;
        }

        // ---- second conv (stride 2): 2 M-tiles of this wave's cout tile; one accumulator per product
        fm_f32x4 acc2[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[j][0] = acc2[j][1] = acc2[j][2] = fm_f32x4{0.f, 0.f, 0.f, 0.f};
        // intermediate row e = 2 ry + ky - 1 relative to the step's first new row: -1 = the carried row, 3 = this step's last row
        int yoff[2][3];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int e = 2 * (rc[j] & 1) + kyt[t] - 1;
                yoff[j][t] = __mul24(e < 0 ? 3 + (sp ^ 1) : e == 3 ? 3 + sp : e, C::YROW) + ((kxt[t] & 1) ? C::YSA * 8 : 0) +
                             (((rc[j] >> 1) & 31) + (kxt[t] >> 1)) * 8 + (t == 2 ? min(gq, 2) * 2 * C::YPLANE : 0);
            }
        auto yfrag = [&](int j, int t, int rd, fm_h8 &h, fm_h8 &md) {
            const unsigned char *p = ys + yoff[j][t] + rd * 2 * C::YPLANE;
            h = fm_join(*reinterpret_cast<const fm_h4 *>(p), *reinterpret_cast<const fm_h4 *>(p + C::YPLANE));
            md = fm_join(*reinterpret_cast<const fm_h4 *>(p + 6 * C::YPLANE), *reinterpret_cast<const fm_h4 *>(p + 7 * C::YPLANE));
        };
        {
            fm_h8 fh[2][2], fmd[2][2];
            auto load_blk = [&](int k, int buf) {
#pragma unroll
                for (int j = 0; j < 2; ++j) yfrag(j, k < 6 ? (k & 1) : 2, k < 6 ? (k >> 1) : 0, fh[buf][j], fmd[buf][j]);
            };
            load_blk(0, 0);
#pragma unroll
            for (int k = 0; k < C::NB2; ++k) {
                if (k + 1 < C::NB2) load_blk(k + 1, (k + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j][0] = PF_MFMA_SPLIT(w2h[k], fmd[k & 1][j], acc2[j][0]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j][1] = PF_MFMA_SPLIT(w2m[k], fh[k & 1][j], acc2[j][1]);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j][2] = PF_MFMA_SPLIT(w2h[k], fh[k & 1][j], acc2[j][2]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FM_PROBE(8);
        {
            const float lo2 = a.relu2 ? 0.f : -__builtin_inff();
            const fm_f32x4 b4 = *reinterpret_cast<const fm_f32x4 *>(bias_lds + 32 + n2 * 16 + 4 * gq);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                fm_f32x4 v = (acc2[j][0] + acc2[j][1]) + acc2[j][2];
                v = __builtin_elementwise_max(__builtin_elementwise_fma(v, fm_f32x4{a.scale2, a.scale2, a.scale2, a.scale2}, b4), fm_f32x4{lo2, lo2, lo2, lo2});
                if (rc[j] >> 6) vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
                pend[j] = v;
            }
            pend_oy = oyA + 2 * step;
        }
        FM_PROBE(9);
        prev = cur; cur = next; next = next == 2 ? 0 : next + 1;
    };
    run_step(std::true_type{}, -1);
    for (int step = 0; step < nsteps; ++step) run_step(std::false_type{}, step);
    store_pending(rc_pack, g, 0, 2);
    if (a.status) {
        range_commit(a.status, a.range_slot_mid, vmax_mid);
        range_commit(a.status, a.range_slot, vmax);
    }
#endif
}

// shapes this kernel is built for: 16 -> 24 -> (16, 32] channels, 3x3 stride 1 then 3x3 stride 2, width % 4 == 0 (16-B pieces), 32-bit
// offsets: the input pieces (h1 * w1 < 2^24) and the stores, whose buffer offsets - (dst_choff + co) * H2 * W2 * 4 in fp32 NCHW,
// 2 * dst_c4 * H2 * W2 * 8 with the second term's plane offset in the packed layout - must stay below the 2^31 records of the
// buffer descriptor for EVERY channel of the destination tensor (a wide concatenation at a large resolution would not)
bool conv_front_supports(int c0, int c1, int c2, int h1, int w1, int dst_ctotal) {
    const long long h2 = (h1 - 1) / 2 + 1, w2 = (w1 - 1) / 2 + 1;
    const long long dst_c4 = (dst_ctotal + 3) / 4;
    const long long max_off = std::max((long long)dst_ctotal * h2 * w2 * 4, 2 * dst_c4 * h2 * w2 * 8);
    return c0 == 16 && c1 == 24 && c2 <= 32 && c2 > 16 && (w1 & 3) == 0 && h1 >= 2 && (long long)h1 * w1 < (1ll << 24) &&
           max_off < 0x7FFFFFFFll;
}

// strips of 31 output columns x segments of `seg_steps` steps (2 output rows each) x frames.  The segment count is the one that
// fills the chip's workgroup slots (2 per CU) best, counting the masked first step of every segment as lost work
int launch_conv_front(const FrontArgs &a0, int B, hipStream_t s) {
    using C = FrontCfg;
    static_assert(C::LDS_BYTES <= 81920, "two workgroups per CU");
    FrontArgs a = a0;
    a.tilesX = (a.W2 + C::TW2 - 1) / C::TW2;
    const int steps = (a.H2 + 1) / 2;
    static int slots = 0;
    if (!slots) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        slots = 2 * cus;
    }
    int best = 1;
    double best_eff = 0.0;
    for (int nseg = 1; nseg <= steps && nseg <= 64; ++nseg) {
        const int seg_steps = (steps + nseg - 1) / nseg, used = (steps + seg_steps - 1) / seg_steps;
        if (used != nseg) continue;
        const long items = (long)a.tilesX * nseg * B;
        const long rounds = (items + slots - 1) / slots;
        const double eff = (double)items / (double)(rounds * slots) * steps / (double)(nseg * (seg_steps + 1));
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = nseg;
        }
    }
    a.seg_steps = (steps + best - 1) / best;
    a.tilesY = (steps + a.seg_steps - 1) / a.seg_steps;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_front_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const double px1 = (double)B * a.H1 * a.W1, px2 = (double)B * a.H2 * a.W2;
    ProfScope ps(s, "pf::conv_front_kernel(pf::FrontArgs)", 2.0 * 9 * (px1 * 16 * a.C1 + px2 * a.C1 * a.C2),
                 4.0 * (px1 * 16 + px2 * a.C2 + 9.0 * (16 * a.C1 + a.C1 * a.C2)));
    hipLaunchKernelGGL(conv_front_kernel, dim3(a.tilesX * a.tilesY, 1, B), dim3(256), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_front_kernel");
    return PF_OK;
}

}  // namespace pf
