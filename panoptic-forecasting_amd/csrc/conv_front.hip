// The 512x1024 front end in one pass: base.1 (3x3, stride 1) and base.2 (3x3, stride 2) of FC-HarDNet (hardnet.py:274-283)
// as ONE kernel - the tensor between them (24 channels at half resolution: 100 MB per 1024x2048 frame, written and read back,
// 39 % of the bytes the three front-end kernels moved) never leaves LDS.
//
// Round 2 ran stem -> conv_split<2, 64> (438 us per 16 frames, 3.1 TB/s of algorithmic bytes: bound by the fabric) ->
// conv_dma stride 2 (332 us, fp32 MFMA): 1.09 ms of a 7.2 ms step.  Here the stem writes the packed-pair layout of
// conv_mfma.h (two fp16 terms per value, [B][2][C/4][H][W][4]) and a workgroup of 5 waves produces a 2 x 32-pixel tile of
// the stride-2 conv's output:
//   1. the 7 x 68-pixel window of the stem output it needs arrives by LDS-DMA (8 planes = 2 terms x 4 channel groups; out-of-
//      image pieces land as zeros = the first conv's padding);
//   2. wave r computes row r of the 5 x 65 intermediate region (5 M-tiles of 16 pixels x 2 cout tiles) with the dense-tap
//      scheme of conv_s4.hip (two full matrix instructions per 8 channels + the collected ninth tap), weights straight from
//      L2 into registers one block ahead (no LDS space);
//   3. bias, ReLU, zero outside the image (the second conv's padding), split into the two fp16 terms, and into LDS - odd and
//      even columns in separate planes, so that a lane's stride-2 pixel (2 ox + kx) is a stride-1 slot for the second conv;
//   4. waves 0..3 compute one 16-pixel M-tile x 2 cout tiles of the stride-2 conv from those planes;
//   5. bias, ReLU, range guard, packed-pair (or fp32 NCHW) store.
// The intermediate region is recomputed at tile borders (5 rows for 4, 65 columns for 64: 1.27x the first conv's flops).
// Arithmetic per layer is that of conv_s4.hip: three fp16 products per fp32 multiply, fp32 accumulation, round-to-nearest
// split (operand bound 2^-23), weights pre-scaled by a power of two.
#include "conv_epilogue.h"
#include "pf_prof.h"

namespace pf {

typedef float fr_f32x4 __attribute__((ext_vector_type(4)));
typedef split_x8 fr_h8;
typedef split_x4 fr_h4;
typedef __attribute__((address_space(3))) void *fr_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kFrOob = 0x80000000u;

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE   // shader-clock stamps of every wave of one workgroup in the middle of the grid (tools/probe_front.py): slot 8 wave + i
#define FR_PROBE(i) do { if (lane == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.z == gridDim.z / 2 && a.probe) a.probe[8 * wave + (i)] = clock64(); } while (0)
#else
#define FR_PROBE(i) do { } while (0)
#endif

template <int TW2_, bool WLDS_>
struct FrontCfg {
    static constexpr int TH2 = 2, TW2 = TW2_;          // output tile of the stride-2 conv
    static constexpr bool WLDS = WLDS_;                // both weight sets resident in LDS (DMA at workgroup start) instead of read from L2 per block
    static constexpr int YR = 2 * TH2 + 1;             // rows of the intermediate region (5)
    static constexpr int YC = 2 * TW2 + 1;             // ... and columns (65 / 33)
    static constexpr int MT1 = (YC + 15) / 16;         // M-tiles per intermediate row (the last holds one column)
    static constexpr int XR = YR + 2, XC = YC + 3;     // stem window: 7 rows x 68 / 36 pixels (starts at an even column)
    static constexpr int XPIECES = XR * (XC / 2);      // 16-B pieces (2 pixels x 4 channels of one term) per plane
    static constexpr int XPLANE = XPIECES * 16;        // bytes
    static constexpr int C0G = 4, C1G = 6;             // channel groups of the stem output (16) and of the intermediate (24)
    static constexpr int XBYTES = 2 * C0G * XPLANE;    // [term][group] planes
    static constexpr int YSA = TW2 + 1, YSB = TW2;     // slots per row of the odd-column (local j even) / even-column plane
    static constexpr int YPA = YR * YSA * 8, YPAIR = YR * (YSA + YSB) * 8;   // bytes: first plane, both planes of a (term, group)
    static constexpr int YBYTES = 2 * C1G * YPAIR;
    static constexpr int NB1 = 5, NB2 = 7;             // weight blocks per cout tile: s4_blocks_total(2 rounds), (3 rounds)
    static constexpr int WBLK = 2 * 64 * 16;           // one block of one tile: [term][lane][8 fp16]
    static constexpr int W1BYTES = WLDS ? 2 * NB1 * WBLK : 0, W2BYTES = WLDS ? 2 * NB2 * WBLK : 0;
    static constexpr int W1_OFF = XBYTES + YBYTES, W2_OFF = W1_OFF + W1BYTES;
    static constexpr int BIAS_OFF = W2_OFF + W2BYTES;  // 2 x 32 floats
    static constexpr int LDS_BYTES = BIAS_OFF + 256;
    static constexpr int NTHR = 64 * YR;               // one wave per intermediate row
    static constexpr int XP = 2 * C0G * XPIECES, W1P = W1BYTES / 16, W2P = W2BYTES / 16;   // 16-B pieces
    static constexpr int NDMA = (XP + W1P + W2P + NTHR - 1) / NTHR;
    static constexpr int NN2 = TW2 / 16;               // cout tiles per wave in the second conv (waves 0..3 = M-tile x NN2 tiles)
};

__device__ __forceinline__ fr_h8 fr_join(fr_h4 lo, fr_h4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// one block of packed weights for both cout tiles: [tile][block][term][lane][8 fp16]; from LDS or straight from L2
struct FrW {
    fr_h8 h[2], m[2];
};
template <typename P>
__device__ __forceinline__ FrW fr_load_w(P w, int nblocks, int blk, int lane) {
    FrW r;
    P p = w + ((size_t)blk * 2 * 64 + lane) * 16;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        r.h[n] = *reinterpret_cast<const fr_h8 *>(p + (size_t)n * nblocks * 2 * 64 * 16);
        r.m[n] = *reinterpret_cast<const fr_h8 *>(p + (size_t)n * nblocks * 2 * 64 * 16 + 64 * 16);
    }
    return r;
}

template <int TW2_, bool WLDS_>
__global__ __launch_bounds__(320, 3) void conv_front_kernel(FrontArgs a) {   // 2 workgroups of 5 waves per CU: up to 3 waves on a SIMD
#if defined(__HIP_DEVICE_COMPILE__)
    using C = FrontCfg<TW2_, WLDS_>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *xs = smem, *ys = smem + C::XBYTES;
    float *bias_lds = reinterpret_cast<float *>(smem + C::BIAS_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_lin, dummy;
    xcd_tile_order(a.tilesX * a.tilesY, tile_lin, dummy);
    const int tileY = tile_lin / a.tilesX, tileX = tile_lin - tileY * a.tilesX, b = blockIdx.z;
    const int oy0 = tileY * C::TH2, ox0 = tileX * C::TW2;
    FR_PROBE(0);
    const int yr0 = 2 * oy0 - 1, yc0 = 2 * ox0 - 1;      // intermediate region origin (stride-1 coordinates)
    const int xr0 = yr0 - 1, xc0 = yc0 - 1;              // stem window origin; xc0 = 2 ox0 - 2 is even

    // ---- 1. stem window (and, WLDS, both weight sets) -> LDS in one DMA burst; bias values of both convs.  The LDS regions
    //      [window | intermediate | weights 1 | weights 2] are one piece-indexed target: piece p goes to byte 16 p past its region's base
    {
        const size_t plane_bytes = (size_t)a.H1 * a.W1 * 8;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(reinterpret_cast<const char *>(a.x) + (size_t)b * 2 * C::C0G * plane_bytes), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t w1rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w1, 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t w2rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, 0x7FFFFFFF, 0x00020000);
        constexpr int XPAD = (C::XP + 63) / 64 * 64, W1PAD = (C::W1P + 63) / 64 * 64;   // every DMA instruction of a wave stays inside one region
        constexpr int TOTAL = XPAD + W1PAD + C::W2P, NIT = (TOTAL + C::NTHR - 1) / C::NTHR;
        static_assert(C::W1P % 64 == 0 && C::W2P % 64 == 0, "weight regions are whole DMA instructions");
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int p0 = it * C::NTHR + wave * 64, p = p0 + lane;     // p0 uniform
            if (p0 >= TOTAL) break;
            if (p0 < XPAD) {
                const int pl = p / C::XPIECES, q = p - pl * C::XPIECES;
                const int row = q / (C::XC / 2), cp = q - row * (C::XC / 2);
                const int gy = xr0 + row, gx = xc0 + 2 * cp;
                const bool ok = p < C::XP && gy >= 0 && gy < a.H1 && gx >= 0 && gx < a.W1;
                const unsigned off = ok ? (unsigned)(pl * plane_bytes) + (unsigned)(gy * a.W1 + gx) * 8u : kFrOob;
                if (p < C::XP) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (fr_lds_ptr_t)(xs + p0 * 16), 16, off, 0, 0, 0);
            } else if (p0 < XPAD + W1PAD) {
                const int q0 = p0 - XPAD;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rs, (fr_lds_ptr_t)(smem + C::W1_OFF + q0 * 16), 16, (unsigned)(q0 + lane) * 16u, 0, 0, 0);
            } else {
                const int q0 = p0 - XPAD - W1PAD;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w2rs, (fr_lds_ptr_t)(smem + C::W2_OFF + q0 * 16), 16, (unsigned)(q0 + lane) * 16u, 0, 0, 0);
            }
        }
        if (wave == 0) {
            const __amdgpu_buffer_rsrc_t b1 = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias1, 0, 0x7FFFFFFF, 0x00020000);
            const __amdgpu_buffer_rsrc_t b2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.bias2, 0, 0x7FFFFFFF, 0x00020000);
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(b1, (fr_lds_ptr_t)bias_lds, 4, (unsigned)lane * 4u, 0, 0, 0);
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(b2, (fr_lds_ptr_t)(bias_lds + 32), 4, (unsigned)lane * 4u, 0, 0, 0);
        }
    }
    const int g = lane >> 4, li = lane & 15;
    // tap of lane group g in the two full instructions / the collected tap (conv_s4.hip): (ky, kx)
    const int ky0 = g >> 1, kx0 = g & 1, ky1 = g < 2 ? 2 : g - 2, kx1 = g < 2 ? g : 2;
    const fr_h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned char *w1l = smem + C::W1_OFF, *w2l = smem + C::W2_OFF;
    auto load_w1 = [&](int blk) { return C::WLDS ? fr_load_w(w1l, C::NB1, blk, lane) : fr_load_w(reinterpret_cast<const char *>(a.w1), C::NB1, blk, lane); };
    auto load_w2 = [&](int blk) { return C::WLDS ? fr_load_w(w2l, C::NB2, blk, lane) : fr_load_w(reinterpret_cast<const char *>(a.w2), C::NB2, blk, lane); };

    // ---- 2. first conv: wave r = row r of the intermediate region, MT1 M-tiles x 2 cout tiles
    fr_f32x4 acc[C::MT1][2];
#pragma unroll
    for (int m = 0; m < C::MT1; ++m) acc[m][0] = acc[m][1] = fr_f32x4{0.f, 0.f, 0.f, 0.f};
    FrW wcur;
    if (!C::WLDS) wcur = load_w1(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the window (and the weights) have landed
    FR_PROBE(1);
    __syncthreads();
    FR_PROBE(2);
    if (C::WLDS) wcur = load_w1(0);
    {
        const int r = wave;
        fr_h8 col_h[C::MT1], col_m[C::MT1];
#pragma unroll
        for (int m = 0; m < C::MT1; ++m) col_h[m] = col_m[m] = zero8;
        auto xfrag = [&](int rd, int ky, int kx, int m, fr_h8 &h, fr_h8 &md) {
            // entries 2 rd, 2 rd + 1 of term 0 / 1: plane (term * 4 + entry)
            const unsigned char *p = xs + ((r + ky) * C::XC + 16 * m + li + kx) * 8 + (2 * rd) * C::XPLANE;
            h = fr_join(*reinterpret_cast<const fr_h4 *>(p), *reinterpret_cast<const fr_h4 *>(p + C::XPLANE));
            md = fr_join(*reinterpret_cast<const fr_h4 *>(p + C::C0G * C::XPLANE), *reinterpret_cast<const fr_h4 *>(p + (C::C0G + 1) * C::XPLANE));
        };
        // the three products of one weight block with all M-tiles: product by product, so that consecutive matrix instructions
        // never hit the same accumulator (a dependent v_mfma pair waits out the first one's latency)
        auto mfmas = [&](const FrW &w, const fr_h8 (&fh)[C::MT1], const fr_h8 (&fm)[C::MT1]) {
#pragma unroll
            for (int m = 0; m < C::MT1; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = PF_MFMA_SPLIT(w.h[n], fm[m], acc[m][n]);
#pragma unroll
            for (int m = 0; m < C::MT1; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = PF_MFMA_SPLIT(w.m[n], fh[m], acc[m][n]);
#pragma unroll
            for (int m = 0; m < C::MT1; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = PF_MFMA_SPLIT(w.h[n], fh[m], acc[m][n]);
        };
        int blk = 0;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const FrW wnext = load_w1(blk + 1);   // (the block after the last full one is the collected block)
                const int ky = s == 0 ? ky0 : ky1, kx = s == 0 ? kx0 : kx1;
                fr_h8 fh[C::MT1], fm[C::MT1];
#pragma unroll
                for (int m = 0; m < C::MT1; ++m) xfrag(rd, ky, kx, m, fh[m], fm[m]);
                mfmas(wcur, fh, fm);
                wcur = wnext;
                ++blk;
            }
            if (g == rd) {   // the ninth tap of this round's entries: K-slice rd of the collected instruction
#pragma unroll
                for (int m = 0; m < C::MT1; ++m) xfrag(rd, 2, 2, m, col_h[m], col_m[m]);
            }
        }
        mfmas(wcur, col_h, col_m);
        FR_PROBE(3);

        // ---- 3. bias, ReLU, zero outside the image, split, -> LDS (odd / even columns in separate planes)
        const int gy = yr0 + r;
        const bool row_in = gy >= 0 && gy < a.H1;
#pragma unroll
        for (int m = 0; m < C::MT1; ++m) {
            const int j = 16 * m + li;                   // local column of this lane's pixel
            if (j >= C::YC) continue;
            const int gx = yc0 + j;
            const bool in = row_in && gx >= 0 && gx < a.W1;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int gi = n * 4 + g;                // channel group of the intermediate tensor
                if (gi >= C::C1G) continue;
                const fr_f32x4 b4 = *reinterpret_cast<const fr_f32x4 *>(bias_lds + n * 16 + 4 * g);
                fr_f32x4 v = acc[m][n];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = v[q] * a.scale1 + b4[q];
                    if (a.relu1) v[q] = fmaxf(v[q], 0.f);
                    v[q] = in ? v[q] : 0.f;              // the second conv's zero padding
                }
                range_commit(a.status, range_acc(0.f, v[0], v[1], v[2], v[3]));   // (rare path: never taken in range)
                fr_h4 hi, mid;
                split_terms4(v, hi, mid);
                unsigned char *p = ys + (size_t)gi * C::YPAIR + ((j & 1) ? C::YPA + (r * C::YSB + (j >> 1)) * 8 : (r * C::YSA + (j >> 1)) * 8);
                *reinterpret_cast<fr_h4 *>(p) = hi;
                *reinterpret_cast<fr_h4 *>(p + C::C1G * C::YPAIR) = mid;
            }
        }
    }
    FrW w2;
    if (!C::WLDS) w2 = load_w2(0);
    FR_PROBE(4);
    __syncthreads();
    FR_PROBE(5);
    if (C::WLDS) w2 = load_w2(0);

    // ---- 4. second conv (stride 2): waves 0..3.  TW2 = 32: wave = (output row, 16-pixel half), both cout tiles;
    //      TW2 = 16: wave = (output row, cout tile)
    if (wave < 4) {
        constexpr int NN = C::NN2;
        const int ry = wave >> 1, hx = NN == 2 ? (wave & 1) : 0, n0 = NN == 2 ? 0 : (wave & 1);
        // one accumulator PER PRODUCT (summed at the end): with 1-2 cout tiles per wave the three products of a block would
        // otherwise be a dependent chain on one register quad
        fr_f32x4 acc2[NN][3];
#pragma unroll
        for (int n = 0; n < NN; ++n) acc2[n][0] = acc2[n][1] = acc2[n][2] = fr_f32x4{0.f, 0.f, 0.f, 0.f};
        // local column 2 (16 hx + i) + kx: kx = 0, 2 -> the j-even plane (slots i, i + 1), kx = 1 -> the j-odd plane (slot i)
        auto yfrag = [&](int rd, int ky, int kx, fr_h8 &h, fr_h8 &md) {
            const int slot = 16 * hx + li + (kx >> 1);
            const unsigned char *p = ys + (size_t)(2 * rd) * C::YPAIR +
                                     ((kx & 1) ? C::YPA + ((2 * ry + ky) * C::YSB + slot) * 8 : ((2 * ry + ky) * C::YSA + slot) * 8);
            h = fr_join(*reinterpret_cast<const fr_h4 *>(p), *reinterpret_cast<const fr_h4 *>(p + C::YPAIR));
            md = fr_join(*reinterpret_cast<const fr_h4 *>(p + C::C1G * C::YPAIR), *reinterpret_cast<const fr_h4 *>(p + (C::C1G + 1) * C::YPAIR));
        };
        auto mfma3 = [&](const FrW &w, const fr_h8 &fh, const fr_h8 &fm) {
#pragma unroll
            for (int n = 0; n < NN; ++n) {
                const fr_h8 wh = NN == 2 ? w.h[n] : (n0 ? w.h[1] : w.h[0]), wm = NN == 2 ? w.m[n] : (n0 ? w.m[1] : w.m[0]);
                acc2[n][0] = PF_MFMA_SPLIT(wh, fm, acc2[n][0]);
                acc2[n][1] = PF_MFMA_SPLIT(wm, fh, acc2[n][1]);
                acc2[n][2] = PF_MFMA_SPLIT(wh, fh, acc2[n][2]);
            }
        };
        fr_h8 col_h = zero8, col_m = zero8;
        int blk = 0;
#pragma unroll
        for (int rd = 0; rd < 3; ++rd) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const FrW wnext = load_w2(blk + 1);
                fr_h8 fh, fm;
                yfrag(rd, s == 0 ? ky0 : ky1, s == 0 ? kx0 : kx1, fh, fm);
                mfma3(w2, fh, fm);
                w2 = wnext;
                ++blk;
            }
            if (g == rd) yfrag(rd, 2, 2, col_h, col_m);
        }
        mfma3(w2, col_h, col_m);
        FR_PROBE(6);

        // ---- 5. bias, ReLU, range guard, store: lane (g, i) = couts 4g..4g+3 of output pixel (oy0 + ry, ox0 + 16 hx + i)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) the compiler can see: nothing left to protect inside the stores (conv_s4.hip)
        const int oy = oy0 + ry, ox = ox0 + 16 * hx + li;
        if (oy < a.H2 && ox < a.W2) {
            const size_t hw = (size_t)a.H2 * a.W2, pix = (size_t)oy * a.W2 + ox;
            const size_t term = (size_t)a.dst_c4 * hw * 8;
            const bool mis = (a.dst_choff & 2) != 0;
            float vmax = 0.f;
            typedef split_x2 h2;
#pragma unroll
            for (int nn = 0; nn < NN; ++nn) {
                const int n = n0 + nn;
                const int co = n * 16 + 4 * g;
                if (co >= a.C2 + 2) continue;
                const fr_f32x4 b4 = *reinterpret_cast<const fr_f32x4 *>(bias_lds + 32 + n * 16 + 4 * g);
                fr_f32x4 v = (acc2[nn][0] + acc2[nn][1]) + acc2[nn][2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = v[q] * a.scale2 + b4[q];
                    if (a.relu2) v[q] = fmaxf(v[q], 0.f);
                }
                vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
                if (a.dst_fmt) {
                    fr_h4 hi, mid;
                    split_terms4(v, hi, mid);
                    const int chb = a.dst_choff + co;
                    const bool ok0 = chb < a.dst_limit, ok1 = chb + 2 < a.dst_limit;
                    char *p = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * term + pix * 8 + (size_t)(chb >> 2) * hw * 8;
                    if (!mis) {
                        if (ok1) {
                            *reinterpret_cast<fr_h4 *>(p) = hi;
                            *reinterpret_cast<fr_h4 *>(p + term) = mid;
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
                    for (int q = 0; q < 4; ++q)
                        if (co + q < a.C2) a.dst[((size_t)b * a.dst_ctotal + a.dst_choff + co + q) * hw + pix] = v[q];
                }
            }
            range_commit(a.status, vmax);
        }
    }
    FR_PROBE(7);
#endif
}

// shapes this kernel is built for: 16 -> C1 <= 24 -> C2 <= 32 channels, 3x3 stride 1 then 3x3 stride 2, even width
bool conv_front_supports(int c0, int c1, int c2, int h1, int w1) { return c0 == 16 && c1 == 24 && c2 <= 32 && c2 > 16 && (w1 & 3) == 0 && h1 >= 2; }

template <int TW2_, bool WLDS_>
static int launch_front_cfg(const FrontArgs &a0, int B, hipStream_t s) {
    using C = FrontCfg<TW2_, WLDS_>;
    static_assert(C::LDS_BYTES <= 81920, "two workgroups per CU");
    FrontArgs a = a0;
    a.tilesX = (a.W2 + C::TW2 - 1) / C::TW2;
    a.tilesY = (a.H2 + C::TH2 - 1) / C::TH2;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_front_kernel<TW2_, WLDS_>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_front_kernel<%d, %d>(pf::FrontArgs)", TW2_, (int)WLDS_);
    const double px1 = (double)B * a.H1 * a.W1, px2 = (double)B * a.H2 * a.W2;
    ProfScope ps(s, label, 2.0 * 9 * (px1 * 16 * a.C1 + px2 * a.C1 * a.C2), 4.0 * (px1 * 16 + px2 * a.C2 + 9.0 * (16 * a.C1 + a.C1 * a.C2)));
    hipLaunchKernelGGL((conv_front_kernel<TW2_, WLDS_>), dim3(a.tilesX * a.tilesY, 1, B), dim3(C::NTHR), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_front_kernel");
    return PF_OK;
}

// variant 1: 2 x 32 output tiles, weights from L2 one block ahead; variant 2: 2 x 16 tiles, both weight sets resident in LDS
int launch_conv_front(const FrontArgs &a, int variant, int B, hipStream_t s) {
    return variant == 2 ? launch_front_cfg<16, true>(a, B, s) : launch_front_cfg<32, false>(a, B, s);
}

}  // namespace pf
