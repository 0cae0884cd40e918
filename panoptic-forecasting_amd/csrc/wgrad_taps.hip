// Backward-weight of the 3x3, stride-1 convolutions of a training step with the TAPS folded into the matrix rows (round 6).
//
//   dW[co][ci][ky][kx] = sum over (b, iy, ix) of dy[b][co][iy - ky + 1][ix - kx + 1] * x[b][ci][iy][ix]
//
// train_kernels.hip: wgrad_tiled_kernel gives every tap its own accumulator: M = 16 couts per v_mfma_f32_16x16x4_f32, so a layer of
// 10 output channels fills 10 of 16 matrix rows and one of 18 fills 18 of 32 (HarDNet's odd layers have 10, 16, 18, 24 ... outputs,
// the even ones 18, 28, 30, 46 ...: models/bg/hardnet.py:177-194).  Here the matrix rows are the (cout, tap) PAIRS, packed densely:
// row j = 9 co + tap, 16 rows per instruction, so 10 outputs are 90 rows = 6 instructions per four pixels instead of 9, 18 are 11
// instead of 18, 28 are 16 instead of 18.  The shifted operand is then dy (every lane reads dy at its own (cout, tap) offset - one
// ds_read_b32 per lane either way), and x is read WITHOUT a halo:
//
//   workgroup = (chunk of <= 32 couts, NC x 16 cins, slab of the items); item = R input rows x TW input pixels (R TW = 128);
//   staged per item by global_load_lds_dwordx4 (no staging registers, two LDS buffers, ONE barrier per item):
//       dy  [couts][R + 2 rows][72]   rows y0 - 1 .. y0 + R, columns x0 - 4 .. x0 + TW + 3, zeros outside the image (a 16-B zero page)
//       x   [NC 16][R TW (+ 4)]
//   pitches: dy row 72 == 8, dy cout == 24, x channel == 4 (mod 64 banks): the 64 lanes of a fragment read hit 64 different banks
//   (same-address lanes aside).  (Dense planes + x pieces XOR-swizzled by the channel - a third fewer DMA instructions, every lane
//   moving a piece that is read - measured 5-20 % SLOWER on every layer of 24+ outputs: profiles/r06_experiments.md);
//   its 8 waves take four 4-pixel groups of the item each, hold all (cout, tap) x cin accumulators and are added up through LDS at
//   the end in a fixed order; partial sums per slab -> wgrad_reduce_kernel as before (no atomics: a step stays bit-reproducible).
//   (TW, R) = (64, 2) for wide levels, (32, 4) for the 25- / 12-pixel levels (their rows are 28 and 12 floats: a 64-pixel item is
//   44 % / 19 % pixels, a 32-pixel one 88 % / 38 %).
#include "conv_epilogue.h"
#include "pf_prof.h"
#include "train_kernels.h"

namespace pf {

typedef float wt_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *wt_lds_ptr_t;
typedef __attribute__((address_space(1))) void *wt_gptr_t;

__device__ __attribute__((aligned(16))) float g_wt_zero[4];     // what a piece outside the image (or past the channels) reads

template <int T, int NC, int TW, int R>
struct WtCfg {
    static constexpr int NTHR = 512, NWAVE = 8;
    static constexpr int ROWS = R + 2, PR = 72;                         // dy rows y0 - 1 .. y0 + R, columns x0 - 4 .. x0 + TW + 3
    static constexpr int PC = (ROWS * PR - 24 + 63) / 64 * 64 + 24;     // >= ROWS * PR, == 24 (mod 64)
    static constexpr int COB = T * 16 / 9;                              // couts a workgroup can take
    static constexpr int PX = R * TW + 4;                               // == 4 (mod 64)
    static constexpr int DY_FLOATS = COB * PC, X_FLOATS = NC * 16 * PX, BUF = DY_FLOATS + X_FLOATS;
    static constexpr int NDY = COB * (PC / 4), NX = NC * 16 * (PX / 4); // 16-B slots
    static constexpr int ITD = (NDY + NTHR - 1) / NTHR, ITX = (NX + NTHR - 1) / NTHR;
    static constexpr int NG = R * TW / 4 / NWAVE;                       // 4-pixel groups per wave and item
    static constexpr int RED_FLOATS = NWAVE * 4 * 256;                  // the final sum: 4 accumulator tiles per pass
    static constexpr int LDS_FLOATS = 2 * BUF > RED_FLOATS ? 2 * BUF : RED_FLOATS;
    static_assert(R * TW == 128 && TW + 8 <= PR && PC >= ROWS * PR && PC % 64 == 24 && PX % 64 == 4 && PC % 4 == 0, "wgrad_taps geometry");
};

template <int T, int NC, int TW, int R>
__global__ __launch_bounds__(512, T <= 6 ? 4 : 2) void wgrad_taps_kernel(ConvArgs a, const float *dy, int B, int slabs, int cob, int co_pad, int ci_pad,
                                                                         float *partial) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = WtCfg<T, NC, TW, R>;
    extern __shared__ __attribute__((aligned(16))) float wt_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * cob, nco = min(cob, a.Cout - co0);
    const int ci0 = blockIdx.y * 16 * NC, slab = blockIdx.z;
    const int H = a.Hout, W = a.Wout;
    const int plane = H * W;
    const int chunks = (W + TW - 1) / TW, row_groups = (H + R - 1) / R;
    const int items = B * row_groups * chunks;
    const int t_act = (nco * 9 + 15) / 16;
    const float *zero = g_wt_zero;

    // ---- this thread's DMA slots (slot = 16 B of a buffer, in buffer order: a wave instruction fills 1 KB)
    int dyo[C::ITD], drow[C::ITD], dcol[C::ITD];
#pragma unroll
    for (int it = 0; it < C::ITD; ++it) {
        const int idx = it * C::NTHR + tid, c = idx / (C::PC / 4), s = idx - c * (C::PC / 4);
        const int row = s / (C::PR / 4), c4 = s - row * (C::PR / 4);
        const bool real = idx < C::NDY && c < nco && row < C::ROWS && c4 < (TW + 8) / 4;
        drow[it] = real ? row - 1 : -(1 << 20);
        dcol[it] = c4 * 4 - 4;
        dyo[it] = c * plane + (row - 1) * W + c4 * 4 - 4;
    }
    const float *xpl[C::ITX];
    int xbs[C::ITX], xo[C::ITX], xrow[C::ITX], xcol[C::ITX];
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
        const int idx = it * C::NTHR + tid, c = idx / (C::PX / 4), s = idx - c * (C::PX / 4);
        const int r = s / (TW / 4), c4 = s - r * (TW / 4);
        const int ci = ci0 + c;
        xpl[it] = nullptr;
        xbs[it] = 0;
        if (idx < C::NX && ci < a.Cin && s < R * TW / 4) {
            int sidx = 0;
            while (sidx + 1 < a.n_src && ci >= a.src_cstart[sidx + 1]) ++sidx;
            xpl[it] = a.src[sidx] + (long long)(a.src_choff[sidx] + (ci - a.src_cstart[sidx])) * plane;
            xbs[it] = a.src_ctotal[sidx] * plane;
        }
        xrow[it] = r;
        xcol[it] = c4 * 4;
        xo[it] = r * W + c4 * 4;
    }
    auto issue = [&](int item, float *buf) {
        const int ch = item % chunks, rowg = item / chunks;
        const int b = rowg / row_groups, y0 = (rowg - b * row_groups) * R, x0 = ch * TW;
        const float *dyb = dy + ((long long)b * a.Cout + co0) * plane + (long long)y0 * W + x0;
#pragma unroll
        for (int it = 0; it < C::ITD; ++it) {
            if (it * C::NTHR + tid >= C::NDY) continue;       // (lanes past the region stay off: they would land in x)
            const bool ok = (unsigned)(y0 + drow[it]) < (unsigned)H && (unsigned)(x0 + dcol[it]) < (unsigned)W;
            const float *p = ok ? dyb + dyo[it] : zero;
            __builtin_amdgcn_global_load_lds((wt_gptr_t)p, (wt_lds_ptr_t)(buf + (it * C::NTHR + wave * 64) * 4), 16, 0, 0);
        }
        float *xb = buf + C::DY_FLOATS;
        const long long xoff = (long long)y0 * W + x0;
#pragma unroll
        for (int it = 0; it < C::ITX; ++it) {
            if (it * C::NTHR + tid >= C::NX) continue;
            const bool ok = xpl[it] && y0 + xrow[it] < H && x0 + xcol[it] < W;
            const float *p = ok ? xpl[it] + (long long)b * xbs[it] + xoff + xo[it] : zero;
            __builtin_amdgcn_global_load_lds((wt_gptr_t)p, (wt_lds_ptr_t)(xb + (it * C::NTHR + wave * 64) * 4), 16, 0, 0);
        }
    };

    // ---- fragment offsets: lane (m, kq) of tile t reads dy of row j = 16 t + m -> (cout j / 9, tap j % 9) at pixel column kq
    const int m = lane & 15, kq = lane >> 4;
    int aoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = 16 * t + m;
        int co = j / 9;
        const int tap = j - co * 9, ky = tap / 3, kx = tap - ky * 3;
        co = co < C::COB ? co : C::COB - 1;       // (rows past the chunk: anything inside the buffer, never stored)
        aoff[t] = co * C::PC + (2 - ky) * C::PR + 5 - kx + kq;
    }
    wt_f32x4 acc[T][NC];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[t][n] = wt_f32x4{0.f, 0.f, 0.f, 0.f};

    int item = slab, cur = 0;
    if (item < items) issue(item, wt_lds);
    for (; item < items; item += slabs, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();       // this item has landed; everyone is done with the other buffer
        if (item + slabs < items) issue(item + slabs, wt_lds + (cur ^ 1) * C::BUF);       // in flight during the matrix instructions
        const float *dys = wt_lds + cur * C::BUF, *xs = dys + C::DY_FLOATS;
#pragma unroll
        for (int g = 0; g < C::NG; ++g) {
            const int gi = wave * C::NG + g, r = gi / (TW / 4), q = gi - r * (TW / 4);
            float bv[NC];
#pragma unroll
            for (int n = 0; n < NC; ++n) bv[n] = xs[(n * 16 + m) * C::PX + r * TW + q * 4 + kq];
            const float *ap = dys + r * C::PR + q * 4;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                if (t < t_act) {       // (uniform)
                    const float av = ap[aoff[t]];
#pragma unroll
                    for (int n = 0; n < NC; ++n) acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[n], acc[t][n], 0, 0, 0);
                }
            }
        }
    }

    // ---- the 8 waves' sums, four accumulator tiles per pass, fixed order -> partial[slab][co][ci][tap]
    float *red = wt_lds;
#pragma unroll
    for (int tb = 0; tb < T * NC; tb += 4) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (tb + u >= T * NC) continue;
            const int t = (tb + u) / NC, n = (tb + u) % NC;
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave * 4 + u) * 4 + i) * 64 + lane] = acc[t][n][i];
        }
        __syncthreads();
        for (int e = tid; e < 4 * 256; e += C::NTHR) {
            const int u = e >> 8, rr = e & 255, i = rr >> 6, l = rr & 63;
            const int t = (tb + u) / NC, n = (tb + u) % NC;
            if (tb + u >= T * NC || t >= t_act) continue;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < C::NWAVE; ++w) v += red[((w * 4 + u) * 4 + i) * 64 + l];
            const int j = 16 * t + 4 * (l >> 4) + i, co = j / 9, tap = j - co * 9, ci = ci0 + n * 16 + (l & 15);
            if (co < nco && ci < ci_pad) partial[(((long long)slab * co_pad + co0 + co) * ci_pad + ci) * 9 + tap] = v;
        }
    }
#endif
}

namespace {
template <int T, int NC, int TW, int R>
int launch_taps_cfg(const ConvArgs &a, const float *dy, int B, int slabs, int chunks, int cob, int co_pad, int ci_pad, float *partial, hipStream_t s) {
    using C = WtCfg<T, NC, TW, R>;
    static bool attr_set = false;
    constexpr int bytes = C::LDS_FLOATS * (int)sizeof(float);
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_taps_kernel<T, NC, TW, R>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        attr_set = true;
    }
    const dim3 grid(chunks, (a.Cin + 16 * NC - 1) / (16 * NC), slabs);
    hipLaunchKernelGGL((wgrad_taps_kernel<T, NC, TW, R>), grid, dim3(C::NTHR), bytes, s, a, dy, B, slabs, cob, co_pad, ci_pad, partial);
    return PF_OK;
}
template <int T, int NC>
int launch_taps_geo(const ConvArgs &a, const float *dy, int B, int slabs, int chunks, int cob, int co_pad, int ci_pad, float *partial, bool narrow, hipStream_t s) {
    if (narrow) return launch_taps_cfg<T, NC, 32, 4>(a, dy, B, slabs, chunks, cob, co_pad, ci_pad, partial, s);
    return launch_taps_cfg<T, NC, 64, 2>(a, dy, B, slabs, chunks, cob, co_pad, ci_pad, partial, s);
}
}  // namespace

bool wgrad_taps_ok(int ks, int stride, int Hin, int Win, int Hout, int Wout) {
    return ks == 3 && stride == 1 && Hin == Hout && Win == Wout && (Wout & 3) == 0;
}

// Per layer of a B = 8, 800 x 800 step, alone on the chip (tools/bench_train.py --no-side-stream --layers, profiles/r06_experiments.md):
//   <= 19 outputs per chunk (6 or 11 row tiles instead of 9 or 18 instructions per four pixels)      0.62 - 0.95 x wgrad_tiled_kernel
//   24+ outputs on the 400 .. 50-pixel levels (14 - 18 row tiles against 18)                           0.87 - 1.27 x: stays tiled
//   28- / 12-float rows (4 x 32 items: 88 % / 38 % pixels against 44 % / 19 %), 64+ / 120+ inputs      0.57 - 0.88 x
//   the smallest layers of those levels (40 -> 24, 54 -> 32, 92 -> 32: a few items per workgroup)       1.07 - 1.65 x: stay tiled
bool wgrad_taps_wanted(int mode, int ks, int stride, int Cin, int Cout, int Hin, int Win, int Hout, int Wout) {
    if (mode <= 0 || !wgrad_taps_ok(ks, stride, Hin, Win, Hout, Wout)) return false;
    if (mode >= 2) return true;
    const int chunks = (Cout + 31) / 32, cob = (Cout + chunks - 1) / chunks;
    if (Wout <= 16) return Cin >= 120;
    if (Wout <= 32) return Cin >= 64;
    return cob <= 19;
}

// workgroups = cout chunks x cin groups x slabs; `max_slabs` = what the partial buffer was sized for (train_kernels.hip: wgrad_slabs)
int launch_wgrad_taps(const ConvArgs &a, const float *dy, int B, int max_slabs, int co_pad, int ci_pad, float *partial, int *slabs_used, hipStream_t s) {
    const int chunks = (a.Cout + 31) / 32, cob = (a.Cout + chunks - 1) / chunks;
    const int t_need = (cob * 9 + 15) / 16;
    const int nc = a.Cin > 16 ? 2 : 1;
    const bool narrow = a.Wout <= 32;
    const int tw = narrow ? 32 : 64, r = narrow ? 4 : 2;
    const long long items = (long long)B * ((a.Hout + r - 1) / r) * ((a.Wout + tw - 1) / tw);
    const long long tiles = (long long)chunks * ((a.Cin + 16 * nc - 1) / (16 * nc));
    // one workgroup of 8 waves per CU where the two buffers take most of the LDS, two where they leave room
    const int target = t_need <= 6 ? 512 : 256;
    long long slabs = target / tiles;
    slabs = slabs < 1 ? 1 : slabs;
    slabs = slabs > items ? items : slabs;
    slabs = slabs > max_slabs ? max_slabs : slabs;
    *slabs_used = (int)slabs;
    int rc;
#define PF_TAPS(TT, NN) rc = launch_taps_geo<TT, NN>(a, dy, B, (int)slabs, chunks, cob, co_pad, ci_pad, partial, narrow, s)
    if (nc == 1) {
        if (t_need <= 6) PF_TAPS(6, 1);
        else if (t_need <= 11) PF_TAPS(11, 1);
        else PF_TAPS(18, 1);
    } else {
        if (t_need <= 6) PF_TAPS(6, 2);
        else if (t_need <= 11) PF_TAPS(11, 2);
        else PF_TAPS(18, 2);
    }
#undef PF_TAPS
    if (rc) return rc;
    PF_LAUNCH_CHECK("wgrad_taps_kernel");
    return PF_OK;
}

}  // namespace pf
