// Wave-autonomous fp32 MFMA convolution for the low-resolution layers (stride 1, input width % 4 == 0).
//
// Same GEMM view as conv_dma.hip (v_mfma_f32_16x16x4_f32: M = 16 pixels of a row, N = 16 couts, K = 4 input
// channels per tap), organised for layers whose tensors fit the XCD L2s (<= 64x128 at 1024x2048 input), where
// the profile (profiles/r01_c_*) showed conv_dma.hip idle: a 16x32 or 32x64 image has too few pixel tiles to
// fill 1024 SIMDs, and its barrier-synchronised rounds expose the staging latency.  Here
//   * a workgroup is WK waves that share NOTHING in the main loop: wave w owns input-channel chunks
//     w, w+WK, ... of the same output tile (MH rows x 16 pixels, NT cout tiles), so there is no barrier until the
//     final K-reduction through LDS - latency is hidden by the other waves on the SIMD, not by a schedule;
//   * the halo tile of a chunk goes global -> LDS by DMA (global_load_lds_dwordx4, L2-hit rate measured at
//     58 B/clk/CU, tools/ubench/dma_bw.hip) into a wave-private two-stage ring;
//   * the weights of a chunk go L2 -> VGPR directly (fragment order, one dwordx4 per 4 MFMAs): no LDS space,
//     no ds_read for the B operand, so 16-32 waves fit a CU;
//   * small tiles (MH = 1 or 2) keep >= 1 wave per SIMD even for a 16x32 image.
#include <cstring>

#include "conv_epilogue.h"
#include "pf_prof.h"

#ifndef PF_PROBE
#define PF_PROBE 0
#endif
#if PF_PROBE
#define PROBE() do { if (lane == 0 && blockIdx.x == 0 && blockIdx.y == 0 && wk == 0 && a.probe && np < 60) a.probe[np++] = clock64(); } while (0)
#else
#define PROBE() do { } while (0)
#endif

namespace pf {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KS, int MH, int NT, int WK>
struct WaveCfg {
    static constexpr int KC = wave_kc_ct(KS);       // input channels per chunk
    static constexpr int KG = KC / 4;               // MFMA k-groups per chunk
    static constexpr int KS2 = KS * KS;
    static constexpr int NV = KG * KS2;             // B values per lane per (cout tile, chunk)
    static constexpr int NQ = NV / 4, NR = NV % 4;  // ... as NQ dwordx4 loads + one NR-dword load (NR in {0, 2})
    static constexpr int WCH = NQ * 256 + NR * 64;  // packed floats per (cout tile, chunk)
    static_assert(NR == 0 || NR == 2, "remainder load is a dwordx2");
    static constexpr int APRON = KS == 3 ? 4 : 0;   // left/right apron (floats): rows stay 16-B aligned
    static constexpr int IW = 16 + 2 * APRON, IH = MH + KS - 1;
    static constexpr int RAW = IH * IW;
    static constexpr int PLANE = (RAW + 15) / 32 * 32 + 16;   // == 16 (mod 32): conflict-free A reads
    static constexpr int PP = PLANE / 4, RP = RAW / 4, RW = IW / 4;
    static constexpr int NPI = KC * PP;             // 16-B pieces per stage
    static constexpr int NIT = (NPI + 63) / 64;     // DMA instructions per stage
    static constexpr int STAGE = NIT * 256;         // floats per stage (whole instructions)
    static constexpr int RING = 2 * STAGE;          // floats per wave
    static constexpr int RED = WK * MH * NT * 256;  // floats: K-split partial sums
    static constexpr int LDS_FLOATS = WK * RING > RED ? WK * RING : RED;
    static constexpr int NL = NIT + NT * (NQ + (NR ? 1 : 0));   // VMEM loads per stage (for s_waitcnt vmcnt)
    static constexpr int NA = MH * NT == 1 ? 2 : 1; // independent accumulator chains per output fragment
    static_assert(PLANE >= RAW && PLANE % 32 == 16, "plane stride");
};

typedef __attribute__((address_space(3))) void *lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Buffer resource over one input tensor (one batch sample): raw buffer, offsets in bytes.  The DMA below is the
// MUBUF form (buffer_load_dwordx4 ... lds), not global_load_lds: (1) a lane whose voffset is >= num_records reads
// zeros, which is exactly the convolution's zero padding - no zero page, no select; (2) hipcc keeps exact in-order
// vmcnt bookkeeping for MUBUF DMA mixed with ordinary loads, whereas the FLAT-encoded global_load_lds makes it fall
// back to vmcnt(0) before every use of a loaded register (seen in the ISA of the first version of this kernel).
[[maybe_unused]] constexpr unsigned kOob = 0x80000000u;       // >= num_records: reads as 0
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0x7FFFFFFF, 0x00020000);
}
#endif

// EPI: 0 = bias + ReLU, 1 = fused stages of conv_epilogue.h (separate instantiations: register budget)
template <int KS, int MH, int NT, int WK, int EPI>
__global__ __launch_bounds__(64 * WK) void conv_wave_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)   // the body uses device-only builtin types; the host pass only needs the stub
    using C = WaveCfg<KS, MH, NT, WK>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    struct WFrag {      // the B operands of one (cout tile, chunk): NV values per lane
        f32x4 q[C::NQ];
        f32x2 r;
    };

    const int lane = threadIdx.x & 63;
    const int wk = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: chunk bookkeeping stays scalar
    const int tileY = blockIdx.x / a.tilesX, tileX = blockIdx.x - tileY * a.tilesX;
    const int tile0 = blockIdx.y * NT, b = blockIdx.z;
    const int oy0 = tileY * MH, ox0 = tileX * 16;
    const int iy0 = oy0 - KS / 2, ix0 = ox0 - C::APRON;
    float *ring = smem + wk * C::RING;
#if PF_PROBE
    int np = 0;
#endif
    PROBE();

    // ---- per-lane constants of the staging pattern: piece p = it*64 + lane covers 4 floats of one input row of
    //      channel pcl (within the chunk); voff = its byte offset from the chunk's first channel plane
    const unsigned in_plane = (unsigned)a.Hin * a.Win;
    unsigned voff[C::NIT];
    int pcl[C::NIT];
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
        const int p = it * 64 + lane;
        const int cl = p / C::PP, q = p - cl * C::PP;
        const int row = q / C::RW, j = q - row * C::RW;
        const int gy = iy0 + row, gx = ix0 + j * 4;
        const bool ok = p < C::NPI && q < C::RP && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        voff[it] = ok ? (cl * in_plane + (unsigned)(gy * a.Win + gx)) * 4u : kOob;
        pcl[it] = cl;
    }
    // input halo tile of `chunk` -> LDS stage `buf` (DMA), its weights -> registers `w`.  Everything about the chunk
    // is wave-uniform (which tensor, first channel, how many of its KC channels exist): select chains over kernel
    // arguments on the scalar ALU (no indexed access: that would go through scratch and a waterfall loop around the
    // descriptor); per piece only a compare + select remain on the vector ALU.
    auto issue = [&](int chunk, float *buf, WFrag(&w)[NT]) {
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
        const int nvalid = (cend - c0) - lc * C::KC;      // channels left in this range
        const unsigned soff = (unsigned)(coff + lc * C::KC) * in_plane * 4u;
        const __amdgpu_buffer_rsrc_t r = make_rsrc(sp + (size_t)b * ctot * in_plane);
#pragma unroll
        for (int it = 0; it < C::NIT; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(buf + it * 256), 16, pcl[it] < nvalid ? voff[it] : kOob,
                                                     soff, 0, 0);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int t = min(tile0 + n, a.ntiles - 1);   // a cout tile past the end recomputes the last one; never stored
            const float *wp = a.wpk + ((size_t)t * a.nchunks + chunk) * C::WCH;
#pragma unroll
            for (int q = 0; q < C::NQ; ++q) w[n].q[q] = *reinterpret_cast<const f32x4 *>(wp + q * 256 + lane * 4);
            if (C::NR) w[n].r = *reinterpret_cast<const f32x2 *>(wp + C::NQ * 256 + lane * 2);
        }
    };

    float biasv[NT];   // fetched now, used in the epilogue: the latency hides behind the main loop
#pragma unroll
    for (int n = 0; n < NT; ++n) biasv[n] = epi_bias(a, (tile0 + n) * 16 + (lane & 15));

    f32x4 acc[MH][NT][C::NA];
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int s = 0; s < C::NA; ++s) acc[m][n][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int abase = (lane >> 4) * C::PLANE + (lane & 15) + (C::APRON - KS / 2);

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

    // A operands are fetched one (k-group, tap) step ahead of the MFMAs that use them, so the LDS latency of step
    // s+1 is covered by the MP*NT MFMAs of step s instead of stalling the matrix pipe at every step.
    auto compute = [&](const float *buf, const WFrag(&w)[NT], int nv) {
        auto aload = [&](int idx, float(&af)[MH]) {
            const int kg = idx / C::KS2, tap = idx - kg * C::KS2;
            const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
            for (int m = 0; m < MH; ++m) af[m] = buf[abase + kg * 4 * C::PLANE + (m + ky) * C::IW + kx];
        };
        float a0[MH], a1[MH];
        aload(0, a0);
#pragma unroll
        for (int idx = 0; idx < C::NV; ++idx) {
            if (idx >= C::KS2 && (idx / C::KS2) * 4 >= nv) break;   // wave-uniform: k-groups past the end of the range
            float(&cur)[MH] = (idx & 1) ? a1 : a0;
            float(&nxt)[MH] = (idx & 1) ? a0 : a1;
            if (idx + 1 < C::NV) aload(idx + 1, nxt);
#pragma unroll
            for (int m = 0; m < MH; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n][idx % C::NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        cur[m], idx < 4 * C::NQ ? w[n].q[(idx / 4) % (C::NQ ? C::NQ : 1)][idx % 4] : w[n].r[idx % 2],
                        acc[m][n][idx % C::NA], 0, 0, 0);
        }
    };

    // ---- main loop: this wave's chunks wk, wk+WK, ...; two-stage ring, registers double-buffered by hand
    const int cb = a.chunk_begin, nch = a.chunk_end - cb;
    const int nmine = nch > wk ? (nch - wk + WK - 1) / WK : 0;
    WFrag w0[NT], w1[NT];
    PROBE();
    // The prefetch is issued unconditionally (past the end it re-reads this wave's last chunk into the idle stage):
    // straight-line code lets the compiler's vmcnt bookkeeping stay exact - a branch around the prefetch makes it
    // fall back to vmcnt(0) before the MFMAs and the pipeline degenerates to load -> wait -> compute.
    const int first = cb + wk, last = first + (nmine - 1) * WK;
    if (nmine > 0) {
        issue(first, ring, w0);
        PROBE();
        for (int i = 0; i < nmine; i += 2) {
            issue(min(first + (i + 1) * WK, last), ring + C::STAGE, w1);
            PROBE();
            wait_vm<C::NL>();
            PROBE();
            compute(ring, w0, chunk_valid(first + i * WK));
            PROBE();
            if (i + 1 < nmine) {
                issue(min(first + (i + 2) * WK, last), ring, w0);
                PROBE();
                wait_vm<C::NL>();
                PROBE();
                compute(ring + C::STAGE, w1, chunk_valid(first + (i + 1) * WK));
                PROBE();
            }
        }
        wait_vm<0>();   // the trailing prefetch must land before the ring is reused for the reduction
    }
    PROBE();

    // ---- K-split reduction through LDS (fixed order => deterministic), fragment (m,n) summed by wave (m*NT+n) % WK
    __syncthreads();   // every wave is done with its ring
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 v = acc[m][n][0];
            if (C::NA == 2) v += acc[m][n][C::NA - 1];
            red[((m * NT + n) * WK + wk) * 64 + lane] = v;
        }
    ResWin rw = ResWin();
    const lds_float *res_lds = nullptr;
    const bool has_res = EPI == 1 && a.res && a.res_lds_off >= 0;
    if (has_res) {   // residual window of this tile -> LDS behind the partial sums
        rw = res_window(a, oy0, MH, ox0, 16);
        res_stage(a, rw, b, tile0 * 16, NT * 16, (lds_float *)(smem + a.res_lds_off), threadIdx.x, 64 * WK);
        res_lds = (const lds_float *)(smem + a.res_lds_off);
    }
    __syncthreads();
    PROBE();

    int lane_e = lane;   // laundered: keeps the epilogue's coordinate / tap arithmetic below the main loop (VGPR pressure)
    asm volatile("" : "+v"(lane_e));
    // unit = one output fragment (or, when pooling, the fragments of rows m, m+1): summed and finished by one wave
    const int rows = (EPI == 1 && a.pool) ? 2 : 1;
    float vmax = 0.f;

#pragma unroll
    for (int u = 0; u < MH * NT; ++u) {
        if (u % WK != wk || u >= (MH / rows) * NT) continue;
        const int m = (u / NT) * rows, n = u % NT;
        const int co = (tile0 + n) * 16 + (lane_e & 15);
        const int oy = oy0 + m, ox = ox0 + (lane_e >> 4) * 4;
        if (co >= a.Cout || oy >= a.Hout || ox >= a.Wout) continue;
        auto total = [&](int mm) {
            const int mn = mm * NT + n;
            f32x4 v = red[(mn * WK) * 64 + lane_e];
#pragma unroll
            for (int k = 1; k < WK; ++k) v += red[(mn * WK + k) * 64 + lane_e];
            return v;
        };
        if (EPI == 0) {
            f32x4 v = total(m);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] += biasv[n];
                if (a.relu) v[r] = fmaxf(v[r], 0.f);
            }
            epi_store(a, b, co, oy, ox, v, vmax);
        } else {
            const lds_float *chan = res_lds + (n * 16 + (lane_e & 15)) * rw.cs;
            ResTaps t0, t1;
            if (has_res) {
                t0 = res_taps(a, rw, oy, ox);
                if (a.pool && oy + 1 < a.Hout) t1 = res_taps(a, rw, oy + 1, ox);
            }
            const f32x4 top = epi_finish(a, b, co, oy, ox, total(m), biasv[n], has_res, chan, &t0);
            if (a.pool) {
                if (MH > 1 && oy + 1 < a.Hout)
                    epi_store_pooled(a, b, co, oy, ox, top, epi_finish(a, b, co, oy + 1, ox, total(MH > 1 ? m + 1 : m), biasv[n], has_res, chan, &t1), vmax);
            } else {
                epi_store(a, b, co, oy, ox, top, vmax);
            }
            __builtin_amdgcn_sched_barrier(0);   // one fragment at a time (register pressure)
        }
    }
    range_commit(a.status, a.range_slot, vmax);
#endif
}

// ------------------------------------------------------------------------------------------------
template <int KS, int MH, int NT, int WK, int EPI>
static int launch_wave_epi(const ConvArgs &a0, int B, hipStream_t s) {
    using C = WaveCfg<KS, MH, NT, WK>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + 15) / 16;
    a.tilesY = (a.Hout + MH - 1) / MH;
    size_t lds = (size_t)C::LDS_FLOATS * sizeof(float);
    if (lds > 64 * 1024) return fail(PF_EUNSUPPORTED, "conv_wave<%d,%d,%d,%d>: %zu B of LDS", KS, MH, NT, WK, lds);
    a.res_lds_off = -1;
    if (a.res) {   // staged residual window lives behind the K-split partial sums
        const size_t need = (size_t)C::RED + (size_t)NT * 16 * res_chan_stride(res_extent(MH, a.res_sh), res_extent(16, a.res_sw));
        if (need * sizeof(float) > 64 * 1024) return fail(PF_EUNSUPPORTED, "residual window of %zu B does not fit LDS", need * sizeof(float));
        a.res_lds_off = C::RED;
        if (need * sizeof(float) > lds) lds = need * sizeof(float);
    }
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_wave_kernel<KS, MH, NT, WK, EPI>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_wave_kernel<%d, %d, %d, %d, %d>(pf::ConvArgs)", KS, MH, NT, WK, EPI);
    if (a.res) strncat(label, " +res", sizeof(label) - strlen(label) - 1);
    if (a.pool) strncat(label, " +pool", sizeof(label) - strlen(label) - 1);
    if (a.no_bias) strncat(label, " lowres-half", sizeof(label) - strlen(label) - 1);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * KS * KS,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * KS * KS));
    hipLaunchKernelGGL((conv_wave_kernel<KS, MH, NT, WK, EPI>), dim3(a.tilesX * a.tilesY, (a.ntiles + NT - 1) / NT, B),
                       dim3(64 * WK), lds, s, a);
    PF_LAUNCH_CHECK("conv_wave_kernel");
    return PF_OK;
}

template <int KS, int MH, int NT, int WK>
static int launch_wave_cfg(const ConvArgs &a, int B, hipStream_t s) {
    if (a.pool || a.res || a.no_bias) {
        if (KS == 1) return launch_wave_epi<KS, MH, NT, WK, KS == 1 ? 1 : 0>(a, B, s);
        return fail(PF_EUNSUPPORTED, "conv_wave: fused epilogue stages are built for 1x1 convs only");
    }
    return launch_wave_epi<KS, MH, NT, WK, 0>(a, B, s);
}

int launch_conv_wave(const ConvArgs &a, int ks, int mh, int nt, int wk, int B, hipStream_t s) {
#define PF_CASE(KS_, MH_, NT_, WK_) \
    if (ks == KS_ && mh == MH_ && nt == NT_ && wk == WK_) return launch_wave_cfg<KS_, MH_, NT_, WK_>(a, B, s);
#define PF_CASES_WK(KS_, MH_, NT_) PF_CASE(KS_, MH_, NT_, 2) PF_CASE(KS_, MH_, NT_, 4) PF_CASE(KS_, MH_, NT_, 8) PF_CASE(KS_, MH_, NT_, 16)
    PF_CASES_WK(3, 1, 1) PF_CASES_WK(3, 1, 2) PF_CASES_WK(3, 2, 1) PF_CASES_WK(3, 2, 2) PF_CASES_WK(3, 4, 1) PF_CASES_WK(3, 4, 2)
    PF_CASES_WK(1, 1, 1) PF_CASES_WK(1, 1, 2) PF_CASES_WK(1, 2, 1) PF_CASES_WK(1, 2, 2) PF_CASES_WK(1, 4, 1) PF_CASES_WK(1, 4, 2)
#undef PF_CASES_WK
#undef PF_CASE
    return fail(PF_EUNSUPPORTED, "conv_wave: no kernel for ks=%d mh=%d nt=%d wk=%d", ks, mh, nt, wk);
}

int wave_chunks(const int *src_ch, int n_src, int ks) {
    const int kc = wave_kc(ks);
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += (src_ch[j] + kc - 1) / kc;
    return n;
}

// fragment-order packing for direct L2 -> VGPR loads, per (cout tile, chunk): [quad][64 lanes][4] then the
// remainder [64 lanes][NR];  value idx = q*4+e (or 4*NQ+e) -> (kg, tap) = (idx / ks2, idx % ks2):
//   W[t*16 + (lane&15)][first channel of the chunk + kg*4 + (lane>>4)][tap]
// K order: the input ranges in order, each padded to whole chunks (zeros), so a chunk lies inside one tensor.
void pack_conv_weights_wave(const float *w, int cin, int cout, int ks, const int *src_ch, int n_src, float *out) {
    const int kc = wave_kc(ks), ks2 = ks * ks, nv = (kc / 4) * ks2, nq = nv / 4, nr = nv % 4;
    const int ntiles = (cout + 15) / 16;
    size_t o = 0;
    for (int t = 0; t < ntiles; ++t) {
        int c0 = 0;
        for (int j = 0; j < n_src; ++j) {
            for (int lc = 0; lc * kc < src_ch[j]; ++lc) {
                auto val = [&](int idx, int lane) {
                    const int kg = idx / ks2, tap = idx % ks2;
                    const int co = t * 16 + (lane & 15), cl = lc * kc + kg * 4 + (lane >> 4);
                    return (co < cout && cl < src_ch[j]) ? w[((size_t)co * cin + c0 + cl) * ks2 + tap] : 0.f;
                };
                for (int q = 0; q < nq; ++q)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 4; ++e) out[o++] = val(q * 4 + e, lane);
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < nr; ++e) out[o++] = val(nq * 4 + e, lane);
            }
            c0 += src_ch[j];
        }
    }
}

size_t wave_packed_floats(const int *src_ch, int n_src, int cout, int ks) {
    const int kc = wave_kc(ks), nv = (kc / 4) * ks * ks;
    return (size_t)((cout + 15) / 16) * wave_chunks(src_ch, n_src, ks) * ((nv / 4) * 256 + (nv % 4) * 64);
}

}  // namespace pf
