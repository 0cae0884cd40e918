// fp32 3x3/s1 convolution on the VECTOR ALU of gfx950, for layers whose output-channel count fits the 16-wide MFMA tile
// badly (FC-HarDNet's 10/18/24/28-channel HarDBlock layers carry a third of the network's time at 256x512).
// STATUS (round 1, profiles/README.md): correct everywhere, but it sustains ~80 TF/s against the MFMA kernels' 100-135
// executed TF/s; it beat them only on 18->10 and 16->24 at >= 256x512, and conv_split.hip now beats it there too: the
// shipped table selects it nowhere.  Kept (tested, forceable) as the measured fp32-only alternative for the layers the
// matrix tile fits worst.
//
// Why not the matrix cores: on gfx950 the fp32 MFMA runs at exactly the fp32 vector rate, and the two pipes share one
// power/clock budget - tools/ubench/valu_rate.hip measures 135 TF/s for MFMA alone, 128 TF/s for v_pk_fma_f32 alone and
// ~137 TF/s for both TOGETHER.  So fp32 MFMA buys operand bandwidth, not flops, and it pays for that with granularity:
// N in steps of 16 channels, K in steps of 4.  A 10-channel layer wastes 37 % of the matrix pipe, an 18-channel one 44 %
// (or needs the hybrid path of conv_dma.hip), a 28-channel one 12 %.  v_pk_fma_f32 has granularity 2 x 1.
//
// Mapping: one lane = one output column, ROWS output rows; a wave = 64 columns; a workgroup = 4 waves = 4*ROWS rows.
//   * the halo tile of KC input channels is staged by LDS-DMA exactly like conv_dma.hip (16-B pieces, 4-float apron, zero
//     fill by out-of-range offsets, double buffered, one barrier per round); lanes read their row neighbourhood with
//     ds_read2_b32 (consecutive lanes -> consecutive banks: conflict-free for any plane stride);
//   * weights never touch LDS or VGPRs: they are wave-uniform, so they are fetched through the scalar cache
//     (constant address space -> s_load_dwordxN) and used as the SGPR-pair operand of v_pk_fma_f32: one instruction =
//     2 output channels x 64 pixels, the activation broadcast to both halves by op_sel;
//   * accumulators: ROWS x Cout registers per lane, fp32 FMA chain (same arithmetic as the MFMA path up to order).
// Per input channel a wave issues (ROWS+2)*2 LDS reads, 9 scalar loads and 9*ROWS*Cout/2 packed FMAs.
#include <cstring>

#include "conv_mfma.h"
#include "pf_prof.h"

namespace pf {

typedef float v_f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) float cfloat;   // constant address space: uniform loads become s_load
typedef __attribute__((address_space(3))) void *valu_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kValuOob = 0x80000000u;

// acc.xy += w.xy * x.lo   /   acc.xy += w.xy * x.hi      (w: SGPR pair, x: VGPR pair)
#define PF_PK_FMA_LO(acc, w, x) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(x))
#define PF_PK_FMA_HI(acc, w, x) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(x))

template <int CP, int ROWS, int KC>
struct ValuCfg {
    static constexpr int TW = 64, TH = 4 * ROWS;
    static constexpr int IW = TW + 8, IH = TH + 2;       // 4-float apron left and right keeps rows 16-B aligned
    static constexpr int PLANE = IH * IW;                // multiple of 4
    static constexpr int PP = PLANE / 4, RW = IW / 4;
    static constexpr int BUF = KC * PLANE;
    static constexpr int CPAD = 2 * CP;                  // floats per (channel, tap) weight row
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * sizeof(float);
};

template <int CP, int ROWS, int KC>
__global__ __launch_bounds__(256) void conv_valu_kernel(ConvArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = ValuCfg<CP, ROWS, KC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tileY = blockIdx.x / a.tilesX, tileX = blockIdx.x - tileY * a.tilesX;
    const int b = blockIdx.z;
    const int iy0 = tileY * C::TH - 1, ix0 = tileX * C::TW - 4;

    v_f32x2 acc[ROWS][CP];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int j = 0; j < CP; ++j) acc[r][j] = v_f32x2{0.f, 0.f};

    // staging pattern of this thread (piece p = it*256 + tid of a KC-channel slot), as in conv_dma.hip
    constexpr int NPI = KC * C::PP, NITI = (NPI + 255) / 256;
    const unsigned in_plane = (unsigned)a.Hin * a.Win;
    unsigned voff[NITI];
    int pcl[NITI];
#pragma unroll
    for (int it = 0; it < NITI; ++it) {
        const int p = it * 256 + tid;
        const int cl = p / C::PP, q = p - cl * C::PP;
        const int row = q / C::RW, j = q - row * C::RW;
        const int gy = iy0 + row, gx = ix0 + j * 4;
        const bool ok = p < NPI && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        voff[it] = ok ? (cl * in_plane + (unsigned)(gy * a.Win + gx)) * 4u : kValuOob;
        pcl[it] = cl;
    }

    // which tensor / first channel / valid channels of a chunk: workgroup-uniform select chains (scalar ALU)
    auto chunk_src = [&](int chunk, const float *&sp, int &ctot, int &choff, int &nvalid) {
        sp = a.src[0];
        ctot = a.src_ctotal[0];
        int coff = a.src_choff[0], ch0 = 0, cend = a.src_cstart[1], c0 = 0;
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
        nvalid = (cend - c0) - lc * KC;
        choff = coff + lc * KC;
    };
    auto stage = [&](int chunk, float *buf) {
        const float *sp;
        int ctot, choff, nvalid;
        chunk_src(chunk, sp, ctot, choff, nvalid);
        const unsigned soff = (unsigned)choff * in_plane * 4u;
        const __amdgpu_buffer_rsrc_t r =
            __builtin_amdgcn_make_buffer_rsrc((void *)(sp + (size_t)b * ctot * in_plane), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int it = 0; it < NITI; ++it)
            if (it * 256 + tid < NPI)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (valu_lds_ptr_t)(buf + (it * 256 + wave * 64) * 4), 16,
                                                         pcl[it] < nvalid ? voff[it] : kValuOob, soff, 0, 0);
    };

    const int cb = a.chunk_begin, nrounds = a.chunk_end - cb;
    if (nrounds > 0) stage(cb, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    cfloat *wbase = (cfloat *)(uintptr_t)a.wpk;
    const int lbase = (wave * ROWS) * C::IW + lane + 3;   // column of tap kx=0 for this lane, first input row of the wave

    for (int round = 0; round < nrounds; ++round) {
        const float *cur = smem + (round & 1) * C::BUF;
        if (round + 1 < nrounds) stage(cb + round + 1, smem + ((round + 1) & 1) * C::BUF);
        const float *sp;
        int ctot, choff, nv;
        chunk_src(cb + round, sp, ctot, choff, nv);
        cfloat *wc = wbase + (size_t)(cb + round) * (KC * 9 * C::CPAD);
#pragma unroll 1
        for (int c = 0; c < KC; ++c) {
            if (c >= nv) break;   // wave-uniform: K has granularity 1 here, nothing is padded
            // the (ROWS+2) x 4 neighbourhood values of this lane: pairs (kx0,kx1) and (kx2, -)
            v_f32x2 p01[ROWS + 2], p2x[ROWS + 2];
#pragma unroll
            for (int r = 0; r < ROWS + 2; ++r) {
                const float *row = cur + c * C::PLANE + lbase + r * C::IW;
                p01[r] = v_f32x2{row[0], row[1]};
                p2x[r] = v_f32x2{row[2], row[3]};
            }
            // One scalar fetch of the channel's 9 x Cout weights, then the FMAs.  (Fetching a batch ahead into a second
            // SGPR set was tried: SMEM shares lgkmcnt with LDS and returns out of order, so every wait is lgkmcnt(0), and
            // two resident batches exceed the SGPR file next to ConvArgs - it measured no faster.  Other waves of the
            // SIMD cover the fetch.)
            cfloat *wch = wc + c * (9 * C::CPAD);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    cfloat *wt = wch + (ky * 3 + kx) * C::CPAD;
#pragma unroll
                    for (int j = 0; j < CP; ++j) {
                        const v_f32x2 w = v_f32x2{wt[2 * j], wt[2 * j + 1]};
#pragma unroll
                        for (int r = 0; r < ROWS; ++r) {
                            if (kx == 0) PF_PK_FMA_LO(acc[r][j], w, p01[r + ky]);
                            else if (kx == 1) PF_PK_FMA_HI(acc[r][j], w, p01[r + ky]);
                            else PF_PK_FMA_LO(acc[r][j], w, p2x[r + ky]);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next round landed
        __syncthreads();                                    // ... and everyone is done with `cur`
    }

    // ---- epilogue: bias + ReLU, one coalesced 256-B row segment per (channel, row)
    const int ox = tileX * C::TW + lane;
    if (ox >= a.Wout) return;
    const size_t opl = (size_t)a.Hout * a.Wout;
    float *dst = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff) * opl + ox;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const int oy = tileY * C::TH + wave * ROWS + r;
        if (oy >= a.Hout) continue;
#pragma unroll
        for (int j = 0; j < CP; ++j) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int co = 2 * j + h;
                if (co >= a.Cout) continue;
                float v = acc[r][j][h] + a.bias[co];
                if (a.relu) v = fmaxf(v, 0.f);
                range_commit(a.status, fabsf(v));   // conv_mfma.h: range guard of the operand split
                dst[(size_t)co * opl + (size_t)oy * a.Wout] = v;
            }
        }
    }
#endif
}

template <int CP, int ROWS, int KC>
static int launch_valu_cfg(const ConvArgs &a0, int B, hipStream_t s) {
    using C = ValuCfg<CP, ROWS, KC>;
    ConvArgs a = a0;
    a.tilesX = (a.Wout + C::TW - 1) / C::TW;
    a.tilesY = (a.Hout + C::TH - 1) / C::TH;
    static bool attr_set = false;
    if (!attr_set) {
        PF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_valu_kernel<CP, ROWS, KC>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    char label[96];
    snprintf(label, sizeof(label), "void pf::conv_valu_kernel<%d, %d, %d>(pf::ConvArgs)", CP, ROWS, KC);
    const double px = (double)B * a.Hout * a.Wout;
    ProfScope ps(s, label, 2.0 * px * a.Cout * a.Cin * 9,
                 4.0 * ((double)B * a.Cin * a.Hin * a.Win + px * a.Cout + (double)a.Cout * a.Cin * 9));
    hipLaunchKernelGGL((conv_valu_kernel<CP, ROWS, KC>), dim3(a.tilesX * a.tilesY, 1, B), dim3(256), C::LDS_BYTES, s, a);
    PF_LAUNCH_CHECK("conv_valu_kernel");
    return PF_OK;
}

// a.wpk = pack_conv_weights_valu() output; chunks of valu_kc() channels (a.src_chunk0 / chunk_begin / chunk_end set for it)
int launch_conv_valu(const ConvArgs &a, int rows, int B, hipStream_t s) {
    if (a.pool || a.res || a.no_bias) return fail(PF_EUNSUPPORTED, "conv_valu: no fused epilogue stages");
    if ((a.Win & 3) != 0 || a.Hin != a.Hout || a.Win != a.Wout) return fail(PF_EUNSUPPORTED, "conv_valu: 3x3/s1, width % 4 == 0 only");
    const int cp = (a.Cout + 1) / 2;
#define PF_VCASE(CP_) \
    if (cp == CP_) return rows == 1 ? launch_valu_cfg<CP_, 1, kValuKc>(a, B, s) : launch_valu_cfg<CP_, 2, kValuKc>(a, B, s);
    PF_VCASE(5) PF_VCASE(8) PF_VCASE(9) PF_VCASE(12) PF_VCASE(14) PF_VCASE(15) PF_VCASE(16)
#undef PF_VCASE
    return fail(PF_EUNSUPPORTED, "conv_valu: no kernel for %d output channels", a.Cout);
}

bool conv_valu_supports(int cout) {
    const int cp = (cout + 1) / 2;
    return cp == 5 || cp == 8 || cp == 9 || cp == 12 || cp == 14 || cp == 15 || cp == 16;
}

int valu_chunks(const int *src_ch, int n_src) {
    int n = 0;
    for (int j = 0; j < n_src; ++j) n += (src_ch[j] + kValuKc - 1) / kValuKc;
    return n;
}

size_t valu_packed_floats(const int *src_ch, int n_src, int cout) {
    return (size_t)valu_chunks(src_ch, n_src) * kValuKc * 9 * (2 * ((cout + 1) / 2));
}

// [chunk][channel in chunk][tap][cout padded to even]; K order = the input ranges in order, each padded to whole chunks
void pack_conv_weights_valu(const float *w, int cin, int cout, const int *src_ch, int n_src, float *out) {
    const int cpad = 2 * ((cout + 1) / 2);
    size_t o = 0;
    int c0 = 0;
    for (int j = 0; j < n_src; ++j) {
        for (int lc = 0; lc * kValuKc < src_ch[j]; ++lc)
            for (int c = 0; c < kValuKc; ++c)
                for (int tap = 0; tap < 9; ++tap)
                    for (int co = 0; co < cpad; ++co) {
                        const int cl = lc * kValuKc + c;
                        out[o++] = (co < cout && cl < src_ch[j]) ? w[((size_t)co * cin + c0 + cl) * 9 + tap] : 0.f;
                    }
        c0 += src_ch[j];
    }
}

}  // namespace pf
