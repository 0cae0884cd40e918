// Training step, forward pass on the packed-pair kernels (round 6; opt-in: option "train_forward_s4", training.forward_packed_pairs).
//
// The forward 3x3 stride-1 and 1x1 convolutions of a training step (models/bg/hardnet.py:16-25 under bg_model.py:73-89) are the
// same functions the inference path runs on conv_s4 (every fp32 operand as two round-to-nearest fp16 terms, three products on
// v_mfma_f32_16x16x32_f16, fp32 accumulation: conv_mfma.h).  What training adds:
//   * weights change every step: s4_pack_weights_dev_kernel packs them ON THE DEVICE from the fp32 arena (theta) into the block layout
//     pack_conv_weights_s4 (conv_s4.hip) produces on the host (checked end to end against float64 autograd:
//     tests/test_gpu_train.py::test_mini_network_forward_on_packed_pairs_vs_autograd) - with ONE fixed scale 2^12 instead of a per-conv 2^k
//     from max|w| (the kernel's 2^-k is a launch argument: a data-dependent scale would need the host to wait for the device).
//     |w| >= 16 does not fit fp16 then: the kernel stores NaN for such a weight, the loss of the step is NaN - loud, never a silently
//     clipped weight;
//   * activations are fp32 NCHW (BatchNorm, pooling, the weight gradients read them): s4_pack_act_kernel writes the packed-pair
//     SHADOW of every slice a producer finishes, for the tensors a packed-pair convolution reads; odd-width levels (50, 25 pixels at
//     800 x 800) get rows padded to a multiple of 4 with zero pad columns = the convolution's own zero padding (train_kernels.hip);
//   * outputs stay fp32 (pre-BatchNorm y): conv_s4's fp32 epilogue; the 3x3 layers run conv_s4_blocked_kernel (conv_s4_kernel.inc with
//     KACC: per-round partial sums, like the fp32 step) - the forward pass is then 0.81-0.95 x as far from float64 as torch-CPU fp32.
// Measured: profiles/r06_experiments.md (section 11).
#include "conv_mfma.h"
#include "conv_s4.h"
#include "train_kernels.h"

namespace pf {

namespace {
[[maybe_unused]] constexpr int kS4MaxEnt = 640;      // group entries of a conv (534 input channels in three ranges, padded to rounds of 8: < 160)

[[maybe_unused]] __device__ __forceinline__ unsigned short s4w_bits(split_t h) { return __builtin_bit_cast(unsigned short, h); }
}  // namespace

// one 256-thread workgroup = 256 (cout tile, block, lane) units of one conv; a unit = the lane's 8 values of both terms (2 x 16 B)
__global__ __launch_bounds__(256) void s4_pack_weights_dev_kernel(const float *theta, float *arena, S4WBatch pb) {
#if defined(__HIP_DEVICE_COMPILE__)
    int j = 0;
    while (j + 1 < pb.n && (int)blockIdx.x >= pb.job[j + 1].first_block) ++j;      // uniform
    const S4WJob &jb = pb.job[j];
    __shared__ short ent_c0[kS4MaxEnt], ent_lo[kS4MaxEnt], ent_hi[kS4MaxEnt];
    __shared__ int n_ent_s;
    const int per = jb.ks == 3 ? 2 : 8;
    if (threadIdx.x == 0) {      // the entry table of pack_conv_weights_s4_ex (conv_s4.hip), a few dozen entries
        int n = 0, c0 = 0;
        for (int s = 0; s < jb.n_src; ++s) {
            const int g0 = jb.choff[s] / 4, g1 = (jb.choff[s] + jb.ch[s] + 3) / 4;
            for (int g = g0; g < g1 && n < kS4MaxEnt; ++g, ++n) {
                ent_c0[n] = (short)(c0 + 4 * g - jb.choff[s]);
                ent_lo[n] = (short)c0;
                ent_hi[n] = (short)(c0 + jb.ch[s]);
            }
            while (jb.pad && n % per != 0 && n < kS4MaxEnt) { ent_c0[n] = 0; ent_lo[n] = 0; ent_hi[n] = 0; ++n; }
            c0 += jb.ch[s];
        }
        n_ent_s = n;
    }
    __syncthreads();
    const int n_ent = n_ent_s, rounds = jb.rounds, nblocks = jb.nblocks, ntiles = (jb.cout + 15) / 16;
    const int unit = ((int)blockIdx.x - jb.first_block) * 256 + (int)threadIdx.x;
    if (unit >= ntiles * nblocks * 64) return;
    const int t = unit / (nblocks * 64), rem = unit - t * (nblocks * 64), kb = rem >> 6, lane = rem & 63;
    const int co = t * 16 + (lane & 15), g = lane >> 4;
    // block kb of a tile -> what lane group g multiplies: entries (e / 4 = 0, 1) and tap
    int ent[2], tap;
    if (jb.ks == 1) {
        ent[0] = kb * 8 + g * 2;
        ent[1] = ent[0] + 1;
        tap = 0;
    } else {
        const int q = kb / 9, r9 = kb - q * 9, nrem = rounds - 4 * q;          // group of four rounds, block inside it
        const int ninstr = 2 * (nrem < 4 ? nrem : 4);
        if (r9 < ninstr) {
            const int rd = 4 * q + (r9 >> 1), s = r9 & 1;
            const int tap3[2][4] = {{0, 1, 3, 4}, {6, 7, 2, 5}};                   // ky * 3 + kx of (instr, lane group): conv_s4.hip
            ent[0] = rd * 2;
            ent[1] = rd * 2 + 1;
            tap = tap3[s][g];
        } else {                                                                   // the collected ninth tap of the group's rounds
            ent[0] = 4 * q + g < rounds ? (4 * q + g) * 2 : -1;
            ent[1] = ent[0] < 0 ? -1 : ent[0] + 1;
            tap = 8;
        }
    }
    const int ks2 = jb.ks * jb.ks;
    const float *w = theta + jb.w_off;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int en = ent[e >> 2];
        v[e] = 0.f;
        if (co < jb.cout && en >= 0 && en < n_ent) {
            const int ci = ent_c0[en] + (e & 3);
            if (ci >= ent_lo[en] && ci < ent_hi[en]) {
                const float x = w[((size_t)co * jb.cin + ci) * ks2 + tap] * pb.scale;
                v[e] = fabsf(x) <= kSplitMaxAbs ? x : __builtin_nanf("");          // (never a silently clipped weight)
            }
        }
    }
    typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
    u16x8 hi8, mid8;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        split_x2 h, m;
        split_terms2(v[e], v[e + 1], h, m);
        hi8[e] = s4w_bits(h[0]); hi8[e + 1] = s4w_bits(h[1]);
        mid8[e] = s4w_bits(m[0]); mid8[e + 1] = s4w_bits(m[1]);
    }
    unsigned short *out = reinterpret_cast<unsigned short *>(arena + jb.out_off) + ((size_t)(t * nblocks + kb) * 2) * 512 + lane * 8;
    *reinterpret_cast<u16x8 *>(out) = hi8;
    *reinterpret_cast<u16x8 *>(out + 512) = mid8;
#endif
}

int launch_s4_pack_weights_dev(const float *theta, float *arena, const S4WJob *jobs, int n, float scale, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kS4WBatch) {
        S4WBatch pb;
        pb.n = n - i0 < kS4WBatch ? n - i0 : kS4WBatch;
        pb.scale = scale;
        int blocks = 0;
        for (int k = 0; k < pb.n; ++k) {
            pb.job[k] = jobs[i0 + k];
            pb.job[k].first_block = blocks;
            blocks += ((pb.job[k].cout + 15) / 16 * pb.job[k].nblocks * 64 + 255) / 256;
        }
        hipLaunchKernelGGL(s4_pack_weights_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, s, theta, arena, pb);
        PF_LAUNCH_CHECK("s4_pack_weights_dev_kernel");
    }
    return PF_OK;
}

// fills what the host needs of a job (rounds, blocks, arena bytes) from the conv's ranges; returns the packed floats
size_t s4_wjob_init(S4WJob &jb, size_t w_off, int cin, int cout, int ks, const S4Range *r, int n_src, int pad_sources) {
    jb.w_off = (long long)w_off;
    jb.cin = cin; jb.cout = cout; jb.ks = ks; jb.n_src = n_src; jb.pad = pad_sources;
    for (int j = 0; j < kConvMaxSrc; ++j) { jb.choff[j] = j < n_src ? r[j].choff : 0; jb.ch[j] = j < n_src ? r[j].ch : 0; }
    jb.rounds = s4_rounds(r, n_src, ks, pad_sources);
    jb.nblocks = ks == 3 ? s4_blocks_total(jb.rounds) : jb.rounds;
    jb.first_block = 0;
    return s4_packed_floats(r, n_src, cout, ks, pad_sources);
}

// ------------------------------------------------------------------------------------------------ activations: fp32 NCHW slice -> shadow
// Channels [c0, c1) of src [B][ctotal][H][W] (c0 even) -> dst [B][2 terms][ceil(ctotal / 4)][H][Wp][4 ch] fp16, Wp >= W a multiple of 4,
// zeros in the pad columns.  A thread = two neighbouring pixels of one channel group: 16 B per term where the slice owns the whole
// group (a slice that ends at the tensor's last channel owns the group's missing channels too: zeros), else the 4-B half it owns.
__global__ __launch_bounds__(256) void s4_pack_act_kernel(const float *src, int ctotal, int c0, int c1, int fill_lo, int fill_up, int H, int W, int Wp,
                                                          unsigned short *dst, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int pairs = H * (Wp / 2), pp = blockIdx.x * 256 + threadIdx.x;
    if (pp >= pairs) return;
    const int g = c0 / 4 + blockIdx.y, b = blockIdx.z, C4 = (ctotal + 3) / 4;
    const int y = pp / (Wp / 2), x = (pp - y * (Wp / 2)) * 2;
    const int c1o = c1 == ctotal ? (c1 + 3) / 4 * 4 : c1;              // owned channels end (zeros past the tensor's last channel)
    // fill_lo / fill_up: the half group in front of / behind the slice has no owner yet in this pass: zeros (see train_plan.hip)
    const bool lo = (4 * g >= c0 && 4 * g < c1o) || (fill_lo && g == c0 / 4), up = (4 * g + 2 >= c0 && 4 * g + 2 < c1o) || (fill_up && g == (c1 - 1) / 4);
    const size_t hw = (size_t)H * W;
    s4_f32x4 v[2];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * g + r;
            v[k][r] = (c >= c0 && c < c1 && x + k < W) ? src[((size_t)b * ctotal + c) * hw + (size_t)y * W + x + k] * scale : 0.f;
        }
    s4_h4 hi[2], mid[2];
    split_terms4(v[0], hi[0], mid[0]);
    split_terms4(v[1], hi[1], mid[1]);
    const size_t plane = (size_t)H * Wp * 4, term = (size_t)C4 * plane;      // in fp16 elements
    unsigned short *p = dst + ((size_t)b * 2 * C4 + g) * plane + ((size_t)y * Wp + x) * 4;
    typedef split_x2 h2;
    if (lo && up) {
        *reinterpret_cast<s4_h8 *>(p) = s4_join(hi[0], hi[1]);
        *reinterpret_cast<s4_h8 *>(p + term) = s4_join(mid[0], mid[1]);
    } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (lo) {
                *reinterpret_cast<h2 *>(p + k * 4) = h2{hi[k][0], hi[k][1]};
                *reinterpret_cast<h2 *>(p + k * 4 + term) = h2{mid[k][0], mid[k][1]};
            }
            if (up) {
                *reinterpret_cast<h2 *>(p + k * 4 + 2) = h2{hi[k][2], hi[k][3]};
                *reinterpret_cast<h2 *>(p + k * 4 + 2 + term) = h2{mid[k][2], mid[k][3]};
            }
        }
    }
#endif
}

int launch_s4_pack_act(const float *src, int B, int ctotal, int c0, int c1, int fill_lo, int fill_up, int H, int W, int Wp, void *dst, float scale,
                       hipStream_t s) {
    if ((c0 & 1) || (Wp & 3) || Wp < W || c1 <= c0) return fail(PF_EINVAL, "s4_pack_act: slice [%d, %d) of %d channels, width %d pitch %d", c0, c1, ctotal, W, Wp);
    const int groups = (c1 + 3) / 4 - c0 / 4, pairs = H * (Wp / 2);
    hipLaunchKernelGGL(s4_pack_act_kernel, dim3((unsigned)((pairs + 255) / 256), groups, B), dim3(256), 0, s, src, ctotal, c0, c1, fill_lo, fill_up, H, W,
                       Wp, reinterpret_cast<unsigned short *>(dst), scale);
    PF_LAUNCH_CHECK("s4_pack_act_kernel");
    return PF_OK;
}

}  // namespace pf
