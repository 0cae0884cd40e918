// Training-side kernels of the bg network (scope row f4): everything `BGModel.loss(...).backward()` + the optimiser
// step do in the reference (models/bg/bg_model.py:73-89, training/train.py:185-222) that the inference path does not:
//   train-mode BatchNorm (batch statistics, running-stat update, backward), the convolution gradients
//   (backward-weight on the fp32 matrix cores; backward-data reuses the forward MFMA kernel on flipped/transposed
//   weights, see train_plan.hip), average-pool / bilinear-upsample backward, bilinear-upsampled cross entropy with its
//   gradient, global-norm / value clipping and SGD with momentum and weight decay.
// All reductions are two-stage with fp64 partials in a fixed order: a step is bit-reproducible run to run.
#include <cstring>

#include "conv_epilogue.h"
#include "pf_prof.h"
#include "train_kernels.h"

namespace pf {

typedef float tr_f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ block reduce helper
template <int NV>
__device__ __forceinline__ void block_reduce_d(double (&v)[NV], double *smem /* [NV][4] */) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NV; ++k) smem[k * 4 + wave] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((smem[k * 4 + 0] + smem[k * 4 + 1]) + smem[k * 4 + 2]) + smem[k * 4 + 3];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ input: one-hot + depth
// bg_model.py:53-69: labels >= n_cls -> zero vector; depth channels = (d - mean) / std * mask, after the T one-hot groups
__global__ __launch_bounds__(256) void onehot_dense_kernel(const void *seg, int seg_i64, const float *depth, const uint8_t *mask,
                                                           float mean, float stdv, int B, int T, int n_cls, long long HW, float *x) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    if (i >= HW) return;
    const long long in = (long long)bt * HW + i;
    const long long lab = seg_i64 ? reinterpret_cast<const long long *>(seg)[in] : (long long)reinterpret_cast<const uint8_t *>(seg)[in];
    const int C = T * (n_cls + 1);
    float *xb = x + (long long)b * C * HW + i;
    for (int c = 0; c < n_cls; ++c) xb[(long long)(t * n_cls + c) * HW] = (lab == c) ? 1.f : 0.f;
    const float dn = (depth[in] - mean) / stdv;
    xb[(long long)(T * n_cls + t) * HW] = dn * (mask[in] ? 1.f : 0.f);
}

int launch_onehot_dense(const void *seg, int seg_i64, const float *depth, const uint8_t *mask, float mean, float stdv, int B, int T,
                        int n_cls, int H, int W, float *x, hipStream_t s) {
    const long long HW = (long long)H * W;
    hipLaunchKernelGGL(onehot_dense_kernel, dim3((unsigned)((HW + 255) / 256), B * T), dim3(256), 0, s, seg, seg_i64, depth, mask, mean,
                       stdv, B, T, n_cls, HW, x);
    PF_LAUNCH_CHECK("onehot_dense_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ weight packing on the device
// OIHW -> the fragment order of conv_mfma.hip::pack_conv_weights ([cout_block][chunk][kgroup][tap][nt][64 lanes]).
// transpose_flip = 0: forward weights.  transpose_flip = 1: the backward-data convolution of input range [c0, c0+ch):
//   Wd[co_d][ci_d][tap] = W[ci_d][c0 + co_d][ks2-1-tap]   (ci_d runs over the forward cout)
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *w, int cin_f, int cout_f, int ks2, int kc, int nt, int nchunks,
                                                           long long total, int transpose_flip, int c0, int ch, float *out) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    long long r = o;
    const int lane = (int)(r % 64); r /= 64;
    const int n = (int)(r % nt); r /= nt;
    const int tap = (int)(r % ks2); r /= ks2;
    const int kg = (int)(r % (kc / 4)); r /= (kc / 4);
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int cb = (int)r;
    const int co = (cb * nt + n) * 16 + (lane & 15), ci = chunk * kc + kg * 4 + (lane >> 4);
    float v = 0.f;
    if (!transpose_flip) {
        if (co < cout_f && ci < cin_f) v = w[((long long)co * cin_f + ci) * ks2 + tap];
    } else {
        if (co < ch && ci < cout_f) v = w[((long long)ci * cin_f + c0 + co) * ks2 + (ks2 - 1 - tap)];
    }
    out[o] = v;
}

int launch_pack_weights(const float *w, int cin_f, int cout_f, const ConvTiling &t, int transpose_flip, int c0, int ch, float *out,
                        hipStream_t s) {
    const long long total = (long long)t.packed_floats();
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, cin_f, cout_f, t.ks * t.ks, t.kc, t.nt,
                       t.nchunks, total, transpose_flip, c0, ch, out);
    PF_LAUNCH_CHECK("pack_weights_kernel");
    return PF_OK;
}

// OIHW -> the per-tile order of the LDS-DMA kernels (conv_dma.hip::pack_conv_weights_tiled: [cout tile][chunk][kgroup][tap][64 lanes],
// K order = the input ranges in order, each padded to whole chunks of kc channels).  Same two modes as above; the
// backward-data convolution has ONE input range (the forward cout channels of dy).
struct TiledPackArgs {
    int cstart[kConvMaxSrc + 1];   // first conv input channel of each range ([n_src] = cin)
    int chunk0[kConvMaxSrc + 1];   // first chunk of each range ([n_src] = nchunks)
    int n_src;
};
__global__ __launch_bounds__(256) void pack_weights_tiled_kernel(const float *w, int cin_f, int cout_f, int ks2, int kc, TiledPackArgs pa,
                                                                 long long total, int transpose_flip, int c0, int ch, float *out) {
    const long long o = (long long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int nchunks = pa.chunk0[pa.n_src];
    long long r = o;
    const int lane = (int)(r % 64); r /= 64;
    const int tap = (int)(r % ks2); r /= ks2;
    const int kg = (int)(r % (kc / 4)); r /= (kc / 4);
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int t = (int)r;
    int j = 0;
    while (j + 1 < pa.n_src && chunk >= pa.chunk0[j + 1]) ++j;
    const int cl = (chunk - pa.chunk0[j]) * kc + kg * 4 + (lane >> 4), nch = pa.cstart[j + 1] - pa.cstart[j];
    const int co = t * 16 + (lane & 15), ci = pa.cstart[j] + cl;
    float v = 0.f;
    if (cl < nch) {
        if (!transpose_flip) {
            if (co < cout_f) v = w[((long long)co * cin_f + ci) * ks2 + tap];
        } else {
            if (co < ch) v = w[((long long)ci * cin_f + c0 + co) * ks2 + (ks2 - 1 - tap)];   // ci runs over the forward cout
        }
    }
    out[o] = v;
}
size_t tiled_packed_floats(const int *src_ch, int n_src, int cout, int ks, int stride) {
    return (size_t)((cout + 15) / 16) * dma_chunks(src_ch, n_src, ks, stride) * (dma_kc(ks, stride) / 4) * ks * ks * 64;
}
int launch_pack_weights_tiled(const float *w, int cin_f, int cout_f, int ks, int stride, const int *src_ch, int n_src, int transpose_flip,
                              int c0, int ch, float *out, hipStream_t s) {
    TiledPackArgs pa;
    const int kc = dma_kc(ks, stride);
    pa.n_src = n_src;
    pa.cstart[0] = 0;
    pa.chunk0[0] = 0;
    for (int j = 0; j < kConvMaxSrc; ++j) {
        pa.cstart[j + 1] = pa.cstart[j] + (j < n_src ? src_ch[j] : 0);
        pa.chunk0[j + 1] = pa.chunk0[j] + (j < n_src ? (src_ch[j] + kc - 1) / kc : 0);
    }
    const long long total = (long long)tiled_packed_floats(src_ch, n_src, transpose_flip ? ch : cout_f, ks, stride);
    hipLaunchKernelGGL(pack_weights_tiled_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, cin_f, cout_f, ks * ks, kc, pa, total,
                       transpose_flip, c0, ch, out);
    PF_LAUNCH_CHECK("pack_weights_tiled_kernel");
    return PF_OK;
}

// All tiled packings of a training step in a handful of launches (train_plan.hip collects the jobs: one per forward conv, one per
// backward-data conv): a step used to spend 190 launches of ~4.6 us on them (0.87 ms of a 20.7 ms step).  The jobs travel as kernel
// arguments, kPackBatch per launch; a workgroup finds its job by a scalar walk over the block prefix
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const float *theta, float *arena, PackBatch pb) {
    int j = 0;
    while (j + 1 < pb.n && (int)blockIdx.x >= pb.job[j + 1].block0) ++j;
    const PackJob &q = pb.job[j];
    const long long o = (long long)((int)blockIdx.x - q.block0) * 256 + threadIdx.x;
    if (o >= q.total) return;
    const float *w = theta + q.w_off;
    const int nchunks = q.chunk0[q.n_src];
    long long r = o;
    const int lane = (int)(r % 64); r /= 64;
    const int tap = (int)(r % q.ks2); r /= q.ks2;
    const int kg = (int)(r % (q.kc / 4)); r /= (q.kc / 4);
    const int chunk = (int)(r % nchunks); r /= nchunks;
    const int t = (int)r;
    int jj = 0;
    while (jj + 1 < q.n_src && chunk >= q.chunk0[jj + 1]) ++jj;
    const int cl = (chunk - q.chunk0[jj]) * q.kc + kg * 4 + (lane >> 4), nch = q.cstart[jj + 1] - q.cstart[jj];
    const int co = t * 16 + (lane & 15), ci = q.cstart[jj] + cl;
    float v = 0.f;
    if (cl < nch) {
        if (!q.tflip) {
            if (co < q.cout_f) v = w[((long long)co * q.cin_f + ci) * q.ks2 + tap];
        } else {
            if (co < q.ch) v = w[((long long)ci * q.cin_f + q.c0 + co) * q.ks2 + (q.ks2 - 1 - tap)];   // ci runs over the forward cout
        }
    }
    arena[q.out_off + o] = v;
}
void pack_job_fill(PackJob &q, size_t w_off, int cin_f, int cout_f, int ks, int stride, const int *src_ch, int n_src, int transpose_flip, int c0,
                   int ch, size_t out_off) {
    const int kc = dma_kc(ks, stride);
    q.w_off = (long long)w_off; q.out_off = (long long)out_off;
    q.cin_f = cin_f; q.cout_f = cout_f; q.ks2 = ks * ks; q.kc = kc; q.tflip = transpose_flip; q.c0 = c0; q.ch = ch; q.n_src = n_src;
    q.cstart[0] = 0; q.chunk0[0] = 0;
    for (int j = 0; j < kConvMaxSrc; ++j) {
        q.cstart[j + 1] = q.cstart[j] + (j < n_src ? src_ch[j] : 0);
        q.chunk0[j + 1] = q.chunk0[j] + (j < n_src ? (src_ch[j] + kc - 1) / kc : 0);
    }
    q.total = (long long)tiled_packed_floats(src_ch, n_src, transpose_flip ? ch : cout_f, ks, stride);
    q.block0 = 0;
}
int launch_pack_weights_batch(const float *theta, float *arena, const PackJob *jobs, int n, hipStream_t s) {
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        PackBatch pb;
        pb.n = n - i0 < kPackBatch ? n - i0 : kPackBatch;
        int blocks = 0;
        for (int k = 0; k < pb.n; ++k) {
            pb.job[k] = jobs[i0 + k];
            pb.job[k].block0 = blocks;
            blocks += (int)((jobs[i0 + k].total + 255) / 256);
        }
        if (blocks == 0) continue;
        hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, s, theta, arena, pb);
        PF_LAUNCH_CHECK("pack_weights_batch_kernel");
    }
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ BatchNorm (training mode)
// partial[c][slab][k]: k = 0 sum(y), 1 sum(y^2)  (stats)   or   0 sum(g'), 1 sum(g' * xhat)  (backward)
constexpr int kBnSlabs = 64;

// y of an odd-width layer is the tiled convolution's own output: rows of Wp >= W floats (pad columns hold garbage and are never
// read).  Element p of plane (b, c) of a [..][H][W] tensor stored with row pitch Wp:
__device__ __forceinline__ long long pitched(long long plane, long long p, long long HW, int W, int Wp) {
    return W == Wp ? plane * HW + p : (plane * (HW / W) + p / W) * Wp + p % W;
}

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float *y, int B, int C, long long HW, int W, int Wp, double *partial) {
    const int c = blockIdx.x, slab = blockIdx.y;
    __shared__ double sm[2 * 4];
    double v[2] = {0.0, 0.0};
    const long long per = (long long)B * HW;
    for (long long i = (long long)slab * 256 + threadIdx.x; i < per; i += (long long)kBnSlabs * 256) {
        const long long b = i / HW, p = i - b * HW;
        const double x = (double)y[pitched((long long)b * C + c, p, HW, W, Wp)];
        v[0] += x;
        v[1] += x * x;
    }
    block_reduce_d<2>(v, sm);
    if (threadIdx.x == 0) {
        partial[((long long)c * kBnSlabs + slab) * 2 + 0] = v[0];
        partial[((long long)c * kBnSlabs + slab) * 2 + 1] = v[1];
    }
}

// The second pass of a BatchNorm (forward: statistics -> scale / shift; backward: sums -> dy) needs the channel's 64 slab
// partials added up.  Every WAVE of the second-pass kernels does that itself - one partial pair per lane, a butterfly of
// shuffles (a fixed order: every wave gets the same bits) - instead of a tiny kernel between the passes: 138 launches of ~5 us
// fewer per training step, on the critical path of the backward chain.  One designated thread per channel (block x = 0 of
// sample 0) writes what the step keeps (mean / invstd / running statistics; dgamma / dbeta).
static_assert(kBnSlabs == 64, "one slab partial per lane");
__device__ __forceinline__ void bn_wave_sums(const double *partial, int c, double &s, double &q) {
    const double *p = partial + ((long long)c * kBnSlabs + (threadIdx.x & 63)) * 2;
    s = p[0];
    q = p[1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
    }
}
struct BnStats {        // forward statistics of one layer (kernel argument)
    const double *partial;
    double n;
    float eps, momentum;
    float *mean, *invstd, *running_mean, *running_var;   // running_*: null = leave them
};
// mean / invstd of the batch (+ running-stat update by the designated thread: nn.BatchNorm2d, momentum 0.1, unbiased variance
// in the running stat).  Called by all lanes of a wave
__device__ __forceinline__ void bn_finish_stats(const BnStats &st, int c, bool writer, float &mean_f, float &invstd_f) {
    double s, q;
    bn_wave_sums(st.partial, c, s, q);
    const double m = s / st.n;
    double var = q / st.n - m * m;
    var = var < 0.0 ? 0.0 : var;
    mean_f = (float)m;
    invstd_f = (float)(1.0 / sqrt(var + (double)st.eps));
    if (writer) {
        st.mean[c] = mean_f;
        st.invstd[c] = invstd_f;
        if (st.running_mean) {
            const double unbiased = st.n > 1.0 ? var * st.n / (st.n - 1.0) : var;
            st.running_mean[c] = (float)((1.0 - st.momentum) * (double)st.running_mean[c] + st.momentum * m);
            st.running_var[c] = (float)((1.0 - st.momentum) * (double)st.running_var[c] + st.momentum * unbiased);
        }
    }
}

// the channel's sums for the second backward pass (dbeta = sum g', dgamma = sum g' xhat, rounded to fp32 as the gradient arenas
// hold them); the designated thread adds them to the arenas (zeroed once per step)
__device__ __forceinline__ void bn_finish_bwd(const double *partial, int c, bool writer, float *dgamma, float *dbeta, float &s_f, float &q_f) {
    double s, q;
    bn_wave_sums(partial, c, s, q);
    s_f = (float)s;
    q_f = (float)q;
    if (writer) {
        dbeta[c] += s_f;
        dgamma[c] += q_f;
    }
}

// z = relu(gamma * (y - mean) * invstd + beta) into channel slice [choff, choff + C) of the destination tensor
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float *y, BnStats st, const float *gamma,
                                                            const float *beta, int C, long long HW, int W, int Wp, float *dst, int dst_ctotal,
                                                            int dst_choff, int relu) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    float mean_c, invstd_c;
    bn_finish_stats(st, c, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, mean_c, invstd_c);
    if (i4 >= HW) return;
    const float sc = gamma[c] * invstd_c, sh = beta[c] - mean_c * sc;
    float *dp = dst + ((long long)b * dst_ctotal + dst_choff + c) * HW + i4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i4 + k < HW) {
            // (y - mean) * invstd * gamma + beta, written as one scale/shift like ATen's batch_norm CPU/CUDA transforms
            float v = __builtin_fmaf(y[pitched(bc, i4 + k, HW, W, Wp)], sc, sh);      // (the backward pass redoes exactly this to know where the ReLU cut)
            dp[k] = relu ? fmaxf(v, 0.f) : v;
        }
}

// ---- the same four passes with 16-B accesses and no per-element index division (planes of HW % 4 == 0 elements): a block
// walks the planes of its channel sample by sample, a thread adds its four values in fp32 and accumulates those in fp64
__global__ __launch_bounds__(256) void bn_stats_partial4_kernel(const float *y, int B, int C, int HW4, double *partial) {
    const int c = blockIdx.x, slab = blockIdx.y;
    __shared__ double sm[2 * 4];
    double v[2] = {0.0, 0.0};
    for (int b = 0; b < B; ++b) {
        const tr_f32x4 *p = reinterpret_cast<const tr_f32x4 *>(y) + ((long long)b * C + c) * HW4;
        for (int i = slab * 256 + threadIdx.x; i < HW4; i += kBnSlabs * 256) {
            const tr_f32x4 x = p[i];
            v[0] += (double)((x[0] + x[1]) + (x[2] + x[3]));
            v[1] += (double)((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]));
        }
    }
    block_reduce_d<2>(v, sm);
    if (threadIdx.x == 0) {
        partial[((long long)c * kBnSlabs + slab) * 2 + 0] = v[0];
        partial[((long long)c * kBnSlabs + slab) * 2 + 1] = v[1];
    }
}
__global__ __launch_bounds__(256) void bn_apply_relu4_kernel(const float *y, BnStats st, const float *gamma,
                                                             const float *beta, int C, int HW4, float *dst, int dst_ctotal, int dst_choff, int relu) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float mean_c, invstd_c;
    bn_finish_stats(st, c, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, mean_c, invstd_c);
    if (i >= HW4) return;
    const float sc = gamma[c] * invstd_c, sh = beta[c] - mean_c * sc;
    tr_f32x4 v = reinterpret_cast<const tr_f32x4 *>(y)[(long long)bc * HW4 + i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = __builtin_fmaf(v[k], sc, sh);
        if (relu) v[k] = fmaxf(v[k], 0.f);
    }
    reinterpret_cast<tr_f32x4 *>(dst)[((long long)b * dst_ctotal + dst_choff + c) * HW4 + i] = v;
}
// Where the ReLU cut (z = relu(fma(y, sc, sh)) <= 0) is recomputed from y, which these kernels read anyway, with the forward
// pass's own instruction - the stored activation is not read again (2 of the 7 tensor passes of a BatchNorm backward)
__global__ __launch_bounds__(256) void bn_bwd_partial4_kernel(const float *g, const float *gamma, const float *beta, int t_ctotal, int choff,
                                                              const float *y, const float *mean, const float *invstd, int B, int C, int HW4,
                                                              int relu, double *partial) {
    const int c = blockIdx.x, slab = blockIdx.y;
    __shared__ double sm[2 * 4];
    double v[2] = {0.0, 0.0};
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    for (int b = 0; b < B; ++b) {
        const long long ti = ((long long)b * t_ctotal + choff + c) * HW4, yi = ((long long)b * C + c) * HW4;
        for (int i = slab * 256 + threadIdx.x; i < HW4; i += kBnSlabs * 256) {
            const tr_f32x4 g4 = reinterpret_cast<const tr_f32x4 *>(g)[ti + i], y4 = reinterpret_cast<const tr_f32x4 *>(y)[yi + i];
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float gp = (!relu || __builtin_fmaf(y4[k], sc, sh) > 0.f) ? g4[k] : 0.f;
                s0 += gp;
                s1 += gp * ((y4[k] - mu) * is);
            }
            v[0] += (double)s0;
            v[1] += (double)s1;
        }
    }
    block_reduce_d<2>(v, sm);
    if (threadIdx.x == 0) {
        partial[((long long)c * kBnSlabs + slab) * 2 + 0] = v[0];
        partial[((long long)c * kBnSlabs + slab) * 2 + 1] = v[1];
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float *g, const float *beta, int t_ctotal, int choff, const float *y,
                                                            const float *mean, const float *invstd, const float *gamma, const double *partial,
                                                            float *dgamma, float *dbeta, int B, int C, int HW4, int relu, float *dy) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    float sum_g, sum_gx;
    bn_finish_bwd(partial, c, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, dgamma, dbeta, sum_g, sum_gx);
    if (i >= HW4) return;
    const float n = (float)((double)B * 4.0 * (double)HW4);
    const long long ti = ((long long)b * t_ctotal + choff + c) * HW4 + i;
    const tr_f32x4 g4 = reinterpret_cast<const tr_f32x4 *>(g)[ti], y4 = reinterpret_cast<const tr_f32x4 *>(y)[(long long)bc * HW4 + i];
    const float mu = mean[c], is = invstd[c], ga = gamma[c], s0 = sum_g / n, s1 = sum_gx / n;
    const float sc = ga * is, sh = beta[c] - mu * sc;
    tr_f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float gp = (!relu || __builtin_fmaf(y4[k], sc, sh) > 0.f) ? g4[k] : 0.f;
        const float xh = (y4[k] - mu) * is;
        o[k] = ga * is * (gp - s0 - xh * s1);       // the scalar kernel's expression, term for term
    }
    reinterpret_cast<tr_f32x4 *>(dy)[(long long)bc * HW4 + i] = o;
}

int launch_bn_forward(const float *y, int B, int C, int H, int W, float eps, float momentum, const float *gamma, const float *beta,
                      float *running_mean, float *running_var, float *mean, float *invstd, double *partial, float *dst, int dst_ctotal,
                      int dst_choff, int relu, int y_pitch, hipStream_t s) {
    const long long HW = (long long)H * W;
    const int Wp = y_pitch > 0 ? y_pitch : W;
    const bool vec = (HW & 3) == 0 && HW / 4 < (1ll << 31) && Wp == W;
    if (vec) hipLaunchKernelGGL(bn_stats_partial4_kernel, dim3(C, kBnSlabs), dim3(256), 0, s, y, B, C, (int)(HW / 4), partial);
    else hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, kBnSlabs), dim3(256), 0, s, y, B, C, HW, W, Wp, partial);
    const BnStats st{partial, (double)B * HW, eps, momentum, mean, invstd, running_mean, running_var};
    if (vec)
        hipLaunchKernelGGL(bn_apply_relu4_kernel, dim3((unsigned)((HW / 4 + 255) / 256), B * C), dim3(256), 0, s, y, st, gamma, beta, C,
                           (int)(HW / 4), dst, dst_ctotal, dst_choff, relu);
    else
        hipLaunchKernelGGL(bn_apply_relu_kernel, dim3((unsigned)((HW / 4 + 256) / 256), B * C), dim3(256), 0, s, y, st, gamma, beta, C, HW,
                           W, Wp, dst, dst_ctotal, dst_choff, relu);
    PF_LAUNCH_CHECK("bn_forward");
    return PF_OK;
}
size_t bn_partial_doubles(int C) { return (size_t)C * kBnSlabs * 2; }

// backward through ReLU + BN:  g' = g * [z > 0];  dbeta = sum g';  dgamma = sum g' * xhat;
//                              dy = gamma * invstd * (g' - dbeta / N - xhat * dgamma / N)
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const float *g, const float *gamma, const float *beta, int t_ctotal, int choff,
                                                             const float *y, const float *mean, const float *invstd, int B, int C, long long HW,
                                                             int W, int Wp, int relu, double *partial) {
    const int c = blockIdx.x, slab = blockIdx.y;
    __shared__ double sm[2 * 4];
    double v[2] = {0.0, 0.0};
    const long long per = (long long)B * HW;
    const float mu = mean[c], is = invstd[c];
    const float sc = gamma[c] * is, sh = beta[c] - mu * sc;
    for (long long i = (long long)slab * 256 + threadIdx.x; i < per; i += (long long)kBnSlabs * 256) {
        const long long b = i / HW, p = i - b * HW;
        const long long ti = ((long long)b * t_ctotal + choff + c) * HW + p;
        const float yv = y[pitched((long long)b * C + c, p, HW, W, Wp)];
        const float gp = (!relu || __builtin_fmaf(yv, sc, sh) > 0.f) ? g[ti] : 0.f;
        const float xh = (yv - mu) * is;
        v[0] += (double)gp;
        v[1] += (double)gp * (double)xh;
    }
    block_reduce_d<2>(v, sm);
    if (threadIdx.x == 0) {
        partial[((long long)c * kBnSlabs + slab) * 2 + 0] = v[0];
        partial[((long long)c * kBnSlabs + slab) * 2 + 1] = v[1];
    }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float *g, const float *beta, int t_ctotal, int choff, const float *y,
                                                           const float *mean, const float *invstd, const float *gamma, const double *partial,
                                                           float *dgamma, float *dbeta, int B, int C, long long HW, int relu, float *dy) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float sum_g, sum_gx;
    bn_finish_bwd(partial, c, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, dgamma, dbeta, sum_g, sum_gx);
    if (i >= HW) return;
    const float n = (float)((double)B * (double)HW);
    const long long ti = ((long long)b * t_ctotal + choff + c) * HW + i;
    const float yv = y[(long long)bc * HW + i], sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
    const float gp = (!relu || __builtin_fmaf(yv, sc, sh) > 0.f) ? g[ti] : 0.f;
    const float xh = (yv - mean[c]) * invstd[c];
    dy[(long long)bc * HW + i] = gamma[c] * invstd[c] * (gp - sum_g / n - xh * sum_gx / n);
}

// the same, dy written with a row pitch of Wp >= W floats and zeros in the pad columns: the layout the tiled convolutions of an
// odd-width level read (launch_pad_gather's, without the copy)
__global__ __launch_bounds__(256) void bn_bwd_apply_pitch_kernel(const float *g, const float *beta, int t_ctotal, int choff, const float *y,
                                                                 const float *mean, const float *invstd, const float *gamma, const double *partial,
                                                                 float *dgamma, float *dbeta, int B, int C, int H, int W, int Wp, int relu, int y_pitched,
                                                                 float *dy) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int ip = blockIdx.x * 256 + threadIdx.x;
    float sum_g, sum_gx;
    bn_finish_bwd(partial, c, blockIdx.x == 0 && b == 0 && threadIdx.x == 0, dgamma, dbeta, sum_g, sum_gx);
    if (ip >= H * Wp) return;
    const int yy = ip / Wp, x = ip - yy * Wp;
    float o = 0.f;
    if (x < W) {
        const long long HW = (long long)H * W, i = (long long)yy * W + x;
        const float n = (float)((double)B * (double)HW);
        const long long ti = ((long long)b * t_ctotal + choff + c) * HW + i;
        const float yv = y[y_pitched ? (long long)bc * H * Wp + ip : (long long)bc * HW + i], sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
        const float gp = (!relu || __builtin_fmaf(yv, sc, sh) > 0.f) ? g[ti] : 0.f;
        const float xh = (yv - mean[c]) * invstd[c];
        o = gamma[c] * invstd[c] * (gp - sum_g / n - xh * sum_gx / n);
    }
    dy[(long long)bc * H * Wp + ip] = o;
}

int launch_bn_backward(const float *g, int t_ctotal, int choff, const float *y, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int B, int C, int H, int W, int relu, float *dgamma, float *dbeta, double *partial, float *dy,
                       int dy_pitch, int y_pitch, hipStream_t s) {
    const long long HW = (long long)H * W;
    const int Wyp = y_pitch > 0 ? y_pitch : W;
    if (Wyp != W && dy_pitch != Wyp) return fail(PF_EINVAL, "bn_backward: a pitched y needs dy with the same pitch");
    const bool vec = (HW & 3) == 0 && HW / 4 < (1ll << 31) && Wyp == W;
    if (vec)
        hipLaunchKernelGGL(bn_bwd_partial4_kernel, dim3(C, kBnSlabs), dim3(256), 0, s, g, gamma, beta, t_ctotal, choff, y, mean, invstd, B, C,
                           (int)(HW / 4), relu, partial);
    else
        hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(C, kBnSlabs), dim3(256), 0, s, g, gamma, beta, t_ctotal, choff, y, mean, invstd, B, C, HW, W, Wyp,
                           relu, partial);
    if (dy_pitch > 0 && dy_pitch != W)
        hipLaunchKernelGGL(bn_bwd_apply_pitch_kernel, dim3((unsigned)((H * dy_pitch + 255) / 256), B * C), dim3(256), 0, s, g, beta, t_ctotal, choff, y, mean,
                           invstd, gamma, partial, dgamma, dbeta, B, C, H, W, dy_pitch, relu, Wyp != W, dy);
    else if (vec)
        hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3((unsigned)((HW / 4 + 255) / 256), B * C), dim3(256), 0, s, g, beta, t_ctotal, choff, y, mean,
                           invstd, gamma, partial, dgamma, dbeta, B, C, (int)(HW / 4), relu, dy);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)((HW + 255) / 256), B * C), dim3(256), 0, s, g, beta, t_ctotal, choff, y, mean, invstd,
                           gamma, partial, dgamma, dbeta, B, C, HW, relu, dy);
    PF_LAUNCH_CHECK("bn_backward");
    return PF_OK;
}

// bias gradient of a conv without BN (finalConv): dbias[c] += sum over (b, pixels) of g; also copies g -> dy (contiguous).
// Grid (C, kBnSlabs) with fp64 partials reduced in slab order (one block per channel took 0.49 ms for the 11 logit planes)
__global__ __launch_bounds__(256) void bias_bwd_kernel(const float *g, int t_ctotal, int choff, int B, int C, long long HW, int W, int Wp,
                                                       double *partial, float *dy) {
    const int c = blockIdx.x, slab = blockIdx.y;
    __shared__ double sm[4];
    double v[1] = {0.0};
    for (int b = 0; b < B; ++b) {
        const float *gp = g + ((long long)b * t_ctotal + choff + c) * HW;
        float *dp = dy + ((long long)b * C + c) * (HW / W) * Wp;      // rows of Wp >= W floats (pad columns: zeroed by the launcher)
        for (long long p = (long long)slab * 256 + threadIdx.x; p < HW; p += (long long)kBnSlabs * 256) {
            const float x = gp[p];
            dp[W == Wp ? p : (p / W) * Wp + p % W] = x;
            v[0] += (double)x;
        }
    }
    block_reduce_d<1>(v, sm);
    if (threadIdx.x == 0) partial[(long long)c * kBnSlabs + slab] = v[0];
}
__global__ void bias_bwd_final_kernel(const double *partial, int C, float *dbias) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int k = 0; k < kBnSlabs; ++k) s += partial[(long long)c * kBnSlabs + k];
    dbias[c] += (float)s;
}
int launch_bias_backward(const float *g, int t_ctotal, int choff, int B, int C, int H, int W, float *dbias, double *partial, float *dy,
                         int dy_pitch, hipStream_t s) {
    const int Wp = dy_pitch > 0 ? dy_pitch : W;
    if (Wp != W) {
        int rc = launch_zero_fill(dy, (size_t)B * C * H * Wp * sizeof(float), s);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(bias_bwd_kernel, dim3(C, kBnSlabs), dim3(256), 0, s, g, t_ctotal, choff, B, C, (long long)H * W, W, Wp, partial, dy);
    hipLaunchKernelGGL(bias_bwd_final_kernel, dim3((C + 63) / 64), dim3(64), 0, s, partial, C, dbias);
    PF_LAUNCH_CHECK("bias_bwd_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ conv backward-weight
// dW[co][ci][tap] = sum over (b, oy, ox) of dy[b][co][oy][ox] * x[b][ci][oy*S + ky - P][ox*S + kx - P]
// GEMM on v_mfma_f32_16x16x4_f32: M = 16 couts, N = 16 cins, K = pixels (4 per instruction).  Workgroup = (cout tile,
// cin tile, slab of output rows); its 4 waves take the 16-pixel chunks of the slab's rows round-robin, keep the k*k
// tap accumulators in registers and are combined through LDS; partial sums go to partial[slab][co][ci][tap] and are
// added up by wgrad_reduce_kernel in slab order (no atomics: deterministic).
template <int KS, int STRIDE>
__global__ __launch_bounds__(256) void wgrad_partial_kernel(ConvArgs a, const float *dy, int B, int slabs, float *partial) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS2 = KS * KS, P = KS / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int co = blockIdx.x * 16 + (lane & 15), ci = blockIdx.y * 16 + (lane & 15), kq = lane >> 4;
    const int slab = blockIdx.z;
    // this lane's input channel -> (source tensor, channel inside it)
    const float *xsrc = nullptr;
    long long xplane0 = 0;
    int x_ctotal = 0;
    if (ci < a.Cin) {
        int sidx = 0;
        while (sidx + 1 < a.n_src && ci >= a.src_cstart[sidx + 1]) ++sidx;
        xsrc = a.src[sidx];
        x_ctotal = a.src_ctotal[sidx];
        xplane0 = a.src_choff[sidx] + (ci - a.src_cstart[sidx]);
    }
    const long long in_plane = (long long)a.Hin * a.Win, out_plane = (long long)a.Hout * a.Wout;
    tr_f32x4 acc[KS2];
#pragma unroll
    for (int t = 0; t < KS2; ++t) acc[t] = tr_f32x4{0.f, 0.f, 0.f, 0.f};
    const int chunks_per_row = (a.Wout + 15) / 16;
    const long long rows = (long long)B * a.Hout;
    int work = 0;
    for (long long row = slab; row < rows; row += slabs) {
        const int b = (int)(row / a.Hout), oy = (int)(row - (long long)b * a.Hout);
        const float *dyp = (co < a.Cout) ? dy + ((long long)b * a.Cout + co) * out_plane + (long long)oy * a.Wout : nullptr;
        const float *xp = xsrc ? xsrc + ((long long)b * x_ctotal + xplane0) * in_plane : nullptr;
        for (int ch = 0; ch < chunks_per_row; ++ch, ++work) {
            if ((work & 3) != wave) continue;
            const int ox0 = ch * 16 + kq * 4;
            float av[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = (dyp && ox0 + j < a.Wout) ? dyp[ox0 + j] : 0.f;
#pragma unroll
            for (int t = 0; t < KS2; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                const int iy = oy * STRIDE + ky - P;
                float bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ix = (ox0 + j) * STRIDE + kx - P;
                    bv[j] = (xp && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && ox0 + j < a.Wout) ? xp[(long long)iy * a.Win + ix] : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[t], 0, 0, 0);
            }
        }
    }
    // combine the 4 waves: D fragment = rows m = 4*(lane>>4)+i (cout), column n = lane&15 (cin)
    __shared__ float red[4][KS2][4][64];
#pragma unroll
    for (int t = 0; t < KS2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][t][i][lane] = acc[t][i];
    __syncthreads();
    const int ci_pad = gridDim.y * 16, co_pad = gridDim.x * 16;
    for (int e = threadIdx.x; e < KS2 * 4 * 64; e += 256) {
        const int l = e & 63, i = (e >> 6) & 3, t = e >> 8;
        const float v = ((red[0][t][i][l] + red[1][t][i][l]) + red[2][t][i][l]) + red[3][t][i][l];
        const int m = blockIdx.x * 16 + 4 * (l >> 4) + i, n = blockIdx.y * 16 + (l & 15);
        partial[(((long long)slab * co_pad + m) * ci_pad + n) * KS2 + t] = v;
    }
#endif
}

// ---- the LDS-tiled form (input and output widths multiples of 4): round 2's kernel above gathered both operands
// straight from memory, 16 B per lane from 16 different planes per request, and every (cout tile, cin tile) pair
// re-read its operands: 12.8 TFLOP/s over the 70 layers of a B = 8, 800 x 800 step (tools/bench_train.py).  Here a
// workgroup owns 32 couts x 16 cins and walks 64-pixel row segments: dy [32][64] and the x rows the taps need
// [16][KS][64 S + 8] are loaded with coalesced 16-B loads one segment AHEAD (registers), parked in LDS, and the 4 waves
// multiply 16 pixels each for all k*k taps from there (pitches == 4 mod 32 floats: at most 2-way bank conflicts on
// the one-float-per-lane fragment reads).  Partial sums per slab as before, same fixed-order reduction.
// R = output rows per work item (round 4).  With R = 1 every x row is staged three times by a 3x3 layer (once per tap row of the
// three output rows it feeds); R = 2 stages KS + 1 rows for two output rows: 2 x rows per output row instead of 3, a fifth
// fewer bytes per multiply-add (the kernel is bound by its L2 -> LDS bytes)
template <int KS, int STRIDE, int R>
struct WgCfg {
    static constexpr int KS2 = KS * KS, P = KS / 2, TW = 64;
    static constexpr int XR = KS + (R - 1) * STRIDE;                // staged x rows
    static constexpr int XW = TW * STRIDE + 8;                      // staged x columns: [x0 S - 4, x0 S + 64 S + 4)
    static constexpr int XCP = (XR * XW + 27) / 32 * 32 + 4;        // channel pitch == 4 (mod 32), >= XR * XW
    static constexpr int DP = TW + 4;                               // dy row pitch: 68 == 4 (mod 32)
    static constexpr int NDY = R * 32 * (TW / 4), NX = 16 * XR * (XW / 4);   // float4 loads per item
    static constexpr int ITD = NDY / 256, ITX = (NX + 255) / 256;
    static constexpr int LDS_FLOATS = R * 32 * DP + 16 * XCP;
    static_assert(XCP >= XR * XW && XCP % 32 == 4 && NDY % 256 == 0, "wgrad tile geometry");
};

template <int KS, int STRIDE, int R>
__global__ __launch_bounds__(256) void wgrad_tiled_kernel(ConvArgs a, const float *dy, int B, int slabs, float *partial) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = WgCfg<KS, STRIDE, R>;
    __shared__ __attribute__((aligned(16))) float lds[C::LDS_FLOATS > 3 * 4 * 256 ? C::LDS_FLOATS : 3 * 4 * 256];
    float *dy_s = lds, *x_s = lds + R * 32 * C::DP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 16, slab = blockIdx.z;
    const long long in_plane = (long long)a.Hin * a.Win, out_plane = (long long)a.Hout * a.Wout;
    const int chunks = (a.Wout + C::TW - 1) / C::TW;
    const int row_groups = (a.Hout + R - 1) / R;
    const long long items = (long long)B * row_groups * chunks;

    // this thread's x loads: (channel, tap row, 16-B column) of the staged window -> plane base of the channel (batch 0)
    const float *xplane[C::ITX];
    int xrow[C::ITX], xcol[C::ITX], xlds[C::ITX];
    int xbstride[C::ITX];   // channels of the tensor the plane lives in (batch stride in planes)
#pragma unroll
    for (int it = 0; it < C::ITX; ++it) {
        const int idx = it * 256 + tid;
        const int c = idx / (C::XR * (C::XW / 4)), r = idx - c * (C::XR * (C::XW / 4));
        const int row = r / (C::XW / 4), c4 = r - row * (C::XW / 4);
        const int ci = ci0 + c;
        xplane[it] = nullptr;
        xbstride[it] = 0;
        if (idx < C::NX && ci < a.Cin) {
            int sidx = 0;
            while (sidx + 1 < a.n_src && ci >= a.src_cstart[sidx + 1]) ++sidx;
            xplane[it] = a.src[sidx] + (long long)(a.src_choff[sidx] + (ci - a.src_cstart[sidx])) * in_plane;
            xbstride[it] = a.src_ctotal[sidx];
        }
        xrow[it] = row;
        xcol[it] = c4 * 4;
        xlds[it] = idx < C::NX ? c * C::XCP + row * C::XW + c4 * 4 : -1;
    }
    tr_f32x4 rdy[C::ITD], rx[C::ITX];
    auto fetch = [&](long long item) {
        const int ch = (int)(item % chunks);
        const long long row = item / chunks;
        const int b = (int)(row / row_groups), oy = (int)(row - (long long)b * row_groups) * R;
        const int x0 = ch * C::TW;
#pragma unroll
        for (int it = 0; it < C::ITD; ++it) {
            const int idx = it * 256 + tid, r2 = idx >> 9, c = (idx >> 4) & 31, c4 = idx & 15;
            const int co = co0 + c, ox = x0 + c4 * 4;
            rdy[it] = tr_f32x4{0.f, 0.f, 0.f, 0.f};
            if (co < a.Cout && ox < a.Wout && oy + r2 < a.Hout)   // Wout % 4 == 0: a 16-B piece is inside the row or outside it
                rdy[it] = *reinterpret_cast<const tr_f32x4 *>(dy + ((long long)b * a.Cout + co) * out_plane + (long long)(oy + r2) * a.Wout + ox);
        }
#pragma unroll
        for (int it = 0; it < C::ITX; ++it) {
            rx[it] = tr_f32x4{0.f, 0.f, 0.f, 0.f};
            const int iy = oy * STRIDE + xrow[it] - C::P, ix = x0 * STRIDE - 4 + xcol[it];
            if (xplane[it] && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                rx[it] = *reinterpret_cast<const tr_f32x4 *>(xplane[it] + (long long)b * xbstride[it] * in_plane + (long long)iy * a.Win + ix);
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int it = 0; it < C::ITD; ++it) {
            const int idx = it * 256 + tid;      // (row r2, cout c) = idx >> 4: rows of 32 couts follow each other
            *reinterpret_cast<tr_f32x4 *>(dy_s + (idx >> 4) * C::DP + (idx & 15) * 4) = rdy[it];
        }
#pragma unroll
        for (int it = 0; it < C::ITX; ++it)
            if (xlds[it] >= 0) *reinterpret_cast<tr_f32x4 *>(x_s + xlds[it]) = rx[it];
    };

    tr_f32x4 acc[2][C::KS2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t = 0; t < C::KS2; ++t) acc[n][t] = tr_f32x4{0.f, 0.f, 0.f, 0.f};
    const int m = lane & 15, kq = lane >> 4;
    const bool two = co0 + 16 < a.Cout;   // the second cout tile exists (uniform)

    long long item = slab;
    if (item < items) fetch(item);
    for (; item < items; item += slabs) {
        __syncthreads();            // the previous segment has been multiplied
        park();
        __syncthreads();
        if (item + slabs < items) fetch(item + slabs);   // in flight during the MFMAs
#pragma unroll
        for (int r2 = 0; r2 < R; ++r2)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int q = wave * 4 + qq;                 // 4-pixel group of the segment
            const float a0 = dy_s[(r2 * 32 + m) * C::DP + q * 4 + kq];
            const float a1 = two ? dy_s[(r2 * 32 + 16 + m) * C::DP + q * 4 + kq] : 0.f;
#pragma unroll
            for (int t = 0; t < C::KS2; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                const float bv = x_s[m * C::XCP + (r2 * STRIDE + ky) * C::XW + (q * 4 + kq) * STRIDE + kx - C::P + 4];
                acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][t], 0, 0, 0);
                if (two) acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][t], 0, 0, 0);
            }
        }
    }
    // combine the 4 waves (3 taps per pass through LDS, fixed order) -> partial[slab][co][ci][tap]
    const int ci_pad = gridDim.y * 16, co_pad = gridDim.x * 32;
    float *red = lds;   // [3 taps][4 waves][256]
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int t0 = 0; t0 < C::KS2; t0 += 3) {
            __syncthreads();
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
                if (t0 + tt < C::KS2)
#pragma unroll
                    for (int i = 0; i < 4; ++i) red[(tt * 4 + wave) * 256 + i * 64 + lane] = acc[n][t0 + tt][i];
            __syncthreads();
            for (int e = tid; e < 3 * 256; e += 256) {
                const int tt = e >> 8, r = e & 255, i = r >> 6, l = r & 63;
                if (t0 + tt >= C::KS2) continue;
                const float v = ((red[(tt * 4 + 0) * 256 + r] + red[(tt * 4 + 1) * 256 + r]) + red[(tt * 4 + 2) * 256 + r]) + red[(tt * 4 + 3) * 256 + r];
                const int mm = co0 + n * 16 + 4 * (l >> 4) + i, nn = ci0 + (l & 15);
                partial[(((long long)slab * co_pad + mm) * ci_pad + nn) * C::KS2 + t0 + tt] = v;
            }
        }
#endif
}

// partial[slab][co_pad][ci_pad][ks2] -> dw[co][ci][tap] += sum over slabs.  A block = 64 consecutive outputs x 4 slab lanes:
// lane j adds the slabs j, j + 4, ... in order, the four sums are combined in fixed order through LDS (deterministic).  (As
// one thread per output walking all slabs this kernel was 5.7 ms of a 27 ms step: 90 blocks for a 91 -> 28 layer, each
// thread 341 dependent, strided loads.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *partial, int slabs, int co_pad, int ci_pad, int ks2, int Cout, int Cin,
                                                           float *dw) {
    __shared__ float sm[4][64];
    const int j = threadIdx.x >> 6, l = threadIdx.x & 63;
    const long long o = (long long)blockIdx.x * 64 + l;
    const long long total = (long long)Cout * Cin * ks2;
    float s = 0.f;
    if (o < total) {
        const int t = (int)(o % ks2);
        const int ci = (int)((o / ks2) % Cin), co = (int)(o / ((long long)ks2 * Cin));
        const long long at = ((long long)co * ci_pad + ci) * ks2 + t, stride = (long long)co_pad * ci_pad * ks2;
        // four independent chains per lane (slabs k, k + 4, k + 8, k + 12 of its residue class): the loads of a chain are dependent
        // round trips to the L2, and a 91 -> 28 layer has 170 slabs; fixed order, so still bit-reproducible
        const float *p = partial + at;
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = j;
        for (; k + 12 < slabs; k += 16) {
            s += p[k * stride];
            s1 += p[(k + 4) * stride];
            s2 += p[(k + 8) * stride];
            s3 += p[(k + 12) * stride];
        }
        for (; k < slabs; k += 4) s += p[k * stride];
        s = (s + s1) + (s2 + s3);
    }
    sm[j][l] = s;
    __syncthreads();
    if (j == 0 && o < total) dw[o] += ((sm[0][l] + sm[1][l]) + sm[2][l]) + sm[3][l];
}

static bool wgrad_tiled_ok(int Win, int Wout) { return (Win & 3) == 0 && (Wout & 3) == 0; }
// cout tiles per workgroup: 2 in the tiled form
int wgrad_slabs(int cout, int cin, int ks, int B, int Hout, int Win, int Wout) {
    const bool tiled = wgrad_tiled_ok(Win, Wout);
    const long long tiles = (long long)(tiled ? (cout + 31) / 32 : (cout + 15) / 16) * ((cin + 15) / 16);
    // tiled: 512 workgroups per layer (2 per CU).  1024 gave each kernel more latency hiding alone, but twice the partial sums to
    // write and reduce, beside a backward chain that fills the other half of the chip anyway: 16.45 -> 16.2 ms per step (256: 17.9,
    // 384: 16.7, 768: 16.65 - same-box runs, profiles/r04_experiments.md)
    // (the 1x1 form - 15 KB of LDS, 58 registers, 8 workgroups per CU - wants the deeper queue: 2048, 15.65 -> 15.4 ms)
    long long s = (tiled ? (ks == 1 ? 2048 : 512) : 2048) / tiles;
    s = s < 1 ? 1 : s;
    const long long units = tiled ? (long long)B * Hout * ((Wout + 63) / 64) : (long long)B * Hout;
    return (int)(s > units ? units : s);
}
size_t wgrad_partial_floats(int cout, int cin, int ks, int B, int Hout, int Win, int Wout) {
    return (size_t)wgrad_slabs(cout, cin, ks, B, Hout, Win, Wout) * ((cout + 31) / 32 * 32) * ((cin + 15) / 16 * 16) * ks * ks;
}

int g_opt_wgrad_taps = 1;   // wgrad_taps: 3x3 stride-1 weight gradients with the taps folded into the matrix rows (wgrad_taps.hip)

int launch_wgrad(const ConvArgs &a, int ks, int stride, const float *dy, int B, float *partial, float *dw, hipStream_t s) {
    int slabs = wgrad_slabs(a.Cout, a.Cin, ks, B, a.Hout, a.Win, a.Wout);
    const bool tiled = wgrad_tiled_ok(a.Win, a.Wout);
    if (wgrad_taps_wanted(g_opt_wgrad_taps, ks, stride, a.Cin, a.Cout, a.Hin, a.Win, a.Hout, a.Wout)) {
        const int co_pad = (a.Cout + 31) / 32 * 32, ci_pad = (a.Cin + 15) / 16 * 16;      // (what wgrad_partial_floats sized the buffer for)
        const double flops = 2.0 * B * a.Hout * a.Wout * (double)a.Cout * a.Cin * ks * ks;
        {
            ProfScope ps(s, "wgrad_taps_kernel", flops, 4.0 * B * ((double)a.Cout * a.Hout * a.Wout + (double)a.Cin * a.Hin * a.Win));
            int rc = launch_wgrad_taps(a, dy, B, slabs, co_pad, ci_pad, partial, &slabs, s);
            if (rc) return rc;
        }
        const long long total = (long long)a.Cout * a.Cin * ks * ks;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, partial, slabs, co_pad, ci_pad, ks * ks, a.Cout, a.Cin, dw);
        PF_LAUNCH_CHECK("wgrad_reduce_kernel");
        return PF_OK;
    }
    const bool two_rows = tiled && ks == 3 && stride == 1;      // R = 2: half as many work items (the partial buffer is sized for R = 1)
    if (two_rows) {
        const long long items2 = (long long)B * ((a.Hout + 1) / 2) * ((a.Wout + 63) / 64);
        slabs = (int)(slabs > items2 ? items2 : slabs);
    }
    const dim3 grid(tiled ? (a.Cout + 31) / 32 : (a.Cout + 15) / 16, (a.Cin + 15) / 16, slabs);
    const double flops = 2.0 * B * a.Hout * a.Wout * (double)a.Cout * a.Cin * ks * ks;
    {
        ProfScope ps(s, tiled ? "wgrad_tiled_kernel" : "wgrad_partial_kernel", flops,
                     4.0 * B * ((double)a.Cout * a.Hout * a.Wout + (double)a.Cin * a.Hin * a.Win));
        if (tiled) {
            if (ks == 3 && stride == 1) hipLaunchKernelGGL((wgrad_tiled_kernel<3, 1, 2>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else if (ks == 3 && stride == 2) hipLaunchKernelGGL((wgrad_tiled_kernel<3, 2, 1>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else if (ks == 1 && stride == 1) hipLaunchKernelGGL((wgrad_tiled_kernel<1, 1, 1>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else return fail(PF_EUNSUPPORTED, "wgrad: k=%d stride=%d", ks, stride);
        } else {
            if (ks == 3 && stride == 1) hipLaunchKernelGGL((wgrad_partial_kernel<3, 1>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else if (ks == 3 && stride == 2) hipLaunchKernelGGL((wgrad_partial_kernel<3, 2>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else if (ks == 1 && stride == 1) hipLaunchKernelGGL((wgrad_partial_kernel<1, 1>), grid, dim3(256), 0, s, a, dy, B, slabs, partial);
            else return fail(PF_EUNSUPPORTED, "wgrad: k=%d stride=%d", ks, stride);
        }
        PF_LAUNCH_CHECK("wgrad kernel");
    }
    const long long total = (long long)a.Cout * a.Cin * ks * ks;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, partial, slabs, (int)grid.x * (tiled ? 32 : 16),
                       (int)grid.y * 16, ks * ks, a.Cout, a.Cin, dw);
    PF_LAUNCH_CHECK("wgrad_reduce_kernel");
    return PF_OK;
}

// ---- widths that are not a multiple of 4 (the 50- and 25-pixel levels of an 800 x 800 crop): the LDS-DMA kernels move
// 16-B pieces and need 16-B aligned rows.  Such a conv runs on COPIES with the row pitch rounded up to 4: its input ranges
// gathered into one contiguous tensor [B][Cin][H][Wp] with zero pad columns (= the convolution's own zero padding, so the
// real columns of the result are unchanged), its result [B][Cout][Hout][Wop] scattered back into the destination slice
// (accumulating for backward-data).  Two small copy kernels around a fast conv instead of the register-staged generic
// kernel (1.7-16 TFLOP/s on these levels: 14 ms of a 44 ms step).
struct PadGatherArgs {
    const float *src[kConvMaxSrc];
    int ctotal[kConvMaxSrc], choff[kConvMaxSrc], cstart[kConvMaxSrc + 1], n_src;
};
__global__ __launch_bounds__(256) void pad_gather_kernel(PadGatherArgs g, int Cin, int H, int W, int Wp, float *dst) {
    const int bc = blockIdx.y, b = bc / Cin, c = bc - b * Cin;
    const int i4 = blockIdx.x * 256 + threadIdx.x;          // 4-column group of the padded plane
    const int wp4 = Wp / 4;
    if (i4 >= H * wp4) return;
    const int y = i4 / wp4, x = (i4 - y * wp4) * 4;
    int j = 0;
    while (j + 1 < g.n_src && c >= g.cstart[j + 1]) ++j;
    const float *sp = g.src[j] + ((long long)b * g.ctotal[j] + g.choff[j] + (c - g.cstart[j])) * H * W + (long long)y * W;
    tr_f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = x + k < W ? sp[x + k] : 0.f;
    *reinterpret_cast<tr_f32x4 *>(dst + ((long long)bc * H + y) * Wp + x) = v;
}
__global__ __launch_bounds__(256) void unpad_scatter_kernel(const float *src, int C, int H, int W, int Wp, float *dst, int dst_ctotal, int dst_choff,
                                                            int accum) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    float *d = dst + ((long long)b * dst_ctotal + dst_choff + c) * H * W + i;
    const float v = src[((long long)bc * H + y) * Wp + x];
    *d = accum ? *d + v : v;
}
// the result of ONE backward-data convolution over all input ranges of a layer ([B][C][H][Wp]) added to the gradient of each
// range's tensor (a null destination - the network input - is skipped)
struct ScatterArgs {
    float *dst[kConvMaxSrc];
    int ctotal[kConvMaxSrc], choff[kConvMaxSrc], cstart[kConvMaxSrc + 1], overwrite[kConvMaxSrc], n;
};
__global__ __launch_bounds__(256) void unpad_scatter_multi_kernel(const float *src, int C, int H, int W, int Wp, ScatterArgs t) {
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    int j = 0;
    while (j + 1 < t.n && c >= t.cstart[j + 1]) ++j;
    if (!t.dst[j]) return;
    const int y = i / W, x = i - y * W;
    float *d = t.dst[j] + ((long long)b * t.ctotal[j] + t.choff[j] + (c - t.cstart[j])) * H * W + i;
    const float v = src[((long long)bc * H + y) * Wp + x];
    *d = t.overwrite[j] ? v : *d + v;
}
int launch_unpad_scatter_multi(const float *src, int B, int C, int H, int W, int Wp, float *const *dst, const int *ctotal, const int *choff,
                               const int *ch, const int *overwrite, int n, hipStream_t s) {
    if (n < 1 || n > kConvMaxSrc) return fail(PF_EINVAL, "unpad_scatter_multi: %d ranges", n);
    ScatterArgs t;
    memset(&t, 0, sizeof(t));
    t.n = n;
    int c0 = 0;
    for (int j = 0; j < n; ++j) {
        t.dst[j] = dst[j]; t.ctotal[j] = ctotal[j]; t.choff[j] = choff[j]; t.cstart[j] = c0; t.overwrite[j] = overwrite[j];
        c0 += ch[j];
    }
    for (int j = n; j <= kConvMaxSrc; ++j) t.cstart[j] = c0;
    if (c0 != C) return fail(PF_EINVAL, "unpad_scatter_multi: ranges cover %d of %d channels", c0, C);
    hipLaunchKernelGGL(unpad_scatter_multi_kernel, dim3((unsigned)((H * W + 255) / 256), B * C), dim3(256), 0, s, src, C, H, W, Wp, t);
    PF_LAUNCH_CHECK("unpad_scatter_multi_kernel");
    return PF_OK;
}
// zeros in channels [c0, c0 + n) of a [B][ctotal][HW] tensor (the few gradient channels whose first writer accumulates)
__global__ __launch_bounds__(256) void zero_channels_kernel(float *t, long long run, long long batch_stride) {
    float *p = t + (long long)blockIdx.y * batch_stride;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < run; i += (long long)gridDim.x * 256) p[i] = 0.f;
}
int launch_zero_channels(float *t, int B, int ctotal, int c0, int n, long long HW, hipStream_t s) {
    const long long run = (long long)n * HW;
    long long blocks = (run + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    hipLaunchKernelGGL(zero_channels_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, t + (long long)c0 * HW, run, (long long)ctotal * HW);
    PF_LAUNCH_CHECK("zero_channels_kernel");
    return PF_OK;
}
int launch_pad_gather(const ConvArgs &a, int B, int Wp, float *dst, hipStream_t s) {
    PadGatherArgs g;
    g.n_src = a.n_src;
    for (int j = 0; j < kConvMaxSrc; ++j) {
        g.src[j] = a.src[j]; g.ctotal[j] = a.src_ctotal[j]; g.choff[j] = a.src_choff[j]; g.cstart[j] = a.src_cstart[j];
    }
    g.cstart[kConvMaxSrc] = a.src_cstart[kConvMaxSrc];
    hipLaunchKernelGGL(pad_gather_kernel, dim3((unsigned)((a.Hin * (Wp / 4) + 255) / 256), B * a.Cin), dim3(256), 0, s, g, a.Cin, a.Hin, a.Win, Wp, dst);
    PF_LAUNCH_CHECK("pad_gather_kernel");
    return PF_OK;
}
int launch_unpad_scatter(const float *src, int B, int C, int H, int W, int Wp, float *dst, int dst_ctotal, int dst_choff, int accum, hipStream_t s) {
    hipLaunchKernelGGL(unpad_scatter_kernel, dim3((unsigned)((H * W + 255) / 256), B * C), dim3(256), 0, s, src, C, H, W, Wp, dst, dst_ctotal, dst_choff,
                       accum);
    PF_LAUNCH_CHECK("unpad_scatter_kernel");
    return PF_OK;
}

// zero-stuffing for the backward-data pass of a stride-2 conv: up[b][c][2*oy][2*ox] = dy[b][c][oy][ox], zeros elsewhere
__global__ __launch_bounds__(256) void zero_stuff_kernel(const float *dy, int Hout, int Wout, int Hin, int Win, float *up) {
    const int bc = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Hin * Win) return;
    const int y = (int)(i / Win), x = (int)(i - (long long)y * Win);
    float v = 0.f;
    if (!(y & 1) && !(x & 1) && (y >> 1) < Hout && (x >> 1) < Wout) v = dy[((long long)bc * Hout + (y >> 1)) * Wout + (x >> 1)];
    up[(long long)bc * Hin * Win + i] = v;
}
int launch_zero_stuff(const float *dy, int planes, int Hout, int Wout, int Hin, int Win, float *up, hipStream_t s) {
    hipLaunchKernelGGL(zero_stuff_kernel, dim3((unsigned)(((long long)Hin * Win + 255) / 256), planes), dim3(256), 0, s, dy, Hout, Wout, Hin, Win, up);
    PF_LAUNCH_CHECK("zero_stuff_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ pool / upsample backward
// AvgPool2d(2,2): gin[2y+dy][2x+dx] += 0.25 * gout[y][x]   (rows/cols dropped by the floor division get no gradient)
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float *gout, int Hin, int Win, int overwrite, float *gin) {
    const int bc = blockIdx.y, Ho = Hin >> 1, Wo = Win >> 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Hin * Win) return;
    const int y = (int)(i / Win), x = (int)(i - (long long)y * Win);
    const float v = ((y >> 1) < Ho && (x >> 1) < Wo) ? 0.25f * gout[((long long)bc * Ho + (y >> 1)) * Wo + (x >> 1)] : 0.f;
    float *d = gin + (long long)bc * Hin * Win + i;
    if (overwrite) *d = v;
    else *d += v;
}
int launch_avgpool2_bwd(const float *gout, int planes, int Hin, int Win, int overwrite, float *gin, hipStream_t s) {
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3((unsigned)(((long long)Hin * Win + 255) / 256), planes), dim3(256), 0, s, gout, Hin, Win, overwrite, gin);
    PF_LAUNCH_CHECK("avgpool2_bwd_kernel");
    return PF_OK;
}

// transpose of bilinear align_corners=True interpolation, gather form (deterministic): every source pixel sums the output
// pixels whose two taps per axis (lin_coord, the same function the forward kernels use) include it.
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float *gout, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                           const double *inv_count /* nullable: multiply by scale / *inv_count */, float scale,
                                                           int accumulate, float *gin) {
    const int bc = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Hi * Wi) return;
    const int iy = (int)(i / Wi), ix = (int)(i - (long long)iy * Wi);
    // candidate output rows: r = sh * oy in (iy - 1, iy + 1)  ->  widen by one on both sides, test the indices exactly
    int oy_lo = 0, oy_hi = Ho - 1, ox_lo = 0, ox_hi = Wo - 1;
    if (sh > 0.f) {
        oy_lo = max(0, (int)floorf((float)(iy - 1) / sh) - 1);
        oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / sh) + 1);
    }
    if (sw > 0.f) {
        ox_lo = max(0, (int)floorf((float)(ix - 1) / sw) - 1);
        ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / sw) + 1);
    }
    const float *gp = gout + (long long)bc * Ho * Wo;
    float sum = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float hy0, hy1;
        lin_coord(oy, sh, Hi, y0, y1, hy0, hy1);
        const float wy = (y0 == iy ? hy0 : 0.f) + (y1 == iy ? hy1 : 0.f);
        if (wy == 0.f) continue;
        float rs = 0.f;
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            int x0, x1;
            float lx0, lx1;
            lin_coord(ox, sw, Wi, x0, x1, lx0, lx1);
            const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
            if (wx != 0.f) rs += wx * gp[(long long)oy * Wo + ox];
        }
        sum += wy * rs;
    }
    if (inv_count) sum = (float)((double)sum * (double)scale / fmax(*inv_count, 1.0));
    float *d = gin + (long long)bc * Hi * Wi + i;
    *d = accumulate ? *d + sum : sum;
}
// The same sum as two passes over a [planes][Ho][Wi] scratch - rows first (tmp[oy][ix] = sum_ox wx g[oy][ox]), then columns
// (gin[iy][ix] = sum_oy wy tmp[oy][ix]): the nesting and the order of the one-pass kernel's own loops, so the same bits, with
// ~10 + 10 taps per source pixel instead of ~10 x 10 (the 4x head of the loss: 0.45 -> 0.1x ms)
__global__ __launch_bounds__(256) void upsample_bwd_rows_kernel(const float *gout, int Wi, int Ho, int Wo, float sw, float *tmp) {
    const int bc = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Ho * Wi) return;
    const int oy = (int)(i / Wi), ix = (int)(i - (long long)oy * Wi);
    int ox_lo = 0, ox_hi = Wo - 1;
    if (sw > 0.f) {
        ox_lo = max(0, (int)floorf((float)(ix - 1) / sw) - 1);
        ox_hi = min(Wo - 1, (int)ceilf((float)(ix + 1) / sw) + 1);
    }
    const float *row = gout + ((long long)bc * Ho + oy) * Wo;
    float rs = 0.f;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        int x0, x1;
        float lx0, lx1;
        lin_coord(ox, sw, Wi, x0, x1, lx0, lx1);
        const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        if (wx != 0.f) rs += wx * row[ox];
    }
    tmp[(long long)bc * Ho * Wi + i] = rs;
}
__global__ __launch_bounds__(256) void upsample_bwd_cols_kernel(const float *tmp, int Hi, int Wi, int Ho, float sh, const double *inv_count,
                                                                float scale, int accumulate, float *gin) {
    const int bc = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)Hi * Wi) return;
    const int iy = (int)(i / Wi), ix = (int)(i - (long long)iy * Wi);
    int oy_lo = 0, oy_hi = Ho - 1;
    if (sh > 0.f) {
        oy_lo = max(0, (int)floorf((float)(iy - 1) / sh) - 1);
        oy_hi = min(Ho - 1, (int)ceilf((float)(iy + 1) / sh) + 1);
    }
    const float *tp = tmp + (long long)bc * Ho * Wi + ix;
    float sum = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        int y0, y1;
        float hy0, hy1;
        lin_coord(oy, sh, Hi, y0, y1, hy0, hy1);
        const float wy = (y0 == iy ? hy0 : 0.f) + (y1 == iy ? hy1 : 0.f);
        if (wy == 0.f) continue;
        sum += wy * tp[(long long)oy * Wi];
    }
    if (inv_count) sum = (float)((double)sum * (double)scale / fmax(*inv_count, 1.0));
    float *d = gin + (long long)bc * Hi * Wi + i;
    *d = accumulate ? *d + sum : sum;
}
size_t upsample_bwd_tmp_floats(int planes, int Hi, int Wi, int Ho, int Wo) {
    return (long long)planes * Ho * Wo >= (1ll << 22) ? (size_t)planes * Ho * Wi : 0;     // small planes: one pass, one launch
}
int launch_upsample_bwd(const float *gout, int planes, int Hi, int Wi, int Ho, int Wo, const double *count, float scale, int accumulate,
                        float *gin, float *tmp, hipStream_t s) {
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    if (tmp && upsample_bwd_tmp_floats(planes, Hi, Wi, Ho, Wo)) {
        hipLaunchKernelGGL(upsample_bwd_rows_kernel, dim3((unsigned)(((long long)Ho * Wi + 255) / 256), planes), dim3(256), 0, s, gout, Wi, Ho, Wo, sw, tmp);
        hipLaunchKernelGGL(upsample_bwd_cols_kernel, dim3((unsigned)(((long long)Hi * Wi + 255) / 256), planes), dim3(256), 0, s, tmp, Hi, Wi, Ho, sh,
                           count, scale, accumulate, gin);
        PF_LAUNCH_CHECK("upsample_bwd_rows/cols_kernel");
        return PF_OK;
    }
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)(((long long)Hi * Wi + 255) / 256), planes), dim3(256), 0, s, gout, Hi, Wi, Ho, Wo, sh, sw,
                       count, scale, accumulate, gin);
    PF_LAUNCH_CHECK("upsample_bwd_kernel");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ loss
// F.interpolate(logits, (Ho,Wo), bilinear, align_corners=True) -> CrossEntropyLoss(ignore_index) (bg_model.py:44,81) and
// its gradient w.r.t. the upsampled logits, (softmax - onehot) on valid pixels (NOT yet divided by the valid count),
// plus the accuracy counters of :82-84.  partial[block][3] = {sum nll, valid, correct} in fp64.
template <int C>
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float *logits, int Hi, int Wi, const void *labels, int lab_i64, int Ho, int Wo,
                                                         int ignore, float sh, float sw, float *dfull, double *partial) {
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long HWo = (long long)Ho * Wo;
    __shared__ double sm[3 * 4];
    double v[3] = {0.0, 0.0, 0.0};
    if (i < HWo) {
        const int oy = (int)(i / Wo), ox = (int)(i - (long long)oy * Wo);
        int y0, y1, x0, x1;
        float hy0, hy1, lx0, lx1;
        lin_coord(oy, sh, Hi, y0, y1, hy0, hy1);
        lin_coord(ox, sw, Wi, x0, x1, lx0, lx1);
        const long long li = (long long)b * HWo + i;
        const long long lab = lab_i64 ? reinterpret_cast<const long long *>(labels)[li] : (long long)reinterpret_cast<const uint8_t *>(labels)[li];
        float z[C];
        float mx = -INFINITY;
        int arg = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float *p = logits + ((long long)b * C + c) * Hi * Wi;
            const float t0 = lx0 * p[(long long)y0 * Wi + x0] + lx1 * p[(long long)y0 * Wi + x1];
            const float t1 = lx0 * p[(long long)y1 * Wi + x0] + lx1 * p[(long long)y1 * Wi + x1];
            z[c] = hy0 * t0 + hy1 * t1;
            if (z[c] > mx) { mx = z[c]; arg = c; }
        }
        const bool valid = lab != ignore && lab >= 0 && lab < C;
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) se += expf(z[c] - mx);
        const float lse = mx + logf(se);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float p = expf(z[c] - lse);
            dfull[((long long)b * C + c) * HWo + i] = valid ? (p - (c == lab ? 1.f : 0.f)) : 0.f;
            if (valid && c == lab) v[0] = (double)(lse - z[c]);
        }
        if (lab != ignore) {          // :83-84: total counts labels != 255, correct compares the argmax with the label
            v[1] = 1.0;
            v[2] = arg == lab ? 1.0 : 0.0;
        }
    }
    block_reduce_d<3>(v, sm);
    if (threadIdx.x == 0) {
        double *p = partial + ((long long)b * gridDim.x + blockIdx.x) * 3;
        p[0] = v[0]; p[1] = v[1]; p[2] = v[2];
    }
}
__global__ __launch_bounds__(256) void sum3_kernel(const double *partial, long long n, double *out3) {
    __shared__ double sm[3 * 4];
    double v[3] = {0.0, 0.0, 0.0};
    for (long long i = threadIdx.x; i < n; i += 256) {
        v[0] += partial[i * 3 + 0]; v[1] += partial[i * 3 + 1]; v[2] += partial[i * 3 + 2];
    }
    block_reduce_d<3>(v, sm);
    if (threadIdx.x == 0) { out3[0] = v[0]; out3[1] = v[1]; out3[2] = v[2]; }
}
size_t ce_partial_doubles(int B, int Ho, int Wo) { return (size_t)B * (((size_t)Ho * Wo + 255) / 256) * 3; }

int launch_ce_fwd_bwd(const float *logits, int B, int C, int Hi, int Wi, const void *labels, int lab_i64, int Ho, int Wo, int ignore,
                      float *dfull, double *partial, double *out3, hipStream_t s) {
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.f, sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.f;
    const dim3 grid((unsigned)(((long long)Ho * Wo + 255) / 256), B);
    if (C == 11) hipLaunchKernelGGL((ce_fwd_bwd_kernel<11>), grid, dim3(256), 0, s, logits, Hi, Wi, labels, lab_i64, Ho, Wo, ignore, sh, sw, dfull, partial);
    else if (C == 19) hipLaunchKernelGGL((ce_fwd_bwd_kernel<19>), grid, dim3(256), 0, s, logits, Hi, Wi, labels, lab_i64, Ho, Wo, ignore, sh, sw, dfull, partial);
    else return fail(PF_EUNSUPPORTED, "cross entropy kernels are built for 11 or 19 classes, got %d", C);
    hipLaunchKernelGGL(sum3_kernel, dim3(1), dim3(256), 0, s, partial, (long long)grid.x * B, out3);
    PF_LAUNCH_CHECK("ce_fwd_bwd");
    return PF_OK;
}

// ------------------------------------------------------------------------------------------------ optimiser
// nn.utils.clip_grad_norm_ (train.py:207-208): coef = min(1, max_norm / (||g||_2 + 1e-6)) over the trainable elements
constexpr int kNormBlocks = 512;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *g, const uint8_t *trainable, long long n, double *partial) {
    __shared__ double sm[4];
    double v[1] = {0.0};
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)kNormBlocks * 256)
        if (trainable[i]) v[0] += (double)g[i] * (double)g[i];
    block_reduce_d<1>(v, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = v[0];
}
__global__ __launch_bounds__(256) void clip_coef_kernel(const double *partial, float max_norm, float *coef_norm2 /* [2]: coef, total norm */) {
    __shared__ double sm[4];
    double v[1] = {0.0};
    for (int i = threadIdx.x; i < kNormBlocks; i += 256) v[0] += partial[i];
    block_reduce_d<1>(v, sm);
    if (threadIdx.x == 0) {
        const double norm = sqrt(v[0]);
        const double c = (double)max_norm / (norm + 1e-6);
        coef_norm2[0] = max_norm > 0.f ? (float)(c < 1.0 ? c : 1.0) : 1.f;
        coef_norm2[1] = (float)norm;
    }
}
// torch.optim.SGD (train.py:138): g += wd * p;  buf = first ? g : momentum * buf + g;  p -= lr * buf
__global__ __launch_bounds__(256) void sgd_kernel(float *theta, float *grad, float *mom, const uint8_t *trainable, long long n, float lr,
                                                  float momentum, float wd, const float *coef, float clip_value, int first) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !trainable[i]) return;
    float g = grad[i] * coef[0];
    if (clip_value > 0.f) g = fminf(fmaxf(g, -clip_value), clip_value);     // clip_grad_value_ (train.py:205-206)
    grad[i] = g;                                                            // what .grad holds after clipping
    g += wd * theta[i];
    const float buf = (first || momentum == 0.f) ? g : momentum * mom[i] + g;
    mom[i] = buf;
    theta[i] -= lr * buf;
}
size_t sgd_ws_bytes() { return kNormBlocks * sizeof(double) + 64; }

int launch_sgd(float *theta, float *grad, float *mom, const uint8_t *trainable, long long n, float lr, float momentum, float wd, float clip_norm,
               float clip_value, int first, void *ws, hipStream_t s) {
    double *partial = (double *)ws;
    float *coef = (float *)((char *)ws + kNormBlocks * sizeof(double));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(kNormBlocks), dim3(256), 0, s, grad, trainable, n, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, s, partial, clip_norm, coef);
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, theta, grad, mom, trainable, n, lr, momentum, wd, coef,
                       clip_value, first);
    PF_LAUNCH_CHECK("sgd");
    return PF_OK;
}

}  // namespace pf
