// Training-side kernels — interface (train_kernels.hip); orchestrated by train_plan.hip.
#pragma once
#include "conv_mfma.h"

namespace pf {

int launch_onehot_dense(const void *seg, int seg_i64, const float *depth, const uint8_t *mask, float mean, float stdv, int B, int T,
                        int n_cls, int H, int W, float *x, hipStream_t s);
// OIHW (device) -> fragment order of the generic MFMA conv (conv_mfma.hip); transpose_flip = 1 packs the backward-data
// weights of forward input range [c0, c0 + ch)
int launch_pack_weights(const float *w, int cin_f, int cout_f, const ConvTiling &t, int transpose_flip, int c0, int ch, float *out,
                        hipStream_t s);
// ... -> the per-tile order of the LDS-DMA kernels (conv_dma.hip): src_ch[n_src] = the conv's input ranges (forward), or
// {forward cout} with transpose_flip (backward-data of forward input range [c0, c0 + ch): a ch-output, cout_f-input conv)
size_t tiled_packed_floats(const int *src_ch, int n_src, int cout, int ks, int stride);
int launch_pack_weights_tiled(const float *w, int cin_f, int cout_f, int ks, int stride, const int *src_ch, int n_src, int transpose_flip,
                              int c0, int ch, float *out, hipStream_t s);
// the same packings for a whole training step in ceil(n / kPackBatch) launches: jobs as kernel arguments
constexpr int kPackBatch = 24;
struct PackJob {
    long long w_off, out_off, total;   // floats: weights inside theta, packing inside the arena, packed size
    int cin_f, cout_f, ks2, kc, tflip, c0, ch, n_src, block0;
    int cstart[kConvMaxSrc + 1], chunk0[kConvMaxSrc + 1];
};
struct PackBatch {
    int n;
    PackJob job[kPackBatch];
};
static_assert(sizeof(PackBatch) <= 3584, "PackBatch travels as kernel arguments");
void pack_job_fill(PackJob &q, size_t w_off, int cin_f, int cout_f, int ks, int stride, const int *src_ch, int n_src, int transpose_flip, int c0,
                   int ch, size_t out_off);
int launch_pack_weights_batch(const float *theta, float *arena, const PackJob *jobs, int n, hipStream_t s);
size_t bn_partial_doubles(int C);
int launch_bn_forward(const float *y, int B, int C, int H, int W, float eps, float momentum, const float *gamma, const float *beta,
                      float *running_mean, float *running_var, float *mean, float *invstd, double *partial, float *dst, int dst_ctotal,
                      int dst_choff, int relu, int y_pitch /* 0 = dense; else floats per row of y (>= W) */, hipStream_t s);
int launch_bn_backward(const float *g, int t_ctotal, int choff, const float *y, const float *mean, const float *invstd,
                       const float *gamma, const float *beta, int B, int C, int H, int W, int relu, float *dgamma, float *dbeta, double *partial,
                       float *dy, int dy_pitch /* 0 = dense rows; else floats per row (>= W), pad columns zeroed */,
                       int y_pitch /* as launch_bn_forward; then dy_pitch must equal it */, hipStream_t s);
int launch_bias_backward(const float *g, int t_ctotal, int choff, int B, int C, int H, int W, float *dbias, double *partial /* bn_partial_doubles(C) */,
                         float *dy, int dy_pitch /* as launch_bn_backward */, hipStream_t s);
int wgrad_slabs(int cout, int cin, int ks, int B, int Hout, int Win, int Wout);
size_t wgrad_partial_floats(int cout, int cin, int ks, int B, int Hout, int Win, int Wout);
// a: the forward conv's arguments (sources, Cin/Cout, Hin/Win/Hout/Wout); dw (OIHW) accumulates
int launch_wgrad(const ConvArgs &a, int ks, int stride, const float *dy, int B, float *partial, float *dw, hipStream_t s);
// train_s4.hip: the forward convolutions of a step on the packed-pair kernels (conv_s4.hip)
constexpr int kS4WBatch = 16;
constexpr float kS4TrainWeightScale = 4096.f;     // 2^12: |w| < 16 fits fp16 (a larger weight is stored as NaN: the step's loss is NaN)
constexpr float kS4TrainActScale = 1.f;           // activations are shadowed as they are (post-BatchNorm values are O(1); |z| < 2^-2 keeps 2^-25 absolute:
                                                  // conv_mfma.h.  A 2^6 pre-scale was measured: same forward distance, profiles/r06_experiments.md)
struct S4WJob {            // one conv's device-side weight packing (pack_conv_weights_s4's layout)
    long long w_off;       // OIHW weights: floats into theta
    unsigned out_off;      // floats into the packed arena
    int cin, cout, ks, n_src, pad, rounds, nblocks, first_block;
    int choff[kConvMaxSrc], ch[kConvMaxSrc];
};
struct S4WBatch {
    int n;
    float scale;
    S4WJob job[kS4WBatch];
};
static_assert(sizeof(S4WBatch) <= 3584, "S4WBatch travels as kernel arguments");
size_t s4_wjob_init(S4WJob &jb, size_t w_off, int cin, int cout, int ks, const S4Range *r, int n_src, int pad_sources);
int launch_s4_pack_weights_dev(const float *theta, float *arena, const S4WJob *jobs, int n, float scale, hipStream_t s);
// channels [c0, c1) of the fp32 tensor src [B][ctotal][H][W], times `scale` (a power of two) -> its packed-pair shadow with rows of
// Wp >= W pixels (zeros in the pad columns).  fill_lo / fill_up: also zero the half group in front of / behind a slice that starts /
// ends in the middle of a 4-channel group
int launch_s4_pack_act(const float *src, int B, int ctotal, int c0, int c1, int fill_lo, int fill_up, int H, int W, int Wp, void *dst, float scale,
                       hipStream_t s);
// wgrad_taps.hip: 3x3 stride-1 layers, the taps folded into the matrix rows; partial[slab][co_pad][ci_pad][9], *slabs_used <= max_slabs
bool wgrad_taps_ok(int ks, int stride, int Hin, int Win, int Hout, int Wout);
// mode = option wgrad_taps: 0 never, 2 wherever the kernel exists, 1 where it measured faster than wgrad_tiled_kernel
bool wgrad_taps_wanted(int mode, int ks, int stride, int Cin, int Cout, int Hin, int Win, int Hout, int Wout);
int launch_wgrad_taps(const ConvArgs &a, const float *dy, int B, int max_slabs, int co_pad, int ci_pad, float *partial, int *slabs_used, hipStream_t s);
// odd-width convs on copies with a row pitch rounded up to 4 (train_kernels.hip): all input ranges of `a` -> [B][Cin][Hin][Wp];
// [B][C][H][Wp] -> channels [dst_choff, dst_choff + C) of a [B][dst_ctotal][H][W] tensor
int launch_pad_gather(const ConvArgs &a, int B, int Wp, float *dst, hipStream_t s);
int launch_unpad_scatter_multi(const float *src, int B, int C, int H, int W, int Wp, float *const *dst /* null = skip */, const int *ctotal,
                               const int *choff, const int *ch, const int *overwrite /* per range: 1 = store, 0 = add */, int n, hipStream_t s);
int launch_unpad_scatter(const float *src, int B, int C, int H, int W, int Wp, float *dst, int dst_ctotal, int dst_choff, int accum, hipStream_t s);
int launch_zero_stuff(const float *dy, int planes, int Hout, int Wout, int Hin, int Win, float *up, hipStream_t s);
int launch_avgpool2_bwd(const float *gout, int planes, int Hin, int Win, int overwrite /* 0: gin += */, float *gin, hipStream_t s);
int launch_zero_channels(float *t, int B, int ctotal, int c0, int n, long long HW, hipStream_t s);
// gin (+)= bilinear^T(gout) [* scale / *count when count != nullptr].  tmp: upsample_bwd_tmp_floats() floats of scratch (0 =
// small planes, none needed; tmp may be null: the one-pass kernel, same bits)
size_t upsample_bwd_tmp_floats(int planes, int Hi, int Wi, int Ho, int Wo);
int launch_upsample_bwd(const float *gout, int planes, int Hi, int Wi, int Ho, int Wo, const double *count, float scale, int accumulate,
                        float *gin, float *tmp, hipStream_t s);
size_t ce_partial_doubles(int B, int Ho, int Wo);
int launch_ce_fwd_bwd(const float *logits, int B, int C, int Hi, int Wi, const void *labels, int lab_i64, int Ho, int Wo, int ignore,
                      float *dfull, double *partial, double *out3, hipStream_t s);
size_t sgd_ws_bytes();
int launch_sgd(float *theta, float *grad, float *mom, const uint8_t *trainable, long long n, float lr, float momentum, float wd, float clip_norm,
               float clip_value, int first, void *ws, hipStream_t s);

}  // namespace pf
