// bg training step (scope row f4): forward in training mode + loss + backward over the op table, and the optimiser step.
//
// Replaces, for task `bg`, what the reference's training loop does per batch (training/train.py:185-222):
//     loss_dict = model.loss(inputs, labels)      models/bg/bg_model.py:73-89  (train-mode BatchNorm, F.interpolate, CrossEntropyLoss)
//     loss.backward()                             autograd through hardnet.py:353-387
//     clip_grad_value_ / clip_grad_norm_ ; opt.step()  (SGD, momentum, weight decay: train.py:130-138)
// Parameters live in ONE flat fp32 arena `theta` (layout below; the Python side maps the reference's state_dict keys onto
// it), gradients in an arena of the same layout — so the data-parallel exchange is a single all-reduce of ~16.5 MB over
// RCCL between pf_train_forward_backward and pf_sgd_step (the reference wraps the model in DDP, train.py:96-103).
//
// theta layout: for every conv op of the table, in table order:  W[cout][cin][k][k],  then
//     with BatchNorm:  gamma[cout], beta[cout], running_mean[cout], running_var[cout]      (ConvLayer, hardnet.py:16-25)
//     without       :  bias[cout]                                                          (finalConv, hardnet.py:325-327)
//
// Convolutions run on the fp32 matrix cores: forward and backward-data through the LDS-DMA kernels of the inference path
// (conv_dma.hip; the generic implicit-GEMM kernel of conv_mfma.hip for widths that are not a multiple of 4) with the
// weights re-packed on the device every step (forward order, or transposed + flipped per input range; stride-2
// backward-data = stride-1 conv over the zero-stuffed gradient; a tensor's gradient accumulates over its consumers in
// the store), backward-weight through the LDS-tiled wgrad kernels of train_kernels.hip.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <vector>

#include "net_kernels.h"
#include "pf_blob.h"
#include "pf_prof.h"
#include "train_kernels.h"

using namespace pf;

#ifndef PF_TRAIN_SIDE_STREAMS
#define PF_TRAIN_SIDE_STREAMS 1
#endif
#ifndef PF_TRAIN_DY_SLOTS
#define PF_TRAIN_DY_SLOTS 4
#endif
static_assert(PF_TRAIN_DY_SLOTS % PF_TRAIN_SIDE_STREAMS == 0, "a dy slot belongs to one side stream");

struct pf_train {
    BlobHeader hdr;
    std::vector<BlobTensor> tensors;
    std::vector<BlobOp> ops;
    std::vector<size_t> w_off, aux_off;   // per op (floats into theta); aux = gamma (BN) or bias
    std::vector<int> bn;                  // per op: 1 = conv + BN (+ ReLU), 0 = plain conv with bias
    size_t n_params = 0;
    float *dev_zero = nullptr;            // 1024 zeros (bias of the BN-less generic conv launches)
    // the weight gradients are leaves of the backward pass: they run on this plan's own lower-priority stream, forked from and
    // joined to the caller's stream inside every pf_train_forward_backward (so a stream capture of the call stays one graph).
    // Option "train_side_stream" (read when the plan is created; 0 = everything on the caller's stream)
    // kSideStreams of them, layer n on stream n % kSideStreams; kDySlots conv-output-gradient buffers decouple the two sides:
    // the caller's stream may run kDySlots layers ahead of the weight gradients
    static constexpr int kSideStreams = PF_TRAIN_SIDE_STREAMS, kDySlots = PF_TRAIN_DY_SLOTS;
    hipStream_t side = nullptr;             // == sides[0]; non-null <=> the side streams are in use
    hipStream_t sides[kSideStreams] = {};
    hipEvent_t ev_dy[kDySlots] = {}, ev_wg[kDySlots] = {}, ev_join[kSideStreams] = {};
    // pf_train_autotune: the workgroup shape (pixel waves x cout tiles) of every forward / backward-data convolution is MEASURED
    // the first time its geometry is seen outside a stream capture (every candidate of conv_dma's shape list, 3 launches each,
    // hipEvents) instead of taken from the inference path's cost model, which was calibrated on 1024x2048 batches.  Off by default:
    // the choice (and with it the summation order of K-split shapes) then depends on timing, i.e. may differ from run to run.
    int autotune = 0;
    int fwd_s4 = 0;                       // option "train_forward_s4" when the plan was created: forward convolutions on conv_s4 (train_s4.hip)
    mutable std::map<std::array<int, 8>, std::pair<int, int>> tuned;   // (ks, stride, Cin, Cout, Hin, Win, B, accum) -> (wm, nt)
    mutable hipEvent_t tune_ev[2] = {nullptr, nullptr};
    // the measurements run in a pass of their own in front of the first real pass of a configuration (B, H, W, out_h, out_w):
    // its launches repeat, so what they accumulate is garbage - it goes to a scratch gradient and leaves theta alone
    mutable bool measuring = false;
    // pf_train_path_stats: which code paths the LAST pf_train_forward_backward took (tests assert that a timed configuration
    // really ran the table's shapes and the padded odd-width forms, not a fallback)
    mutable int stats[8] = {};
    mutable std::vector<std::array<int, 5>> measured_configs;
};

namespace pf {
extern int g_opt_train_kacc;
extern int g_opt_train_side, g_opt_use_tuned, g_opt_up_two_pass, g_opt_train_table_batch;
extern int g_opt_train_s4;      // train_forward_s4: the forward convolutions of a step on the packed-pair kernels (train_s4.hip)
}

namespace {

// measured shapes of the training step's convolutions for the configurations tools/tune_train.py was run on (the reference's
// configs/bg/bg_train.yaml: batch 8 of 800x800 crops): {ks, stride, Cin, Cout, Hin, Win, B, accum, wm, nt}.  Consulted when option
// "use_tuned_table" is on (default) and the plan does not measure for itself; other geometries take the cost model's shape
struct TrainTuned {
    int key[8], wm, nt;
};
const TrainTuned kTrainTuned[] = {
#include "train_tuned.inc"
    {{0, 0, 0, 0, 0, 0, 0, 0}, 0, 0}};

// conv_dma with the measured shape for the geometry: this plan's own measurement (pf_train::autotune, measuring first if need
// be), else the table's, else the cost model's
int train_conv_dma(const pf_train *p, const ConvArgs &c0, int ks, int stride, int B, hipStream_t s) {
    // blocked summation in the 3x3 convolutions of a training step (conv_dma.hip: KACC; option train_blocked_sum, on)
    ConvArgs c = c0;
    c.kacc = (g_opt_train_kacc && ks == 3) ? 1 : 0;
    const std::array<int, 8> key{ks, stride, c.Cin, c.Cout, c.Hin, c.Win, B, c.accum};
    auto it = p->autotune ? p->tuned.find(key) : p->tuned.end();
    if (it == p->tuned.end() && !p->measuring) {
        if (g_opt_use_tuned) {
            // option "train_table_batch" = n > 0: look the table up as if the batch were n (the rows are keyed on the batch they
            // were measured at; a parity test of the timed configuration's kernels on a batch the CPU oracle can afford pins n = 8)
            std::array<int, 8> tkey = key;
            if (g_opt_train_table_batch > 0) tkey[6] = g_opt_train_table_batch;
            for (const TrainTuned &t : kTrainTuned)
                if (t.wm && std::equal(tkey.begin(), tkey.end(), t.key)) {
                    const int rc = launch_conv_dma(c, ks, stride, B, s, t.wm, t.nt);
                    if (rc != PF_EUNSUPPORTED) {
                        ++p->stats[0];
                        return rc;
                    }
                    break;      // a row this build has no kernel for: the cost model's shape
                }
        }
        ++p->stats[1];
        return launch_conv_dma(c, ks, stride, B, s);
    }
    if (it == p->tuned.end()) {
        if (!p->tune_ev[0] && (hipEventCreate(&p->tune_ev[0]) != hipSuccess || hipEventCreate(&p->tune_ev[1]) != hipSuccess))
            return fail(PF_EHIP, "autotune: hipEventCreate failed");
        // candidate (0, 0) = the cost model's own choice: measured first, kept unless a forced shape is at least 3 % faster
        float model_ms = 0.f, best = 1e30f;
        std::pair<int, int> pick{0, 0};
        const int wms[4] = {0, 4, 2, 1};
        for (int wi = 0; wi < 4; ++wi) {
            const int wm = wms[wi], ntmax = wm == 0 ? 1 : (wm == 1 ? 2 : 4);
            for (int nt = 1; nt <= ntmax && (wm == 0 || nt <= c.ntiles); ++nt) {
                int rc = PF_OK;
                for (int rep = 0; rep < 4 && rc == PF_OK; ++rep) {      // 1 warm-up + 3 timed
                    if (rep == 1) PF_HIP_CHECK(hipEventRecord(p->tune_ev[0], s));
                    rc = wm ? launch_conv_dma(c, ks, stride, B, s, wm, nt) : launch_conv_dma(c, ks, stride, B, s);
                }
                if (rc == PF_EUNSUPPORTED && wm) continue;
                if (rc) return rc;
                PF_HIP_CHECK(hipEventRecord(p->tune_ev[1], s));
                PF_HIP_CHECK(hipEventSynchronize(p->tune_ev[1]));
                float ms = 0.f;
                PF_HIP_CHECK(hipEventElapsedTime(&ms, p->tune_ev[0], p->tune_ev[1]));
                if (!wm) model_ms = ms;
                else if (ms < best) {
                    best = ms;
                    pick = {wm, nt};
                }
            }
        }
        if (!(best < 0.97f * model_ms)) pick = {0, 0};
        it = p->tuned.emplace(key, pick).first;
    }
    ++p->stats[2];
    return it->second.first ? launch_conv_dma(c, ks, stride, B, s, it->second.first, it->second.second) : launch_conv_dma(c, ks, stride, B, s);
}

struct TDims {
    int h = 0, w = 0;
};

int t_propagate(const pf_train *p, int H, int W, std::vector<TDims> &d) {
    d.assign(p->tensors.size(), TDims());
    d[p->ops[0].src[0].tensor] = {H, W};
    for (const BlobOp &o : p->ops) {
        const TDims in = d[o.src[0].tensor];
        if (in.h <= 0 || in.w <= 0) return fail(PF_EBLOB, "op reads a tensor that was never produced");
        TDims out = in;
        if (o.kind == OP_STEM || o.kind == OP_CONV) {
            const int pad = o.k / 2;
            out.h = (in.h + 2 * pad - (int)o.k) / (int)o.stride + 1;
            out.w = (in.w + 2 * pad - (int)o.k) / (int)o.stride + 1;
        } else if (o.kind == OP_POOL) {
            out = {in.h / 2, in.w / 2};
        } else if (o.kind == OP_UPSAMPLE) {
            out = d[o.src[1].tensor];
        }
        if (out.h <= 0 || out.w <= 0) return fail(PF_EINVAL, "input %dx%d is too small for this network", H, W);
        if (o.kind != OP_HEAD) d[o.dst] = out;
    }
    return PF_OK;
}

// Who writes a tensor's gradient first?  The backward pass visits the ops in reverse; every consumer of a tensor adds its
// contribution to the tensor's gradient.  The first visitor of a channel range STORES instead (no cleared arena needed: the
// arena is as large as all activations, 0.2 ms of fill per step at batch 8 of 800x800); a range that is only partly fresh keeps
// the add and has its fresh channels cleared beforehand, as are channels no consumer ever writes (their producer reads them).
struct GradFirst {
    std::vector<std::array<uint8_t, kMaxSrc>> store;     // per op, per input range (conv / stem) or [0] (pool, upsample, head)
    std::vector<std::array<int, 3>> clear;               // (tensor, first channel, channels)
};
GradFirst grad_first_writers(const pf_train *p, const std::vector<TDims> &d) {
    GradFirst g;
    g.store.assign(p->ops.size(), std::array<uint8_t, kMaxSrc>{});
    const uint32_t input = p->ops[0].src[0].tensor;
    std::vector<std::vector<uint8_t>> touched(p->tensors.size());
    for (size_t t = 0; t < p->tensors.size(); ++t) touched[t].assign(p->tensors[t].channels, 0);
    auto add_clear = [&](uint32_t t, const std::vector<uint8_t> &want) {       // runs of channels
        for (size_t c = 0; c < want.size();) {
            if (!want[c]) { ++c; continue; }
            size_t e = c;
            while (e < want.size() && want[e]) ++e;
            g.clear.push_back({(int)t, (int)c, (int)(e - c)});
            c = e;
        }
    };
    for (size_t ii = p->ops.size(); ii-- > 0;) {
        const BlobOp &o = p->ops[ii];
        const uint32_t nj = (o.kind == OP_STEM || o.kind == OP_CONV) ? o.n_src : 1;
        for (uint32_t j = 0; j < nj; ++j) {
            const uint32_t t = o.src[j].tensor;
            if (t == input) continue;
            const bool whole = !(o.kind == OP_STEM || o.kind == OP_CONV);      // pool / upsample / head write the whole tensor
            const uint32_t c0 = whole ? 0 : o.src[j].choff, n = whole ? p->tensors[t].channels : o.src[j].ch;
            uint32_t fresh = 0;
            for (uint32_t c = c0; c < c0 + n; ++c) fresh += !touched[t][c];
            if (fresh == n) {
                g.store[ii][j] = 1;
            } else if (fresh) {
                std::vector<uint8_t> want(p->tensors[t].channels, 0);
                for (uint32_t c = c0; c < c0 + n; ++c) want[c] = !touched[t][c];
                add_clear(t, want);
            }
            for (uint32_t c = c0; c < c0 + n; ++c) touched[t][c] = 1;
        }
    }
    for (size_t t = 0; t < p->tensors.size(); ++t) {
        if (t == input || !d[t].h) continue;
        std::vector<uint8_t> want(p->tensors[t].channels, 0);
        bool any = false;
        for (size_t c = 0; c < want.size(); ++c) any |= (want[c] = !touched[t][c]) != 0;
        if (any) add_clear((uint32_t)t, want);
    }
    return g;
}

struct TLayout {
    std::vector<size_t> act, grad;     // per tensor (bytes); act[input] = the dense one-hot/depth tensor
    std::vector<size_t> ypre, stat;    // per op: pre-BN conv output, {mean[cout], invstd[cout]}
    std::vector<size_t> xpad;          // per op of an odd-width level: its gathered, row-padded input, kept from the forward pass for the weight gradient
    size_t dy = 0, wpk = 0, wpart = 0, dfull = 0, cepart = 0, bnpart = 0, out3 = 0, total = 0;
    size_t pad_in = 0, pad_out = 0;    // odd-width convs: gathered input / result with the row pitch rounded up to 4
    size_t dy_more[pf_train::kDySlots] = {}, wpart_more[pf_train::kSideStreams] = {};   // side streams: further dy slots, per stream the partial sums
    size_t up_tmp = 0;                 // scratch of the two-pass bilinear transpose
    size_t tune_grad = 0;              // autotune: the measuring pass's parameter gradients (discarded)
    size_t grad_begin = 0, grad_end = 0;
    // every tiled weight packing of the step (forward convs in op order, then the backward-data convs of every op and input
    // range in op order): packed by ONE batch of launches at the start of the step into wpk_arena
    std::vector<PackJob> jobs;
    std::vector<int> fwd_job;                  // per op: its forward job, or -1
    std::vector<std::vector<int>> bwd_job;     // per op, per input range: its backward-data job, or -1 (the network input needs none)
    std::vector<int> bwd_all_job;              // per op of an odd-width level: ONE backward-data conv over all input ranges, or -1
    size_t wpk_arena = 0;
    // forward pass on the packed-pair kernels (train_s4.hip): per tensor its shadow, per op its device-side weight packing
    std::vector<size_t> s4act;
    std::vector<int> s4_job;
    std::vector<S4WJob> s4jobs;
    size_t s4w_arena = 0;
};

// may op i's forward convolution run on conv_s4?  conv + BatchNorm, 3x3 / 1x1, stride 1, every input range at an even channel of a
// tensor some op produced (the network input stays dense fp32: the stem reads it)
bool s4_fwd_ok(const pf_train *p, size_t i) {
    const BlobOp &o = p->ops[i];
    if (!p->fwd_s4 || o.kind != OP_CONV || !p->bn[i] || o.stride != 1 || (o.k != 1 && o.k != 3) || o.n_src > (uint32_t)kConvMaxSrc) return false;
    const uint32_t input = p->ops[0].src[0].tensor;
    for (uint32_t j = 0; j < o.n_src; ++j)
        if ((o.src[j].choff & 1) || o.src[j].tensor == input) return false;
    return true;
}

TLayout t_layout(const pf_train *p, int B, const std::vector<TDims> &d, int out_h, int out_w) {
    TLayout L;
    const size_t nt = p->tensors.size();
    L.act.assign(nt, (size_t)-1);
    L.grad.assign(nt, (size_t)-1);
    L.ypre.assign(p->ops.size(), (size_t)-1);
    L.stat.assign(p->ops.size(), (size_t)-1);
    L.xpad.assign(p->ops.size(), (size_t)-1);
    size_t cur = 0;
    auto take = [&](size_t bytes) {
        const size_t o = cur;
        cur += align_up(bytes, 256);
        return o;
    };
    auto tbytes = [&](size_t t) { return (size_t)B * p->tensors[t].channels * d[t].h * d[t].w * sizeof(float); };
    for (size_t t = 0; t < nt; ++t)
        if (d[t].h) L.act[t] = take(tbytes(t));
    L.grad_begin = cur;
    const uint32_t input = p->ops[0].src[0].tensor;
    for (size_t t = 0; t < nt; ++t)
        if (d[t].h && t != input) L.grad[t] = take(tbytes(t));
    L.grad_end = cur;
    size_t max_dy = 0, max_wpk = 0, max_wpart = 0, max_c = 16, max_pin = 256, max_pout = 256;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const BlobOp &o = p->ops[i];
        if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
        const TDims in = d[o.src[0].tensor], out = d[o.dst];
        const size_t ybytes = (size_t)B * o.cout * out.h * out.w * sizeof(float);
        if (p->bn[i]) {
            // (odd width, stride 1: y keeps the tiled kernel's padded rows)
            L.ypre[i] = take((in.w & 3) && o.stride == 1 ? (size_t)B * o.cout * out.h * ((out.w + 3) / 4 * 4) * sizeof(float) : ybytes);
            L.stat[i] = take(2 * (size_t)o.cout * sizeof(float));
        }
        // dy scratch: the conv-output gradient, and (stride 2) its zero-stuffed copy at the input resolution
        size_t need = ybytes;
        if ((in.w & 3) && o.stride == 1) need = (size_t)B * o.cout * out.h * ((out.w + 3) / 4 * 4) * sizeof(float);   // written with padded rows
        if (o.stride == 2) need += align_up((size_t)B * o.cout * in.h * in.w * sizeof(float), 256);
        max_dy = need > max_dy ? need : max_dy;
        const ConvTiling tf = choose_tiling((int)o.k, (int)o.stride, (int)o.cin, (int)o.cout, 0);
        max_wpk = tf.packed_floats() > max_wpk ? tf.packed_floats() : max_wpk;
        int src_ch[kMaxSrc];
        for (uint32_t j = 0; j < o.n_src; ++j) src_ch[j] = (int)o.src[j].ch;
        size_t tp = tiled_packed_floats(src_ch, (int)o.n_src, (int)o.cout, (int)o.k, (int)o.stride);
        max_wpk = tp > max_wpk ? tp : max_wpk;
        for (uint32_t j = 0; j < o.n_src; ++j) {
            const ConvTiling tb = choose_tiling((int)o.k, 1, (int)o.cout, (int)o.src[j].ch, 0);
            max_wpk = tb.packed_floats() > max_wpk ? tb.packed_floats() : max_wpk;
            const int one = (int)o.cout;
            tp = tiled_packed_floats(&one, 1, (int)o.src[j].ch, (int)o.k, 1);
            max_wpk = tp > max_wpk ? tp : max_wpk;
        }
        if ((in.w & 3) && o.stride == 1) L.xpad[i] = take((size_t)B * o.cin * in.h * ((in.w + 3) / 4 * 4) * sizeof(float));
        if (in.w & 3) {     // padded copies (forward: cin -> cout at the output size; backward-data: cout -> cin at the input size)
            const size_t wp_in = (size_t)(in.w + 3) / 4 * 4, cmax = o.cin > o.cout ? o.cin : o.cout;
            const size_t bytes = (size_t)B * cmax * in.h * wp_in * sizeof(float);
            max_pin = bytes > max_pin ? bytes : max_pin;
            max_pout = bytes > max_pout ? bytes : max_pout;
        }
        const int wpi = o.stride == 1 ? (in.w + 3) / 4 * 4 : in.w, wpo = o.stride == 1 ? (out.w + 3) / 4 * 4 : out.w;   // (padded copies for odd widths)
        const size_t wp = wgrad_partial_floats((int)o.cout, (int)o.cin, (int)o.k, B, out.h, wpi, wpo);
        max_wpart = wp > max_wpart ? wp : max_wpart;
        max_c = o.cout > max_c ? o.cout : max_c;
    }
    // the step's packing jobs (the padded copies of odd-width levels read ONE gathered range)
    {
        const uint32_t input_t = p->ops[0].src[0].tensor;
        L.fwd_job.assign(p->ops.size(), -1);
        L.bwd_job.assign(p->ops.size(), std::vector<int>());
        L.bwd_all_job.assign(p->ops.size(), -1);
        size_t arena = 0;
        auto add = [&](size_t w_off, int cin_f, int cout_f, int ks, int stride, const int *chs, int ns, int tflip, int c0, int ch) {
            PackJob q;
            pack_job_fill(q, w_off, cin_f, cout_f, ks, stride, chs, ns, tflip, c0, ch, arena);
            arena += align_up((size_t)q.total, 64);
            L.jobs.push_back(q);
            return (int)L.jobs.size() - 1;
        };
        for (size_t i = 0; i < p->ops.size(); ++i) {
            const BlobOp &o = p->ops[i];
            if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
            const TDims in = d[o.src[0].tensor];
            int src_ch[kMaxSrc];
            for (uint32_t j = 0; j < o.n_src; ++j) src_ch[j] = (int)o.src[j].ch;
            const int one = (int)o.cin;
            const bool aligned = (in.w & 3) == 0;
            L.fwd_job[i] = add(p->w_off[i], (int)o.cin, (int)o.cout, (int)o.k, (int)o.stride, aligned ? src_ch : &one, aligned ? (int)o.n_src : 1, 0, 0, 0);
        }
        for (size_t i = 0; i < p->ops.size(); ++i) {
            const BlobOp &o = p->ops[i];
            if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
            L.bwd_job[i].assign(o.n_src, -1);
            int c0 = 0;
            const int dy_ch = (int)o.cout;
            if ((d[o.src[0].tensor].w & 3) && o.stride == 1) {
                L.bwd_all_job[i] = add(p->w_off[i], (int)o.cin, (int)o.cout, (int)o.k, 1, &dy_ch, 1, 1, 0, (int)o.cin);
                continue;
            }
            for (uint32_t j = 0; j < o.n_src; ++j) {
                if (o.src[j].tensor != input_t) L.bwd_job[i][j] = add(p->w_off[i], (int)o.cin, (int)o.cout, (int)o.k, 1, &dy_ch, 1, 1, c0, (int)o.src[j].ch);
                c0 += (int)o.src[j].ch;
            }
        }
        L.wpk_arena = take(arena * sizeof(float) + 256);
    }
    // the forward convolutions that run on the packed-pair kernels: their device-side weight packings, and the shadows of the tensors they read
    {
        L.s4act.assign(nt, (size_t)-1);
        L.s4_job.assign(p->ops.size(), -1);
        std::vector<char> need(nt, 0);
        size_t arena = 0;
        for (size_t i = 0; i < p->ops.size(); ++i) {
            if (!s4_fwd_ok(p, i)) continue;
            const BlobOp &o = p->ops[i];
            S4Range rg[kConvMaxSrc];
            for (uint32_t j = 0; j < o.n_src; ++j) {
                rg[j] = S4Range{(int)o.src[j].choff, (int)o.src[j].ch};
                need[o.src[j].tensor] = 1;
            }
            S4WJob jb;
            const size_t fl = s4_wjob_init(jb, p->w_off[i], (int)o.cin, (int)o.cout, (int)o.k, rg, (int)o.n_src, o.k == 1 && o.n_src == 2);
            jb.out_off = (unsigned)arena;
            arena += align_up(fl, 64);
            L.s4_job[i] = (int)L.s4jobs.size();
            L.s4jobs.push_back(jb);
        }
        for (size_t t = 0; t < nt; ++t)
            if (need[t] && d[t].h) L.s4act[t] = take((size_t)B * 2 * ((p->tensors[t].channels + 3) / 4) * d[t].h * ((d[t].w + 3) / 4 * 4) * 8);
        L.s4w_arena = take(arena * sizeof(float) + 256);
    }
    if (p->autotune) L.tune_grad = take(p->n_params * sizeof(float));
    L.dy = take(max_dy + 256);
    if (p->side) {
        for (int k = 1; k < pf_train::kDySlots; ++k) L.dy_more[k] = take(max_dy + 256);
    }
    L.wpk = take(max_wpk * sizeof(float));
    L.wpart = take(max_wpart * sizeof(float));
    if (p->side)
        for (int k = 1; k < pf_train::kSideStreams; ++k) L.wpart_more[k] = take(max_wpart * sizeof(float));
    L.pad_in = take(max_pin);
    L.pad_out = take(max_pout);
    L.dfull = take((size_t)B * p->hdr.n_cls * out_h * out_w * sizeof(float));
    {
        size_t mx = 0;
        for (size_t i = 0; i < p->ops.size(); ++i) {
            const BlobOp &o = p->ops[i];
            const TDims in = d[o.src[0].tensor];
            size_t n = 0;
            if (o.kind == OP_HEAD) n = upsample_bwd_tmp_floats(B * (int)o.cin, in.h, in.w, out_h, out_w);
            else if (o.kind == OP_UPSAMPLE) n = upsample_bwd_tmp_floats(B * (int)o.cin, in.h, in.w, d[o.dst].h, d[o.dst].w);
            mx = n > mx ? n : mx;
        }
        L.up_tmp = take(mx * sizeof(float) + 256);
    }
    L.cepart = take(ce_partial_doubles(B, out_h, out_w) * sizeof(double));
    L.bnpart = take(bn_partial_doubles((int)max_c) * sizeof(double));
    L.out3 = take(4 * sizeof(double));
    L.total = cur;
    return L;
}

}  // namespace

extern "C" int pf_train_create(const void *blob, size_t bytes, int in_ch, int n_cls, pf_train **out) {
    if (!blob || !out) return fail(PF_EINVAL, "pf_train_create: null argument");
    if (bytes < sizeof(BlobHeader)) return fail(PF_EBLOB, "blob shorter than its header");
    BlobHeader h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, kBlobMagic, 8) != 0 || h.version != kBlobVersion) return fail(PF_EBLOB, "bad blob magic/version");
    if (h.total_bytes != bytes || h.tensor_off + (uint64_t)h.n_tensors * sizeof(BlobTensor) > bytes ||
        h.op_off + (uint64_t)h.n_ops * sizeof(BlobOp) > bytes)
        return fail(PF_EBLOB, "blob table offsets out of range");
    if ((int)h.in_ch != in_ch || (int)h.n_cls != n_cls) return fail(PF_EINVAL, "blob is for in_ch=%u n_cls=%u", h.in_ch, h.n_cls);
    pf_train *p = new pf_train();
    p->hdr = h;
    p->tensors.resize(h.n_tensors);
    p->ops.resize(h.n_ops);
    memcpy(p->tensors.data(), (const char *)blob + h.tensor_off, h.n_tensors * sizeof(BlobTensor));
    memcpy(p->ops.data(), (const char *)blob + h.op_off, h.n_ops * sizeof(BlobOp));
    p->w_off.assign(h.n_ops, 0);
    p->aux_off.assign(h.n_ops, 0);
    p->bn.assign(h.n_ops, 0);
    size_t cur = 0;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const BlobOp &o = p->ops[i];
        bool ok = o.n_src >= 1 && o.n_src <= (uint32_t)kMaxSrc && o.dst < h.n_tensors;
        uint32_t cin = 0;
        for (uint32_t j = 0; ok && j < o.n_src; ++j) {
            ok = o.src[j].tensor < h.n_tensors && o.src[j].choff + o.src[j].ch <= p->tensors[o.src[j].tensor].channels;
            cin += o.src[j].ch;
        }
        if (!ok) {
            delete p;
            return fail(PF_EBLOB, "op %zu is inconsistent with the tensor table", i);
        }
        if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
        if (cin != o.cin || !((o.k == 3 && (o.stride == 1 || o.stride == 2)) || (o.k == 1 && o.stride == 1))) {
            delete p;
            return fail(PF_EUNSUPPORTED, "op %zu: training supports 3x3 (stride 1/2) and 1x1 convs", i);
        }
        // packing.py writes pad[0] = 1 + has_bn; blobs without the field: ConvLayer (conv + BN + ReLU) <=> relu flag
        p->bn[i] = o.pad[0] ? (int)o.pad[0] - 1 : (int)(o.relu != 0);
        p->w_off[i] = cur;
        cur += (size_t)o.cout * o.cin * o.k * o.k;
        p->aux_off[i] = cur;
        cur += (size_t)o.cout * (p->bn[i] ? 4 : 1);
    }
    p->n_params = cur;
    if (hipMalloc((void **)&p->dev_zero, 1024 * sizeof(float)) != hipSuccess || hipMemset(p->dev_zero, 0, 1024 * sizeof(float)) != hipSuccess) {
        delete p;
        return fail(PF_EHIP, "pf_train_create: device allocation failed");
    }
    p->fwd_s4 = g_opt_train_s4;
    if (g_opt_train_side) {
        int lo = 0, hi = 0;
        bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        for (int k = 0; ok && k < pf_train::kSideStreams; ++k) ok = hipStreamCreateWithPriority(&p->sides[k], hipStreamNonBlocking, lo) == hipSuccess;
        p->side = p->sides[0];
        for (int k = 0; ok && k < pf_train::kDySlots; ++k)
            ok = hipEventCreateWithFlags(&p->ev_dy[k], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&p->ev_wg[k], hipEventDisableTiming) == hipSuccess;
        for (int k = 0; ok && k < pf_train::kSideStreams; ++k) ok = hipEventCreateWithFlags(&p->ev_join[k], hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            pf_train_destroy(p);
            return fail(PF_EHIP, "pf_train_create: could not create the weight-gradient stream");
        }
    }
    *out = p;
    return PF_OK;
}

extern "C" void pf_train_destroy(pf_train *p) {
    if (!p) return;
    for (int k = 0; k < pf_train::kDySlots; ++k) {
        if (p->ev_dy[k]) (void)hipEventDestroy(p->ev_dy[k]);
        if (p->ev_wg[k]) (void)hipEventDestroy(p->ev_wg[k]);
    }
    for (hipEvent_t e : p->tune_ev)
        if (e) (void)hipEventDestroy(e);
    for (int k = 0; k < pf_train::kSideStreams; ++k) {
        if (p->ev_join[k]) (void)hipEventDestroy(p->ev_join[k]);
        if (p->sides[k]) (void)hipStreamDestroy(p->sides[k]);
    }
    if (p->dev_zero) (void)hipFree(p->dev_zero);
    delete p;
}

extern "C" int pf_train_autotune(pf_train *p, int enable) {
    if (!p) return fail(PF_EINVAL, "pf_train_autotune: null plan");
    p->autotune = enable != 0;      // (changes pf_train_workspace: the measuring pass has its own parameter-gradient buffer)
    if (!enable) {
        p->tuned.clear();
        p->measured_configs.clear();
    }
    return PF_OK;
}

extern "C" int pf_train_tuned_shapes(const pf_train *p, int *rows, int cap_rows, int *n_rows) {
    if (!p || !n_rows || (cap_rows > 0 && !rows)) return fail(PF_EINVAL, "pf_train_tuned_shapes: null argument");
    *n_rows = (int)p->tuned.size();
    int i = 0;
    for (const auto &kv : p->tuned) {
        if (i >= cap_rows) break;
        for (int k = 0; k < 8; ++k) rows[i * 10 + k] = kv.first[k];
        rows[i * 10 + 8] = kv.second.first;
        rows[i * 10 + 9] = kv.second.second;
        ++i;
    }
    return PF_OK;
}

extern "C" int pf_train_path_stats(const pf_train *p, int *stats, int cap, int *n) {
    if (!p || !n || (cap > 0 && !stats)) return fail(PF_EINVAL, "pf_train_path_stats: null argument");
    *n = 8;
    for (int k = 0; k < 8 && k < cap; ++k) stats[k] = p->stats[k];
    return PF_OK;
}

extern "C" int pf_train_param_count(const pf_train *p, size_t *n_floats) {
    if (!p || !n_floats) return fail(PF_EINVAL, "pf_train_param_count: null");
    *n_floats = p->n_params;
    return PF_OK;
}

extern "C" int pf_train_param_layout(const pf_train *p, int op_index, size_t *w_off, size_t *aux_off, int *has_bn) {
    if (!p || op_index < 0 || (size_t)op_index >= p->ops.size() || !w_off || !aux_off || !has_bn) return fail(PF_EINVAL, "pf_train_param_layout: bad argument");
    const BlobOp &o = p->ops[op_index];
    if (o.kind != OP_STEM && o.kind != OP_CONV) return fail(PF_EINVAL, "op %d is not a convolution", op_index);
    *w_off = p->w_off[op_index];
    *aux_off = p->aux_off[op_index];
    *has_bn = p->bn[op_index];
    return PF_OK;
}

extern "C" int pf_train_workspace(const pf_train *p, int B, int H, int W, int out_h, int out_w, size_t *bytes) {
    if (!p || !bytes || B <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return fail(PF_EINVAL, "pf_train_workspace: bad argument");
    std::vector<TDims> d;
    int rc = t_propagate(p, H, W, d);
    if (rc) return rc;
    *bytes = t_layout(p, B, d, out_h, out_w).total;
    return PF_OK;
}

extern "C" int pf_train_tensor_view(const pf_train *p, const char *name, int want_grad, int B, int H, int W, int out_h, int out_w,
                                    size_t *ws_offset, int *channels, int *h, int *w) {
    if (!p || !name || !ws_offset || !channels || !h || !w) return fail(PF_EINVAL, "pf_train_tensor_view: null");
    std::vector<TDims> d;
    int rc = t_propagate(p, H, W, d);
    if (rc) return rc;
    const TLayout L = t_layout(p, B, d, out_h, out_w);
    for (size_t t = 0; t < p->tensors.size(); ++t)
        if (strncmp(p->tensors[t].name, name, sizeof(p->tensors[t].name)) == 0) {
            const size_t off = want_grad ? L.grad[t] : L.act[t];
            if (off == (size_t)-1) return fail(PF_EINVAL, "tensor '%s' has no %s buffer", name, want_grad ? "gradient" : "activation");
            *ws_offset = off; *channels = (int)p->tensors[t].channels; *h = d[t].h; *w = d[t].w;
            return PF_OK;
        }
    return fail(PF_EINVAL, "no tensor named '%s'", name);
}

static int train_pass(const pf_train *p, float *theta, float *grad, int accumulate_grads, const void *seg, int seg_is_i64,
                      const float *depth, const uint8_t *depth_mask, float depth_mean, float depth_std, int T,
                      const float *x_dense, int B, int H, int W, const void *labels, int labels_i64, int out_h, int out_w,
                      int ignore_index, float loss_scale, float bn_momentum, float bn_eps, int update_running_stats,
                      double *out3, void *ws, size_t ws_bytes, void *stream) {
    if (!p || !theta || !grad || !labels || !out3 || !ws) return fail(PF_EINVAL, "pf_train_forward_backward: null pointer argument");
    if (!x_dense && (!seg || !depth || !depth_mask)) return fail(PF_EINVAL, "pf_train_forward_backward: pass seg+depth+depth_mask or x_dense");
    if (B <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return fail(PF_EINVAL, "pf_train_forward_backward: bad dims");
    hipStream_t s = (hipStream_t)stream;
    std::vector<TDims> d;
    int rc = t_propagate(p, H, W, d);
    if (rc) return rc;
    const TLayout L = t_layout(p, B, d, out_h, out_w);
    if (ws_bytes < L.total) return fail(PF_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, L.total);
    char *wsb = (char *)ws;
    for (int &v : p->stats) v = 0;
    // Every exit of this function - an error return in the middle of the backward pass included - leaves the side streams
    // JOINED to the caller's stream: a fork that is never joined invalidates an enclosing stream capture and lets the caller's
    // stream run ahead of weight gradients still in flight.
    struct SideJoin {
        const pf_train *p;
        hipStream_t s;
        int forked = 0;         // number of layers handed to the side streams so far
        bool done = false;
        int join() {
            if (done || !p->side || forked == 0) return PF_OK;
            done = true;
            for (int k = 0; k < pf_train::kSideStreams && k < forked; ++k) {   // (a stream that got no layer was never forked)
                PF_HIP_CHECK(hipEventRecord(p->ev_join[k], p->sides[k]));
                PF_HIP_CHECK(hipStreamWaitEvent(s, p->ev_join[k], 0));
            }
            return PF_OK;
        }
        ~SideJoin() { (void)join(); }
    } side_join{p, s};
    auto act = [&](uint32_t t) { return reinterpret_cast<float *>(wsb + L.act[t]); };
    auto gradt = [&](uint32_t t) { return reinterpret_cast<float *>(wsb + L.grad[t]); };
    const uint32_t input = p->ops[0].src[0].tensor;
    const int in_ch = (int)p->tensors[input].channels, n_cls = (int)p->hdr.n_cls;
    float *dy_slot[pf_train::kDySlots];
    for (int k = 0; k < pf_train::kDySlots; ++k) dy_slot[k] = reinterpret_cast<float *>(wsb + (p->side && k ? L.dy_more[k] : L.dy));
    float *wpk = reinterpret_cast<float *>(wsb + L.wpk);
    float *wpart = reinterpret_cast<float *>(wsb + L.wpart);
    float *pad_in = reinterpret_cast<float *>(wsb + L.pad_in), *pad_out = reinterpret_cast<float *>(wsb + L.pad_out);
    float *gather_to = pad_in;      // where run_conv's odd-width path puts its gathered input (forward: the op's kept copy)
    bool keep_padded = false;       // run_conv's odd-width path leaves its result in a.dst with padded rows (forward, conv + BN: the
                                    // BatchNorm kernels read y with that pitch, forward and backward - no unpad pass)
    float *dfull = reinterpret_cast<float *>(wsb + L.dfull);
    double *cepart = reinterpret_cast<double *>(wsb + L.cepart);
    float *up_tmp = g_opt_up_two_pass ? reinterpret_cast<float *>(wsb + L.up_tmp) : nullptr;
    double *bnpart = reinterpret_cast<double *>(wsb + L.bnpart);
    double *loss3 = reinterpret_cast<double *>(wsb + L.out3);

    // ---- input tensor (bg_model.py:61-69)
    if (x_dense) {
        if ((rc = launch_copy(act(input), x_dense, (size_t)B * in_ch * H * W * sizeof(float), s))) return rc;
    } else {
        if (T * (n_cls + 1) != in_ch) return fail(PF_EINVAL, "T=%d frames x (%d classes + depth) != %d input channels", T, n_cls, in_ch);
        if ((rc = launch_onehot_dense(seg, seg_is_i64, depth, depth_mask, depth_mean, depth_std, B, T, n_cls, H, W, act(input), s))) return rc;
    }
    float *wpk_arena = reinterpret_cast<float *>(wsb + L.wpk_arena);
    if ((rc = launch_pack_weights_batch(theta, wpk_arena, L.jobs.data(), (int)L.jobs.size(), s))) return rc;   // theta does not move inside this call
    float *s4w_arena = reinterpret_cast<float *>(wsb + L.s4w_arena);
    if (!L.s4jobs.empty() && (rc = launch_s4_pack_weights_dev(theta, s4w_arena, L.s4jobs.data(), (int)L.s4jobs.size(), kS4TrainWeightScale, s))) return rc;
    // the packed-pair shadow of channels [c0, c1) of tensor t, for the convolutions that read it through conv_s4
    // (a slice that starts or ends in the middle of a 4-channel group also zero-fills the group's other half while no producer has
    //  written it in this pass: a reader of this slice multiplies those channels by zero weights, and 0 x a stale NaN pattern is NaN)
    std::vector<std::vector<char>> s4_written(p->tensors.size());
    auto shadow = [&](uint32_t t, int c0, int c1) -> int {
        if (L.s4act[t] == (size_t)-1) return PF_OK;
        const int ct = (int)p->tensors[t].channels;
        std::vector<char> &wr = s4_written[t];
        if (wr.empty()) wr.assign((size_t)ct + 4, 0);
        const int fill_lo = (c0 & 2) && !wr[c0 - 2], fill_up = (c1 & 3) == 2 && c1 < ct && !wr[c1];
        for (int c = c0; c < c1; ++c) wr[c] = 1;
        return launch_s4_pack_act(act(t), B, ct, c0, c1, fill_lo, fill_up, d[t].h, d[t].w, (d[t].w + 3) / 4 * 4, wsb + L.s4act[t], kS4TrainActScale, s);
    };
    const GradFirst gfirst = grad_first_writers(p, d);      // (no cleared gradient arena: first writers store)
    for (const auto &c : gfirst.clear)
        if ((rc = launch_zero_channels(gradt((uint32_t)c[0]), B, (int)p->tensors[c[0]].channels, c[1], c[2], (long long)d[c[0]].h * d[c[0]].w, s))) return rc;
    if (!accumulate_grads && (rc = launch_zero_fill(grad, p->n_params * sizeof(float), s))) return rc;

    auto conv_args = [&](const BlobOp &o, const TDims &in, const TDims &out, ConvArgs &a) {
        memset(&a, 0, sizeof(a));
        a.n_src = (int)o.n_src;
        int c0 = 0;
        for (int j = 0; j < a.n_src; ++j) {
            a.src[j] = act(o.src[j].tensor);
            a.src_ctotal[j] = (int)p->tensors[o.src[j].tensor].channels;
            a.src_choff[j] = (int)o.src[j].choff;
            a.src_cstart[j] = c0;
            c0 += (int)o.src[j].ch;
        }
        for (int j = a.n_src; j <= kConvMaxSrc; ++j) a.src_cstart[j] = c0;
        a.bias = p->dev_zero;
        a.Cin = (int)o.cin; a.Cout = (int)o.cout;
        a.Hin = in.h; a.Win = in.w; a.Hout = out.h; a.Wout = out.w;
        a.zero_page = p->dev_zero;
        a.ntiles = ((int)o.cout + 15) / 16;
        a.src_end = a.n_src;
    };

    // one convolution: a = sources / destination / shapes filled, weights = OIHW in theta.  fwd: the op's own conv (input
    // ranges src_ch); else the backward-data conv of forward input range [c0, c0 + ch) (one range: the cout_f channels of dy)
    // job: the conv's entry in L.jobs (its tiled packing is already in the arena), or -1
    auto run_conv = [&](ConvArgs &a, int ks, int stride, const float *w, int cin_f, int cout_f, const int *src_ch, int n_src, int tflip, int c0,
                        int ch, int job) -> int {
        int rc2 = PF_EUNSUPPORTED;
        auto fast = [&](ConvArgs &c, const int *chs, int ns) -> int {
            if (job >= 0 && L.jobs[job].n_src == ns) {
                c.wpk = wpk_arena + L.jobs[job].out_off;
            } else {
                int r2 = launch_pack_weights_tiled(w, cin_f, cout_f, ks, stride, chs, ns, tflip, c0, ch, wpk, s);
                if (r2) return r2;
                c.wpk = wpk;
            }
            const int kc = dma_kc(ks, stride);
            c.src_chunk0[0] = 0;
            for (int j = 0; j < kConvMaxSrc; ++j) c.src_chunk0[j + 1] = c.src_chunk0[j] + (j < ns ? (chs[j] + kc - 1) / kc : 0);
            c.nchunks = c.src_chunk0[ns];
            c.chunk_begin = 0;
            c.chunk_end = c.nchunks;
            c.rem = 0;
            return train_conv_dma(p, c, ks, stride, B, s);
        };
        if ((a.Win & 3) == 0) {
            rc2 = fast(a, src_ch, n_src);
            if (rc2 != PF_EUNSUPPORTED) return rc2;
        } else {
            // odd width: the same kernels on copies whose rows are padded to a multiple of 4 (train_kernels.hip)
            const int Wp = (a.Win + 3) / 4 * 4, pad = ks / 2;
            const int Wop = (Wp + 2 * pad - ks) / stride + 1;
            if ((rc2 = launch_pad_gather(a, B, Wp, gather_to, s))) return rc2;
            ++p->stats[3];
            ConvArgs c = a;
            c.n_src = 1;
            c.src[0] = gather_to; c.src_ctotal[0] = a.Cin; c.src_choff[0] = 0; c.src_cstart[0] = 0;
            for (int k = 1; k <= kConvMaxSrc; ++k) c.src_cstart[k] = a.Cin;
            c.src_begin = 0; c.src_end = 1;
            c.Win = Wp; c.Wout = Wop;
            c.dst = keep_padded ? a.dst : pad_out; c.dst_ctotal = a.Cout; c.dst_choff = 0; c.accum = 0;
            const int one = a.Cin;
            rc2 = fast(c, &one, 1);
            if (rc2 == PF_OK && keep_padded) {
                ++p->stats[4];
                return PF_OK;
            }
            if (keep_padded && rc2 == PF_EUNSUPPORTED) return fail(PF_EUNSUPPORTED, "training: no tiled kernel for an odd-width conv + BatchNorm layer");
            if (rc2 == PF_OK) return launch_unpad_scatter(pad_out, B, a.Cout, a.Hout, a.Wout, Wop, a.dst, a.dst_ctotal, a.dst_choff, a.accum, s);
            if (rc2 != PF_EUNSUPPORTED) return rc2;
        }
        ++p->stats[6];      // the generic register-staged kernel (no tiled kernel took the geometry)
        const ConvTiling t = choose_tiling(ks, stride, tflip ? cout_f : cin_f, tflip ? ch : cout_f, 0);
        if ((rc2 = launch_pack_weights(w, cin_f, cout_f, t, tflip, c0, ch, wpk, s))) return rc2;
        a.wpk = wpk;
        a.nchunks = t.nchunks;
        return launch_conv(a, t, B, s);
    };

    // ================================================================ forward (training mode)
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const BlobOp &o = p->ops[i];
        const TDims in = d[o.src[0].tensor];
        const TDims out = o.kind == OP_HEAD ? in : d[o.dst];
        if (o.kind == OP_STEM || o.kind == OP_CONV) {
            int src_ch[kMaxSrc];
            for (uint32_t j = 0; j < o.n_src; ++j) src_ch[j] = (int)o.src[j].ch;
            ConvArgs a;
            conv_args(o, in, out, a);
            gather_to = L.xpad[i] != (size_t)-1 ? reinterpret_cast<float *>(wsb + L.xpad[i]) : pad_in;
            if (p->bn[i]) {
                float *y = reinterpret_cast<float *>(wsb + L.ypre[i]);
                a.dst = y; a.dst_ctotal = (int)o.cout; a.dst_choff = 0; a.relu = 0;
                const bool odd_f = (in.w & 3) != 0 && o.stride == 1;
                if (L.s4_job[i] >= 0) {
                    // train_s4.hip: conv_s4 on the shadows of the input ranges; y in fp32 (rows padded to 4 on an odd-width level,
                    // as the tiled fp32 path leaves them).  The weight gradient of an odd-width layer still reads the gathered,
                    // row-padded fp32 copy of x
                    const S4WJob &jb = L.s4jobs[L.s4_job[i]];
                    const int Wp = (in.w + 3) / 4 * 4, per = o.k == 3 ? 2 : 8;
                    if (odd_f) {
                        if ((rc = launch_pad_gather(a, B, Wp, gather_to, s))) return rc;
                        ++p->stats[3];
                        ++p->stats[4];
                    }
                    ConvArgs c = a;
                    int e = 0;
                    for (int j = 0; j < c.n_src; ++j) {
                        const int ch0 = (int)o.src[j].choff, chn = (int)o.src[j].ch;
                        c.src[j] = reinterpret_cast<const float *>(wsb + L.s4act[o.src[j].tensor]);
                        c.src_c4[j] = (c.src_ctotal[j] + 3) / 4;
                        c.src_g0[j] = ch0 / 4;
                        c.src_gn[j] = (ch0 + chn + 3) / 4 - ch0 / 4;
                        c.src_ent0[j] = e;
                        e += jb.pad ? (c.src_gn[j] + per - 1) / per * per : c.src_gn[j];
                    }
                    for (int j = c.n_src; j <= kConvMaxSrc; ++j) c.src_ent0[j] = e;
                    c.src_fmt = 1;
                    c.dst_fmt = 0;
                    c.src_begin = 0;
                    c.Win = Wp; c.Wout = Wp;
                    c.acc_scale = 1.0f / (kS4TrainWeightScale * kS4TrainActScale);
                    c.wpk = s4w_arena + jb.out_off;
                    c.nchunks = jb.rounds;
                    c.chunk_begin = 0;
                    c.chunk_end = jb.rounds;
                    c.kacc = (g_opt_train_kacc && o.k == 3) ? 1 : 0;      // blocked summation, as in the fp32 step (conv_s4_kernel.inc: KACC)
                    ConvChoice ch4;
                    int nt4 = c.ntiles == 3 ? 3 : (c.ntiles < 2 ? 1 : 2), wide4 = 0;
                    if (g_opt_use_tuned && choose_s4((int)o.k, c.Cin, c.Cout, c.Hout, c.Wout, B, &ch4) && ch4.kind == 5) { nt4 = ch4.p0; wide4 = ch4.p1; }
                    if ((rc = launch_conv_s4(c, (int)o.k, nt4, wide4, B, s))) return rc;
                    ++p->stats[7];
                } else {
                keep_padded = odd_f;
                rc = run_conv(a, (int)o.k, (int)o.stride, theta + p->w_off[i], (int)o.cin, (int)o.cout, src_ch, (int)o.n_src, 0, 0, 0, L.fwd_job[i]);
                keep_padded = false;
                if (rc) return rc;
                }
                float *aux = theta + p->aux_off[i];
                float *stat = reinterpret_cast<float *>(wsb + L.stat[i]);
                if ((rc = launch_bn_forward(y, B, (int)o.cout, out.h, out.w, bn_eps, bn_momentum, aux, aux + o.cout,
                                            update_running_stats ? aux + 2 * o.cout : nullptr, update_running_stats ? aux + 3 * o.cout : nullptr,
                                            stat, stat + o.cout, bnpart, act(o.dst), (int)p->tensors[o.dst].channels, (int)o.dst_choff,
                                            (int)o.relu, odd_f ? (out.w + 3) / 4 * 4 : 0, s)))
                    return rc;
                if ((rc = shadow(o.dst, (int)o.dst_choff, (int)o.dst_choff + (int)o.cout))) return rc;
            } else {
                a.bias = theta + p->aux_off[i];
                a.dst = act(o.dst); a.dst_ctotal = (int)p->tensors[o.dst].channels; a.dst_choff = (int)o.dst_choff; a.relu = (int)o.relu;
                if ((rc = run_conv(a, (int)o.k, (int)o.stride, theta + p->w_off[i], (int)o.cin, (int)o.cout, src_ch, (int)o.n_src, 0, 0, 0, L.fwd_job[i]))) return rc;
                if ((rc = shadow(o.dst, (int)o.dst_choff, (int)o.dst_choff + (int)o.cout))) return rc;
            }
        } else if (o.kind == OP_POOL) {
            if ((rc = launch_avgpool2(act(o.src[0].tensor), act(o.dst), B * (int)o.cin, in.h, in.w, nullptr, nullptr, s))) return rc;
            if ((rc = shadow(o.dst, 0, (int)p->tensors[o.dst].channels))) return rc;
        } else if (o.kind == OP_UPSAMPLE) {
            if ((rc = launch_upsample(act(o.src[0].tensor), act(o.dst), B * (int)o.cin, in.h, in.w, out.h, out.w, nullptr, nullptr, s))) return rc;
            if ((rc = shadow(o.dst, 0, (int)p->tensors[o.dst].channels))) return rc;
        } else if (o.kind == OP_HEAD) {
            if ((rc = launch_ce_fwd_bwd(act(o.src[0].tensor), B, (int)o.cin, in.h, in.w, labels, labels_i64, out_h, out_w, ignore_index, dfull,
                                        cepart, loss3, s)))
                return rc;
            if ((rc = launch_copy(out3, loss3, 3 * sizeof(double), s))) return rc;
        }
    }

    // ================================================================ backward
    gather_to = pad_in;
    // With the side streams: layer n's conv-output gradient goes to dy slot n % kDySlots; the weight gradient (and its padded
    // copy of x) reads it on side stream n % kSideStreams while the caller's stream goes on to the input gradients and the next
    // layers; before it overwrites a slot it waits for the weight gradient of layer n - kDySlots that last read it.
    int n_conv = 0;
    for (size_t ii = p->ops.size(); ii-- > 0;) {
        const BlobOp &o = p->ops[ii];
        const TDims in = d[o.src[0].tensor];
        const TDims out = o.kind == OP_HEAD ? in : d[o.dst];
        if (o.kind == OP_HEAD) {
            // d loss / d logits = bilinear^T (softmax - onehot) * loss_scale / n_valid   (mean over the valid pixels, bg_model.py:81)
            if ((rc = launch_upsample_bwd(dfull, B * (int)o.cin, in.h, in.w, out_h, out_w, loss3 + 1, loss_scale, gfirst.store[ii][0] ? 0 : 1, gradt(o.src[0].tensor), up_tmp, s))) return rc;
        } else if (o.kind == OP_POOL) {
            if ((rc = launch_avgpool2_bwd(gradt(o.dst), B * (int)o.cin, in.h, in.w, gfirst.store[ii][0], gradt(o.src[0].tensor), s))) return rc;
        } else if (o.kind == OP_UPSAMPLE) {
            if ((rc = launch_upsample_bwd(gradt(o.dst), B * (int)o.cin, in.h, in.w, out.h, out.w, nullptr, 1.f, gfirst.store[ii][0] ? 0 : 1, gradt(o.src[0].tensor), up_tmp, s))) return rc;
        } else if (o.kind == OP_STEM || o.kind == OP_CONV) {
            const int t_ctotal = (int)p->tensors[o.dst].channels;
            float *aux = theta + p->aux_off[ii], *gaux = grad + p->aux_off[ii];
            const int slot = n_conv % pf_train::kDySlots, sidx = n_conv % pf_train::kSideStreams;
            float *dy = dy_slot[slot];
            // odd-width levels (stride 1): dy goes out with its rows padded to a multiple of 4 floats and zero pad columns - the
            // form the tiled weight-gradient and backward-data kernels read - and ONE backward-data conv covers all input ranges
            const bool odd = (in.w & 3) != 0 && o.stride == 1;
            const int Wp = (in.w + 3) / 4 * 4;
            if (p->side && n_conv >= pf_train::kDySlots) PF_HIP_CHECK(hipStreamWaitEvent(s, p->ev_wg[slot], 0));
            ++n_conv;
            if (p->bn[ii]) {
                const float *stat = reinterpret_cast<const float *>(wsb + L.stat[ii]);
                if ((rc = launch_bn_backward(gradt(o.dst), t_ctotal, (int)o.dst_choff, reinterpret_cast<const float *>(wsb + L.ypre[ii]),
                                             stat, stat + o.cout, aux, aux + o.cout, B, (int)o.cout, out.h, out.w, (int)o.relu, gaux, gaux + o.cout, bnpart,
                                             dy, odd ? Wp : 0, odd ? Wp : 0, s)))
                    return rc;
            } else {
                if (o.relu) return fail(PF_EUNSUPPORTED, "training: ReLU without BatchNorm (op %zu)", ii);
                if ((rc = launch_bias_backward(gradt(o.dst), t_ctotal, (int)o.dst_choff, B, (int)o.cout, out.h, out.w, gaux, bnpart, dy, odd ? Wp : 0, s))) return rc;
            }
            // dW
            hipStream_t sw = s;
            float *wpart_l = wpart;
            if (p->side) {
                PF_HIP_CHECK(hipEventRecord(p->ev_dy[slot], s));
                sw = p->sides[sidx];
                PF_HIP_CHECK(hipStreamWaitEvent(sw, p->ev_dy[slot], 0));
                side_join.forked = n_conv;
                if (sidx) wpart_l = reinterpret_cast<float *>(wsb + L.wpart_more[sidx]);
            }
            ConvArgs a;
            conv_args(o, in, out, a);
            if (prof_enabled()) {      // per-layer rows of tools/bench_train.py --layers
                char tag[64];
                snprintf(tag, sizeof(tag), "%02d %u->%u k%u s%u %dx%d", (int)ii, o.cin, o.cout, o.k, o.stride, out.h, out.w);
                prof_set_tag(tag);
            }
            if (odd) {
                // odd width: the tiled kernel on the padded copy of x the forward pass gathered (all ranges) and the padded dy; zero
                // pad columns add nothing
                ConvArgs ap = a;
                ap.n_src = 1;
                ap.src[0] = reinterpret_cast<float *>(wsb + L.xpad[ii]); ap.src_ctotal[0] = (int)o.cin; ap.src_choff[0] = 0; ap.src_cstart[0] = 0;
                for (int k = 1; k <= kConvMaxSrc; ++k) ap.src_cstart[k] = (int)o.cin;
                ap.Win = Wp; ap.Wout = Wp;
                if ((rc = launch_wgrad(ap, (int)o.k, 1, dy, B, wpart_l, grad + p->w_off[ii], sw))) return rc;
            } else if ((rc = launch_wgrad(a, (int)o.k, (int)o.stride, dy, B, wpart_l, grad + p->w_off[ii], sw))) {
                return rc;
            }
            if (prof_enabled()) prof_set_tag(nullptr);
            if (p->side) PF_HIP_CHECK(hipEventRecord(p->ev_wg[slot], sw));
            // dX per input range (the network input needs none)
            const float *dsrc = dy;
            if (o.stride == 2) {
                float *up = dy + align_up((size_t)B * o.cout * out.h * out.w * sizeof(float), 256) / sizeof(float);
                bool needed = false;
                for (uint32_t j = 0; j < o.n_src; ++j) needed = needed || o.src[j].tensor != input;
                if (needed && (rc = launch_zero_stuff(dy, B * (int)o.cout, out.h, out.w, in.h, in.w, up, s))) return rc;
                dsrc = up;
            }
            if (odd) {
                const PackJob &q = L.jobs[L.bwd_all_job[ii]];
                ConvArgs b;
                memset(&b, 0, sizeof(b));
                b.n_src = 1;
                b.src[0] = dy; b.src_ctotal[0] = (int)o.cout; b.src_choff[0] = 0; b.src_cstart[0] = 0;
                for (int k = 1; k <= kConvMaxSrc; ++k) b.src_cstart[k] = (int)o.cout;
                b.bias = p->dev_zero; b.zero_page = p->dev_zero;
                b.dst = pad_out; b.dst_ctotal = (int)o.cin; b.dst_choff = 0;
                b.Cin = (int)o.cout; b.Cout = (int)o.cin; b.Hin = in.h; b.Win = Wp; b.Hout = in.h; b.Wout = Wp;
                b.ntiles = ((int)o.cin + 15) / 16; b.src_end = 1;
                b.wpk = wpk_arena + q.out_off;
                const int kc = dma_kc((int)o.k, 1);
                b.src_chunk0[0] = 0;
                for (int j = 0; j < kConvMaxSrc; ++j) b.src_chunk0[j + 1] = j == 0 ? ((int)o.cout + kc - 1) / kc : b.src_chunk0[j];
                b.nchunks = b.src_chunk0[1];
                b.chunk_begin = 0;
                b.chunk_end = b.nchunks;
                if ((rc = train_conv_dma(p, b, (int)o.k, 1, B, s))) return rc;
                ++p->stats[5];
                float *dsts[kMaxSrc];
                int ct[kMaxSrc], co[kMaxSrc], chs[kMaxSrc], ow[kMaxSrc];
                for (uint32_t j = 0; j < o.n_src; ++j) {
                    ow[j] = gfirst.store[ii][j];
                    dsts[j] = o.src[j].tensor != input ? gradt(o.src[j].tensor) : nullptr;
                    ct[j] = (int)p->tensors[o.src[j].tensor].channels;
                    co[j] = (int)o.src[j].choff;
                    chs[j] = (int)o.src[j].ch;
                }
                if ((rc = launch_unpad_scatter_multi(pad_out, B, (int)o.cin, in.h, in.w, Wp, dsts, ct, co, chs, ow, (int)o.n_src, s))) return rc;
                continue;
            }
            int c0 = 0;
            for (uint32_t j = 0; j < o.n_src; ++j) {
                const int ch = (int)o.src[j].ch;
                if (o.src[j].tensor != input) {
                    ConvArgs b;
                    memset(&b, 0, sizeof(b));
                    b.n_src = 1;
                    b.src[0] = dsrc; b.src_ctotal[0] = (int)o.cout; b.src_choff[0] = 0; b.src_cstart[0] = 0;
                    for (int k = 1; k <= kConvMaxSrc; ++k) b.src_cstart[k] = (int)o.cout;
                    b.bias = p->dev_zero; b.zero_page = p->dev_zero;
                    b.dst = gradt(o.src[j].tensor); b.dst_ctotal = (int)p->tensors[o.src[j].tensor].channels; b.dst_choff = (int)o.src[j].choff;
                    b.Cin = (int)o.cout; b.Cout = ch; b.Hin = in.h; b.Win = in.w; b.Hout = in.h; b.Wout = in.w;
                    b.ntiles = (ch + 15) / 16; b.src_end = 1; b.accum = gfirst.store[ii][j] ? 0 : 1;
                    const int dy_ch = (int)o.cout;
                    if ((rc = run_conv(b, (int)o.k, 1, theta + p->w_off[ii], (int)o.cin, (int)o.cout, &dy_ch, 1, 1, c0, ch, L.bwd_job[ii][j]))) return rc;
                }
                c0 += ch;
            }
        }
    }
    return side_join.join();   // the caller's stream carries every gradient when this call's work is done
}

extern "C" int pf_train_forward_backward(const pf_train *p, float *theta, float *grad, int accumulate_grads, const void *seg, int seg_is_i64,
                                         const float *depth, const uint8_t *depth_mask, float depth_mean, float depth_std, int T,
                                         const float *x_dense, int B, int H, int W, const void *labels, int labels_i64, int out_h, int out_w,
                                         int ignore_index, float loss_scale, float bn_momentum, float bn_eps, int update_running_stats,
                                         double *out3, void *ws, size_t ws_bytes, void *stream) {
    if (p && p->autotune && ws) {
        const std::array<int, 5> cfg{B, H, W, out_h, out_w};
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool seen = std::find(p->measured_configs.begin(), p->measured_configs.end(), cfg) != p->measured_configs.end();
        if (!seen && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
            std::vector<TDims> d;
            int rc = t_propagate(p, H, W, d);
            if (rc) return rc;
            const TLayout L = t_layout(p, B, d, out_h, out_w);
            if (ws_bytes < L.total) return fail(PF_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, L.total);
            p->measuring = true;
            rc = train_pass(p, theta, reinterpret_cast<float *>((char *)ws + L.tune_grad), 0, seg, seg_is_i64, depth, depth_mask, depth_mean, depth_std, T,
                            x_dense, B, H, W, labels, labels_i64, out_h, out_w, ignore_index, loss_scale, bn_momentum, bn_eps, 0, out3, ws, ws_bytes, stream);
            p->measuring = false;
            if (rc) return rc;
            p->measured_configs.push_back(cfg);
        }
    }
    return train_pass(p, theta, grad, accumulate_grads, seg, seg_is_i64, depth, depth_mask, depth_mean, depth_std, T, x_dense, B, H, W, labels,
                      labels_i64, out_h, out_w, ignore_index, loss_scale, bn_momentum, bn_eps, update_running_stats, out3, ws, ws_bytes, stream);
}

extern "C" int pf_sgd_workspace(size_t *bytes) {
    if (!bytes) return fail(PF_EINVAL, "pf_sgd_workspace: null");
    *bytes = sgd_ws_bytes();
    return PF_OK;
}

extern "C" int pf_sgd_step(float *theta, float *grad, float *momentum_buf, const uint8_t *trainable, size_t n, float lr, float momentum,
                           float weight_decay, float clip_norm, float clip_value, int first_step, void *ws, size_t ws_bytes, void *stream) {
    if (!theta || !grad || !momentum_buf || !trainable || !ws || n == 0) return fail(PF_EINVAL, "pf_sgd_step: null argument");
    if (ws_bytes < sgd_ws_bytes()) return fail(PF_EWORKSPACE, "pf_sgd_step: workspace %zu B < required %zu B", ws_bytes, sgd_ws_bytes());
    return launch_sgd(theta, grad, momentum_buf, trainable, (long long)n, lr, momentum, weight_decay, clip_norm, clip_value, first_step, ws,
                      (hipStream_t)stream);
}
