// Plan = parsed op table + device-resident, MFMA-tiled weights; forward = walk the table and enqueue.
//
// Replaces the module walk of reference hardnet.py:353-387 (hardnet.forward) and the glue of
// bg_model.py:61-71,91-102.  The op table comes from the blob (packing.py / hardnet_arch.py); nothing
// about FC-HarDNet-70 is hard-coded here, so single-op test networks use the same code.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "conv_epilogue.h"
#include "net_kernels.h"
#include "pf_blob.h"
#include "pf_prof.h"

using namespace pf;

struct ConvPlan {
    ConvTiling tiling;
    size_t wpk_off = 0;   // floats into dev_weights
    size_t bias_off = 0;  // floats (padded to 16*n_tiles)
    size_t raw_off = 0;   // folded OIHW copy (stem only)
    size_t dep_off = 0;   // stem only: depth-channel columns [tap][t][16]
    size_t oh_off = 0;    // stem only: one-hot rows [tap][t][n_cls + 1][16] with a zero row per group
    bool has_oh = false;
    size_t tiled_off = 0; // per-cout-tile packing for the DMA fast path
    int tiled_chunks = 0;
    size_t rem_off = 0;   // conv_dma vector-ALU cout weights (3x3/s1 convs, rem_count trailing couts), else 0
    int rem_count = 0;
    size_t wave_off = 0;  // fragment-order packing for the wave-autonomous path (stride 1 only)
    int wave_chunks = 0;
    size_t split_off = 0; // fp16 hi/mid fragments for the split path (3x3/s1 and 1x1), else 0
    int split_chunks = 0;
    bool has_split = false;
    size_t s4_off = 0;    // conv_s4.hip packing (stride-1 3x3 and 1x1), else 0
    int s4_rounds = 0;
    bool has_s4 = false, s4_pad = false;   // s4_pad: ranges padded to whole rounds (the conv may run one range at a time)
    size_t front_off = 0; // conv_s4-style packing of a 3x3 STRIDE-2 conv with one input range (second conv of conv_front.hip), else 0
    bool has_front = false;
    float split_acc_scale = 1.0f;          // 2^-k: the split / S4 packings hold fp16 terms of w * 2^k (conv_mfma.h)
    // conv_pair.hip, kept on the CONSUMER (op i; its producer is op i - 1): the consumer's weights in the K order [S, others, P] with
    // every range padded to whole rounds, the producer's two-instruction stream and its ninth-tap stream
    bool has_pair = false;
    size_t pair_c_off = 0, pair_two_off = 0, pair_nine_off = 0;
    int pair_rounds = 0;
    bool pair_merged = false;              // pair_c_off carries P's weights in C's padding rows (conv_mfma.h: PairArgs::merged)
};

struct pf_plan {
    BlobHeader hdr;
    std::vector<BlobTensor> tensors;
    std::vector<BlobOp> ops;
    std::vector<ConvPlan> conv;  // parallel to ops
    std::vector<int> readers;    // per tensor: number of ops that read it
    float *dev_weights = nullptr;
    uint8_t *dev_lut = nullptr;
    size_t dev_floats = 0;
    // execution options of THIS plan (pf_hardnet_plan_set_option); initialised from the process-wide defaults
    // (pf_set_option) when the plan is created and read only by forwards of this plan
    int opt_fuse_pool = 1, opt_fuse_upsample = 1, opt_valu_rem = 1, opt_split = 1, opt_use_tuned = 1;
    int opt_table_batch = 0;   // > 0: per-layer kernel choice as if the batch were this (batch-invariant numerics)
    int opt_fuse_front = 0;    // stem -> conv_front.hip (3x3 s1 + 3x3 s2 in one kernel, the tensor between them never stored)
    int opt_range_guard = 1;   // kernels raise PF_STATUS_RANGE in the workspace's status word when they store |v| > 65504 while
                               // two-term fp16 operands are in use (conv_mfma.h); 0 = no checks (the clamp-free fp32 path needs none)
    int opt_packed_acts = 1;   // tensors whose producers and consumers all support it live in the S4 layout (conv_s4.hip)
    // formats of the last forward (pf_hardnet_tensor_read): 1 = S4, 0xFF = elided (never stored: conv_front.hip)
    mutable std::vector<uint8_t> last_fmt;
    // range normalisation (conv_mfma.h): tensor t, channel c is stored multiplied by chan_scale[t][c] (a power of two; 1 for
    // the network input, the head's input and every tensor no convolution reads); inv_scale_off[t] = offset in dev_weights
    // of the reciprocals (pf_hardnet_tensor_read), or 0 when all are 1
    std::vector<std::vector<float>> chan_scale;
    std::vector<size_t> inv_scale_off;
    std::vector<uint8_t> feeds_conv;   // per tensor: a convolution reads it (directly or through pool / upsample ops)
    int opt_normalize = 1;
    int opt_tag_ops = 0;       // pf_profile_* records carry one label per op of the table (tools/)
    int opt_fuse_pairs = 1;    // conv_pair.hip: an odd HarDBlock layer runs inside its consumer where both read / write packed pairs (0 = two launches)
};

namespace pf {
int g_opt_fuse_pool = 1, g_opt_fuse_upsample = 1, g_opt_valu_rem = 1, g_opt_split = 1, g_opt_packed_acts = 1, g_opt_tag_ops = 0;
int g_opt_range_guard = 1, g_opt_fuse_front = 1;   // fuse_front: conv_front.hip (0: stem -> conv_split -> conv_dma stride 2, three kernels)
int g_opt_fuse_pairs = 1;  // fuse_pairs: conv_pair.hip (0: every layer a launch of its own)
int g_opt_normalize = 1;   // normalize_ranges: per-channel power-of-two scaling of the stored activations, fixed at plan creation
extern int g_opt_use_tuned;
int g_opt_up_two_pass = 1;  // upsample_bwd_two_pass: the bilinear transposes of large planes as rows-then-columns passes (train_kernels.hip)
int g_opt_train_table_batch = 0;   // train_table_batch: batch size the rows of train_tuned.inc are looked up with (0 = the call's own)
int g_opt_train_kacc = 1;   // train_blocked_sum: per-round partial sums in the 3x3 convolutions of a training step (conv_dma.hip: KACC)
extern int g_opt_wgrad_taps;   // wgrad_taps (train_kernels.hip)
int g_opt_train_s4 = 0;         // train_forward_s4: the forward convolutions of a training step on conv_s4 with blocked sums (train_s4.hip); opt-in
int g_opt_train_side = 1;   // train_side_stream: weight gradients on the training plan's own stream (train_plan.hip)
}

extern "C" int pf_set_option(const char *name, int value) {
    if (!name) return fail(PF_EINVAL, "pf_set_option: null name");
    if (!strcmp(name, "fuse_pool")) g_opt_fuse_pool = value;
    else if (!strcmp(name, "fuse_upsample")) g_opt_fuse_upsample = value;
    else if (!strcmp(name, "use_tuned_table")) g_opt_use_tuned = value;
    else if (!strcmp(name, "valu_remainder")) g_opt_valu_rem = value;
    else if (!strcmp(name, "split_f16") || !strcmp(name, "split_bf16")) g_opt_split = value;   // (round-1 name kept)
    else if (!strcmp(name, "packed_acts")) g_opt_packed_acts = value;
    else if (!strcmp(name, "range_guard")) g_opt_range_guard = value;
    else if (!strcmp(name, "fuse_front")) g_opt_fuse_front = value;
    else if (!strcmp(name, "normalize_ranges")) g_opt_normalize = value;
    else if (!strcmp(name, "fuse_pairs")) g_opt_fuse_pairs = value;
    else if (!strcmp(name, "profile_tag_ops")) g_opt_tag_ops = value;
    else if (!strcmp(name, "train_side_stream")) g_opt_train_side = value;
    else if (!strcmp(name, "wgrad_taps")) g_opt_wgrad_taps = value;
    else if (!strcmp(name, "train_forward_s4")) g_opt_train_s4 = value;
    else if (!strcmp(name, "train_blocked_sum")) g_opt_train_kacc = value;
    else if (!strcmp(name, "train_table_batch")) g_opt_train_table_batch = value < 0 ? 0 : value;
    else if (!strcmp(name, "upsample_bwd_two_pass")) g_opt_up_two_pass = value;
    else return fail(PF_EINVAL, "pf_set_option: unknown option '%s'", name);
    return PF_OK;
}

extern "C" int pf_hardnet_plan_set_option(pf_plan *p, const char *name, int value) {
    if (!p || !name) return fail(PF_EINVAL, "pf_hardnet_plan_set_option: null argument");
    if (!strcmp(name, "fuse_pool")) p->opt_fuse_pool = value;
    else if (!strcmp(name, "fuse_upsample")) p->opt_fuse_upsample = value;
    else if (!strcmp(name, "use_tuned_table")) p->opt_use_tuned = value;
    else if (!strcmp(name, "valu_remainder")) p->opt_valu_rem = value;
    else if (!strcmp(name, "split_f16") || !strcmp(name, "split_bf16")) p->opt_split = value;
    else if (!strcmp(name, "packed_acts")) p->opt_packed_acts = value;
    else if (!strcmp(name, "range_guard")) p->opt_range_guard = value;
    else if (!strcmp(name, "fuse_front")) p->opt_fuse_front = value;
    else if (!strcmp(name, "table_batch")) p->opt_table_batch = value < 0 ? 0 : value;
    else if (!strcmp(name, "fuse_pairs")) p->opt_fuse_pairs = value;
    else if (!strcmp(name, "profile_tag_ops")) p->opt_tag_ops = value;
    else return fail(PF_EINVAL, "pf_hardnet_plan_set_option: unknown option '%s'", name);
    return PF_OK;
}

namespace {

struct Dims {
    int h = 0, w = 0;
};

// ops i (P) and i + 1 (C) form a pair conv_pair.hip can run as one launch: P = 3x3 conv of ONE range S, C = 3x3 conv whose first range
// is exactly P's output and whose second range is exactly S (hardnet.py:177-194: the links of an even layer start with the odd layer
// in front of it and contain that layer's input)
bool is_conv_pair(const pf_plan *p, size_t i) {
    if (i + 1 >= p->ops.size()) return false;
    const BlobOp &P = p->ops[i], &C = p->ops[i + 1];
    if (P.kind != OP_CONV || C.kind != OP_CONV || P.k != 3 || C.k != 3 || P.stride != 1 || C.stride != 1) return false;
    if (P.n_src != 1 || C.n_src < 2 || (P.dst_choff & 1) || (C.dst_choff & 1)) return false;
    if (C.src[0].tensor != P.dst || C.src[0].choff != P.dst_choff || C.src[0].ch != P.cout) return false;
    if (C.src[1].tensor != P.src[0].tensor || C.src[1].choff != P.src[0].choff || C.src[1].ch != P.src[0].ch) return false;
    for (uint32_t j = 0; j < C.n_src; ++j)
        if (C.src[j].choff & 1) return false;
    return conv_pair_supports((int)C.cout, (int)P.cout);
}

// Cityscapes id -> trainId (public label table; ids outside 0..33 -> 0, like the zeros_like init of
// export_cityscapes_segmentation_results.py:34-38)
void fill_lut(uint8_t *lut) {
    memset(lut, 0, 256);
    for (int i = 0; i < 34; ++i) lut[i] = 255;
    const int ids[19] = {7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33};
    for (int t = 0; t < 19; ++t) lut[ids[t]] = (uint8_t)t;
}

// spatial size of every tensor for an H x W network input
int propagate_dims(const pf_plan *p, int H, int W, std::vector<Dims> &d) {
    d.assign(p->tensors.size(), Dims());
    if (p->ops.empty()) return fail(PF_EBLOB, "empty op table");
    d[p->ops[0].src[0].tensor] = {H, W};
    for (const BlobOp &o : p->ops) {
        const Dims in = d[o.src[0].tensor];
        if (in.h <= 0 || in.w <= 0) return fail(PF_EBLOB, "op reads a tensor that was never produced");
        Dims out = in;
        switch (o.kind) {
            case OP_STEM:
            case OP_CONV: {
                const int pad = o.k / 2;
                out.h = (in.h + 2 * pad - (int)o.k) / (int)o.stride + 1;
                out.w = (in.w + 2 * pad - (int)o.k) / (int)o.stride + 1;
                break;
            }
            case OP_POOL: out = {in.h / 2, in.w / 2}; break;
            case OP_UPSAMPLE: out = d[o.src[1].tensor]; break;
            case OP_HEAD: break;
            default: return fail(PF_EBLOB, "unknown op kind %u", o.kind);
        }
        if (out.h <= 0 || out.w <= 0) return fail(PF_EINVAL, "input %dx%d is too small for this network", H, W);
        if (d[o.dst].h && (d[o.dst].h != out.h || d[o.dst].w != out.w) && o.kind != OP_HEAD)
            return fail(PF_EBLOB, "tensor %u written with two different sizes", o.dst);
        if (o.kind != OP_HEAD) d[o.dst] = out;
    }
    return PF_OK;
}

// workspace: PF_WS_STATUS_BYTES of status words - word 0 = PF_STATUS_* bits of the last forward (written once, by the
// range_finalize launch that ends a forward), word 1 = the same bits ORed over every forward since the host cleared it
// (sticky), word 2 = the word the kernels of the running forward OR their flags into, words kSlot0 + i = max |v| that op i
// of the table reported so far (bit pattern; low side of the range guard, conv_mfma.h), words kSlot0 + kMaxSlots + i = the
// same for the last finished forward (pf_hardnet_range_maxima).  The finalizer leaves word 2 and the live slots cleared for
// the next forward: the host zeroes the block once (pf_hardnet_status_reset) and no forward starts with a memset.  Then
// every tensor except the network input in its own 256-B aligned region
constexpr size_t kStatusBytes = PF_WS_STATUS_BYTES;
constexpr int kStickyWord = PF_WS_STICKY_OFFSET / 4, kLiveWord = 2, kSlot0 = 16, kMaxSlots = ((int)(kStatusBytes / 4) - kSlot0) / 2;
int layout(const pf_plan *p, int B, const std::vector<Dims> &d, std::vector<size_t> &off, size_t &total) {
    off.assign(p->tensors.size(), (size_t)-1);
    size_t cur = kStatusBytes;
    const uint32_t input = p->ops[0].src[0].tensor;
    for (size_t t = 0; t < p->tensors.size(); ++t) {
        if (t == input || d[t].h == 0) continue;
        off[t] = cur;
        // channels padded to whole groups of 4: the same region holds the tensor as fp32 NCHW or in the S4 layout
        cur += align_up((size_t)B * ((p->tensors[t].channels + 3) / 4 * 4) * d[t].h * d[t].w * sizeof(float), 256);
    }
    total = cur;
    return PF_OK;
}

int run_net(const pf_plan *p, const StemArgs *stem, const float *dense_x, int B, int H, int W, int out_h, int out_w,
            void *out_seg, int out_seg_is_i64, float *out_logits, float *out_orig, void *ws, size_t ws_bytes,
            hipStream_t s) {
    std::vector<Dims> d;
    int rc = propagate_dims(p, H, W, d);
    if (rc) return rc;
    std::vector<size_t> off;
    size_t need = 0;
    layout(p, B, d, off, need);
    if (ws_bytes < need) return fail(PF_EWORKSPACE, "workspace %zu B < required %zu B", ws_bytes, need);
    const uint32_t input = p->ops[0].src[0].tensor;
    auto tptr = [&](uint32_t t) -> float * {
        return t == input ? const_cast<float *>(dense_x) : reinterpret_cast<float *>((char *)ws + off[t]);
    };

    const bool tag_ops = p->opt_tag_ops != 0 || g_opt_tag_ops != 0;   // option "profile_tag_ops" (per plan, or process-wide for tools that switch it on late): per-op labels in pf_profile_* records
    const bool fuse = p->opt_fuse_pool != 0;      // option "fuse_pool"
    // range guard of the two-term operand split (conv_mfma.h): producers of tensors a split kernel may read raise
    // PF_STATUS_RANGE in the live status word and report their max |v| to their op's slot; the forward ends with range_finalize
    // (PF_STATUS_RANGE_LOW, published status word, sticky word, everything live cleared for the next forward).  fp32-only plans
    // clamp nothing and check nothing
    unsigned *const st_words = reinterpret_cast<unsigned *>(ws);
    unsigned *const status_all = (p->opt_split && p->opt_range_guard) ? st_words + kLiveWord : nullptr;
    // the guard of a launch: only if what it stores can become an operand of a split kernel (the logits cannot)
    auto status_of = [&](uint32_t dst_t) -> unsigned * { return p->feeds_conv[dst_t] ? status_all : nullptr; };
    auto slot_of = [&](size_t op_i, uint32_t dst_t) -> unsigned * {
        return (status_all && p->feeds_conv[dst_t]) ? st_words + kSlot0 + op_i : nullptr;
    };

    // ---- tensor formats.  The op loop below runs twice: a dry pass records every launch (which tensors it reads and
    // writes, whether its kernel can read / write the S4 layout of conv_s4.hip), the formats are then decided - a tensor is
    // S4 iff every launch that reads it reads all its inputs as S4 and every launch that writes it can write S4 - and
    // the second pass enqueues.
    struct ConvRec {
        uint32_t src_t[kConvMaxSrc], dst_t;
        int sb, se;            // ranges it accumulates
        bool can_read, can_write;   // can_write: whatever it reads; can_write_s4: only when it reads S4 itself
        bool can_write_s4;
        int dst_limit;         // S4 stores: first buffer channel this launch must NOT write (conv_mfma.h)
        int nt, wide;          // conv_s4 shape if it reads S4
    };
    const size_t nT = p->tensors.size();
    std::vector<ConvRec> recs;
    std::vector<uint8_t> fmt(nT, 0), cand(nT, 1);
    std::vector<std::vector<uint8_t>> written(nT);
    for (size_t t = 0; t < nT; ++t) written[t].assign(p->tensors[t].channels + 8, 0);
    const bool s4_allowed = p->opt_packed_acts && p->opt_split && (g_conv_force.kind == 0 || g_conv_force.kind == 5);
    bool dry = true;
    size_t rec_i = 0;
    struct ConvMeta {
        uint32_t src_t[kConvMaxSrc], dst_t;
    };

    auto fill_conv_args = [&](const BlobOp &o, size_t i, const Dims &in, const Dims &out, ConvArgs &a, ConvMeta &mt) {
        memset(&a, 0, sizeof(a));
        a.n_src = (int)o.n_src;
        int c0 = 0;
        for (int j = 0; j < a.n_src; ++j) {
            a.src[j] = tptr(o.src[j].tensor);
            mt.src_t[j] = o.src[j].tensor;
            a.src_ctotal[j] = (int)p->tensors[o.src[j].tensor].channels;
            a.src_choff[j] = (int)o.src[j].choff;
            a.src_cstart[j] = c0;
            c0 += (int)o.src[j].ch;
        }
        for (int j = a.n_src; j <= kConvMaxSrc; ++j) a.src_cstart[j] = c0;
        a.bias = p->dev_weights + p->conv[i].bias_off;
        a.dst = tptr(o.dst);
        mt.dst_t = o.dst;
        a.dst_ctotal = (int)p->tensors[o.dst].channels;
        a.dst_choff = (int)o.dst_choff;
        a.Cin = (int)o.cin; a.Cout = (int)o.cout;
        a.Hin = in.h; a.Win = in.w; a.Hout = out.h; a.Wout = out.w;
        a.relu = (int)o.relu;
        a.zero_page = p->dev_weights;   // first 64 floats of the weight arena are zeros
        a.ntiles = ((int)o.cout + 15) / 16;
        a.src_begin = 0;
        a.src_end = a.n_src;
        a.acc_scale = 1.0f;
        a.status = status_of(o.dst);
        a.range_slot = slot_of(i, o.dst);
        static const bool probe_on = ab_env("PF_PROBE") != nullptr;
        a.probe = probe_on ? probe_buffer() : nullptr;
    };
    // need: bit 1 = even tile rows (pooling epilogue), bit 2 = fused stage (not available on the generic path)
    auto launch_conv_op = [&](const BlobOp &o, size_t i, ConvArgs &a, const ConvMeta &mt, int need) -> int {
        const bool generic = (a.Win & 3) != 0;
        ConvChoice ch{0, 0, 0, 0};
        if (!generic) {
            ch = choose_conv((int)o.k, (int)o.stride, a.Cin, a.Cout, a.Hout, a.Wout, p->opt_table_batch > 0 ? p->opt_table_batch : B, need,
                             p->opt_use_tuned);
            if (g_conv_force.kind == 2 && o.stride == 1) {
                ch = g_conv_force;
                if ((need & 2) && ch.p0 == 1) ch.p0 = 2;
            }
            if (g_conv_force.kind == 1) ch = g_conv_force;
            if (g_conv_force.kind == 4 && p->conv[i].has_split && (!need || o.k == 1)) ch = g_conv_force;
            if (ch.kind == 4 && (!p->conv[i].has_split || (need && o.k != 1) || !p->opt_split)) ch = ConvChoice{1, 0, 0, 0};
            if (ch.kind == 3) ch = ConvChoice{1, 0, 0, 0};   // (kind 3 was conv_valu, removed in round 3: selected by no table row)
            // pf_debug_force_conv(5, ..): launches that cannot read S4 (fp32 sources) still have to be able to WRITE it
            if (g_conv_force.kind == 5) ch = ConvChoice{1, 0, 0, 0};
        }
        if (dry) {
            ConvRec r;
            memset(&r, 0, sizeof(r));
            memcpy(r.src_t, mt.src_t, sizeof(r.src_t));
            r.dst_t = mt.dst_t;
            r.sb = a.src_begin;
            r.se = a.src_end;
            // reads S4: the layers the table gives to the split kernels (same tile parameters), big 1x1 convs, or all
            // eligible convs under pf_debug_force_conv(5, nt, wide)
            const bool forced = g_conv_force.kind == 5;
            const long px = (long)(p->opt_table_batch > 0 ? p->opt_table_batch : B) * a.Hout * a.Wout;   // (the pinned batch decides, like every other choice)
            bool want = ch.kind == 4 || (o.k == 1 && px >= 32768 && ch.kind == 1) || forced;
            r.nt = forced ? g_conv_force.p0 : (o.k == 1 ? 4 : (ch.kind == 4 ? ch.p0 : 2));
            r.wide = forced ? g_conv_force.p1 : (ch.kind == 4 ? ch.p1 : 0);
            ConvChoice s4c;   // measured conv_s4 row of this layer (conv_select.cpp): decides, and names the shape
            if (!forced && !generic && p->opt_use_tuned &&
                choose_s4((int)o.k, a.Cin, a.Cout, a.Hout, a.Wout, p->opt_table_batch > 0 ? p->opt_table_batch : B, &s4c)) {
                want = s4c.kind == 5;
                if (want) { r.nt = s4c.p0; r.wide = s4c.p1; }
            }
            bool res_fits = true;
            if (a.res) {
                const int nt1 = r.nt < 1 ? 1 : (r.nt > a.ntiles ? a.ntiles : r.nt);
                res_fits = (size_t)nt1 * 16 * res_chan_stride(res_extent(8, a.res_sh), res_extent(32, a.res_sw)) * sizeof(float) <= 60 * 1024;
            }
            r.can_read = s4_allowed && !generic && o.stride == 1 && o.kind == OP_CONV && p->conv[i].has_s4 && (need == 0 || o.k == 1) && want &&
                         res_fits && ((a.src_begin == 0 && a.src_end == a.n_src) || p->conv[i].s4_pad);
            r.can_write_s4 = s4_allowed && !generic && (a.dst_choff & 1) == 0;
            r.can_write = r.can_write_s4 && (ch.kind == 1 || ch.kind == 4);
            // channel bookkeeping of the destination: zero-fill the tail of the last group unless its owner wrote it already
            std::vector<uint8_t> &wr = written[mt.dst_t];
            const int lo = a.dst_choff, hi = a.dst_choff + a.Cout;
            if ((lo & 3) == 2 && !wr[lo - 2]) cand[mt.dst_t] = 0;   // would expose an unwritten half group to readers of this range
            r.dst_limit = ((hi & 3) != 0 && !wr[hi]) ? (hi + 3) / 4 * 4 : hi;
            for (int c = lo; c < hi; ++c) wr[c] = 1;
            recs.push_back(r);
            return PF_OK;
        }
        const ConvRec &rec = recs[rec_i++];
        a.dst_fmt = fmt[rec.dst_t];
        a.dst_c4 = (a.dst_ctotal + 3) / 4;
        a.dst_limit = rec.dst_limit;
        bool read_s4 = rec.can_read;
        for (int j = rec.sb; j < rec.se; ++j) read_s4 = read_s4 && fmt[rec.src_t[j]];
        if (read_s4) {
            const int per = o.k == 3 ? 2 : 8;
            int e = 0;
            for (int j = 0; j < a.n_src; ++j) {
                const int ch0 = a.src_choff[j], chn = a.src_cstart[j + 1] - a.src_cstart[j];
                a.src_c4[j] = (a.src_ctotal[j] + 3) / 4;
                a.src_g0[j] = ch0 / 4;
                a.src_gn[j] = (ch0 + chn + 3) / 4 - ch0 / 4;
                a.src_ent0[j] = e;
                e += p->conv[i].s4_pad ? (a.src_gn[j] + per - 1) / per * per : a.src_gn[j];
            }
            for (int j = a.n_src; j <= kConvMaxSrc; ++j) a.src_ent0[j] = e;
            a.src_fmt = 1;
            a.acc_scale = p->conv[i].split_acc_scale;
            a.wpk = p->dev_weights + p->conv[i].s4_off;
            a.nchunks = p->conv[i].s4_rounds;
            a.chunk_begin = a.src_ent0[a.src_begin] / per;
            a.chunk_end = (a.src_ent0[a.src_end] + per - 1) / per;
            return launch_conv_s4(a, (int)o.k, rec.nt, rec.wide, B, s);
        }
        if (generic) {
            if (need) return fail(PF_EUNSUPPORTED, "fused conv on a width that is not a multiple of 4");
            a.wpk = p->dev_weights + p->conv[i].wpk_off;
            a.nchunks = p->conv[i].tiling.nchunks;
            return launch_conv(a, p->conv[i].tiling, B, s);
        }
        int rc = PF_EUNSUPPORTED;
        auto set_chunks = [&](int kc) {
            a.src_chunk0[0] = 0;
            for (int j = 0; j < kConvMaxSrc; ++j)
                a.src_chunk0[j + 1] = a.src_chunk0[j] + (j < a.n_src ? ((int)o.src[j].ch + kc - 1) / kc : 0);
            a.chunk_begin = a.src_chunk0[a.src_begin];
            a.chunk_end = a.src_chunk0[a.src_end];
        };
        if (ch.kind == 4) {
            a.wpk = p->dev_weights + p->conv[i].split_off;
            a.acc_scale = p->conv[i].split_acc_scale;
            a.nchunks = p->conv[i].split_chunks;
            set_chunks(o.k == 1 ? 32 : 8);
            rc = o.k == 1 ? launch_conv_split1(a, ch.p0, B, s) : launch_conv_split(a, ch.p0, ch.p1, B, s);
            if (rc != PF_EUNSUPPORTED) return rc;
            ch = ConvChoice{1, 0, 0, 0};
        }
        if (ch.kind == 2) {
            a.wpk = p->dev_weights + p->conv[i].wave_off;
            a.nchunks = p->conv[i].wave_chunks;
            set_chunks(wave_kc((int)o.k));
            rc = launch_conv_wave(a, (int)o.k, ch.p0, ch.p1 < a.ntiles ? ch.p1 : a.ntiles, ch.p2, B, s);
            if (rc == PF_EUNSUPPORTED) ch = ConvChoice{1, 0, 0, 0};   // shape not built: conv_dma with its cost model
        }
        if (ch.kind != 2) {
            a.wpk = p->dev_weights + p->conv[i].tiled_off;
            a.nchunks = p->conv[i].tiled_chunks;
            set_chunks(dma_kc((int)o.k, (int)o.stride));
            // the trailing rem_count output channels of a big image go to the vector ALU instead of an MFMA tile
            // (conv_dma WM=4 shapes only: forced or cost-model-chosen WM is checked inside, which falls back)
            a.rem = 0;
            if (p->opt_valu_rem && p->conv[i].rem_off && !need && !a.dst_fmt && a.src_begin == 0 && a.src_end == a.n_src &&
                (ch.p0 == 0 || ch.p0 == 4)) {
                a.rem = p->conv[i].rem_count;
                a.wrem = p->dev_weights + p->conv[i].rem_off;
                a.ntiles = ((int)o.cout - a.rem) / 16;
                rc = launch_conv_dma(a, (int)o.k, (int)o.stride, B, s, 4, ch.p0 == 4 && ch.p1 > 0 ? ch.p1 : 0, p->opt_table_batch);
                if (rc == PF_EUNSUPPORTED) {
                    a.rem = 0;
                    a.ntiles = ((int)o.cout + 15) / 16;
                }
            }
            if (a.rem == 0) rc = launch_conv_dma(a, (int)o.k, (int)o.stride, B, s, ch.p0, ch.p1, p->opt_table_batch);
        }
        return rc;
    };
    // pf_set_option("fuse_upsample", 0/1); at B=4: transUp.3 + conv1x1_up.3 144 us fused vs 233 us as two passes
    const bool fuse_up = p->opt_fuse_upsample != 0;   // option "fuse_upsample"
    auto can_commute_upsample = [&](size_t i, const Dims &in, const Dims &out) -> bool {
        if (!fuse || !fuse_up || i + 1 >= p->ops.size()) return false;
        const BlobOp &o = p->ops[i], &n = p->ops[i + 1];
        return n.kind == OP_CONV && n.k == 1 && n.stride == 1 && n.n_src == 2 && n.src[0].tensor == o.dst &&
               n.src[0].choff == 0 && n.src[0].ch == p->tensors[o.dst].channels && p->readers[o.dst] == 1 &&
               (in.w & 3) == 0 && (out.w & 3) == 0 && n.cout <= p->tensors[o.dst].channels &&
               2 * in.h <= out.h + 1 && 2 * in.w <= out.w + 1;   // >= ~2x upsampling: the residual window of a tile stays small
    };

    static const bool sync_ops = ab_env("PF_SYNC_OPS") != nullptr;   // debugging: localise a faulting launch
    cand[input] = 0;
    for (int pass = 0; pass < 2; ++pass) {
    dry = pass == 0;
    for (size_t i = 0; i < p->ops.size(); ++i) {
        if (sync_ops && !dry) {
            const hipError_t e = hipStreamSynchronize(s);
            fprintf(stderr, "[pf] before op %zu (%s): %s\n", i, p->tensors[p->ops[i].dst].name, hipGetErrorString(e));
        }
        const BlobOp &o = p->ops[i];
        const Dims in = d[o.src[0].tensor];
        const Dims out = o.kind == OP_HEAD ? in : d[o.dst];
        if (tag_ops && prof_enabled() && !dry) {
            char tag[96];
            snprintf(tag, sizeof(tag), "%02zu %s %u->%u %dx%d", i, p->tensors[o.dst].name, o.cin, o.cout, out.h, out.w);
            prof_set_tag(tag);
        }
        // ops executed by kernels that only know fp32 NCHW pin their tensors to it
        auto pin_fp32 = [&](const BlobOp &op) {
            for (uint32_t j = 0; j < op.n_src; ++j) cand[op.src[j].tensor] = 0;
            cand[op.dst] = 0;
        };
        // stem -> [3x3 s1 16 -> 24] -> [3x3 s2 24 -> <= 32] with single readers: the two convs run as conv_front.hip on the
        // packed-pair stem output; the tensor between them is never stored
        bool front = false;
        if (o.kind == OP_STEM && stem && s4_allowed && p->opt_fuse_front && i + 2 < p->ops.size()) {
            const BlobOp &n1 = p->ops[i + 1], &n2 = p->ops[i + 2];
            StemArgs probe = *stem;
            probe.wdep = p->dev_weights + p->conv[i].dep_off;
            probe.woh = p->conv[i].has_oh ? p->dev_weights + p->conv[i].oh_off : nullptr;
            probe.Hout = out.h; probe.Wout = out.w;
            front = n1.kind == OP_CONV && n1.k == 3 && n1.stride == 1 && n1.n_src == 1 && n1.src[0].tensor == o.dst && n1.src[0].choff == 0 &&
                    n1.src[0].ch == p->tensors[o.dst].channels && n1.dst_choff == 0 && n1.cout == p->tensors[n1.dst].channels &&
                    p->readers[o.dst] == 1 && p->readers[n1.dst] == 1 && n1.relu && p->conv[i + 1].has_s4 && p->conv[i + 1].s4_rounds == 2 &&
                    n2.kind == OP_CONV && n2.k == 3 && n2.stride == 2 && n2.n_src == 1 && n2.src[0].tensor == n1.dst && n2.src[0].choff == 0 &&
                    n2.src[0].ch == n1.cout && p->conv[i + 2].has_front && (n2.dst_choff & 3) == 0 &&
                    conv_front_supports((int)o.cout, (int)n1.cout, (int)n2.cout, out.h, out.w, (int)p->tensors[n2.dst].channels) && stem_writes_s4(probe) && g_conv_force.kind == 0;
        }
        if (front) {
            const BlobOp &n1 = p->ops[i + 1], &n2 = p->ops[i + 2];
            const Dims o2 = d[n2.dst];
            if (dry) {
                cand[o.dst] = 0; cand[n1.dst] = 0;   // (not subject to the format decision: the stem output is packed, the middle tensor never exists)
                ConvRec r;
                memset(&r, 0, sizeof(r));
                r.dst_t = n2.dst;
                r.sb = r.se = 0;
                r.can_read = false; r.can_write = true; r.can_write_s4 = true;
                std::vector<uint8_t> &wr = written[n2.dst];
                const int lo = (int)n2.dst_choff, hi = lo + (int)n2.cout;
                if ((lo & 3) == 2 && !wr[lo - 2]) cand[n2.dst] = 0;
                r.dst_limit = ((hi & 3) != 0 && !wr[hi]) ? (hi + 3) / 4 * 4 : hi;
                for (int c = lo; c < hi; ++c) wr[c] = 1;
                recs.push_back(r);
                i += 2;
                continue;
            }
            const ConvRec &rec = recs[rec_i++];
            StemArgs a = *stem;
            a.w = p->dev_weights + p->conv[i].raw_off;
            a.wdep = p->dev_weights + p->conv[i].dep_off;
            a.woh = p->conv[i].has_oh ? p->dev_weights + p->conv[i].oh_off : nullptr;
            a.bias = p->dev_weights + p->conv[i].bias_off;
            a.status = status_of(o.dst);
            a.range_slot = slot_of(i, o.dst);
            a.lut = p->dev_lut;
            a.dst = tptr(o.dst);
            a.dst_fmt = 1;
            a.Hout = out.h; a.Wout = out.w;
            if ((rc = launch_stem(a, s))) return rc;
            FrontArgs f;
            memset(&f, 0, sizeof(f));
            f.x = tptr(o.dst);
            f.w1 = p->dev_weights + p->conv[i + 1].s4_off;
            f.w2 = p->dev_weights + p->conv[i + 2].front_off;
            f.bias1 = p->dev_weights + p->conv[i + 1].bias_off;
            f.bias2 = p->dev_weights + p->conv[i + 2].bias_off;
            f.scale1 = p->conv[i + 1].split_acc_scale;
            f.scale2 = p->conv[i + 2].split_acc_scale;
            f.dst = tptr(n2.dst);
            f.dst_fmt = fmt[n2.dst];
            f.dst_ctotal = (int)p->tensors[n2.dst].channels;
            f.dst_c4 = (f.dst_ctotal + 3) / 4;
            f.dst_choff = (int)n2.dst_choff;
            f.dst_limit = rec.dst_limit;
            f.H1 = out.h; f.W1 = out.w; f.H2 = o2.h; f.W2 = o2.w;
            f.C1 = (int)n1.cout; f.C2 = (int)n2.cout; f.relu1 = (int)n1.relu; f.relu2 = (int)n2.relu;
            f.status = status_all;
            f.range_slot_mid = slot_of(i + 1, n1.dst);
            f.range_slot = slot_of(i + 2, n2.dst);
            static const bool front_probe = ab_env("PF_PROBE") != nullptr;
            f.probe = front_probe ? probe_buffer() : nullptr;
            if (tag_ops && prof_enabled()) {
                char tag[96];
                snprintf(tag, sizeof(tag), "%02zu+%02zu %s+%s %u->%u->%u %dx%d", i + 1, i + 2, p->tensors[n1.dst].name, p->tensors[n2.dst].name, o.cout,
                         n1.cout, n2.cout, o2.h, o2.w);
                prof_set_tag(tag);
            }
            if ((rc = launch_conv_front(f, B, s))) return rc;
            fmt[o.dst] = 1;            // pf_hardnet_tensor_read unpacks the stem output
            p->last_fmt = fmt;
            p->last_fmt[n1.dst] = 0xFF;   // ... and refuses the tensor between the two convs: it is never stored
            i += 2;
            continue;
        }
        // ---- an odd HarDBlock layer inside its consumer (conv_pair.hip): decided in the REAL pass only - the dry pass recorded the two
        //      launches as they would run alone, so the formats are what the unfused plan has, and any pair whose tensors did not all
        //      end up as packed pairs simply runs as two launches
        if (!dry && p->opt_fuse_pairs && is_conv_pair(p, i) && p->conv[i + 1].has_pair && (in.w & 3) == 0 && (g_conv_force.kind == 0 || g_conv_force.kind == 5) &&
            rec_i + 1 < recs.size()) {
            const BlobOp &C = p->ops[i + 1];
            const ConvRec &rp = recs[rec_i], &rc_ = recs[rec_i + 1];
            bool ok = rp.can_read && rc_.can_read && fmt[o.dst] && fmt[C.dst] && fmt[o.src[0].tensor];
            for (uint32_t j = 1; j < C.n_src; ++j) ok = ok && fmt[C.src[j].tensor];
            const int Bt = p->opt_table_batch > 0 ? p->opt_table_batch : B;
            ok = ok && pair_wanted((int)o.cin, (int)o.cout, (int)C.cin, (int)C.cout, out.h, out.w, Bt, p->opt_fuse_pairs);
            if (ok) {
                PairArgs pa;
                memset(&pa, 0, sizeof(pa));
                ConvMeta mt;
                fill_conv_args(C, i + 1, in, out, pa.c, mt);
                ConvArgs &a = pa.c;
                // K order [S, others.., P]: source j of the launch = source (j + 1) % n of the op
                const int n = (int)C.n_src;
                int e = 0;
                for (int j = 0; j < n; ++j) {
                    const BlobSrc &sj = C.src[(j + 1) % n];
                    const bool isP = j == n - 1;
                    a.src[j] = tptr(sj.tensor);
                    a.src_ctotal[j] = (int)p->tensors[sj.tensor].channels;
                    a.src_choff[j] = (int)sj.choff;
                    a.src_c4[j] = (a.src_ctotal[j] + 3) / 4;
                    a.src_g0[j] = isP ? 0 : (int)sj.choff / 4;
                    a.src_gn[j] = isP ? ((int)sj.ch + 3) / 4 : ((int)sj.choff + (int)sj.ch + 3) / 4 - (int)sj.choff / 4;
                    a.src_ent0[j] = e;
                    e += (a.src_gn[j] + 1) / 2 * 2;
                }
                for (int j = n; j <= kConvMaxSrc; ++j) a.src_ent0[j] = e;
                a.src_fmt = 1;
                a.dst_fmt = 1;
                a.dst_c4 = (a.dst_ctotal + 3) / 4;
                a.dst_limit = rc_.dst_limit;
                a.acc_scale = p->conv[i + 1].split_acc_scale;
                a.wpk = p->dev_weights + p->conv[i + 1].pair_c_off;
                a.nchunks = p->conv[i + 1].pair_rounds;
                pa.p_wpk = p->dev_weights + p->conv[i + 1].pair_two_off;
                pa.p_w9 = p->dev_weights + p->conv[i + 1].pair_nine_off;
                pa.p_bias = p->dev_weights + p->conv[i].bias_off;
                pa.p_acc_scale = p->conv[i].split_acc_scale;
                pa.p_dst = tptr(o.dst);
                pa.p_dst_c4 = ((int)p->tensors[o.dst].channels + 3) / 4;
                pa.p_dst_choff = (int)o.dst_choff;
                pa.p_dst_limit = rp.dst_limit;
                pa.p_cout = (int)o.cout;
                pa.p_cin = (int)o.cin;
                pa.p_ntiles = ((int)o.cout + 15) / 16;
                pa.p_relu = (int)o.relu;
                pa.p_range_slot = slot_of(i, o.dst);
                pa.rounds_s = a.src_ent0[1] / 2;
                pa.round_d = a.src_ent0[n - 1] / 2;
                pa.merged = p->conv[i + 1].pair_merged && p->opt_fuse_pairs != 3;
                if (tag_ops && prof_enabled()) {
                    char tag[96];
                    snprintf(tag, sizeof(tag), "%02zu+%02zu %s+%s %u->%u->%u %dx%d", i, i + 1, p->tensors[o.dst].name, p->tensors[C.dst].name, o.cin,
                             o.cout, C.cout, out.h, out.w);
                    prof_set_tag(tag);
                }
                if ((rc = launch_conv_pair(pa, B, s))) return rc;
                rec_i += 2;
                ++i;
                continue;
            }
        }
        if (o.kind == OP_STEM && stem) {
            if (dry) { pin_fp32(o); continue; }
            StemArgs a = *stem;
            a.w = p->dev_weights + p->conv[i].raw_off;
            a.wdep = p->dev_weights + p->conv[i].dep_off;
            a.woh = p->conv[i].has_oh ? p->dev_weights + p->conv[i].oh_off : nullptr;
            static const bool stem_probe = ab_env("PF_PROBE") != nullptr;     // read once, not per forward
            static const int stem_plane_pad = ab_env("PF_DBG_PLANE_PAD") ? atoi(ab_env("PF_DBG_PLANE_PAD")) : 0;
            a.probe = stem_probe ? probe_buffer() : nullptr;
            a.dbg_plane_pad = stem_plane_pad;
            a.bias = p->dev_weights + p->conv[i].bias_off;
            a.status = status_of(o.dst);
            a.range_slot = slot_of(i, o.dst);
            a.lut = p->dev_lut;
            a.dst = tptr(o.dst);
            a.Hout = out.h;
            a.Wout = out.w;
            if (o.cout != 16 || o.k != 3 || o.stride != 2 || o.dst_choff != 0 ||
                p->tensors[o.dst].channels != 16 || (uint32_t)(a.T * (a.n_cls + 1)) != o.cin)
                return fail(PF_EUNSUPPORTED, "fused stem expects a 3x3/s2 conv %d->16, got %u->%u k%u s%u",
                            a.T * (a.n_cls + 1), o.cin, o.cout, o.k, o.stride);
            if ((rc = launch_stem(a, s))) return rc;
        } else if (o.kind == OP_STEM || o.kind == OP_CONV) {
            if (o.src[0].tensor == input && !dense_x)
                return fail(PF_EINVAL, "network input is consumed by a generic conv: use pf_hardnet_forward_dense");
            // a caller-provided dense input has no producer kernel that could have checked its range
            // (its maximum goes to the last slot of the block: the input has no op of its own)
            if (o.src[0].tensor == input && !dry && status_all &&
                (rc = launch_range_check(dense_x, (size_t)B * p->tensors[input].channels * in.h * in.w, status_all, st_words + kSlot0 + kMaxSlots - 1, s)))
                return rc;
            // conv + AvgPool2d(2,2): pool in the conv epilogue, the full-resolution tensor is never written
            const BlobOp *pool = nullptr;
            if (fuse && i + 1 < p->ops.size() && p->ops[i + 1].kind == OP_POOL && o.stride == 1 && o.k == 1 && (in.w & 3) == 0 &&
                p->ops[i + 1].src[0].tensor == o.dst && o.dst_choff == 0 && o.cout == p->tensors[o.dst].channels &&
                p->readers[o.dst] == 1 && out.h >= 2 && out.w >= 2)
                pool = &p->ops[i + 1];
            ConvArgs a;
            ConvMeta mt;
            fill_conv_args(o, i, in, out, a, mt);
            if (pool) {
                a.pool = 1;
                a.dst = tptr(pool->dst);
                mt.dst_t = pool->dst;
                a.dst_ctotal = (int)p->tensors[pool->dst].channels;
                a.status = status_of(pool->dst);
                a.range_slot = slot_of(i, pool->dst);
            }
            if ((rc = launch_conv_op(o, i, a, mt, pool ? 2 : 0))) return rc;
            if (pool) ++i;   // the pool op is done
        } else if (o.kind == OP_POOL) {
            if (dry) { pin_fp32(o); continue; }
            if ((rc = launch_avgpool2(tptr(o.src[0].tensor), tptr(o.dst), B * (int)o.cin, in.h, in.w, status_of(o.dst), slot_of(i, o.dst), s))) return rc;
        } else if (o.kind == OP_UPSAMPLE && can_commute_upsample(i, in, out)) {
            // TransitionUp + 1x1 conv over cat([up(x), skip])  ==  W_skip*skip + up(W_x*x)   (conv_epilogue.h)
            const BlobOp &n = p->ops[i + 1];
            const Dims hi = out;   // = size of the skip tensor
            ConvArgs lo;
            ConvMeta lo_mt;
            fill_conv_args(n, i + 1, in, in, lo, lo_mt);
            lo.src[0] = tptr(o.src[0].tensor);                 // x at the low resolution
            lo_mt.src_t[0] = o.src[0].tensor;
            lo.src_ctotal[0] = (int)p->tensors[o.src[0].tensor].channels;
            lo.src_choff[0] = (int)o.src[0].choff;
            lo.dst = tptr(o.dst);                              // scratch: the slot of the (never built) upsampled tensor
            lo_mt.dst_t = o.dst;
            if (dry) cand[o.dst] = 0;                          // sampled as an fp32 residual by the other half
            lo.dst_ctotal = (int)n.cout;
            lo.dst_choff = 0;
            lo.relu = 0;
            lo.no_bias = 1;
            lo.status = nullptr;                               // an fp32 residual of the other half, never a split operand
            lo.range_slot = nullptr;
            lo.Cin = (int)n.src[0].ch;
            lo.src_begin = 0;
            lo.src_end = 1;
            auto tag_half = [&](const char *half, int cin, const Dims &dd) {
                if (!(tag_ops && prof_enabled()) || dry) return;
                char tag[96];
                snprintf(tag, sizeof(tag), "%02zu%c %s.%s %d->%u %dx%d", i + 1, half[0] == 'l' ? 'a' : 'b', p->tensors[n.dst].name,
                         half, cin, n.cout, dd.h, dd.w);
                prof_set_tag(tag);
            };
            tag_half("lo", lo.Cin, in);
            if ((rc = launch_conv_op(n, i + 1, lo, lo_mt, 4))) return rc;
            if (sync_ops && !dry) fprintf(stderr, "[pf]   low-resolution half: %s\n", hipGetErrorString(hipStreamSynchronize(s)));
            ConvArgs hi_a;
            ConvMeta hi_mt;
            fill_conv_args(n, i + 1, hi, hi, hi_a, hi_mt);
            hi_a.Cin = (int)n.src[1].ch;
            hi_a.src_begin = 1;
            hi_a.src_end = 2;
            tag_half("hi", hi_a.Cin, hi);
            hi_a.res = tptr(o.dst);
            hi_a.res_ctotal = (int)n.cout;
            hi_a.res_choff = 0;
            hi_a.Hres = in.h;
            hi_a.Wres = in.w;
            hi_a.res_sh = hi.h > 1 ? (float)(in.h - 1) / (float)(hi.h - 1) : 0.f;
            hi_a.res_sw = hi.w > 1 ? (float)(in.w - 1) / (float)(hi.w - 1) : 0.f;
            if ((rc = launch_conv_op(n, i + 1, hi_a, hi_mt, 4))) return rc;
            ++i;   // the 1x1 conv is done
        } else if (o.kind == OP_UPSAMPLE) {
            if (dry) { pin_fp32(o); continue; }
            if ((rc = launch_upsample(tptr(o.src[0].tensor), tptr(o.dst), B * (int)o.cin, in.h, in.w, out.h, out.w, status_of(o.dst), slot_of(i, o.dst), s)))
                return rc;
        } else if (o.kind == OP_HEAD) {
            if (dry) { pin_fp32(o); continue; }
            HeadArgs a;
            a.logits = tptr(o.src[0].tensor);
            a.out_seg = out_seg; a.out_logits = out_logits; a.out_is_i64 = out_seg_is_i64;
            a.B = B; a.C = (int)o.cin; a.Hin = in.h; a.Win = in.w; a.Hout = out_h; a.Wout = out_w;
            if (out_orig && (rc = launch_copy(out_orig, a.logits, (size_t)B * a.C * in.h * in.w * sizeof(float), s))) return rc;
            if ((rc = launch_head(a, s))) return rc;
        }
    }
    if (dry) {
        // fixpoint: a launch reads S4 only if ALL the ranges it accumulates are S4; a tensor stays S4 only while all its
        // readers do and all its writers can
        if (!s4_allowed) std::fill(cand.begin(), cand.end(), 0);
        for (bool changed = true; changed;) {
            changed = false;
            for (const ConvRec &r : recs) {
                bool all = r.can_read;
                for (int j = r.sb; j < r.se; ++j) all = all && cand[r.src_t[j]];
                if (!all)
                    for (int j = r.sb; j < r.se; ++j)
                        if (cand[r.src_t[j]]) { cand[r.src_t[j]] = 0; changed = true; }
                if (!(r.can_write || (all && r.can_write_s4)) && cand[r.dst_t]) { cand[r.dst_t] = 0; changed = true; }
            }
        }
        // a tensor nobody reads as S4 (network outputs tapped by the caller) stays fp32
        std::vector<uint8_t> read_s4(nT, 0);
        for (const ConvRec &r : recs)
            for (int j = r.sb; j < r.se; ++j) read_s4[r.src_t[j]] = 1;
        for (size_t t = 0; t < nT; ++t) fmt[t] = cand[t] && read_s4[t];
        p->last_fmt = fmt;
    }
    }
    if (tag_ops) prof_set_tag(nullptr);
    // the dense input's slot is the last word of the block: the finalizer scans all of it (unused slots stay 0)
    return launch_range_finalize(st_words, kLiveWord, kSlot0, kMaxSlots, kStickyWord, s);
}

}  // namespace

// Range normalisation (conv_mfma.h, "low side"): a static range propagation over the op table.  est[t][c] = expected magnitude
// (rms-like) of channel c of tensor t in the ORIGINAL units: 1 for dense inputs (sqrt(1/n_cls) for the one-hot channels of the
// fused stem), sqrt(sum_k |w_ok|^2 est_k^2 + b_o^2) behind a conv (uncorrelated-inputs model), / sqrt(2) behind a ReLU, copied
// through pool / upsample.  Channel c is then STORED multiplied by s = 2^round(log2(kRangeTarget / est)): the producer's
// weight row and bias are multiplied by s, every consumer's weight column divided by it - powers of two, so the network
// computes bit-identical fp32 products; only where the fp16 pair's subnormal floor (2^-25 absolute) and its ceiling (65504)
// fall relative to the data changes.  A checkpoint re-parameterised across a BatchNorm (gamma * alpha, next weights / alpha)
// gets s / alpha and stores the same values.  Channels of tensors no convolution reads (network outputs, the head's input)
// keep s = 1.  The guarantee itself is the run-time guard (PF_STATUS_RANGE / PF_STATUS_RANGE_LOW); this only decides how
// often it fires.
static void normalize_ranges(pf_plan *p, std::vector<float> &w) {
    const size_t nT = p->tensors.size();
    p->chan_scale.assign(nT, std::vector<float>());
    std::vector<std::vector<double>> est(nT);
    for (size_t t = 0; t < nT; ++t) {
        p->chan_scale[t].assign(p->tensors[t].channels, 1.0f);
        est[t].assign(p->tensors[t].channels, 1.0);
    }
    // feeds_conv: backwards over the table (ops are in topological order)
    p->feeds_conv.assign(nT, 0);
    for (size_t i = p->ops.size(); i-- > 0;) {
        const BlobOp &o = p->ops[i];
        if (o.kind == OP_STEM || o.kind == OP_CONV)
            for (uint32_t j = 0; j < o.n_src; ++j) p->feeds_conv[o.src[j].tensor] = 1;
        else if ((o.kind == OP_POOL || o.kind == OP_UPSAMPLE) && p->feeds_conv[o.dst])
            p->feeds_conv[o.src[0].tensor] = 1;
    }
    // pinned to s = 1: read by the head, by nobody (outputs tapped by the caller), or the network input
    std::vector<uint8_t> pinned(nT, 0), read(nT, 0);
    for (const BlobOp &o : p->ops) {
        for (uint32_t j = 0; j < (o.kind == OP_UPSAMPLE ? 1u : o.n_src); ++j) read[o.src[j].tensor] = 1;
        if (o.kind == OP_HEAD) pinned[o.src[0].tensor] = 1;
    }
    for (size_t t = 0; t < nT; ++t) pinned[t] = pinned[t] || !read[t] || !p->feeds_conv[t];
    if (p->ops.empty()) return;
    const uint32_t input = p->ops[0].src[0].tensor;
    pinned[input] = 1;
    if (p->ops[0].kind == OP_STEM && p->hdr.n_cls > 0) {   // fused stem: T * n_cls one-hot channels, then T depth channels
        const uint32_t C = p->tensors[input].channels, T = C / (p->hdr.n_cls + 1);
        if (T * (p->hdr.n_cls + 1) == C)
            for (uint32_t c = 0; c < T * p->hdr.n_cls; ++c) est[input][c] = std::sqrt(1.0 / p->hdr.n_cls);
    }
    // pool / upsample outputs inherit their source's scale: they are pinned iff ... their source is; a pinned destination of
    // such an op pins the source channel too (the scale must be the same on both sides), so walk backwards first
    for (size_t i = p->ops.size(); i-- > 0;) {
        const BlobOp &o = p->ops[i];
        if ((o.kind == OP_POOL || o.kind == OP_UPSAMPLE) && pinned[o.dst]) pinned[o.src[0].tensor] = 1;
    }
    for (const BlobOp &o : p->ops) {
        if (o.kind == OP_POOL || o.kind == OP_UPSAMPLE) {
            for (uint32_t c = 0; c < o.src[0].ch; ++c) {
                p->chan_scale[o.dst][o.dst_choff + c] = p->chan_scale[o.src[0].tensor][o.src[0].choff + c];
                est[o.dst][o.dst_choff + c] = est[o.src[0].tensor][o.src[0].choff + c];
            }
            continue;
        }
        if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
        const size_t kk = (size_t)o.k * o.k;
        // input channel k of the conv -> (estimate, stored scale)
        std::vector<double> e_in(o.cin);
        std::vector<float> s_in(o.cin);
        uint32_t k0 = 0;
        for (uint32_t j = 0; j < o.n_src; ++j)
            for (uint32_t c = 0; c < o.src[j].ch; ++c, ++k0) {
                e_in[k0] = est[o.src[j].tensor][o.src[j].choff + c];
                s_in[k0] = p->chan_scale[o.src[j].tensor][o.src[j].choff + c];
            }
        for (uint32_t co = 0; co < o.cout; ++co) {
            float *wr = w.data() + o.w_off + (size_t)co * o.cin * kk;
            float &b = w[o.b_off + co];
            double var = (double)b * b;
            for (uint32_t k = 0; k < o.cin; ++k) {
                double ss = 0;
                for (size_t q = 0; q < kk; ++q) ss += (double)wr[k * kk + q] * wr[k * kk + q];
                var += ss * e_in[k] * e_in[k];
            }
            double e = std::sqrt(var);
            if (o.relu) e *= 0.70710678118654752;
            float s_out = 1.0f;
            if (p->opt_normalize && !pinned[o.dst] && e > 0 && std::isfinite(e)) {
                int ex = (int)std::lround(std::log2((double)kRangeTarget / e));
                ex = ex < -60 ? -60 : (ex > 60 ? 60 : ex);
                s_out = std::ldexp(1.0f, ex);
            }
            for (uint32_t k = 0; k < o.cin; ++k) {
                const float f = s_out / s_in[k];   // a power of two
                if (f != 1.0f)
                    for (size_t q = 0; q < kk; ++q) wr[k * kk + q] *= f;
            }
            b *= s_out;
            p->chan_scale[o.dst][o.dst_choff + co] = s_out;
            est[o.dst][o.dst_choff + co] = e;
        }
    }
}

extern "C" int pf_hardnet_plan_create(const void *blob, size_t bytes, int in_ch, int n_cls, pf_plan **out) {
    if (!blob || !out) return fail(PF_EINVAL, "pf_hardnet_plan_create: null argument");
    if (bytes < sizeof(BlobHeader)) return fail(PF_EBLOB, "blob shorter than its header");
    BlobHeader h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, kBlobMagic, 8) != 0 || h.version != kBlobVersion)
        return fail(PF_EBLOB, "bad blob magic/version");
    if (h.total_bytes != bytes || h.tensor_off + (uint64_t)h.n_tensors * sizeof(BlobTensor) > bytes ||
        h.op_off + (uint64_t)h.n_ops * sizeof(BlobOp) > bytes || h.weights_off > bytes || (h.weights_off & 3))
        return fail(PF_EBLOB, "blob table offsets out of range (total %llu, got %zu)",
                    (unsigned long long)h.total_bytes, bytes);
    if ((int)h.in_ch != in_ch || (int)h.n_cls != n_cls)
        return fail(PF_EINVAL, "blob is for in_ch=%u n_cls=%u, caller asked for %d/%d", h.in_ch, h.n_cls, in_ch, n_cls);
    pf_plan *p = new pf_plan();
    p->opt_fuse_pool = g_opt_fuse_pool; p->opt_fuse_upsample = g_opt_fuse_upsample; p->opt_valu_rem = g_opt_valu_rem;
    p->opt_split = g_opt_split; p->opt_use_tuned = g_opt_use_tuned; p->opt_packed_acts = g_opt_packed_acts;
    p->opt_range_guard = g_opt_range_guard;
    p->opt_fuse_front = g_opt_fuse_front;
    p->opt_normalize = g_opt_normalize;
    p->opt_fuse_pairs = g_opt_fuse_pairs;
    p->opt_tag_ops = g_opt_tag_ops;
    p->hdr = h;
    p->tensors.resize(h.n_tensors);
    p->ops.resize(h.n_ops);
    memcpy(p->tensors.data(), (const char *)blob + h.tensor_off, h.n_tensors * sizeof(BlobTensor));
    memcpy(p->ops.data(), (const char *)blob + h.op_off, h.n_ops * sizeof(BlobOp));
    const size_t n_w = (bytes - h.weights_off) / sizeof(float);
    if ((int)p->ops.size() >= kMaxSlots - 1) {
        delete p;
        return fail(PF_EUNSUPPORTED, "op table of %zu ops: the status block has %d per-op words", p->ops.size(), kMaxSlots - 1);
    }

    // validate the op table
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const BlobOp &o = p->ops[i];
        bool ok = o.n_src >= 1 && o.n_src <= (uint32_t)kMaxSrc && o.dst < h.n_tensors;
        uint32_t cin = 0;
        for (uint32_t j = 0; ok && j < o.n_src; ++j) {
            ok = o.src[j].tensor < h.n_tensors &&
                 o.src[j].choff + o.src[j].ch <= p->tensors[o.src[j].tensor].channels;
            cin += o.src[j].ch;
        }
        if (ok && (o.kind == OP_STEM || o.kind == OP_CONV))
            ok = cin == o.cin && o.dst_choff + o.cout <= p->tensors[o.dst].channels && (o.k == 1 || o.k == 3) &&
                 (o.stride == 1 || o.stride == 2) && o.w_off + (uint64_t)o.cout * o.cin * o.k * o.k <= n_w &&
                 o.b_off + o.cout <= n_w;
        if (ok && (o.kind == OP_POOL || o.kind == OP_UPSAMPLE)) ok = o.src[0].ch <= p->tensors[o.dst].channels;
        if (!ok) {
            delete p;
            return fail(PF_EBLOB, "op %zu is inconsistent with the tensor table", i);
        }
    }
    // the folded weights, re-parameterised so that every stored channel has an expected magnitude of kRangeTarget
    std::vector<float> wnorm(reinterpret_cast<const float *>((const char *)blob + h.weights_off),
                             reinterpret_cast<const float *>((const char *)blob + h.weights_off) + n_w);
    normalize_ranges(p, wnorm);
    const float *wts = wnorm.data();

    // tile the weights on the host
    std::vector<float> host(64, 0.f);   // zero page
    p->conv.resize(p->ops.size());
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const BlobOp &o = p->ops[i];
        if (o.kind != OP_STEM && o.kind != OP_CONV) continue;
        if (o.k == 1 && o.stride != 1) {
            delete p;
            return fail(PF_EUNSUPPORTED, "1x1 conv with stride %u", o.stride);
        }
        ConvPlan &c = p->conv[i];
        c.tiling = choose_tiling((int)o.k, (int)o.stride, (int)o.cin, (int)o.cout, 0);
        c.wpk_off = host.size();
        host.resize(host.size() + c.tiling.packed_floats());
        pack_conv_weights(wts + o.w_off, (int)o.cin, (int)o.cout, c.tiling, host.data() + c.wpk_off);
        c.bias_off = host.size();
        const size_t nb = (size_t)c.tiling.cout_blocks * c.tiling.nt * 16;
        host.resize(host.size() + nb, 0.f);
        memcpy(host.data() + c.bias_off, wts + o.b_off, o.cout * sizeof(float));
        int src_ch[kMaxSrc];
        for (uint32_t j = 0; j < o.n_src; ++j) src_ch[j] = (int)o.src[j].ch;
        {
            const int kc = dma_kc((int)o.k, (int)o.stride);
            c.tiled_chunks = dma_chunks(src_ch, (int)o.n_src, (int)o.k, (int)o.stride);
            c.tiled_off = host.size();
            host.resize(host.size() + (size_t)((o.cout + 15) / 16) * c.tiled_chunks * (kc / 4) * o.k * o.k * 64);
            pack_conv_weights_tiled(wts + o.w_off, (int)o.cin, (int)o.cout, (int)o.k, kc, src_ch, (int)o.n_src,
                                    host.data() + c.tiled_off);
        }
        // trailing couts that may run on the vector ALU beside the MFMA tiles (conv_dma.hip); env knobs for A/B runs:
        // PF_VALU_MAX = largest such group (default 8: beyond that the padded MFMA tile measured faster; 0 disables), PF_VALU_PEEL = 1 also peels a full tile of cout % 16 == 0
        static const int valu_max = ab_env("PF_VALU_MAX") ? atoi(ab_env("PF_VALU_MAX")) : 8;
        static const bool valu_peel = ab_env("PF_VALU_PEEL") ? atoi(ab_env("PF_VALU_PEEL")) != 0 : false;
        const int split = (o.k == 3 && o.stride == 1) ? dma_valu_split((int)o.cout, valu_peel) : 0;
        if (split > 0 && split <= valu_max) {
            const int kc = dma_kc(3, 1), rv = dma_rem_rv(split);
            host.resize(align_up(host.size(), 4), 0.f);
            c.rem_off = host.size();
            c.rem_count = split;
            host.resize(host.size() + (size_t)c.tiled_chunks * (kc / 4) * 9 * rv * 4);
            pack_conv_weights_rem(wts + o.w_off, (int)o.cin, (int)o.cout, split, 3, kc, src_ch, (int)o.n_src, host.data() + c.rem_off);
        }
        // the split packings hold fp16 terms of w * 2^k (exact scaling, k per conv: conv_mfma.h)
        std::vector<float> wsc;
        const float *wsplit = wts + o.w_off;
        if ((o.stride == 1 && (o.k == 3 || o.k == 1)) || (o.stride == 2 && o.k == 3 && o.n_src == 1)) {
            const size_t nw = (size_t)o.cout * o.cin * o.k * o.k;
            const float sc = split_weight_scale(wts + o.w_off, nw);
            wsc.resize(nw);
            for (size_t q = 0; q < nw; ++q) wsc[q] = wts[o.w_off + q] * sc;
            wsplit = wsc.data();
            c.split_acc_scale = 1.0f / sc;
        }
        if (o.k == 3 && o.stride == 1) {
            host.resize(align_up(host.size(), 16), 0.f);
            c.split_off = host.size();
            c.split_chunks = split_chunks(src_ch, (int)o.n_src);
            c.has_split = true;
            host.resize(host.size() + split_packed_floats(src_ch, (int)o.n_src, (int)o.cout));
            pack_conv_weights_split(wsplit, (int)o.cin, (int)o.cout, src_ch, (int)o.n_src, host.data() + c.split_off);
        }
        if (o.k == 1 && o.stride == 1) {
            host.resize(align_up(host.size(), 16), 0.f);
            c.split_off = host.size();
            c.split_chunks = split1_chunks(src_ch, (int)o.n_src);
            c.has_split = true;
            host.resize(host.size() + split1_packed_floats(src_ch, (int)o.n_src, (int)o.cout));
            pack_conv_weights_split1(wsplit, (int)o.cin, (int)o.cout, src_ch, (int)o.n_src, host.data() + c.split_off);
        }
        if (o.stride == 1 && o.kind == OP_CONV) {
            S4Range rg[kMaxSrc];
            bool even = true;
            for (uint32_t j = 0; j < o.n_src; ++j) {
                rg[j] = S4Range{(int)o.src[j].choff, (int)o.src[j].ch};
                even = even && (o.src[j].choff & 1) == 0;
            }
            if (even) {
                c.s4_pad = o.k == 1 && o.n_src == 2;
                host.resize(align_up(host.size(), 16), 0.f);
                c.s4_off = host.size();
                c.s4_rounds = s4_rounds(rg, (int)o.n_src, (int)o.k, c.s4_pad);
                c.has_s4 = true;
                host.resize(host.size() + s4_packed_floats(rg, (int)o.n_src, (int)o.cout, (int)o.k, c.s4_pad));
                pack_conv_weights_s4(wsplit, (int)o.cin, (int)o.cout, (int)o.k, rg, (int)o.n_src, c.s4_pad, host.data() + c.s4_off);
            }
        }
        if (i > 0 && is_conv_pair(p, i - 1)) {
            // conv_pair.hip: this op is the consumer C of the pair (P = op i - 1).  C's weights in the K order [S, others.., P], every
            // range padded to whole rounds, P's range declared at channel 0 of its own planes; P's weights as two-instruction rounds + a
            // ninth-tap stream (scaled by P's own 2^k)
            const BlobOp &P = p->ops[i - 1];
            const int n = (int)o.n_src;
            S4Range rg[kMaxSrc];
            int cstart[kMaxSrc], c0s[kMaxSrc], acc0 = 0;
            for (int j = 0; j < n; ++j) { c0s[j] = acc0; acc0 += (int)o.src[j].ch; }
            for (int j = 0; j < n; ++j) {
                const int sj = (j + 1) % n;
                rg[j] = sj == 0 ? S4Range{0, (int)o.src[0].ch} : S4Range{(int)o.src[sj].choff, (int)o.src[sj].ch};
                cstart[j] = c0s[sj];
            }
            host.resize(align_up(host.size(), 16), 0.f);
            c.pair_c_off = host.size();
            c.pair_rounds = s4_rounds(rg, n, 3, 1);
            host.resize(host.size() + s4_packed_floats(rg, n, (int)o.cout, 3, 1));
            const size_t nwp = (size_t)P.cout * P.cin * 9;
            const float scp = split_weight_scale(wts + P.w_off, nwp);
            std::vector<float> wp(nwp);
            for (size_t q = 0; q < nwp; ++q) wp[q] = wts[P.w_off + q] * scp;
            c.pair_merged = conv_pair_merged_supports((int)o.cout, (int)P.cout);
            if (c.pair_merged) {
                // P's couts ride in the rows C's last cout tile pads with zeros (from the next multiple of four on), over the columns of
                // S: C's matrix instructions over S then produce P at the tile's own pixels for nothing.  Harmless for the plain kernel
                // (its epilogue never looks at those rows)
                const int nt = ((int)o.cout + 15) / 16, row0 = ((int)o.cout + 3) / 4 * 4, cS = c0s[1 % n];
                std::vector<float> waug((size_t)nt * 16 * o.cin * 9, 0.f);
                std::copy(wsplit, wsplit + (size_t)o.cout * o.cin * 9, waug.begin());
                for (int pc = 0; pc < (int)P.cout; ++pc)
                    for (int ci = 0; ci < (int)P.cin; ++ci)
                        for (int t = 0; t < 9; ++t) waug[((size_t)(row0 + pc) * o.cin + cS + ci) * 9 + t] = wp[((size_t)pc * P.cin + ci) * 9 + t];
                pack_conv_weights_s4_ex(waug.data(), (int)o.cin, nt * 16, 3, rg, cstart, n, 1, host.data() + c.pair_c_off);
            } else {
                pack_conv_weights_s4_ex(wsplit, (int)o.cin, (int)o.cout, 3, rg, cstart, n, 1, host.data() + c.pair_c_off);
            }
            const S4Range rs{(int)P.src[0].choff, (int)P.src[0].ch};
            host.resize(align_up(host.size(), 16), 0.f);
            c.pair_two_off = host.size();
            host.resize(host.size() + pair_p_two_floats(rs, (int)P.cout));
            host.resize(align_up(host.size(), 16), 0.f);
            c.pair_nine_off = host.size();
            host.resize(host.size() + pair_p_nine_floats(rs, (int)P.cout));
            pack_conv_weights_pair_p(wp.data(), (int)P.cin, (int)P.cout, rs, host.data() + c.pair_two_off, host.data() + c.pair_nine_off);
            c.has_pair = true;
        }
        if (o.k == 3 && o.stride == 2 && o.kind == OP_CONV && o.n_src == 1 && (o.src[0].choff & 3) == 0 && o.cout <= 32) {
            const S4Range rg{(int)o.src[0].choff, (int)o.src[0].ch};
            host.resize(align_up(host.size(), 16), 0.f);
            c.front_off = host.size();
            c.has_front = true;
            host.resize(host.size() + s4_packed_floats(&rg, 1, (int)o.cout, 3, 0));
            pack_conv_weights_s4(wsplit, (int)o.cin, (int)o.cout, 3, &rg, 1, 0, host.data() + c.front_off);
        }
        if (o.stride == 1) {
            c.wave_chunks = wave_chunks(src_ch, (int)o.n_src, (int)o.k);
            c.wave_off = host.size();
            host.resize(host.size() + wave_packed_floats(src_ch, (int)o.n_src, (int)o.cout, (int)o.k));
            pack_conv_weights_wave(wts + o.w_off, (int)o.cin, (int)o.cout, (int)o.k, src_ch, (int)o.n_src,
                                   host.data() + c.wave_off);
        }
        if (o.kind == OP_STEM) {
            c.raw_off = host.size();
            host.insert(host.end(), wts + o.w_off, wts + o.w_off + (size_t)o.cout * o.cin * o.k * o.k);
            // depth channels are the last T of the T*(n_cls+1) inputs (bg_model.py:68-69)
            const int T = (int)o.cin / ((int)h.n_cls + 1), ks2 = (int)(o.k * o.k);
            if (T >= 1 && (uint32_t)(T * ((int)h.n_cls + 1)) == o.cin && o.cout == 16) {
                host.resize(align_up(host.size(), 16), 0.f);
                c.dep_off = host.size();
                for (int tap = 0; tap < ks2; ++tap)
                    for (int t = 0; t < T; ++t)
                        for (int co = 0; co < 16; ++co)
                            host.push_back(wts[o.w_off + ((size_t)co * o.cin + T * h.n_cls + t) * ks2 + tap]);
                host.resize(align_up(host.size(), 16), 0.f);
                c.oh_off = host.size();
                c.has_oh = true;
                for (int tap = 0; tap < ks2; ++tap)
                    for (int t = 0; t < T; ++t)
                        for (int r = 0; r <= (int)h.n_cls; ++r)
                            for (int co = 0; co < 16; ++co)
                                host.push_back(r < (int)h.n_cls ? wts[o.w_off + ((size_t)co * o.cin + t * h.n_cls + r) * ks2 + tap] : 0.f);
            }
        }
        host.resize(align_up(host.size(), 64), 0.f);
    }
    p->readers.assign(p->tensors.size(), 0);
    for (const BlobOp &o : p->ops)
        for (uint32_t j = 0; j < o.n_src; ++j) p->readers[o.src[j].tensor]++;
    p->inv_scale_off.assign(p->tensors.size(), 0);
    for (size_t t = 0; t < p->tensors.size(); ++t) {
        bool any = false;
        for (float v : p->chan_scale[t]) any = any || v != 1.0f;
        if (!any) continue;
        p->inv_scale_off[t] = host.size();
        for (float v : p->chan_scale[t]) host.push_back(1.0f / v);   // exact: powers of two
        host.resize(align_up(host.size(), 64), 0.f);
    }
    p->dev_floats = host.size();
    uint8_t lut[256];
    fill_lut(lut);
    hipError_t e = hipMalloc((void **)&p->dev_weights, (host.size() + 64) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&p->dev_lut, 256);
    if (e == hipSuccess) e = hipMemcpy(p->dev_weights, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->dev_lut, lut, 256, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        pf_hardnet_plan_destroy(p);
        return fail(PF_EHIP, "plan upload: %s", hipGetErrorString(e));
    }
    *out = p;
    return PF_OK;
}

extern "C" void pf_hardnet_plan_destroy(pf_plan *p) {
    if (!p) return;
    if (p->dev_weights) (void)hipFree(p->dev_weights);
    if (p->dev_lut) (void)hipFree(p->dev_lut);
    delete p;
}

extern "C" int pf_hardnet_workspace(const pf_plan *p, int B, int H, int W, size_t *bytes) {
    if (!p || !bytes || B <= 0 || H <= 0 || W <= 0) return fail(PF_EINVAL, "pf_hardnet_workspace: bad argument");
    std::vector<Dims> d;
    int rc = propagate_dims(p, H, W, d);
    if (rc) return rc;
    std::vector<size_t> off;
    return layout(p, B, d, off, *bytes);
}

extern "C" int pf_bg_forward(const pf_plan *p, const void *seg, int seg_is_i64, const float *depth,
                             const uint8_t *depth_mask, float depth_mean, float depth_std, int hop_flags,
                             float min_depth, float max_depth, int B, int T, int H, int W, int out_h, int out_w,
                             void *out_seg, int out_seg_is_i64, float *out_logits, float *out_orig_logits, void *ws,
                             size_t ws_bytes, void *stream) {
    if (!p || !seg || !depth || !out_seg || !ws) return fail(PF_EINVAL, "pf_bg_forward: null pointer argument");
    if (!depth_mask && !(hop_flags & PF_HOP_DEPTH_U16))
        return fail(PF_EINVAL, "pf_bg_forward: depth_mask is required unless PF_HOP_DEPTH_U16 is set");
    if (B <= 0 || T <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0)
        return fail(PF_EINVAL, "pf_bg_forward: bad dims");
    if (p->ops.empty() || p->ops[0].kind != OP_STEM)
        return fail(PF_EUNSUPPORTED, "pf_bg_forward: plan does not start with a fused stem");
    StemArgs st;
    memset(&st, 0, sizeof(st));
    st.seg = seg; st.depth = depth; st.mask = depth_mask;
    st.depth_mean = depth_mean; st.depth_std = depth_std; st.min_depth = min_depth; st.max_depth = max_depth;
    st.seg_is_i64 = seg_is_i64; st.hop = hop_flags; st.B = B; st.T = T; st.n_cls = (int)p->hdr.n_cls;
    st.H = H; st.W = W;
    return run_net(p, &st, nullptr, B, H, W, out_h, out_w, out_seg, out_seg_is_i64, out_logits, out_orig_logits, ws,
                   ws_bytes, (hipStream_t)stream);
}

extern "C" int pf_hardnet_forward_dense(const pf_plan *p, const float *x, int B, int H, int W, int out_h, int out_w,
                                        void *out_seg, int out_seg_is_i64, float *out_logits, float *out_orig_logits,
                                        void *ws, size_t ws_bytes, void *stream) {
    if (!p || !x || !ws) return fail(PF_EINVAL, "pf_hardnet_forward_dense: null pointer argument");
    if (B <= 0 || H <= 0 || W <= 0) return fail(PF_EINVAL, "pf_hardnet_forward_dense: bad dims");
    const bool has_head = !p->ops.empty() && p->ops.back().kind == OP_HEAD;
    if (has_head && (!out_seg || out_h <= 0 || out_w <= 0))
        return fail(PF_EINVAL, "pf_hardnet_forward_dense: out_seg/out size required");
    return run_net(p, nullptr, x, B, H, W, out_h, out_w, out_seg, out_seg_is_i64, out_logits, out_orig_logits, ws,
                   ws_bytes, (hipStream_t)stream);
}

extern "C" int pf_hardnet_status(const void *ws, unsigned *status, void *stream) {
    if (!ws || !status) return fail(PF_EINVAL, "pf_hardnet_status: null argument");
    PF_HIP_CHECK(hipMemcpyAsync(status, ws, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return PF_OK;
}

extern "C" int pf_hardnet_status_sticky(void *ws, unsigned *status, int clear, void *stream) {
    if (!ws || !status) return fail(PF_EINVAL, "pf_hardnet_status_sticky: null argument");
    char *w = (char *)ws + PF_WS_STICKY_OFFSET;
    PF_HIP_CHECK(hipMemcpyAsync(status, w, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (clear) {
        int rc = launch_zero_fill(w, sizeof(unsigned), (hipStream_t)stream);
        if (rc) return rc;
    }
    PF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return PF_OK;
}

extern "C" int pf_hardnet_status_reset(void *ws, void *stream) {
    if (!ws) return fail(PF_EINVAL, "pf_hardnet_status_reset: null argument");
    return launch_zero_fill(ws, PF_WS_STATUS_BYTES, (hipStream_t)stream);
}

extern "C" int pf_hardnet_range_maxima(const pf_plan *p, const void *ws, float *maxima, int cap, int *n_ops, void *stream) {
    if (!p || !ws || !maxima || !n_ops) return fail(PF_EINVAL, "pf_hardnet_range_maxima: null argument");
    *n_ops = (int)p->ops.size();
    if (cap < *n_ops) return fail(PF_EINVAL, "pf_hardnet_range_maxima: room for %d values, the plan has %d ops", cap, *n_ops);
    PF_HIP_CHECK(hipMemcpyAsync(maxima, (const char *)ws + (kSlot0 + kMaxSlots) * 4, p->ops.size() * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return PF_OK;
}

extern "C" int pf_hardnet_tensor_view(const pf_plan *p, const char *name, int B, int H, int W, size_t *ws_offset,
                                      int *channels, int *h, int *w) {
    if (!p || !name || !ws_offset || !channels || !h || !w) return fail(PF_EINVAL, "pf_hardnet_tensor_view: null");
    std::vector<Dims> d;
    int rc = propagate_dims(p, H, W, d);
    if (rc) return rc;
    std::vector<size_t> off;
    size_t total;
    layout(p, B, d, off, total);
    for (size_t t = 0; t < p->tensors.size(); ++t) {
        if (strncmp(p->tensors[t].name, name, sizeof(p->tensors[t].name)) == 0) {
            if (off[t] == (size_t)-1) return fail(PF_EINVAL, "tensor '%s' is not materialised", name);
            *ws_offset = off[t];
            *channels = (int)p->tensors[t].channels;
            *h = d[t].h;
            *w = d[t].w;
            return PF_OK;
        }
    }
    return fail(PF_EINVAL, "no tensor named '%s'", name);
}

__global__ void unscale_channels_kernel(float *x, const float *inv_scale, int C, size_t hw, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        x[i] *= inv_scale[(i / hw) % C];
}

extern "C" int pf_hardnet_tensor_read(const pf_plan *p, const char *name, int B, int H, int W, const void *ws, float *dst,
                                      void *stream) {
    if (!p || !name || !ws || !dst) return fail(PF_EINVAL, "pf_hardnet_tensor_read: null");
    size_t off;
    int c, h, w;
    int rc = pf_hardnet_tensor_view(p, name, B, H, W, &off, &c, &h, &w);
    if (rc) return rc;
    size_t t = 0;
    while (strncmp(p->tensors[t].name, name, sizeof(p->tensors[t].name)) != 0) ++t;
    const char *src = (const char *)ws + off;
    if (t < p->last_fmt.size() && p->last_fmt[t] == 0xFF)
        return fail(PF_EUNSUPPORTED, "tensor '%s' was elided by the fused front end (never stored); set plan option fuse_front = 0 to tap it", name);
    if (t < p->last_fmt.size() && p->last_fmt[t]) {
        if ((rc = launch_s4_unpack(src, dst, B, c, h, w, (hipStream_t)stream))) return rc;
    } else {
        int rc = launch_copy(dst, src, (size_t)B * c * h * w * sizeof(float), (hipStream_t)stream);
        if (rc) return rc;
    }
    // tensors are stored multiplied by the plan's per-channel powers of two (normalize_ranges): undo it for the caller
    if (p->inv_scale_off[t]) {
        const size_t n = (size_t)B * c * h * w;
        hipLaunchKernelGGL(unscale_channels_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0, (hipStream_t)stream,
                           dst, p->dev_weights + p->inv_scale_off[t], c, (size_t)h * w, n);
        PF_LAUNCH_CHECK("unscale_channels_kernel");
    }
    return PF_OK;
}

extern "C" int pf_s4_pack(const float *src, void *dst, int B, int C, int H, int W, unsigned *status, void *stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(PF_EINVAL, "pf_s4_pack: bad argument");
    return launch_s4_pack(src, dst, B, C, H, W, status, (hipStream_t)stream);
}
extern "C" int pf_s4_unpack(const void *src, float *dst, int B, int C, int H, int W, void *stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(PF_EINVAL, "pf_s4_unpack: bad argument");
    return launch_s4_unpack(src, dst, B, C, H, W, (hipStream_t)stream);
}

extern "C" int pf_hardnet_flops(const pf_plan *p, int H, int W, double *flops) {
    if (!p || !flops) return fail(PF_EINVAL, "pf_hardnet_flops: null");
    std::vector<Dims> d;
    int rc = propagate_dims(p, H, W, d);
    if (rc) return rc;
    double f = 0;
    for (const BlobOp &o : p->ops)
        if (o.kind == OP_STEM || o.kind == OP_CONV)
            f += 2.0 * o.cout * d[o.dst].h * d[o.dst].w * o.cin * o.k * o.k;
    *flops = f;
    return PF_OK;
}
