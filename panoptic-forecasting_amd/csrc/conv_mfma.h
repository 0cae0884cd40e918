// Direct (implicit-GEMM) fp32 convolution on the gfx950 matrix cores — interface.
//
// Replaces the ATen/cuDNN convolutions behind reference hardnet.py:16-25 (ConvLayer = conv + BN + ReLU,
// BN folded into weight/bias) for 3x3 (stride 1/2, pad 1) and 1x1 kernels, reading the HarDBlock
// "concatenation" (hardnet.py:225-230) as up to four channel ranges of earlier tensors instead of a
// torch.cat copy.
#pragma once
#include "pf_common.h"

namespace pf {

constexpr int kConvMaxSrc = 4;

struct ConvArgs {
    const float *src[kConvMaxSrc];   // base of each source tensor [B, ctotal, Hin, Win]
    int src_ctotal[kConvMaxSrc];     // channels of the tensor the range lives in
    int src_choff[kConvMaxSrc];      // first channel of the range inside that tensor
    int src_cstart[kConvMaxSrc + 1]; // prefix sums of range lengths (conv input channel numbering)
    int src_chunk0[kConvMaxSrc + 1]; // first K chunk of each range ([n_src] = nchunks) in the launched kernel's chunk
                                     // size; every range is padded to whole chunks: a chunk never straddles tensors
    int n_src;
    const float *wpk;  // packed weights, see pack_conv_weights()
    const float *bias; // [n_tiles_total*16], zero padded
    float *dst;        // [B, dst_ctotal, Hout, Wout]
    int dst_ctotal, dst_choff;
    int Cin, Cout, Hin, Win, Hout, Wout;
    int tilesX, tilesY, nchunks, relu;
    // fast path (conv_dma.hip) only:
    const float *zero_page;  // >= 16 B of zeros (source of out-of-image / padding DMA pieces)
    int ntiles;              // ceil(Cout/16)
    // fused epilogue stages (conv_epilogue.h); zero-initialised = plain conv
    int no_bias;             // skip the bias add (the low-resolution half of a commuted upsample+1x1 conv)
    int pool;                // 2x2 average pool in the epilogue: dst is [B, dst_ctotal, Hout/2, Wout/2]
    const float *res;        // residual [B, res_ctotal, Hres, Wres], bilinearly upsampled (align_corners) and added
    int res_ctotal, res_choff, Hres, Wres;
    float res_sh, res_sw;    // (Hres-1)/(Hout-1), (Wres-1)/(Wout-1)
    int res_lds_off;         // float offset of the staged residual window in the kernel's LDS, or -1: sample from memory
    int res_rows, res_cols;  // conv_s4 1x1: fixed window size per channel (upper bounds of what a tile touches)
    unsigned res_magic_cs, res_magic_cols;   // ... and 2^32 / d + 1 of its channel stride and its columns (lane index -> element)
    const float *wrem;       // conv_dma remainder path: weights of the last `rem` couts, [chunk][kgroup][tap][RV][4 ch]
    int rem;                 // > 0: couts handled on the vector ALU; ntiles then counts FULL 16-cout tiles only
    int src_begin, src_end;      // input ranges to accumulate (whole conv: 0, n_src)
    int chunk_begin, chunk_end;  // = src_chunk0[src_begin], src_chunk0[src_end] (filled at launch)
    long long *probe;        // PF_PROBE builds only: in-kernel timestamps of workgroup 0 / wave 0 (else nullptr)
    unsigned *status;        // range guard of the two-term operand split (below): word 0 of the forward's workspace, or
                             // nullptr (fp32-only plans, training): epilogues OR PF_STATUS_RANGE into it when they store |v| > 65504
    unsigned *range_slot;    // ... and keep max |v| of what this launch stored here (low side of the guard, below); nullable
    int kacc;                // conv_dma 3x3 (training forward): per-round partial sums added into a second accumulator set (blocked summation)
    int accum;               // fp32 NCHW stores of conv_mfma.hip and of epi_store (conv_dma / conv_wave): dst += result (gradient
                             // accumulation of the training path); not combined with rem / pool / S4 destinations
    float acc_scale;         // split kernels (conv_split.hip, conv_s4.hip) only: their weights are packed as fp16 terms of
                             // w * 2^k (k per conv, split_weight_scale()); the raw sums are multiplied by 2^-k (exact)
    // ---- packed-pair ("S4") activation layout, conv_s4.hip: a tensor of C channels is stored as
    //      [B][2 terms: hi, mid][C4 = ceil(C/4)][H][W][4] fp16 with x ~= hi + mid (split_terms() below): the same
    //      4 B per element as fp32 and exactly the two terms the split kernels feed the matrix pipe with
    int dst_fmt;             // 0: fp32 NCHW, 1: S4 (epi_store / epi_store_pooled of conv_epilogue.h)
    int dst_c4;              // channel groups of the dst tensor
    int dst_limit;           // S4: stores cover buffer channels [dst_choff, dst_limit) = dst_choff + Cout, rounded up to a
                             // whole group when this conv also zero-fills the tail of its last group
    int src_fmt;             // 1: every source is S4 (conv_s4 kernels only)
    int src_c4[kConvMaxSrc];      // channel groups of each source tensor
    int src_g0[kConvMaxSrc];      // first group a range touches = src_choff / 4
    int src_gn[kConvMaxSrc];      // number of groups it touches
    int src_ent0[kConvMaxSrc + 1];// first K entry of each range ("group entries": one group of one range; padded per range to
                                  // whole rounds when the conv was packed with pad_sources)
};

// ---- the two-term operand split of conv_split.hip / conv_s4.hip ------------------------------------------------------
// Every fp32 operand x of a convolution is fed to the 16-bit matrix pipe as x ~= hi + mid with TWO fp16 TERMS, both
// rounded to nearest even:
//     hi = fp16_rne(x),  mid = fp16_rne(x - hi)          (x - hi is exact in fp32)
// Operand bound (proved in tests/test_host_logic.py::test_split_operand_bound by exhausting every fp32 exponent, checked
// bit for bit against the device in tests/test_gpu_conv.py::test_s4_layout_round_trip): |x - hi| <= half an fp16 ulp of x,
// so the residual needs at most 12 significand bits below hi's last one and mid (11 bits, nearest) leaves
//     |x - hi - mid| <= 2^-23 |x|                      for 2^-2 <= |x| <= 65504   (fp32's own rounding: 2^-24)
//     |x - hi - mid| <= 2^-25  (half an fp16 subnormal step: absolute)   for |x| < 2^-2
// against 2^-21 |x| for the round-toward-zero pair of round 2 and 2^-17 |x| for a pair of bf16 terms, at the same storage,
// the same instruction rate (v_mfma_f32_16x16x32_f16) and the same three products hi*hi + hi*mid + mid*hi; the dropped
// product mid*mid is <= 2^-22 of a product.  The price is fp16's range:
//   * activations: |x| > 65504 cannot be represented.  It is never silent: every kernel that produces a tensor a split
//     kernel may read (the conv epilogues, the stem, the layout packer) compares what it stores against 65504 and raises
//     bit PF_STATUS_RANGE of the forward's status word (ConvArgs::status = word 0 of the workspace; range_acc / range_commit
//     below); the caller re-runs that forward on the fp32 matrix instructions or fails (pfhip.h: pf_hardnet_status,
//     bg_model.py: on_range_overflow).  FC-HarDNet activations behind folded BatchNorm are O(1..100);
//   * weights: packed on the host (round to nearest even) after an exact per-conv scaling by 2^k that puts max|w| into
//     [2^14, 2^15) - small weights keep both terms in fp16's normal range - and the kernels multiply their raw sums
//     by 2^-k (ConvArgs::acc_scale) before the bias: exact, so the scaling is invisible in the result.
// v_mfma_f32_16x16x32_f16 multiplies fp16 subnormals exactly (measured: tools/ubench/f16_split.hip).
typedef _Float16 split_t;
typedef split_t split_x2 __attribute__((ext_vector_type(2)));
typedef split_t split_x4 __attribute__((ext_vector_type(4)));
typedef split_t split_x8 __attribute__((ext_vector_type(8)));
constexpr float kSplitMaxAbs = 65504.f;   // largest |x| the pair represents (to 2^-23)
// two values at a time: {hi0, hi1}, {mid0, mid1}.  hi: v_cvt_pk_f16_f32 (gfx950) rounds to nearest even, two values per issue.
// mid = fp16(x - hi): one mixed-precision FMA per value (v_fma_mixlo / mixhi_f16: hi read as fp16, x as fp32, the fp32 difference
// - exact - rounded to nearest even into one half of the result register) - 3 instructions per pair where convert back, subtract,
// convert takes 5; the same bits (tests/test_gpu_conv.py::test_s4_layout_round_trip pins the device's patterns to the host's)
__device__ __forceinline__ void split_terms2(float x0, float x1, split_x2 &hi, split_x2 &mid) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    hi = __builtin_convertvector(f32x2{x0, x1}, split_x2);
    const unsigned hu = __builtin_bit_cast(unsigned, hi);
    unsigned mu;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(mu) : "v"(hu), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(mu) : "v"(hu), "v"(x1));
    mid = __builtin_bit_cast(split_x2, mu);
#endif
}
// range guard: m = running max |v| of what a lane stores; one compare + (never taken) branch at the end.  NaN cannot arise
// from finite guarded inputs (products <= 65504 * 2^15, K <= a few thousand: the fp32 accumulators cannot overflow)
// (two v_max3_f32 with |.| source modifiers; written with fmaxf / fabsf hipcc quiets every input first: 7 instructions)
__device__ __forceinline__ float range_acc(float m, float a, float b, float c, float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b));
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(c), "v"(d));
#endif
    return m;
}
// Low side of the same guard.  Below |x| = 2^-2 the pair keeps an ABSOLUTE 2^-25 (mid is an fp16 subnormal), so a tensor
// whose values are all tiny loses relative precision where the reference's fp32 Conv2d (hardnet.py:16-25) does not.  Two
// measures (hardnet_plan.hip): every tensor channel is stored multiplied by a power of two chosen at plan creation so that
// its expected magnitude is kRangeTarget (exact re-parameterisation of the folded weights: producers' rows * s, consumers'
// columns / s), and every launch that produces a split operand reports the maximum |v| of what it stored to its own word of
// the forward's status block (ConvArgs::range_slot); range_finalize_kernel raises PF_STATUS_RANGE_LOW when a launch's
// reported maximum is below kRangeLowMax: the unflagged operand bound is then |x - hi - mid| <= 2^-23 |x| + 2^-25, with
// 2^-25 <= 2^-19 max|tensor| - and the caller re-runs a flagged forward on the fp32 matrix instructions as for overflow.
// Only a SAMPLE of a launch's workgroups reports (range_sampled: 8-15 of them, picked by a hash of the workgroup id):
// agent-scope atomics on one address execute memory-side at ~10 ns each (measured: every wave of a 16 384-workgroup launch
// reporting tripled the kernel's time, and a cached pre-check of the word does not help - the XCD L2s keep serving the value
// from before the atomics).  Sampling is conservative for this test: a tensor whose values are ALL below the threshold is
// below it on every sample, so it is always flagged; a tensor with larger values escapes the flag as soon as one sampled
// workgroup stored one (8 x 32 pixels x all its output channels each).  A reporting wave ORs 1 into the bit pattern, so a word
// of 0 means "no report" (an op without a slot) and a word of 1 "every sampled value was exactly zero": neither is low - exact
// zeros lose nothing in the pair (a dead-ReLU layer, blank frames); the flag needs a NON-ZERO maximum below kRangeLowMax.
constexpr float kRangeTarget = 8.0f;          // expected magnitude of a stored channel after the plan's scaling
constexpr float kRangeLowMax = 0.015625f;     // 2^-6: a launch whose reported max |v| is below it raises PF_STATUS_RANGE_LOW
__device__ __forceinline__ bool range_sampled() {   // uniform per workgroup
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned total = gridDim.x * gridDim.y * gridDim.z;
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned mask = total >= 16u ? (1u << (28 - __builtin_clz(total))) - 1u : 0u;   // keeps one workgroup in 2^floor(log2(total)) / 8
    return (((id * 0x9E3779B1u) >> 9) & mask) == 0u;
#else
    return false;
#endif
}
// m = max |v| over what the lane stored (>= 0; v_max3 drops NaN); lanes that have exited are simply not counted
__device__ __forceinline__ void range_commit(unsigned *status, unsigned *slot, float m) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (status == nullptr) return;
    if (!(m <= kSplitMaxAbs)) atomicOr(status, 1u);   // PF_STATUS_RANGE
    if (slot == nullptr || !range_sampled()) return;
    // wave maximum by a scalar walk over the lanes that exceed the running value (~5 steps for unordered data)
    const unsigned mb = __builtin_bit_cast(unsigned, m);
    unsigned wm = 0;
    unsigned long long above = __builtin_amdgcn_ballot_w64(mb > wm);
    while (above != 0) {
        wm = (unsigned)__builtin_amdgcn_readlane((int)mb, (int)__builtin_ctzll(above));
        above = __builtin_amdgcn_ballot_w64(mb > wm);
    }
    const unsigned lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (lane_id == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicMax(slot, wm | 1u);   // first active lane
#endif
}
template <typename V4>   // any 4-float vector type
__device__ __forceinline__ void split_terms4(const V4 &v, split_x4 &hi, split_x4 &mid) {
    split_x2 h0, m0, h1, m1;
    split_terms2(v[0], v[1], h0, m0);
    split_terms2(v[2], v[3], h1, m1);
    hi = split_x4{h0[0], h0[1], h1[0], h1[1]};
    mid = split_x4{m0[0], m0[1], m1[0], m1[1]};
}
#define PF_MFMA_SPLIT(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
// host side (weight packing): fp32 -> fp16 bits, round to nearest even, subnormals kept; and back
unsigned short split_host_f16(float x);
float split_host_f32(unsigned short h);
// 2^k with max|w| * 2^k in [2^14, 2^15) (1 for an all-zero tensor)
float split_weight_scale(const float *w, size_t n);

// Workgroup -> (pixel tile, cout group) for the grids dim3(pixel tiles, cout groups, B) of the fast kernels.  Workgroups are
// dispatched in linear id order (x fastest) to the 8 XCDs round-robin, and each XCD has its own L2.  With a tile count that
// is a multiple of 8, id % 8 names the XCD and id / 8 the position in that XCD's queue; the queue is laid out as
//     XCD x  <-  a contiguous band of tiles;  consecutive positions = the cout groups of ONE tile, then the next tile
// so that (a) the halo rows / partially used 128-B lines neighbouring tiles share and (b) the whole input tile that the
// cout groups of a tile share (a 1x1 conv with Cout = 119 at NT = 2 is four workgroups reading the same pixels) are read
// from HBM once and then from that XCD's L2.  In plain (x, y) order the cout groups of a tile are a whole image of
// workgroups apart and every one of them streams the input from HBM again.
__device__ __forceinline__ void xcd_tile_order(int ntiles_xy, int &tile_lin, int &group) {
#if defined(__HIP_DEVICE_COMPILE__)
    if ((ntiles_xy & 7) != 0) {
        tile_lin = (int)blockIdx.x;
        group = (int)blockIdx.y;
        return;
    }
    const int ngroups = (int)gridDim.y;
    const int lin = (int)blockIdx.x + ntiles_xy * (int)blockIdx.y;
    const int xcd = lin & 7, pos = lin >> 3;
    const int t = pos / ngroups;
    group = pos - t * ngroups;
    tile_lin = xcd * (ntiles_xy >> 3) + t;
#endif
}

// Tiling choice for one conv (depends on shape only; fixed at plan time for the weight packing).
struct ConvTiling {
    int ks, stride;
    int kc;   // input channels staged per LDS chunk (multiple of 4)
    int nt;   // 16-wide cout tiles per workgroup
    int twt;  // workgroup tile width in 16-pixel units (4: 4x64, 2: 8x32, 1: 16x16 pixels)
    int cout_blocks, nchunks;
    size_t packed_floats() const { return (size_t)cout_blocks * nchunks * (kc / 4) * ks * ks * nt * 64; }
};

ConvTiling choose_tiling(int ks, int stride, int cin, int cout, int wout_hint);
// OIHW fp32 -> [cout_block][chunk][kgroup][tap][nt][64 lanes] with
//   value = W[(cb*nt+t)*16 + (lane&15)][chunk*kc + kg*4 + (lane>>4)][tap]  (0 outside Cin/Cout)
void pack_conv_weights(const float *w_oihw, int cin, int cout, const ConvTiling &t, float *out);
// Enqueue one conv for a batch of B images (generic path: any stride/width).
int launch_conv(const ConvArgs &a, const ConvTiling &t, int B, hipStream_t stream);

// Fast path (Win % 4 == 0): a.wpk must point at pack_conv_weights_tiled() output and
// a.nchunks = ceil(Cin / kc) with kc = dma_kc(ks, stride) input channels per double-buffered stage.
constexpr int dma_kc_ct(int ks, int stride) { return ks == 1 ? 16 : (stride == 2 ? 4 : 8); }
inline int dma_kc(int ks, int stride) { return dma_kc_ct(ks, stride); }
int dma_chunks(const int *src_ch, int n_src, int ks, int stride);
// conv_dma's vector-ALU cout path: the last `rem` (1..16) output channels; accumulators rv in {2,4,8,12,16} >= rem
inline int dma_rem_rv(int rem) { return rem <= 2 ? 2 : (rem + 3) / 4 * 4; }
// how many trailing output channels of a 3x3/s1 conv go to the vector ALU (0 = none): the partial tile, or with
// peel_full a whole 16-wide tile of a conv whose cout is a multiple of 16
inline int dma_valu_split(int cout, bool peel_full) {
    if (cout < 16) return 0;
    const int r = cout % 16;
    if (r == 0) return (peel_full && cout >= 32) ? 16 : 0;
    return r;
}
void pack_conv_weights_rem(const float *w_oihw, int cin, int cout, int rem, int ks, int kc, const int *src_ch, int n_src, float *out);
void pack_conv_weights_tiled(const float *w_oihw, int cin, int cout, int ks, int kc, const int *src_ch, int n_src, float *out);
// force_wm/force_nt > 0 override the cost model (tuning runs); model_B > 0: the cost model sees this batch size
// instead of B (batch-invariant shape choice, plan option "table_batch")
int launch_conv_dma(const ConvArgs &a, int ks, int stride, int B, hipStream_t stream, int force_wm = 0, int force_nt = 0, int model_B = 0);

// Wave-autonomous path (conv_wave.hip; stride 1, Win % 4 == 0): a.wpk must point at pack_conv_weights_wave()
// output and a.nchunks = ceil(Cin / wave_kc(ks)).  Tile = mh rows x 16 pixels, nt cout tiles, wk-way K split.
constexpr int wave_kc_ct(int ks) { return ks == 1 ? 32 : 8; }
inline int wave_kc(int ks) { return wave_kc_ct(ks); }
// src_ch[n_src]: channels of each input range (sum = cin); K order = ranges in order, each padded to whole chunks
int wave_chunks(const int *src_ch, int n_src, int ks);
void pack_conv_weights_wave(const float *w_oihw, int cin, int cout, int ks, const int *src_ch, int n_src, float *out);
size_t wave_packed_floats(const int *src_ch, int n_src, int cout, int ks);
int launch_conv_wave(const ConvArgs &a, int ks, int mh, int nt, int wk, int B, hipStream_t stream);

// split path (conv_split.hip; 3x3/s1, Wout % 4 == 0, no fused epilogue; two fp16 terms per operand, see split_terms2
// above): a.wpk must point at pack_conv_weights_split() output, chunks of 8 channels; the packers take the weights
// ALREADY multiplied by split_weight_scale() and a.acc_scale = 1 / that scale.
int split_chunks(const int *src_ch, int n_src);
size_t split_packed_floats(const int *src_ch, int n_src, int cout);
void pack_conv_weights_split(const float *w_oihw, int cin, int cout, const int *src_ch, int n_src, float *out);
int launch_conv_split(const ConvArgs &a, int nt, int wide, int B, hipStream_t stream);
// 1x1/s1 on the same scheme (chunks of 32 channels; pack_conv_weights_split1()); supports the fused epilogue stages
int split1_chunks(const int *src_ch, int n_src);
size_t split1_packed_floats(const int *src_ch, int n_src, int cout);
void pack_conv_weights_split1(const float *w_oihw, int cin, int cout, const int *src_ch, int n_src, float *out);
int launch_conv_split1(const ConvArgs &a, int nt, int B, hipStream_t stream);

// S4 path (conv_s4.hip; stride 1, W % 4 == 0): every source in the packed-pair layout, tiles arrive by LDS-DMA, no split
// phase.  K order = the group entries of the ranges in order, two entries (8 channels) per 3x3 round, eight (32 channels)
// per 1x1 round; weights of channels a range does not own inside its first/last group are zero.
struct S4Range { int choff, ch; };   // a source range in the channel numbering of its tensor
// pad_sources: every range padded to whole rounds (convs that may be launched one range at a time)
int s4_entries(const S4Range *r, int n_src, int ks, int pad_sources);
int s4_rounds(const S4Range *r, int n_src, int ks, int pad_sources);
size_t s4_packed_floats(const S4Range *r, int n_src, int cout, int ks, int pad_sources);
void pack_conv_weights_s4(const float *w_oihw, int cin, int cout, int ks, const S4Range *r, int n_src, int pad_sources, float *out);
int launch_conv_s4(const ConvArgs &a, int ks, int nt, int wide, int B, hipStream_t stream);
// the same with an explicit start of every range in the conv's input-channel numbering (cstart[j]; nullptr = the ranges are
// consecutive): ranges may then be packed in ANOTHER order than the one they are concatenated in (conv_pair.hip)
void pack_conv_weights_s4_ex(const float *w_oihw, int cin, int cout, int ks, const S4Range *r, const int *cstart, int n_src, int pad_sources, float *out);

// conv_pair.hip: an odd HarDBlock layer P = conv3x3(S) computed inside its consumer C = conv3x3(P ++ S ++ others) (hardnet.py:177-194)
struct PairArgs {
    ConvArgs c;              // the consumer; sources in the K order [S, other ranges.., P's output] (the last one is never read from
                             // memory: src_ent0 / nchunks describe it, its src pointer is unused), weights packed in that order with
                             // pad_sources = 1 and P's range declared as {0, p_cout}
    const float *p_wpk;      // P: pack_conv_weights_pair_p() of its single range: [tile][round][instr 0, instr 1] blocks ...
    const float *p_w9;       // ... and the ninth tap [tile][round][term][64 lanes][4 fp16] (lane group 0 / 1 = the round's entries); bias zero padded
    const float *p_bias;
    float p_acc_scale;
    float *p_dst;            // P's destination (packed pairs): its slice of the block's output tensor
    int p_dst_c4, p_dst_choff, p_dst_limit, p_cout, p_cin, p_ntiles, p_relu;
    unsigned *p_range_slot;  // max |v| of what P stored (range guard, low side); nullable
    int rounds_s;            // rounds of S = P's rounds = C's first rounds
    int round_d;             // C's first round over P's planes
    int merged;              // 1: c.wpk was packed from C's weights WITH P's weights in rows 4 .. 4 + p_cout - 1 of C's last cout tile
                             // (columns of S only; each conv's own 2^k): P's values at the tile's own pixels come out of C's matrix
                             // instructions, conv_pair computes P only on the halo ring.  Needs roundup4(C cout) + P cout <= 16 * tiles
};
bool conv_pair_supports(int c_cout, int p_cout);
bool conv_pair_merged_supports(int c_cout, int p_cout);     // PairArgs::merged possible for these channel counts?
// conv_select.cpp: run this pair as one conv_pair launch?  mode = the plan's fuse_pairs option (1: where measured / modelled faster, 2: wherever possible,
// 3: wherever possible and never merged, 4: the merged pairs only)
bool pair_wanted(int p_cin, int p_cout, int c_cin, int c_cout, int h, int w, int B, int mode);
// P's weights (already scaled by 2^k like every split packing): `two` = [tile][round][2 blocks], `nine` = [tile][round][term][lane][4]
size_t pair_p_two_floats(const S4Range &r, int cout);
size_t pair_p_nine_floats(const S4Range &r, int cout);
void pack_conv_weights_pair_p(const float *w_oihw, int cin, int cout, const S4Range &r, float *two, float *nine);
int launch_conv_pair(const PairArgs &pa, int B, hipStream_t stream);
// layout conversion (tests, tensor taps): fp32 NCHW <-> S4
// status (nullable, device): PF_STATUS_RANGE is raised when an element exceeds what the pair represents (|x| > 65504)
int launch_s4_pack(const float *src, void *dst, int B, int C, int H, int W, unsigned *status, hipStream_t stream);
// raises PF_STATUS_RANGE in *status if any of the n floats at x is not |x| <= 65504 (NaN included); max |x| -> *slot (nullable)
int launch_range_check(const float *x, size_t n, unsigned *status, unsigned *slot, hipStream_t stream);
// status block of a forward's workspace (hardnet_plan.hip): word 0 = published PF_STATUS_* bits of the last forward, word
// sticky_word = OR over all forwards since the host cleared it, word live_word = what the running forward's kernels OR into,
// words first_slot .. + n_slots - 1 = reported max |v| per launch (bit patterns), the next n_slots words = those of the last forward
int launch_range_finalize(unsigned *st, int live_word, int first_slot, int n_slots, int sticky_word, hipStream_t stream);
int launch_s4_unpack(const void *src, float *dst, int B, int C, int H, int W, hipStream_t stream);

// Fused front end (conv_front.hip): 3x3 stride-1 conv (16 -> 24 channels) + 3x3 stride-2 conv (24 -> <= 32) in one kernel; the
// input is the stem output in the packed-pair layout, the tensor between the two convs stays in LDS
struct FrontArgs {
    const void *x;          // stem output, packed pairs [B][2][4][H1][W1][4] fp16
    const void *w1, *w2;    // pack_conv_weights_s4() of the two convs (already scaled by 2^k)
    const float *bias1, *bias2;   // zero padded to 32
    float scale1, scale2;   // 2^-k
    float *dst;             // [B][dst_ctotal][H2][W2] fp32 or packed pairs
    int dst_fmt, dst_c4, dst_ctotal, dst_choff, dst_limit;
    int H1, W1, H2, W2, C1, C2, relu1, relu2;
    int tilesX, tilesY;     // strips x vertical segments (set by the launcher)
    int seg_steps;          // steps (2 output rows each) per segment (set by the launcher)
    unsigned *status;
    unsigned *range_slot_mid, *range_slot;   // max |v| of the tensor between the two convs / of the output (low side of the range guard); nullable
    long long *probe;       // PF_PROBE builds only (tools/probe_front.py), else nullptr
};

bool conv_front_supports(int c0, int c1, int c2, int h1, int w1, int dst_ctotal);
int launch_conv_front(const FrontArgs &a, int B, hipStream_t s);

// Kernel/shape choice for one stride-1 conv (conv_select.cpp): kind 1 = conv_dma (p0 = WM, p1 = NT),
// kind 2 = conv_wave (p0 = MH, p1 = NT, p2 = WK), (kind 3 was conv_valu: the whole 3x3 conv on v_pk_fma_f32, measured and removed - DESIGN.md 3.2),
// kind 4 = conv_split (p0 = NT, p1 = 1: 8x64-pixel tiles), kind 5 = conv_s4 (same parameters; S4 sources).
struct ConvChoice {
    int kind, p0, p1, p2;
};
// need: bit 1 = even tile rows if conv_wave is chosen (pooling epilogue)
ConvChoice choose_conv(int ks, int stride, int cin, int cout, int hout, int wout, int B, int need = 0, int use_tuned = 1);
bool choose_s4(int ks, int cin, int cout, int hout, int wout, int B, ConvChoice *out);   // conv_s4 table row, if any
long long *probe_buffer();
// tuning hook (pf_debug_force_conv): kind 0 = automatic
extern ConvChoice g_conv_force;

}  // namespace pf
