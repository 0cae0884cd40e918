// Non-GEMM stages of the bg network — interface (see net_kernels.hip).
#pragma once
#include "pf_common.h"

namespace pf {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct StemArgs {
    const void *seg;       // [B,T,H,W] u8 or i64
    const float *depth;    // [B,T,H,W]
    const uint8_t *mask;   // [B,T,H,W] (ignored with PF_HOP_DEPTH_U16)
    const float *w;        // folded OIHW [16][T*(n_cls+1)][3][3]
    const float *wdep;     // depth-channel columns re-packed [tap][t][16] (uniform 64-B rows for scalar loads)
    const float *woh;      // one-hot rows re-packed [tap][t][n_cls + 1][16], last row of each group zero (nullable)
    const float *bias;     // [16]
    const uint8_t *lut;    // [256] id -> trainId (device)
    float *dst;            // [B,16,Hout,Wout] fp32, or (dst_fmt = 1) the packed-pair layout [B][2][4][Hout][Wout][4] fp16 of conv_mfma.h
    int dst_fmt;           // 1: only the 2x2-outputs-per-lane kernel writes it (stem_writes_s4() tells whether a launch will use that one)
    float depth_mean, depth_std, min_depth, max_depth;
    int seg_is_i64, hop, B, T, n_cls, H, W, Hout, Wout;
    unsigned *status;      // range guard of the operand split (conv_mfma.h): |output| > 65504 raises PF_STATUS_RANGE; nullable
    unsigned *range_slot;  // ... and max |output| goes to this word of the status block (low side of the guard); nullable
    long long *probe;      // PF_PROBE builds only
    int dbg_plane_pad;     // timing experiment only (PF_DBG_PLANE_PAD): extra floats between output planes
};

struct HeadArgs {
    const float *logits;  // [B,C,Hin,Win]
    void *out_seg;        // [B,Hout,Wout] u8 or i64
    float *out_logits;    // nullable [B,C,Hout,Wout]
    int out_is_i64, B, C, Hin, Win, Hout, Wout;
};

int launch_stem(const StemArgs &a, hipStream_t s);
bool stem_writes_s4(const StemArgs &a);   // the kernel launch_stem() picks for these arguments can write dst_fmt = 1
// status / slot (nullable): max |output| of the launch for the low side of the range guard (conv_mfma.h)
int launch_avgpool2(const float *src, float *dst, int planes, int Hin, int Win, unsigned *status, unsigned *slot, hipStream_t s);
int launch_upsample(const float *src, float *dst, int planes, int Hin, int Win, int Hout, int Wout, unsigned *status, unsigned *slot, hipStream_t s);
int launch_head(const HeadArgs &a, hipStream_t s);

}  // namespace pf
