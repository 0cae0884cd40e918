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
};

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
void pack_conv_weights_tiled(const float *w_oihw, int cin, int cout, int ks, int kc, float *out);
int launch_conv_dma(const ConvArgs &a, int ks, int stride, int B, hipStream_t stream);

}  // namespace pf
