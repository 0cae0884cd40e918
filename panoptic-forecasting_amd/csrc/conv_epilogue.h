// Shared convolution epilogue: D fragment of v_mfma_f32_16x16x4_f32 -> bias -> (+ bilinearly upsampled residual)
// -> ReLU -> (2x2 average pool) -> NCHW store.  Used by conv_dma.hip and conv_wave.hip.
//
// Fragment map (every MFMA conv kernel here): a lane holds 4 consecutive output pixels (oy, ox .. ox+3) of output
// channel co.  The fused stages replace separate passes of the reference network:
//   residual : TransitionUp (hardnet.py:248-258) followed by the 1x1 conv over cat([up(x), skip]) is evaluated as
//              W_skip*skip + up(W_x*x) - bilinear interpolation is linear and per-channel, so it commutes with the
//              1x1 channel mix; W_x*x is computed at the LOW resolution by a preceding launch and sampled here.
//   pool     : nn.AvgPool2d(2,2) after the block-transition 1x1 conv (hardnet.py:296): rows oy, oy+1 of one lane.
#pragma once
#include "conv_mfma.h"

namespace pf {

typedef float epi_f32x4 __attribute__((ext_vector_type(4)));
typedef float epi_f32x2 __attribute__((ext_vector_type(2)));

// align_corners=True source index (ATen area_pixel_compute_scale / compute_source_index)
__device__ __forceinline__ void lin_coord(int o, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
    const float r = scale * (float)o;
    i0 = min((int)r, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

// raw accumulator sums of (oy, ox..ox+3, co) -> finished activations
__device__ __forceinline__ epi_f32x4 epi_finish(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 v) {
    const float bias = a.no_bias ? 0.f : a.bias[co];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bias;
    if (a.res) {
        const float *s = a.res + ((size_t)b * a.res_ctotal + a.res_choff + co) * ((size_t)a.Hres * a.Wres);
        int y0, y1;
        float hy0, hy1;
        lin_coord(oy, a.res_sh, a.Hres, y0, y1, hy0, hy1);
        const float *r0 = s + (size_t)y0 * a.Wres, *r1 = s + (size_t)y1 * a.Wres;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int x0, x1;
            float lx0, lx1;
            lin_coord(ox + r, a.res_sw, a.Wres, x0, x1, lx0, lx1);
            const float t0 = lx0 * r0[x0] + lx1 * r0[x1];
            const float t1 = lx0 * r1[x0] + lx1 * r1[x1];
            v[r] += hy0 * t0 + hy1 * t1;
        }
    }
    if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    return v;
}

__device__ __forceinline__ void epi_store(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 v) {
    float *p = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co) * ((size_t)a.Hout * a.Wout) + (size_t)oy * a.Wout + ox;
    if ((a.Wout & 3) == 0 && ox + 3 < a.Wout) {
        *reinterpret_cast<epi_f32x4 *>(p) = v;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ox + r < a.Wout) p[r] = v[r];
    }
}

// rows oy (even) and oy+1 of the conv output -> row oy/2 of the pooled tensor [.., Hout/2, Wout/2]
__device__ __forceinline__ void epi_store_pooled(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 top, epi_f32x4 bot) {
    const int Hp = a.Hout >> 1, Wp = a.Wout >> 1, py = oy >> 1, px = ox >> 1;
    if (py >= Hp) return;
    float *p = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co) * ((size_t)Hp * Wp) + (size_t)py * Wp + px;
    const float p0 = (((top[0] + top[1]) + bot[0]) + bot[1]) * 0.25f;   // summation order of avgpool2_kernel
    const float p1 = (((top[2] + top[3]) + bot[2]) + bot[3]) * 0.25f;
    if ((Wp & 1) == 0 && px + 1 < Wp) {
        *reinterpret_cast<epi_f32x2 *>(p) = epi_f32x2{p0, p1};
    } else {
        if (px < Wp) p[0] = p0;
        if (px + 1 < Wp) p[1] = p1;
    }
}

}  // namespace pf
