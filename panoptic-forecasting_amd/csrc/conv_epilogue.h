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
typedef __attribute__((address_space(3))) float lds_float;   // explicit LDS pointers: ds_read, not flat loads

// align_corners=True source index (ATen area_pixel_compute_scale / compute_source_index)
__device__ __forceinline__ void lin_coord(int o, float scale, int in_size, int &i0, int &i1, float &l0, float &l1) {
    const float r = scale * (float)o;
    i0 = min((int)r, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(r - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}

// ---- residual staged in LDS ---------------------------------------------------------------------------------------
// The residual tensor is sampled 16 times per output fragment; gathered straight from memory that is 16 uncoalesced
// loads per lane per fragment and costs more than the upsample pass it replaces.  Instead the workgroup copies the
// source window of its output tile ([channels of the tile][rows][cols], a few KB) into LDS once, coalesced, and the
// lanes sample from there.  Channel stride is odd so the 16 channels of a fragment column hit 16 different banks.

// upper bound of source rows (cols) touched by n_out consecutive output rows (cols) at `scale` (host + device)
__host__ __device__ inline int res_extent(int n_out, float scale) { return (int)(scale * (float)(n_out - 1)) + 3; }
__host__ __device__ inline int res_chan_stride(int rows, int cols) { return (rows * cols) | 1; }

struct ResWin {
    int sy0, sx0, rows, cols, cs;   // window origin in the residual tensor, its size, channel stride (floats)
};

// window = every (y0, y0+1) x (x0, x0+1) tap pair of the tile; rows/cols past the tensor edge are staged as copies
// of the edge (their interpolation weight is 0 up to rounding, exactly like the clamped index of lin_coord)
__device__ __forceinline__ ResWin res_window(const ConvArgs &a, int oy0, int th, int ox0, int tw) {
    ResWin w;
    const int oy1 = min(oy0 + th, a.Hout) - 1, ox1 = min(ox0 + tw, a.Wout) - 1;
    w.sy0 = min((int)(a.res_sh * (float)oy0), a.Hres - 1);
    w.sx0 = min((int)(a.res_sw * (float)ox0), a.Wres - 1);
    w.rows = min((int)(a.res_sh * (float)oy1), a.Hres - 1) + 2 - w.sy0;
    w.cols = min((int)(a.res_sw * (float)ox1), a.Wres - 1) + 2 - w.sx0;
    w.cs = res_chan_stride(w.rows, w.cols);
    return w;
}

// all `nthr` threads of the workgroup: channels [co0, co0+nco) of the window -> lds[c*cs + r*cols + x]
__device__ __forceinline__ void res_stage(const ConvArgs &a, const ResWin &w, int b, int co0, int nco, lds_float *lds,
                                          int tid, int nthr) {
    const int per = w.rows * w.cols;
    const size_t plane = (size_t)a.Hres * a.Wres;
    const float *base = a.res + ((size_t)b * a.res_ctotal + a.res_choff + co0) * plane;
    for (int e0 = 0; e0 < per; e0 += 64) {            // a wave walks whole channels: lanes = window elements
        const int e = e0 + (tid & 63);
        const int r = e / w.cols, x = e - r * w.cols;
        const size_t off = (size_t)min(w.sy0 + r, a.Hres - 1) * a.Wres + min(w.sx0 + x, a.Wres - 1);
#pragma unroll 4
        for (int c = tid >> 6; c < nco; c += nthr >> 6)
            if (e < per && co0 + c < a.Cout) lds[c * w.cs + e] = base[(size_t)c * plane + off];
    }
}

// interpolation taps of the 4 pixels (oy, ox..ox+3) inside the staged window: shared by every channel of the tile
struct ResTaps {
    int o0[4], o1[4];     // float offsets (from the channel's base) of the (y0, x0) and (y1, x0) taps; x1 = x0 + 1
    float lx1[4], hy1;
};
__device__ __forceinline__ ResTaps res_taps(const ConvArgs &a, const ResWin &w, int oy, int ox) {
    ResTaps t;
    int y0, y1;
    float hy0;
    lin_coord(oy, a.res_sh, a.Hres, y0, y1, hy0, t.hy1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int x0, x1;
        float lx0;
        lin_coord(min(ox + r, a.Wout - 1), a.res_sw, a.Wres, x0, x1, lx0, t.lx1[r]);
        t.o0[r] = (y0 - w.sy0) * w.cols + (x0 - w.sx0);
        t.o1[r] = (y1 - w.sy0) * w.cols + (x0 - w.sx0);
    }
    return t;
}
__device__ __forceinline__ epi_f32x4 res_apply(const lds_float *chan, const ResTaps &t, epi_f32x4 v) {
    const float hy0 = 1.f - t.hy1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lx0 = 1.f - t.lx1[r];
        const float t0 = lx0 * chan[t.o0[r]] + t.lx1[r] * chan[t.o0[r] + 1];
        const float t1 = lx0 * chan[t.o1[r]] + t.lx1[r] * chan[t.o1[r] + 1];
        v[r] += hy0 * t0 + t.hy1 * t1;
    }
    return v;
}

// raw accumulator sums of (oy, ox..ox+3, co) -> finished activations.  chan/taps: this channel's staged residual
// window and the pixel group's taps (nullptr: no residual)
// `bias` is loaded by the caller BEFORE its main loop (epi_bias): fetched here it is a cold, dependent global load
// at the very end of every workgroup - measured at 8-9 us of a 25 us workgroup in conv_dma (tools/probe).
__device__ __forceinline__ float epi_bias(const ConvArgs &a, int co) {
    return (a.no_bias || co >= a.Cout) ? 0.f : a.bias[co];
}
__device__ __forceinline__ epi_f32x4 epi_finish(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 v, float bias,
                                                bool has_res = false, const lds_float *chan = nullptr, const ResTaps *taps = nullptr) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bias;
    // (an explicit flag, not chan != nullptr: the window may sit at LDS address 0, which IS the null pointer there)
    if (has_res) v = res_apply(chan, *taps, v);   // residual: always from the staged LDS window (the launchers refuse
                                               // shapes whose window does not fit; a gather from memory here costs
                                               // ~100 VGPRs of addresses and halves the occupancy of the whole kernel)
    if (a.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    return v;
}

// ---- S4 stores (ConvArgs::dst_fmt == 1) ---------------------------------------------------------------------------
// The D fragment gives a quad of lanes (same lane >> 2) 4 consecutive output channels x 4 consecutive pixels.  A 4x4
// transpose inside the quad (two butterfly steps of DPP quad_perm moves) turns that into "lane k = pixel ox + k, 4
// consecutive channels" = one 8-B unit of the packed layout per term.  EVERY lane of a quad must arrive here: callers
// skip fragments with epi_skip(), which keeps lanes whose channel is past Cout (they contribute zeros) in S4 mode.
__device__ __forceinline__ bool epi_skip(const ConvArgs &a, int co) { return co >= a.Cout && !a.dst_fmt; }

__device__ __forceinline__ float quad_swap1(float x) {   // value of lane ^ 1
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_swap2(float x) {   // value of lane ^ 2
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));
}
// lane k holds row k of a 4x4 matrix -> lane k holds column k
__device__ __forceinline__ epi_f32x4 quad_transpose(epi_f32x4 v) {
    const int k = threadIdx.x & 3;
    const bool hi2 = (k & 2) != 0, hi1 = (k & 1) != 0;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const float lo = v[r], up = v[r + 2];
        const float got = quad_swap2(hi2 ? lo : up);
        v[r] = hi2 ? got : lo;
        v[r + 2] = hi2 ? up : got;
    }
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const float lo = v[r], up = v[r + 1];
        const float got = quad_swap1(hi1 ? lo : up);
        v[r] = hi1 ? got : lo;
        v[r + 1] = hi1 ? up : got;
    }
    return v;
}

typedef split_x2 epi_h2;   // fp16 terms of the operand split (conv_mfma.h)
typedef split_x4 epi_h4;

// c = 4 consecutive channels (buffer channels chb .. chb+3, chb even) of the pixel at element offset `pix` of an h x w plane
__device__ __forceinline__ void s4_store_unit(const ConvArgs &a, int b, int chb, size_t plane_px, size_t pix, epi_f32x4 c) {
    epi_h4 hi, mid;
    split_terms4(c, hi, mid);
    const size_t term = (size_t)a.dst_c4 * plane_px * 8;                      // bytes between the hi and the mid block
    char *base = reinterpret_cast<char *>(a.dst) + (size_t)b * 2 * term + pix * 8;
    const bool ok0 = chb < a.dst_limit, ok1 = chb + 2 < a.dst_limit;
    if ((chb & 3) == 0) {
        char *p = base + (size_t)(chb >> 2) * plane_px * 8;
        if (ok1) {
            *reinterpret_cast<epi_h4 *>(p) = hi;
            *reinterpret_cast<epi_h4 *>(p + term) = mid;
        } else if (ok0) {
            *reinterpret_cast<epi_h2 *>(p) = epi_h2{hi[0], hi[1]};
            *reinterpret_cast<epi_h2 *>(p + term) = epi_h2{mid[0], mid[1]};
        }
    } else {   // the range starts in the middle of a group: upper half of one group, lower half of the next
        char *p = base + (size_t)(chb >> 2) * plane_px * 8 + 4;
        if (ok0) {
            *reinterpret_cast<epi_h2 *>(p) = epi_h2{hi[0], hi[1]};
            *reinterpret_cast<epi_h2 *>(p + term) = epi_h2{mid[0], mid[1]};
        }
        p += plane_px * 8 - 4;
        if (ok1) {
            *reinterpret_cast<epi_h2 *>(p) = epi_h2{hi[2], hi[3]};
            *reinterpret_cast<epi_h2 *>(p + term) = epi_h2{mid[2], mid[3]};
        }
    }
}

// vmax: the caller's running max |v| of what it stores (range guard of the operand split, conv_mfma.h: the caller commits it
// once, range_commit, at the end of its epilogue)
__device__ __forceinline__ void epi_store(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 v, float &vmax) {
    vmax = range_acc(vmax, v[0], v[1], v[2], v[3]);
    if (a.dst_fmt) {
        if (co >= a.Cout) v = epi_f32x4{0.f, 0.f, 0.f, 0.f};
        const epi_f32x4 c = quad_transpose(v);
        const int k = threadIdx.x & 3;
        if (ox + k < a.Wout)
            s4_store_unit(a, b, a.dst_choff + (co & ~3), (size_t)a.Hout * a.Wout, (size_t)oy * a.Wout + ox + k, c);
        return;
    }
    float *p = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co) * ((size_t)a.Hout * a.Wout) + (size_t)oy * a.Wout + ox;
    if ((a.Wout & 3) == 0 && ox + 3 < a.Wout) {
        if (a.accum) v += *reinterpret_cast<const epi_f32x4 *>(p);   // training: the gradient of a tensor collects all its consumers'
        *reinterpret_cast<epi_f32x4 *>(p) = v;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (ox + r < a.Wout) p[r] = a.accum ? p[r] + v[r] : v[r];
    }
}

// rows oy (even) and oy+1 of the conv output -> row oy/2 of the pooled tensor [.., Hout/2, Wout/2]
__device__ __forceinline__ void epi_store_pooled(const ConvArgs &a, int b, int co, int oy, int ox, epi_f32x4 top, epi_f32x4 bot, float &vmax) {
    const int Hp = a.Hout >> 1, Wp = a.Wout >> 1, py = oy >> 1, px = ox >> 1;
    const float p0 = (((top[0] + top[1]) + bot[0]) + bot[1]) * 0.25f;   // summation order of avgpool2_kernel
    const float p1 = (((top[2] + top[3]) + bot[2]) + bot[3]) * 0.25f;
    vmax = range_acc(vmax, p0, p1, 0.f, 0.f);
    if (a.dst_fmt) {
        const bool live = co < a.Cout;
        const epi_f32x4 c = quad_transpose(epi_f32x4{live ? p0 : 0.f, live ? p1 : 0.f, 0.f, 0.f});
        const int k = threadIdx.x & 3;
        if (py < Hp && k < 2 && px + k < Wp)
            s4_store_unit(a, b, a.dst_choff + (co & ~3), (size_t)Hp * Wp, (size_t)py * Wp + px + k, c);
        return;
    }
    if (py >= Hp) return;
    float *p = a.dst + ((size_t)b * a.dst_ctotal + a.dst_choff + co) * ((size_t)Hp * Wp) + (size_t)py * Wp + px;
    if ((Wp & 1) == 0 && px + 1 < Wp) {
        *reinterpret_cast<epi_f32x2 *>(p) = epi_f32x2{p0, p1};
    } else {
        if (px < Wp) p[0] = p0;
        if (px + 1 < Wp) p[1] = p1;
    }
}

}  // namespace pf
