// fg -> panoptic merge for gfx950: depth-ordered pasting of forecast instance masks over the
// background canvas, and the id/colour encoding of the exported panoptic PNG.  HBM-bound, integer-exact.
//
// Replaces, in reference panoptic_forecasting/:
//   models/fg/fg_model.py:548-588   predict_panoptic's pasting loop (seg value (class+11)*1000 + per-class instance id)
//   models/fg/fg_model.py:455-480   predict_semantics' pasting loop  (seg value  class+11)
//   models/fg/model_utils.py:30-57  paste_mask: box -> normalised grid -> F.grid_sample(bilinear, zeros, align_corners=False)
//   experiments/export_cityscapes_panoptic_results.py:27-68  convert_labels, create_pan_img, get_segments_info
//
// The reference pastes instance after instance: 2 full-image grid_samples + ~8 full-image elementwise passes per
// instance (an 80-instance street scene touches ~6 GB).  Here every output pixel walks the instance list once:
//
//   merge_prep_kernel   one workgroup per image: stable descending depth rank (fg_model.py:560-562, ATen's CPU sort is
//                       stable), per-class running instance ids in paste order (:568-572), box corners exactly as
//                       paste_mask forms them (:33-40), and a conservative pixel bounding box per instance.
//   merge_kernel        one workgroup per 8x128 pixel tile, 4 consecutive pixels per lane; instances whose bounding box
//                       misses the tile are skipped on the scalar unit; the rest are sampled in the reference's exact
//                       fp32 order (this file is built with -ffp-contract=off; the two fused steps ATen's vectorised
//                       CPU kernel has are explicit fmaf) so that ">= 0.5" decides identically:
//                         g  = ((p + 0.5 - x0) / (x1 - x0)) * 2 - 1                 model_utils.py:41-44 (4 roundings)
//                         i  = fmaf(g + 1, size/2, -0.5)                            GridSamplerKernel.cpp unnormalize
//                         v  = fmaf(se_v, se, fmaf(sw_v, sw, fmaf(ne_v, ne, nw_v * nw)))     bilinear, zeros padding
//                       then the sequential paste rule per pixel (:574-588): with a background depth map an instance
//                       replaces the pixel iff its depth is strictly nearer than what is there; without one, later
//                       (nearer) instances overwrite earlier ones.
//   panoptic_encode_kernel  trainId-based panoptic ids -> Cityscapes ids (255 -> 0; >100: id2label; else trainId2label),
//                       RGB = (id % 256, id // 256, id // 65536) and a presence byte per id for segments_info.
#include "pf_common.h"
#include "pf_prof.h"

namespace pf {

constexpr int kMergeTH = 8, kMergeTW = 128;   // tile of one workgroup: 8 rows x 32 lanes x 4 px
constexpr int kMaxIds = 34000;                // panoptic ids < 34 * 1000

struct InstRec {          // 48 B, written by merge_prep_kernel in paste order
    float x0, y0, dx, dy; // box origin and extent (x1 - x0, y1 - y0) as paste_mask computes them
    float depth;
    int seg_val;          // value pasted where the mask is >= 0.5
    int mask_idx;         // row of `masks`
    int empty;            // zero/non-finite extent: samples nothing (the reference's grid is inf/NaN there -> all taps out of range)
    int px_lo, px_hi, py_lo, py_hi;   // conservative pixel bounds (inclusive) outside which every tap is out of the mask
};

struct MergeArgs {
    const void *background;   // [B,H,W] u8 / i32 / i64 or null (canvas = 255)
    const float *bg_depth;    // [B,H,W] or null
    const uint8_t *bg_dmask;  // [B,H,W] or null
    const float *masks;       // [N,MH,MW] probabilities (already sigmoid-ed, fg_model.py:532)
    const float *boxes;       // [N,4] (cx,cy,w,h) or (x0,y0,x1,y1)
    const float *inst_depth;  // [N] or null
    const long long *classes; // [N] thing class 0..7
    const int *offsets;       // [B+1]
    InstRec *recs;            // [N]
    void *out;                // [B,H,W] i32 or i64
    int bg_kind;              // 0 u8, 1 i32, 2 i64
    int B, H, W, MH, MW, ulbr, sort_by_depth, panoptic, clear_things, out_is_i64;
};

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_prep_kernel(MergeArgs a) {
    const int b = blockIdx.x;
    const int i0 = a.offsets[b], n = a.offsets[b + 1] - i0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float di = a.sort_by_depth ? a.inst_depth[i0 + i] : 0.f;
        const long long ci = a.classes[i0 + i];
        int rank = i;
        if (a.sort_by_depth) {   // stable descending: everything deeper, or equally deep and earlier, is pasted first
            rank = 0;
            for (int j = 0; j < n; ++j) {
                const float dj = a.inst_depth[i0 + j];
                rank += (dj > di || (dj == di && j < i)) ? 1 : 0;
            }
        }
        int inst_id = 0;         // how many instances of this class were pasted before this one (:568-571)
        for (int j = 0; j < n; ++j) {
            if (a.classes[i0 + j] != ci || j == i) continue;
            bool before = j < i;
            if (a.sort_by_depth) {
                const float dj = a.inst_depth[i0 + j];
                before = dj > di || (dj == di && j < i);
            }
            inst_id += before ? 1 : 0;
        }
        InstRec r;
        const float *bx = a.boxes + (size_t)(i0 + i) * 4;
        float x0, y0, x1, y1;
        if (a.ulbr) {
            x0 = bx[0]; y0 = bx[1]; x1 = bx[2]; y1 = bx[3];
        } else {                 // model_utils.py:36-40: w/2 is exact, one rounding per corner
            const float hw = bx[2] / 2.f, hh = bx[3] / 2.f;
            x0 = bx[0] - hw; x1 = bx[0] + hw; y0 = bx[1] - hh; y1 = bx[1] + hh;
        }
        r.x0 = x0; r.y0 = y0; r.dx = x1 - x0; r.dy = y1 - y0;
        r.depth = di;
        r.seg_val = a.panoptic ? ((int)ci + 11) * 1000 + inst_id : (int)ci + 11;
        r.mask_idx = i0 + i;
        const bool fin = isfinite(x0) && isfinite(y0) && isfinite(r.dx) && isfinite(r.dy) && fabsf(x0) < 1e7f && fabsf(y0) < 1e7f &&
                         fabsf(r.dx) < 1e7f && fabsf(r.dy) < 1e7f;
        r.empty = (!fin || r.dx == 0.f || r.dy == 0.f) ? 1 : 0;
        // taps exist only for -1 < i < size, i.e. p + 0.5 strictly inside [x0 - |dx|/size, x1 + |dx|/size] (either orientation);
        // one extra pixel each side absorbs the fp32 roundings of the exact evaluation
        if (!r.empty) {
            const float mx = fabsf(r.dx) / (float)a.MW + 1.5f, my = fabsf(r.dy) / (float)a.MH + 1.5f;
            const float xl = fminf(x0, x1) - mx, xh = fmaxf(x0, x1) + mx, yl = fminf(y0, y1) - my, yh = fmaxf(y0, y1) + my;
            r.px_lo = (int)fmaxf(floorf(xl), 0.f);
            r.px_hi = (int)fminf(ceilf(xh), (float)(a.W - 1));
            r.py_lo = (int)fmaxf(floorf(yl), 0.f);
            r.py_hi = (int)fminf(ceilf(yh), (float)(a.H - 1));
            if (xh < 0.f || yh < 0.f || xl > (float)a.W || yl > (float)a.H) r.empty = 1;   // wholly outside the image
        } else {
            r.px_lo = r.py_lo = 1; r.px_hi = r.py_hi = 0;
        }
        a.recs[i0 + rank] = r;
    }
}

// position along one axis: source index of the low tap, its weight pair, exactly as ATen's CPU kernel forms them
struct AxisTap {
    int i0;        // floor(i)
    float w1, w0;  // w1 = i - floor(i) (weight of tap i0+1), w0 = 1 - w1
};

__device__ __forceinline__ AxisTap axis_tap(int p, float o, float d, float half_size) {
    const float c = ((float)p + 0.5f) - o;     // img_x - x0
    const float g = (c / d) * 2.f - 1.f;       // / (x1 - x0) * 2 - 1
    const float i = fmaf(g + 1.f, half_size, -0.5f);
    const float fl = floorf(i);
    AxisTap t;
    t.w1 = i - fl;
    t.w0 = 1.f - t.w1;
    // int conversion saturates for the huge values a near-degenerate box gives; any such tap is out of range anyway
    t.i0 = (int)fminf(fmaxf(fl, -4.f), 1e6f);
    return t;
}

template <typename OutT>
__global__ __launch_bounds__(256) void merge_kernel(MergeArgs a) {
    const int b = blockIdx.z;
    const int ty0 = blockIdx.y * kMergeTH, tx0 = blockIdx.x * kMergeTW;
    const int y = ty0 + (threadIdx.x >> 5), x = tx0 + (threadIdx.x & 31) * 4;
    const bool live = y < a.H && x < a.W;     // W % 4 == 0 is checked by the host entry
    const size_t pix = ((size_t)b * a.H + (live ? y : 0)) * a.W + (live ? x : 0);

    int val[4] = {255, 255, 255, 255};
    float cur[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.background && live) {   // one vector load per lane whatever the canvas dtype
        if (a.bg_kind == 0) {
            const uchar4 v = *reinterpret_cast<const uchar4 *>(reinterpret_cast<const uint8_t *>(a.background) + pix);
            val[0] = v.x; val[1] = v.y; val[2] = v.z; val[3] = v.w;
        } else if (a.bg_kind == 1) {
            const int4 v = *reinterpret_cast<const int4 *>(reinterpret_cast<const int *>(a.background) + pix);
            val[0] = v.x; val[1] = v.y; val[2] = v.z; val[3] = v.w;
        } else {
            const longlong2 *p = reinterpret_cast<const longlong2 *>(reinterpret_cast<const long long *>(a.background) + pix);
            const longlong2 v0 = p[0], v1 = p[1];
            val[0] = (int)v0.x; val[1] = (int)v0.y; val[2] = (int)v1.x; val[3] = (int)v1.y;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (a.clear_things && val[k] >= 11) val[k] = 255;     // fg_model.py:515
    }
    if (a.bg_depth && live) {
        const float4 dv = *reinterpret_cast<const float4 *>(a.bg_depth + pix);
        cur[0] = dv.x; cur[1] = dv.y; cur[2] = dv.z; cur[3] = dv.w;
        if (a.bg_dmask) {                                          // :563-564
            const uchar4 mv = *reinterpret_cast<const uchar4 *>(a.bg_dmask + pix);
            if (!mv.x) cur[0] = 1000000000.f;
            if (!mv.y) cur[1] = 1000000000.f;
            if (!mv.z) cur[2] = 1000000000.f;
            if (!mv.w) cur[3] = 1000000000.f;
        }
    }
    const bool ztest = a.sort_by_depth && a.bg_depth != nullptr;
    const float hsx = (float)a.MW / 2.f, hsy = (float)a.MH / 2.f;

    const int i0 = a.offsets[b], i1 = a.offsets[b + 1];
    for (int i = i0; i < i1; ++i) {
        const InstRec *rp = a.recs + i;      // uniform address: scalar loads
        if (rp->empty || rp->px_hi < tx0 || rp->px_lo >= tx0 + kMergeTW || rp->py_hi < ty0 || rp->py_lo >= ty0 + kMergeTH) continue;
        if (!live || y < rp->py_lo || y > rp->py_hi || x + 3 < rp->px_lo || x > rp->px_hi) continue;
        const float *m = a.masks + (size_t)rp->mask_idx * a.MH * a.MW;
        const AxisTap ty = axis_tap(y, rp->y0, rp->dy, hsy);
        const bool rn = ty.i0 >= 0 && ty.i0 < a.MH, rs = ty.i0 + 1 >= 0 && ty.i0 + 1 < a.MH;
        if (!rn && !rs) continue;
        const float *mn = m + (rn ? ty.i0 : 0) * a.MW, *ms = m + (rs ? ty.i0 + 1 : 0) * a.MW;
        const float d = rp->depth;
        const int sv = rp->seg_val;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const AxisTap tx = axis_tap(x + k, rp->x0, rp->dx, hsx);
            const bool cw = tx.i0 >= 0 && tx.i0 < a.MW, ce = tx.i0 + 1 >= 0 && tx.i0 + 1 < a.MW;
            const float nw_v = (rn && cw) ? mn[tx.i0] : 0.f, ne_v = (rn && ce) ? mn[tx.i0 + 1] : 0.f;
            const float sw_v = (rs && cw) ? ms[tx.i0] : 0.f, se_v = (rs && ce) ? ms[tx.i0 + 1] : 0.f;
            const float nw = ty.w0 * tx.w0, ne = ty.w0 * tx.w1, sw = ty.w1 * tx.w0, se = ty.w1 * tx.w1;
            const float v = fmaf(se_v, se, fmaf(sw_v, sw, fmaf(ne_v, ne, nw_v * nw)));
            bool hit = v >= 0.5f;
            if (ztest) {
                hit = hit && d < cur[k];
                cur[k] = hit ? d : cur[k];
            }
            val[k] = hit ? sv : val[k];
        }
    }
    if (!live) return;
    OutT *o = reinterpret_cast<OutT *>(a.out) + pix;
    if (sizeof(OutT) == 4) {
        *reinterpret_cast<int4 *>(o) = make_int4(val[0], val[1], val[2], val[3]);
    } else {
        reinterpret_cast<longlong2 *>(o)[0] = make_longlong2(val[0], val[1]);
        reinterpret_cast<longlong2 *>(o)[1] = make_longlong2(val[2], val[3]);
    }
}

// ------------------------------------------------------------------------------------------------
__constant__ uint8_t kPanTrainId2Id[19] = {7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33};

struct EncodeArgs {
    const void *seg;     // [B,H,W] i32 or i64 trainId-based panoptic ids
    uint8_t *rgb;        // [B,H,W,3]
    int *ids;            // nullable [B,H,W] converted ids
    uint8_t *present;    // [B,kMaxIds] zeroed by the entry point
    size_t n_per_image;
    int B, seg_is_i64, convert;
};

__device__ __forceinline__ int pan_convert(int v, int convert) {
    if (!convert) return v;
    if (v == 255) return 0;                                   // export_cityscapes_panoptic_results.py:31-32
    if (v > 100) {                                            // :33-37
        const int cat = v / 1000, inst = v % 1000;
        return (cat >= 0 && cat < 19) ? (int)kPanTrainId2Id[cat] * 1000 + inst : 0;
    }
    return (v >= 0 && v < 19) ? (int)kPanTrainId2Id[v] : 0;   // :38-39
}

__global__ __launch_bounds__(256) void panoptic_encode_kernel(EncodeArgs a) {
    const int b = blockIdx.y;
    const size_t n4 = a.n_per_image >> 2;
    const size_t base = (size_t)b * a.n_per_image;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        int v[4];
        if (a.seg_is_i64) {
            const longlong2 *s = reinterpret_cast<const longlong2 *>(reinterpret_cast<const long long *>(a.seg) + base) + 2 * i;
            const longlong2 s0 = s[0], s1 = s[1];
            v[0] = (int)s0.x; v[1] = (int)s0.y; v[2] = (int)s1.x; v[3] = (int)s1.y;
        } else {
            const int4 s = reinterpret_cast<const int4 *>(reinterpret_cast<const int *>(a.seg) + base)[i];
            v[0] = s.x; v[1] = s.y; v[2] = s.z; v[3] = s.w;
        }
        unsigned c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int id = pan_convert(v[k], a.convert);
            v[k] = id;
            if (id >= 0 && id < kMaxIds) a.present[(size_t)b * kMaxIds + id] = 1;   // same constant from every writer
            c[k] = ((unsigned)id & 255u) | ((((unsigned)id >> 8) & 255u) << 8) | ((((unsigned)id >> 16) & 255u) << 16);
        }
        // 4 pixels x 3 bytes = three 32-bit words
        uint3 w;
        w.x = c[0] | (c[1] << 24);
        w.y = (c[1] >> 8) | (c[2] << 16);
        w.z = (c[2] >> 16) | (c[3] << 8);
        reinterpret_cast<uint3 *>(a.rgb + base * 3)[i] = w;
        if (a.ids) reinterpret_cast<int4 *>(a.ids + base)[i] = make_int4(v[0], v[1], v[2], v[3]);
    }
}

}  // namespace pf

extern "C" int pf_panoptic_merge_workspace(int n_instances, size_t *bytes) {
    if (!bytes || n_instances < 0) return pf::fail(PF_EINVAL, "pf_panoptic_merge_workspace: bad arguments");
    *bytes = pf::align_up((size_t)(n_instances > 0 ? n_instances : 1) * sizeof(pf::InstRec), 256);
    return PF_OK;
}

extern "C" int pf_panoptic_merge(const void *background, int bg_kind, const float *bg_depth, const uint8_t *bg_depth_mask,
                                 const float *masks, int MH, int MW, const float *boxes, int box_is_ulbr,
                                 const float *inst_depth, const int64_t *classes, const int32_t *inst_offsets,
                                 int n_instances, int B, int H, int W, int use_depth_sorting, int panoptic_ids,
                                 int clear_things, void *out, int out_is_i64, void *ws, size_t ws_bytes, void *stream) {
    if (!out || !inst_offsets || B <= 0 || H <= 0 || W <= 0 || n_instances < 0)
        return pf::fail(PF_EINVAL, "pf_panoptic_merge: bad arguments (B=%d H=%d W=%d n=%d)", B, H, W, n_instances);
    if (n_instances > 0 && (!masks || !boxes || !classes || !ws || MH <= 0 || MW <= 0))
        return pf::fail(PF_EINVAL, "pf_panoptic_merge: instances given without masks/boxes/classes/workspace");
    if (use_depth_sorting && n_instances > 0 && !inst_depth)
        return pf::fail(PF_EINVAL, "pf_panoptic_merge: use_depth_sorting needs inst_depth");
    if (bg_kind < 0 || bg_kind > 2) return pf::fail(PF_EINVAL, "pf_panoptic_merge: bg_kind must be 0 (u8), 1 (i32) or 2 (i64)");
    if (W % 4) return pf::fail(PF_EUNSUPPORTED, "pf_panoptic_merge: W must be a multiple of 4 (got %d)", W);
    size_t need = 0;
    pf_panoptic_merge_workspace(n_instances, &need);
    if (n_instances > 0 && ws_bytes < need)
        return pf::fail(PF_EWORKSPACE, "pf_panoptic_merge: workspace %zu B < required %zu B", ws_bytes, need);
    pf::MergeArgs a;
    a.background = background; a.bg_depth = bg_depth; a.bg_dmask = bg_depth_mask; a.masks = masks; a.boxes = boxes;
    a.inst_depth = inst_depth; a.classes = (const long long *)classes; a.offsets = inst_offsets; a.recs = (pf::InstRec *)ws;
    a.out = out; a.bg_kind = bg_kind; a.B = B; a.H = H; a.W = W; a.MH = MH; a.MW = MW; a.ulbr = box_is_ulbr ? 1 : 0;
    a.sort_by_depth = use_depth_sorting ? 1 : 0; a.panoptic = panoptic_ids ? 1 : 0; a.clear_things = clear_things ? 1 : 0;
    a.out_is_i64 = out_is_i64 ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_instances > 0) {
        pf::ProfScope ps(s, "pf::merge_prep_kernel(pf::MergeArgs)", 0.0, (double)n_instances * (sizeof(pf::InstRec) + 32.0));
        hipLaunchKernelGGL(pf::merge_prep_kernel, dim3(B), dim3(256), 0, s, a);
        PF_LAUNCH_CHECK("merge_prep_kernel");
    }
    const double px = (double)B * H * W;
    const double bg_b = background ? (bg_kind == 0 ? 1.0 : bg_kind == 1 ? 4.0 : 8.0) : 0.0;
    const double bytes = px * (bg_b + (bg_depth ? 4.0 : 0.0) + (bg_depth_mask ? 1.0 : 0.0) + (out_is_i64 ? 8.0 : 4.0));
    const dim3 grid((W + pf::kMergeTW - 1) / pf::kMergeTW, (H + pf::kMergeTH - 1) / pf::kMergeTH, B);
    if (out_is_i64) {
        pf::ProfScope ps(s, "void pf::merge_kernel<long long>(pf::MergeArgs)", 0.0, bytes);
        hipLaunchKernelGGL(pf::merge_kernel<long long>, grid, dim3(256), 0, s, a);
    } else {
        pf::ProfScope ps(s, "void pf::merge_kernel<int>(pf::MergeArgs)", 0.0, bytes);
        hipLaunchKernelGGL(pf::merge_kernel<int>, grid, dim3(256), 0, s, a);
    }
    PF_LAUNCH_CHECK("merge_kernel");
    return PF_OK;
}

extern "C" int pf_panoptic_encode(const void *seg, int seg_is_i64, int convert_to_ids, int B, int H, int W, uint8_t *out_rgb,
                                  int32_t *out_ids, uint8_t *out_present, void *stream) {
    if (!seg || !out_rgb || !out_present || B <= 0 || H <= 0 || W <= 0)
        return pf::fail(PF_EINVAL, "pf_panoptic_encode: bad arguments");
    const size_t n = (size_t)H * W;
    if (n % 4) return pf::fail(PF_EUNSUPPORTED, "pf_panoptic_encode: H*W must be a multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = pf::launch_zero_fill(out_present, (size_t)B * pf::kMaxIds, s)) return rc;
    pf::EncodeArgs a{seg, out_rgb, out_ids, out_present, n, B, seg_is_i64 ? 1 : 0, convert_to_ids ? 1 : 0};
    size_t blocks = ((n >> 2) + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    pf::ProfScope ps(s, "pf::panoptic_encode_kernel(pf::EncodeArgs)", 0.0,
                     (double)B * n * ((seg_is_i64 ? 8.0 : 4.0) + 3.0 + (out_ids ? 4.0 : 0.0)));
    hipLaunchKernelGGL(pf::panoptic_encode_kernel, dim3((unsigned)blocks, B), dim3(256), 0, s, a);
    PF_LAUNCH_CHECK("panoptic_encode_kernel");
    return PF_OK;
}

extern "C" int pf_panoptic_max_ids(void) { return pf::kMaxIds; }
