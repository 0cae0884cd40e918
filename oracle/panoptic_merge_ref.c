/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
 *
 * Scalar C restatement of the reference's fg -> panoptic merge, in the reference's own loop structure
 * (instance after instance over the whole image):
 *   /root/reference/panoptic_forecasting/models/fg/fg_model.py:548-588   predict_panoptic pasting loop
 *   /root/reference/panoptic_forecasting/models/fg/fg_model.py:455-480   predict_semantics pasting loop
 *   /root/reference/panoptic_forecasting/models/fg/model_utils.py:30-57  paste_mask
 *   F.grid_sample(bilinear, padding zeros, align_corners=False) as ATen's vectorised CPU kernel evaluates it
 *     (aten/src/ATen/native/cpu/GridSamplerKernel.cpp: unnormalize = (g + 1) * (size/2) - 0.5 contracted to one
 *      FMA; interpolated = nw_v*nw + ne_v*ne + sw_v*sw + se_v*se contracted to mul + 3 FMAs, left to right) —
 *     established by probing torch 2.10 CPU bit for bit (tests/golden/make_golden_fg.py) and pinned by
 *     tests/golden/g5_*.npz, which hold outputs of the reference's own predict_panoptic / predict.
 *   /root/reference/panoptic_forecasting/experiments/export_cityscapes_panoptic_results.py:27-52 (pfo_panoptic_encode)
 *
 * Built with -ffp-contract=off: every fused step is an explicit fmaf().
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float w0, w1; long i0; } tap_t;

static tap_t axis_tap(int p, float o, float d, int size) {
    /* model_utils.py:41-44: img = arange + 0.5; (img - x0) / (x1 - x0) * 2 - 1 */
    float c = ((float)p + 0.5f) - o;
    float g = (c / d) * 2.0f - 1.0f;
    float i = fmaf(g + 1.0f, (float)size / 2.0f, -0.5f);
    float fl = floorf(i);
    tap_t t;
    t.w1 = i - fl;
    t.w0 = 1.0f - t.w1;
    /* the vector float->int conversion gives INT_MIN for NaN / out-of-range: out of bounds either way */
    t.i0 = (fl >= -4.0f && fl <= 1.0e6f) ? (long)fl : -1000;
    return t;
}

static float tap_val(const float *m, int MH, int MW, long y, long x) {
    return (x >= 0 && x < MW && y >= 0 && y < MH) ? m[y * MW + x] : 0.0f;
}

/* order[k] = index of the k-th pasted instance: stable sort by descending depth (seq_depths.sort(descending=True)) */
static void paste_order(const float *depth, int n, int use_sort, int *order) {
    for (int i = 0; i < n; ++i) order[i] = i;
    if (!use_sort) return;
    for (int i = 1; i < n; ++i) {          /* insertion sort: stable */
        int v = order[i], j = i - 1;
        while (j >= 0 && depth[order[j]] < depth[v]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = v;
    }
}

int pfo_panoptic_merge(const int64_t *background /* [B,H,W] or NULL */, const float *bg_depth, const uint8_t *bg_dmask,
                       const float *masks, int MH, int MW, const float *boxes, int ulbr, const float *inst_depth,
                       const int64_t *classes, const int32_t *offsets, int B, int H, int W, int use_sort,
                       int panoptic_ids, int clear_things, int64_t *out) {
    const size_t N = (size_t)H * W;
    float *cur = (float *)malloc(N * sizeof(float));
    uint8_t *pasted = (uint8_t *)malloc(N);
    if (!cur || !pasted) { free(cur); free(pasted); return -1; }
    for (int b = 0; b < B; ++b) {
        int64_t *res = out + (size_t)b * N;
        for (size_t p = 0; p < N; ++p) {
            int64_t v = background ? background[(size_t)b * N + p] : 255;          /* :513-518 */
            if (background && clear_things && v >= 11) v = 255;                    /* :515 */
            res[p] = v;
        }
        const int i0 = offsets[b], n = offsets[b + 1] - i0;
        const int ztest = use_sort && bg_depth != NULL;                            /* :580 */
        if (ztest)
            for (size_t p = 0; p < N; ++p)                                         /* :561-564 */
                cur[p] = (bg_dmask && !bg_dmask[(size_t)b * N + p]) ? 1000000000.0f : bg_depth[(size_t)b * N + p];
        int *order = (int *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int));
        int cl_ids[64];
        memset(cl_ids, 0, sizeof(cl_ids));
        paste_order(inst_depth ? inst_depth + i0 : NULL, n, use_sort, order);
        for (int k = 0; k < n; ++k) {
            const int inst = i0 + order[k];
            const float *bx = boxes + (size_t)inst * 4;
            float x0, y0, x1, y1;
            if (ulbr) { x0 = bx[0]; y0 = bx[1]; x1 = bx[2]; y1 = bx[3]; }
            else {                                                                  /* model_utils.py:36-40 */
                x0 = bx[0] - bx[2] / 2.0f; x1 = bx[0] + bx[2] / 2.0f;
                y0 = bx[1] - bx[3] / 2.0f; y1 = bx[1] + bx[3] / 2.0f;
            }
            const float dx = x1 - x0, dy = y1 - y0;
            const int cls = (int)classes[inst];
            const int inst_id = cl_ids[cls & 63]++;                                /* :568-571 */
            const int64_t seg_val = panoptic_ids ? (int64_t)(cls + 11) * 1000 + inst_id : cls + 11;   /* :572 / :470 */
            const float *m = masks + (size_t)inst * MH * MW;
            /* paste_mask + grid_sample over the whole image, then >= 0.5 (:573) */
            for (int y = 0; y < H; ++y) {
                const tap_t ty = axis_tap(y, y0, dy, MH);
                for (int x = 0; x < W; ++x) {
                    const tap_t tx = axis_tap(x, x0, dx, MW);
                    const float nw = ty.w0 * tx.w0, ne = ty.w0 * tx.w1, sw = ty.w1 * tx.w0, se = ty.w1 * tx.w1;
                    const float nw_v = tap_val(m, MH, MW, ty.i0, tx.i0), ne_v = tap_val(m, MH, MW, ty.i0, tx.i0 + 1);
                    const float sw_v = tap_val(m, MH, MW, ty.i0 + 1, tx.i0), se_v = tap_val(m, MH, MW, ty.i0 + 1, tx.i0 + 1);
                    const float v = fmaf(se_v, se, fmaf(sw_v, sw, fmaf(ne_v, ne, nw_v * nw)));
                    pasted[(size_t)y * W + x] = v >= 0.5f;
                }
            }
            if (ztest) {                                                            /* :580-585 */
                const float d = inst_depth[inst];
                for (size_t p = 0; p < N; ++p)
                    if (pasted[p] && d < cur[p]) { res[p] = seg_val; cur[p] = d; }
            } else {                                                                /* :586-588 */
                for (size_t p = 0; p < N; ++p)
                    if (pasted[p]) res[p] = seg_val;
            }
        }
        free(order);
    }
    free(cur);
    free(pasted);
    return 0;
}

static const int kTrainId2Id[19] = {7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33};

/* export_cityscapes_panoptic_results.py:27-52: convert_labels + create_pan_img; present[id] = 1 for np.unique (:54-68) */
int pfo_panoptic_encode(const int64_t *seg, size_t n, int convert, uint8_t *rgb, int32_t *ids, uint8_t *present, int max_ids) {
    memset(present, 0, (size_t)max_ids);
    for (size_t p = 0; p < n; ++p) {
        int64_t v = seg[p], id = v;
        if (convert) {
            if (v == 255) id = 0;
            else if (v > 100) {
                const int64_t cat = v / 1000, inst = v % 1000;
                id = (cat >= 0 && cat < 19) ? (int64_t)kTrainId2Id[cat] * 1000 + inst : 0;
            } else id = (v >= 0 && v < 19) ? kTrainId2Id[v] : 0;
        }
        if (ids) ids[p] = (int32_t)id;
        if (id >= 0 && id < max_ids) present[id] = 1;
        rgb[3 * p + 0] = (uint8_t)(id % 256);
        rgb[3 * p + 1] = (uint8_t)((id / 256) % 256);
        rgb[3 * p + 2] = (uint8_t)((id / 256 / 256) % 256);
    }
    return 0;
}
