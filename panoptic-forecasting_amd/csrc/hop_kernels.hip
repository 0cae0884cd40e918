// The on-disk hop between the reference's tasks, device side — HBM-bound elementwise kernels.
//
//   hop_export_kernel : what export_results does to a prediction before PIL/cv2 write it
//                       (experiments/export_cityscapes_segmentation_results.py):
//                         seg   :27-32  convert_labels        trainId -> label id  (zeros_like init: others -> 0)
//                               :34-38  convert_labels_to_trainid  label id -> trainId (others -> 0)
//                               :108-110 astype(uint8)
//                         depth :119-121 round(clamp(d + 1, 0, 255) * 256) -> uint16
//   hop_load_kernel   : what BGDataset.__getitem__ does to the u16 depth it reads back
//                       (data/datasets/bg_dataset.py:224-228,166-170): x/256 - 1, mask = d > 0, d[~mask] = -1,
//                       clamp masked values to [min_depth, max_depth].
//
// The fused forecast path (task bg_forecast) performs the same arithmetic in registers inside the stem kernel
// (net_kernels.hip); these kernels exist so that the two-stage, file-based pipeline of the reference
// (scripts/bg/run_export_bg_val.sh -> data/bg/* -> task bg) runs on the device up to the PNG encoder, with 3 B per
// pixel crossing PCIe instead of 5, and so that tests can prove "in-register hop == file hop" bit for bit.
// One lane = 4 consecutive pixels (16-B depth loads, 8-B u16 stores, 4-B label stores).
#include "pf_common.h"
#include "pf_prof.h"

namespace pf {

typedef float hop_f4 __attribute__((ext_vector_type(4)));

// Cityscapes label table (public dataset constants, cityscapesscripts.helpers.labels)
__constant__ uint8_t kTrainId2Id[19] = {7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 31, 32, 33};

__device__ __forceinline__ uint8_t hop_label(int v, int mode) {
    if (mode == 0) return (uint8_t)v;                                   // --no_convert
    if (mode == 1) return (v >= 0 && v < 19) ? kTrainId2Id[v] : 0;      // convert_labels
    // convert_labels_to_trainid: every id of the table, unlisted ids keep the zeros_like init
    if (v < 0 || v > 33) return 0;
    switch (v) {
        case 7: return 0; case 8: return 1; case 11: return 2; case 12: return 3; case 13: return 4;
        case 17: return 5; case 19: return 6; case 20: return 7; case 21: return 8; case 22: return 9;
        case 23: return 10; case 24: return 11; case 25: return 12; case 26: return 13; case 27: return 14;
        case 28: return 15; case 31: return 16; case 32: return 17; case 33: return 18;
        default: return 255;
    }
}

__device__ __forceinline__ uint16_t hop_quant(float d) {
    return (uint16_t)rintf(fminf(fmaxf(d + 1.f, 0.f), 255.f) * 256.f);   // <= 65280: fits
}

struct HopExportArgs {
    const void *seg;      // [n] u8 or i64 (nullable)
    const float *depth;   // [n] (nullable)
    uint8_t *out_seg;     // [n]
    uint16_t *out_depth;  // [n]
    size_t n;
    int seg_is_i64, mode;
};

__global__ __launch_bounds__(256) void hop_export_kernel(HopExportArgs a) {
    const size_t n4 = a.n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if (a.seg) {
            uchar4 o;
            if (a.seg_is_i64) {
                const longlong2 *s = reinterpret_cast<const longlong2 *>(a.seg) + 2 * i;
                const longlong2 s0 = s[0], s1 = s[1];
                o = make_uchar4(hop_label((int)s0.x, a.mode), hop_label((int)s0.y, a.mode), hop_label((int)s1.x, a.mode),
                                hop_label((int)s1.y, a.mode));
            } else {
                const uchar4 s = reinterpret_cast<const uchar4 *>(a.seg)[i];
                o = make_uchar4(hop_label(s.x, a.mode), hop_label(s.y, a.mode), hop_label(s.z, a.mode), hop_label(s.w, a.mode));
            }
            reinterpret_cast<uchar4 *>(a.out_seg)[i] = o;
        }
        if (a.depth) {
            const hop_f4 d = reinterpret_cast<const hop_f4 *>(a.depth)[i];
            reinterpret_cast<ushort4 *>(a.out_depth)[i] = make_ushort4(hop_quant(d[0]), hop_quant(d[1]), hop_quant(d[2]), hop_quant(d[3]));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {   // ragged tail
        const size_t i = (a.n & ~(size_t)3) + threadIdx.x;
        if (a.seg) {
            const int v = a.seg_is_i64 ? (int)reinterpret_cast<const long long *>(a.seg)[i] : (int)reinterpret_cast<const uint8_t *>(a.seg)[i];
            a.out_seg[i] = hop_label(v, a.mode);
        }
        if (a.depth) a.out_depth[i] = hop_quant(a.depth[i]);
    }
}

struct HopLoadArgs {
    const uint16_t *q;
    float *depth;
    uint8_t *mask;
    size_t n;
    float min_depth, max_depth;
};

__device__ __forceinline__ float hop_decode(uint16_t q, float lo, float hi, uint8_t &m) {
    float d = (float)q / 256.f - 1.f;
    const bool mk = d > 0.f;
    m = mk ? 1 : 0;
    if (!mk) return -1.f;
    d = d > hi ? hi : d;    // _clamp_depths order: upper bound first, then lower (bg_dataset.py:166-170)
    d = d < lo ? lo : d;
    return d;
}

__global__ __launch_bounds__(256) void hop_load_kernel(HopLoadArgs a) {
    const size_t n4 = a.n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const ushort4 q = reinterpret_cast<const ushort4 *>(a.q)[i];
        uchar4 m;
        hop_f4 d;
        d[0] = hop_decode(q.x, a.min_depth, a.max_depth, m.x);
        d[1] = hop_decode(q.y, a.min_depth, a.max_depth, m.y);
        d[2] = hop_decode(q.z, a.min_depth, a.max_depth, m.z);
        d[3] = hop_decode(q.w, a.min_depth, a.max_depth, m.w);
        reinterpret_cast<hop_f4 *>(a.depth)[i] = d;
        reinterpret_cast<uchar4 *>(a.mask)[i] = m;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = (a.n & ~(size_t)3) + threadIdx.x;
        uint8_t m;
        a.depth[i] = hop_decode(a.q[i], a.min_depth, a.max_depth, m);
        a.mask[i] = m;
    }
}

static int hop_grid(size_t n) {
    const size_t blocks = ((n >> 2) + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks));
}

}  // namespace pf

extern "C" int pf_hop_export(const void *seg, int seg_is_i64, int seg_mode, const float *depth, size_t n,
                             uint8_t *out_seg, uint16_t *out_depth_u16, void *stream) {
    if ((!seg && !depth) || (seg && !out_seg) || (depth && !out_depth_u16))
        return pf::fail(PF_EINVAL, "pf_hop_export: each given input needs its output buffer (and at least one input)");
    if (seg_mode < 0 || seg_mode > 2) return pf::fail(PF_EINVAL, "pf_hop_export: seg_mode must be 0, 1 or 2, got %d", seg_mode);
    if (n == 0) return PF_OK;
    const uintptr_t al = (uintptr_t)seg | (uintptr_t)depth | (uintptr_t)out_seg | (uintptr_t)out_depth_u16;
    if (al & 15) return pf::fail(PF_EINVAL, "pf_hop_export: buffers must be 16-byte aligned");
    pf::HopExportArgs a{seg, depth, out_seg, out_depth_u16, n, seg_is_i64 ? 1 : 0, seg_mode};
    hipStream_t s = (hipStream_t)stream;
    const double bytes = (double)n * ((seg ? (seg_is_i64 ? 8.0 : 1.0) + 1.0 : 0.0) + (depth ? 6.0 : 0.0));
    pf::ProfScope ps(s, "pf::hop_export_kernel(pf::HopExportArgs)", 0.0, bytes);
    hipLaunchKernelGGL(pf::hop_export_kernel, dim3(pf::hop_grid(n)), dim3(256), 0, s, a);
    PF_LAUNCH_CHECK("hop_export_kernel");
    return PF_OK;
}

extern "C" int pf_hop_load(const uint16_t *depth_u16, size_t n, float min_depth, float max_depth, float *out_depth,
                           uint8_t *out_mask, void *stream) {
    if (!depth_u16 || !out_depth || !out_mask) return pf::fail(PF_EINVAL, "pf_hop_load: null pointer argument");
    if (n == 0) return PF_OK;
    if (((uintptr_t)depth_u16 | (uintptr_t)out_depth | (uintptr_t)out_mask) & 15)
        return pf::fail(PF_EINVAL, "pf_hop_load: buffers must be 16-byte aligned");
    pf::HopLoadArgs a{depth_u16, out_depth, out_mask, n, min_depth, max_depth};
    hipStream_t s = (hipStream_t)stream;
    pf::ProfScope ps(s, "pf::hop_load_kernel(pf::HopLoadArgs)", 0.0, 7.0 * (double)n);
    hipLaunchKernelGGL(pf::hop_load_kernel, dim3(pf::hop_grid(n)), dim3(256), 0, s, a);
    PF_LAUNCH_CHECK("hop_load_kernel");
    return PF_OK;
}
