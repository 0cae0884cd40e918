// Pieces shared by the packed-pair ("S4") convolution kernels: conv_s4.hip and conv_pair.hip (an odd HarDBlock layer computed
// inside its consumer).  Layout, operand split and weight stream: conv_mfma.h, conv_s4.hip.
#pragma once
#include "conv_epilogue.h"

namespace pf {

typedef float s4_f32x4 __attribute__((ext_vector_type(4)));
typedef split_x8 s4_h8;   // 8 / 4 fp16 terms (conv_mfma.h: split_terms2)
typedef split_x4 s4_h4;
typedef __attribute__((address_space(3))) void *s4_lds_ptr_t;
[[maybe_unused]] constexpr unsigned kS4Oob = 0x80000000u;

__device__ __forceinline__ s4_h8 s4_join(s4_h4 lo, s4_h4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// group entry e of the conv's K order -> (source tensor frame base, byte offset of the hi plane of its group); all scalar
__device__ __forceinline__ bool s4_entry(const ConvArgs &a, int e, int b, size_t plane_bytes, const char *&base, unsigned &goff,
                                         unsigned &term_stride) {
    const float *sp = a.src[0];
    int c4 = a.src_c4[0], g0 = a.src_g0[0], gn = a.src_gn[0], e0 = 0;
#pragma unroll
    for (int k = 1; k < kConvMaxSrc; ++k) {
        const bool take = k < a.n_src && e >= a.src_ent0[k];
        sp = take ? a.src[k] : sp;
        c4 = take ? a.src_c4[k] : c4;
        g0 = take ? a.src_g0[k] : g0;
        gn = take ? a.src_gn[k] : gn;
        e0 = take ? a.src_ent0[k] : e0;
    }
    // entries past the groups of their range (round padding) have zero weights; the caller fetches nothing for them
    // (out-of-range pieces land as zeros), the address stays inside the tensor anyway
    base = reinterpret_cast<const char *>(sp) + (size_t)b * 2 * c4 * plane_bytes;
    goff = (unsigned)(g0 + min(e - e0, gn - 1)) * (unsigned)plane_bytes;
    term_stride = (unsigned)c4 * (unsigned)plane_bytes;
    return e - e0 < gn;
}

// weight blocks in front of round r (2 per round + one collected-tap block per started group of 4 rounds before it)
__host__ __device__ inline int s4_blocks_before(int r) { return 2 * r + r / 4; }
__host__ __device__ inline int s4_blocks_total(int rounds) { return 2 * rounds + (rounds + 3) / 4; }

}  // namespace pf
