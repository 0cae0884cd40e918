// Zero fill and device-to-device copy as kernels of this library.
//
// Why not hipMemsetAsync / hipMemcpyAsync: every entry point of the library may be captured into a hipGraph by its caller
// (bench.py, bg_train.py).  Under torch.cuda.graph (torch 2.10 with its bundled HIP 7.0.2 runtime) a captured memset NODE inside a
// graph that is one linear chain of nodes did its job on the first replay and filled its range with a stale 64-bit pattern on
// later ones (round 4: the z-buffer slots of the splat and the gradient arenas of the training step held garbage from the second
// replay on; graphs with forked branches were not affected; tools/ubench/graph_memset_torch.py reproduces it, the plain HIP
// program tools/ubench/graph_memset.hip does not).  A captured call of this library therefore holds kernel nodes only;
// tests/test_gpu_graph_replay.py replays every captured path five times against the eager result.
#include "pf_common.h"

namespace pf {

typedef unsigned fill_u32x4 __attribute__((ext_vector_type(4)));

// body: 16-byte pieces of [p16, p16 + n16); edges: the < 16 bytes in front of it and the < 16 bytes behind it
__global__ __launch_bounds__(256) void zero_fill_kernel(fill_u32x4 *p16, long long n16, unsigned char *head, int nhead, unsigned char *tail, int ntail) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) p16[i] = fill_u32x4{0u, 0u, 0u, 0u};
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < nhead) head[threadIdx.x] = 0;
        if ((int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
    }
}

__global__ __launch_bounds__(256) void copy_kernel(fill_u32x4 *d16, const fill_u32x4 *s16, long long n16, unsigned char *dtail, const unsigned char *stail, int ntail) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) d16[i] = s16[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dtail[threadIdx.x] = stail[threadIdx.x];
}

__global__ __launch_bounds__(256) void copy_bytes_kernel(unsigned char *d, const unsigned char *s, long long n) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) d[i] = s[i];
}

static unsigned fill_blocks(long long n16) {
    long long b = (n16 + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

int launch_zero_fill(void *p, size_t bytes, hipStream_t s) {
    if (!bytes) return PF_OK;
    if (!p) return fail(PF_EINVAL, "zero fill: null pointer");
    unsigned char *b = (unsigned char *)p;
    size_t nhead = (16 - ((uintptr_t)b & 15)) & 15;
    if (nhead > bytes) nhead = bytes;
    const size_t n16 = (bytes - nhead) / 16, ntail = bytes - nhead - n16 * 16;
    hipLaunchKernelGGL(zero_fill_kernel, dim3(fill_blocks((long long)n16)), dim3(256), 0, s, (fill_u32x4 *)(b + nhead), (long long)n16, b, (int)nhead,
                       b + nhead + n16 * 16, (int)ntail);
    PF_LAUNCH_CHECK("zero_fill_kernel");
    return PF_OK;
}

int launch_copy(void *dst, const void *src, size_t bytes, hipStream_t s) {
    if (!bytes || dst == src) return PF_OK;
    if (!dst || !src) return fail(PF_EINVAL, "copy: null pointer");
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t n16 = bytes / 16;
        hipLaunchKernelGGL(copy_kernel, dim3(fill_blocks((long long)n16)), dim3(256), 0, s, (fill_u32x4 *)dst, (const fill_u32x4 *)src, (long long)n16,
                           (unsigned char *)dst + n16 * 16, (const unsigned char *)src + n16 * 16, (int)(bytes - n16 * 16));
    } else {
        hipLaunchKernelGGL(copy_bytes_kernel, dim3(fill_blocks((long long)(bytes / 16))), dim3(256), 0, s, (unsigned char *)dst, (const unsigned char *)src,
                           (long long)bytes);
    }
    PF_LAUNCH_CHECK("copy_kernel");
    return PF_OK;
}

}  // namespace pf
