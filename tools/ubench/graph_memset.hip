// Does a captured hipMemsetAsync node run on EVERY launch of a hipGraph?  (round 4: in this repository's captures it ran on the
// first launch only when the graph was a single chain of nodes - see DESIGN.md 3.7.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/graph_memset.hip -o /tmp/graph_memset && /tmp/graph_memset
// Per case: capture [pre kernels] -> memset(buf, 0) -> add1(buf) [-> post kernels] on one stream (or with a forked branch), launch the
// graph 4 times, print buf[0] after each launch (1 = the memset ran; n = it did not).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e = (x);                                                       \
        if (e != hipSuccess) {                                                    \
            printf("%s: %s\n", #x, hipGetErrorString(e));                         \
            return 1;                                                             \
        }                                                                         \
    } while (0)

__global__ void add1(unsigned *p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1;
}
__global__ void touch(unsigned *p, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 3 + 1;
}

static int run_case(const char *name, size_t bytes, int pre, int post, bool fork, bool two_memsets, bool autofree = false, bool d2d = false, bool destroy_early = false, bool null_launch = false, size_t sub_off = 0) {
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    unsigned *buf, *other, *third;
    const size_t n = bytes / 4;
    unsigned *base;
    CK(hipMalloc(&base, bytes + sub_off));      // sub_off: the target lies INSIDE a larger allocation (a caching allocator's block)
    buf = base + sub_off / 4;
    CK(hipMalloc(&other, 1 << 20));
    CK(hipMalloc(&third, bytes));
    CK(hipMemset(buf, 0, bytes));
    CK(hipMemset(other, 0, 1 << 20));
    CK(hipMemset(third, 0, bytes));
    hipEvent_t e1, e2;
    CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < pre; ++k) hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, other, (size_t)(1 << 18));
    if (fork) {
        CK(hipEventRecord(e1, s));
        CK(hipStreamWaitEvent(s2, e1, 0));
        hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s2, other, (size_t)(1 << 18));
        CK(hipEventRecord(e2, s2));
    }
    CK(hipMemsetAsync(buf, 0, bytes, s));
    if (two_memsets) CK(hipMemsetAsync(third, 0, bytes, s));
    hipLaunchKernelGGL(add1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, buf, n);
    if (two_memsets) hipLaunchKernelGGL(add1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, third, n);
    for (int k = 0; k < post; ++k) hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, s, other, (size_t)(1 << 18));
    if (d2d) CK(hipMemcpyAsync(other + (1 << 17), other, 64, hipMemcpyDeviceToDevice, s));
    if (fork) CK(hipStreamWaitEvent(s, e2, 0));
    CK(hipStreamEndCapture(s, &g));
    if (autofree) CK(hipGraphInstantiateWithFlags(&ge, g, hipGraphInstantiateFlagAutoFreeOnLaunch));   // what torch.cuda.CUDAGraph does
    else CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    if (destroy_early) CK(hipGraphDestroy(g));      // torch.cuda.CUDAGraph keeps only the executable graph
    printf("%-58s", name);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipGraphLaunch(ge, null_launch ? (hipStream_t)0 : s));
        CK(hipDeviceSynchronize());
        unsigned h[2] = {0, 0};
        CK(hipMemcpy(&h[0], buf, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&h[1], buf + n - 1, 4, hipMemcpyDeviceToHost));
        unsigned t = 0;
        if (two_memsets) CK(hipMemcpy(&t, third, 4, hipMemcpyDeviceToHost));
        printf("  launch %d: %u/%u%s", rep + 1, h[0], h[1], two_memsets ? (t == 1 ? "+ok" : "+BAD") : "");
    }
    printf("\n");
    CK(hipGraphExecDestroy(ge));
    if (!destroy_early) CK(hipGraphDestroy(g));
    CK(hipFree(base));
    CK(hipFree(other));
    CK(hipFree(third));
    return 0;
}

int main() {
    int rv = 0;
    rv |= run_case("chain: memset 4 KB -> add1", 4096, 0, 0, false, false);
    rv |= run_case("chain: memset 64 MB -> add1", 64u << 20, 0, 0, false, false);
    rv |= run_case("chain: 3 kernels -> memset 4 KB -> add1 -> 3 kernels", 4096, 3, 3, false, false);
    rv |= run_case("chain: 3 kernels -> memset 64 MB -> add1 -> 3 kernels", 64u << 20, 3, 3, false, false);
    rv |= run_case("chain: 40 kernels -> memset 1 MB -> add1 -> 40 kernels", 1 << 20, 40, 40, false, false);
    rv |= run_case("chain: two memsets 1 MB back to back -> add1, add1", 1 << 20, 2, 2, false, true);
    rv |= run_case("fork : 3 kernels -> (branch) memset 1 MB -> add1 -> join", 1 << 20, 3, 3, true, false);
    rv |= run_case("chain, AutoFreeOnLaunch: 3 k -> memset 1 MB -> add1 -> 3 k", 1 << 20, 3, 3, false, false, true);
    rv |= run_case("chain, AutoFreeOnLaunch: 500 k -> memset 64 MB -> add1 -> 500", 64u << 20, 500, 500, false, false, true);
    rv |= run_case("chain: 500 k -> memset 64 MB -> add1 -> 500 k", 64u << 20, 500, 500, false, false, false);
    rv |= run_case("chain, AutoFree + D2D copy: 3 k -> memset 1 MB -> add1 -> 3 k", 1 << 20, 3, 3, false, false, true, true);
    rv |= run_case("chain, AutoFree, 2 memsets + D2D: 8 k -> ... -> 1000 k", 16u << 20, 8, 1000, false, true, true, true);
    rv |= run_case("chain, source graph destroyed after instantiate: 3 k -> memset", 1 << 20, 3, 3, false, false, false, false, true);
    rv |= run_case("chain, AutoFree, source graph destroyed: 3 k -> memset 1 MB", 1 << 20, 3, 3, false, false, true, false, true);
    rv |= run_case("fork , AutoFree, source graph destroyed: 3 k -> memset 1 MB", 1 << 20, 3, 3, true, false, true, false, true);
    rv |= run_case("chain, launched on the NULL stream: 3 k -> memset 1 MB", 1 << 20, 3, 3, false, false, false, false, false, true);
    rv |= run_case("chain, AutoFree + destroyed + NULL stream: 3 k -> memset 1 MB", 1 << 20, 3, 3, false, false, true, false, true, true);
    rv |= run_case("fork , AutoFree + destroyed + NULL stream: 3 k -> memset 1 MB", 1 << 20, 3, 3, true, false, true, false, true, true);
    rv |= run_case("chain, target 64 MB into a larger allocation: 3 k -> memset 1 MB", 1 << 20, 3, 3, false, false, false, false, false, false, 64u << 20);
    rv |= run_case("chain, torch-like (AutoFree, destroyed, NULL, sub-alloc): memset 1 MB", 1 << 20, 3, 3, false, false, true, false, true, true, (64u << 20) + 512);
    rv |= run_case("fork , torch-like (AutoFree, destroyed, NULL, sub-alloc): memset 1 MB", 1 << 20, 3, 3, true, false, true, false, true, true, (64u << 20) + 512);
    return rv;
}
