// What does a device-wide barrier inside ONE persistent kernel cost next to a kernel boundary inside a hipGraph?
// (round 5: the B = 1 forward is a dependent chain of 80 launches and rocprofv3 shows a floor of 4.5-5 us per dependent
// launch in the replayed graph - 360 of 1116 us.  The low-resolution HarDBlocks could run as one kernel with grid barriers
// between their layers if a barrier is clearly cheaper than that.)
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Per case (G workgroups x T threads): every phase each workgroup writes `bytes` to its slot of a ping-pong buffer, all
// workgroups meet, each reads the slot of a workgroup on another XCD and checks it (so the barrier has to carry the
// agent-scope release/acquire the layers need: L2s are per XCD).  The same phases as G-block kernels chained in a graph
// give the launch floor on the same box.
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

struct Args {
    float4 *buf;          // [2][G][words]
    unsigned *counter;    // monotonically increasing arrival counter
    unsigned *errors;
    int words;            // float4 per workgroup and phase
    int phases;
};

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {   // bounded: a workgroup that is not resident must not hang the box
                ok = false;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

__device__ __forceinline__ void phase_write(const Args &a, int ph, int wg, int G) {
    float4 *dst = a.buf + ((size_t)(ph & 1) * G + wg) * a.words;
    for (int i = threadIdx.x; i < a.words; i += blockDim.x) dst[i] = float4{(float)ph, (float)wg, (float)i, 1.f};
}
__device__ __forceinline__ void phase_check(const Args &a, int ph, int wg, int G) {
    const int other = (wg + G / 2 + 1) % G;
    const float4 *src = a.buf + ((size_t)(ph & 1) * G + other) * a.words;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < a.words; i += blockDim.x) {
        const float4 v = src[i];
        bad += !(v.x == (float)ph && v.y == (float)other && v.z == (float)i);
    }
    if (bad) atomicAdd(a.errors, bad);
}

__global__ void persistent(Args a) {
    const int G = gridDim.x, wg = blockIdx.x;
    unsigned target = 0;
    for (int ph = 0; ph < a.phases; ++ph) {
        phase_write(a, ph, wg, G);
        target += G;
        if (!grid_barrier(a.counter, target)) {
            if (threadIdx.x == 0) atomicAdd(a.errors, 1u << 20);
            return;
        }
        phase_check(a, ph, wg, G);
        // the next phase writes the OTHER half of the ping-pong buffer: no second barrier needed
    }
}

__global__ void one_phase(Args a, int ph) {
    const int G = gridDim.x, wg = blockIdx.x;
    if (ph > 0) phase_check(a, ph - 1, wg, G);
    phase_write(a, ph, wg, G);
}

static int run_case(int G, int T, int words, int phases) {
    hipStream_t s;
    CK(hipStreamCreate(&s));
    Args a;
    a.words = words;
    a.phases = phases;
    CK(hipMalloc(&a.buf, sizeof(float4) * 2 * G * words));
    CK(hipMalloc(&a.counter, 4));
    CK(hipMalloc(&a.errors, 4));
    CK(hipMemset(a.errors, 0, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent, T, 0));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    if (G > occ * prop.multiProcessorCount) {
        printf("G=%d does not fit (%d x %d)\n", G, occ, prop.multiProcessorCount);
        return 0;
    }
    float ms_p = 0, ms_g = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(a.counter, 0, 4, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(persistent, dim3(G), dim3(T), 0, s, a);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms_p, e0, e1));
    }
    unsigned err = 0;
    CK(hipMemcpy(&err, a.errors, 4, hipMemcpyDeviceToHost));
    // the same phases as a chain of kernels in a graph
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int ph = 0; ph < phases; ++ph) hipLaunchKernelGGL(one_phase, dim3(G), dim3(T), 0, s, a, ph);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms_g, e0, e1));
    }
    unsigned err2 = 0;
    CK(hipMemcpy(&err2, a.errors, 4, hipMemcpyDeviceToHost));
    printf("G=%4d T=%4d %6d B/wg: persistent %6.2f us/phase (errors %u)   graph of kernels %6.2f us/phase (errors %u)\n", G, T,
           words * 16, 1e3 * ms_p / phases, err, 1e3 * ms_g / phases, err2 - err);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipFree(a.buf));
    CK(hipFree(a.counter));
    CK(hipFree(a.errors));
    CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    const int phases = 200;
    for (int G : {32, 64, 128, 256, 512})
        for (int words : {64, 1024})
            if (run_case(G, 512, words, phases)) return 1;
    run_case(256, 256, 256, phases);
    run_case(1024, 256, 256, phases);
    return 0;
}
