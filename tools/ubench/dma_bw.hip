// Microbenchmark: global -> LDS DMA (global_load_lds_dwordx4) and global -> VGPR (dwordx4) bandwidth as a
// function of the working-set size (L2 / Infinity Cache / HBM) and waves per CU.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void dma16(const float *g, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// each wave streams `iters` x 8 KiB (8 DMA instructions) from its own window of the buffer
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const float *buf, size_t ws_floats, int iters, float *sink) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gw = (size_t)blockIdx.x * (blockDim.x >> 6) + wave;          // global wave id
    float *my = smem + wave * 2048;                                            // 8 KiB per wave
    float acc = 0.f;
    size_t pos = (gw * 2048 * 7) % ws_floats;                                  // de-correlate waves
    for (int it = 0; it < iters; ++it) {
        const float *src = buf + pos;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) dma16(src + k * 256 + lane * 4, my + k * 256);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            float4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4 *>(src + k * 256 + lane * 4);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        pos += 2048 * 4099;           // stride through the working set
        pos %= ws_floats;
        pos &= ~(size_t)2047;
    }
    if (MODE == 0) acc = my[lane];
    if (acc == 123456.789f) sink[0] = acc;
}

int main() {
    const size_t max_bytes = 1ull << 30;
    float *buf, *sink;
    CHECK(hipMalloc(&buf, max_bytes + (1 << 20)));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, max_bytes + (1 << 20)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t sizes[] = {1ull << 20, 4ull << 20, 16ull << 20, 64ull << 20, 512ull << 20};
    const int wgs_per_cu[] = {1, 2, 4};
    for (int mode = 0; mode < 2; ++mode)
        for (size_t ws : sizes)
            for (int wpc : wgs_per_cu) {
                const int grid = 256 * wpc, iters = 400;
                const size_t lds = 4 * 8192;
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), lds, 0, buf, ws / 4, iters, sink);
                    else hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), lds, 0, buf, ws / 4, iters, sink);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                for (int r = 0; r < 5; ++r) launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = 5.0 * grid * 4 * iters * 8192.0;
                printf("%s ws=%5zu MiB  waves/CU=%2d  %8.1f GB/s  (%.1f B/clk/CU @2.4GHz)\n", mode == 0 ? "lds-dma " : "vgpr-x4 ",
                       ws >> 20, wpc * 4, bytes / ms / 1e6, bytes / ms / 1e6 / 256 / 2.4);
            }
    return 0;
}
