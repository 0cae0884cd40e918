// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate with random operands, alone and with an LDS-read stream
// shaped like the convolution inner loop.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: registers only; 1: A operand re-read from LDS every MFMA group (4 reads per 8 MFMAs)
__global__ __launch_bounds__(256) void mfma_kernel(const float *in, float *out, int iters) {
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = in[lane + i * 64];
    for (int i = 0; i < 2; ++i) b[i] = in[lane + 512 + i * 64];
    int off = lane;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = lds[(off + i * 80) & 4095];
            off += 3;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m * 2 + n], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    *(f32x4 *)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = s;
}

int main() {
    float *in, *out;
    CHECK(hipMalloc(&in, 4096 * 4));
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * 16));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode)
        for (int wpc : {1, 2, 3}) {
            const int grid = 256 * wpc, iters = 20000;
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(mfma_kernel<1>, dim3(grid), dim3(256), 0, 0, in, out, iters);
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 3.0 * grid * 4 * (double)iters * 8 * 2048;
            printf("mode %d (%s)  waves/SIMD=%d  %.1f TF/s  (%.1f us per launch)\n", mode, mode ? "A from LDS" : "registers", wpc,
                   flops / ms / 1e9, ms / 3 * 1e3);
        }
    return 0;
}
