// Microbenchmark: how v_mfma_f32_16x16x32_f16 and plain vector instructions share a SIMD on gfx950 - the question behind
// conv_front.hip's step (162 matrix + ~500 vector instructions per wave at two waves per SIMD, matrix pipe 44 % busy).
//   per iteration and wave: M matrix instructions (independent accumulators) and V vector FMAs, either interleaved
//   (V / M vector instructions behind every matrix instruction) or in two blocks (all M, then all V).
//   Reported: cycles per iteration and wave (s_memtime, one wave per SIMD and 2 / 4 waves per SIMD), and what the two streams
//   would take alone.
// hipcc -O3 --offload-arch=gfx950 mfma_valu_f16.hip -o mfma_valu_f16 && ./mfma_valu_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int M, int V, int INTERLEAVE>
__global__ __launch_bounds__(256) void k(const float *in, float *out, long long *clk, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[lane + i]; b[i] = (_Float16)in[lane + 64 + i]; }
    float v[8], w = in[lane + 200], x = in[lane + 300];
    for (int i = 0; i < 8; ++i) v[i] = in[lane + 400 + i];
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                if (M) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
                for (int j = 0; j < (M ? V / M : 0); ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 7]) : "v"(w), "v"(x));
            }
            if (!M)
#pragma unroll
                for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 7]) : "v"(w), "v"(x));
        } else {
#pragma unroll
            for (int m = 0; m < M; ++m) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 7]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < V; ++j) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[j & 7]) : "v"(w), "v"(x));
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int M, int V, int IL>
static void run(const char *name, const float *in, float *out, long long *clk, int wg_per_cu) {
    const int iters = 2000, cus = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<M, V, IL>), dim3(cus * wg_per_cu), dim3(256), 0, 0, in, out, clk, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<M, V, IL>), dim3(cus * wg_per_cu), dim3(256), 0, 0, in, out, clk, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long c;
    CHECK(hipMemcpy(&c, clk, sizeof(c), hipMemcpyDeviceToHost));
    const double tf = 2.0 * 16 * 16 * 32 * M * (double)iters * cus * wg_per_cu * 4 / (ms * 1e-3) / 1e12;
    printf("%-34s waves/SIMD %d: %7.3f ms = %6.2f ns per iteration and SIMD, matrix %7.1f TFLOP/s  (first wave: %7.1f clocks per iteration)\n", name, wg_per_cu, ms,
           ms * 1e6 / iters, tf, (double)c / iters);
}

int main() {
    float *in, *out;
    long long *clk;
    CHECK(hipMalloc(&in, 4096 * 4));
    CHECK(hipMalloc(&out, 256 * 256 * 8 * 4));
    CHECK(hipMalloc(&clk, 64));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int wg : {1, 2, 4}) {
        run<8, 0, 0>("8 matrix", in, out, clk, wg);
        run<0, 24, 1>("24 vector", in, out, clk, wg);
        run<8, 24, 1>("8 matrix + 24 vector, interleaved", in, out, clk, wg);
        run<8, 24, 0>("8 matrix + 24 vector, two blocks", in, out, clk, wg);
        run<8, 48, 1>("8 matrix + 48 vector, interleaved", in, out, clk, wg);
        run<8, 8, 1>("8 matrix + 8 vector, interleaved", in, out, clk, wg);
    }
    return 0;
}
