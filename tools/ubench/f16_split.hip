// Microbenchmark / probe: is a (hi, mid) fp16 pair a usable replacement for the (hi, mid) bf16 pair of the split convolutions?
//   1. does v_mfma_f32_16x16x32_f16 keep fp16 SUBNORMAL inputs (mid terms of |x| < 0.125 are subnormal)?
//   2. what do v_cvt_f16_f32 / v_cvt_pkrtz_f16_f32 give for subnormal results and beyond 65504?
//   3. f16 vs bf16 16x16x32 instruction rate.
// hipcc -O3 --offload-arch=gfx950 f16_split.hip -o f16_split
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));

// every lane: A[i][k] = a for all i,k; B[k][j] = b  ->  D = 32*a*b
__global__ void denorm_kernel(const float *in, float *out, int n) {
    for (int t = 0; t < n; ++t) {
        const _Float16 a = (_Float16)in[2 * t], b = (_Float16)in[2 * t + 1];
        f16x8 av, bv;
        for (int k = 0; k < 8; ++k) { av[k] = a; bv[k] = b; }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
        if (threadIdx.x == 0) { out[3 * t] = (float)a; out[3 * t + 1] = (float)b; out[3 * t + 2] = acc[0]; }
    }
}

__global__ void cvt_kernel(const float *in, float *out, int n) {
    const int t = threadIdx.x;
    if (t >= n) return;
    const float x = in[t];
    const _Float16 rne = (_Float16)x;
    const fp16x2 rtz = __builtin_amdgcn_cvt_pkrtz(x, x);
    out[4 * t] = x; out[4 * t + 1] = (float)rne; out[4 * t + 2] = (float)rtz[0];
    const float hi = (float)rtz[0];
    const fp16x2 mid = __builtin_amdgcn_cvt_pkrtz(x - hi, 0.f);
    out[4 * t + 3] = hi + (float)mid[0];
}

template <int F16>
__global__ __launch_bounds__(256) void rate_kernel(const float *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 ah[4], bh[2];
    bf16x8 ab[4], bb[2];
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) { ah[i][k] = (_Float16)in[(lane + i * 64 + k) & 4095]; ab[i][k] = (__bf16)in[(lane + i * 64 + k) & 4095]; }
    for (int i = 0; i < 2; ++i) for (int k = 0; k < 8; ++k) { bh[i][k] = (_Float16)in[(lane + 512 + i * 64 + k) & 4095]; bb[i][k] = (__bf16)in[(lane + 512 + i * 64 + k) & 4095]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (F16) acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], acc[m * 2 + n], 0, 0, 0);
                else acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[m], bb[n], acc[m * 2 + n], 0, 0, 0);
            }
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    *(f32x4 *)(out + ((size_t)blockIdx.x * 256 + threadIdx.x) * 4) = s;
}

int main() {
    float *in, *out;
    CHECK(hipMalloc(&in, 4096 * 4));
    CHECK(hipMalloc(&out, (size_t)256 * 8 * 256 * 16));
    // 1. denormal inputs
    {
        const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {5.96e-8f, 1.0f}, {3.0e-5f, 2.0f}, {1.0e-6f, 1024.0f}, {3.0e-5f, 3.0e-5f}, {6.1e-5f, 6.1e-5f}, {0.01f, 1.0e-5f}};
        const int n = sizeof(cases) / sizeof(cases[0]);
        CHECK(hipMemcpy(in, cases, sizeof(cases), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, in, out, n);
        float r[3 * 16];
        CHECK(hipMemcpy(r, out, sizeof(float) * 3 * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            printf("mfma_f16 a=%.9g b=%.9g  D=%.9g  expect 32*a*b=%.9g  %s\n", r[3 * i], r[3 * i + 1], r[3 * i + 2], 32.0 * r[3 * i] * r[3 * i + 1],
                   fabs(r[3 * i + 2] - 32.0 * r[3 * i] * r[3 * i + 1]) <= 1e-6 * fabs(32.0 * r[3 * i] * r[3 * i + 1]) ? "ok" : "DIFFERS");
    }
    // 2. conversions
    {
        const float xs[] = {1.0f, 0.1f, 3.0e-5f, 1.0e-7f, 2.0e-8f, 65504.0f, 65519.0f, 65520.0f, 70000.0f, 131000.0f, 1.0e6f, -1.0e6f, 1.2345678f, 1.0e-3f * 1.2345678f};
        const int n = sizeof(xs) / sizeof(xs[0]);
        CHECK(hipMemcpy(in, xs, sizeof(xs), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, in, out, n);
        float r[4 * 32];
        CHECK(hipMemcpy(r, out, sizeof(float) * 4 * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i)
            printf("cvt x=%.9g  rne=%.9g  rtz=%.9g  rtz hi+mid=%.9g  rel err %.3g\n", r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], (r[4 * i + 3] - r[4 * i]) / r[4 * i]);
    }
    // 3. rate
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int f16 = 0; f16 < 2; ++f16)
        for (int wpc : {1, 2}) {
            const int grid = 256 * wpc, iters = 20000;
            auto launch = [&]() {
                if (f16) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(256), 0, 0, in, out, iters);
                else hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(256), 0, 0, in, out, iters);
            };
            launch();
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < 3; ++r) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 3.0 * grid * 4.0 * iters * 8.0 * 2.0 * 16 * 16 * 32;
            printf("%s 16x16x32, %d workgroup(s)/CU: %.1f TFLOP/s\n", f16 ? "f16 " : "bf16", wpc, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
