// Microbenchmark: sustained fp32 vector-ALU MAC rate on gfx950 for the instruction forms the convolution's vector co-tile
// could use, alone and next to a saturating v_mfma_f32_16x16x4_f32 stream in the SAME wave.
//   MODE 0  v_fma_f32                    (16 independent accumulators)
//   MODE 1  v_fmac_f32_dpp quad_perm     (multiplier picked out of the quad by DPP)
//   MODE 2  v_pk_fma_f32                 (2 MACs per lane per instruction, multiplicand broadcast by op_sel)
//   MODE 3  v_pk_fma_f32 with an SGPR-pair multiplier
// hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE, int WITH_MFMA>
__global__ __launch_bounds__(256) void valu_kernel(const float *in, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 macc[4];
    for (int i = 0; i < 4; ++i) macc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float w[4], x[4];
    for (int i = 0; i < 4; ++i) { w[i] = in[lane + i * 64]; x[i] = in[lane + 256 + i * 64]; }
    const float s0 = in[1024], s1 = in[1025];
    for (int it = 0; it < iters; ++it) {
        if (WITH_MFMA) {
#pragma unroll
            for (int m = 0; m < 4; ++m) macc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[m], w[m], macc[m], 0, 0, 0);
        }
        // 32 MACs per lane per iteration in every mode
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[i & 3]), "v"(x[r]));
        } else if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(w[i & 3]), "v"(x[r]));
        } else if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x2 a = {acc[2 * i], acc[2 * i + 1]}, ww = {w[i & 1], w[2 + (i & 1)]}, xx = {x[r], x[r + 2]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "v"(ww), "v"(xx));
                    acc[2 * i] = a[0]; acc[2 * i + 1] = a[1];
                }
        } else {
            f32x2 sw = {s0, s1};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    f32x2 a = {acc[2 * i], acc[2 * i + 1]}, xx = {x[r], x[r + 2]};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a) : "s"(sw), "v"(xx));
                    acc[2 * i] = a[0]; acc[2 * i + 1] = a[1];
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += macc[i][0] + macc[i][1] + macc[i][2] + macc[i][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WITH_MFMA>
static void run(const float *in, float *out, const char *name) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wpc : {1, 2, 4}) {
        const int grid = 256 * wpc, iters = 20000;
        hipLaunchKernelGGL((valu_kernel<MODE, WITH_MFMA>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((valu_kernel<MODE, WITH_MFMA>), dim3(grid), dim3(256), 0, 0, in, out, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double waves = (double)grid * 4, macs = waves * 64.0 * 32.0 * iters * 3;
        const double mf = WITH_MFMA ? waves * 4.0 * 1024.0 * iters * 3 : 0.0;
        printf("%-28s mfma=%d waves/SIMD=%d  VALU %.1f TF/s  MFMA %.1f TF/s  (%.2f ms)\n", name, WITH_MFMA, wpc, 2.0 * macs / ms / 1e9,
               2.0 * mf / ms / 1e9, ms / 3);
    }
}

int main() {
    float *in, *out;
    CHECK(hipMalloc(&in, 4096 * 4));
    CHECK(hipMalloc(&out, (size_t)1024 * 256 * 4));
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    run<0, 0>(in, out, "v_fma_f32");
    run<1, 0>(in, out, "v_fmac_f32_dpp");
    run<2, 0>(in, out, "v_pk_fma_f32");
    run<3, 0>(in, out, "v_pk_fma_f32 sgpr");
    run<0, 1>(in, out, "v_fma_f32");
    run<1, 1>(in, out, "v_fmac_f32_dpp");
    run<2, 1>(in, out, "v_pk_fma_f32");
    run<3, 1>(in, out, "v_pk_fma_f32 sgpr");
    return 0;
}
