// Reproducer for the hazard conv_pair.hip ran into (profiles/r06_experiments.md 1.2): a v_mfma_f32_16x16x16_f16 issued
// directly behind a v_mfma_f32_16x16x32_f16 on the SAME accumulator registers.  Both instructions are written in inline asm, so
// the compiler's hazard recognizer sees what it sees in conv_pair's generated code (it pads back-to-back matrix instructions with a
// register dependency; the question is whether its padding for this opcode pair is enough on gfx950).
//   variant 0: K32 then K16 back to back on one accumulator (a chain of CH K32 products in front, like a round of conv_pair)
//   variant 1: the same with NOPS s_nop 7 between the two
//   variant 2: K16 into a second accumulator, added afterwards (what conv_pair.hip does now)
//   variant 3: builtins instead of inline asm for the last K32 + the K16 (one more K32 product per iteration: its own reference)
// Every wave computes the same products from the same inputs; the host compares all waves of all workgroups against wave 0 of
// variant 2 and counts differing lanes.  Many workgroups and an LDS read in front of the K16 (as in the kernel: its operand comes
// from LDS, so the issue time of the K16 instruction jitters) make the timing vary.
// hipcc -O3 --offload-arch=gfx950 mfma_k16_after_k32.hip -o mfma_k16_after_k32 && ./mfma_k16_after_k32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int VARIANT>
__global__ __launch_bounds__(512) void k(const float *in, float *out, int iters) {
    __shared__ h4 lds[512];
    const int lane = threadIdx.x & 63;
    h8 a, b;
    h4 c, d;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[lane * 8 + i]; b[i] = (_Float16)in[512 + lane * 8 + i]; }
    for (int i = 0; i < 4; ++i) { c[i] = (_Float16)in[1024 + lane * 4 + i]; d[i] = (_Float16)in[1536 + lane * 4 + i]; }
    lds[threadIdx.x] = d;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc9 = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        // the K16 operand arrives from LDS right before its use (volatile: not hoisted out of the loop)
        const h4 dd = *reinterpret_cast<const volatile h4 *>(&lds[(threadIdx.x + it * 64) & 511]);
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(b), "v"(a));
        if (VARIANT == 0) {
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(c), "v"(dd));
        } else if (VARIANT == 3) {
            // the same through the compiler's builtins: whatever wait states hipcc's hazard recognizer inserts are in
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(c, dd, acc, 0, 0, 0);
        } else if (VARIANT == 1) {
            asm volatile("s_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(c), "v"(dd));
        } else {
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc9) : "v"(c), "v"(dd));
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const f32x4 r = acc + acc9;
    float *o = out + ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    for (int i = 0; i < 4; ++i) o[i] = r[i];
}

int main() {
    const int blocks = 2048, threads = 512, iters = 8;
    std::vector<float> in(2048);
    srand(1);
    for (auto &v : in) v = (float)(rand() % 17 - 8) / 8.0f;      // small exact values: every product and sum is exact in fp32
    float *din, *dout;
    CHECK(hipMalloc(&din, in.size() * 4));
    CHECK(hipMalloc(&dout, (size_t)blocks * threads * 16));
    CHECK(hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ref, got((size_t)blocks * threads * 4);
    for (int variant : {2, 0, 1, 3}) {
        long long bad_total = 0;
        for (int rep = 0; rep < 20; ++rep) {
            CHECK(hipMemset(dout, 0, got.size() * 4));
            if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
            else if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
            else if (variant == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
            else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, din, dout, iters);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
            if (ref.empty() || (variant == 3 && rep == 0)) {
                // reference: the lds index differs per wave and iteration, so take each thread's own value of variant 2's first run
                ref = got;
            }
            long long bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != ref[i];
            bad_total += bad;
        }
        printf("variant %d (%s): %lld differing values in 20 launches of %d x %d threads\n", variant,
               variant == 0 ? "K16 directly behind K32, same accumulator (inline asm: no compiler padding)" : variant == 1 ? "16 wait states between them" :
               variant == 3 ? "builtins: K32, K32, K32, K16 on one accumulator (compiler padding)" : "K16 into its own accumulator",
               bad_total, blocks, threads);
    }
    return 0;
}
