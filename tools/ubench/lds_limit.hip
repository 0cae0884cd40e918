// Can a kernel get more than 64 KiB of dynamic LDS on gfx950 / this ROCm?  (design question for conv staging depth)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, int n) {
    extern __shared__ float s[];
    for (int i = threadIdx.x; i < n; i += blockDim.x) s[i] = i;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[n - 1];
}
int main() {
    float *d; hipMalloc(&d, 1024);
    for (int kb : {48, 64, 65, 96, 128, 160}) {
        for (int threads : {256, 1024}) {
            hipError_t e1 = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            hipLaunchKernelGGL(k, dim3(4), dim3(threads), kb * 1024, 0, d, kb * 256);
            hipError_t e2 = hipGetLastError();
            hipError_t e3 = hipDeviceSynchronize();
            printf("%3d KiB, %4d threads: setattr=%s launch=%s sync=%s\n", kb, threads, hipGetErrorName(e1), hipGetErrorName(e2), hipGetErrorName(e3));
        }
    }
    return 0;
}
