// Sustained fp32 MFMA rate of the chip (no memory traffic): what "peak" means under the clocks MFMA load actually runs at.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, int nacc) {
  v16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0;
  for (int j = 0; j < 16; ++j) s += a0[j] + a1[j] + a2[j] + a3[j];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4096 * 256 * 4);
  for (int wpb : {1, 2}) {
    int blocks = 256 * wpb * 4;   // wpb blocks of 4 waves per CU ... x4 rounds
    int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 100, 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 4);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 4096.0;
      printf("blocks=%d (%d per CU-round) %.2f ms  %.1f TFLOP/s\n", blocks, wpb, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
