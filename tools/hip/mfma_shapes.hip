// fp32 MFMA shape comparison under the chip's power management: v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32, operands
// in registers or re-read from LDS every step (as a GEMM main loop does), random data, ~100 ms runs.  Reports TFLOP/s and
// the in-kernel clock (s_memtime / s_memrealtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned i) {
  unsigned h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x400000);
}

// MODE 0: 32x32x2, 4 accumulators (64x64 wave tile), operands from LDS (2 A + 2 B ds_read_b32 per k pair)
// MODE 1: 16x16x4, 16 accumulators (64x64 wave tile), operands from LDS (4 A + 4 B ds_read_b32 per k quad)
// MODE 2: 32x32x2 from registers; MODE 3: 16x16x4 from registers
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
  __shared__ float smem[2 * 16 * 129 * 2];
  for (int i = threadIdx.x; i < 2 * 16 * 129 * 2; i += 256) smem[i] = rnd(i + blockIdx.x * 7919);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if (MODE == 0 || MODE == 2) {
    const int l31 = lane & 31, h = lane >> 5;
    const float* as = smem + (wave / 2) * 64 + l31;
    const float* bs = smem + 2 * 16 * 129 + (wave % 2) * 64 + l31;
    v16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float ra[2] = {rnd(lane), rnd(lane + 64)}, rb[2] = {rnd(lane + 128), rnd(lane + 192)};
    for (int it = 0; it < iters; ++it) {
      const float* a_ = as + (it & 1) * 16 * 129;
      const float* b_ = bs + (it & 1) * 16 * 129;
#pragma unroll
      for (int kk = 0; kk < 16; kk += 2) {
        float av[2], bv[2];
        if (MODE == 0) {
          av[0] = a_[(kk + h) * 129]; av[1] = a_[(kk + h) * 129 + 32];
          bv[0] = b_[(kk + h) * 129]; bv[1] = b_[(kk + h) * 129 + 32];
        } else { av[0] = ra[0]; av[1] = ra[1]; bv[0] = rb[0]; bv[1] = rb[1]; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    const int l15 = lane & 15, q = lane >> 4;
    const float* as = smem + (wave / 2) * 64 + l15;
    const float* bs = smem + 2 * 16 * 129 + (wave % 2) * 64 + l15;
    v4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    float ra[4] = {rnd(lane), rnd(lane + 64), rnd(lane + 300), rnd(lane + 400)}, rb[4] = {rnd(lane + 128), rnd(lane + 192), rnd(lane + 500), rnd(lane + 600)};
    for (int it = 0; it < iters; ++it) {
      const float* a_ = as + (it & 1) * 16 * 129;
      const float* b_ = bs + (it & 1) * 16 * 129;
#pragma unroll
      for (int kk = 0; kk < 16; kk += 4) {
        float av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (MODE == 1) { av[i] = a_[(kk + q) * 129 + 16 * i]; bv[i] = b_[(kk + q) * 129 + 16 * i]; }
          else { av[i] = ra[i]; bv[i] = rb[i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}
template <int MODE>
void run(float* out, unsigned long long* clk, int bpc, const char* nm) {
  const int blocks = 256 * bpc, iters = 30000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[2 * 1024];
  (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * (blocks < 1024 ? blocks : 1024), hipMemcpyDeviceToHost);
  double mhz = 0; int n = blocks < 1024 ? blocks : 1024;
  for (int i = 0; i < n; ++i) mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
  const double fl = (double)blocks * 4 * iters * 8 * 4 * 4096.0;    // per wave and iteration: 64x64x16 MACs x 2
  printf("%-28s blocks/CU=%d: %7.2f ms %6.1f TFLOP/s  clock %4.0f MHz\n", nm, bpc, best, fl / best / 1e9, mhz / n);
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  unsigned long long* clk; (void)hipMalloc(&clk, 4096 * 2 * 8);
  for (int b : {1, 2, 4}) {
    run<2>(out, clk, b, "32x32x2 registers");
    run<3>(out, clk, b, "16x16x4 registers");
    run<0>(out, clk, b, "32x32x2 LDS operands");
    run<1>(out, clk, b, "16x16x4 LDS operands");
  }
  return 0;
}
