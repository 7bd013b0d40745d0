// Main-loop candidates for the fp32 MFMA GEMM (operands re-read from LDS every K step, no global traffic, no barrier):
//   0: 32x32x2, k-major LDS, ds_read_b32, next k pair prefetched (the production loop)
//   1: 16x16x4, k-major LDS (row stride 144: conflict-free), ds_read_b32, no explicit prefetch
//   2: 16x16x4, k-major LDS, ds_read_b32, next k quad prefetched
//   3: 16x16x4, m-major LDS [m][20], ds_read_b128 (4 k per lane per read: one read feeds 4 MFMA steps)
//   4: 32x32x2, m-major LDS [m][20], ds_read_b128 (lane half h takes k = 8j+4h+s)
// 64x64 output per wave in every variant.  Random data, ~50-100 ms per run, in-kernel clock reported.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v16 __attribute__((ext_vector_type(16)));
typedef float v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float rnd(unsigned i) {
  unsigned h = i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  return ((int)(h & 0xffffff) - 0x800000) * (1.0f / 0x400000);
}
constexpr int LDSF = 2 * 2 * 16 * 144;   // two buffers x (A, B) x 16 k x 144

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) float smem[LDSF];
  for (int i = threadIdx.x; i < LDSF; i += 256) smem[i] = rnd(i + blockIdx.x * 7919);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  const float* Abase = smem + (wave / 2) * 64;
  const float* Bbase = smem + 2 * 16 * 144 + (wave % 2) * 64;
  if (MODE == 0 || MODE == 4) {
    const int l31 = lane & 31, h = lane >> 5;
    v16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 0) {
        constexpr int S = 129;
        const float* a_ = Abase + (it & 1) * 16 * 144 + l31;
        const float* b_ = Bbase + (it & 1) * 16 * 144 + l31;
        float av[2][2], bv[2][2];
        auto rd = [&](int kk, int slot) {
          av[slot][0] = a_[(kk + h) * S]; av[slot][1] = a_[(kk + h) * S + 32];
          bv[slot][0] = b_[(kk + h) * S]; bv[slot][1] = b_[(kk + h) * S + 32];
        };
        rd(0, 0);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
          const int slot = (kk >> 1) & 1;
          if (kk + 2 < 16) rd(kk + 2, slot ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][i], bv[slot][j], acc[i][j], 0, 0, 0);
        }
      } else {
        // m-major: element (m, k) at m*20 + k; lane (l31, h) reads k = 8j + 4h .. +3 of its row
        const float* a_ = smem + (it & 1) * 2 * 16 * 144 + ((wave / 2) * 64 + l31) * 20 + 4 * h;
        const float* b_ = smem + (it & 1) * 2 * 16 * 144 + 128 * 20 + ((wave % 2) * 64 + l31) * 20 + 4 * h;
#pragma unroll
        for (int j8 = 0; j8 < 2; ++j8) {
          v4 a0 = *(const v4*)(a_ + 8 * j8), a1 = *(const v4*)(a_ + 8 * j8 + 32 * 20);
          v4 b0 = *(const v4*)(b_ + 8 * j8), b1 = *(const v4*)(b_ + 8 * j8 + 32 * 20);
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[st], b0[st], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[st], b1[st], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[st], b0[st], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[st], b1[st], acc[1][1], 0, 0, 0);
          }
        }
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  } else {
    const int l15 = lane & 15, q = lane >> 4;
    v4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1 || MODE == 2) {
        constexpr int S = 144;
        const float* a_ = Abase + (it & 1) * 16 * 144 + l15;
        const float* b_ = Bbase + (it & 1) * 16 * 144 + l15;
        float av[2][4], bv[2][4];
        auto rd = [&](int kk, int slot) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { av[slot][i] = a_[(kk + q) * S + 16 * i]; bv[slot][i] = b_[(kk + q) * S + 16 * i]; }
        };
        if (MODE == 2) rd(0, 0);
#pragma unroll
        for (int kk = 0; kk < 16; kk += 4) {
          const int slot = MODE == 2 ? (kk >> 2) & 1 : 0;
          if (MODE == 2) { if (kk + 4 < 16) rd(kk + 4, slot ^ 1); __builtin_amdgcn_sched_barrier(0); }
          else rd(kk, 0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][i], bv[slot][j], acc[i][j], 0, 0, 0);
        }
      } else {
        // m-major [m][20]: lane (l15, q) reads k = 4q .. 4q+3 of its row: one b128 read per 16-row tile feeds 4 MFMA steps
        const float* a_ = smem + (it & 1) * 2 * 16 * 144 + ((wave / 2) * 64 + l15) * 20 + 4 * q;
        const float* b_ = smem + (it & 1) * 2 * 16 * 144 + 128 * 20 + ((wave % 2) * 64 + l15) * 20 + 4 * q;
        v4 av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = *(const v4*)(a_ + 16 * 20 * i); bv[i] = *(const v4*)(b_ + 16 * 20 * i); }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i][st], bv[j][st], acc[i][j], 0, 0, 0);
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
  const int blocks = 256 * bpc, iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, 10);
  (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  unsigned long long h[2 * 1024];
  int n = blocks < 1024 ? blocks : 1024;
  (void)hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost);
  double mhz = 0;
  for (int i = 0; i < n; ++i) mhz += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
  const double fl = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 16;
  printf("%-44s blocks/CU=%d: %7.2f ms %6.1f TFLOP/s  clock %4.0f MHz\n", nm, bpc, best, fl / best / 1e9, mhz / n);
}
int main() {
  float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
  unsigned long long* clk; (void)hipMalloc(&clk, 4096 * 2 * 8);
  for (int b : {1, 2, 3, 4}) {
    run<0>(out, clk, b, "0: 32x32x2 k-major b32 prefetched (prod)");
    run<1>(out, clk, b, "1: 16x16x4 k-major b32");
    run<2>(out, clk, b, "2: 16x16x4 k-major b32 prefetched");
    run<3>(out, clk, b, "3: 16x16x4 m-major b128");
    run<4>(out, clk, b, "4: 32x32x2 m-major b128");
  }
  return 0;
}
