// Streaming bandwidth of MI355X by direction: write-only (plain / non-temporal 16-byte stores), read-only, copy.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE> __global__ __launch_bounds__(256) void k(f4* __restrict__ dst, const f4* __restrict__ src, long n, float* sink) {
  f4 acc = {0, 0, 0, 0};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (MODE == 0) dst[i] = f4{1.f, 2.f, 3.f, (float)i};
    else if (MODE == 1) __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, dst + i);
    else if (MODE == 2) acc += src[i];
    else if (MODE == 3) dst[i] = src[i];
    else __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
  }
  if (MODE == 2 && acc.x == 123.456f) sink[0] = acc.y;
}
template <int MODE> void run(const char* nm, f4* d, f4* s, long n, float* sink, int blocks, double bytes_per_elem) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, s, n, sink);
  (void)hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, s, n, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  printf("%-28s blocks=%5d: %7.3f ms  %6.2f TB/s\n", nm, blocks, best, n * bytes_per_elem / best / 1e9);
}
int main() {
  const long n = (1L << 30) / 16 * 2;   // 2 GiB per buffer
  f4 *d, *s; float* sink;
  (void)hipMalloc(&d, n * 16); (void)hipMalloc(&s, n * 16); (void)hipMalloc(&sink, 64);
  (void)hipMemset(s, 0, n * 16);
  for (int blocks : {2048, 8192, 65536}) {
    run<0>("write (plain 16 B)", d, s, n, sink, blocks, 16);
    run<1>("write (non-temporal 16 B)", d, s, n, sink, blocks, 16);
    run<2>("read", d, s, n, sink, blocks, 16);
    run<3>("copy (plain)", d, s, n, sink, blocks, 32);
    run<4>("copy (non-temporal)", d, s, n, sink, blocks, 32);
  }
  return 0;
}
