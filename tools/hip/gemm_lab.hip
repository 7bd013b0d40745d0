// GEMM laboratory (diagnostic build, not part of the library): compiles muscle_amd/csrc/gemm.hip with the stamp hook
// enabled and times / dissects the pointwise GEMMs on the B7 layer shapes without PyTorch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics tools/hip/gemm_lab.hip -o gpurun_out/gemm_lab
//   gemm_lab time            per-shape time / TFLOP/s of fwd, dgrad, wgrad (production dispatch)
//   gemm_lab planes [M K N]  first- against second-generation split kernel (time, results against each other and fp64)
//   gemm_lab stamps M K N    per-workgroup phase shares of the forward GEMM (prologue / main loop / epilogue) and clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>

#define MX_GEMM_STAMP(g, slot)                                                                         \
  do {                                                                                                 \
    if ((g).stamps && threadIdx.x == 0) {                                                              \
      const long wg__ = blockIdx.x + (long)gridDim.x * (blockIdx.y + (long)gridDim.y * blockIdx.z);    \
      (g).stamps[wg__ * 8 + (slot)] = __builtin_amdgcn_s_memtime();                                    \
      if ((slot) == 0) {                                                                               \
        (g).stamps[wg__ * 8 + 4] = __builtin_amdgcn_s_memrealtime();                                   \
        (g).stamps[wg__ * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4) |                         \
                                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32); \
      }                                                                                                \
      if ((slot) == 3) (g).stamps[wg__ * 8 + 5] = __builtin_amdgcn_s_memrealtime();                    \
    }                                                                                                  \
  } while (0)

#include "../../muscle_amd/csrc/gemm.hip"
#include "../../muscle_amd/csrc/wgrad.hip"
#include "../../muscle_amd/csrc/api.cpp"
#include "../../muscle_amd/csrc/bn.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_kernel(float* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((int)(h & 0xffffff) - 0x800000) * (scale / 0x800000);
  }
}
static float* dalloc(long n, unsigned seed, float scale) {
  float* p; CK(hipMalloc(&p, n * sizeof(float)));
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, p, n, seed, scale);
  return p;
}
template <class F> static float time_us(F f, int reps = 7) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f(); CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms * 1e3f);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

struct Shape { int M, K, N; };
static const Shape kShapes[] = {{25088, 640, 3840}, {25088, 3840, 640}, {25088, 384, 2304}, {25088, 2304, 384}, {25088, 224, 1344},
                                {25088, 1344, 224}, {25088, 160, 960}, {25088, 960, 160}, {100352, 80, 480}, {100352, 480, 80},
                                {401408, 48, 288}, {401408, 288, 48}, {401408, 192, 48}, {1605632, 32, 192}, {1605632, 32, 32}};

static void* ws = nullptr;
static const long ws_bytes = 512l << 20;

static void run_time() {
  if (!ws) CK(hipMalloc(&ws, ws_bytes));
  for (const Shape& s : kShapes) {
    float* A = dalloc((long)s.M * s.K, 1, 1.f); float* W = dalloc((long)s.N * s.K, 2, 0.05f);
    float* G = dalloc((long)s.M * s.N, 3, 1.f); float* C = dalloc((long)s.M * s.N, 4, 0.f);
    float* dX = dalloc((long)s.M * s.K, 5, 0.f); float* dW = dalloc((long)s.N * s.K, 6, 0.f);
    const int parts = mx_pw_fwd_parts(s.M, s.N, s.K);
    float* st = dalloc((long)parts * 2 * s.N, 7, 0.f);
    const double fl = 2.0 * s.M * s.K * s.N;
    float t1 = time_us([&] { mx_pw_fwd(A, 0, nullptr, nullptr, nullptr, 1, W, C, s.M, s.K, s.N, s.K, s.N, nullptr, nullptr, 0, st, nullptr); });
    // data gradient as the product runs it: a forward GEMM against W^T (ops.pw_dgrad)
    float* Wt = dalloc((long)s.N * s.K, 8, 0.05f);
    float t2 = time_us([&] { mx_pw_fwd(G, 0, nullptr, nullptr, nullptr, 1, Wt, dX, s.M, s.N, s.K, s.N, s.K, nullptr, nullptr, 0, nullptr, nullptr); });
    CK(hipFree(Wt));
    float t3 = time_us([&] {
      if (mx_pw_wgrad_tile_ws(s.M, s.N, s.K, 0) > 0) mx_pw_wgrad_tile(G, A, 0, nullptr, nullptr, nullptr, 1, dW, s.M, s.N, s.K, s.N, s.K, ws, ws_bytes, nullptr);
      else if (mx_pw_wgrad_small_ws(s.M, s.N, s.K, 0) > 0) mx_pw_wgrad_small(G, A, 0, nullptr, nullptr, nullptr, 1, dW, s.M, s.N, s.K, s.N, s.K, ws, ws_bytes, nullptr);
      else mx_pw_wgrad(G, A, 0, nullptr, nullptr, nullptr, 1, dW, s.M, s.N, s.K, s.N, s.K, ws, ws_bytes, nullptr);
    });
    printf("  M=%d K=%d N=%d: fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad %7.1f us %6.1f TF\n", s.M, s.K, s.N, t1,
           fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6);
    fflush(stdout);
    for (float* p : {A, W, G, C, dX, dW, st}) CK(hipFree(p));
  }
}

static void run_stamps(int M, int K, int N, int which) {
  float* A = dalloc((long)M * K, 1, 1.f); float* W = dalloc((long)N * K, 2, 0.05f);
  float* G = dalloc((long)M * N, 3, 1.f); float* C = dalloc((long)M * N, 4, 0.f);
  float* dX = dalloc((long)M * K, 5, 0.f); float* dW = dalloc((long)N * K, 6, 0.f);
  const int parts = mx_pw_fwd_parts(M, N, K);
  float* st = dalloc((long)parts * 2 * N, 7, 0.f);
  const long maxwg = 1 << 20;
  unsigned long long* stamps; CK(hipMalloc(&stamps, maxwg * 8 * sizeof(unsigned long long)));
  void* planes = nullptr;
  if (which == 3) {                       // the second-generation split kernel on the pre-split image of W
    const long pb = mx_pw_planes_bytes(N, K);
    if (pb <= 0) { printf("planes M=%d K=%d N=%d: no image for this shape\n", M, K, N); return; }
    CK(hipMalloc(&planes, pb));
    long row[5] = {(long)W, (long)planes, N, K, 0};
    long* table; CK(hipMalloc(&table, sizeof(row))); CK(hipMemcpy(table, row, sizeof(row), hipMemcpyHostToDevice));
    mx_pw_planes_batch(table, 1, mx_pw_planes_tiles(N, K), nullptr);
    CK(hipDeviceSynchronize());
  }
  auto call = [&] {
    if (which == 3) mx_pw_fwd_planes(A, planes, C, M, K, N, K, N, nullptr, nullptr, 0, st, nullptr);
    else if (which == 0) mx_pw_fwd(A, 0, nullptr, nullptr, nullptr, 1, W, C, M, K, N, K, N, nullptr, nullptr, 0, st, nullptr);
    else if (which == 1) mx_pw_dgrad(G, W, dX, M, N, K, N, K, nullptr, nullptr);
    else mx_pw_wgrad(G, A, 0, nullptr, nullptr, nullptr, 1, dW, M, N, K, N, K, ws, ws_bytes, nullptr);
  };
  if (!ws) CK(hipMalloc(&ws, ws_bytes));
  float t_plain = time_us(call);
  CK(hipMemset(stamps, 0, maxwg * 8 * sizeof(unsigned long long)));
  mx_gemm_stamps = stamps;
  float t_st = time_us(call, 3);
  // the clock the chip HOLDS under this kernel: ~1.5 s of back-to-back launches before the launch whose stamps are read
  for (int i = 0, n = (int)(1.5e6f / t_st) + 1; i < n; ++i) call();
  CK(hipDeviceSynchronize());
  CK(hipMemset(stamps, 0, maxwg * 8 * sizeof(unsigned long long)));
  call();
  CK(hipDeviceSynchronize());
  mx_gemm_stamps = nullptr;
  std::vector<unsigned long long> h(maxwg * 8);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  long nwg = 0;
  double pro = 0, loop = 0, epi = 0, clk = 0, epi_store = 0;
  unsigned long long t0 = ~0ull, t1 = 0;
  std::vector<double> starts, ends;
  for (long w = 0; w < maxwg; ++w) {
    const unsigned long long* s = &h[w * 8];
    if (!s[0] || !s[3]) continue;
    ++nwg;
    pro += (double)(s[1] - s[0]); loop += (double)(s[2] - s[1]); epi += (double)(s[3] - s[2]);
    if (s[7] > s[2]) epi_store += (double)(s[7] - s[2]);
    if (s[5] > s[4]) clk += (double)(s[3] - s[0]) / (double)(s[5] - s[4]) * 100.0;   // MHz (memrealtime = 100 MHz)
    t0 = std::min(t0, s[4]); t1 = std::max(t1, s[5]);
    starts.push_back((double)s[4]); ends.push_back((double)s[5]);
  }
  const char* nm[] = {"fwd", "dgrad", "wgrad", "fwd (planes kernel)"};
  printf("%s M=%d K=%d N=%d: %.1f us plain, %.1f us stamped; %ld workgroups stamped\n", nm[which], M, K, N, t_plain, t_st, nwg);
  if (nwg) {
    const double tot = pro + loop + epi;
    printf("  per-workgroup cycles: prologue %.0f (%.1f%%)  main loop %.0f (%.1f%%)  epilogue %.0f (%.1f%%)  total %.0f; in-kernel clock %.0f MHz\n",
           pro / nwg, 100 * pro / tot, loop / nwg, 100 * loop / tot, epi / nwg, 100 * epi / tot, tot / nwg, clk / nwg);
    if (epi_store > 0) printf("  epilogue: %.0f cycles up to the last C store issued, %.0f for the statistics after it\n", epi_store / nwg, (epi - epi_store) / nwg);
    printf("  first start -> last end (realtime): %.1f us\n", (double)(t1 - t0) / 100.0);
    // concurrency histogram: how many workgroups are alive over time (20 buckets)
    const int NB = 20;
    for (int b = 0; b < NB; ++b) {
      const double t = (double)t0 + ((double)(t1 - t0)) * (b + 0.5) / NB;
      long alive = 0;
      for (size_t i = 0; i < starts.size(); ++i) alive += (starts[i] <= t && ends[i] >= t);
      printf("%s%ld", b ? " " : "  alive: ", alive);
    }
    printf("\n");
    if (which == 3) {
      // the workgroups of ONE compute unit (XCC 0, the CU / SE bits of the first stamped workgroup), in start order: are their phases locked?
      struct Ev { double s, l, e; long w; };
      std::vector<Ev> ev;
      unsigned long long key0 = ~0ull;
      for (long w = 0; w < maxwg; ++w) {
        const unsigned long long* s = &h[w * 8];
        if (!s[0] || !s[3]) continue;
        const unsigned long long key = ((s[6] >> 8) & 0xff) | (((s[6] >> 32) & 0xf) << 8);
        if (key0 == ~0ull) key0 = key;
        if (key != key0) continue;
        const double cyc2us = (double)(s[5] - s[4]) / 100.0 / (double)(s[3] - s[0]);
        ev.push_back({(double)(s[4] - t0) / 100.0, (double)(s[4] - t0) / 100.0 + (double)(s[2] - s[0]) * cyc2us, (double)(s[5] - t0) / 100.0, w});
      }
      std::sort(ev.begin(), ev.end(), [](const Ev& a, const Ev& b) { return a.s < b.s; });
      printf("  one CU (key %llx), %zu workgroups: [id start -> loop end -> end] us:", key0, ev.size());
      for (const Ev& e : ev) printf(" [%ld %.0f>%.0f>%.0f]", e.w, e.s, e.l, e.e);
      printf("\n");
    }
    // start-time rounds: sort starts, print deciles in us
    std::sort(starts.begin(), starts.end());
    printf("  start deciles (us):");
    for (int d = 0; d <= 10; ++d) printf(" %.0f", (starts[std::min(starts.size() - 1, starts.size() * d / 10)] - (double)t0) / 100.0);
    printf("\n");
  }
}

// forward GEMM (bias + residual + statistics) against a plain fp64 host reference on sampled rows, and the two kernel
// generations against each other (MX_GEMM_NT_V2 is read once per process: the second generation is checked by default)
static int run_check(int M, int K, int N) {
  float* A = dalloc((long)M * K, 11, 1.f); float* W = dalloc((long)N * K, 12, 0.05f);
  float* R = dalloc((long)M * N, 13, 1.f); float* bias = dalloc(N, 14, 1.f);
  float* C1 = dalloc((long)M * N, 4, 0.f);
  const int parts = mx_pw_fwd_parts(M, N, K);
  float* st1 = dalloc((long)parts * 2 * N, 7, 0.f);
  if (mx_pw_fwd(A, 0, nullptr, nullptr, nullptr, 1, W, C1, M, K, N, K, N, bias, R, 0, st1, nullptr)) { printf("fwd failed: %s\n", mx_last_error()); return 1; }
  CK(hipDeviceSynchronize());
  std::vector<float> hA((long)M * K), hW((long)N * K), hR((long)M * N), hb(N), h1((long)M * N), s1((long)parts * 2 * N);
  CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bias, hb.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s1.data(), st1, s1.size() * 4, hipMemcpyDeviceToHost));
  double worst_ref = 0;
  for (int t = 0; t < 400; ++t) {
    const long r = t < 8 ? (t < 4 ? t : M - 1 - (t - 4)) : ((long)t * 7919 + 13) % M;
    for (int c = 0; c < N; c += (t < 8 ? 1 : 37)) {
      double acc = (double)hb[c] + hR[r * N + c];
      for (int k = 0; k < K; ++k) acc += (double)hA[r * K + k] * hW[(long)c * K + k];
      worst_ref = std::max(worst_ref, fabs(acc - h1[r * N + c]));
    }
  }
  // statistics: column sums over all partial rows against sums of the output itself
  double worst_st = 0, scale_st = 0;
  for (int c = 0; c < N; c += 5) {
    double a = 0, a2 = 0, b = 0, b2 = 0;
    for (int p = 0; p < parts; ++p) { a += s1[((long)p * 2 + 0) * N + c]; a2 += s1[((long)p * 2 + 1) * N + c]; }
    for (long r = 0; r < M; ++r) { const double v = h1[r * N + c]; b += v; b2 += v * v; }
    worst_st = std::max(worst_st, std::max(fabs(a - b) / (fabs(b) + 1.0), fabs(a2 - b2) / (fabs(b2) + 1.0)));
    scale_st = std::max(scale_st, fabs(b2));
  }
  printf("check M=%d K=%d N=%d: vs fp64 reference max|d| %.3g, statistics rel err %.3g\n", M, K, N, worst_ref, worst_st);
  for (float* p : {A, W, R, bias, C1, st1}) CK(hipFree(p));
  return (worst_ref < 1e-3 && worst_st < 1e-4) ? 0 : 2;
}

// first-generation split kernel (mx_pw_fwd) against the second generation (mx_pw_fwd_planes: pre-split weight planes by
// LDS-DMA, activations straight to registers, 16x16x32 MFMA) on one shape: interleaved timing in one process, results against
// each other and against fp64
static int run_planes(int M, int K, int N, bool check) {
  float* A = dalloc((long)M * K, 1, 1.f); float* W = dalloc((long)N * K, 2, 0.05f);
  float* C1 = dalloc((long)M * N, 4, 0.f); float* C2 = dalloc((long)M * N, 5, 0.f);
  float* R = dalloc((long)M * N, 13, 1.f); float* bias = dalloc(N, 14, 1.f);
  const int parts = mx_pw_fwd_parts(M, N, K);
  float* st1 = dalloc((long)parts * 2 * N, 7, 0.f); float* st2 = dalloc((long)parts * 2 * N, 8, 0.f);
  const long pb = mx_pw_planes_bytes(N, K);
  if (pb <= 0) { printf("  M=%d K=%d N=%d: no planes for this shape\n", M, K, N); return 0; }
  void* planes; CK(hipMalloc(&planes, pb));
  long row[5] = {(long)W, (long)planes, N, K, 0};
  long* table; CK(hipMalloc(&table, sizeof(row))); CK(hipMemcpy(table, row, sizeof(row), hipMemcpyHostToDevice));
  const int tiles = mx_pw_planes_tiles(N, K);
  float tp = time_us([&] { mx_pw_planes_batch(table, 1, tiles, nullptr); });
  const double fl = 2.0 * M * K * N;
  // variants: 0 gen1 | 1 gen3 128 wide | 2 gen3 64 wide | 3 gen3, the library's own choice
  float best[4] = {1e30f, 1e30f, 1e30f, 1e30f};
  bool extras = false;                 // timing: statistics only (the expand forward); check: bias + residual as well
  auto run_v = [&](int v, float* Cout, float* stout) {
    const float* b_ = extras ? bias : nullptr; const float* r_ = extras ? R : nullptr;
    if (v == 0) { mx_pw_fwd(A, 0, nullptr, nullptr, nullptr, 1, W, Cout, M, K, N, K, N, b_, r_, 0, stout, nullptr); return; }
    g_split_nj = v == 1 ? 2 : v == 2 ? 1 : 0;
    mx_pw_fwd_planes(A, planes, Cout, M, K, N, K, N, b_, r_, 0, stout, nullptr);
    g_split_nj = 0;
  };
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 4; ++v) best[v] = std::min(best[v], time_us([&] { run_v(v, v ? C2 : C1, v ? st2 : st1); }, 5));
  printf("  M=%d K=%d N=%d: gen1 %7.1f us %5.1f TF | gen3 128w %7.1f %5.1f | gen3 64w %7.1f %5.1f | gen3 auto %7.1f %5.1f | planes %4.1f us\n", M, K, N,
         best[0], fl / best[0] / 1e6, best[1], fl / best[1] / 1e6, best[2], fl / best[2] / 1e6, best[3], fl / best[3] / 1e6, tp);
  fflush(stdout);
  int rc = 0;
  if (check) {
    extras = true;
    for (int v = 1; v < 3; ++v) {
      CK(hipMemset(C2, 0xff, (size_t)M * N * 4));
      CK(hipMemset(st2, 0xff, (size_t)parts * 2 * N * 4));
      run_v(0, C1, st1);
      run_v(v, C2, st2);
      CK(hipDeviceSynchronize());
      std::vector<float> hA((long)M * K), hW((long)N * K), hR((long)M * N), hb(N), h1((long)M * N), h2((long)M * N), s1((long)parts * 2 * N), s2((long)parts * 2 * N);
      CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hR.data(), R, hR.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), bias, hb.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), C2, h2.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(s1.data(), st1, s1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s2.data(), st2, s2.size() * 4, hipMemcpyDeviceToHost));
      double d12 = 0, e1 = 0, e2 = 0, ds = 0;
      for (long i = 0; i < (long)M * N; ++i) { const double d = fabs((double)h1[i] - h2[i]); if (!(d <= d12)) d12 = d; }   // (NaN-proof maximum)
      for (long i = 0; i < (long)parts * 2 * N; ++i) { const double d = fabs((double)s1[i] - s2[i]) / (fabs((double)s1[i]) + 1.0); if (!(d <= ds)) ds = d; }
      for (int t = 0; t < 200; ++t) {
        const long r = t < 8 ? (t < 4 ? t : M - 1 - (t - 4)) : ((long)t * 7919 + 13) % M;
        for (int c = 0; c < N; c += (t < 8 ? 1 : 37)) {
          double acc = (double)hb[c] + hR[r * N + c];
          for (int k = 0; k < K; ++k) acc += (double)hA[r * K + k] * hW[(long)c * K + k];
          e1 = std::max(e1, fabs(acc - h1[r * N + c])); e2 = std::max(e2, fabs(acc - h2[r * N + c]));
        }
      }
      const bool ok = d12 < 1e-4 && e2 < 1.5 * e1 + 1e-6 && ds < 5e-5;
      printf("    check %s: vs gen1 max|d| %.3g; vs fp64: gen1 %.3g gen3 %.3g; statistics rel %.3g %s\n", v == 1 ? "128w" : " 64w", d12, e1, e2, ds, ok ? "" : "FAILED");
      if (!ok) rc = 2;
    }
  }
  for (float* p : {A, W, C1, C2, R, bias, st1, st2}) CK(hipFree(p));
  CK(hipFree(planes)); CK(hipFree(table));
  return rc;
}

// second-generation split kernel: N tiles per XCD-local chunk (MX_SPLIT3_XCD_CHUNK)
static void run_chunk(int M, int K, int N) {
  float* A = dalloc((long)M * K, 1, 1.f); float* W = dalloc((long)N * K, 2, 0.05f);
  float* C2 = dalloc((long)M * N, 5, 0.f);
  float* st2 = dalloc((long)mx_pw_fwd_parts(M, N, K) * 2 * N, 8, 0.f);
  void* planes; CK(hipMalloc(&planes, mx_pw_planes_bytes(N, K)));
  long row[5] = {(long)W, (long)planes, N, K, 0};
  long* table; CK(hipMalloc(&table, sizeof(row))); CK(hipMemcpy(table, row, sizeof(row), hipMemcpyHostToDevice));
  mx_pw_planes_batch(table, 1, mx_pw_planes_tiles(N, K), nullptr);
  const int cv[] = {1000, 16, 12, 10, 8, 6, 4};      // (lists of up to 16 N tiles are never cut: only N > 2048 reacts)
  float best[7]; for (float& b : best) b = 1e30f;
  for (int round = 0; round < 3; ++round)
    for (int i = 0; i < 7; ++i) {
      g_split3_xcd_chunk = cv[i];
      best[i] = std::min(best[i], time_us([&] { mx_pw_fwd_planes(A, planes, C2, M, K, N, K, N, nullptr, nullptr, 0, st2, nullptr); }, 5));
    }
  g_split3_xcd_chunk = 6;
  printf("  M=%d K=%d N=%d:", M, K, N);
  for (int i = 0; i < 7; ++i) printf("  chunk %-4d %6.1f", cv[i], best[i]);
  printf("\n"); fflush(stdout);
  for (float* p : {A, W, C2, st2}) CK(hipFree(p));
  CK(hipFree(planes)); CK(hipFree(table));
}

// second-generation split kernel: every tile width on one shape (forced through MX_GEMM_SPLIT_NJ's lab values)
static void run_widths(int M, int K, int N) {
  float* A = dalloc((long)M * K, 1, 1.f); float* W = dalloc((long)N * K, 2, 0.05f);
  float* C2 = dalloc((long)M * N, 5, 0.f);
  float* st2 = dalloc((long)mx_pw_fwd_parts(M, N, K) * 2 * N, 8, 0.f);
  void* planes; CK(hipMalloc(&planes, mx_pw_planes_bytes(N, K) + 8192));      // (slack: a padded last tile reads up to 16 rows past the image)
  long row[5] = {(long)W, (long)planes, N, K, 0};
  long* table; CK(hipMalloc(&table, sizeof(row))); CK(hipMemcpy(table, row, sizeof(row), hipMemcpyHostToDevice));
  mx_pw_planes_batch(table, 1, mx_pw_planes_tiles(N, K), nullptr);
  const int wv[] = {64, 80, 96, 112, 128, 0};
  float best[6]; for (float& b : best) b = 1e30f;
  for (int round = 0; round < 3; ++round)
    for (int i = 0; i < 6; ++i) {
      // odd widths: where they divide N, or pad it by at most 16 columns that stay inside the image's 128-row padding + slack
      if (wv[i] && wv[i] != 64 && wv[i] != 128 && N % wv[i] && ((N + wv[i] - 1) / wv[i] * wv[i] - ((N + 127) & ~127) > 16 || (N + wv[i] - 1) / wv[i] * wv[i] - N > N / 16)) continue;
      g_split_nj = wv[i];
      best[i] = std::min(best[i], time_us([&] { mx_pw_fwd_planes(A, planes, C2, M, K, N, K, N, nullptr, nullptr, 0, st2, nullptr); }, 5));
    }
  g_split_nj = 0;
  printf("  M=%d K=%d N=%d:", M, K, N);
  for (int i = 0; i < 6; ++i) if (best[i] < 1e29f) printf("  w%-3d %6.1f", wv[i], best[i]);
  printf("   (w0 = the library's choice)\n"); fflush(stdout);
  for (float* p : {A, W, C2, st2}) CK(hipFree(p));
  CK(hipFree(planes)); CK(hipFree(table));
}

// wgradf32 R Co Ci: exact-fp32 weight gradient, the tiled kernel against wgrad_f32_ws_kernel (gemm mode 0), same process: time, max difference,
// both against fp64 on a sample of output elements
static int run_wgradf32(int R, int Co, int Ci) {
  if (!ws) CK(hipMalloc(&ws, ws_bytes));
  mx_set_gemm_mode(0);
  float* G = dalloc((long)R * Co, 11, 1.f); float* X = dalloc((long)R * Ci, 12, 1.f);
  float* d0 = dalloc((long)Co * Ci, 13, 0.f); float* d1 = dalloc((long)Co * Ci, 13, 0.f);
  float t[2] = {1e30f, 1e30f};
  for (int round = 0; round < 3; ++round)
    for (int v = 0; v < 2; ++v) {
      g_wgrad_f32ws = v;
      if (mx_pw_wgrad_tile_ws(R, Co, Ci, 0) > ws_bytes) { printf("  workspace too small\n"); return 1; }
      t[v] = std::min(t[v], time_us([&] { mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, v ? d1 : d0, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
    }
  CK(hipMemset(d0, 0, (size_t)Co * Ci * 4)); CK(hipMemset(d1, 0, (size_t)Co * Ci * 4));
  g_wgrad_f32ws = 0; mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d0, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr);
  g_wgrad_f32ws = 1; mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d1, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr);
  CK(hipDeviceSynchronize());
  std::vector<float> h0((size_t)Co * Ci), h1((size_t)Co * Ci), hG((size_t)R * Co), hX((size_t)R * Ci);
  CK(hipMemcpy(h0.data(), d0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), d1, h1.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hG.data(), G, hG.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hX.data(), X, hX.size() * 4, hipMemcpyDeviceToHost));
  double dmax = 0, e0 = 0, e1 = 0, vmax = 0;
  for (size_t i = 0; i < h0.size(); ++i) { dmax = std::max(dmax, (double)fabs(h0[i] - h1[i])); vmax = std::max(vmax, (double)fabs(h0[i])); }
  for (int k = 0; k < 64; ++k) {
    const int co = (int)((k * 7919L + 3) % Co), ci = (int)((k * 104729L + 5) % Ci);
    double acc = 0;
    for (long r = 0; r < R; ++r) acc += (double)hG[r * Co + co] * hX[r * Ci + ci];
    e0 = std::max(e0, fabs(acc - h0[(size_t)co * Ci + ci])); e1 = std::max(e1, fabs(acc - h1[(size_t)co * Ci + ci]));
  }
  const double fl = 2.0 * R * Co * Ci;
  printf("  R=%d Co=%d Ci=%d: tiled %7.1f us %6.1f TF | ws %7.1f us %6.1f TF | max|d| = %g (max|v| = %g); vs fp64 on 64 elements: tiled %.3g ws %.3g%s\n", R, Co, Ci,
         t[0], fl / t[0] * 1e-6, t[1], fl / t[1] * 1e-6, dmax, vmax, e0, e1, (dmax > 2e-3 * vmax || e1 > 2.0 * e0 + 1e-4 * vmax) ? "  FAILED" : "");
  fflush(stdout);
  for (float* p : {G, X, d0, d1}) CK(hipFree(p));
  g_wgrad_f32ws = 1;
  return 0;
}

static int wlab_mode() { return getenv("WLAB_MODE") ? atoi(getenv("WLAB_MODE")) : 2; }
// wgrad [R Co Ci]: the first split weight-gradient kernel against the round-5 ones (WLAB_MODE = 1 single-stream pipeline, 2 wave-specialised), same process: time, bits
static int run_wgrad(int R, int Co, int Ci) {
  if (!ws) CK(hipMalloc(&ws, ws_bytes));
  float* G = dalloc((long)R * Co, 11, 1.f); float* X = dalloc((long)R * Ci, 12, 1.f);
  float* d0 = dalloc((long)Co * Ci, 13, 0.f); float* d1 = dalloc((long)Co * Ci, 13, 0.f);
  if (mx_pw_wgrad_tile_ws(R, Co, Ci, 0) <= 0) { printf("  R=%d Co=%d Ci=%d: not a tiled shape\n", R, Co, Ci); return 0; }
  const double fl = 2.0 * R * Co * Ci;
  float t[2];
  for (int v = 0; v < 2; ++v) {
    mx_wgrad_pipe_override = v ? wlab_mode() : 0;
    float* d = v ? d1 : d0;
    CK(hipMemset(d, 0, (long)Co * Ci * 4));
    if (mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr)) { printf("error: %s\n", mx_last_error()); return 1; }
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
      best = std::min(best, time_us([&] { mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
    t[v] = best;
  }
  mx_wgrad_pipe_override = -1;
  // bits: one fresh call each into zeroed outputs
  std::vector<float> h0((long)Co * Ci), h1((long)Co * Ci);
  for (int v = 0; v < 2; ++v) {
    mx_wgrad_pipe_override = v ? wlab_mode() : 0;
    float* d = v ? d1 : d0;
    CK(hipMemset(d, 0, (long)Co * Ci * 4));
    mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr);
    CK(hipMemcpy((v ? h1 : h0).data(), d, (long)Co * Ci * 4, hipMemcpyDeviceToHost));
  }
  mx_wgrad_pipe_override = -1;
#ifdef WPIPE_STAMPS
  {
    const long nwg = 1 << 16;
    unsigned long long* st; CK(hipMalloc(&st, nwg * 8 * sizeof(unsigned long long))); CK(hipMemset(st, 0, nwg * 8 * sizeof(unsigned long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wpipe_stamps), &st, sizeof(st)));
    mx_wgrad_pipe_override = wlab_mode();
    for (int i = 0; i < 200; ++i) mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d1, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr);     // the clock the chip HOLDS
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), st, nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> loop, tot, ghz, pro;
    for (long w = 0; w < nwg; ++w) {
      if (!h[w * 8 + 3]) continue;
      loop.push_back((double)(h[w * 8 + 2] - h[w * 8 + 1])); tot.push_back((double)(h[w * 8 + 3] - h[w * 8 + 0])); pro.push_back((double)(h[w * 8 + 1] - h[w * 8 + 0]));
      ghz.push_back((double)(h[w * 8 + 3] - h[w * 8 + 0]) / ((double)(h[w * 8 + 5] - h[w * 8 + 4]) * 10.0));
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    WtPlan pl; wt_plan(R, Co, Ci, 0, &pl);
    const int ns = (pl.rows_per_group + 31) / 32;
    printf("    stamps: %zu workgroups, %d slabs each: loop %.0f cycles = %.1f per slab (%.1f per MFMA), prologue %.0f, whole %.0f, clock %.2f GHz\n", loop.size(), ns,
           med(loop), med(loop) / ns, med(loop) / ns / 48, med(pro), med(tot), med(ghz));
    unsigned long long* z = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(wpipe_stamps), &z, sizeof(z)));
    mx_wgrad_pipe_override = -1;
    CK(hipFree(st));
  }
#endif
  double md = 0, mx = 0;
  for (long i = 0; i < (long)Co * Ci; ++i) { md = std::max(md, (double)fabsf(h0[i] - h1[i])); mx = std::max(mx, (double)fabsf(h0[i])); }
  printf("  R=%d Co=%d Ci=%d: v1 %7.1f us %6.1f TF | pipe %7.1f us %6.1f TF | max|d| = %g (max|v1| = %g)\n", R, Co, Ci, t[0], fl / t[0] / 1e6, t[1],
         fl / t[1] / 1e6, md, mx);
  fflush(stdout);
  for (float* p : {G, X, d0, d1}) CK(hipFree(p));
  return md == 0 ? 0 : 2;
}

// wgradf [R Co Ci]: the weight-gradient-side BatchNorm-backward fold (wgrad_split_ws_kernel<true>, dZ stored) against the plain kernel + the apply pass it replaces
static int run_wgradf(int R, int Co, int Ci) {
  if (!ws) CK(hipMalloc(&ws, ws_bytes));
  float* G = dalloc((long)R * Co, 11, 1.f); float* G2 = dalloc((long)R * Co, 14, 1.f); float* X = dalloc((long)R * Ci, 12, 1.f);
  float* coef = dalloc(3l * Co, 15, 1.f); float* dz = dalloc((long)R * Co, 16, 0.f);
  float* d0 = dalloc((long)Co * Ci, 13, 0.f);
  if (!mx_pw_wgrad_tile_bnbwd_dz_ok(R, Co, Ci, Co, Ci)) { printf("  R=%d Co=%d Ci=%d: not taken\n", R, Co, Ci); return 0; }
  const double fl = 2.0 * R * Co * Ci;
  float t0 = 1e30f, t1 = 1e30f, t2 = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    t0 = std::min(t0, time_us([&] { mx_pw_wgrad_tile(G, X, 0, nullptr, nullptr, nullptr, 1, d0, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
    t1 = std::min(t1, time_us([&] { mx_pw_wgrad_tile_bnbwd_dz(G, G2, coef, X, d0, dz, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
    t2 = std::min(t2, time_us([&] { mx_bn_bwd_apply(G, G2, nullptr, nullptr, nullptr, nullptr, nullptr, coef, coef + Co, coef + 2 * Co, dz, R, Co, 1, nullptr); }, 5));
  }
  printf("  R=%d Co=%d Ci=%d: plain %7.1f us %6.1f TF | folded + dZ %7.1f us | bn_bwd_apply alone %6.1f us -> %+.1f us\n", R, Co, Ci, t0, fl / t0 / 1e6, t1, t2, t1 - t0 - t2);
  {   // what of the fold costs: without the dZ stores; with G2 = G (the second tensor's requests hit the lines just fetched)
    float t3 = 1e30f, t4 = 1e30f, t5 = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      t3 = std::min(t3, time_us([&] { mx_pw_wgrad_tile_bnbwd(G, G2, coef, X, d0, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
      t4 = std::min(t4, time_us([&] { mx_pw_wgrad_tile_bnbwd(G, G, coef, X, d0, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
      t5 = std::min(t5, time_us([&] { mx_pw_wgrad_tile_bnbwd_dz(G, G, coef, X, d0, dz, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr); }, 5));
    }
    printf("    folded without dZ stores %7.1f us | with G2 = G, no stores %7.1f us | with G2 = G and stores %7.1f us\n", t3, t4, t5);
  }
#ifdef WPIPE_STAMPS
  {
    const long nwg = 1 << 16;
    unsigned long long* st; CK(hipMalloc(&st, nwg * 8 * sizeof(unsigned long long))); CK(hipMemset(st, 0, nwg * 8 * sizeof(unsigned long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(wpipe_stamps), &st, sizeof(st)));
    for (int i = 0; i < 100; ++i) mx_pw_wgrad_tile_bnbwd_dz(G, G2, coef, X, d0, dz, R, Co, Ci, Co, Ci, ws, ws_bytes, nullptr);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nwg * 8);
    CK(hipMemcpy(h.data(), st, nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> loop, ghz;
    for (long w = 0; w < nwg; ++w) {
      if (!h[w * 8 + 3]) continue;
      loop.push_back((double)(h[w * 8 + 2] - h[w * 8 + 1]));
      ghz.push_back((double)(h[w * 8 + 3] - h[w * 8 + 0]) / ((double)(h[w * 8 + 5] - h[w * 8 + 4]) * 10.0));
    }
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    WtPlan pl; wt_plan(R, Co, Ci, 0, &pl);
    const int ns = (pl.rows_per_group + 31) / 32;
    printf("    stamps (folded): %zu workgroups, %d slabs: %.1f cycles per slab (%.1f per MFMA), clock %.2f GHz\n", loop.size(), ns, med(loop) / ns, med(loop) / ns / 48, med(ghz));
    unsigned long long* z = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(wpipe_stamps), &z, sizeof(z)));
    CK(hipFree(st));
  }
#endif
  fflush(stdout);
  for (float* p : {G, G2, X, coef, dz, d0}) CK(hipFree(p));
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 5 && !strcmp(argv[1], "wgradf32")) return run_wgradf32(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  if (argc >= 2 && !strcmp(argv[1], "wgradf")) {
    if (argc >= 5) return run_wgradf(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    static const Shape sh[] = {{25088, 2304, 384}, {25088, 3840, 640}, {25088, 1344, 224}, {25088, 960, 160}};
    for (const Shape& s : sh) run_wgradf(s.M, s.K, s.N);
    return 0;
  }

  if (argc >= 2 && !strcmp(argv[1], "wgrad")) {
    if (argc >= 5) return run_wgrad(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    static const Shape sh[] = {{25088, 2304, 384}, {25088, 384, 2304}, {25088, 3840, 640}, {25088, 640, 3840}, {25088, 1344, 224}, {25088, 224, 1344},
                               {25088, 960, 160}, {25088, 160, 960}, {6272, 2304, 384}, {12544, 1152, 192}, {25088 + 19, 640, 2304}, {100352, 480, 80}};
    int rc = 0;
    for (const Shape& s : sh) rc |= run_wgrad(s.M, s.K, s.N);
    return rc;
  }

  if (argc >= 2 && !strcmp(argv[1], "widths")) {
    if (argc >= 5) { run_widths(atoi(argv[2]), atoi(argv[3]), atoi(argv[4])); return 0; }
    static const Shape sh[] = {{25088, 160, 960}, {25088, 960, 160}, {25088, 224, 1344}, {25088, 1344, 224}, {25088, 384, 2304}, {25088, 2304, 384},
                               {25088, 640, 3840}, {25088, 3840, 640}, {6272, 384, 2304}, {6272, 2304, 384}, {6272, 640, 3840}, {6272, 3840, 640},
                               {6272, 224, 1344}, {6272, 1344, 224}, {401408, 288, 48}, {100352, 480, 80}, {12544, 192, 1152}, {12544, 1152, 192}};
    for (const Shape& s : sh) run_widths(s.M, s.K, s.N);
    return 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "chunk")) {
    static const Shape sh[] = {{25088, 384, 2304}, {25088, 640, 3840}, {25088, 224, 1344}, {25088, 160, 960}, {25088, 2304, 384}, {25088, 3840, 640}, {6272, 384, 2304}};
    for (const Shape& s : sh) run_chunk(s.M, s.K, s.N);
    return 0;
  }
  if (argc >= 2 && !strcmp(argv[1], "planes")) {
    int rc = 0;
    if (argc >= 5) return run_planes(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), true);
    static const Shape sh[] = {{25088, 384, 2304}, {25088, 2304, 384}, {25088, 640, 3840}, {25088, 3840, 640}, {25088, 224, 1344},
                               {25088, 1344, 224}, {25088, 160, 960}, {25088, 960, 160}, {6272, 384, 2304}, {6272, 2304, 384},
                               {12544, 192, 1152}, {12544, 1152, 192}, {25000, 160, 960}, {1000, 320, 200}};
    for (const Shape& s : sh) rc |= run_planes(s.M, s.K, s.N, s.M <= 25000 || s.K == 384);
    return rc;
  }
  if (argc >= 5 && !strcmp(argv[1], "check")) return run_check(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
  if (argc >= 2 && !strcmp(argv[1], "time")) { run_time(); return 0; }
  if (argc >= 5 && !strcmp(argv[1], "time1")) {      // forward GEMM of one shape (plain A, statistics)
    const int M = atoi(argv[2]), K = atoi(argv[3]), N = atoi(argv[4]);
    float* A = dalloc((long)M * K, 1, 1.f); float* W = dalloc((long)(N + 256) * (K + 32) * 2, 2, 0.05f); float* C = dalloc((long)M * N, 4, 0.f);
    float* st = dalloc((long)mx_pw_fwd_parts(M, N, K) * 2 * N, 7, 0.f);
    float t1 = time_us([&] { mx_pw_fwd(A, 0, nullptr, nullptr, nullptr, 1, W, C, M, K, N, K, N, nullptr, nullptr, 0, st, nullptr); });
    printf("  M=%d K=%d N=%d: fwd %7.1f us %6.1f TF\n", M, K, N, t1, 2.0 * M * K * N / t1 / 1e6);
    return 0;
  }
  if (argc >= 5 && !strcmp(argv[1], "stamps")) {
    for (int which = (argc >= 6 ? atoi(argv[5]) : 0); which < (argc >= 6 ? atoi(argv[5]) + 1 : 3); ++which) run_stamps(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), which);
    return 0;
  }
  printf("usage: gemm_lab time | stamps M K N\n");
  return 1;
}
