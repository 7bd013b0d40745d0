// Exact-fp32 pointwise-convolution GEMMs on the CDNA4 matrix cores
// (v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain, 64 cycles / SIMD).
//
// One kernel template covers the three contractions the 1x1 convolutions of an MBConv block need
// (reference: src/efficientnet_pytorch/model.py:77,86 forward; autograd's conv backward):
//   NT : C[M,N]  = A'[M,K] * W[N,K]^T          forward   (W = conv weight [Cout,Cin])
//   NN : C[M,N]  = A'[M,K] * W[K,N]            dgrad     (W = conv weight [Cout,Cin], K=Cout, N=Cin)
//   TN : C[M,N] += A'[R,M]^T * B'[R,N]         wgrad     (reduction over the R pixel rows, split over blocks)
// A' / B' are activation matrices [rows, channels] (NHWC) read through an MxOperand prologue, so the
// BN affine + SiLU + SE gate of the *producer* is applied on the consumer's load and never
// round-trips HBM.  Epilogue: optional bias / residual / relu, and per-column sum and sum of
// squares (train-mode BatchNorm statistics of the conv output) accumulated in fp64.
//
// Tiling: 256 threads = 4 waves; each wave owns TM x TN MFMA tiles of 32x32; both operand tiles are
// staged k-major in LDS (As[BK][BM+pad]) so the per-lane MFMA operand read is a conflict-free
// ds_read_b32 of consecutive m (the 64-cycle fp32 MFMA leaves the LDS ~10 % busy, so wide reads buy
// nothing here).  Global->register->LDS staging is double buffered with one barrier per K step.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  MxOperand a, b;
  float* c;
  int M, N, K;
  int lda, ldb, ldc;
  long sa, sb, sc;          // batch strides (elements)
  const float* bias;        // [N] or null
  const float* residual;    // [M, ldc] or null
  int relu;
  float* stats;             // partial statistics [Mtiles][2][N] or null
  int ksplit;               // TN: rows of R per z-slice
  int xcd_nt;               // > 0: 1-D grid, the xcd_nt N tiles of an M tile run back to back on one XCD (see kernel)
  int mt, zt;               // M tiles, reduction slices (for the 1-D grids)
  unsigned long long* stamps;   // diagnostic builds only (tools/hip/gemm_lab.hip); null in the library
  float* a_out;             // MX_BNBWD: the prologue's result [M, lda] is also written here (by the N tile 0 workgroups)
  float* ws;                // TN: scratch for the per-slice partial matrices [slices][M][N]
};

// Diagnostic hook: tools/hip/gemm_lab.hip compiles this file with MX_GEMM_STAMP defined to record s_memtime stamps per
// workgroup (prologue / main loop / epilogue shares, in-kernel clock).  In the library build it expands to nothing.
#ifndef MX_GEMM_STAMP
#define MX_GEMM_STAMP(g, slot)
#endif
static unsigned long long* mx_gemm_stamps = nullptr;

enum { L_NT = 0, L_NN = 1, L_TN = 2 };

// load a [rows x BK] slab of a row-major [R, ld] matrix (k contiguous) and store it k-major
template <int ROWS, int PAD, int BK>
struct SlabK {   // matrix is [rows_total, K] with k contiguous -> transposing store
  static constexpr int TOTAL = ROWS * (BK / 4);
  static constexpr int PER = (TOTAL + 255) / 256;
  float4 v[PER];
  __device__ __forceinline__ void load(const MxOperand& o, const float* base, int ld, int r0, int rows_total,
                                       int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + 256 * i;
      int k4 = idx % (BK / 4), row = idx / (BK / 4);
      int r = r0 + row, k = k0 + 4 * k4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < TOTAL && r < rows_total && k < K) {
        x = ld4(base + (long)r * ld + k);
        x = mx_apply(o, x, r, k, K);
      }
      v[i] = x;
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + 256 * i;
      int k4 = idx % (BK / 4), row = idx / (BK / 4);
      if (idx < TOTAL) {
        float* p = lds + (4 * k4) * (ROWS + PAD) + row;
        p[0] = v[i].x; p[ROWS + PAD] = v[i].y; p[2 * (ROWS + PAD)] = v[i].z; p[3 * (ROWS + PAD)] = v[i].w;
      }
    }
  }
};

// load a [BK x COLS] slab of a row-major [Kdim, ld] matrix (cols contiguous) -> direct store
template <int COLS, int PAD, int BK>
struct SlabN {
  static constexpr int C4 = COLS / 4;
  static constexpr int TOTAL = BK * C4;
  static constexpr int PER = (TOTAL + 255) / 256;
  float4 v[PER];
  __device__ __forceinline__ void load(const MxOperand& o, const float* base, int ld, int c0, int cols_total,
                                       int k0, int Kdim, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + 256 * i;
      int c4 = idx % C4, kk = idx / C4;
      int k = k0 + kk, c = c0 + 4 * c4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < TOTAL && k < Kdim && c < cols_total) {
        x = ld4(base + (long)k * ld + c);
        x = mx_apply(o, x, k, c, cols_total);
      }
      v[i] = x;
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      int idx = tid + 256 * i;
      int c4 = idx % C4, kk = idx / C4;
      if (idx < TOTAL) st4(lds + kk * (COLS + PAD) + 4 * c4, v[i]);
    }
  }
};

// The 128x128 tile is held to 128 registers (4 workgroups per CU instead of 3; no spills): the N = 384 / 640 layers have
// 588 / 980 tiles, which 768 slots run as 0.77 / 1.28 rounds but 1024 slots run as one.
constexpr int gemm_min_waves(int acc_regs, int bk) { return (acc_regs == 64 && bk == 16) ? 4 : 1; }

template <int LAYOUT, int WM, int WN, int TM, int TN, int BK>
__global__ __launch_bounds__(256, gemm_min_waves(TM * TN * 16, BK)) void gemm_kernel(GemmArgs g) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PADA = (LAYOUT == L_TN) ? 4 : 1;
  constexpr int PADB = (LAYOUT == L_NT) ? 1 : 4;
  constexpr int SA = BM + PADA, SB = BN + PADB;
  __shared__ __attribute__((aligned(16))) float smem[2 * BK * SA + 2 * BK * SB];
  float* As = smem;
  float* Bs = smem + 2 * BK * SA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  // Rasterisation.  Default: M tiles fastest.  For the big layers (30 N tiles) an XCD-aware N-fastest order was measured
  // twice: unchanged for fwd/dgrad (the re-streamed A rows come from the 256 MiB Infinity Cache, and the per-XCD window
  // of 128 resident workgroups bounds what L2 can capture), 1.5x slower for the weight gradient, and binding reduction
  // slices to XCDs cost 13 %.  The one case where it pays is below: few N tiles over a long, HBM-bound A operand.
  // With few N tiles (2..16) and a streamed A operand the forward / data-gradient grid is 1-D and remapped: workgroup L
  // runs on XCD L % 8 (round-robin dispatch), and the N tiles of one M tile are given to consecutive workgroups OF THE
  // SAME XCD, so the A rows are fetched from beyond L2 once instead of once per N tile (N = 224 on 128x32 tiles: 7x).
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  int zz = blockIdx.z;
  if (LAYOUT == L_TN && g.xcd_nt > 0) {
    // weight gradient: all output tiles of one reduction slice on one XCD, back to back (they read the same rows)
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int tiles = g.mt * g.xcd_nt;
    const int t = j % tiles;
    zz = (j / tiles) * 8 + xcd;
    if (zz >= g.zt) return;
    tile_m = t % g.mt; tile_n = t / g.mt;
  } else if (g.xcd_nt > 0) {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    tile_n = j % g.xcd_nt;
    tile_m = (j / g.xcd_nt) * 8 + xcd;
    if (tile_m >= g.mt) return;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int z = zz;

  const float* A = g.a.p;
  const float* B = g.b.p;
  float* C = g.c;
  int kbeg = 0, kend = g.K;
  if (LAYOUT == L_TN) {
    kbeg = z * g.ksplit;
    kend = min(g.K, kbeg + g.ksplit);
    C += z * g.sc;          // sc != 0: every reduction slice adds into its own (zeroed) partial matrix, joined in slice order
  } else {
    A += z * g.sa; B += z * g.sb; C += z * g.sc;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  typename std::conditional<LAYOUT == L_TN, SlabN<BM, PADA, BK>, SlabK<BM, PADA, BK>>::type sa;
  typename std::conditional<LAYOUT == L_NT, SlabK<BN, PADB, BK>, SlabN<BN, PADB, BK>>::type sb;

  auto load = [&](int k0) {
    if constexpr (LAYOUT == L_TN) sa.load(g.a, A, g.lda, m0, g.M, k0, kend, tid);
    else sa.load(g.a, A, g.lda, m0, g.M, k0, kend, tid);
    if constexpr (LAYOUT == L_NT) sb.load(g.b, B, g.ldb, n0, g.N, k0, kend, tid);
    else sb.load(g.b, B, g.ldb, n0, g.N, k0, kend, tid);
  };

  MX_GEMM_STAMP(g, 0);
  const int nk = (kend - kbeg + BK - 1) / BK;
  if (nk > 0) {
    load(kbeg);
    sa.store(As, tid);
    sb.store(Bs, tid);
    if (nk > 1) load(kbeg + BK);
  }
  __syncthreads();
  MX_GEMM_STAMP(g, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    // The register slab holds K tile kt+1, requested a whole iteration ago.  It goes to the other LDS buffer right after
    // the barrier (every wave has finished reading that buffer) and the request for tile kt+2 is issued at once, so
    // each global load has a full iteration of MFMA work to land (stores in mid-iteration left it half of one).
    if (kt + 1 < nk) {
      sa.store(As + (cur ^ 1) * BK * SA, tid);
      sb.store(Bs + (cur ^ 1) * BK * SB, tid);
      if (kt + 2 < nk) load(kbeg + (kt + 2) * BK);
    }
    const float* as = As + cur * BK * SA + wm * TM * 32 + l31;
    const float* bs = Bs + cur * BK * SB + wn * TN * 32 + l31;
    // No "k < kend" test inside the MFMA stream: the slab loaders zero-fill rows/columns beyond kend, so the tail of a
    // partial last K tile multiplies zeros.  (With the test, every k-pair became its own basic block and hipcc put an
    // s_waitcnt lgkmcnt(0) between each pair's ds_reads and its four MFMAs, exposing the LDS latency 8 times per tile.)
    // the LDS stores of the next tile sit in the middle of the MFMA stream (the matrix pipe keeps draining the
    // already issued MFMAs while they issue) instead of in front of the barrier, where every wave would stall
    // operands of k-pair kk+2 are read from LDS before the MFMAs of pair kk issue (one pair of register sets, static
    // indices after unrolling), so the ~100-cycle LDS latency hides under the 4 x 64 MFMA cycles in flight
    float av[2][TM], bv[2][TN];
    auto lds_read = [&](int kk, int slot) {
#pragma unroll
      for (int i = 0; i < TM; ++i) av[slot][i] = as[(kk + h) * SA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[slot][j] = bs[(kk + h) * SB + j * 32];
    };
    auto mfma_range = [&](int k_lo, int k_hi) {
#pragma unroll
      for (int kk = k_lo; kk < k_hi; kk += 2) {
        const int slot = (kk >> 1) & 1;
        if (kk + 2 < BK) lds_read(kk + 2, slot ^ 1);
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this pair's MFMAs (hipcc sinks it otherwise)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[slot][i], bv[slot][j], acc[i][j], 0, 0, 0);
      }
    };
    lds_read(0, 0);
    mfma_range(0, BK);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------
  MX_GEMM_STAMP(g, 2);
  float csum[TN], csq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) csum[j] = csq[j] = 0.f;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * TN * 32 + j * 32 + l31;
    const bool cok = col < g.N;
    const float bias = (g.bias && cok) ? g.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (cok && row < g.M) {
          float v = acc[i][j][r];
          const long idx = (long)row * g.ldc + col;
          if (LAYOUT == L_TN) {
            unsafeAtomicAdd(C + idx, v);
          } else {
            v += bias;
            if (g.residual) v += g.residual[idx];
            if (g.relu) v = fmaxf(v, 0.f);
            __builtin_nontemporal_store(v, C + idx);
            csum[j] += v;
            csq[j] += v * v;
          }
        }
      }
    }
  }
  if (LAYOUT != L_TN && g.stats) {
    // column sums of the tile: each wave leaves its 32*TM-row partial in LDS (plain stores; the main loop's last barrier
    // already retired every read of the operand tiles this aliases), one barrier, then the WM partials are added and
    // written as this M tile's partial row (summed in fp64 later).  No LDS atomics, one barrier.
    float* red = smem;   // [WM][2][BN]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float s = csum[j] + __shfl_xor(csum[j], 32, 64);
      float q = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (h == 0) {
        red[(wm * 2 + 0) * BN + wn * TN * 32 + j * 32 + l31] = s;
        red[(wm * 2 + 1) * BN + wn * TN * 32 + j * 32 + l31] = q;
      }
    }
    __syncthreads();
    float* prow = g.stats + (long)tile_m * 2 * g.N;
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, col = i - which * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + which) * BN + col];
      if (n0 + col < g.N) prow[which * g.N + n0 + col] = v;
    }
  }
  MX_GEMM_STAMP(g, 3);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int LAYOUT, int WM, int WN, int TM, int TN, int BK = 16>
static void launch(const GemmArgs& g, int batch_or_splits, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int mt = cdiv(g.M, BM), nt = cdiv(g.N, BN);
  static const int xcd_mode = getenv("MX_GEMM_XCD") ? atoi(getenv("MX_GEMM_XCD")) : 1;
  // experiment knob: unused dynamic LDS for the weight-gradient kernels, which caps their workgroups per CU and so leaves
  // wave slots to the HBM-bound kernels of the main stream while they run on the side stream
  static const int tn_pad = getenv("MX_WGRAD_LDS_PAD") ? atoi(getenv("MX_WGRAD_LDS_PAD")) : 0;
  const int dyn = LAYOUT == L_TN ? tn_pad : 0;
  if (LAYOUT != L_TN && xcd_mode && nt >= 2 && nt <= 16 && mt >= 64) {     // 17..32 N tiles measured: 1-5 % slower
    GemmArgs a = g;
    a.xcd_nt = nt; a.mt = mt;
    dim3 grid(8 * cdiv(mt, 8) * nt, 1, batch_or_splits);
    hipLaunchKernelGGL((gemm_kernel<LAYOUT, WM, WN, TM, TN, BK>), grid, dim3(256), dyn, st, a);
    return;
  }
  // weight gradient: measured win only for the stage-2 layers (401 408 rows, 2-6 output tiles: 277 -> 236, 331 -> 255 us);
  // 5-13 % slower on the 25 088-row layers and 3 % slower at 1.6 M rows, where it stays off
  if (LAYOUT == L_TN && xcd_mode && batch_or_splits >= 8 && mt * nt >= 2 && mt * nt <= 8 && g.K >= 200000 && g.K < 1000000) {
    GemmArgs a = g;
    a.xcd_nt = nt; a.mt = mt; a.zt = batch_or_splits;
    dim3 grid(8 * cdiv(batch_or_splits, 8) * mt * nt, 1, 1);
    hipLaunchKernelGGL((gemm_kernel<LAYOUT, WM, WN, TM, TN, BK>), grid, dim3(256), dyn, st, a);
    return;
  }
  GemmArgs a = g;
  a.xcd_nt = 0; a.mt = mt; a.zt = batch_or_splits;
  dim3 grid(mt, nt, batch_or_splits);
  hipLaunchKernelGGL((gemm_kernel<LAYOUT, WM, WN, TM, TN, BK>), grid, dim3(256), dyn, st, a);
}

// Tile configurations (block tile BM x BN; 4 waves).  EfficientNet's channel counts (48, 80, 160, 224, 288, 480,
// 1344, ...) are multiples of 32 but rarely of 128, so BN is chosen per call.  Measured on MI355X over the B7
// shapes (tools/gemm_sweep.py, profiles/r01_gemm_sweep.txt): the small 128x32 tile (7 workgroups per CU hide the
// staging latency; the A re-reads it causes are absorbed by L2) wins or ties everywhere except where N is a
// multiple of 96 or 128; tiles wider than 128 and 256-row tiles lose at every shape and were dropped.
struct TileCfg { int bm, bn; };
static const TileCfg kCfgs[] = {
    {128, 128},   // 0: 2x2 waves, 2x2 MFMA tiles each
    {128, 96},    // 1: 4x1 waves, 1x3
    {128, 32},    // 2: 4x1 waves, 1x1
    {256, 64},    // 3: 4x1 waves, 2x2   (kept for the tuning sweep)
    {128, 64},    // 4: 4x1 waves, 1x2   (kept for the tuning sweep)
    {128, 128},   // 5: as 0 with BK = 32
    {128, 64},    // 6: 2x2 waves, 2x1
    {128, 64},    // 7: 2x2 waves, 2x1, BK = 32
    {128, 32},    // 8: as 2 with BK = 32
    {128, 96},    // 9: as 1 with BK = 32
    {256, 128},   // 10: 2x2 waves, 4x2 tiles
    {128, 256},   // 11: 2x2 waves, 2x4 tiles
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// Tile choice.  For the weight gradient (TN: the reduction is split until the grid fills the chip anyway) the measured
// rule stands: 128x128 when N is a multiple of 128, 128x96 when a multiple of 96, else 128x32.  For the forward and
// data-gradient GEMMs one more effect matters on 256 CUs: with few tiles per CU the slowest CU sets the time.
// N = 384 at M = 25088 is 588 tiles of 128x128 = 2.3 per CU (some CUs run 3: 77 % balance) but 1176 tiles of 128x64 =
// 4.6 per CU (92 %): measured 487 -> 407 us.  score = tile efficiency x (1 - N padding) x per-CU balance.
static int pick_cfg(int layout, int M, int N, int K) {
  (void)K;
  if (const char* e = getenv("MX_GEMM_CFG")) {           // tuning override (tools/gemm_sweep.py)
    int forced = atoi(e);
    if (forced >= 0 && forced < kNumCfgs) return forced;
  }
  if (layout == L_TN) {
    if (N >= 384 && N % 128 == 0) return 0;
    if (N % 96 == 0) return 1;
    return 2;
  }
  static const struct { int cfg; double eff; } cand[] = {{0, 1.00}, {1, 0.97}, {4, 0.95}, {2, 0.85}};   // big-shape TFLOP/s ratios
  int best = 2;
  double best_score = -1.0;
  for (const auto& c : cand) {
    const TileCfg t = kCfgs[c.cfg];
    const long nt = cdiv(N, t.bn), tiles = (long)cdiv(M, t.bm) * nt;
    const double pad = (double)N / (double)(nt * t.bn);
    const double per_cu = (double)tiles / 256.0;
    const double balance = per_cu / (double)(long)(per_cu + 0.999999);
    if (c.cfg == 4 && N % 64 != 0) continue;               // 128x64 is only measured where it needs no padding
    double score = c.eff * pad * balance;
    if (per_cu < 2.0) score *= 0.85;                         // a lone workgroup per CU cannot hide its own latencies
    if (score > best_score + 1e-9) { best_score = score; best = c.cfg; }
  }
  return best;
}

template <int LAYOUT>
static void dispatch(const GemmArgs& g, int zdim, hipStream_t st) {
  switch (pick_cfg(LAYOUT, g.M, g.N, g.K)) {
    case 0: launch<LAYOUT, 2, 2, 2, 2>(g, zdim, st); break;
    case 1: launch<LAYOUT, 4, 1, 1, 3>(g, zdim, st); break;
    case 2: launch<LAYOUT, 4, 1, 1, 1>(g, zdim, st); break;
    case 3: launch<LAYOUT, 4, 1, 2, 2>(g, zdim, st); break;
    case 4: launch<LAYOUT, 4, 1, 1, 2>(g, zdim, st); break;
    case 5: launch<LAYOUT, 2, 2, 2, 2, 32>(g, zdim, st); break;
    case 6: launch<LAYOUT, 2, 2, 2, 1>(g, zdim, st); break;
    case 7: launch<LAYOUT, 2, 2, 2, 1, 32>(g, zdim, st); break;
    case 8: launch<LAYOUT, 4, 1, 1, 1, 32>(g, zdim, st); break;
    case 9: launch<LAYOUT, 4, 1, 1, 3, 32>(g, zdim, st); break;
    case 10: launch<LAYOUT, 2, 2, 4, 2>(g, zdim, st); break;
    default: launch<LAYOUT, 2, 2, 2, 4>(g, zdim, st); break;
  }
}

// =====================================================================================================================
// NT GEMM, second generation: C[M,N] = A'[M,K] * W[N,K]^T with ROW-MAJOR LDS images and 16-byte operand reads.
//
// Both operands of the forward GEMM have K contiguous, so a [rows][16] slab is a plain copy: one 16-byte global load and
// ONE ds_write_b128 per 4 values (the k-major image above needs four scalar ds_write_b32 for them), and the MFMA operand
// fetch is ONE ds_read_b128 per 16-row tile and K step: lane (l15, q) takes k = 4q..4q+3 of its row and feeds them to four
// consecutive v_mfma_f32_16x16x4_f32 (the hardware sums over the four lane groups; which actual k a group carries is
// free as long as A and B agree).  Row stride 20 floats: conflict-free for the 16 rows a b128 read group touches.
// Measured on the bare loop (tools/hip/mfma_loops.hip, 3 workgroups per CU): 149-152 TFLOP/s against 147 for the
// k-major / ds_read_b32 loop.  The 16x16 MFMA also gives 16-column granularity: N = 48 / 80 / 160 / 224 need no padded
// columns (a 32-wide tile computes 33 % / 20 % / 0 / 14 % zeros there).
// The operand prologue (BN affine + SiLU + SE gate) is applied when the slab moves from registers to LDS, one K step
// after its load was issued: the raw load no longer has to be waited for in the same step.
// The data gradient runs through the same kernel against the transposed weight (mx_transpose).
// Built and measured, not kept (gpurun_out/r2_lab3..8): (i) a persistent stream-K schedule of the 128x128 tile (partial
// tiles handed between workgroups through slabs + agent-scope flags; perfectly balanced, deterministic): 3-8 % SLOWER
// than the plain grid on every MFMA-bound layer (K = 384 -> N = 2304: 429-436 vs 403 us) although in-kernel stamps show
// the plain grid's rounds in lockstep - under this load the chip runs at ~2.0 GHz, and what the schedule gains in
// matrix-pipe occupancy it gives back in clock; (ii) persistent short-K workgroups (whole K in one slab, next tile
// prefetched into registers during MFMA + store): 173 registers, 2 workgroups per CU, 205 vs 163 us on K = 48 -> N = 288.
// More resident workgroups of the simple structure beat software pipelining inside one workgroup here.

// ---- epilogue of the NT kernel.  The main loop feeds the WEIGHT rows as the MFMA's first operand and the activation rows
// as its second, so a lane's four accumulator registers are four CONSECUTIVE OUTPUT CHANNELS of one pixel row
// (acc[i][j][r] = C[m = 16 i + l15][n = 16 j + 4 q + r]): bias, residual, relu and the 16-byte non-temporal store need
// no exchange at all (round-2 first version: through a per-wave LDS patch, 64 ds_write_b32 + 16 ds_read_b128 per lane).
// Column statistics: per-lane sums over the TM row tiles, then a rotate-and-add over the 16 lanes of a DPP row.
// `smem` needs WM*2*BN floats (statistics only); every wave must be out of the main loop.
static __device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

template <int WM, int WN, int TM, int TN>
static __device__ __forceinline__ void nt_epilogue(const GemmArgs& g, float* C, int tile_m, int m0, int n0, f32x4 (&acc)[TM][TN],
                                                   float* smem, int tid) {
  constexpr int BN = WN * TN * 16;
  const int lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  float* red = smem;                                         // [WM][2][BN]
  const bool vec = ((g.ldc | g.N) & 3) == 0;
  f32x4 cs[TN], cq[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) cs[j] = cq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // Interior tiles without bias / ReLU (every expand, project and data-gradient GEMM of the backbone) take a lean path: one base
  // offset per lane, constant strides, packed statistics, no per-element bounds.  The general loop below is ~40 instructions per
  // 16 x 16 sub-tile; beside two other workgroups' MFMA streams on the same SIMDs it took 19 000 cycles per 128 x 128 tile
  // (tools/hip/gemm_lab stamps, profiles/r04_planes_epilogue.txt), a quarter of the workgroup's life at K = 640.
  const bool lean = vec && !g.bias && !g.relu && m0 + WM * TM * 16 <= g.M && n0 + BN <= g.N;
  if (lean) {
    const long idx0 = (long)(m0 + wm * TM * 16 + l15) * g.ldc + n0 + wn * TN * 16 + 4 * q;
    float* cp = C + idx0;
    const long istep = 16L * g.ldc;
    if (g.residual) {
      const float* rp = g.residual + idx0;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const f32x4 v = acc[i][j] + *reinterpret_cast<const f32x4*>(rp + i * istep + 16 * j);
          cs[j] += v; cq[j] += v * v;
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cp + i * istep + 16 * j));
        }
    } else if (g.stats) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const f32x4 v = acc[i][j];
          cs[j] += v; cq[j] += v * v;
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(cp + i * istep + 16 * j));
        }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          __builtin_nontemporal_store(acc[i][j], reinterpret_cast<f32x4*>(cp + i * istep + 16 * j));
    }
  } else
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn * TN * 16 + 16 * j + 4 * q;
    const bool full = vec && col + 3 < g.N;
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) if (col + e < g.N) b[e] = g.bias[col + e];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = m0 + wm * TM * 16 + 16 * i + l15;
      if (row >= g.M || col >= g.N) continue;
      const long idx = (long)row * g.ldc + col;
      float v[4] = {acc[i][j][0] + b[0], acc[i][j][1] + b[1], acc[i][j][2] + b[2], acc[i][j][3] + b[3]};
      if (g.residual) {
        if (full) {
          const float4 r4 = ld4(g.residual + idx);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) if (col + e < g.N) v[e] += g.residual[idx + e];
        }
      }
      if (g.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u = (col + e < g.N) ? v[e] : 0.f;
        cs[j][e] += u; cq[j][e] += u * u;
      }
      if (full) {
        f32x4 o = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(C + idx));
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (col + e < g.N) C[idx + e] = v[e];
      }
    }
  }
  MX_GEMM_STAMP(g, 7);
  if (g.stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      f32x4 s, sq;
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] = row16_sum(cs[j][e]); sq[e] = row16_sum(cq[j][e]); }
      if (l15 == 0) {
        *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + wn * TN * 16 + 16 * j + 4 * q) = s;
        *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + wn * TN * 16 + 16 * j + 4 * q) = sq;
      }
    }
    __syncthreads();
    float* prow = g.stats + (long)tile_m * 2 * g.N;
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, col = i - which * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[(w * 2 + which) * BN + col];
      if (n0 + col < g.N) prow[which * g.N + n0 + col] = v;
    }
  }
}

// one 16-byte chunk of an A' slab through the operand prologue
template <int AMODE>
static __device__ __forceinline__ float4 nt_prologue(float4 v, float4 sc, float4 sh, float4 gt) {
  if (AMODE == MX_PLAIN) return v;
  v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
  if (AMODE == MX_BNACT) {
    v.x = swishf_(v.x) * gt.x; v.y = swishf_(v.y) * gt.y; v.z = swishf_(v.z) * gt.z; v.w = swishf_(v.w) * gt.w;
  }
  return v;
}

constexpr int gemm_nt_waves(int acc) { return acc <= 48 ? 4 : acc <= 64 ? 3 : 2; }

// ---- streaming kernel: one tile per workgroup, K in 16-wide slabs, double-buffered -----------------------------------
template <int WM, int WN, int TM, int TN, int AMODE>
__global__ __launch_bounds__(256, gemm_nt_waves(TM * TN * 4 + (AMODE == MX_BNBWD ? 16 : 0))) void gemm_nt_kernel(GemmArgs g) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16, LS = 20;
  constexpr int PA = (BM * 4 + 255) / 256, PB = (BN * 4 + 255) / 256;
  constexpr int SMEM = 2 * (BM + BN) * LS;
  static_assert(WM * WN == 4, "four waves");
  static_assert(WM * 2 * BN <= SMEM, "statistics staging does not fit the operand buffers");
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if (g.xcd_nt > 0) {        // N tiles of one M tile back to back on one XCD (see gemm_kernel)
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    tile_n = j % g.xcd_nt;
    tile_m = (j / g.xcd_nt) * 8 + xcd;
    if (tile_m >= g.mt) return;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long zb = blockIdx.z;
  const float* A = g.a.p + zb * g.sa;
  const float* B = g.b.p + zb * g.sb;
  float* C = g.c + zb * g.sc;
  const int K = g.K;

  // slab movers: thread t owns rows (t + 256 i) / 4, 16-byte column chunk t % 4
  const int ck = (tid & 3) * 4;
  float4 ra[PA], rb[PB];
  float4 gt[(AMODE == MX_BNACT || AMODE == MX_BNBWD) ? PA : 1];   // BNACT: SE gate of the row's sample; BNBWD: the second tensor
  float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;  // BN scale / shift of this thread's 4 channels (BNBWD: c1, c2)
  float4 c34 = sc4;                                         // BNBWD: c3
  long ga_off[AMODE == MX_BNACT ? PA : 1];
  const float* pa[PA];
  const float* pb[PB];
  bool oka[PA], okb[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (tid + 256 * i) >> 2;
    oka[i] = (tid + 256 * i) < BM * 4 && m0 + row < g.M;
    pa[i] = A + (long)(m0 + row) * g.lda + ck;
    if (AMODE == MX_BNACT) ga_off[i] = g.a.rowp ? (long)((m0 + row) / g.a.rps) * K + ck : -1;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (tid + 256 * i) >> 2;
    okb[i] = (tid + 256 * i) < BN * 4 && n0 + row < g.N;
    pb[i] = B + (long)(n0 + row) * g.ldb + ck;
  }
  auto load = [&](int k0) {
    const bool kin = k0 + ck < K;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      ra[i] = (oka[i] && kin) ? ld4(pa[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == MX_BNACT) gt[i] = (oka[i] && kin && ga_off[i] >= 0) ? ld4(g.a.rowp + ga_off[i] + k0) : make_float4(1.f, 1.f, 1.f, 1.f);
      if (AMODE == MX_BNBWD) gt[i] = (oka[i] && kin) ? ld4(g.a.rowp + (pa[i] - g.a.p) + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (AMODE != MX_PLAIN) {
      sc4 = kin ? ld4(g.a.c1 + k0 + ck) : make_float4(0.f, 0.f, 0.f, 0.f);
      sh4 = kin ? ld4(g.a.c2 + k0 + ck) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == MX_BNBWD) c34 = kin ? ld4(g.a.c1 + 2 * K + k0 + ck) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = (okb[i] && kin) ? ld4(pb[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto store = [&](float* As, float* Bs, int k0) {
    const bool kin = k0 + ck < K;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      if ((tid + 256 * i) < BM * 4) {
        float4 v = ra[i];
        if (AMODE == MX_BNBWD) {
          if (oka[i] && kin) {                               // dX = c1*g + c2*x + c3 (bn_bwd_apply), also kept for the weight gradient
            const float4 x = gt[AMODE == MX_BNBWD ? i : 0];
            v.x = sc4.x * v.x + sh4.x * x.x + c34.x; v.y = sc4.y * v.y + sh4.y * x.y + c34.y;
            v.z = sc4.z * v.z + sh4.z * x.z + c34.z; v.w = sc4.w * v.w + sh4.w * x.w + c34.w;
            if (tile_n == 0) st4(g.a_out + (pa[i] - g.a.p) + k0, v);
          }
        } else if (AMODE != MX_PLAIN && oka[i] && kin) v = nt_prologue<AMODE>(v, sc4, sh4, gt[AMODE == MX_BNACT ? i : 0]);
        st4(As + ((tid + 256 * i) >> 2) * LS + ck, v);
      }
    }
#pragma unroll
    for (int i = 0; i < PB; ++i)
      if ((tid + 256 * i) < BN * 4) st4(Bs + ((tid + 256 * i) >> 2) * LS + ck, rb[i]);
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float* As0 = smem;
  float* Bs0 = smem + 2 * BM * LS;
  MX_GEMM_STAMP(g, 0);
  const int nk = (K + 15) / 16;
  load(0);
  store(As0, Bs0, 0);
  if (nk > 1) load(16);
  __syncthreads();
  MX_GEMM_STAMP(g, 1);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      store(As0 + (cur ^ 1) * BM * LS, Bs0 + (cur ^ 1) * BN * LS, (kt + 1) * 16);
      if (kt + 2 < nk) load((kt + 2) * 16);
    }
    const float* as = As0 + cur * BM * LS + (wm * TM * 16 + l15) * LS + 4 * q;
    const float* bs = Bs0 + cur * BN * LS + (wn * TN * 16 + l15) * LS + 4 * q;
    f32x4 av[TM], bv[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const f32x4*>(as + 16 * LS * i);
#pragma unroll
    for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const f32x4*>(bs + 16 * LS * j);
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j][st], av[i][st], acc[i][j], 0, 0, 0);   // D[n][m]: see nt_epilogue
    __syncthreads();
  }
  MX_GEMM_STAMP(g, 2);
  nt_epilogue<WM, WN, TM, TN>(g, C, tile_m, m0, n0, acc, smem, tid);
  MX_GEMM_STAMP(g, 3);
}

// =====================================================================================================================
// NT GEMM on the bf16 matrix pipe with fp32-exact operands ("split" mode; OPT-IN, see mx_set_gemm_mode).
//
// Every fp32 operand value is split by truncation into three terms whose low 16 bits are zero, x = h + m + l EXACTLY
// (8 + 8 + 8 significand bits), i.e. three bf16 numbers.  a*b = (ah+am+al)(bh+bm+bl); the six products of total order
// <= 2 (hh, hm, mh, hl, lh, mm) are each exact in fp32 and are accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the three
// dropped products are <= 3 * 2^-24 |a b|, the size of ONE fp32 rounding of the product - the accumulation roundings,
// which dominate the error of an fp32 dot product, are the same as in the fp32-MFMA kernel.  Cost: 6 bf16 MFMAs of
// 32 cycles per 32x32x16 block against 32 cycles per 32x32x2 on the fp32 pipe: 2.67x the rate.
// Tile 128 x 128, K in 16-wide slabs, 4 waves of 64 x 64 (2 x 2 MFMA tiles).  LDS: per stage and operand three planes
// [128 rows][16 bf16] (32-byte rows, the two 16-byte halves of a row swapped when bit 3 of the row is set: conflict-free
// for the lane groups of ds_read_b128), 48 KB for two stages: 3 workgroups per CU.  The split runs on the VALU when a
// slab moves from registers to LDS (5.5 operations per value), after the operand prologue.  As in gemm_nt_kernel the
// weight rows are the MFMA's first operand, so a lane's accumulators are runs of four consecutive output channels.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static __device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0) & 0xffff0000u, u1 = __float_as_uint(x1) & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(u0), r1 = x1 - __uint_as_float(u1);                // exact
  const unsigned v0 = __float_as_uint(r0) & 0xffff0000u, v1 = __float_as_uint(r1) & 0xffff0000u;
  const float s0 = r0 - __uint_as_float(v0), s1 = r1 - __uint_as_float(v1);                // exact, <= 8 significant bits
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);     // {bf16(x1), bf16(x0)}: the top halves of both words
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// NJ = 32-column MFMA tiles per wave: tile 128 x (64 NJ).  NJ = 1 doubles the tile count where 128 x 128 leaves CUs idle.
template <int AMODE, int NJ>
__global__ __launch_bounds__(256, NJ == 4 ? 2 : 3) void gemm_nt_split_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 64 * NJ, WN = 2, WNC = 32 * NJ;   // WNC: columns per wave
  constexpr int PA_ = 128 * 32, PB_ = BN * 32;              // bytes of one [rows][16 bf16] plane of A / B
  constexpr int STAGE = 3 * PA_ + 3 * PB_;
  constexpr int PS = WNC + 4;                               // epilogue patch row stride (floats)
  constexpr int SMEM = (2 * STAGE > 4 * 32 * PS * 4 + 2 * 2 * BN * 4) ? 2 * STAGE : 4 * 32 * PS * 4 + 2 * 2 * BN * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if (g.xcd_nt > 0) {
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    tile_n = j % g.xcd_nt;
    tile_m = (j / g.xcd_nt) * 8 + xcd;
    if (tile_m >= g.mt) return;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long zb = blockIdx.z;
  const float* A = g.a.p + zb * g.sa;
  const float* B = g.b.p + zb * g.sb;
  float* C = g.c + zb * g.sc;
  const int K = g.K;

  // slab movers: thread t owns rows t/4 (+64) of A and of B (B: only while < BN), 16-byte column chunk t % 4
  constexpr int NB = BN / 64;
  const int ck = (tid & 3) * 4;
  float4 ra[2], rb[NB];
  float4 gt[AMODE == MX_BNACT ? 2 : 1];
  float4 sc4 = make_float4(0.f, 0.f, 0.f, 0.f), sh4 = sc4;
  long ga_off[AMODE == MX_BNACT ? 2 : 1];
  const float* pa[2];
  const float* pb[NB];
  bool oka[2], okb[NB];
  int woff[2];                                              // byte offset of this thread's 8 bytes inside a plane
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 2) + 64 * i;
    oka[i] = m0 + row < g.M;
    pa[i] = A + (long)(m0 + row) * g.lda + ck;
    if (AMODE == MX_BNACT) ga_off[i] = g.a.rowp ? (long)((m0 + row) / g.a.rps) * K + ck : -1;
    if (i < NB) {
      okb[i] = n0 + row < g.N;
      pb[i] = B + (long)(n0 + row) * g.ldb + ck;
    }
    const int kq = tid & 3;
    woff[i] = row * 32 + (((kq >> 1) ^ ((row >> 3) & 1)) << 4) + ((kq & 1) << 3);
  }
  auto load = [&](int k0) {
    const bool kin = k0 + ck < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      ra[i] = (oka[i] && kin) ? ld4(pa[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (AMODE == MX_BNACT) gt[i] = (oka[i] && kin && ga_off[i] >= 0) ? ld4(g.a.rowp + ga_off[i] + k0) : make_float4(1.f, 1.f, 1.f, 1.f);
      if (i < NB) rb[i] = (okb[i] && kin) ? ld4(pb[i] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (AMODE != MX_PLAIN) {
      sc4 = kin ? ld4(g.a.c1 + k0 + ck) : make_float4(0.f, 0.f, 0.f, 0.f);
      sh4 = kin ? ld4(g.a.c2 + k0 + ck) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto put = [&](unsigned char* base, int plane, int off, float4 v) {
    unsigned h0, m0_, l0, h1, m1, l1;
    split3_pair(v.x, v.y, h0, m0_, l0);
    split3_pair(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2*>(base + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(base + plane + off) = make_uint2(m0_, m1);
    *reinterpret_cast<uint2*>(base + 2 * plane + off) = make_uint2(l0, l1);
  };
  auto store = [&](unsigned char* st, int k0) {
    const bool kin = k0 + ck < K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 v = ra[i];
      if (AMODE != MX_PLAIN && oka[i] && kin) v = nt_prologue<AMODE>(v, sc4, sh4, gt[AMODE == MX_BNACT ? i : 0]);
      put(st, PA_, woff[i], v);
      if (i < NB) put(st + 3 * PA_, PB_, woff[i], rb[i]);
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses of this lane inside a plane
  int fa[2], fb[NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra_ = wm * 64 + 32 * i + l31;
    fa[i] = ra_ * 32 + ((hf ^ ((ra_ >> 3) & 1)) << 4);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int rb_ = wn * WNC + 32 * j + l31;
    fb[j] = rb_ * 32 + ((hf ^ ((rb_ >> 3) & 1)) << 4);
  }
  const int nk = (K + 15) / 16;
  load(0);
  store(smem, 0);
  if (nk > 1) load(16);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      store(smem + (cur ^ 1) * STAGE, (kt + 1) * 16);
      if (kt + 2 < nk) load((kt + 2) * 16);
    }
    const unsigned char* sa = smem + cur * STAGE;
    const unsigned char* sb = sa + 3 * PA_;
    bf16x8 av[2][3], bv[NJ][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i][p] = *reinterpret_cast<const bf16x8*>(sa + p * PA_ + fa[i]);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bv[j][p] = *reinterpret_cast<const bf16x8*>(sb + p * PB_ + fb[j]);
    }
    // six products, smallest terms first: (w,a) = (h,l) (l,h) (m,m) (h,m) (m,h) (h,h); product-major so that consecutive
    // MFMAs write different accumulators; the weight fragment is the first operand (see the epilogue)
    constexpr int PW[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][PW[t]], av[i][PX[t]], acc[i][j], 0, 0, 0);
    __syncthreads();
  }
  // epilogue: acc[i][j][4 gq + e] = C[m = 32 i + l31][n = 32 j + 8 gq + 4 hf + e].  Bias / residual / relu / statistics in
  // registers; the values then cross a per-wave LDS patch [32][WNC] so that a store instruction writes whole rows
  // (64 lanes x 16 bytes = WNC*4-byte row segments) instead of 32-byte pieces.
  float* patch = reinterpret_cast<float*>(smem) + wave * (32 * PS);
  float* red = reinterpret_cast<float*>(smem) + 4 * 32 * PS;           // [2 (wm)][2][BN]
  const bool vec = ((g.ldc | g.N) & 3) == 0;
  f32x4 cs[NJ][4], cq[NJ][4];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) cs[j][gq] = cq[j][gq] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rbase = m0 + wm * 64 + 32 * i;
    const int row = rbase + l31;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int cl = 32 * j + 8 * gq + 4 * hf;             // column inside the wave's WNC columns
        const int col = n0 + wn * WNC + cl;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e] + ((g.bias && col + e < g.N) ? g.bias[col + e] : 0.f);
        if (g.residual && row < g.M) {
          const long idx = (long)row * g.ldc + col;
          if (vec && col + 3 < g.N) {
            const float4 r4 = ld4(g.residual + idx);
            v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (col + e < g.N) v[e] += g.residual[idx + e];
          }
        }
        if (g.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (row < g.M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float u = (col + e < g.N) ? v[e] : 0.f;
            cs[j][gq][e] += u; cq[j][gq][e] += u * u;
          }
        }
        *reinterpret_cast<f32x4*>(patch + l31 * PS + cl) = f32x4{v[0], v[1], v[2], v[3]};
      }
    // patch -> rows: WNC/4 float4 per row, 64 lanes cover 64/(WNC/4) rows per pass
    constexpr int C4 = WNC / 4, RPP = 64 / C4;
#pragma unroll
    for (int t = 0; t < 32 / RPP; ++t) {
      const int rl = t * RPP + lane / C4, c4 = (lane % C4) * 4;
      const int orow = rbase + rl, ocol = n0 + wn * WNC + c4;
      const float4 v4 = ld4(patch + rl * PS + c4);
      if (orow < g.M && ocol < g.N) {
        const long idx = (long)orow * g.ldc + ocol;
        if (vec && ocol + 3 < g.N) {
          __builtin_nontemporal_store(f32x4{v4.x, v4.y, v4.z, v4.w}, reinterpret_cast<f32x4*>(C + idx));
        } else {
          const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) if (ocol + e < g.N) C[idx + e] = v[e];
        }
      }
    }
  }
  if (g.stats) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        f32x4 s, sq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a = row16_sum(cs[j][gq][e]), q = row16_sum(cq[j][gq][e]);
          a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
          s[e] = a; sq[e] = q;
        }
        if (l31 == 0) {
          const int coll = wn * WNC + 32 * j + 8 * gq + 4 * hf;
          *reinterpret_cast<f32x4*>(red + (wm * 2 + 0) * BN + coll) = s;
          *reinterpret_cast<f32x4*>(red + (wm * 2 + 1) * BN + coll) = sq;
        }
      }
    __syncthreads();
    float* prow = g.stats + (long)tile_m * 2 * g.N;
    for (int i = tid; i < 2 * BN; i += 256) {
      const int which = i / BN, col = i - which * BN;
      const float v = red[(0 * 2 + which) * BN + col] + red[(1 * 2 + which) * BN + col];
      if (n0 + col < g.N) prow[which * g.N + n0 + col] = v;
    }
  }
}

// =====================================================================================================================
// Split-arithmetic NT GEMM, second generation (round 4): no operand crosses a VGPR on its way to LDS, and only the weights
// go through LDS at all.
//
// What bounded gemm_nt_split_kernel (profiles/r03_split_gemm_loop_dissection.txt): ~100 of 286 us on the large layers were
// the operand path - the VALU split of BOTH operands in EVERY workgroup (the weights re-split by each of the 196..3136 M
// tiles), twelve ds_write_b64 per thread and K step, register staging one short K step ahead.  Here:
//   * the WEIGHTS are split once per step into three bf16 planes stored as the LDS image itself (mx_pw_planes_batch: per
//     32-deep K step and plane a run of [Npad rows][64 B], the 16-byte chunks of a row permuted by swz_w), so a workgroup's
//     share is copied by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, linear on both sides), two stages;
//   * the ACTIVATIONS never enter LDS: with a 4 x 1 wave layout a wave is the ONLY reader of its 32 rows, so each lane loads
//     its own fragments straight from HBM one K step ahead (two 16-byte loads per row tile; the four k groups of a row take
//     chunks q and 4 + q of the 128-byte line, so one load instruction covers 16 rows x 64 contiguous bytes - the weight
//     image is built in the same k order) and splits them in registers: 88 VALU operations per 96 MFMAs, no redundancy;
//   * v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, the shape that holds the higher clock on random data,
//     MI355X_MICROARCH.md "DVFS give-back" item 7): a lane's weight fragment is one ds_read_b128 of a plane; the accumulator
//     layout is gemm_nt_kernel's, so the epilogue (bias / residual / relu / statistics / 16-byte stores straight from
//     registers) is nt_epilogue unchanged;
//   * LDS: 24 KB per stage at 128 columns, 3 workgroups per CU; one barrier per K step.
// Same arithmetic as the first generation: x = h + m + l exactly, six products (h,l) (l,h) (m,m) (h,m) (m,h) (h,h).
// Conflict-free ds_read_b128: the lane groups of a b128 read ({0-3,12-15,20-27}, ...) hold every row l15 = 0..15 once, with
// the k group q = lane >> 4 taking two values that differ exactly where bit2 ^ bit3 of the row differs; XOR-ing the chunk
// index with 3 * bit3(row) makes the 16-byte slot index a bijection of the four row bits in every group
// (MI355X_MICROARCH.md, LDS).
// Measured against the first generation in one process (tools/hip/gemm_lab planes, profiles/r04_gemm_lab_planes.txt), M = 25088:
// 384 -> 2304 278 -> 234 us, 2304 -> 384 289 -> 231, 640 -> 3840 711 -> 596, 3840 -> 640 722 -> 573, 224 -> 1344 111 -> 93,
// 1344 -> 224 111 -> 97, 160 -> 960 65 -> 55, 960 -> 160 75 -> 65 (190-215 TFLOP/s fp32-equivalent on the K >= 384 layers).
// Also built and measured there, not kept: the activations through LDS as an fp32 image filled by LDS-DMA (80 KB of LDS, 2
// workgroups per CU: equal on the long-K layers, 5-12 % slower on the short-K ones), three LDS stages at 1 workgroup per CU
// (30-50 % slower), a start delay for the second / third resident workgroup of a CU against lockstep epilogues (slower).
static __device__ __forceinline__ int swz_w(int row) { return ((row >> 3) & 1) * 3; }                        // bf16 plane, 4 chunks per row

static __device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// pre-split image of a weight matrix W[N][K] (K % 32 == 0 or 16: a half K step at the end is filled with zeros): planes[((kt * 3 + p) * Npad + n) * 64 + c' * 16 + 2 e] =
// bf16 term p of W[n][32 kt + 4 c + (e < 4 ? e : 12 + e)] with c = c' ^ swz_w(n) (chunk c = k 4c..4c+3 and 16+4c..16+4c+3 of
// the K step: the order gemm_nt_split3_kernel's lanes load the activations in); rows n >= N zero; Npad = N rounded up to 128
// table: rows of 5 longs {src, dst, N, K, first_tile}; a tile = (K step, group of 64 rows) -> Npad/64 * K/32 tiles per matrix
__global__ __launch_bounds__(256) void split_planes_kernel(const long* __restrict__ table, int njobs) {
  __shared__ int job_s;
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < njobs && table[(j + 1) * 5 + 4] <= (long)blockIdx.x) ++j;
    job_s = j;
  }
  __syncthreads();
  const long* e = table + job_s * 5;
  const float* src = reinterpret_cast<const float*>(e[0]);
  unsigned char* dst = reinterpret_cast<unsigned char*>(e[1]);
  const int N = (int)e[2], K = (int)e[3], Npad = (N + 127) & ~127;
  const int tile = (int)(blockIdx.x - e[4]);
  const int rg = Npad / 64, kt = tile / rg, n = (tile % rg) * 64 + (threadIdx.x >> 2);
  const int cp = threadIdx.x & 3, c = cp ^ swz_w(n);
  unsigned h[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, l[4] = {0, 0, 0, 0};
  if (n < N) {
    const float* s = src + (long)n * K + 32 * kt + 4 * c;
    const float4 x0 = ld4(s), x1 = (32 * kt + 16 < K) ? ld4(s + 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    split3_pair(x0.x, x0.y, h[0], m[0], l[0]);
    split3_pair(x0.z, x0.w, h[1], m[1], l[1]);
    split3_pair(x1.x, x1.y, h[2], m[2], l[2]);
    split3_pair(x1.z, x1.w, h[3], m[3], l[3]);
  }
  const long ps = (long)Npad * 64;
  unsigned char* d = dst + ((long)(kt * 3) * Npad + n) * 64 + cp * 16;
  *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(d + ps) = make_uint4(m[0], m[1], m[2], m[3]);
  *reinterpret_cast<uint4*>(d + 2 * ps) = make_uint4(l[0], l[1], l[2], l[3]);
}

// AMODE = MX_BNBWD: the A operand is the BatchNorm backward apply dZ = c1*G + c2*X + c3 (per channel k) formed in registers from
// TWO tensors as the lane loads its fragments (g.a.p = G, g.a.rowp = X, same leading dimension; g.a.c1 = the [3][K] coefficient
// table): the expand convolution's data gradient reads (G, e_raw) instead of a materialised dZ, and the 2R + 1W pass that wrote
// dZ is gone (the weight gradient takes the same prologue on its G operand, wgrad.hip).  Costs 4 more activation loads, 6
// L1-resident coefficient loads and 32 FMAs per lane and K step; the kernel then runs two workgroups per CU.
template <int TN, int AMODE>
__global__ __launch_bounds__(256, AMODE == MX_BNBWD ? 2 : 3) void gemm_nt_split3_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 16 * TN;
  constexpr int W_PLANE = BN * 64, STAGE = 3 * W_PLANE;
  constexpr int WPIECES = 3 * BN / 16, WPW = (WPIECES + 3) / 4;   // (uneven shares are fine: every wave waits for all of its own loads)
  static_assert(4 * 2 * BN * 4 <= 2 * STAGE, "statistics staging does not fit");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, q = lane >> 4;
  int tile_m = blockIdx.x, tile_n = blockIdx.y;
  if (g.xcd_nt > 0) {
    // 1-D grid: the N tiles are cut into chunks of xcd_nt; within a chunk the tiles of one M tile run back to back on ONE XCD
    // (ids L, L + 8, ... share an XCD under round-robin dispatch), so the activation rows cross the fabric once per chunk
    // and the chunk's weight planes (xcd_nt x 128 x K x 6 B) stay in that XCD's L2
    const int per_chunk = 8 * ((g.mt + 7) / 8) * g.xcd_nt;
    const int chunk = blockIdx.x / per_chunk, L = blockIdx.x % per_chunk, xcd = L & 7, j = L >> 3;
    tile_n = chunk * g.xcd_nt + j % g.xcd_nt;
    tile_m = (j / g.xcd_nt) * 8 + xcd;
    if (tile_m >= g.mt || tile_n >= g.zt) return;          // zt: number of N tiles
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const long zb = blockIdx.z;
  const float* A = g.a.p + zb * g.sa;
  const unsigned char* WP = reinterpret_cast<const unsigned char*>(g.b.p);
  float* C = g.c + zb * g.sc;
  const int nk = (g.K + 31) >> 5;
  const bool ktail = (g.K & 31) != 0;        // K % 32 == 16: the last step's second half is zero in the image and is not loaded here
  const long npad64 = (long)g.ldb * 64;

  const float* ap[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int gr = min(m0 + 32 * wave + 16 * i + l15, g.M - 1);   // rows past M re-read the last row (their outputs are not stored)
    ap[i] = A + (long)gr * g.lda + 4 * q;
  }
  const unsigned char* wsrc[WPW];
  int wdst[WPW];
#pragma unroll
  for (int i = 0; i < WPW; ++i) {
    const int pw = wave + 4 * i, p = pw / (BN / 16), rp = pw % (BN / 16);
    wsrc[i] = WP + ((long)p * g.ldb + n0 + 16 * rp) * 64 + lane * 16;
    wdst[i] = p * W_PLANE + rp * 1024;
  }
  auto issue_w = [&](int kt) {
    unsigned char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int i = 0; i < WPW; ++i)
      if (WPIECES % 4 == 0 || wave + 4 * i < WPIECES) glds16(wsrc[i] + 3 * npad64 * kt, st + wdst[i]);
  };
  f32x4 raw[2][2];
  f32x4 rawx[(AMODE == MX_BNBWD || AMODE == MX_BNACT) ? 2 : 1][2], cf[AMODE == MX_BNBWD ? 3 : AMODE == MX_BNACT ? 2 : 1][2];
  const long xoff = AMODE == MX_BNBWD ? (g.a.rowp + zb * g.sa) - A : 0;      // X[r][k] sits xoff floats from G[r][k]
  // AMODE = MX_BNACT (the project convolution of stages 1-3, model.py:86 on swish(bn1(d_raw)) * gate): the lane applies scale / shift per
  // channel k, SiLU and the SE gate of its row's sample to the fragments it loaded, in registers; rawx holds the gate values, cf the
  // scale and shift.  16 sigmoids per lane and K step: only for the narrow outputs of the HBM-bound stages (the launcher keeps TN <= 6)
  const float* gp[AMODE == MX_BNACT ? 2 : 1];
  if (AMODE == MX_BNACT) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int gr = min(m0 + 32 * wave + 16 * i + l15, g.M - 1);
      gp[AMODE == MX_BNACT ? i : 0] = g.a.rowp + (long)(gr / g.a.rps) * g.K + 4 * q;
    }
  }
  auto load_a = [&](int kt) {
    const bool half = AMODE != MX_BNBWD && ktail && kt == nk - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      raw[i][0] = *reinterpret_cast<const f32x4*>(ap[i] + 32 * kt);
      raw[i][1] = half ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(ap[i] + 32 * kt + 16);
      if (AMODE == MX_BNBWD) {
        rawx[AMODE == MX_BNBWD ? i : 0][0] = *reinterpret_cast<const f32x4*>(ap[i] + xoff + 32 * kt);
        rawx[AMODE == MX_BNBWD ? i : 0][1] = *reinterpret_cast<const f32x4*>(ap[i] + xoff + 32 * kt + 16);
      }
      if (AMODE == MX_BNACT) {
        rawx[AMODE == MX_BNACT ? i : 0][0] = *reinterpret_cast<const f32x4*>(gp[AMODE == MX_BNACT ? i : 0] + 32 * kt);
        rawx[AMODE == MX_BNACT ? i : 0][1] = *reinterpret_cast<const f32x4*>(gp[AMODE == MX_BNACT ? i : 0] + 32 * kt + 16);
      }
    }
    if (AMODE == MX_BNACT) {
      cf[0][0] = *reinterpret_cast<const f32x4*>(g.a.c1 + 32 * kt + 4 * q);
      cf[0][1] = *reinterpret_cast<const f32x4*>(g.a.c1 + 32 * kt + 4 * q + 16);
      cf[AMODE == MX_BNACT ? 1 : 0][0] = *reinterpret_cast<const f32x4*>(g.a.c2 + 32 * kt + 4 * q);
      cf[AMODE == MX_BNACT ? 1 : 0][1] = *reinterpret_cast<const f32x4*>(g.a.c2 + 32 * kt + 4 * q + 16);
    }
    if (AMODE == MX_BNBWD) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        cf[AMODE == MX_BNBWD ? c : 0][0] = *reinterpret_cast<const f32x4*>(g.a.c1 + (long)c * g.K + 32 * kt + 4 * q);
        cf[AMODE == MX_BNBWD ? c : 0][1] = *reinterpret_cast<const f32x4*>(g.a.c1 + (long)c * g.K + 32 * kt + 4 * q + 16);
      }
    }
  };

  f32x4 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int woff = l15 * 64 + ((q ^ swz_w(l15)) << 4);

  MX_GEMM_STAMP(g, 0);
  issue_w(0);
  load_a(0);
  MX_GEMM_STAMP(g, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's loads of step kt (weight pieces by LDS-DMA, its own activation rows) are done; split before the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (AMODE == MX_BNBWD) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            raw[i][u][e] = cf[0][u][e] * raw[i][u][e] + (cf[AMODE == MX_BNBWD ? 1 : 0][u][e] * rawx[AMODE == MX_BNBWD ? i : 0][u][e] +
                                                         cf[AMODE == MX_BNBWD ? 2 : 0][u][e]);
      }
      if (AMODE == MX_BNACT) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            raw[i][u][e] = swishf_(cf[0][u][e] * raw[i][u][e] + cf[AMODE == MX_BNACT ? 1 : 0][u][e]) * rawx[AMODE == MX_BNACT ? i : 0][u][e];
      }
      unsigned h[4], m[4], l[4];
      split3_pair(raw[i][0][0], raw[i][0][1], h[0], m[0], l[0]);
      split3_pair(raw[i][0][2], raw[i][0][3], h[1], m[1], l[1]);
      split3_pair(raw[i][1][0], raw[i][1][1], h[2], m[2], l[2]);
      split3_pair(raw[i][1][2], raw[i][1][3], h[3], m[3], l[3]);
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      af[i][0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
      af[i][1] = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
      af[i][2] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    }
    // s_barrier is not a memory clobber for the compiler: without the two compiler fences the ds_reads of `st` below and the next
    // stage's LDS-DMA could legally be scheduled across it (the hardware ordering is the s_waitcnt above + the barrier itself)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave's share of stage kt has landed; every read of the other stage is retired
    asm volatile("" ::: "memory");
    if (kt + 1 < nk) { issue_w(kt + 1); load_a(kt + 1); }
    const unsigned char* st = smem + (kt & 1) * STAGE;
    constexpr int PW[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      bf16x8 wf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) wf[p] = *reinterpret_cast<const bf16x8*>(st + woff + p * W_PLANE + j * 1024);
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[PW[t]], af[i][PX[t]], acc[i][j], 0, 0, 0);
    }
  }
  MX_GEMM_STAMP(g, 2);
  __syncthreads();
  nt_epilogue<4, 1, 2, TN>(g, C, tile_m, m0, n0, acc, reinterpret_cast<float*>(smem), tid);
  MX_GEMM_STAMP(g, 3);
}

// 0 = exact-fp32 MFMA everywhere; 1 (default since round 3) = split arithmetic for the MFMA-bound forward / data-gradient /
// weight-gradient shapes, exact-fp32 MFMA for the rest; 2 = split arithmetic for every NT GEMM (tests).  Process-wide;
// MX_GEMM_SPLIT sets the initial value.  Every golden / oracle parity test runs in modes 0 AND 1 with the same tolerances
// (tests/conftest.py, marker both_arith).
static int g_gemm_mode = getenv("MX_GEMM_SPLIT") ? atoi(getenv("MX_GEMM_SPLIT")) : 1;
// which forward / data-gradient GEMMs take the split-bf16 kernel: mode 2 all; mode 1 the MFMA-bound shapes only
// (K and N large enough, <= 26 % padded columns)
static bool nt_uses_split(int N, int K) {
  return g_gemm_mode == 2 || (g_gemm_mode == 1 && K >= 128 && N >= 96 && (double)N / (64.0 * cdiv(N, 64)) >= 0.74);
}

template <int NJ>
static void launch_nt_split_t(const GemmArgs& g, int batch, hipStream_t st) {
  constexpr int BN = 64 * NJ;
  const int mt = cdiv(g.M, 128), nt = cdiv(g.N, BN);
  GemmArgs a = g;
  a.mt = mt;
  a.xcd_nt = 0;
  dim3 grid(mt, nt, batch);
  if (nt >= 2 && nt <= 16 && mt >= 64) { a.xcd_nt = nt; grid = dim3(8 * cdiv(mt, 8) * nt, 1, batch); }
  switch (g.a.mode) {
    case MX_PLAIN: hipLaunchKernelGGL((gemm_nt_split_kernel<MX_PLAIN, NJ>), grid, dim3(256), 0, st, a); break;
    case MX_BNACT: hipLaunchKernelGGL((gemm_nt_split_kernel<MX_BNACT, NJ>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_nt_split_kernel<MX_AFFINE, NJ>), grid, dim3(256), 0, st, a); break;
  }
}

// 128 x 128 unless the narrower tile pads fewer columns or balances the 768 resident workgroups (3 per CU) better
static void launch_nt_split(const GemmArgs& g, int batch, hipStream_t st) {
  static const int forced = getenv("MX_GEMM_SPLIT_NJ") ? atoi(getenv("MX_GEMM_SPLIT_NJ")) : 0;
  auto score = [&](int bn, double eff) {
    const long nt = cdiv(g.N, bn), tiles = (long)cdiv(g.M, 128) * nt * batch;
    const double pad = (double)g.N / (double)(nt * bn);
    const double rounds = (double)tiles / 768.0;
    const double balance = rounds / (double)(long)(rounds + 0.999999);
    return eff * pad * balance;
  };
  // Long reductions (the project convs and their mirror, the expand data gradient: K = Cexp) are priced with a model fitted to
  // measurements (tools/gemm_nj.py): a launch costs floor(rounds) + min(1, tail + 0.12) rounds of 768 workgroups - a partly
  // filled last round is cheaper than a full one because its workgroups share their CU with fewer neighbours - and the
  // 64-wide tile does 0.87 of the 128-wide tile's work per unit time.  (K = 3840 -> N = 640 took 876 us with the 64-wide tile
  // the first rule picked and 788 us with the 128-wide one.)
  static const int fitted = getenv("MX_GEMM_SPLIT_FITTED") ? atoi(getenv("MX_GEMM_SPLIT_FITTED")) : 1;   // 0: first rule only (A/B)
  auto cost = [&](int bn, double eff) {
    const double rounds = (double)cdiv(g.M, 128) * cdiv(g.N, bn) * batch / 768.0;
    const double full = (double)(long)rounds, tail = rounds - full;
    return (full + (tail > 0 ? (tail + 0.12 < 1.0 ? tail + 0.12 : 1.0) : 0.0)) * bn / eff;
  };
  const bool narrow = forced ? forced == 1
                    : (g.K >= 1024 && fitted) ? cost(64, 0.87) < cost(128, 1.0)
                                  : score(64, 0.92) > score(128, 1.0) + 1e-9;
  if (forced == 4) launch_nt_split_t<4>(g, batch, st);
  else if (narrow) launch_nt_split_t<1>(g, batch, st);
  else launch_nt_split_t<2>(g, batch, st);
}

// ---- second-generation split kernel: pre-split weight planes (mx_pw_planes_batch) -----------------------------------
static int g_split3_xcd_chunk = getenv("MX_SPLIT3_XCD_CHUNK") ? atoi(getenv("MX_SPLIT3_XCD_CHUNK")) : 6;   // N tiles per XCD-local chunk when there are more than 16
template <int TN>
static void launch_nt_split3_t(const GemmArgs& g, int batch, hipStream_t st) {
  constexpr int BN = 16 * TN;
  const int mt = cdiv(g.M, 128), nt = cdiv(g.N, BN);
  GemmArgs a = g;
  a.mt = mt;
  a.xcd_nt = 0;
  dim3 grid(mt, nt, batch);
  if (nt >= 2 && mt >= 64) {
    // up to 16 N tiles: one chunk; more (N = 2304: 18, N = 3840: 30): chunks of ~6 (measured 227 -> 219 us and 572 -> 545 us;
    // eleven tiles cut into 6 + 5 lost 10 %, so the short lists stay whole)
    const int chunks = nt <= 16 ? 1 : cdiv(nt, g_split3_xcd_chunk);
    a.xcd_nt = cdiv(nt, chunks); a.zt = nt;
    grid = dim3(chunks * 8 * cdiv(mt, 8) * a.xcd_nt, 1, batch);
  }
  if (g.a.mode == MX_BNBWD) hipLaunchKernelGGL((gemm_nt_split3_kernel<TN, MX_BNBWD>), grid, dim3(256), 0, st, a);
  else if (g.a.mode == MX_BNACT) {
    if constexpr (TN <= 6) hipLaunchKernelGGL((gemm_nt_split3_kernel<TN, MX_BNACT>), grid, dim3(256), 0, st, a);     // (launch_nt_split3 keeps the activated form at <= 96 columns)
  } else hipLaunchKernelGGL((gemm_nt_split3_kernel<TN, MX_PLAIN>), grid, dim3(256), 0, st, a);
}

// Tile width: 128 columns, or 112 / 96 / 80 / 64 where that pads less or balances better.  Tiles are dealt over 256 CUs: time ~
// (tiles per CU + half a tile of tail) x tile width / efficiency of the width (a narrower tile re-loads and re-splits the same
// activation rows for fewer columns: 0.85 at 64, measured 0.84-0.96 on the M = 25088 layers); a grid of fewer than ~1.2 wide tiles
// per CU always takes the 64-wide tile (M = 6272: 2304 -> 384 108 -> 91 us, M = 12544: 1152 -> 192 61 -> 51).  The odd widths are
// only taken where they divide N (N = 160 = 2 x 80, 224 = 2 x 112, 960 = 10 x 96, 2304 = 24 x 96): no padded columns at all
// (960 -> 160: 69 us at 64 columns, 55 at 80; 1344 -> 224: 99 at 128, 91 at 112; 480 -> 80 at 100 352 rows: 73 -> 54).
static int g_split_nj = getenv("MX_GEMM_SPLIT_NJ") ? atoi(getenv("MX_GEMM_SPLIT_NJ")) : 0;     // 1: force 128 x 64, 2: force 128 x 128, 3: 64 / 128 only
static void launch_nt_split3(const GemmArgs& g, int batch, hipStream_t st) {
  auto cost = [&](int bn, double eff) {
    const double per_cu = (double)cdiv(g.M, 128) * cdiv(g.N, bn) * batch / 256.0;
    return (per_cu + 0.5) * bn / eff;
  };
  const long wide_tiles = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * batch;
  int bn = 128;
  if (g_split_nj >= 16) bn = g_split_nj;                                   // lab: force this width
  else if (g_split_nj == 1 || (g_split_nj != 2 && wide_tiles <= 300)) bn = 64;
  else if (g_split_nj != 2) {
    // short reductions (K <= 512: few K steps per tile, so a tile's fill and epilogue weigh more) reward the widths that run four
    // workgroups per CU (96 and 80 columns: 112-128 registers, <= 36 KB of LDS): tools/hip/gemm_lab widths,
    // profiles/r04_gemm_lab_widths.txt - 384 -> 2304: 236 (128) / 216 (96); 224 -> 1344: 97 / 89.5; 160 -> 960: 65 / 57.5
    const bool shortk = g.K <= 512;
    const double e64 = shortk ? 0.90 : 0.85, e80 = shortk ? 0.97 : 0.90, e96 = shortk ? 1.03 : 0.94, e112 = 0.97;
    double best = cost(128, 1.0);
    if (cost(64, e64) < best) { best = cost(64, e64); bn = 64; }
    if (g_split_nj != 3) {
      if (g.N % 112 == 0 && cost(112, e112) < best) { best = cost(112, e112); bn = 112; }
      if (g.N % 96 == 0 && cost(96, e96) < best) { best = cost(96, e96); bn = 96; }
      if (g.N % 80 == 0 && cost(80, e80) < best) { best = cost(80, e80); bn = 80; }
    }
  }
  if (g.a.mode == MX_BNACT && bn > 96) bn = (g.N % 96 == 0) ? 96 : 64;     // (the activated form exists up to 96 columns)
  switch (bn) {
    case 64: launch_nt_split3_t<4>(g, batch, st); break;
    case 80: launch_nt_split3_t<5>(g, batch, st); break;
    case 96: launch_nt_split3_t<6>(g, batch, st); break;
    case 112: launch_nt_split3_t<7>(g, batch, st); break;
    default: launch_nt_split3_t<8>(g, batch, st); break;
  }
}

// tile table of the second-generation NT kernel: {BM, BN}; every entry has 128 rows
struct NtCfg { int bm, bn; };
static const NtCfg kNtCfgs[] = {{128, 128}, {128, 96}, {128, 64}, {128, 48}, {128, 80}, {128, 32}, {128, 160}};
constexpr int kNumNtCfgs = sizeof(kNtCfgs) / sizeof(kNtCfgs[0]);

template <int WM, int WN, int TM, int TN>
static void launch_nt(const GemmArgs& g, int batch, hipStream_t st) {
  constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
  const int mt = cdiv(g.M, BM), nt = cdiv(g.N, BN);
  GemmArgs a = g;
  a.mt = mt;
  a.xcd_nt = 0;
  dim3 grid(mt, nt, batch);
  if (nt >= 2 && nt <= 16 && mt >= 64) { a.xcd_nt = nt; grid = dim3(8 * cdiv(mt, 8) * nt, 1, batch); }
  switch (g.a.mode) {
    case MX_PLAIN: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, MX_PLAIN>), grid, dim3(256), 0, st, a); break;
    case MX_BNACT: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, MX_BNACT>), grid, dim3(256), 0, st, a); break;
    case MX_BNBWD: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, MX_BNBWD>), grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, MX_AFFINE>), grid, dim3(256), 0, st, a); break;
  }
}

// Choice among the NT tiles: fewest padded columns first (16-column granularity), then the widest tile whose grid still
// gives every CU at least ~2 workgroups.
static int pick_nt_cfg(int M, int N) {
  if (const char* e = getenv("MX_GEMM_NT_CFG")) {
    int forced = atoi(e);
    if (forced >= 0 && forced < kNumNtCfgs) return forced;
  }
  int best = 0;
  double best_score = -1.0;
  for (int c = 0; c < kNumNtCfgs; ++c) {
    const NtCfg t = kNtCfgs[c];
    const long nt = cdiv(N, t.bn), tiles = (long)cdiv(M, t.bm) * nt;
    const double pad = (double)N / (double)(nt * t.bn);
    const double per_cu = (double)tiles / 256.0;
    const double balance = per_cu / (double)(long)(per_cu + 0.999999);
    double eff = t.bn >= 128 ? 1.0 : t.bn >= 96 ? 0.98 : t.bn >= 64 ? 0.95 : t.bn >= 48 ? 0.92 : 0.88;
    double score = eff * pad * balance;
    if (per_cu < 2.0) score *= 0.85;
    if (score > best_score + 1e-9) { best_score = score; best = c; }
  }
  return best;
}

static void dispatch_nt(const GemmArgs& g, int batch, hipStream_t st) {
  if (g.a.mode != MX_BNBWD && nt_uses_split(g.N, g.K)) {
    launch_nt_split(g, batch, st);       // MFMA-bound shapes only: K and N large enough, <= 26 % padded columns
    return;
  }
  switch (pick_nt_cfg(g.M, g.N)) {
    case 0: launch_nt<2, 2, 4, 4>(g, batch, st); break;     // 128 x 128
    case 1: launch_nt<2, 2, 4, 3>(g, batch, st); break;     // 128 x 96
    case 2: launch_nt<2, 2, 4, 2>(g, batch, st); break;     // 128 x 64
    case 3: launch_nt<4, 1, 2, 3>(g, batch, st); break;     // 128 x 48
    case 4: launch_nt<4, 1, 2, 5>(g, batch, st); break;     // 128 x 80
    case 5: launch_nt<4, 1, 2, 2>(g, batch, st); break;     // 128 x 32
    default: launch_nt<2, 2, 4, 5>(g, batch, st); break;    // 128 x 160
  }
}

static int check_operand(const MxOperand& o, const char* nm) {
  MX_CHECK_ARG(o.p != nullptr, "gemm: operand %s is null", nm);
  MX_CHECK_ARG(((uintptr_t)o.p & 15) == 0, "gemm: operand %s not 16-byte aligned", nm);
  MX_CHECK_ARG(o.mode >= 0 && o.mode <= 3, "gemm: operand %s bad mode %d", nm, o.mode);
  if (o.mode != MX_PLAIN) {
    MX_CHECK_ARG(o.c1 && o.c2, "gemm: operand %s needs scale/shift", nm);
    MX_CHECK_ARG(o.rps > 0, "gemm: operand %s rows_per_sample must be > 0", nm);
  }
  return MX_OK;
}

// TN (weight gradient): the pixel reduction is split so the grid fills the chip (>= ~1024 blocks), >= 512 rows per slice
static int tn_splits(int M, int N, int K, int* ksplit) {
  const TileCfg tc = kCfgs[pick_cfg(L_TN, M, N, K)];
  long tiles = (long)cdiv(M, tc.bm) * cdiv(N, tc.bn);
  // workgroup target of the split (tuning override MX_GEMM_SPLIT_TARGET; swept 1024/1536/2048/3072/4096 on the final
  // code: 150.7 / 149.6 / 148.9 / 149.8 / 149.6 ms per step)
  static const long split_target = getenv("MX_GEMM_SPLIT_TARGET") ? atol(getenv("MX_GEMM_SPLIT_TARGET")) : 2048;
  int splits = (int)((split_target + tiles - 1) / tiles);
  int maxs = cdiv(K, 512);
  if (splits > maxs) splits = maxs;
  if (splits < 1) splits = 1;
  int ks = cdiv(cdiv(K, splits), 32) * 32;
  *ksplit = ks;
  return cdiv(K, ks);
}

static int gemm_common(int layout, GemmArgs& g, int batch, hipStream_t st) {
  if (int e = check_operand(g.a, "A")) return e;
  if (int e = check_operand(g.b, "B")) return e;
  MX_CHECK_ARG(g.c != nullptr, "gemm: C is null");
  MX_CHECK_ARG(g.M > 0 && g.N > 0 && g.K > 0, "gemm: bad extents M=%d N=%d K=%d", g.M, g.N, g.K);
  MX_CHECK_ARG(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: leading dims must be multiples of 4 (lda=%d ldb=%d)", g.lda, g.ldb);
  MX_CHECK_ARG(batch >= 1, "gemm: batch must be >= 1");
  g.stamps = mx_gemm_stamps;
  if (layout == L_NT) {
    MX_CHECK_ARG(g.K % 4 == 0, "gemm NT: K=%d must be a multiple of 4", g.K);
    static const int nt_v2 = getenv("MX_GEMM_NT_V2") ? atoi(getenv("MX_GEMM_NT_V2")) : 1;
    if (nt_v2 && g.b.mode == MX_PLAIN) dispatch_nt(g, batch, st);
    else dispatch<L_NT>(g, batch, st);
  } else if (layout == L_NN) {
    MX_CHECK_ARG(g.K % 4 == 0 && g.N % 4 == 0, "gemm NN: K=%d and N=%d must be multiples of 4", g.K, g.N);
    dispatch<L_NN>(g, batch, st);
  } else {
    MX_CHECK_ARG(g.M % 4 == 0 && g.N % 4 == 0, "gemm TN: M=%d and N=%d must be multiples of 4", g.M, g.N);
    MX_CHECK_ARG(batch == 1, "gemm TN: not batched");
    const int splits = tn_splits(g.M, g.N, g.K, &g.ksplit);
    if (splits > 1) {
      // deterministic join: slice z accumulates into partial matrix z of the scratch (one adder per element), a second
      // kernel adds the partial matrices to dW in slice order.  (The slices used to meet in dW through fp32 atomics.)
      float* dW = g.c;
      g.c = g.ws; g.sc = (long)g.M * g.N; g.ldc = g.N;
      hipMemsetAsync(g.ws, 0, sizeof(float) * (size_t)splits * g.M * g.N, st);
      dispatch<L_TN>(g, splits, st);
      MX_LAUNCH_CHECK();
      mx_launch_parts_reduce(g.ws, splits, g.M * g.N, dW, st);
    } else {
      dispatch<L_TN>(g, 1, st);
    }
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// dst[cols][rows] = src[rows][cols]^T (weights only: a few MB at most), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float t[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = by + ty + i, c = bx + tx;
    if (r < rows && c < cols) t[ty + i][tx] = src[(long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = bx + ty + i, r = by + tx;
    if (r < rows && c < cols) dst[(long)c * rows + r] = t[tx][ty + i];
  }
}

// Every conv weight of the network in one launch: table rows of 5 longs {src, dst, rows, cols, first tile}; workgroup id ->
// (matrix, 32x32 tile) by a scan of the first-tile column (~110 rows).
__global__ __launch_bounds__(256) void transpose_batch_kernel(const long* __restrict__ table, int n) {
  __shared__ float t[32][33];
  __shared__ int job;
  if (threadIdx.x == 0) {
    int j = 0;
    while (j + 1 < n && table[(j + 1) * 5 + 4] <= (long)blockIdx.x) ++j;
    job = j;
  }
  __syncthreads();
  const long* e = table + job * 5;
  const float* src = reinterpret_cast<const float*>(e[0]);
  float* dst = reinterpret_cast<float*>(e[1]);
  const int rows = (int)e[2], cols = (int)e[3];
  const int tile = (int)(blockIdx.x - e[4]), tx_n = (cols + 31) / 32;
  const int bx = (tile % tx_n) * 32, by = (tile / tx_n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = by + ty + i, c = bx + tx;
    if (r < rows && c < cols) t[ty + i][tx] = src[(long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = bx + ty + i, r = by + tx;
    if (r < rows && c < cols) dst[(long)c * rows + r] = t[tx][ty + i];
  }
}

extern "C" {

// n transposes in one launch.  table: DEVICE array of n rows {src, dst, rows, cols, first_tile} (longs), first_tile = the
// running sum of ceil(rows/32) * ceil(cols/32) over the rows before; total_tiles = that sum over all rows.
int mx_transpose_batch(const long* table, int n, int total_tiles, void* stream) {
  MX_CHECK_ARG(table && n > 0 && total_tiles > 0, "transpose_batch: bad arguments");
  hipLaunchKernelGGL(transpose_batch_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// dst[cols, rows] = src[rows, cols]^T.  The data gradient of a 1x1 convolution is run as a forward GEMM against the
// transposed weight (mx_pw_fwd with W^T: both operands K-contiguous, 16-byte LDS traffic, 16-column tiles).
int mx_transpose(const float* src, float* dst, int rows, int cols, void* stream) {
  MX_CHECK_ARG(src && dst && rows > 0 && cols > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, cols);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// ---- pre-split weight planes --------------------------------------------------------------------------------------
// bytes of the image mx_pw_planes writes for a weight W[N][K]; MX_EARG when the shape has none (K % 32 is neither 0 nor 16)
long mx_pw_planes_bytes(int N, int K) {
  if (N <= 0 || K <= 0 || K % 16) return MX_EARG;
  return (long)((N + 127) & ~127) * ((K + 31) & ~31) * 6;
}

// number of workgroups (tiles) mx_pw_planes_batch needs for one matrix
int mx_pw_planes_tiles(int N, int K) {
  if (N <= 0 || K <= 0 || K % 16) return MX_EARG;
  return (((N + 127) & ~127) / 64) * ((K + 31) / 32);
}

// n weight matrices in one launch.  table: DEVICE array of n rows of 5 longs {src W[N][K] fp32, dst image, N, K, first_tile};
// first_tile = running sum of mx_pw_planes_tiles over the rows before; total_tiles = the full sum
int mx_pw_planes_batch(const long* table, int n, int total_tiles, void* stream) {
  MX_CHECK_ARG(table && n > 0 && total_tiles > 0, "pw_planes_batch: bad arguments");
  hipLaunchKernelGGL(split_planes_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, n);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// 1 when, in the current mode, a forward / data-gradient GEMM of this shape should be run through mx_pw_fwd_planes
int mx_pw_fwd_uses_planes(int M, int K, int N) {
  static const int on = getenv("MX_SPLIT2") ? atoi(getenv("MX_SPLIT2")) : 1;
  static const int early = getenv("MX_SPLIT3_EARLY") ? atoi(getenv("MX_SPLIT3_EARLY")) : 1;
  static const int tail = getenv("MX_SPLIT3_KTAIL") ? atoi(getenv("MX_SPLIT3_KTAIL")) : 1;
  if (!on || M <= 0 || K % 16) return 0;
  // K = 48 / 80 (the expand convolutions of stages 2-3 and the project data gradients beside them, 0.1-0.4 M rows): HBM-bound, and
  // the second-generation kernel's register-direct activation stream wins there as well; the half K step it pads costs MFMA time only
  if (K % 32) return (g_gemm_mode != 0 && tail && K >= 48 && N >= 96) ? 1 : 0;
  if (nt_uses_split(N, K)) return 1;
  // the HBM-bound data gradients of stages 2-3 (K = 288 -> 48 at 401 408 rows, 480 -> 80 at 100 352): the second-generation
  // kernel streams their long operand straight into registers and beats the exact-fp32 kernel there too (137 -> 127 us,
  // 102 -> 78 us; tools/hip/gemm_lab planes); narrower outputs / shorter reductions stay where they are (32 -> 32: 73 vs 90 us)
  return (g_gemm_mode == 1 && early && K >= 192 && N >= 48) ? 1 : 0;
}

// C[M,N] = A[M,K] * W[N,K]^T (+bias) (+residual) (relu) with W given as its pre-split image (mx_pw_planes_batch) and a plain A:
// split arithmetic, second-generation kernel.
int mx_pw_fwd_planes(const float* A, const void* Wplanes, float* C, int M, int K, int N, int lda, int ldc,
                     const float* bias, const float* residual, int relu, float* stats, void* stream) {
  MX_CHECK_ARG(A && Wplanes && C, "pw_fwd_planes: null pointer");
  MX_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 16 == 0, "pw_fwd_planes: bad extents M=%d N=%d K=%d (K must be a multiple of 16)", M, N, K);
  MX_CHECK_ARG(lda % 4 == 0 && lda >= K && ldc >= N, "pw_fwd_planes: bad leading dimensions lda=%d ldc=%d", lda, ldc);
  MX_CHECK_ARG((((uintptr_t)A | (uintptr_t)Wplanes | (uintptr_t)C) & 15) == 0, "pw_fwd_planes: pointers must be 16-byte aligned");
  GemmArgs g{};
  g.a = MxOperand{A, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.b = MxOperand{reinterpret_cast<const float*>(Wplanes), nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = (N + 127) & ~127; g.ldc = ldc;
  g.bias = bias; g.residual = residual; g.relu = relu; g.stats = stats;
  g.stamps = mx_gemm_stamps;
  launch_nt_split3(g, 1, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// 1 when, in the current mode, the project convolution's forward GEMM on swish(scale*A + shift) * gate should go through
// mx_pw_fwd_planes_act: the narrow outputs of stages 2-3 (K = 192 / 288 -> 48, 288 / 480 -> 80 at 0.1-0.4 M rows), HBM-bound - the
// register-direct activation stream beats the exact-fp32 kernel's LDS slabs there as it does for the plain form (MX_SPLIT3_ACT=0: off)
int mx_pw_fwd_act_uses_planes(int M, int K, int N) {
  static const int on = getenv("MX_SPLIT3_ACT") ? atoi(getenv("MX_SPLIT3_ACT")) : 1;
  static const int split2 = getenv("MX_SPLIT2") ? atoi(getenv("MX_SPLIT2")) : 1;
  return (on && split2 && g_gemm_mode != 0 && M >= 65536 && K % 32 == 0 && K >= 192 && N >= 48 && N <= 128) ? 1 : 0;
}

// C[M,N] = (swish(scale[k]*A[m,k] + shift[k]) * gate[m / rows_per_sample, k]) * W[N,K]^T with W as its pre-split image: the operand
// prologue of mx_pw_fwd's a_mode 1 (model.py:83-86) in the second-generation split kernel's register loads.  K % 32 == 0.
int mx_pw_fwd_planes_act(const float* A, const float* scale, const float* shift, const float* gate, int rows_per_sample,
                         const void* Wplanes, float* C, int M, int K, int N, int lda, int ldc, float* stats, void* stream) {
  MX_CHECK_ARG(A && scale && shift && gate && Wplanes && C, "pw_fwd_planes_act: null pointer");
  MX_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 32 == 0 && rows_per_sample > 0, "pw_fwd_planes_act: bad extents M=%d N=%d K=%d rows_per_sample=%d", M, N, K, rows_per_sample);
  MX_CHECK_ARG(lda % 4 == 0 && lda >= K && ldc >= N, "pw_fwd_planes_act: bad leading dimensions lda=%d ldc=%d", lda, ldc);
  MX_CHECK_ARG((((uintptr_t)A | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)gate | (uintptr_t)Wplanes | (uintptr_t)C) & 15) == 0,
               "pw_fwd_planes_act: pointers must be 16-byte aligned");
  GemmArgs g{};
  g.a = MxOperand{A, scale, shift, gate, MX_BNACT, rows_per_sample};
  g.b = MxOperand{reinterpret_cast<const float*>(Wplanes), nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = (N + 127) & ~127; g.ldc = ldc;
  g.stats = stats;
  launch_nt_split3(g, 1, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// dX[M,N] = dZ[M,K] * Wt[N,K]^T (+ residual) with dZ = c1*G + c2*X + c3 (coef = [3][K], as mx_bn_bwd_finalize leaves it) formed in
// the operand load of the second-generation split kernel from G and X [M, ldg]; Wt given as its pre-split image.  dZ is NOT written.
int mx_pw_dgrad_bnbwd_planes(const float* G, const float* X, const float* coef, const void* WtPlanes, float* dX, int M, int K, int N,
                             int ldg, int ldx, const float* residual, void* stream) {
  MX_CHECK_ARG(G && X && coef && WtPlanes && dX, "pw_dgrad_bnbwd_planes: null pointer");
  MX_CHECK_ARG(M > 0 && N > 0 && K > 0 && K % 32 == 0, "pw_dgrad_bnbwd_planes: bad extents M=%d N=%d K=%d (K must be a multiple of 32)", M, N, K);
  MX_CHECK_ARG(ldg % 4 == 0 && ldg >= K && ldx >= N, "pw_dgrad_bnbwd_planes: bad leading dimensions");
  MX_CHECK_ARG((((uintptr_t)G | (uintptr_t)X | (uintptr_t)coef | (uintptr_t)WtPlanes | (uintptr_t)dX) & 15) == 0, "pw_dgrad_bnbwd_planes: pointers must be 16-byte aligned");
  GemmArgs g{};
  g.a = MxOperand{G, coef, coef + K, X, MX_BNBWD, 1};
  g.b = MxOperand{reinterpret_cast<const float*>(WtPlanes), nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = dX; g.M = M; g.N = N; g.K = K; g.lda = ldg; g.ldb = (N + 127) & ~127; g.ldc = ldx;
  g.residual = residual;
  launch_nt_split3(g, 1, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_set_gemm_mode(int mode) {
  MX_CHECK_ARG(mode >= 0 && mode <= 2, "set_gemm_mode: mode %d (0 fp32 MFMA, 1 split for MFMA-bound shapes, 2 split everywhere)", mode);
  g_gemm_mode = mode;
  return MX_OK;
}

int mx_get_gemm_mode(void) { return g_gemm_mode; }

bool mx_wgrad_uses_split(int R, int Co, int Ci);      // wgrad.hip

// 1 when, in the current mode, the GEMM runs on the bf16 matrix pipe in split arithmetic (kind 0: mx_pw_fwd / the data
// gradient as C[M,N] = A[M,K] W[N,K]^T; kind 1: the weight gradient dW[M=Co, N=Ci] over K = R rows), else 0.
// For measurement code that prices each launch against the right peak (bench.py).
int mx_gemm_uses_split(int kind, int M, int N, int K) {
  if (kind == 0) return nt_uses_split(N, K) ? 1 : 0;
  return mx_wgrad_uses_split(K, M, N) ? 1 : 0;
}

// number of partial-statistics rows mx_pw_fwd writes for an [M, N] output
int mx_pw_fwd_parts(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return MX_EARG;
  return cdiv(M, 128);                  // every forward tile configuration (both kernel generations) has 128 rows
}

// C[M,N] = A'[M,K] * W[N,K]^T (+bias) (+residual) (relu); stats = partial column sum / sumsq rows.
// A' = prologue(A; a_mode, a_scale, a_shift, a_gate, rows_per_sample).

int mx_pw_fwd(const float* A, int a_mode, const float* a_scale, const float* a_shift, const float* a_gate,
              int rows_per_sample, const float* W, float* C, int M, int K, int N, int lda, int ldc,
              const float* bias, const float* residual, int relu, float* stats, void* stream) {
  GemmArgs g{};
  g.a = MxOperand{A, a_scale, a_shift, a_gate, a_mode, rows_per_sample};
  g.b = MxOperand{W, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = K; g.ldc = ldc;
  g.bias = bias; g.residual = residual; g.relu = relu; g.stats = stats;
  return gemm_common(L_NT, g, 1, (hipStream_t)stream);
}

// dX[M,N] = G[M,K] * W[K,N] (+residual): data gradient of a 1x1 conv whose weight is W[K=Cout, N=Cin].
int mx_pw_dgrad(const float* G, const float* W, float* dX, int M, int K, int N, int ldg, int ldx,
                const float* residual, void* stream) {
  GemmArgs g{};
  g.a = MxOperand{G, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.b = MxOperand{W, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = dX; g.M = M; g.N = N; g.K = K; g.lda = ldg; g.ldb = N; g.ldc = ldx;
  g.residual = residual;
  return gemm_common(L_NN, g, 1, (hipStream_t)stream);
}

// Data gradient with the BatchNorm backward apply folded into its operand load: dZ = c1*G + c2*X + c3 per channel (coef =
// [3][K]), dX[M,N] = dZ[M,K] * Wt[N,K]^T (+ residual); dZ [M, ldg] is ALSO written (the weight gradient reads it).
int mx_pw_dgrad_bnbwd(const float* G, const float* X, const float* coef, const float* Wt, float* dX, float* dZ, int M, int K, int N,
                      int ldg, int ldx, const float* residual, void* stream) {
  MX_CHECK_ARG(G && X && coef && Wt && dX && dZ, "pw_dgrad_bnbwd: null pointer");
  MX_CHECK_ARG(dZ != G && dZ != X, "pw_dgrad_bnbwd: dZ must not alias G or X (other N tiles still read them)");
  MX_CHECK_ARG((((uintptr_t)X | (uintptr_t)dZ | (uintptr_t)coef) & 15) == 0 && K % 4 == 0, "pw_dgrad_bnbwd: alignment");
  GemmArgs g{};
  g.a = MxOperand{G, coef, coef + K, X, MX_BNBWD, 1};
  g.b = MxOperand{Wt, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = dX; g.M = M; g.N = N; g.K = K; g.lda = ldg; g.ldb = K; g.ldc = ldx;
  g.residual = residual;
  g.a_out = dZ;
  return gemm_common(L_NT, g, 1, (hipStream_t)stream);
}

// bytes of scratch mx_pw_wgrad needs for this shape (0: a single reduction slice adds straight into dW)
long mx_pw_wgrad_ws(int R, int Co, int Ci) {
  if (R <= 0 || Co <= 0 || Ci <= 0 || Co % 4 || Ci % 4) return MX_EARG;
  int ks;
  const int splits = tn_splits(Co, Ci, R, &ks);
  return splits > 1 ? (long)splits * Co * Ci * 4 : 0;
}

// dW[Co,Ci] += G[R,Co]^T * X'[R,Ci] (dW must be zeroed or hold a running sum); the general kernel for the shapes
// mx_pw_wgrad_small / _tile do not take.  Deterministic: see gemm_common.
int mx_pw_wgrad(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                void* ws, long ws_bytes, void* stream) {
  const long need = mx_pw_wgrad_ws(R, Co, Ci);
  MX_CHECK_ARG(need >= 0, "pw_wgrad: bad extents R=%d Co=%d Ci=%d", R, Co, Ci);
  MX_CHECK_ARG(need == 0 || (ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0), "pw_wgrad: %ld bytes of scratch required (mx_pw_wgrad_ws)", need);
  GemmArgs g{};
  g.a = MxOperand{G, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.b = MxOperand{X, x_scale, x_shift, x_gate, x_mode, rows_per_sample};
  g.c = dW; g.M = Co; g.N = Ci; g.K = R; g.lda = ldg; g.ldb = ldx; g.ldc = Ci;
  g.ws = (float*)ws;
  return gemm_common(L_TN, g, 1, (hipStream_t)stream);
}

// Batched plain GEMMs for the PCM head (MuSCLe.py:213-223): layout 0 NT, 1 NN, 2 is not batched.
int mx_bgemm(int layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             long sa, long sb, long sc, int batch, int relu, void* stream) {
  MX_CHECK_ARG(layout == L_NT || layout == L_NN, "bgemm: layout must be 0 (NT) or 1 (NN)");
  GemmArgs g{};
  g.a = MxOperand{A, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.b = MxOperand{B, nullptr, nullptr, nullptr, MX_PLAIN, 1};
  g.c = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sa = sa; g.sb = sb; g.sc = sc; g.relu = relu;
  return gemm_common(layout, g, batch, (hipStream_t)stream);
}

}  // extern "C"
