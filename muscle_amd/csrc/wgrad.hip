// Weight gradient of the 1x1 convolutions with a SMALL output and a long pixel reduction (the first four stages of
// EfficientNet: dW[Co,Ci] = sum_r G[r,Co]^T X'[r,Ci] with Co*Ci <= ~40k and R = 100k .. 1.6M rows).
//
// These are HBM-bound: both operands should cross the fabric exactly once.  The tiled TN GEMM (gemm.hip) cuts the output
// into 128x32 tiles, so every row slab is read by several workgroups, and joins its reduction slices with fp32 atomics
// (run-to-run different sums): 237-285 us for Co x Ci = 288 x 48 at R = 401 408, whose operands take ~90 us to stream.
// Here ONE workgroup owns the WHOLE output: it walks a contiguous range of rows in 16-row slabs (both operands through
// LDS, double buffered), keeps all Co/16 x Ci/16 MFMA tiles in registers (v_mfma_f32_16x16x4_f32, split over its waves)
// and leaves one partial matrix; a second kernel adds the partials in a fixed order: deterministic, no atomics.
#include "common.h"

typedef float f32x4w __attribute__((ext_vector_type(4)));

struct WgArgs {
  const float* G;          // [R, ldg] upstream gradient (plain)
  MxOperand X;             // [R, ldx] layer input through the operand prologue
  float* part;             // [groups][Co*Ci]
  int R, Co, Ci, ldg, ldx;
  int rows_per_wg;         // multiple of 16
  int WCO, WCI;            // wave grid over (Co tiles, Ci tiles): WCO * WCI = waves
  int SG, SX;              // LDS row strides (floats), == 16 mod 32: the 4 rows of a fragment fall on disjoint banks
  const float* G2;         // GBN instantiations: the G operand is c1*G + c2*G2 + c3 per column (BatchNorm backward apply folded
  const float* gcoef;      //   into the load; gcoef = [3][Co]); G2 shares G's leading dimension
};

// GI / XI: 16-byte chunks of a G / X row per loader lane (a row is shared by NT/16 lanes); XMODE: operand prologue of X.
template <int TCO, int TCI, int NW, int GI, int XI, int XMODE, bool GBN = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 3 : 4) void wgrad_small_kernel(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RB = 16, NT = NW * 64, LPR = NT / RB;       // loader lanes per slab row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wco = wave % a.WCO, wci = wave / a.WCO;
  const int slab = RB * (a.SG + a.SX);
  const int gch = a.Co >> 2, xch = a.Ci >> 2;               // 16-byte chunks per row
  const long r_beg = (long)blockIdx.x * a.rows_per_wg;
  const long r_end = min((long)a.R, r_beg + a.rows_per_wg);
  const int lrow = tid / LPR, lc = tid - lrow * LPR;         // this lane's slab row and first chunk

  float4 rg[GI], rx[XI], gt[XMODE == MX_BNACT ? XI : 1], rg2[GBN ? GI : 1];
  auto load = [&](long r0) {
    const long r = r0 + lrow;
    const bool rin = r < r_end;
    const float* pg = a.G + r * a.ldg;
    const float* px = a.X.p + r * a.ldx;
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const int c = lc + LPR * i;
      rg[i] = (rin && c < gch) ? ld4(pg + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (GBN) rg2[GBN ? i : 0] = (rin && c < gch) ? ld4(a.G2 + r * a.ldg + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = lc + LPR * i;
      rx[i] = (rin && c < xch) ? ld4(px + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (XMODE == MX_BNACT)
        gt[i] = (rin && c < xch && a.X.rowp) ? ld4(a.X.rowp + (long)((unsigned)r / (unsigned)a.X.rps) * a.Ci + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
  };
  auto store = [&](float* buf, long r0) {
    const bool rin = r0 + lrow < r_end;
#pragma unroll
    for (int i = 0; i < GI; ++i) {
      const int c = lc + LPR * i;
      if (c < gch) {
        float4 v = rg[i];
        if (GBN && rin) {               // dZ = c1*G + c2*G2 + c3 (rows past the range stay zero: they must not add c3)
          const float4 k1 = ld4(a.gcoef + 4 * c), k2 = ld4(a.gcoef + a.Co + 4 * c), k3 = ld4(a.gcoef + 2 * a.Co + 4 * c);
          const float4 x = rg2[GBN ? i : 0];
          v = make_float4(k1.x * v.x + (k2.x * x.x + k3.x), k1.y * v.y + (k2.y * x.y + k3.y), k1.z * v.z + (k2.z * x.z + k3.z),
                          k1.w * v.w + (k2.w * x.w + k3.w));
        }
        st4(buf + lrow * a.SG + 4 * c, v);
      }
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = lc + LPR * i;
      if (c < xch) {
        float4 v = rx[i];
        if (XMODE != MX_PLAIN && rin) {
          const float4 sc = ld4(a.X.c1 + 4 * c), sh = ld4(a.X.c2 + 4 * c);
          v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
          if (XMODE == MX_BNACT) {
            const float4 g4 = gt[XMODE == MX_BNACT ? i : 0];
            v.x = swishf_(v.x) * g4.x; v.y = swishf_(v.y) * g4.y; v.z = swishf_(v.z) * g4.z; v.w = swishf_(v.w) * g4.w;
          }
        }
        st4(buf + RB * a.SG + lrow * a.SX + 4 * c, v);
      }
    }
  };
  // columns beyond Co / Ci (up to the wave grid's 16-column tiles) stay zero: fragments read them without a test
  for (int i = tid; i < 2 * slab; i += NT) smem[i] = 0.f;
  __syncthreads();

  f32x4w acc[TCO][TCI];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j) acc[i][j] = f32x4w{0.f, 0.f, 0.f, 0.f};

  const int ns = (int)((r_end - r_beg + RB - 1) / RB);
  if (ns > 0) {
    load(r_beg);
    store(smem, r_beg);
    if (ns > 1) load(r_beg + RB);
  }
  __syncthreads();
  for (int s = 0; s < ns; ++s) {
    const int cur = s & 1;
    if (s + 1 < ns) {
      store(smem + (cur ^ 1) * slab, r_beg + (long)(s + 1) * RB);
      if (s + 2 < ns) load(r_beg + (long)(s + 2) * RB);
    }
    const float* gs = smem + cur * slab + q * a.SG + 16 * wco + l15;
    const float* xs = smem + cur * slab + RB * a.SG + q * a.SX + 16 * wci + l15;
    const int gstep = 16 * a.WCO, xstep = 16 * a.WCI;
#pragma unroll
    for (int qd = 0; qd < RB / 4; ++qd) {
      float av[TCO], bv[TCI];
#pragma unroll
      for (int i = 0; i < TCO; ++i) av[i] = gs[(4 * qd) * a.SG + gstep * i];
#pragma unroll
      for (int j = 0; j < TCI; ++j) bv[j] = xs[(4 * qd) * a.SX + xstep * j];
#pragma unroll
      for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // partial matrix of this workgroup: acc[i][j][r] = dW[16 t_co + 4q + r][16 t_ci + l15]
  float* out = a.part + (long)blockIdx.x * a.Co * a.Ci;
#pragma unroll
  for (int i = 0; i < TCO; ++i) {
    const int tco = wco + a.WCO * i;
#pragma unroll
    for (int j = 0; j < TCI; ++j) {
      const int col = 16 * (wci + a.WCI * j) + l15;
      if (col < a.Ci) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * tco + 4 * q + r;
          if (row < a.Co) out[(long)row * a.Ci + col] = acc[i][j][r];
        }
      }
    }
  }
}

// dW[e] += sum_g part[g][e]: 16 lanes per 4 consecutive elements walk the partial matrices (g = lane, lane + 16, ...; four
// loads in flight each), then the 16 lane sums are added in a fixed tree: deterministic.
__global__ __launch_bounds__(256) void wgrad_parts_reduce_kernel(const float* __restrict__ part, int groups, int n, float* __restrict__ dW) {
  __shared__ float4 red[16][16];
  const int el = threadIdx.x & 15, gl = threadIdx.x >> 4;
  const int e4 = blockIdx.x * 16 + el;                       // float4 index
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
  if (4 * e4 < n) {
    const float* p = part + 4L * e4;
    int g = gl;
    for (; g + 48 < groups; g += 64) {
      const float4 a0 = ld4(p + (long)g * n), a1 = ld4(p + (long)(g + 16) * n), a2 = ld4(p + (long)(g + 32) * n), a3 = ld4(p + (long)(g + 48) * n);
      s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
      s1.x += a1.x; s1.y += a1.y; s1.z += a1.z; s1.w += a1.w;
      s2.x += a2.x; s2.y += a2.y; s2.z += a2.z; s2.w += a2.w;
      s3.x += a3.x; s3.y += a3.y; s3.z += a3.z; s3.w += a3.w;
    }
    for (; g < groups; g += 16) {
      const float4 a0 = ld4(p + (long)g * n);
      s0.x += a0.x; s0.y += a0.y; s0.z += a0.z; s0.w += a0.w;
    }
  }
  red[gl][el] = make_float4((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w));
  __syncthreads();
  if (gl == 0 && 4 * e4 < n) {
    float4 t = red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 v = red[k][el]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    float4 d = ld4(dW + 4L * e4);
    d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
    st4(dW + 4L * e4, d);
  }
}

static int wg_pad16(int c) {           // smallest stride >= c (rounded to 16) that is 16 mod 32
  int s = (c + 15) / 16 * 16;
  return (s % 32 == 16) ? s : s + 16;
}

struct WgPlan { int nw, wco, wci, tco, tci, groups, rows_per_wg, SG, SX; long lds_bytes; };

void mx_launch_parts_reduce(const float* part, int groups, int n, float* dW, hipStream_t st) {
  hipLaunchKernelGGL(wgrad_parts_reduce_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, part, groups, n, dW);
}

static bool wg_plan(int R, int Co, int Ci, WgPlan* p) {
  const int cot = cdiv(Co, 16), cit = cdiv(Ci, 16);
  static const int rmin = getenv("MX_WGRAD_SMALL_RMIN") ? atoi(getenv("MX_WGRAD_SMALL_RMIN")) : 65536;
  if (Co % 4 || Ci % 4 || (long)Co * Ci > 40960 || R < rmin) return false;
  // waves: 4, or 8 when the tiles of one wave would exceed 24 (96 accumulator registers)
  int best_nw = 0, best_wco = 0, best_cost = 1 << 30;
  for (int nw : {4, 8}) {
    for (int wco = 1; wco <= nw; wco *= 2) {
      const int wci = nw / wco;
      const int tco = cdiv(cot, wco), tci = cdiv(cit, wci);
      if (tco > 5 || tci > 5 || tco * tci > (nw == 4 ? 24 : 20)) continue;
      const int cost = tco * tci * nw;                        // issued MFMA slots per 4 rows
      if (cost < best_cost) { best_cost = cost; best_nw = nw; best_wco = wco; }
    }
    if (best_nw) break;                                       // 4 waves suffice
  }
  if (!best_nw) return false;
  p->nw = best_nw; p->wco = best_wco; p->wci = best_nw / best_wco;
  p->tco = cdiv(cot, p->wco); p->tci = cdiv(cit, p->wci);
  p->SG = wg_pad16(16 * p->wco * p->tco); p->SX = wg_pad16(16 * p->wci * p->tci);
  p->lds_bytes = 2L * 16 * (p->SG + p->SX) * 4;
  if (p->lds_bytes > 80 * 1024) return false;
  int per_cu = (int)(160 * 1024 / p->lds_bytes);
  const int cap = best_nw == 8 ? 2 : 3;
  if (per_cu > cap) per_cu = cap;
  int rows = cdiv(cdiv(R, 256 * per_cu), 16) * 16;
  if (rows < 64) rows = 64;
  p->rows_per_wg = rows;
  p->groups = cdiv(R, rows);
  return true;
}

template <int TCO, int TCI, int NW, int GI, int XI, int XMODE, bool GBN = false>
static void wg_launch_one(const WgArgs& a, const WgPlan& p, hipStream_t st) {
  static bool big_lds_ok = false;                      // > 64 KB of dynamic LDS needs the attribute, once per kernel
  if (p.lds_bytes > 64 * 1024 && !big_lds_ok) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_small_kernel<TCO, TCI, NW, GI, XI, XMODE, GBN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    big_lds_ok = true;
  }
  hipLaunchKernelGGL((wgrad_small_kernel<TCO, TCI, NW, GI, XI, XMODE, GBN>), dim3(p.groups), dim3(NW * 64), p.lds_bytes, st, a);
}

// The instantiations that exist: the (Co, Ci, prologue) combinations of EfficientNet-B7's first four stages (project convs
// read d_raw through BN1 + SiLU + gate, expand convs and the stem read plain inputs).  Anything else falls back to the
// tiled kernel of gemm.hip - including stage 3 (480 x 80 / 80 x 480 at 100 352 rows): its 150 tiles need 8 waves of 20
// tiles each, which at the 128-register bound of two 512-thread workgroups per CU spilled 160-300 registers.            tiles/wave  waves  G chunks  X chunks  prologue       B7 layer
#define WG_TABLE(V)                                                                                             \
  V(2, 1, 4, 1, 1, MX_BNACT)  /* 32 x 64              project of block 0                                    */ \
  V(3, 3, 4, 1, 3, MX_BNACT)  /* 48 x 192             project of block 4                                    */ \
  V(3, 5, 4, 1, 5, MX_BNACT)  /* 48 x 288             project of stage 2                                    */ \
  V(3, 2, 4, 3, 1, MX_PLAIN)  /* 192 x 32             expand of block 4                                     */ \
  V(5, 3, 4, 5, 1, MX_PLAIN)  /* 288 x 48             expand of stage 2                                     */ \
  V(2, 1, 4, 1, 1, MX_PLAIN)  /* 64 x 28              stem                                                  */ \
  V(1, 1, 4, 1, 1, MX_BNACT)  /* 32 x 32              project of blocks 1-3                                 */ \
  V(1, 1, 4, 1, 1, MX_PLAIN)  /* 32 x 32              (plain)                                               */

static bool wg_dispatch(const WgArgs& a, const WgPlan& p, hipStream_t st, bool launch) {
  const int lpr = p.nw * 64 / 16;
  const int gi = cdiv(a.Co / 4, lpr), xi = cdiv(a.Ci / 4, lpr);
#define WG_TRY(TCO, TCI, NW, GI, XI, MODE)                                                             \
  if (p.tco == TCO && p.tci == TCI && p.nw == NW && gi == GI && xi == XI && a.X.mode == MODE) {        \
    if (launch) wg_launch_one<TCO, TCI, NW, GI, XI, MODE>(a, p, st);                                   \
    return true;                                                                                       \
  }
  if (a.G2) {                       // folded BatchNorm backward apply on G: the expand convs of block 4 and of stage 2 (plain X)
#define WG_TRY_GBN(TCO, TCI, NW, GI, XI)                                                               \
    if (p.tco == TCO && p.tci == TCI && p.nw == NW && gi == GI && xi == XI && a.X.mode == MX_PLAIN) {  \
      if (launch) wg_launch_one<TCO, TCI, NW, GI, XI, MX_PLAIN, true>(a, p, st);                       \
      return true;                                                                                     \
    }
    WG_TRY_GBN(3, 2, 4, 3, 1)       /* 192 x 32 */
    WG_TRY_GBN(5, 3, 4, 5, 1)       /* 288 x 48 */
#undef WG_TRY_GBN
    return false;
  }
  WG_TABLE(WG_TRY)
#undef WG_TRY
  return false;
}

// =====================================================================================================================
// Tiled, deterministic weight gradient for the LARGE outputs (stages 4-7: Co*Ci up to 3840 x 640).
//
// dW is cut into TCO x TCI tiles (128/64 wide), the pixel rows into `groups` contiguous ranges; workgroup (tile, group)
// walks its rows in 16-row slabs through k-major LDS and leaves its partial tile in part[group]; the reduce kernel adds
// the groups in a fixed order (no atomics: run-to-run identical, and no read-modify-write traffic).  k-major LDS means a
// slab row is a plain copy of 512 contiguous bytes of the operand row, and ONE ds_read_b128 per operand and 4 reduction
// rows feeds 16 v_mfma_f32_16x16x4_f32: lane (l15, q) reads columns 4 l15 .. 4 l15 + 3 of row 4 qd + q, value e goes to
// the MFMA of sub-tile e, whose row/column l15 therefore IS column 4 l15 + e of the wave's 64 (a fixed permutation of the
// output coordinates, undone when the accumulators are stored: each lane owns runs of four consecutive Ci).  Row stride
// 128 floats (= 0 mod 64 banks: the 16 lanes of a ds_read_b128 group differ in l15 only) or 96 for the 64-wide operand.
// Workgroup ids are XCD-aware: all tiles of one row group run back to back on ONE XCD, so a slab crosses the fabric once
// (per-XCD L2) instead of once per tile - the tiled TN kernel of gemm.hip re-read its operands 2.8x (profiles/traffic).
struct WtArgs {
  const float* G;
  MxOperand X;
  float* part;             // [groups][Co*Ci], or dW itself when groups == 1 (then accumulated in place)
  int R, Co, Ci, ldg, ldx;
  int rows_per_group;      // multiple of 16
  int groups, tiles_co, tiles_ci;
  int accumulate;          // groups == 1: part == dW, add instead of store
  int total;               // number of workgroup ids (8 * ceil(groups / 8) * tiles)
  int order;               // 0: XCD-aware (tiles of a group on one XCD); 1: group-major; 2: tile-major
  const float* G2;         // wgrad_split_kernel<*, true>: the G operand is c1*G + c2*G2 + c3 per column (BatchNorm backward apply
  const float* gcoef;      //   folded into the load; gcoef = [3][Co]); G2 shares G's leading dimension
  float* dz;               // wgrad_split_ws_kernel<true> only: if set, that operand is also WRITTEN here ([R, ldg]; by the ci-tile-0 items)
};

template <int TE, int TF, int XMODE>
__global__ __launch_bounds__(256, (TE * TF >= 16 && XMODE == MX_BNACT) ? 3 : 4) void wgrad_tile_kernel(WtArgs a) {
  constexpr int TCO = 32 * TE, TCI = 32 * TF;                 // 2 x 2 waves, wave tile 16 TE x 16 TF
  constexpr int SG = (TE == 4) ? 128 : 96, SX = (TF == 4) ? 128 : 96;
  constexpr int SLAB = 16 * (SG + SX);
  constexpr int GP = (16 * TCO / 4) / 256 > 0 ? (16 * TCO / 4) / 256 : 1;   // float4 per thread per slab
  constexpr int XP = (16 * TCI / 4) / 256 > 0 ? (16 * TCI / 4) / 256 : 1;
  __shared__ __attribute__((aligned(16))) float smem[2 * SLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int wco = wave >> 1, wci = wave & 1;
  // XCD-aware ids: L % 8 is the XCD (round-robin dispatch); the tiles of one group are consecutive ON that XCD.
  // a.total > gridDim.x: persistent workgroups (a multiple of 8 of them) walk the ids L, L + grid, ...: same XCD each time.
  const int tiles = a.tiles_co * a.tiles_ci;
  for (int L = blockIdx.x; L < a.total; L += gridDim.x) {
  const int xcd = L & 7, j = L >> 3;
  int group = (j / tiles) * 8 + xcd, tile = j % tiles;
  if (a.order == 1) { group = L / tiles; tile = L % tiles; }
  else if (a.order == 2) { const int g8 = 8 * ((a.groups + 7) / 8); group = L % g8; tile = L / g8; }
  if (group >= a.groups) continue;
  const int co0 = (tile / a.tiles_ci) * TCO, ci0 = (tile % a.tiles_ci) * TCI;
  const long r_beg = (long)group * a.rows_per_group;
  const long r_end = min((long)a.R, r_beg + a.rows_per_group);

  float4 rg[GP], rx[XP], gt[XMODE == MX_BNACT ? XP : 1];
  auto load = [&](long r0) {
#pragma unroll
    for (int i = 0; i < GP; ++i) {
      const int idx = tid + 256 * i, row = idx / (TCO / 4), c = idx % (TCO / 4);
      const long r = r0 + row;
      rg[i] = (idx < 16 * TCO / 4 && r < r_end && co0 + 4 * c < a.Co) ? ld4(a.G + r * a.ldg + co0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int idx = tid + 256 * i, row = idx / (TCI / 4), c = idx % (TCI / 4);
      const long r = r0 + row;
      const bool ok = idx < 16 * TCI / 4 && r < r_end && ci0 + 4 * c < a.Ci;
      rx[i] = ok ? ld4(a.X.p + r * a.ldx + ci0 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (XMODE == MX_BNACT) gt[i] = (ok && a.X.rowp) ? ld4(a.X.rowp + (long)((unsigned)r / (unsigned)a.X.rps) * a.Ci + ci0 + 4 * c) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
  };
  auto store = [&](float* buf, long r0) {
#pragma unroll
    for (int i = 0; i < GP; ++i) {
      const int idx = tid + 256 * i, row = idx / (TCO / 4), c = idx % (TCO / 4);
      if (idx < 16 * TCO / 4) st4(buf + row * SG + 4 * c, rg[i]);
    }
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int idx = tid + 256 * i, row = idx / (TCI / 4), c = idx % (TCI / 4);
      if (idx < 16 * TCI / 4) {
        float4 v = rx[i];
        if (XMODE != MX_PLAIN && r0 + row < r_end && ci0 + 4 * c < a.Ci) {
          const float4 sc = ld4(a.X.c1 + ci0 + 4 * c), sh = ld4(a.X.c2 + ci0 + 4 * c);
          v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
          if (XMODE == MX_BNACT) {
            const float4 g4 = gt[XMODE == MX_BNACT ? i : 0];
            v.x = swishf_(v.x) * g4.x; v.y = swishf_(v.y) * g4.y; v.z = swishf_(v.z) * g4.z; v.w = swishf_(v.w) * g4.w;
          }
        }
        st4(buf + 16 * SG + row * SX + 4 * c, v);
      }
    }
  };

  f32x4w acc[TE][TF];
#pragma unroll
  for (int e = 0; e < TE; ++e)
#pragma unroll
    for (int f = 0; f < TF; ++f) acc[e][f] = f32x4w{0.f, 0.f, 0.f, 0.f};

  const int ns = (int)((r_end - r_beg + 15) / 16);
  if (ns > 0) {
    load(r_beg);
    store(smem, r_beg);
    if (ns > 1) load(r_beg + 16);
  }
  __syncthreads();
  for (int s = 0; s < ns; ++s) {
    const int cur = s & 1;
    if (s + 1 < ns) {
      store(smem + (cur ^ 1) * SLAB, r_beg + (long)(s + 1) * 16);
      if (s + 2 < ns) load(r_beg + (long)(s + 2) * 16);
    }
    const float* gs = smem + cur * SLAB + q * SG + 16 * TE * wco + TE * l15;
    const float* xs = smem + cur * SLAB + 16 * SG + q * SX + 16 * TF * wci + TF * l15;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      float av[TE], bv[TF];
      if (TE == 4) { const float4 t = ld4(gs + 4 * qd * SG); av[0] = t.x; av[1] = t.y; av[2] = t.z; av[TE - 1] = t.w; }
      else { const float2 t = *reinterpret_cast<const float2*>(gs + 4 * qd * SG); av[0] = t.x; av[TE - 1] = t.y; }
      if (TF == 4) { const float4 t = ld4(xs + 4 * qd * SX); bv[0] = t.x; bv[1] = t.y; bv[2] = t.z; bv[TF - 1] = t.w; }
      else { const float2 t = *reinterpret_cast<const float2*>(xs + 4 * qd * SX); bv[0] = t.x; bv[TF - 1] = t.y; }
#pragma unroll
      for (int e = 0; e < TE; ++e)
#pragma unroll
        for (int f = 0; f < TF; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[f], acc[e][f], 0, 0, 0);
    }
    __syncthreads();
  }
  // acc[e][f][r] = dW[co0 + 16 TE wco + TE (4 q + r) + e][ci0 + 16 TF wci + TF l15 + f]
  float* out = a.part + (a.accumulate ? 0 : (long)group * a.Co * a.Ci);
#pragma unroll
  for (int e = 0; e < TE; ++e)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + 16 * TE * wco + TE * (4 * q + r) + e;
      const int ci = ci0 + 16 * TF * wci + TF * l15;
      if (co >= a.Co || ci >= a.Ci) continue;                 // Ci % 4 == 0 and TF | 4: a lane's run is in or out as a whole
      float* o = out + (long)co * a.Ci + ci;
      if (TF == 4) {
        float4 v = make_float4(acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][TF - 1][r]);
        if (a.accumulate) { const float4 d = ld4(o); v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w; }
        st4(o, v);
      } else {
        float2 v = make_float2(acc[e][0][r], acc[e][TF - 1][r]);
        if (a.accumulate) { const float2 d = *reinterpret_cast<const float2*>(o); v.x += d.x; v.y += d.y; }
        *reinterpret_cast<float2*>(o) = v;
      }
    }
  __syncthreads();                                            // the next work item overwrites the LDS slabs
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The tiled weight gradient in the opt-in "split" arithmetic (mx_set_gemm_mode >= 1; see gemm.hip: fp32 operands split
// exactly into three bf16 terms, six of nine products on v_mfma_f32_32x32x16_bf16, fp32 accumulation).  The reduction
// runs over pixel ROWS, so the bf16 fragments (8 consecutive k of one output row / column) are columns of the operand
// slabs: a thread loads 8 rows x 4 columns, splits in registers and writes, per column, 8 bf16 along k as one 16-byte
// chunk - the transposition costs nothing.  LDS image per operand and plane: [128 columns][32 k] bf16 (64-byte rows, the
// four 16-byte chunks of a row XOR-ed with (column >> 2) & 3: conflict-free ds_read_b128).  128 x 128 tile, 32-row slabs,
// one LDS stage (48 KB: 3 workgroups per CU) + register prefetch.  Same (tile, group) decomposition, partial tiles and
// fixed-order reduce as wgrad_tile_kernel.
// Round 4 built a second structure in the lab and measured it slower (profiles/r04_wgrad_split2_lab.txt): 16-row slabs, two LDS
// stages, ONE barrier per slab, a thread owning 8 rows x 2 columns (8-byte loads), the split of slab s + 1 in the same instruction
// stream as the MFMAs of slab s.  Bit-identical results; at 2 waves per SIMD (175 registers) level with this kernel (317 vs 317 us at
// 2304 x 384) to 8 % slower (3840 x 640), at 3 waves per SIMD (fragments fetched plane by plane: 144 registers) 10-15 % slower - the
// barrier now comes every 768 MFMA cycles and the fragment reads sit in front of their MFMAs.  This kernel stays.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ void wsplit3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0) & 0xffff0000u, u1 = __float_as_uint(x1) & 0xffff0000u;
  const float r0 = x0 - __uint_as_float(u0), r1 = x1 - __uint_as_float(u1);
  const unsigned v0 = __float_as_uint(r0) & 0xffff0000u, v1 = __float_as_uint(r1) & 0xffff0000u;
  const float s0 = r0 - __uint_as_float(v0), s1 = r1 - __uint_as_float(v1);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}

// GBN: the G operand is the BatchNorm backward apply c1*G + c2*G2 + c3 (per column), formed from two tensors by the loader threads
// (their columns are fixed, so the three coefficient vectors are loaded once): the expand convolution's weight gradient reads
// (G, e_raw) instead of a materialised dZ - see gemm_nt_split3_kernel<*, MX_BNBWD> for the data-gradient half.
template <int XMODE, bool GBN>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WtArgs a) {
  constexpr int PLANE = 128 * 64;                           // bytes: [128 columns][32 k] bf16
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * PLANE];     // G h/m/l, X h/m/l
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hf = lane >> 5;
  const int wco = wave >> 1, wci = wave & 1;
  const int tiles = a.tiles_co * a.tiles_ci;
  const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
  const int group = (j / tiles) * 8 + xcd, tile = j % tiles;
  if (group >= a.groups) return;
  const int co0 = (tile / a.tiles_ci) * 128, ci0 = (tile % a.tiles_ci) * 128;
  const long r_beg = (long)group * a.rows_per_group;
  const long r_end = min((long)a.R, r_beg + a.rows_per_group);

  // loader: threads 0-127 move G, 128-255 move X; thread = (4-column chunk c of 32, row group rg of 4): rows 8 rg .. 8 rg + 7
  const bool isx = tid >= 128;
  const int lt = tid & 127, c = lt & 31, rg = lt >> 5;
  const int col0 = (isx ? ci0 : co0) + 4 * c;
  const bool colok = col0 < (isx ? a.Ci : a.Co);
  const float* base = isx ? a.X.p : a.G;
  const int ld = isx ? a.ldx : a.ldg;
  float4 rv[8], gt[(XMODE == MX_BNACT || GBN) ? 8 : 1];      // gt: the SE gate rows (X side) / the second tensor (G side)
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gc1 = sc, gc2 = sh, gc3 = sh;
  if (XMODE != MX_PLAIN && isx && colok) { sc = ld4(a.X.c1 + col0); sh = ld4(a.X.c2 + col0); }
  if (GBN && !isx && colok) { gc1 = ld4(a.gcoef + col0); gc2 = ld4(a.gcoef + a.Co + col0); gc3 = ld4(a.gcoef + 2 * a.Co + col0); }
  auto load = [&](long r0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long r = r0 + 8 * rg + i;
      const bool ok = colok && r < r_end;
      rv[i] = ok ? ld4(base + r * ld + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (XMODE == MX_BNACT && (!GBN || isx)) gt[i] = (ok && isx && a.X.rowp) ? ld4(a.X.rowp + (long)((unsigned)r / (unsigned)a.X.rps) * a.Ci + col0) : make_float4(1.f, 1.f, 1.f, 1.f);
      if (GBN && !isx) gt[i] = ok ? ld4(a.G2 + r * ld + col0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store = [&](long r0) {
    if (GBN && !isx) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!(colok && r0 + 8 * rg + i < r_end)) continue;      // rows past the group's end stay zero
        const float4 g4 = rv[i], x4 = gt[GBN ? i : 0];
        rv[i] = make_float4(gc1.x * g4.x + (gc2.x * x4.x + gc3.x), gc1.y * g4.y + (gc2.y * x4.y + gc3.y),
                            gc1.z * g4.z + (gc2.z * x4.z + gc3.z), gc1.w * g4.w + (gc2.w * x4.w + gc3.w));
      }
    }
    if (XMODE != MX_PLAIN && isx) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!(colok && r0 + 8 * rg + i < r_end)) continue;
        float4 v = rv[i];
        v.x = sc.x * v.x + sh.x; v.y = sc.y * v.y + sh.y; v.z = sc.z * v.z + sh.z; v.w = sc.w * v.w + sh.w;
        if (XMODE == MX_BNACT) {
          const float4 g4 = gt[XMODE == MX_BNACT ? i : 0];
          v.x = swishf_(v.x) * g4.x; v.y = swishf_(v.y) * g4.y; v.z = swishf_(v.z) * g4.z; v.w = swishf_(v.w) * g4.w;
        }
        rv[i] = v;
      }
    }
    unsigned char* pl = smem + (isx ? 3 * PLANE : 0);
#pragma unroll
    for (int jc = 0; jc < 4; ++jc) {                          // column 4 c + jc: 8 values along k -> one 16-byte chunk per plane
      const int col = 4 * c + jc;
      unsigned h[4], m[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x0 = jc == 0 ? rv[2 * q].x : jc == 1 ? rv[2 * q].y : jc == 2 ? rv[2 * q].z : rv[2 * q].w;
        const float x1 = jc == 0 ? rv[2 * q + 1].x : jc == 1 ? rv[2 * q + 1].y : jc == 2 ? rv[2 * q + 1].z : rv[2 * q + 1].w;
        wsplit3_pair(x0, x1, h[q], m[q], l[q]);
      }
      const int off = col * 64 + ((rg ^ ((col >> 2) & 3)) << 4);
      *reinterpret_cast<uint4*>(pl + off) = make_uint4(h[0], h[1], h[2], h[3]);
      *reinterpret_cast<uint4*>(pl + PLANE + off) = make_uint4(m[0], m[1], m[2], m[3]);
      *reinterpret_cast<uint4*>(pl + 2 * PLANE + off) = make_uint4(l[0], l[1], l[2], l[3]);
    }
  };

  wf32x16 acc[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;

  const int ns = (int)((r_end - r_beg + 31) / 32);
  if (ns > 0) load(r_beg);
  for (int s = 0; s < ns; ++s) {
    __syncthreads();                                          // the previous slab's fragments have been read
    store(r_beg + (long)s * 32);
    if (s + 1 < ns) load(r_beg + (long)(s + 1) * 32);
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                           // two 16-row MFMA steps per slab
      wbf16x8 gv[2][3], xv[2][3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int cg = wco * 64 + 32 * e + l31, cx = wci * 64 + 32 * e + l31;
          gv[e][p] = *reinterpret_cast<const wbf16x8*>(smem + p * PLANE + cg * 64 + (((2 * ks + hf) ^ ((cg >> 2) & 3)) << 4));
          xv[e][p] = *reinterpret_cast<const wbf16x8*>(smem + (3 + p) * PLANE + cx * 64 + (((2 * ks + hf) ^ ((cx >> 2) & 3)) << 4));
        }
      constexpr int PG[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int f = 0; f < 2; ++f)
            acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gv[e][PG[t]], xv[f][PX[t]], acc[e][f], 0, 0, 0);
    }
  }
  // acc[e][f][4 g + r] = dW[co0 + 64 wco + 32 e + 8 g + 4 hf + r][ci0 + 64 wci + 32 f + l31]
  float* out = a.part + (a.accumulate ? 0 : (long)group * a.Co * a.Ci);
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ci = ci0 + 64 * wci + 32 * f + l31;
      if (ci >= a.Ci) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 64 * wco + 32 * e + 8 * (r >> 2) + 4 * hf + (r & 3);
        if (co >= a.Co) continue;
        float* o = out + (long)co * a.Ci + ci;
        *o = a.accumulate ? *o + acc[e][f][r] : acc[e][f][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the split weight gradient as ONE software-pipelined instruction stream (plain operands; the prologue modes keep
// the kernel above).  Same tile, same loader mapping, same LDS image, same MFMA order as wgrad_split_kernel - the results
// are bit-identical - but:
//   * two LDS stages (96 KB, one workgroup = one wave per SIMD per CU) and ONE barrier per 32-row slab;
//   * the fragments of a 16-row step are read one step AHEAD of their MFMAs (two fragment register sets), so no ds_read
//     latency sits between a barrier and the matrix pipe; the split + plane stores of slab s+1 are placed BETWEEN the MFMAs of
//     slab s, ~4 vector instructions per MFMA gap (an MFMA holds the vector issue for 8 of its 32 cycles,
//     MI355X_MICROARCH.md 'vector-instruction ISSUE cost'); the raw rows of slab s+2 are in flight meanwhile (two raw sets);
//   * every global load is a buffer load through a descriptor that covers exactly the group's rows: rows past the group's
//     end and the columns past the matrix read as zeros by the range check - no compares, no branches, so the slab is one
//     basic block the scheduler can order.
// The first structure spent ~3500 cycles per slab and CU with three workgroups resident (MFMA work: 1536): a split/store
// phase, a barrier, and an MFMA phase whose ds_reads were waited for right in front of their MFMAs.
#ifndef WPIPE_PF
#define WPIPE_PF 3         // raw row sets: slab s + WPIPE_PF is requested while slab s is multiplied (bytes in flight per CU = 32 KB x (WPIPE_PF - 1))
#endif
#ifndef WPIPE_KNOCK
#define WPIPE_KNOCK 0      // diagnostic builds of tools/hip/gemm_lab only: 1 no split, 2 no split / plane stores, 3 no MFMA, 4 no fragment reads, 5 no global traffic, 6 split but no plane stores
#endif

static __device__ __forceinline__ __amdgpu_buffer_rsrc_t wbuf_rsrc(const float* base, long bytes) {
  // wave-uniform by construction (blockIdx / wave-id only): the readfirstlanes make that provable, no waterfall loops
  const unsigned long long p = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  const unsigned n = __builtin_amdgcn_readfirstlane((unsigned)bytes);
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, n, 0x00020000);
}

#define WPIPE_MFMA(GV, XV, T, EF)                                                                                              \
  acc[(EF) >> 1][(EF) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(GV[(EF) >> 1][PG[T]], XV[(EF) & 1][PX[T]], acc[(EF) >> 1][(EF) & 1], 0, 0, 0)

#ifdef WPIPE_STAMPS        // diagnostic builds of tools/hip/gemm_lab only: shader-clock stamps per workgroup (start, loop start, loop end, end, 2 x realtime)
__device__ unsigned long long* wpipe_stamps;
#define WPIPE_STAMP(slot) do { if (wpipe_stamps && threadIdx.x == 0) { wpipe_stamps[blockIdx.x * 8l + (slot)] = __builtin_amdgcn_s_memtime(); \
    if ((slot) == 0 || (slot) == 3) wpipe_stamps[blockIdx.x * 8l + 4 + ((slot) == 3)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define WPIPE_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(256, 1) void wgrad_split_pipe_kernel(WtArgs a) {
  // LDS image of one plane: [4 k-groups][4 column classes (col & 3)][36 = 32 columns (col >> 2) + 4 pad] chunks of 16 bytes (8 k of one
  // column).  A loader pass (16 lanes = 16 consecutive 4-column chunks, one column class) writes 16 consecutive chunks; a fragment
  // pass (16 consecutive columns of one k-group) reads chunks 36 b + a + 4 m (b = col & 3, a = (col >> 2) & 3): 16 different bank groups
  // either way.  The first kernel's [128 columns][64 bytes] image is conflict-free for the reads only: its plane stores hit 4 bank
  // groups per pass (4-way), which made the LDS, not the matrix pipe, the busiest unit (profiles/r05_wgrad_pipe_lab.txt).
  constexpr int PLANE = 4 * 144 * 16, STAGE = 6 * PLANE;    // G h/m/l, X h/m/l per stage
  extern __shared__ __attribute__((aligned(16))) unsigned char wp_smem[];      // 2 * STAGE
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform facts in SGPRs
  const int l31 = lane & 31, hf = lane >> 5;
  const int wco = wave >> 1, wci = wave & 1;
  const int tiles = a.tiles_co * a.tiles_ci;
  const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
  const int group = (j / tiles) * 8 + xcd, tile = j % tiles;
  if (group >= a.groups) return;
  const int co0 = (tile / a.tiles_ci) * 128, ci0 = (tile % a.tiles_ci) * 128;
  const long r_beg = (long)group * a.rows_per_group;
  const long r_end = min((long)a.R, r_beg + a.rows_per_group);
  const int rows = (int)(r_end - r_beg);

  // loader: waves 0-1 move G, 2-3 move X; thread = (4-column chunk c of 32, row group rg of 4): rows 8 rg .. 8 rg + 7 of the slab
  const bool isx = wave >= 2;
  const int lt = tid & 127, c = lt & 31, rg = lt >> 5;
  const int col0 = (isx ? ci0 : co0) + 4 * c;
  const bool colok = col0 < (isx ? a.Ci : a.Co);
  const int ld = __builtin_amdgcn_readfirstlane(isx ? a.ldx : a.ldg);
  const __amdgpu_buffer_rsrc_t rs = wbuf_rsrc((isx ? a.X.p : a.G) + r_beg * ld, WPIPE_KNOCK == 5 ? 0 : (long)rows * ld * 4);
  const unsigned vbase = colok ? (unsigned)((8 * rg * ld + col0) * 4) : 0x7f000000u;      // past every record: reads as zero
  const unsigned slab_bytes = (unsigned)(32 * ld * 4), row_bytes = (unsigned)(ld * 4);

  typedef float wf4 __attribute__((ext_vector_type(4)));
  wf4 raw[WPIPE_PF][8];
  auto gload = [&](wf4 (&rv)[8], int s) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      rv[i] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + (unsigned)i * row_bytes, (unsigned)s * slab_bytes, 0));
  };
  // split of column 4 c + jc, row pair q (rows 2 q, 2 q + 1 of the thread's eight): slices A / B / C of 4 / 4 / 3 vector instructions
  unsigned sh_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, sm_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, sl_[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // two columns' planes
  float r0_, r1_, s0_, s1_;
  unsigned u0_, u1_, v0_, v1_;
  unsigned char* const wbase = wp_smem + (isx ? 3 * PLANE : 0);
  const int woff0 = (rg * 144 + c) * 16;                     // + 36 * 16 per column class

#define WPIPE_X(RV, JC, Q, ROW) ((JC) == 0 ? RV[2 * (Q) + (ROW)][0] : (JC) == 1 ? RV[2 * (Q) + (ROW)][1] : (JC) == 2 ? RV[2 * (Q) + (ROW)][2] : RV[2 * (Q) + (ROW)][3])
#define WPIPE_SLICE_A(RV, JC, Q)                                                              \
  { const float x0 = WPIPE_X(RV, JC, Q, 0), x1 = WPIPE_X(RV, JC, Q, 1);                       \
    u0_ = __float_as_uint(x0) & 0xffff0000u; u1_ = __float_as_uint(x1) & 0xffff0000u;        \
    r0_ = x0 - __uint_as_float(u0_); r1_ = x1 - __uint_as_float(u1_); }
#define WPIPE_SLICE_B()                                                                       \
  { v0_ = __float_as_uint(r0_) & 0xffff0000u; v1_ = __float_as_uint(r1_) & 0xffff0000u;      \
    s0_ = r0_ - __uint_as_float(v0_); s1_ = r1_ - __uint_as_float(v1_); }
#define WPIPE_SLICE_C(B, Q)                                                                   \
  { sh_[B][Q] = __builtin_amdgcn_perm(u1_, u0_, 0x07060302u); sm_[B][Q] = __builtin_amdgcn_perm(v1_, v0_, 0x07060302u); \
    sl_[B][Q] = __builtin_amdgcn_perm(__float_as_uint(s1_), __float_as_uint(s0_), 0x07060302u); }
#define WPIPE_WRITE(STG, JC)                                                                                  \
  { unsigned char* o_ = wbase + (STG) * STAGE + woff0 + (JC) * 576;                                                    \
    *reinterpret_cast<uint4*>(o_) = make_uint4(sh_[0][0], sh_[0][1], sh_[0][2], sh_[0][3]);                   \
    *reinterpret_cast<uint4*>(o_ + PLANE) = make_uint4(sm_[0][0], sm_[0][1], sm_[0][2], sm_[0][3]);           \
    *reinterpret_cast<uint4*>(o_ + 2 * PLANE) = make_uint4(sl_[0][0], sl_[0][1], sl_[0][2], sl_[0][3]); }

  wf32x16 acc[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
  wbf16x8 gv0[2][3], xv0[2][3], gv1[2][3], xv1[2][3];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int i = 0; i < 8; ++i) { gv1[e][p][i] = (__bf16)0.f; xv1[e][p][i] = (__bf16)0.f; }
  const int fro = (hf * 144 + (l31 & 3) * 36 + (l31 >> 2)) * 16;         // this lane's chunk of k-group hf, column l31 of a 32-column block
  const int cgo = fro + wco * 256, cxo = fro + 3 * PLANE + wci * 256;       // + 128 per 32-column block e, + 2 * 144 * 16 per 16-row step
  auto frd = [&](const unsigned char* st, int ks, wbf16x8 (&gv)[2][3], wbf16x8 (&xv)[2][3]) {
    if (WPIPE_KNOCK == 4) return;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        gv[e][p] = *reinterpret_cast<const wbf16x8*>(st + cgo + p * PLANE + e * 128 + ks * 4608);
        xv[e][p] = *reinterpret_cast<const wbf16x8*>(st + cxo + p * PLANE + e * 128 + ks * 4608);
      }
  };
  constexpr int PG[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};

  const int ns = (rows + 31) / 32;
  WPIPE_STAMP(0);
  // prologue: slab 0 split into stage 0, slab 1 in flight
  gload(raw[0], 0);
  gload(raw[1], 1);
  if (WPIPE_PF >= 3) gload(raw[WPIPE_PF >= 3 ? 2 : 0], 2);
  if (WPIPE_PF >= 4) gload(raw[WPIPE_PF >= 4 ? 3 : 0], 3);
#pragma unroll
  for (int jc = 0; jc < 4; ++jc) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { WPIPE_SLICE_A(raw[0], jc, q); WPIPE_SLICE_B(); WPIPE_SLICE_C(0, q); }
    WPIPE_WRITE(0, jc);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);      /* lgkmcnt(0), as a builtin: the compiler's own wait counting sees it */
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // One slab: MFMAs of (slab s - 1, second step) from set 1, then of (slab s, first step) from set 0; between them the split of the raw
  // set RB (slab s + 1) into the other stage.  48 MFMAs, 32 row pairs: a pair's three slices go behind three consecutive MFMAs,
  // a column's three plane stores behind its fourth pair.
  // Every gap (the 24 cycles an MFMA leaves the vector issue free) gets one slice of the split (3-4 vector instructions) and at most one
  // memory instruction: the 12 fragment reads of the NEXT 16-row step in gaps 0-11 (consumption order), 4 of the 8 raw-row requests of
  // slab s + WPIPE_PF in gaps 13 / 16 / 19 / 22, and the three plane stores of a finished column two gaps apart behind it (second register
  // set, so the next column's split goes on).  Bursts cost: 8 x 1 KB requests in a row hold the CU's one address unit for ~130 cycles, a
  // 16-byte LDS store takes its wave's issue for >= 13 (MI355X_MICROARCH.md, LDS table) - neither fits into one gap.
#define WPIPE_FRD1(ST, KS, R, GVT, XVT)                                                                       \
  { const int g_ = (R) >> 2, e_ = (R) & 1;                                                                     \
    if (WPIPE_KNOCK != 4) {                                                                                    \
      if ((((R) >> 1) & 1) == 0) GVT[e_][g_ == 0 ? 0 : g_ == 1 ? 2 : 1] = *reinterpret_cast<const wbf16x8*>((ST) + cgo + (g_ == 0 ? 0 : g_ == 1 ? 2 : 1) * PLANE + e_ * 128 + (KS) * 4608); \
      else XVT[e_][g_ == 0 ? 2 : g_ == 1 ? 0 : 1] = *reinterpret_cast<const wbf16x8*>((ST) + cxo + (g_ == 0 ? 2 : g_ == 1 ? 0 : 1) * PLANE + e_ * 128 + (KS) * 4608); } }
#define WPIPE_GLD1(RVT, I, S)                                                                                 \
  RVT[I] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + (unsigned)(I) * row_bytes, (unsigned)(S) * slab_bytes, 0));
#define WPIPE_WRITE1(STG, JC, P)                                                                              \
  if (WPIPE_KNOCK == 6) { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) asm volatile("" :: "v"((P) == 0 ? sh_[(JC) & 1][k_] : (P) == 1 ? sm_[(JC) & 1][k_] : sl_[(JC) & 1][k_])); } else \
  { unsigned char* o_ = wbase + (STG) * STAGE + woff0 + (JC) * 576 + (P) * PLANE;                              \
    if ((P) == 0) *reinterpret_cast<uint4*>(o_) = make_uint4(sh_[(JC) & 1][0], sh_[(JC) & 1][1], sh_[(JC) & 1][2], sh_[(JC) & 1][3]);      \
    else if ((P) == 1) *reinterpret_cast<uint4*>(o_) = make_uint4(sm_[(JC) & 1][0], sm_[(JC) & 1][1], sm_[(JC) & 1][2], sm_[(JC) & 1][3]); \
    else *reinterpret_cast<uint4*>(o_) = make_uint4(sl_[(JC) & 1][0], sl_[(JC) & 1][1], sl_[(JC) & 1][2], sl_[(JC) & 1][3]); }
  // HALF: 24 MFMAs on the fragment set (GV, XV); reads the other set (GVN, XVN) for step KSN of stage STN; splits columns JC0, JC0 + 1 of RV
  // into stage NSTG; requests rows I0 .. I0 + 3 of slab SL into RVL
#define WPIPE_HALF(GV, XV, GVN, XVN, STN, KSN, RV, JC0, NSTG, RVL, I0, SL)                                    \
  _Pragma("unroll") for (int n_ = 0; n_ < 24; ++n_) {                                                         \
    if (WPIPE_KNOCK != 3) { WPIPE_MFMA(GV, XV, n_ >> 2, n_ & 3); }                                            \
    const int jc_ = (JC0) + n_ / 12, q_ = (n_ % 12) / 3, sl3_ = n_ % 3;                                       \
    if (n_ < 12) { WPIPE_FRD1(STN, KSN, n_, GVN, XVN); }                                                      \
    else if (n_ % 3 == 1) { WPIPE_GLD1(RVL, (I0) + (n_ - 13) / 3, SL); }                                      \
    if (WPIPE_KNOCK != 1 && WPIPE_KNOCK != 2) {                                                               \
      if (sl3_ == 0) { WPIPE_SLICE_A(RV, jc_, q_); }                                                          \
      else if (sl3_ == 1) { WPIPE_SLICE_B(); }                                                                \
      else { WPIPE_SLICE_C(jc_ & 1, q_); }                                                                    \
    }                                                                                                         \
    if (WPIPE_KNOCK != 2) {                                                                                   \
      if ((JC0) == 2 && (n_ == 1 || n_ == 3 || n_ == 5)) { WPIPE_WRITE1(NSTG, 1, (n_ - 1) / 2); }             \
      if (n_ == 13 || n_ == 15 || n_ == 17) { WPIPE_WRITE1(NSTG, JC0, (n_ - 13) / 2); }                       \
      if ((JC0) == 2 && n_ == 23) { WPIPE_WRITE1(NSTG, 3, 0); WPIPE_WRITE1(NSTG, 3, 1); WPIPE_WRITE1(NSTG, 3, 2); } \
    }                                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#define WPIPE_BODY(S, RB)                                                                                     \
  {                                                                                                           \
    const unsigned char* st_ = wp_smem + ((S) & 1) * STAGE;                                                   \
    const int nstg_ = ((S) + 1) & 1;                                                                          \
    WPIPE_HALF(gv1, xv1, gv0, xv0, st_, 0, raw[RB], 0, nstg_, raw[((RB) + WPIPE_PF - 1) % WPIPE_PF], 0, (S) + WPIPE_PF) \
    WPIPE_HALF(gv0, xv0, gv1, xv1, st_, 1, raw[RB], 2, nstg_, raw[((RB) + WPIPE_PF - 1) % WPIPE_PF], 4, (S) + WPIPE_PF) \
    __builtin_amdgcn_s_waitcnt(0xc07f);      /* lgkmcnt(0), as a builtin: the compiler's own wait counting sees it */ \
    __builtin_amdgcn_s_barrier();                                                                             \
    asm volatile("" ::: "memory");                                                                            \
  }
  WPIPE_STAMP(1);
  for (int s = 0; s < ns; s += WPIPE_PF) {        // unrolled by the number of raw sets: their indices are compile-time constants
    WPIPE_BODY(s, 1)
    if (s + 1 < ns) WPIPE_BODY(s + 1, 2 % WPIPE_PF)
    if (WPIPE_PF >= 3 && s + 2 < ns) WPIPE_BODY(s + 2, 3 % WPIPE_PF)
    if (WPIPE_PF >= 4 && s + 3 < ns) WPIPE_BODY(s + 3, 0)
  }
  WPIPE_STAMP(2);
  // the last slab's second step
#pragma unroll
  for (int n_ = 0; n_ < 24; ++n_) { WPIPE_MFMA(gv1, xv1, n_ >> 2, n_ & 3); }
#undef WPIPE_BODY
#undef WPIPE_HALF
#undef WPIPE_FRD1
#undef WPIPE_GLD1
#undef WPIPE_WRITE1
#undef WPIPE_WRITE
#undef WPIPE_SLICE_A
#undef WPIPE_SLICE_B
#undef WPIPE_SLICE_C
#undef WPIPE_X

  // acc[e][f][4 g + r] = dW[co0 + 64 wco + 32 e + 8 g + 4 hf + r][ci0 + 64 wci + 32 f + l31]
  float* out = a.part + (a.accumulate ? 0 : (long)group * a.Co * a.Ci);
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int ci = ci0 + 64 * wci + 32 * f + l31;
      if (ci >= a.Ci) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 64 * wco + 32 * e + 8 * (r >> 2) + 4 * hf + (r & 3);
        if (co >= a.Co) continue;
        float* o = out + (long)co * a.Ci + ci;
        *o = a.accumulate ? *o + acc[e][f][r] : acc[e][f][r];
      }
    }
  WPIPE_STAMP(3);
}
#undef WPIPE_MFMA

// ---------------------------------------------------------------------------------------------------------------------
// The same product with the two kinds of work on DIFFERENT waves (round 5).  Stamps of the single-stream kernel above say what the
// matrix pipe tolerates beside it in ONE wave (cycles per 32-row slab and workgroup, floor 48 MFMAs x 32 = 1536): MFMAs + fragment
// reads + row requests 1585; + the plane stores 1824; + the 176 vector instructions of the split 2298 with the stores left out, 2538
// with them - in one wave's stream a vector instruction adds its 4 issue cycles to the MFMA's 32 instead of hiding under them.
// A second wave on the SIMD does hide (MI355X_MICROARCH.md 'Two waves per SIMD'): so 512 threads - waves 0-3 only read fragments and
// issue MFMAs (2 x 2 wave tiles of 64 x 64 as before), waves 4-7 only request rows, split and store planes; one of each per SIMD, one
// barrier per slab for all eight, two LDS stages, same image, same MFMA order: bit-identical results again.
#ifdef WPIPE_STAMPS
#define WWS_STAMP(slot) do { if (wpipe_stamps && threadIdx.x == 0) { wpipe_stamps[blockIdx.x * 8l + (slot)] = __builtin_amdgcn_s_memtime(); \
    if ((slot) == 0 || (slot) == 3) wpipe_stamps[blockIdx.x * 8l + 4 + ((slot) == 3)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define WWS_STAMP(slot) do { } while (0)
#endif
#ifndef WWS_PF
#define WWS_PF 3           // raw row sets of a loader wave (slabs requested ahead)
#endif
#ifndef WWS_CHAIN
#define WWS_CHAIN 49       // slabs (32 rows) of one fp32 accumulation chain: 1568 rows
#endif
#ifndef WWS_PRIO
#define WWS_PRIO 1         // s_setprio of the MFMA waves
#endif
// GBN: the G operand is the BatchNorm backward apply dZ = c1*G + c2*G2 + c3 (per column), formed by the G loader waves from two tensors
// (their vector work grows by 2 FMAs per element, their requests double); with a.dz set the items of ci tile 0 also STORE it, so the
// data gradient that follows reads a materialised dZ and the separate 2R + 1W pass (bn_bwd_apply) is gone - unlike round 4's fold into
// BOTH consumers, no GEMM kernel carries a second operand stream in its matrix waves.
template <bool GBN>
__global__ __launch_bounds__(512, 1) void wgrad_split_ws_kernel(WtArgs a) {
  constexpr int PLANE = 4 * 144 * 16, STAGE = 6 * PLANE;    // the image of wgrad_split_pipe_kernel
  constexpr int PF = WWS_PF;                                // raw row sets (the folded form holds two tensors' rows in each)
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_smem[];      // 2 * STAGE
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.tiles_co * a.tiles_ci;
  // Persistent: one workgroup per CU walks work items (group, tile), tile fastest.  Items are dealt in chunks of 32 consecutive ones
  // (gridDim / 8; mostly one group's tiles: they read the same rows) to the XCDs in turn; blocks b and b + 8 share an XCD (round-robin dispatch -
  // a placement for speed only), so block b takes item (b >> 3) of the chunks (b & 7), (b & 7) + 8, ...  The loader waves start an
  // item's first slabs while the MFMA waves still store the previous item's tile.
  const int items = tiles * a.groups;
  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7, chunk = gridDim.x >> 3;      // (the grid is a multiple of 8: 256, or fewer for a short list)
  WWS_STAMP(0);

  if (wave >= 4) {
    // ---- loader waves: thread = (4-column chunk c of 32, row group rg of 4) of G (waves 4-5) or X (waves 6-7): rows 8 rg .. 8 rg + 7 of the slab
    const bool isx = wave >= 6;
    const int lt = tid & 127, c = lt & 31, rg = lt >> 5;
    const int ld = __builtin_amdgcn_readfirstlane(isx ? a.ldx : a.ldg);
    const unsigned slab_bytes = (unsigned)(32 * ld * 4), row_bytes = (unsigned)(ld * 4);
    typedef float wf4 __attribute__((ext_vector_type(4)));
    wf4 raw[PF][8];
    wf4 raw2[GBN ? PF : 1][8];
    const bool isg = GBN && !isx;                            // (wave-uniform: waves 4-5)
    unsigned char* const wb = ws_smem + (isx ? 3 * PLANE : 0) + (rg * 144 + c) * 16;
#define WWS_GLOAD(K, S)                                                                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                              \
      raw[K][i_] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs, vbase + (unsigned)i_ * row_bytes, (unsigned)(S) * slab_bytes, 0)); \
      if (isg) raw2[GBN ? (K) : 0][i_] = __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs2, vbase + (unsigned)i_ * row_bytes, (unsigned)(S) * slab_bytes, 0)); \
    }
    // the folded operand: dZ in place of the raw rows of slab S (rows past the group give c3 - they meet zero rows of X), stored once per G block
#define WWS_FOLD(K, S)                                                                                                              \
    if (isg) {                                                                                                                      \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                                                            \
        _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                                            \
          raw[K][i_][e_] = __builtin_fmaf(gc1[e_], raw[K][i_][e_], __builtin_fmaf(gc2[e_], raw2[GBN ? (K) : 0][i_][e_], gc3[e_])); \
        if (wr_dz) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu4, raw[K][i_]), rdz, vbase + (unsigned)i_ * row_bytes, (unsigned)(S) * slab_bytes, 0); \
      }                                                                                                                             \
    }
// (lab, round 5: the two residual subtractions of adjacent columns as ONE v_pk_add_f32 - 144 instead of 176 vector instructions per
// slab - took the MFMA waves from 39.8 to 62.6 cycles per MFMA: packed f32 arithmetic on the partner wave stalls the matrix pipe)
#define WWS_SPLIT(RV, STG)                                                                                                          \
    _Pragma("unroll") for (int jc_ = 0; jc_ < 4; ++jc_) {                                                                           \
      unsigned h_[4], m_[4], l_[4];                                                                                                 \
      _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                                            \
        if (WPIPE_KNOCK == 1) { h_[q_] = __float_as_uint(RV[2 * q_][jc_]); m_[q_] = __float_as_uint(RV[2 * q_ + 1][jc_]); l_[q_] = 0; }    \
        else wsplit3_pair(RV[2 * q_][jc_], RV[2 * q_ + 1][jc_], h_[q_], m_[q_], l_[q_]); }                                          \
      unsigned char* o_ = wb + (STG) * STAGE + jc_ * 576;                                                                           \
      *reinterpret_cast<uint4*>(o_) = make_uint4(h_[0], h_[1], h_[2], h_[3]);                                                       \
      *reinterpret_cast<uint4*>(o_ + PLANE) = make_uint4(m_[0], m_[1], m_[2], m_[3]);                                               \
      *reinterpret_cast<uint4*>(o_ + 2 * PLANE) = make_uint4(l_[0], l_[1], l_[2], l_[3]);                                           \
    }
#define WWS_LBODY(S, RB)                                                                                                            \
    {                                                                                                                               \
      WWS_GLOAD(((RB) + PF - 1) % PF, (S) + PF)                                                                                     \
      WWS_FOLD(RB, (S) + 1)                                                                                                         \
      WWS_SPLIT(raw[RB], ((S) + 1) & 1)                                                                                             \
      __builtin_amdgcn_s_waitcnt(0xc07f);                                                                                           \
      __builtin_amdgcn_s_barrier();                                                                                                 \
      asm volatile("" ::: "memory");                                                                                                \
    }
    for (int it = 0;; ++it) {
      const int item = (xcd + 8 * it) * chunk + slot;
      if (item >= items) break;
      const int group = item / tiles, tile = item - group * tiles;
      const int c0 = isx ? (tile % a.tiles_ci) * 128 : (tile / a.tiles_ci) * 128;
      const long r_beg = (long)group * a.rows_per_group;
      const int rows = (int)(min((long)a.R, r_beg + a.rows_per_group) - r_beg);
      const int ns = (rows + 31) / 32;
      const int col0 = c0 + 4 * c;
      const bool colok = col0 < (isx ? a.Ci : a.Co);
      // the descriptor covers exactly the group's rows: rows past its end and columns past the matrix read as zeros by the range check
      const __amdgpu_buffer_rsrc_t rs = wbuf_rsrc((isx ? a.X.p : a.G) + r_beg * ld, (long)rows * ld * 4);
      const unsigned vbase = colok ? (unsigned)((8 * rg * ld + col0) * 4) : 0x7f000000u;
      const __amdgpu_buffer_rsrc_t rs2 = isg ? wbuf_rsrc(a.G2 + r_beg * ld, (long)rows * ld * 4) : rs;
      const bool wr_dz = isg && a.dz != nullptr && tile % a.tiles_ci == 0;
      const __amdgpu_buffer_rsrc_t rdz = wr_dz ? wbuf_rsrc(a.dz + r_beg * ld, (long)rows * ld * 4) : rs;
      typedef unsigned wu4 __attribute__((ext_vector_type(4)));
      wf4 gc1 = {1.f, 1.f, 1.f, 1.f}, gc2 = {0.f, 0.f, 0.f, 0.f}, gc3 = gc2;
      if (isg && colok) {
        gc1 = *reinterpret_cast<const wf4*>(a.gcoef + col0); gc2 = *reinterpret_cast<const wf4*>(a.gcoef + a.Co + col0);
        gc3 = *reinterpret_cast<const wf4*>(a.gcoef + 2 * a.Co + col0);
      }
      WWS_GLOAD(0, 0)
      WWS_GLOAD(1, 1)
      if (PF >= 3) { WWS_GLOAD(PF >= 3 ? 2 : 0, 2) }
      if (PF >= 4) { WWS_GLOAD(PF >= 4 ? 3 : 0, 3) }
      WWS_FOLD(0, 0)
      WWS_SPLIT(raw[0], 0)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int s = 0; s < ns; s += PF) {
        WWS_LBODY(s, 1)
        if (s + 1 < ns) WWS_LBODY(s + 1, 2 % PF)
        if (PF >= 3 && s + 2 < ns) WWS_LBODY(s + 2, 3 % PF)
        if (PF >= 4 && s + 3 < ns) WWS_LBODY(s + 3, 0)
      }
    }
#undef WWS_LBODY
#undef WWS_FOLD
#undef WWS_SPLIT
#undef WWS_GLOAD
    return;
  }

  // ---- MFMA waves
  if (WWS_PRIO) __builtin_amdgcn_s_setprio(WWS_PRIO);
  const int l31 = lane & 31, hf = lane >> 5;
  const int wco = wave >> 1, wci = wave & 1;
  const int fro = (hf * 144 + (l31 & 3) * 36 + (l31 >> 2)) * 16;
  const int cgo = fro + wco * 256, cxo = fro + 3 * PLANE + wci * 256;
  constexpr int PG[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
#define WWS_MFMA(GV, XV, T, EF)                                                                                                \
  acc[(EF) >> 1][(EF) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(GV[(EF) >> 1][PG[T]], XV[(EF) & 1][PX[T]], acc[(EF) >> 1][(EF) & 1], 0, 0, 0)
#define WWS_FRD1(ST, KS, R, GVT, XVT)                                                                          \
  { const int g_ = (R) >> 2, e_ = (R) & 1;                                                                     \
    if ((((R) >> 1) & 1) == 0) GVT[e_][g_ == 0 ? 0 : g_ == 1 ? 2 : 1] = *reinterpret_cast<const wbf16x8*>((ST) + cgo + (g_ == 0 ? 0 : g_ == 1 ? 2 : 1) * PLANE + e_ * 128 + (KS) * 4608); \
    else XVT[e_][g_ == 0 ? 2 : g_ == 1 ? 0 : 1] = *reinterpret_cast<const wbf16x8*>((ST) + cxo + (g_ == 0 ? 2 : g_ == 1 ? 0 : 1) * PLANE + e_ * 128 + (KS) * 4608); }
  // 24 MFMAs on (GV, XV); the 12 fragments of the next 16-row step go into the other set, one read per MFMA gap, in the order the MFMAs want them
#define WWS_HALF(GV, XV, GVN, XVN, STN, KSN)                                                                   \
  _Pragma("unroll") for (int n_ = 0; n_ < 24; ++n_) {                                                          \
    WWS_MFMA(GV, XV, n_ >> 2, n_ & 3);                                                                         \
    if (n_ < 12) { WWS_FRD1(STN, KSN, n_, GVN, XVN); }                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  }
  for (int it = 0;; ++it) {
    const int item = (xcd + 8 * it) * chunk + slot;
    if (item >= items) break;
    const int group = item / tiles, tile = item - group * tiles;
    const int co0 = (tile / a.tiles_ci) * 128, ci0 = (tile % a.tiles_ci) * 128;
    const long r_beg = (long)group * a.rows_per_group;
    const int rows = (int)(min((long)a.R, r_beg + a.rows_per_group) - r_beg);
    const int ns = (rows + 31) / 32;
    wf32x16 acc[2][2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    wbf16x8 gv0[2][3], xv0[2][3], gv1[2][3], xv1[2][3];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < 8; ++i) { gv1[e][p][i] = (__bf16)0.f; xv1[e][p][i] = (__bf16)0.f; }      // the first half step multiplies zeros
    __builtin_amdgcn_s_barrier();                // slab 0 of this item is in stage 0
    asm volatile("" ::: "memory");
    if (it == 0) WWS_STAMP(1);
    // A group may hold several accumulation chains: every WWS_CHAIN slabs (1568 rows, the bound the fp64 tests set for ONE fp32 chain)
    // the accumulators are added into a second set and cleared - the same roundings as that many separate groups added by the reduce
    // kernel, without their items, partial tiles and reduce traffic.
    wf32x16 acc2[2][2];
    bool flushed = false;
    int flush_at = WWS_CHAIN;
    for (int s = 0; s < ns; ++s) {
      const unsigned char* st_ = ws_smem + (s & 1) * STAGE;
      WWS_HALF(gv1, xv1, gv0, xv0, st_, 0)
      if (s == flush_at) {                       // slabs < s are complete in acc (the half above was slab s - 1's second step)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc2[e][f][r] = flushed ? acc2[e][f][r] + acc[e][f][r] : acc[e][f][r]; acc[e][f][r] = 0.f; }
        flushed = true;
        flush_at += WWS_CHAIN;
      }
      WWS_HALF(gv0, xv0, gv1, xv1, st_, 1)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (it == 0) WWS_STAMP(2);
#pragma unroll
    for (int n_ = 0; n_ < 24; ++n_) { WWS_MFMA(gv1, xv1, n_ >> 2, n_ & 3); }
    if (flushed) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[e][f][r] = acc2[e][f][r] + acc[e][f][r];
    }
    // acc[e][f][4 g + r] = dW[co0 + 64 wco + 32 e + 8 g + 4 hf + r][ci0 + 64 wci + 32 f + l31]
    if (!a.accumulate) {
      // partial tile of this group: 64 one-dword buffer stores per lane, the row / block part of the address in the scalar offset, rows past
      // Co clipped by the descriptor's range check and columns past Ci parked behind it - no compares, no branches, no 64-bit address math
      const __amdgpu_buffer_rsrc_t ro = wbuf_rsrc(a.part + (long)group * a.Co * a.Ci, (long)a.Co * a.Ci * 4);
      const int cib = ci0 + 64 * wci + l31;
      const unsigned rowo = (unsigned)(((co0 + 64 * wco + 4 * hf) * a.Ci) * 4);
      const unsigned vo[2] = {cib < a.Ci ? rowo + (unsigned)cib * 4u : 0x7f000000u, cib + 32 < a.Ci ? rowo + (unsigned)(cib + 32) * 4u : 0x7f000000u};
      const unsigned rstep = (unsigned)(a.Ci * 4);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e][f][r]), ro, vo[f], (unsigned)(32 * e + 8 * (r >> 2) + (r & 3)) * rstep, 0);
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int ci = ci0 + 64 * wci + 32 * f + l31;
          if (ci >= a.Ci) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + 64 * wco + 32 * e + 8 * (r >> 2) + 4 * hf + (r & 3);
            if (co >= a.Co) continue;
            float* o = a.part + (long)co * a.Ci + ci;
            *o = *o + acc[e][f][r];
          }
        }
    }
    if (it == 0) WWS_STAMP(3);
  }
#undef WWS_HALF
#undef WWS_FRD1
#undef WWS_MFMA
}

// ---------------------------------------------------------------------------------------------------------------------
// The exact-fp32 weight gradient on specialised waves (round 5; mx_set_gemm_mode(0) only - the default arithmetic never gets here).
// fp32 MFMA operands need no split and no transposition: v_mfma_f32_32x32x2_f32 takes ONE float per lane for each operand, element
// (column l % 32, row l / 32 of a two-row K step), which is how the rows of G and X lie in memory.  So the loader waves (4-7) only MOVE rows,
// with buffer_load_dwordx4 ... lds (range-checked LDS-DMA: rows past the group and columns past the matrix land as zeros), two rows of 128
// columns per instruction, 8 instructions per wave and 32-row slab, no vector arithmetic and no ds_write at all; the MFMA waves (0-3, a 64 x 64
// quadrant each) read 4 fragments (ds_read_b32) per 4 MFMAs of 64 cycles.  LDS image of an operand's slab: [32 rows][128 floats], the 16-byte unit u
// of an odd row stored at position u ^ 8, so that the two rows one fragment read touches sit in different halves of the 64 banks.  Three stages
// (96 KB).  Same persistent item loop, row groups, 1568-row chains and partial-tile stores as wgrad_split_ws_kernel.
#ifndef WF32_NST
#define WF32_NST 3
#endif
// E, F: 32-row / 32-column blocks of a wave's quadrant - the workgroup's tile is 64 E x 64 F (128 x 128, 128 x 64 for narrow Ci such as 960 x 160,
// 64 x 128 for its mirror); the LDS image keeps its 128-float rows, the loader lanes past a 64-wide tile ask for nothing.
template <int E, int F>
__global__ __launch_bounds__(512, 1) void wgrad_f32_ws_kernel(WtArgs a) {
  constexpr int OPB = 32 * 512, STAGE = 2 * OPB, NST = WF32_NST;
  constexpr int BCO = 64 * E, BCI = 64 * F;
  extern __shared__ __attribute__((aligned(16))) unsigned char wf_smem[];      // NST * STAGE
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles = a.tiles_co * a.tiles_ci;
  const int items = tiles * a.groups;
  const int slot = blockIdx.x >> 3, xcd = blockIdx.x & 7, chunk = gridDim.x >> 3;
  typedef __attribute__((address_space(3))) void* wlds_t;

  if (wave >= 4) {
    // ---- loader waves: 4 / 5 move rows 0-15 / 16-31 of G's slab, 6 / 7 those of X's; lane = (row parity, 16-byte unit of the 128 columns)
    const bool isx = wave >= 6;
    const int half = wave & 1;
    const int ld = __builtin_amdgcn_readfirstlane(isx ? a.ldx : a.ldg);
    const int rpar = lane >> 5, gu = (lane & 31) ^ (rpar << 3);
    const unsigned slab_bytes = (unsigned)(32 * ld * 4), row2_bytes = (unsigned)(2 * ld * 4);
    unsigned char* const lbase = wf_smem + (isx ? OPB : 0) + 16 * half * 512;
#define WF32_ISSUE(S)                                                                                                              \
    { unsigned char* d_ = lbase + ((S) % NST) * STAGE;                                                                              \
      _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                                                              \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (wlds_t)(d_ + i_ * 1024), 16, vbase + (unsigned)i_ * row2_bytes, (unsigned)(S) * slab_bytes, 0, 0); }
    for (int it = 0;; ++it) {
      const int item = (xcd + 8 * it) * chunk + slot;
      if (item >= items) break;
      const int group = item / tiles, tile = item - group * tiles;
      const int c0 = isx ? (tile % a.tiles_ci) * BCI : (tile / a.tiles_ci) * BCO;
      const long r_beg = (long)group * a.rows_per_group;
      const int rows = (int)(min((long)a.R, r_beg + a.rows_per_group) - r_beg);
      const int ns = (rows + 31) / 32;
      const int col0 = c0 + 4 * gu;
      const __amdgpu_buffer_rsrc_t rs = wbuf_rsrc((isx ? a.X.p : a.G) + r_beg * ld, (long)rows * ld * 4);
      const unsigned vbase = (4 * gu < (isx ? BCI : BCO) && col0 < (isx ? a.Ci : a.Co)) ? (unsigned)(((16 * half + rpar) * ld + col0) * 4) : 0x7f000000u;
      WF32_ISSUE(0)
      if (ns > 1) {
        WF32_ISSUE(1)
        __builtin_amdgcn_s_waitcnt(0x0f78);       // vmcnt(8): slab 0 has landed
      } else __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      for (int s = 0; s < ns; ++s) {
        if (s + 2 < ns) {
          WF32_ISSUE(s + 2)                         // into the stage the MFMA waves left at the last barrier
          __builtin_amdgcn_s_waitcnt(0x0f78);       // slab s + 1 has landed
        } else __builtin_amdgcn_s_waitcnt(0x0f70);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
#undef WF32_ISSUE
    return;
  }

  // ---- MFMA waves
  const int l31 = lane & 31, hf = lane >> 5;
  const int wco = wave >> 1, wci = wave & 1;
  // float f of a row sits at byte ((f >> 2) ^ (8 * row parity)) * 16 + (f & 3) * 4; this lane reads rows of parity hf
  int og[E], ox[F];
#pragma unroll
  for (int e = 0; e < E; ++e) { const int f_ = 32 * E * wco + 32 * e + l31; og[e] = hf * 512 + (((f_ >> 2) ^ (hf << 3)) << 4) + ((f_ & 3) << 2); }
#pragma unroll
  for (int f = 0; f < F; ++f) { const int f_ = 32 * F * wci + 32 * f + l31; ox[f] = OPB + hf * 512 + (((f_ >> 2) ^ (hf << 3)) << 4) + ((f_ & 3) << 2); }
  for (int it = 0;; ++it) {
    const int item = (xcd + 8 * it) * chunk + slot;
    if (item >= items) break;
    const int group = item / tiles, tile = item - group * tiles;
    const int co0 = (tile / a.tiles_ci) * BCO, ci0 = (tile % a.tiles_ci) * BCI;
    const long r_beg = (long)group * a.rows_per_group;
    const int rows = (int)(min((long)a.R, r_beg + a.rows_per_group) - r_beg);
    const int ns = (rows + 31) / 32;
    wf32x16 acc[E][F], acc2[E][F];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
      for (int f = 0; f < F; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][f][r] = 0.f;
    bool flushed = false;
    int flush_at = WWS_CHAIN;
    __builtin_amdgcn_s_barrier();                // slab 0 of this item is in stage 0
    asm volatile("" ::: "memory");
    for (int s = 0; s < ns; ++s) {
      if (s == flush_at) {                       // one fp32 accumulation chain is at most 1568 rows (as in the split kernel)
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int f = 0; f < F; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc2[e][f][r] = flushed ? acc2[e][f][r] + acc[e][f][r] : acc[e][f][r]; acc[e][f][r] = 0.f; }
        flushed = true;
        flush_at += WWS_CHAIN;
      }
      const unsigned char* st_ = wf_smem + (s % NST) * STAGE;
      float gv[E], xv[F];
#pragma unroll
      for (int e = 0; e < E; ++e) gv[e] = *reinterpret_cast<const float*>(st_ + og[e]);
#pragma unroll
      for (int f = 0; f < F; ++f) xv[f] = *reinterpret_cast<const float*>(st_ + ox[f]);
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        float ng[E], nx[F];
#pragma unroll
        for (int e = 0; e < E; ++e) ng[e] = kk < 15 ? *reinterpret_cast<const float*>(st_ + og[e] + (kk + 1) * 1024) : 0.f;      // the next two rows' fragments,
#pragma unroll
        for (int f = 0; f < F; ++f) nx[f] = kk < 15 ? *reinterpret_cast<const float*>(st_ + ox[f] + (kk + 1) * 1024) : 0.f;      // under this step's MFMAs
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
          for (int f = 0; f < F; ++f) acc[e][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(gv[e], xv[f], acc[e][f], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < E; ++e) gv[e] = ng[e];
#pragma unroll
        for (int f = 0; f < F; ++f) xv[f] = nx[f];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);        // every LDS read of this stage has returned
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (flushed) {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[e][f][r] = acc2[e][f][r] + acc[e][f][r];
    }
    // acc[e][f][4 g + r] = dW[co0 + 32 E wco + 32 e + 8 g + 4 hf + r][ci0 + 32 F wci + 32 f + l31]  (the layout of every 32 x 32 MFMA result)
    if (!a.accumulate) {
      const __amdgpu_buffer_rsrc_t ro = wbuf_rsrc(a.part + (long)group * a.Co * a.Ci, (long)a.Co * a.Ci * 4);
      const int cib = ci0 + 32 * F * wci + l31;
      const unsigned rowo = (unsigned)(((co0 + 32 * E * wco + 4 * hf) * a.Ci) * 4);
      unsigned vo[F];
#pragma unroll
      for (int f = 0; f < F; ++f) vo[f] = cib + 32 * f < a.Ci ? rowo + (unsigned)(cib + 32 * f) * 4u : 0x7f000000u;
      const unsigned rstep = (unsigned)(a.Ci * 4);
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[e][f][r]), ro, vo[f], (unsigned)(32 * e + 8 * (r >> 2) + (r & 3)) * rstep, 0);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e)
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const int ci = ci0 + 32 * F * wci + 32 * f + l31;
          if (ci >= a.Ci) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + 32 * E * wco + 32 * e + 8 * (r >> 2) + 4 * hf + (r & 3);
            if (co >= a.Co) continue;
            float* o = a.part + (long)co * a.Ci + ci;
            *o = *o + acc[e][f][r];
          }
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WtPlan { int te, tf, tiles_co, tiles_ci, groups, rows_per_group, f32ws; };
int mx_wgrad_pipe_override = -1;      // lab hook (tools/hip/gemm_lab.hip): 0 / 1 selects the first / pipelined split kernel per call

static int wt_order() {
  static const int order = getenv("MX_WGRAD_TILE_ORDER") ? atoi(getenv("MX_WGRAD_TILE_ORDER")) : 0;
  return order;
}

extern "C" int mx_get_gemm_mode(void);

static bool wt_use_split(int Co, int Ci) {
  if (mx_get_gemm_mode() < 1) return false;
  const double eff = ((double)Co / (128.0 * cdiv(Co, 128))) * ((double)Ci / (128.0 * cdiv(Ci, 128)));
  // 128 x 128 tiles only.  0.8 until late round 4 (not where they pad much); with the row groups filling the chip for few-tile outputs
  // (wt_plan) the padded shapes win as well - 960 x 160 and 480 x 80 (0.59 of their tiles used): 105 -> 76 us and 136 -> 77 us, and the
  // step 96.9 -> 95.15 ms on one box (profiles/r04_knob_sweep.txt)
  static const double min_eff = getenv("MX_WGRAD_SPLIT_EFF") ? atof(getenv("MX_WGRAD_SPLIT_EFF")) : 0.55;
  return mx_get_gemm_mode() == 2 || eff >= min_eff;
}

// which split weight-gradient kernel takes the plain-operand launches: 0 wgrad_split_kernel, 1 wgrad_split_pipe_kernel, 2 wgrad_split_ws_kernel
static int wt_pipe_mode() {
  static const int pipe_env = getenv("MX_WGRAD_PIPE") ? atoi(getenv("MX_WGRAD_PIPE")) : 2;
  return mx_wgrad_pipe_override >= 0 ? mx_wgrad_pipe_override : pipe_env;
}
static int g_wgrad_groups = 0;         // mx_set_wgrad_kernel: > 0 fixes the row groups of the plain split launches
static int g_wgrad_f32ws = getenv("MX_WGRAD_F32_WS") ? atoi(getenv("MX_WGRAD_F32_WS")) : 1;      // exact-fp32 mode: wgrad_f32_ws_kernel for the 128 x 128 tiles (0: the tiled kernel)

static bool wt_plan(int R, int Co, int Ci, int x_mode, WtPlan* p) {
  if (Co % 4 || Ci % 4 || R < 1024 || (long)Co * Ci < 16384) return false;
  p->f32ws = 0;
  if (wt_use_split(Co, Ci)) {
    p->te = p->tf = 4;
    p->tiles_co = cdiv(Co, 128); p->tiles_ci = cdiv(Ci, 128);
    // Row groups of the split kernel.  Standalone, many small work items are fastest (g = 32 at up to ~100 tiles: 1344 x 224 132 ->
    // 118 us; profiles/r04_wgrad_split_groups.txt) - but the weight gradients run on the side stream BESIDE the HBM-bound kernels
    // of the backward chain, and there every extra group is a partial matrix written and read back through the memory system
    // those kernels live on: in the step FEWER groups win (B7 / 448 / batch 32, one box, every shape at g = 2 / 4 / 6 / 8 / 16 / 24 /
    // the standalone-best rule: 131.8 / 111.3 / 104.9 / 102.6 / 104.9 / 105.5 / 105.8 ms per step).  The floor is set by the
    // arithmetic, not by speed: a group is ONE fp32 accumulation chain per output element, and the error against fp64 grows with
    // its length - at g = 8 (3136 rows at R = 25 088, 1568 at 12 544) it is 2.2-2.6x the exact-fp32 tiled kernel's, whose own
    // groups are short; tests/test_gpu_split.py holds the split kernel to 1.5x.  Sixteen groups (two per XCD under the XCD-aware
    // ids) keep every tested shape level with the fp32 kernel and are within 0.3 ms per step of g = 8.
    // The bound is on the CHAIN, not on the count: at most 1568 rows per group, so R = 50 176 (B7 at batch 64, or batch 32 at larger
    // images) takes 32 groups instead of running 3136-row chains at 16 (tests/test_gpu_split.py has R = 50 176 and 62 720 cases).
    const int maxg = R / 128 > 0 ? R / 128 : 1;
    if (x_mode == MX_PLAIN && wt_pipe_mode() >= 2) {
      // wgrad_split_ws_kernel: ONE persistent workgroup per CU deals the (group, tile) items out evenly, so the count that matters is
      // items / 256 rounded up.  Cost model from the kernel's own stamps at the clock it holds (~1.75 GHz): 1.09 us per 32-row slab,
      // 3.4 us per item (first slabs + storing the partial tile), and the partial matrices once more through the reduce kernel.
      // The chain bound (<= 1568 rows per group) is the floor; 54 tiles take 18 groups (3.8 rounds) instead of 16 (3.4 -> 4 rounds).
      const int tiles = p->tiles_co * p->tiles_ci;
      // (a group may hold up to MX_WGRAD_WS_CHAINS chains of 1568 rows, flushed inside the kernel: fewer items and partial tiles)
      static const int chains = getenv("MX_WGRAD_WS_CHAINS") ? atoi(getenv("MX_WGRAD_WS_CHAINS")) : 3;
      int gmin = cdiv(R, 1568 * (chains > 0 ? chains : 1));
      if (gmin < 1) gmin = 1;
      if (gmin > maxg) gmin = maxg;
      int best_g = gmin;
      double best_t = 1e30;
      // (every count up to 128-row groups is priced: outputs of a few tiles - 672 x 112 at 12 544 rows, B0 at batch 16 - fill the 256
      //  workgroups only with 40+ groups; the search used to stop at 3 gmin + 8)
      static const int wide = getenv("MX_WGRAD_WS_WIDE") ? atoi(getenv("MX_WGRAD_WS_WIDE")) : 1;
      for (int g = gmin; g <= maxg && (wide || g <= 3 * gmin + 8); ++g) {
        const int rpg = cdiv(cdiv(R, g), 32) * 32, ga = cdiv(R, rpg);
        if (ga != g) continue;                                 // (the rounding of the rows makes some counts unreachable)
        const double t = cdiv(tiles * g, 256) * (rpg / 32 * 1.09 + 3.4) + g * ((double)Co * Ci * 4.0 / 4e6);
        if (t < best_t - 1e-9) { best_t = t; best_g = g; }
      }
      static const int forced_env = getenv("MX_WGRAD_WS_GROUPS") ? atoi(getenv("MX_WGRAD_WS_GROUPS")) : 0;
      const int forced_ws = g_wgrad_groups > 0 ? g_wgrad_groups : forced_env;
      if (forced_ws > 0) best_g = forced_ws < maxg ? forced_ws : maxg;
      p->rows_per_group = cdiv(cdiv(R, best_g), 32) * 32;
      p->groups = cdiv(R, p->rows_per_group);
      return true;
    }
    int groups = (cdiv(R, 1568) + 7) / 8 * 8;
    if (groups < 16) groups = 16;
    // few output tiles (960 x 160: 16, 480 x 80: 4 - taken by this kernel when MX_WGRAD_SPLIT_EFF admits their padding): their partial
    // matrices are small, so the groups are what fills the chip - ~768 workgroups, at most 128 groups of at least 512 rows
    static const int fill = getenv("MX_WGRAD_SPLIT_FILL") ? atoi(getenv("MX_WGRAD_SPLIT_FILL")) : 1;
    const int tiles = p->tiles_co * p->tiles_ci;
    if (fill && tiles < 48) {
      int want = (cdiv(768, tiles) + 7) / 8 * 8;
      static const int minrows = getenv("MX_WGRAD_SPLIT_MINROWS") ? atoi(getenv("MX_WGRAD_SPLIT_MINROWS")) : 512;
      const int cap = R / minrows >= 8 ? R / minrows / 8 * 8 : 8;
      if (want > 128) want = 128;
      if (want > cap) want = cap;
      if (want > groups) groups = want;
    }
    if (groups > maxg) groups = maxg >= 8 ? maxg / 8 * 8 : maxg;
    static const int forced_split = getenv("MX_WGRAD_SPLIT_GROUPS") ? atoi(getenv("MX_WGRAD_SPLIT_GROUPS")) : 0;
    if (forced_split > 0) groups = forced_split < maxg ? forced_split : maxg;
    if (g_wgrad_groups > 0 && x_mode == MX_PLAIN) groups = g_wgrad_groups < maxg ? g_wgrad_groups : maxg;
    p->rows_per_group = cdiv(cdiv(R, groups), 32) * 32;
    p->groups = cdiv(R, p->rows_per_group);
    return true;
  }
  // tile: least padded area, ties to the larger tile
  long best = -1;
  for (int te : {4, 2})
    for (int tf : {4, 2}) {
      const long area = (long)cdiv(Co, 32 * te) * 32 * te * cdiv(Ci, 32 * tf) * 32 * tf;
      const long cost = area * 100 / (te * tf >= 16 ? 100 : te * tf >= 8 ? 95 : 88);      // smaller tiles feed the MFMA less well
      if (best < 0 || cost < best) { best = cost; p->te = te; p->tf = tf; }
    }
  p->tiles_co = cdiv(Co, 32 * p->te); p->tiles_ci = cdiv(Ci, 32 * p->tf);
  const int tiles = p->tiles_co * p->tiles_ci;
  // exact-fp32 arithmetic (mx_set_gemm_mode(0)), 128 x 128 tiles, plain operands: wgrad_f32_ws_kernel - one persistent workgroup per CU, so
  // the row groups come from the same 256-slot model as the split kernel's (2.15 us per 32-row slab: 64 MFMAs of 64 cycles per wave at the
  // clock the chip holds; 3.4 us per item; chains of 1568 rows flushed inside the kernel)
  if (g_wgrad_f32ws && mx_get_gemm_mode() == 0 && x_mode == MX_PLAIN && g_wgrad_groups <= 0) {
    // tile of the wave-specialised fp32 kernel: 128 x 128, or 128 x 64 / 64 x 128 where that pads less MFMA work (960 x 160, 480 x 80 and mirrors)
    int be = 2, bf = 2;
    long bw = (long)cdiv(Co, 128) * cdiv(Ci, 128) * 4;
    if ((long)cdiv(Co, 128) * cdiv(Ci, 64) * 2 < bw) { bw = (long)cdiv(Co, 128) * cdiv(Ci, 64) * 2; be = 2; bf = 1; }
    if ((long)cdiv(Co, 64) * cdiv(Ci, 128) * 2 < bw) { bw = (long)cdiv(Co, 64) * cdiv(Ci, 128) * 2; be = 1; bf = 2; }
    p->te = 2 * be; p->tf = 2 * bf;
    p->tiles_co = cdiv(Co, 64 * be); p->tiles_ci = cdiv(Ci, 64 * bf);
    const int tiles = p->tiles_co * p->tiles_ci;
    const double slab_us = 0.55 * be * bf;                   // 16 K steps x E x F MFMAs of 64 cycles per wave and slab, at the clock the chip holds
    const int maxg = R / 128 > 0 ? R / 128 : 1;
    int gmin = cdiv(R, 1568 * 3);
    if (gmin > maxg) gmin = maxg;
    int best_g = gmin;
    double best_t = 1e30;
    for (int g = gmin; g <= maxg; ++g) {
      const int rpg = cdiv(cdiv(R, g), 32) * 32, ga = cdiv(R, rpg);
      if (ga != g) continue;
      const double t = cdiv(tiles * g, 256) * (rpg / 32 * slab_us + 3.4) + g * ((double)Co * Ci * 4.0 / 4e6);
      if (t < best_t - 1e-9) { best_t = t; best_g = g; }
    }
    p->rows_per_group = cdiv(cdiv(R, best_g), 32) * 32;
    p->groups = cdiv(R, p->rows_per_group);
    p->f32ws = 1;
    return true;
  }
  // Row groups: tiles x groups should fill a whole number of residency rounds (256 CUs x 4 workgroups, 3 for the 128 x 128
  // tile with the BN+SiLU+gate prologue).  1026 workgroups on 1024 slots cost 25 % (two stragglers run alone after a full
  // round); every extra group costs a partial matrix written and read back.
  static const int forced = getenv("MX_WGRAD_TILE_GROUPS") ? atoi(getenv("MX_WGRAD_TILE_GROUPS")) : 0;
  const int slots = 256 * ((p->te * p->tf >= 16 && x_mode == MX_BNACT) ? 3 : 4);
  const int maxg = R / 256 > 0 ? R / 256 : 1;
  int groups = 1;
  double best_score = -1.0;
  static const int rmin = getenv("MX_WGRAD_TILE_RMIN") ? atoi(getenv("MX_WGRAD_TILE_RMIN")) : 2;    // measured in the step: 1 / 2 / 3 -> 132.7 / 131.5 / 131.6 ms
  for (int rounds = rmin; rounds <= rmin + 2; ++rounds) {
    int g = slots * rounds / tiles;
    if (g < 1) g = 1;
    if (g > maxg) g = maxg;
    if (wt_order() == 0 && g >= 8) g = g / 8 * 8;             // XCD-aware ids hand out groups eight at a time
    const double fill = (double)tiles * g / ((double)slots * cdiv(tiles * g, slots));
    const double score = fill - 0.0025 * g;
    if (score > best_score + 1e-9) { best_score = score; groups = g; }
  }
  if (forced > 0) groups = forced < maxg ? forced : maxg;
  p->rows_per_group = cdiv(cdiv(R, groups), 16) * 16;
  p->groups = cdiv(R, p->rows_per_group);
  return true;
}

template <int TE, int TF>
static void wt_launch(const WtArgs& a, hipStream_t st) {
  // MX_WGRAD_TILE_PERSIST = n > 0: at most 256 * n persistent workgroups (n per CU) instead of one per id - leaves wave
  // slots and LDS to the main stream's kernels while this one runs beside them on the side stream
  static const int persist = getenv("MX_WGRAD_TILE_PERSIST") ? atoi(getenv("MX_WGRAD_TILE_PERSIST")) : 0;
  WtArgs b = a;
  b.total = 8 * cdiv(a.groups, 8) * a.tiles_co * a.tiles_ci;
  int g = b.total;
  if (persist > 0 && g > 256 * persist) g = 256 * persist;
  const dim3 grid(g);
  const WtArgs& a_ = b;
  if (a.X.mode == MX_PLAIN) hipLaunchKernelGGL((wgrad_tile_kernel<TE, TF, MX_PLAIN>), grid, dim3(256), 0, st, a_);
  else if (a.X.mode == MX_BNACT) hipLaunchKernelGGL((wgrad_tile_kernel<TE, TF, MX_BNACT>), grid, dim3(256), 0, st, a_);
  else hipLaunchKernelGGL((wgrad_tile_kernel<TE, TF, MX_AFFINE>), grid, dim3(256), 0, st, a_);
}

extern "C" {

int mx_set_wgrad_kernel(int kernel, int groups) {
  MX_CHECK_ARG(kernel >= -1 && kernel <= 2, "set_wgrad_kernel: kernel %d (0 first split kernel, 1 pipelined, 2 wave-specialised, -1 keep)", kernel);
  MX_CHECK_ARG(groups >= -1, "set_wgrad_kernel: groups %d (> 0 fixed, 0 planner, -1 keep)", groups);
  if (kernel >= 0) mx_wgrad_pipe_override = kernel;
  if (groups >= 0) g_wgrad_groups = groups;
  return MX_OK;
}
int mx_get_wgrad_kernel(void) { return wt_pipe_mode(); }

// bytes of scratch mx_pw_wgrad_small needs for (R, Co, Ci, x_mode), or 0 when the shape is not one it takes
long mx_pw_wgrad_small_ws(int R, int Co, int Ci, int x_mode) {
  WgPlan p;
  if (!wg_plan(R, Co, Ci, &p)) return 0;
  WgArgs a{};
  a.Co = Co; a.Ci = Ci; a.X.mode = x_mode;
  if (!wg_dispatch(a, p, nullptr, false)) return 0;
  return (long)p.groups * Co * Ci * 4;
}

// dW[Co,Ci] += G[R,Co]^T X'[R,Ci] for small outputs and long reductions; deterministic (fixed summation order).
// ws: caller-owned scratch of mx_pw_wgrad_small_ws bytes.
int mx_pw_wgrad_small(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                      const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                      void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(G && X && dW && ws, "wgrad_small: null pointer");
  MX_CHECK_ARG(((uintptr_t)dW & 15) == 0 && ((uintptr_t)ws & 15) == 0, "wgrad_small: dW / workspace must be 16-byte aligned");
  MX_CHECK_ARG(ldg % 4 == 0 && ldx % 4 == 0, "wgrad_small: leading dimensions must be multiples of 4");
  MX_CHECK_ARG(x_mode == MX_PLAIN || (x_scale && x_shift && rows_per_sample > 0), "wgrad_small: prologue needs scale/shift");
  WgPlan p;
  MX_CHECK_ARG(wg_plan(R, Co, Ci, &p), "wgrad_small: shape R=%d Co=%d Ci=%d not supported", R, Co, Ci);
  MX_CHECK_ARG(ws_bytes >= (long)p.groups * Co * Ci * 4, "wgrad_small: workspace too small");
  WgArgs a{};
  a.G = G; a.X = MxOperand{X, x_scale, x_shift, x_gate, x_mode, rows_per_sample};
  a.part = (float*)ws; a.R = R; a.Co = Co; a.Ci = Ci; a.ldg = ldg; a.ldx = ldx;
  a.rows_per_wg = p.rows_per_wg; a.WCO = p.wco; a.WCI = p.wci; a.SG = p.SG; a.SX = p.SX;
  hipStream_t st = (hipStream_t)stream;
  MX_CHECK_ARG(wg_dispatch(a, p, st, true), "wgrad_small: no kernel for Co=%d Ci=%d mode=%d (ask mx_pw_wgrad_small_ws first)", Co, Ci, x_mode);
  MX_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_parts_reduce_kernel, dim3(cdiv((long)Co * Ci, 64)), dim3(256), 0, st, (const float*)ws, p.groups, Co * Ci, dW);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// 1 when mx_pw_wgrad_small_bnbwd takes this shape (a small-output kernel with the folded G operand exists), else 0
int mx_pw_wgrad_small_bnbwd_ok(int R, int Co, int Ci) {
  WgPlan p;
  if (!wg_plan(R, Co, Ci, &p)) return 0;
  WgArgs a{};
  a.Co = Co; a.Ci = Ci; a.X.mode = MX_PLAIN; a.G2 = reinterpret_cast<const float*>(16);
  return wg_dispatch(a, p, nullptr, false) ? 1 : 0;
}

// dW[Co,Ci] += dZ[R,Co]^T X[R,Ci] with dZ = c1*G + c2*G2 + c3 per column formed in the loader (small outputs, HBM-bound: the fold
// saves the 2R + 1W pass that materialised dZ); scratch: mx_pw_wgrad_small_ws(R, Co, Ci, 0) bytes
int mx_pw_wgrad_small_bnbwd(const float* G, const float* G2, const float* coef, const float* X, float* dW, int R, int Co, int Ci,
                            int ldg, int ldx, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(G && G2 && coef && X && dW && ws, "wgrad_small_bnbwd: null pointer");
  MX_CHECK_ARG((((uintptr_t)dW | (uintptr_t)ws | (uintptr_t)G2 | (uintptr_t)coef) & 15) == 0, "wgrad_small_bnbwd: pointers must be 16-byte aligned");
  MX_CHECK_ARG(ldg % 4 == 0 && ldx % 4 == 0, "wgrad_small_bnbwd: leading dimensions must be multiples of 4");
  WgPlan p;
  MX_CHECK_ARG(wg_plan(R, Co, Ci, &p), "wgrad_small_bnbwd: shape R=%d Co=%d Ci=%d not supported", R, Co, Ci);
  MX_CHECK_ARG(ws_bytes >= (long)p.groups * Co * Ci * 4, "wgrad_small_bnbwd: workspace too small");
  WgArgs a{};
  a.G = G; a.X = MxOperand{X, nullptr, nullptr, nullptr, MX_PLAIN, 1}; a.G2 = G2; a.gcoef = coef;
  a.part = (float*)ws; a.R = R; a.Co = Co; a.Ci = Ci; a.ldg = ldg; a.ldx = ldx;
  a.rows_per_wg = p.rows_per_wg; a.WCO = p.wco; a.WCI = p.wci; a.SG = p.SG; a.SX = p.SX;
  hipStream_t st = (hipStream_t)stream;
  MX_CHECK_ARG(wg_dispatch(a, p, st, true), "wgrad_small_bnbwd: no kernel for Co=%d Ci=%d (mx_pw_wgrad_small_bnbwd_ok)", Co, Ci);
  MX_LAUNCH_CHECK();
  hipLaunchKernelGGL(wgrad_parts_reduce_kernel, dim3(cdiv((long)Co * Ci, 64)), dim3(256), 0, st, (const float*)ws, p.groups, Co * Ci, dW);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

bool mx_wgrad_uses_split(int R, int Co, int Ci) {
  // the small-output kernel is tried first (ops.pw_wgrad) and stays fp32 - where it has an instantiation for the shape (any prologue):
  // 480 x 80 has a plan but no kernel, and goes to the tiled kernels
  if (mx_pw_wgrad_small_ws(R, Co, Ci, MX_PLAIN) > 0 || mx_pw_wgrad_small_ws(R, Co, Ci, MX_BNACT) > 0) return false;
  WtPlan p;
  return wt_plan(R, Co, Ci, MX_PLAIN, &p) && wt_use_split(Co, Ci);
}

// bytes of scratch mx_pw_wgrad_tile needs (0 = shape not taken; a single row group accumulates straight into dW)
long mx_pw_wgrad_tile_ws(int R, int Co, int Ci, int x_mode) {
  WtPlan p;
  if (!wt_plan(R, Co, Ci, x_mode, &p)) return 0;
  return p.groups > 1 ? (long)p.groups * Co * Ci * 4 : 16;
}

static int wgrad_tile_impl(const float* G, const float* G2, const float* gcoef, const float* X, int x_mode, const float* x_scale,
                           const float* x_shift, const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg,
                           int ldx, void* ws, long ws_bytes, void* stream, float* dz = nullptr);

// dW[Co,Ci] += G[R,Co]^T X'[R,Ci], large outputs: tiled, deterministic (partial tiles per row group, fixed-order reduce).
int mx_pw_wgrad_tile(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                     const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                     void* ws, long ws_bytes, void* stream) {
  return wgrad_tile_impl(G, nullptr, nullptr, X, x_mode, x_scale, x_shift, x_gate, rows_per_sample, dW, R, Co, Ci, ldg, ldx, ws, ws_bytes, stream);
}

// 1 when mx_pw_wgrad_tile_bnbwd takes this shape in the current mode (the split-arithmetic tiled kernel), else 0
int mx_pw_wgrad_tile_bnbwd_ok(int R, int Co, int Ci) { return mx_wgrad_uses_split(R, Co, Ci) ? 1 : 0; }

// dW[Co,Ci] += dZ[R,Co]^T X[R,Ci] with dZ = c1*G + c2*G2 + c3 per column (coef = [3][Co], as mx_bn_bwd_finalize leaves it) formed in
// the operand load: the weight-gradient half of the BatchNorm-backward fold (mx_pw_dgrad_bnbwd_planes is the other).  Shapes:
// mx_pw_wgrad_tile_bnbwd_ok; scratch: mx_pw_wgrad_tile_ws(R, Co, Ci, 0).
int mx_pw_wgrad_tile_bnbwd(const float* G, const float* G2, const float* coef, const float* X, float* dW, int R, int Co, int Ci,
                           int ldg, int ldx, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(G2 && coef, "wgrad_tile_bnbwd: null pointer");
  MX_CHECK_ARG((((uintptr_t)G2 | (uintptr_t)coef) & 15) == 0, "wgrad_tile_bnbwd: pointers must be 16-byte aligned");
  MX_CHECK_ARG(mx_wgrad_uses_split(R, Co, Ci), "wgrad_tile_bnbwd: shape R=%d Co=%d Ci=%d is not one the split kernel takes (mx_pw_wgrad_tile_bnbwd_ok)", R, Co, Ci);
  return wgrad_tile_impl(G, G2, coef, X, MX_PLAIN, nullptr, nullptr, nullptr, 1, dW, R, Co, Ci, ldg, ldx, ws, ws_bytes, stream);
}

// 1 when mx_pw_wgrad_tile_bnbwd_dz can leave dZ for this shape (the wave-specialised kernel takes it), else 0
int mx_pw_wgrad_tile_bnbwd_dz_ok(int R, int Co, int Ci, int ldg, int ldx) {
  WtPlan p;
  if (!mx_wgrad_uses_split(R, Co, Ci) || wt_pipe_mode() < 2 || !wt_plan(R, Co, Ci, MX_PLAIN, &p)) return 0;
  return (((long)p.rows_per_group + 96) * (ldg > ldx ? ldg : ldx) * 4 < (1l << 30) && (long)Co * Ci * 4 < (1l << 30)) ? 1 : 0;
}

// The same, and dZ[R, ldg] = c1*G + c2*G2 + c3 is also WRITTEN (round 5): the weight gradient runs FIRST and its loader waves leave the
// materialised operand for the data gradient (mx_pw_fwd_planes / mx_pw_dgrad on dZ), so the separate apply pass (mx_bn_bwd_apply:
// 2 reads + 1 write of the Cexp-wide tensors) is not launched at all.  dz must not alias G or G2.  Shapes: mx_pw_wgrad_tile_bnbwd_dz_ok.
int mx_pw_wgrad_tile_bnbwd_dz(const float* G, const float* G2, const float* coef, const float* X, float* dW, float* dz, int R, int Co,
                              int Ci, int ldg, int ldx, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(G2 && coef && dz, "wgrad_tile_bnbwd_dz: null pointer");
  MX_CHECK_ARG((((uintptr_t)G2 | (uintptr_t)coef | (uintptr_t)dz) & 15) == 0, "wgrad_tile_bnbwd_dz: pointers must be 16-byte aligned");
  MX_CHECK_ARG(dz != G && dz != G2, "wgrad_tile_bnbwd_dz: dz aliases an input (other workgroups still read it)");
  MX_CHECK_ARG(mx_pw_wgrad_tile_bnbwd_dz_ok(R, Co, Ci, ldg, ldx), "wgrad_tile_bnbwd_dz: shape R=%d Co=%d Ci=%d not taken (mx_pw_wgrad_tile_bnbwd_dz_ok)", R, Co, Ci);
  return wgrad_tile_impl(G, G2, coef, X, MX_PLAIN, nullptr, nullptr, nullptr, 1, dW, R, Co, Ci, ldg, ldx, ws, ws_bytes, stream, dz);
}

}  // extern "C"

static int wgrad_tile_impl(const float* G, const float* G2, const float* gcoef, const float* X, int x_mode, const float* x_scale,
                           const float* x_shift, const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg,
                           int ldx, void* ws, long ws_bytes, void* stream, float* dz) {
  MX_CHECK_ARG(G && X && dW && ws, "wgrad_tile: null pointer");
  MX_CHECK_ARG(((uintptr_t)dW & 15) == 0 && ((uintptr_t)ws & 15) == 0 && ((uintptr_t)G & 15) == 0 && ((uintptr_t)X & 15) == 0,
               "wgrad_tile: pointers must be 16-byte aligned");
  MX_CHECK_ARG(ldg % 4 == 0 && ldx % 4 == 0, "wgrad_tile: leading dimensions must be multiples of 4");
  MX_CHECK_ARG(x_mode == MX_PLAIN || (x_scale && x_shift && rows_per_sample > 0), "wgrad_tile: prologue needs scale/shift");
  WtPlan p;
  MX_CHECK_ARG(wt_plan(R, Co, Ci, x_mode, &p), "wgrad_tile: shape R=%d Co=%d Ci=%d not supported", R, Co, Ci);
  MX_CHECK_ARG(p.groups == 1 || ws_bytes >= (long)p.groups * Co * Ci * 4, "wgrad_tile: workspace too small");
  WtArgs a{};
  a.G = G; a.X = MxOperand{X, x_scale, x_shift, x_gate, x_mode, rows_per_sample};
  a.R = R; a.Co = Co; a.Ci = Ci; a.ldg = ldg; a.ldx = ldx;
  a.rows_per_group = p.rows_per_group; a.groups = p.groups; a.tiles_co = p.tiles_co; a.tiles_ci = p.tiles_ci;
  a.accumulate = p.groups == 1;
  a.order = wt_order();
  a.G2 = G2; a.gcoef = gcoef; a.dz = dz;
  a.part = a.accumulate ? dW : (float*)ws;
  hipStream_t st = (hipStream_t)stream;
  if (wt_use_split(Co, Ci)) {
    const dim3 grid(8 * cdiv(a.groups, 8) * a.tiles_co * a.tiles_ci);
    a.total = grid.x;
    // experiment knob: unused dynamic LDS caps the kernel's workgroups per CU (32 KB -> 2 per CU, 64 KB -> 1), leaving wave slots
    // and registers to the main stream's kernels it runs beside
    static const int pad = getenv("MX_WGRAD_SPLIT_LDS_PAD") ? atoi(getenv("MX_WGRAD_SPLIT_LDS_PAD")) : 0;
    // the pipelined kernel addresses a group through 32-bit buffer offsets: (rows + 64) * ld * 4 bytes must stay far below 2^31
    const int pipe = wt_pipe_mode();
    // (and the partial tile is stored through 32-bit buffer offsets as well: Co x Ci x 4 bytes)
    const bool pipe_ok = pipe && !a.G2 && a.X.mode == MX_PLAIN && ((long)a.rows_per_group + 96) * (a.ldg > a.ldx ? a.ldg : a.ldx) * 4 < (1l << 30) &&
                         (long)a.Co * a.Ci * 4 < (1l << 30);
    const bool ws_gbn = pipe >= 2 && a.G2 && a.X.mode == MX_PLAIN && ((long)a.rows_per_group + 96) * (a.ldg > a.ldx ? a.ldg : a.ldx) * 4 < (1l << 30) &&
                        (long)a.Co * a.Ci * 4 < (1l << 30);
    MX_CHECK_ARG(!a.dz || ws_gbn, "wgrad_tile_bnbwd: the dZ output exists in the wave-specialised kernel only (mx_set_wgrad_kernel 2, rows per group x ld < 2^28)");
    if (ws_gbn) {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_ws_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 4 * 144 * 16), 0);
      (void)once;
      const int items = a.groups * a.tiles_co * a.tiles_ci;
      hipLaunchKernelGGL(wgrad_split_ws_kernel<true>, dim3(items < 256 ? 8 * cdiv(items, 8) : 256), dim3(512), 12 * 4 * 144 * 16, st, a);
    } else if (pipe_ok && pipe >= 2) {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_ws_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 4 * 144 * 16), 0);
      (void)once;
      const int items = a.groups * a.tiles_co * a.tiles_ci;
      hipLaunchKernelGGL(wgrad_split_ws_kernel<false>, dim3(items < 256 ? 8 * cdiv(items, 8) : 256), dim3(512), 12 * 4 * 144 * 16, st, a);
    } else if (pipe_ok) {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_split_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 12 * 4 * 144 * 16), 0);
      (void)once;
      hipLaunchKernelGGL(wgrad_split_pipe_kernel, grid, dim3(256), 12 * 4 * 144 * 16, st, a);
    } else if (a.G2) hipLaunchKernelGGL((wgrad_split_kernel<MX_PLAIN, true>), grid, dim3(256), 0, st, a);
    else if (a.X.mode == MX_PLAIN) hipLaunchKernelGGL((wgrad_split_kernel<MX_PLAIN, false>), grid, dim3(256), pad, st, a);
    else if (a.X.mode == MX_BNACT) hipLaunchKernelGGL((wgrad_split_kernel<MX_BNACT, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_split_kernel<MX_AFFINE, false>), grid, dim3(256), 0, st, a);
  } else if (p.f32ws && ((long)a.rows_per_group + 96) * (a.ldg > a.ldx ? a.ldg : a.ldx) * 4 < (1l << 30) && (long)a.Co * a.Ci * 4 < (1l << 30)) {
    const int items = a.groups * a.tiles_co * a.tiles_ci;
    const dim3 grid(items < 256 ? 8 * cdiv(items, 8) : 256);
    constexpr int lds = WF32_NST * 2 * 32 * 512;
    if (p.te == 4 && p.tf == 4) {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_ws_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), 0);
      (void)once;
      hipLaunchKernelGGL((wgrad_f32_ws_kernel<2, 2>), grid, dim3(512), lds, st, a);
    } else if (p.te == 4) {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_ws_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), 0);
      (void)once;
      hipLaunchKernelGGL((wgrad_f32_ws_kernel<2, 1>), grid, dim3(512), lds, st, a);
    } else {
      static const int once = (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_f32_ws_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds), 0);
      (void)once;
      hipLaunchKernelGGL((wgrad_f32_ws_kernel<1, 2>), grid, dim3(512), lds, st, a);
    }
  } else if (p.te == 4 && p.tf == 4) wt_launch<4, 4>(a, st);
  else if (p.te == 4) wt_launch<4, 2>(a, st);
  else if (p.tf == 4) wt_launch<2, 4>(a, st);
  else wt_launch<2, 2>(a, st);
  MX_LAUNCH_CHECK();
  if (!a.accumulate) {
    hipLaunchKernelGGL(wgrad_parts_reduce_kernel, dim3(cdiv((long)Co * Ci, 64)), dim3(256), 0, st, (const float*)ws, p.groups, Co * Ci, dW);
    MX_LAUNCH_CHECK();
  }
  return MX_OK;
}
