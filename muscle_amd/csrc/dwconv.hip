// Depthwise k x k convolution (k = 3/5, stride 1/2, static TF-"same" padding) for NHWC fp32
// activations: forward, data gradient and weight gradient.
//
// Reference: MBConvBlock._depthwise_conv, src/efficientnet_pytorch/model.py:50-52,78 with
// Conv2dStaticSamePadding (utils.py:122-145): zero padding (pad_lo before, pad_hi after) is applied
// to the *activated* tensor, then a VALID convolution.  Weight layout [C,1,k,k].
//
// HBM-bound: every kernel stages one spatial tile x 32 channels in LDS (128 B per pixel, so global
// loads are whole lines), applying the producer's BatchNorm affine + SiLU while staging
// (MxOperand BNACT) so the activated tensor never exists in HBM.  Each thread produces OX
// neighbouring outputs for 4 channels (float4) from LDS.  The forward epilogue accumulates the
// per-channel sum / sum-of-squares that the following train-mode BatchNorm needs.
#include "common.h"

constexpr int CB = 32;       // channels per block
constexpr int C4B = CB / 4;  // float4 groups per pixel

struct DwArgs {
  const float* x;      // fwd: input [N,H,W,C]; bwd_data: dY [N,Ho,Wo,C]; bwd_weight: input
  const float* sc;     // BNACT prologue on x (null = plain)
  const float* sh;
  const float* w;      // [C,1,K,K]
  const float* dy;     // bwd_weight: dY
  const float* res;    // bwd_data: optional residual added to dX
  float* y;            // fwd: Y; bwd_data: dX; bwd_weight: dW [C,1,K,K] (+=)
  float* stats;        // fwd: partial statistics [N*tiles][2][C] or null
  const float* ps;     // fwd, inference: BN1 scale / shift of the OUTPUT (running statistics) ...
  const float* pb;
  float* pooled;       // ... and the SE squeeze sums pooled[n][c] = sum_hw swish(ps*y + pb), or null
  float* poolpart;     // per-group partial squeeze rows [gpp][N][C] and the arrival counters [N][channel chunks] (mx_last_arriver)
  unsigned* counters;
  int N, H, W, Ho, Wo, C, pad;
  int tiles_x, tiles_y;
  int tiles_per_block;   // bwd_weight
  int gpp;               // fwd: groups of tiles per sample (inference squeeze), 0 = groups run over the whole (sample, tile) sequence
};

__device__ __forceinline__ float4 dw_load(const DwArgs& a, const float* p, int c) {
  float4 v = ld4(p);
  if (a.sc) {
    float4 s = ld4(a.sc + c), t = ld4(a.sh + c);
    v.x = swishf_(s.x * v.x + t.x); v.y = swishf_(s.y * v.y + t.y);
    v.z = swishf_(s.z * v.z + t.z); v.w = swishf_(s.w * v.w + t.w);
  }
  return v;
}

template <int K>
__device__ __forceinline__ void dw_stage_weights(const DwArgs& a, float* wl, int c0, int tid) {
  // wl[tap][CB]  <-  w[c][tap]
  for (int i = tid; i < K * K * CB; i += 256) {
    int cc = i % CB, tap = i / CB;
    wl[i] = (c0 + cc < a.C) ? a.w[(long)(c0 + cc) * K * K + tap] : 0.f;
  }
}

// Stage one input tile in two halves: dw_tile_load issues all global loads of a thread (the tile needs <= 12 float4 per
// thread), so a workgroup has its whole tile in flight instead of one line per wave; dw_tile_store applies the producer's
// BatchNorm affine + SiLU and writes the LDS tile.  Split so that the tile loop of the forward can request tile t + 1 before it
// computes tile t.
template <int K, int S, int TH, int TW>
struct DwTile {
  static constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  static constexpr int TOT = IH * IW * C4B, PER = (TOT + 255) / 256;
  float4 v[PER];
  unsigned ok;
  __device__ __forceinline__ void load(const DwArgs& a, int n, int oy0, int ox0, int c0, int tid) {
    const int iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;
    const int c = c0 + 4 * (tid % C4B);
    ok = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = tid + 256 * k;
      int pix = i / C4B;
      int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
      v[k] = make_float4(0, 0, 0, 0);
      if (i < TOT && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.C) {
        v[k] = ld4(a.x + (((long)n * a.H + iy) * a.W + ix) * a.C + c);
        ok |= 1u << k;
      }
    }
  }
  __device__ __forceinline__ void store(const DwArgs& a, float4* tile, int c0, int tid) const {
    const int c = c0 + 4 * (tid % C4B);
    float4 s = make_float4(0, 0, 0, 0), t = s;
    if (a.sc && c < a.C) { s = ld4(a.sc + c); t = ld4(a.sh + c); }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = tid + 256 * k;
      if (i < TOT) {
        float4 x = v[k];
        if (a.sc && ((ok >> k) & 1u)) {
          x.x = swishf_(s.x * x.x + t.x); x.y = swishf_(s.y * x.y + t.y);
          x.z = swishf_(s.z * x.z + t.z); x.w = swishf_(s.w * x.w + t.w);
        }
        tile[i] = x;
      }
    }
  }
};

template <int K, int S, int TH, int TW>
__device__ __forceinline__ void dw_stage_input(const DwArgs& a, float4* tile, int n, int oy0, int ox0, int c0, int tid) {
  DwTile<K, S, TH, TW> t;
  t.load(a, n, oy0, ox0, c0, tid);
  t.store(a, tile, c0, tid);
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
// A workgroup walks `tiles_per_block` consecutive tiles of one channel chunk (weights staged once) and leaves ONE partial: the
// statistics row of its tiles (training) or its share of a sample's squeeze sum (inference).  One tile per workgroup, the
// round-1/2 form, cost a partial row and - with the ordered squeeze of round 3 - a write-through store, a drain and an arrival
// ticket per TILE: the inference forward's depthwise kernels ran 45 % over their training twins.
//   gpp == 0 (training): group g covers tiles [g tpb, (g+1) tpb) of the (sample, tile) sequence, statistics row g;
//   gpp  > 0 (inference squeeze): a sample's tiles are cut into gpp groups that do not straddle samples; group (n, gi) parks its
//            partial squeeze row and the last of the gpp groups to arrive adds them in group order (mx_last_arriver).
// Software pipelining of the tile loop (tile t + 1's global loads in flight while tile t is computed; -DDW_FWD_PREFETCH=1): measured
// in round 3 and OFF - the 40 staging registers it keeps live cost a wave per SIMD (4 -> 3), 3x3 loses 10-20 %, 5x5 gains 5-12 % on the
// widest layers only and loses on the rest; whole step 109.0 -> 110.6 ms.  Occupancy, not intra-workgroup overlap, feeds these kernels.
#ifndef DW_FWD_PREFETCH
#define DW_FWD_PREFETCH 0
#endif
// ROWS (28-wide stride-1 tiles, round 4): as in dw_bwd_fused_kernel the staged rows are pitched at 32 pixels, so one staging pass of
// the 256 threads is one tile row - the row's address, clamp and bounds are scalar, the thread's column is fixed for the tile - and a
// thread computes 7 pixels x 4 channels (11 tile reads per kernel row for 70 packed FMAs; 8 for 40 at 4 pixels).  The 16-wide form
// spent ~75 instructions per output element against ~26 here, on a kernel whose VALU is 80 % busy at four waves per SIMD (PMC).
template <int K, int S, int TH, int TW, int OX>
__global__ __launch_bounds__(256, (TW == 28) ? 3 : 1) void dw_fwd_kernel(DwArgs a) {
  constexpr bool ROWS = (TW == 28 && S == 1);
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K, IWP = ROWS ? 32 : IW;
  static_assert(TH * (TW / OX) * C4B == 256, "thread mapping");
  static_assert(!ROWS || (IW <= 32 && 256 / C4B == 32), "one padded tile row per staging pass");
  __shared__ float4 tile[IH * IWP * C4B];
  __shared__ __attribute__((aligned(16))) float wl[K * K * CB];
  __shared__ __attribute__((aligned(16))) float red[4][2 * CB];     // one row per wave: the four waves are added in wave order (LDS atomics gave sums
  __shared__ unsigned last_flag;       // whose last bit depended on which wave came first)
  const int tid = threadIdx.x;
  const int ntile = a.tiles_x * a.tiles_y;
  const int c0 = blockIdx.y * CB;
  long t_beg, t_end;
  int n_fix = 0, gi = 0;
  if (a.gpp > 0) {
    n_fix = blockIdx.x / a.gpp; gi = blockIdx.x % a.gpp;
    t_beg = (long)n_fix * ntile + (long)gi * a.tiles_per_block;
    t_end = min((long)(n_fix + 1) * ntile, t_beg + a.tiles_per_block);
  } else {
    t_beg = (long)blockIdx.x * a.tiles_per_block;
    t_end = min((long)a.N * ntile, t_beg + a.tiles_per_block);
  }
  dw_stage_weights<K>(a, wl, c0, tid);
  const int c4 = tid % C4B, q = tid / C4B;
  const int oyl = q / (TW / OX), oxl = (q % (TW / OX)) * OX;
  const int c = c0 + 4 * c4;
  const int wave = tid >> 6;
  float4 s = make_float4(0, 0, 0, 0), sq = make_float4(0, 0, 0, 0), p = make_float4(0, 0, 0, 0);
  float4 psc = make_float4(0, 0, 0, 0), psh = psc;
  if (a.pooled && c < a.C) { psc = ld4(a.ps + c); psh = ld4(a.pb + c); }
  // software pipeline over the tiles: tile t + 1's global loads are in flight while tile t is computed from LDS
  DwTile<K, S, TH, TW> stg;
  if (DW_FWD_PREFETCH && t_beg < t_end) {
    const int rem0 = (int)(t_beg % ntile);
    stg.load(a, (int)(t_beg / ntile), (rem0 / a.tiles_x) * TH, (rem0 % a.tiles_x) * TW, c0, tid);
  }
  for (long t = t_beg; t < t_end; ++t) {
    const int n = (int)(t / ntile), rem = (int)(t % ntile);
    const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
    if constexpr (ROWS) {
      const int col = tid / C4B, ix = ox0 - a.pad + col;
      const bool colok = col < IW && ix >= 0 && ix < a.W && c < a.C;
      const int voff = min(max(ix, 0), a.W - 1) * a.C + (c < a.C ? c : 0);
      float4 sv = make_float4(0, 0, 0, 0), tv = sv;
      if (a.sc && c < a.C) { sv = ld4(a.sc + c); tv = ld4(a.sh + c); }
      float4 v[IH];
#pragma unroll
      for (int r = 0; r < IH; ++r) {
        const int iy = oy0 - a.pad + r;                                  // wave-uniform
        v[r] = ld4(a.x + (((long)n * a.H + min(max(iy, 0), a.H - 1)) * a.W) * a.C + voff);
      }
      __syncthreads();                             // the previous tile's readers are done (first pass: the weights are staged)
#pragma unroll
      for (int r = 0; r < IH; ++r) {
        const int iy = oy0 - a.pad + r;
        float4 x = v[r];
        if (a.sc) {
          x.x = swishf_(sv.x * x.x + tv.x); x.y = swishf_(sv.y * x.y + tv.y);
          x.z = swishf_(sv.z * x.z + tv.z); x.w = swishf_(sv.w * x.w + tv.w);
        }
        const bool ok = colok && iy >= 0 && iy < a.H;                    // the convolution pads the ACTIVATED tensor with zeros
        tile[r * 256 + tid] = ok ? x : make_float4(0, 0, 0, 0);
      }
      __syncthreads();
    } else {
    if (!DW_FWD_PREFETCH) stg.load(a, n, oy0, ox0, c0, tid);
    __syncthreads();                               // the previous tile's readers are done (first pass: the weights are staged)
    stg.store(a, tile, c0, tid);
    __syncthreads();
    }
    if (DW_FWD_PREFETCH && t + 1 < t_end) {
      const int rem1 = (int)((t + 1) % ntile);
      stg.load(a, (int)((t + 1) / ntile), (rem1 / a.tiles_x) * TH, (rem1 % a.tiles_x) * TW, c0, tid);
    }
    float4 acc[OX];
#pragma unroll
    for (int o = 0; o < OX; ++o) acc[o] = make_float4(0, 0, 0, 0);
    // one kernel row at a time (8-12 input float4 live): fully unrolling ky for k=5 took 256 VGPRs = 1 wave/SIMD
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      float4 in[(OX - 1) * S + K];
#pragma unroll
      for (int j = 0; j < (OX - 1) * S + K; ++j) in[j] = tile[((oyl * S + ky) * IWP + oxl * S + j) * C4B + c4];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float4 w = ld4(wl + (ky * K + kx) * CB + 4 * c4);
#pragma unroll
        for (int o = 0; o < OX; ++o) {
          float4 v = in[o * S + kx];
          acc[o].x += w.x * v.x; acc[o].y += w.y * v.y; acc[o].z += w.z * v.z; acc[o].w += w.w * v.w;
        }
      }
    }
    const int oy = oy0 + oyl;
    if (c < a.C && oy < a.Ho) {
      float* yrow = a.y + (((long)n * a.Ho + oy) * a.Wo + ox0 + oxl) * a.C + c;     // one 64-bit base, the pixels C floats apart
#pragma unroll
      for (int o = 0; o < OX; ++o) {
        int ox = ox0 + oxl + o;
        if (ox < a.Wo) {
          st4(yrow + o * a.C, acc[o]);
          if (a.pooled) {
            // Inference (eval-mode BatchNorm: its affine is known before the batch is seen): the squeeze of the SE block,
            // sum over the image of swish(bn1(y)), leaves with the tiles instead of costing a second pass over d (model.py:81-82)
            p.x += swishf_(psc.x * acc[o].x + psh.x); p.y += swishf_(psc.y * acc[o].y + psh.y);
            p.z += swishf_(psc.z * acc[o].z + psh.z); p.w += swishf_(psc.w * acc[o].w + psh.w);
          } else {
            s.x += acc[o].x; s.y += acc[o].y; s.z += acc[o].z; s.w += acc[o].w;
            sq.x += acc[o].x * acc[o].x; sq.y += acc[o].y * acc[o].y; sq.z += acc[o].z * acc[o].z; sq.w += acc[o].w * acc[o].w;
          }
        }
      }
    }
  }
  if (a.pooled) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      p.x += __shfl_xor(p.x, o, 64); p.y += __shfl_xor(p.y, o, 64); p.z += __shfl_xor(p.z, o, 64); p.w += __shfl_xor(p.w, o, 64);
    }
    __syncthreads();
    if ((tid & 63) < C4B) st4(&red[wave][4 * c4], p);
    __syncthreads();
    const bool own = tid < CB && c0 + tid < a.C;
    const float v = own ? ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid] : 0.f;
    if (a.gpp == 1) {
      if (own) a.pooled[(long)n_fix * a.C + c0 + tid] = v;
      return;
    }
    // the groups of a sample are joined by the last workgroup to arrive, in group order (no atomics: the squeeze, and with it
    // the whole eval forward, gives the same bits in every run; across batch sizes only while dw_fwd_geom cuts a sample into the
    // same groups - tiles per group follows target / N - which tests/test_gpu_determinism.py checks for the sizes it runs)
    if (own) mx_st_wt(a.poolpart + ((long)gi * a.N + n_fix) * a.C + c0 + tid, v);
    if (!mx_last_arriver(a.counters + n_fix * gridDim.y + blockIdx.y, a.gpp, &last_flag)) return;
    if (own) {
      float u = a.poolpart[(long)n_fix * a.C + c0 + tid];
      for (int g = 1; g < a.gpp; ++g) u += a.poolpart[((long)g * a.N + n_fix) * a.C + c0 + tid];
      a.pooled[(long)n_fix * a.C + c0 + tid] = u;
    }
    return;
  }
  if (a.stats) {
    // lanes with equal c4 sit 8 apart: fold the wave first, then one LDS row per wave, added in wave order
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      s.x += __shfl_xor(s.x, o, 64); s.y += __shfl_xor(s.y, o, 64); s.z += __shfl_xor(s.z, o, 64); s.w += __shfl_xor(s.w, o, 64);
      sq.x += __shfl_xor(sq.x, o, 64); sq.y += __shfl_xor(sq.y, o, 64); sq.z += __shfl_xor(sq.z, o, 64); sq.w += __shfl_xor(sq.w, o, 64);
    }
    __syncthreads();
    if ((tid & 63) < C4B) { st4(&red[wave][4 * c4], s); st4(&red[wave][CB + 4 * c4], sq); }
    __syncthreads();
    if (tid < 2 * CB) {
      const int cc = tid % CB;
      if (c0 + cc < a.C) {
        float* prow = a.stats + (long)blockIdx.x * 2 * a.C;                      // one partial row per workgroup
        prow[(tid / CB) * a.C + c0 + cc] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// data gradient: dX[n,iy,ix,c] = sum_{ky,kx} dY[n,oy,ox,c] * w[c,ky,kx],  oy*S - pad + ky == iy
// tile over input pixels; dY staged with its halo.
// ---------------------------------------------------------------------------
template <int K, int S, int TH, int TW>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(DwArgs a) {
  // output rows that can touch input rows [iy0, iy0+TH): oy in [ceil((iy0+pad-K+1)/S), floor((iy0+TH-1+pad)/S)]
  constexpr int OH = (TH + K - 2) / S + 2, OW = (TW + K - 2) / S + 2;
  static_assert(TH * TW * C4B == 256 * 4, "4 input pixels per thread");
  __shared__ float4 tile[OH * OW * C4B];
  __shared__ __attribute__((aligned(16))) float wl[K * K * CB];
  const int tid = threadIdx.x;
  const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
  const int c0 = blockIdx.y * CB, n = blockIdx.z;
  const int iy0 = ty * TH, ix0 = tx * TW;
  // floor division for possibly negative numerators
  auto fdiv = [](int p, int q) { return (p >= 0) ? p / q : -((-p + q - 1) / q); };
  const int oy_lo = fdiv(iy0 + a.pad - K + 1 + S - 1, S), ox_lo = fdiv(ix0 + a.pad - K + 1 + S - 1, S);
  dw_stage_weights<K>(a, wl, c0, tid);
  {
    constexpr int TOT = OH * OW * C4B, PER = (TOT + 255) / 256;
    float4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = tid + 256 * k;
      int c4s = i % C4B, pix = i / C4B;
      int oy = oy_lo + pix / OW, ox = ox_lo + pix % OW;
      int cs = c0 + 4 * c4s;
      v[k] = make_float4(0, 0, 0, 0);
      if (i < TOT && oy >= 0 && oy < a.Ho && ox >= 0 && ox < a.Wo && cs < a.C)
        v[k] = ld4(a.x + (((long)n * a.Ho + oy) * a.Wo + ox) * a.C + cs);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      int i = tid + 256 * k;
      if (i < TOT) tile[i] = v[k];
    }
  }
  __syncthreads();
  const int c4 = tid % C4B, q = tid / C4B;   // q in [0,32)
  const int c = c0 + 4 * c4;
#pragma unroll
  for (int rep = 0; rep < 4; ++rep) {
    const int p = q + 32 * rep;
    const int iyl = p / TW, ixl = p % TW;
    const int iy = iy0 + iyl, ix = ix0 + ixl;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int ny = iy + a.pad - ky;
      if (ny % S != 0 || ny < 0) continue;
      const int oy = ny / S - oy_lo;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int nx = ix + a.pad - kx;
        if (nx % S != 0 || nx < 0) continue;
        const int ox = nx / S - ox_lo;
        // (oy, ox) is inside the staged tile by construction; out-of-range outputs were staged as 0
        float4 w = ld4(wl + (ky * K + kx) * CB + 4 * c4);
        float4 v = tile[(oy * OW + ox) * C4B + c4];
        acc.x += w.x * v.x; acc.y += w.y * v.y; acc.z += w.z * v.z; acc.w += w.w * v.w;
      }
    }
    if (c < a.C && iy < a.H && ix < a.W) {
      const long o = (((long)n * a.H + iy) * a.W + ix) * a.C + c;
      if (a.res) { float4 r = ld4(a.res + o); acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w; }
      st4(a.y + o, acc);
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient: dW[c,ky,kx] += sum_{n,oy,ox} dY[n,oy,ox,c] * act(X)[n,oy*S-pad+ky,ox*S-pad+kx,c]
// each block walks `tiles_per_block` (n, tile) pairs of one channel chunk with its partials in
// registers, then leaves through LDS + one round of fp32 atomics.
// ---------------------------------------------------------------------------
template <int K, int S, int TH, int TW, int OX>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(DwArgs a) {
  constexpr int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
  static_assert(TH * (TW / OX) * C4B == 256, "thread mapping");
  __shared__ float4 tile[IH * IW * C4B];
  static_assert(IH * IW * C4B * 4 >= 4 * K * K * CB, "the per-wave partial rows reuse the tile");
  const int tid = threadIdx.x;
  const int c0 = blockIdx.y * CB;
  const int c4 = tid % C4B, q = tid / C4B;
  const int oyl = q / (TW / OX), oxl = (q % (TW / OX)) * OX;
  const int c = c0 + 4 * c4;
  float4 part[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) part[t] = make_float4(0, 0, 0, 0);
  const long ntiles = (long)a.N * a.tiles_x * a.tiles_y;
  const long t_beg = (long)blockIdx.x * a.tiles_per_block;
  const long t_end = min(ntiles, t_beg + a.tiles_per_block);
  for (long t = t_beg; t < t_end; ++t) {
    const int n = (int)(t / (a.tiles_x * a.tiles_y));
    const int rem = (int)(t % (a.tiles_x * a.tiles_y));
    const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
    __syncthreads();
    dw_stage_input<K, S, TH, TW>(a, tile, n, oy0, ox0, c0, tid);
    __syncthreads();
    const int oy = oy0 + oyl;
    float4 g[OX];
#pragma unroll
    for (int o = 0; o < OX; ++o) {
      int ox = ox0 + oxl + o;
      g[o] = (c < a.C && oy < a.Ho && ox < a.Wo) ? ld4(a.dy + (((long)n * a.Ho + oy) * a.Wo + ox) * a.C + c)
                                                  : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      float4 in[(OX - 1) * S + K];
#pragma unroll
      for (int j = 0; j < (OX - 1) * S + K; ++j) in[j] = tile[((oyl * S + ky) * IW + oxl * S + j) * C4B + c4];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
#pragma unroll
        for (int o = 0; o < OX; ++o) {
          float4 v = in[o * S + kx];
          float4& p = part[ky * K + kx];
          p.x += g[o].x * v.x; p.y += g[o].y * v.y; p.z += g[o].z * v.z; p.w += g[o].w * v.w;
        }
      }
    }
  }
  // leave: fold each wave, one partial row per wave in the (now free) tile memory, the four rows added in wave order
  // (LDS atomics here made the last bit of dW depend on which wave arrived first)
  __syncthreads();
  float* slots = reinterpret_cast<float*>(tile);          // [4][K*K*CB]
  const int wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    float4 p = part[t];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      p.x += __shfl_xor(p.x, o, 64); p.y += __shfl_xor(p.y, o, 64); p.z += __shfl_xor(p.z, o, 64); p.w += __shfl_xor(p.w, o, 64);
    }
    if ((tid & 63) < C4B) st4(slots + (wave * K * K + t) * CB + 4 * c4, p);
  }
  __syncthreads();
  for (int i = tid; i < K * K * CB; i += 256) {
    int cc = i % CB, tap = i / CB;
    if (c0 + cc >= a.C) continue;
    const float v = ((slots[i] + slots[K * K * CB + i]) + slots[2 * K * K * CB + i]) + slots[3 * K * K * CB + i];
    // a.stats = scratch [gridDim.x][C*K*K]: one partial row per workgroup, added by dw_parts_reduce_kernel (up to 228
    // workgroups per channel chunk used to add into the same 16*K*K addresses: contended atomics, 323 us for a 5x5 layer)
    a.stats[(long)blockIdx.x * a.C * K * K + (long)(c0 + cc) * K * K + tap] = v;
  }
}


// ---------------------------------------------------------------------------
// Fused stride-1 backward of  [BN0+SiLU] -> depthwise -> BN1 -> SiLU -> SE gate  in ONE pass over the tensors:
//   * the BatchNorm-1 data gradient dd = c1*g + c2*d + c3, g = (dA*gate + add) * swish'(a1*d + b1), is formed while the
//     tile is staged (it is never written to HBM),
//   * weight gradient dW[c,ky,kx] += sum dd * act(X) and data gradient gX = dwconv^T(dd) are both computed from the
//     two staged tiles (same tile geometry: odd kernel, symmetric pad),
//   * for expand blocks the epilogue multiplies by swish'(a0*x + b0) and accumulates the BatchNorm-0 backward sums
//     (sum g, sum g*x) as one partial row per workgroup, so no separate reduction pass over (gX, X) is needed.
// Replaces bn_bwd_apply (2 reads + 1 write), dw_bwd_data, dw_bwd_weight and the BN0 reduction (2 reads).
// ---------------------------------------------------------------------------
struct DwFusedArgs {
  const float* dA; const float* d;                    // [N,H,W,C] each (stride 1: output size == input size)
  const float* gate; const float* add;                // [N,C]
  const float* a1; const float* b1;                   // BN1 scale / shift
  const float* c1; const float* c2; const float* c3;  // BN1 backward coefficients
  const float* x; const float* a0; const float* b0;   // dw input (raw) and its BN0 scale / shift (null = plain input)
  const float* w; const float* res;                   // weights [C,1,K,K]; residual added to gX (plain-input case)
  float* gx; float* dwpart; float* part;              // outputs: gX (or g*swish'), dW partial rows [groups][C*K*K], BN0 partial sums [groups][2][C]
  int N, H, W, C, pad, tiles_x, tiles_y, tiles_per_block;
  int xcd, gpp, chunks;    // xcd = 1: 1-D grid, the gpp tile groups of one (sample, channel chunk) plane run on ONE XCD (dw_fwd_kernel)
  unsigned* fin_counters;  // non-null: the last workgroup of a channel chunk to arrive finishes the BatchNorm-0 backward statistics itself
  BnBwdFin fin;            //   (dgamma / dbeta +=, c1 c2 c3) from the partial rows - the separate bn_reduce_finalize launch is gone
};

// Tile shapes (TH x TW output pixels per staged tile, PX consecutive pixels per thread "group", 16 groups per pass):
//   8 x 16, PX 8  - one group per thread; 30 KB of LDS for 5x5: the round-1/2 shape, best where the image is not a multiple of 14/28
//   14 x 28, PX 7 - four passes of groups per thread; 72 KB for 5x5 (2 workgroups per CU).  B7's stride-1 layers are 224 / 112 / 56 /
//                   28 pixels wide: 28 = 2 x 14 rows x 1 x 28 columns EXACTLY, where 8 x 16 tiles cover 32 x 32 (1.31x the pixels) and stage
//                   8 x 240 elements for 784 outputs (2.45x); 14 x 28 stages 2 x 576 (1.47x): -40 % staging work and dA / D reads.
// an optimisation barrier on one float2 (kept as a 64-bit register pair, so v_pk_* survives): the value is materialised HERE
__device__ __forceinline__ void pin2(float2& v) {
  double d = __builtin_bit_cast(double, v);
  asm volatile("" : "+v"(d));
  v = __builtin_bit_cast(float2, d);
}

#ifdef MX_DW_STAMPS
// diagnostic build only (-DMX_DW_STAMPS, never shipped): wall-clock stamps (s_memrealtime, 10 ns) of thread 0 of the first 64 workgroups
// around the phases of each tile: [wg][tile][4] = tile start, staging issued+transformed, compute start (after the barrier), compute end
__device__ unsigned long long mx_dw_stamps[64 * 32 * 4];
extern "C" int mx_dw_stamps_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(mx_dw_stamps), sizeof(mx_dw_stamps));
}
#define DW_STAMP(k) do { if (tid == 0 && blockIdx.x < 64 && blockIdx.y == 0 && (t - t_beg) < 32) \
    mx_dw_stamps[(blockIdx.x * 32 + (int)(t - t_beg)) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DW_STAMP(k) do {} while (0)
#endif

#ifndef MX_DW_STAGE_MAX
#define MX_DW_STAGE_MAX 4
#endif
// float4 pairs (dA, d) a thread has in flight per staging chunk: the largest divisor of its share of the tile up to MX_DW_STAGE_MAX
constexpr int dw_stage_depth(int per, int th) {
  int best = 1;
  for (int c = 2; c <= (th == 8 ? 4 : MX_DW_STAGE_MAX); ++c) if (per % c == 0) best = c;   // (8 x 16 at 3 waves per SIMD spills above 4)
  return best;
}

template <int K, int TH, int TW, int PX>
__global__ __launch_bounds__(256, (TH == 7) ? 3 : (K == 5 || TH != 8) ? 2 : 3) void dw_bwd_fused_kernel(DwFusedArgs a) {
  // Only dd is staged with its halo.  The weight gradient is taken over INPUT pixels,
  //     dW[ky,kx] = sum_q act(X)[q] * dd[q - (ky,kx) + pad],
  // so the activated input is needed at the tile's centre pixels only: each thread reads its 2 channels x PX pixels of X
  // straight into registers (1.0x instead of the halo's 1.9x, no LDS), reuses the raw values for the swish'(bn0(x))
  // factor and the BN0 sums of the epilogue, and the dW and dX loops walk the SAME shifted dd rows, so every LDS read
  // feeds both.  One staged array instead of two (round 1 PMC: the two-array kernel sat at SQ_WAIT_ANY 45-60 %, 1.5-1.9 TB/s).
  // ROWS (28-pixel-wide tiles): the staged rows are pitched at 32 pixels, so that the 256 threads' float4s of one staging pass are
  // exactly one tile row - row index wave-uniform (scalar address part, scalar bounds), column fixed per thread for the whole tile
  constexpr bool ROWS = (TW == 28);
  constexpr int IH = TH + K - 1, IW = TW + K - 1, IWP = ROWS ? 32 : IW, TOT = IH * IWP * C4B, PER = (TOT + 255) / 256;
  static_assert(!ROWS || (IW <= 32 && PER == IH && 256 / C4B == 32), "one padded tile row per staging pass");
  constexpr int CH = (ROWS && TH == 7) ? 4 : dw_stage_depth(PER, TH);     // (7-row tiles: 11 or 9 staged rows in chunks of 4, the last one short)
  static_assert(ROWS || PER % CH == 0, "staging chunks");
  __shared__ float4 td[TOT + C4B];     // dd with halo (+ one pixel of zeros: the masked eighth slot of an odd PX reads past the last row); at the end the per-wave partial rows of dW and of the BN0 sums
  static_assert(TOT * 4 >= 4 * K * K * CB, "the per-wave partial rows reuse the staged tile");
  __shared__ __attribute__((aligned(16))) float wl[K * K * CB];
  __shared__ __attribute__((aligned(16))) float cst[9 * CB];   // a1 b1 c1 c2 c3 a0 b0 | gate add (per tile)
  // workgroup -> (group of tiles, channel chunk); XCD-aware ids as in dw_fwd_kernel when a plane is cut into several groups
  int grp_id = blockIdx.x, chunk_id = blockIdx.y;
  long t_beg, t_end;
  const long ntiles = (long)a.N * a.tiles_x * a.tiles_y;
  if (a.xcd) {
    const int ntile = a.tiles_x * a.tiles_y;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int plane = (j / a.gpp) * 8 + xcd, gi = j % a.gpp;
    if (plane >= a.chunks * a.N) return;
    chunk_id = plane % a.chunks;
    const int n_ = plane / a.chunks;
    grp_id = n_ * a.gpp + gi;
    t_beg = (long)n_ * ntile + (long)gi * a.tiles_per_block;
    t_end = min((long)(n_ + 1) * ntile, t_beg + a.tiles_per_block);
  } else {
    t_beg = (long)blockIdx.x * a.tiles_per_block;
    t_end = min(ntiles, t_beg + a.tiles_per_block);
  }
  const int tid = threadIdx.x, c0 = chunk_id * CB;
  const int c4 = tid % C4B;                        // staging: 4 channels per thread
  const int c = c0 + 4 * c4;
  const bool cok = c < a.C;
  constexpr int C2B = CB / 2;                      // compute: 2 channels x PX pixels per thread and pass
  constexpr int GX = TW / PX, NGRP = TH * GX, NG = (NGRP + 15) / 16;   // pixel groups per tile; passes of 16 groups
  static_assert(TW % PX == 0 && PX >= 5 && PX <= 8, "compute mapping");
  const int c2 = tid % C2B, pq = tid / C2B;
  const int cc2 = c0 + 2 * c2;
  const bool cok2 = cc2 < a.C;
  const bool has_bn0 = a.a0 != nullptr;
  if (tid < C4B) td[TOT + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
  // weights wl[tap][CB]
  for (int i = tid; i < K * K * CB; i += 256) {
    int cc = i % CB, tap = i / CB;
    wl[i] = (c0 + cc < a.C) ? a.w[(long)(c0 + cc) * K * K + tap] : 0.f;
  }
  // per-channel constants live in LDS (9 float4 per thread would otherwise sit in VGPRs for the whole tile loop)
  for (int i = tid; i < 7 * CB; i += 256) {
    int cc = i % CB, j = i / CB;
    const float* src = j == 0 ? a.a1 : j == 1 ? a.b1 : j == 2 ? a.c1 : j == 3 ? a.c2 : j == 4 ? a.c3 : j == 5 ? a.a0 : a.b0;
    cst[i] = (src && c0 + cc < a.C) ? src[c0 + cc] : 0.f;
  }
  float2 part[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) part[t] = make_float2(0, 0);
  float2 s0 = make_float2(0, 0), s1 = s0;
  // this thread's centre pixels of X (raw) for one group: clamped addresses, masked at use.  One 64-bit row base; the pixel
  // offsets are 32-bit (a row of one sample is far below 2^31 floats) and advance by C with a clamp at the last column
  auto load_x = [&](auto& xr, int n, int oy, int oxb) {
    const float* row = a.x + (((long)n * a.H + min(oy, a.H - 1)) * a.W) * a.C + (cok2 ? cc2 : 0);
    const int last = (a.W - 1) * a.C;
    int off = min(oxb, a.W - 1) * a.C;
#pragma unroll
    for (int o = 0; o < PX; ++o) {
      xr[o] = *reinterpret_cast<const float2*>(row + off);
      off = min(off + a.C, last);
    }
  };
  for (long t = t_beg; t < t_end; ++t) {
    const int n = (int)(t / (a.tiles_x * a.tiles_y));
    const int rem = (int)(t % (a.tiles_x * a.tiles_y));
    const int oy0 = (rem / a.tiles_x) * TH, ox0 = (rem % a.tiles_x) * TW;
    const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
    DW_STAMP(0);
    __syncthreads();
    if (tid < 2 * CB) {
      int cc = tid % CB;
      cst[7 * CB + tid] = (c0 + cc < a.C) ? (tid < CB ? a.gate : a.add)[(long)n * a.C + c0 + cc] : 0.f;
    }
    // the first group's centre pixels are requested before the staging loop so they land under it
    float2 xr[PX];
    load_x(xr, n, oy0 + pq / GX, ox0 + (pq % GX) * PX);
    __syncthreads();
    // stage dd, CH float4 pairs in flight per thread; out-of-image / out-of-range elements read a clamped (valid) address and
    // are multiplied by 0
#define DD1(f) dd.f = okf * (C1.f * ((vg[k].f * G.f + AD.f) * swish_gradf_(A1.f * vd[k].f + B1.f)) + C2.f * vd[k].f + C3.f);
    if constexpr (ROWS) {
      const int col = tid / C4B, ix = ix0 + col;
      const float colf = (col < IW && ix >= 0 && ix < a.W && cok) ? 1.f : 0.f;     // (columns IW..31 of the pitch are stored as zeros)
      const int voff = min(max(ix, 0), a.W - 1) * a.C + (cok ? c : 0);
#pragma unroll 1
      for (int r0 = 0; r0 < IH; r0 += CH) {
        asm volatile("" ::: "memory");   // keep the constant reloads inside the loop
        float4 vg[CH], vd[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const long rb = (((long)n * a.H + min(max(iy0 + min(r0 + k, IH - 1), 0), a.H - 1)) * a.W) * a.C;     // wave-uniform
          vg[k] = ld4(a.dA + rb + voff); vd[k] = ld4(a.d + rb + voff);
        }
        const float4 A1 = ld4(cst + 4 * c4), B1 = ld4(cst + CB + 4 * c4), C1 = ld4(cst + 2 * CB + 4 * c4),
                     C2 = ld4(cst + 3 * CB + 4 * c4), C3 = ld4(cst + 4 * CB + 4 * c4),
                     G = ld4(cst + 7 * CB + 4 * c4), AD = ld4(cst + 8 * CB + 4 * c4);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int iy = iy0 + r0 + k;
          const float okf = (iy >= 0 && iy < a.H) ? colf : 0.f;
          float4 dd;
          DD1(x) DD1(y) DD1(z) DD1(w)
          if (IH % CH == 0 || r0 + k < IH) td[(r0 + k) * 256 + tid] = dd;
        }
      }
    } else {
#pragma unroll 1
      for (int k0 = 0; k0 < PER; k0 += CH) {
        asm volatile("" ::: "memory");   // keep the constant reloads inside the loop
        float4 vg[CH], vd[CH];
        unsigned okm = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          int i = min(tid + 256 * (k0 + k), TOT - 1), pix = i / C4B;
          int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
          okm |= (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && cok) ? (1u << k) : 0u;
          iy = min(max(iy, 0), a.H - 1); ix = min(max(ix, 0), a.W - 1);
          const long off = (((long)n * a.H + iy) * a.W + ix) * a.C + (cok ? c : 0);
          vg[k] = ld4(a.dA + off); vd[k] = ld4(a.d + off);
        }
        const float4 A1 = ld4(cst + 4 * c4), B1 = ld4(cst + CB + 4 * c4), C1 = ld4(cst + 2 * CB + 4 * c4),
                     C2 = ld4(cst + 3 * CB + 4 * c4), C3 = ld4(cst + 4 * CB + 4 * c4),
                     G = ld4(cst + 7 * CB + 4 * c4), AD = ld4(cst + 8 * CB + 4 * c4);
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const int i = min(tid + 256 * (k0 + k), TOT - 1);   // duplicates of the last element rewrite the same value
          float4 dd;
          const float okf = ((okm >> k) & 1u) ? 1.f : 0.f;
          DD1(x) DD1(y) DD1(z) DD1(w)
          td[i] = dd;
        }
      }
    }
#undef DD1
    DW_STAMP(1);
    __syncthreads();
    DW_STAMP(2);
    // Compute: a thread owns 2 channels x PX consecutive pixels of one tile row per pass (4 channels x 4 pixels needs 100
    // accumulator VGPRs for K=5 and the allocator then spills; 2 x 8 needs 50 and keeps v_pk_fma_f32 over the channel pair).
    // Row ky of the kernel pairs this pixel row with dd row pyl + K-1-ky of the halo tile, for the weight gradient
    // (times act(X) of the centre pixel) and for the data gradient (times the flipped weight) alike.  The pixels go in two
    // halves of QX (each half: taps, then its own epilogue and stores - no accumulator survives a half), both unrolled, so
    // every register index and every LDS offset is a constant and the odd PX's eighth slot is never computed; scheduling
    // barriers keep the halves and the kernel rows apart (hoisted together their LDS reads spill).
    const float2* td2 = reinterpret_cast<const float2*>(td);
    const float2 A0 = *reinterpret_cast<const float2*>(cst + 5 * CB + 2 * c2), B0 = *reinterpret_cast<const float2*>(cst + 6 * CB + 2 * c2);
#ifndef MX_DW_NH
#define MX_DW_NH 1
#endif
    constexpr int NH = MX_DW_NH;                   // the PX pixels go in NH parts of QX
    constexpr int QX = (PX + NH - 1) / NH;
    constexpr bool UNROLL_H = true;
    constexpr int HU = NH;
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
      const int grp = pq + 16 * gi;
      const bool gok = grp < NGRP;                 // (NGRP is a multiple of 16 for the 8 x 16 shape; 56 of 64 for 14 x 28)
      const int pyl = min(grp, NGRP - 1) / GX, pxl = (min(grp, NGRP - 1) % GX) * PX;
      const int oy = oy0 + pyl;
      constexpr bool XAHEAD = (TH != 7);           // (the 3-workgroup 7-row variant has no registers for it, and a third wave to wait behind)
      float2 xn[XAHEAD ? PX : 1];                  // the next pass's centre pixels, in flight under this pass's arithmetic
      if (XAHEAD && gi + 1 < NG) {
        const int g2 = min(grp + 16, NGRP - 1);
        load_x(xn, n, oy0 + g2 / GX, ox0 + (g2 % GX) * PX);
      }
      if (!XAHEAD && gi > 0) load_x(xr, n, oy, ox0 + pxl);
      const bool rok = gok && cok2 && oy < a.H;
      const float2* tdg = td2 + (pyl * IWP + pxl) * C2B + c2;
      const long rowoff = (((long)n * a.H + min(oy, a.H - 1)) * a.W + ox0 + pxl) * a.C + cc2;
      float2 cur[QX];                              // the centre pixels of the current half
#pragma unroll
      for (int o = 0; o < QX; ++o) cur[o] = xr[o];
      // a wave none of whose lanes has a pixel group in this pass (14 x 28: 56 groups in passes of 16 - waves 2 and 3 of the fourth) leaves
      // its issue slots to the other workgroup's waves instead of running 500 masked instructions
      const bool wave_has_work = __builtin_amdgcn_ballot_w64(rok) != 0;
#pragma unroll HU
      for (int h = 0; h < (wave_has_work ? NH : 0); ++h) {
        __builtin_amdgcn_sched_barrier(0);
        const int pxh = h * QX;
        int wlh = c2;                              // opaque per half: the K*K weight pairs are loop-invariant and would otherwise be
        asm volatile("" : "+v"(wlh));              // hoisted out of the loops as 2 K*K live registers (an index: ds_read, not flat_load)
        const float2* tdh = tdg + pxh * C2B;
        float2 xa[QX], ah[QX], sg[QX];
        bool img[QX];
#pragma unroll
        for (int o = 0; o < QX; ++o) {
          if (UNROLL_H && pxh + o >= PX) continue;
          const float2 r = cur[o];
          img[o] = rok && pxh + o < PX && ox0 + pxl + pxh + o < a.W;
          float2 v = r, sv = make_float2(0.f, 0.f);
          if (has_bn0) {
            const float zx = A0.x * r.x + B0.x, zy = A0.y * r.y + B0.y;
            sv.x = sigmoidf_(zx); sv.y = sigmoidf_(zy);
            v.x = zx * sv.x; v.y = zy * sv.y;
          }
          xa[o] = img[o] ? v : make_float2(0.f, 0.f);
          sg[o] = sv;
          ah[o] = make_float2(0.f, 0.f);
        }
        // the tile row of kernel row ky + 1 is requested before the products of row ky: two rows of reads live, never more
        // (not in the 3-workgroups-per-CU variant: 168 registers)
#ifndef MX_DW_PIPE
#define MX_DW_PIPE 1
#endif
        constexpr bool PIPE = MX_DW_PIPE && TH != 7 && (K == 5 || TH != 8);
        float2 inb[PIPE ? 2 : 1][QX - 1 + K];
#pragma unroll
        for (int j = 0; j < QX - 1 + K; ++j) {
          if (!PIPE || (UNROLL_H && pxh + j - (K - 1) >= PX)) continue;  // (a slot past PX feeds the left-out pixel only)
          inb[0][j] = tdh[((K - 1) * IWP + j) * C2B];
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          __builtin_amdgcn_sched_barrier(0);       // one kernel row at a time (hoisted together the reads spill)
          if (PIPE ? ky + 1 < K : true) {
            const int kr = PIPE ? ky + 1 : ky;
#pragma unroll
            for (int j = 0; j < QX - 1 + K; ++j) {
              if (UNROLL_H && pxh + j - (K - 1) >= PX) continue;
              inb[PIPE ? (ky + 1) & 1 : 0][j] = tdh[((K - 1 - kr) * IWP + j) * C2B];
            }
          }
          float2 (&in)[QX - 1 + K] = inb[PIPE ? ky & 1 : 0];
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const float2 w = reinterpret_cast<const float2*>(wl)[wlh + (ky * K + kx) * C2B];
            float2& p = part[ky * K + kx];
#pragma unroll
            for (int o = 0; o < QX; ++o) {
              if (UNROLL_H && pxh + o >= PX) continue;
              const float2 v = in[o + K - 1 - kx];
              ah[o].x += w.x * v.x; ah[o].y += w.y * v.y;
              p.x += xa[o].x * v.x; p.y += xa[o].y * v.y;
            }
          }
          // (this row's weight-gradient sums are pinned too: left free, the optimiser moves all K*K*PX of them to the end of the
          // pass and keeps every tile value it has read alive until then)
#pragma unroll
          for (int kx = 0; kx < K; ++kx) pin2(part[ky * K + kx]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // this part's epilogue and stores: no accumulator survives a part.  (The data gradient is pinned here: its only use is
        // under `if (img)`, and the optimiser otherwise sinks its K*K products into that branch, after ALL the tile reads.)
#pragma unroll
        for (int o = 0; o < QX; ++o) {
          if (UNROLL_H && pxh + o >= PX) continue;
          pin2(ah[o]);
        }
#pragma unroll
        for (int o = 0; o < QX; ++o) {
          if (UNROLL_H && pxh + o >= PX) continue;
          if (img[o]) {
            const long off = rowoff + (long)(pxh + o) * a.C;
            float2 v = ah[o];
            if (has_bn0) {
              const float2 r = cur[o];
              const float zx = A0.x * r.x + B0.x, zy = A0.y * r.y + B0.y;
              const float2 sv = sg[o];
              v.x *= sv.x * (1.0f + zx * (1.0f - sv.x)); v.y *= sv.y * (1.0f + zy * (1.0f - sv.y));   // swish'(z) from the same sigma
              s0.x += v.x; s0.y += v.y;
              s1.x += v.x * r.x; s1.y += v.y * r.y;
            } else if (a.res) {
              const float2 r = *reinterpret_cast<const float2*>(a.res + off);
              v.x += r.x; v.y += r.y;
            }
            *reinterpret_cast<float2*>(a.gx + off) = v;
          }
        }
#pragma unroll
        for (int o = 0; o < QX; ++o) cur[o] = ((h + 1) * QX + o < PX) ? xr[(h + 1) * QX + o] : make_float2(0.f, 0.f);   // (the next part's pixels)
      }
      if (XAHEAD && gi + 1 < NG) {
#pragma unroll
        for (int o = 0; o < PX; ++o) xr[o] = xn[XAHEAD ? o : 0];
      }
    }
    DW_STAMP(3);
  }
  // leave: dW and the BN0 sums as this workgroup's partial rows (global atomics here would have every workgroup of a
  // channel chunk contend on the same K*K*32 addresses: measured ~20 us per workgroup).  Inside the workgroup each wave
  // folds its lanes and writes one row into the (now free) tile memory; the four rows are added in wave order.
  __syncthreads();
  float* slots = reinterpret_cast<float*>(td);             // [4][K*K*CB]
  const int wave = tid >> 6;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    float2 p = part[t];
    p.x += __shfl_xor(p.x, 16, 64); p.y += __shfl_xor(p.y, 16, 64);
    p.x += __shfl_xor(p.x, 32, 64); p.y += __shfl_xor(p.y, 32, 64);
    if ((tid & 63) < C2B) *reinterpret_cast<float2*>(slots + (wave * K * K + t) * CB + 2 * c2) = p;
  }
  __syncthreads();
  for (int i = tid; i < K * K * CB; i += 256) {
    int cc = i % CB, tap = i / CB;
    if (c0 + cc < a.C)
      a.dwpart[(long)grp_id * a.C * K * K + (long)(c0 + cc) * K * K + tap] =
          ((slots[i] + slots[K * K * CB + i]) + slots[2 * K * K * CB + i]) + slots[3 * K * K * CB + i];
  }
  if (a.part) {
    __syncthreads();
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      s0.x += __shfl_xor(s0.x, o, 64); s0.y += __shfl_xor(s0.y, o, 64);
      s1.x += __shfl_xor(s1.x, o, 64); s1.y += __shfl_xor(s1.y, o, 64);
    }
    if ((tid & 63) < C2B) {
      *reinterpret_cast<float2*>(slots + wave * 2 * CB + 2 * c2) = s0;
      *reinterpret_cast<float2*>(slots + wave * 2 * CB + CB + 2 * c2) = s1;
    }
    __syncthreads();
    if (tid < 2 * CB) {
      const int cc = tid % CB;
      if (c0 + cc < a.C) {
        const float v = ((slots[tid] + slots[2 * CB + tid]) + slots[4 * CB + tid]) + slots[6 * CB + tid];
        float* dst = a.part + (long)grp_id * 2 * a.C + (tid / CB) * a.C + c0 + cc;
        if (a.fin_counters) mx_st_wt(dst, v); else *dst = v;
      }
    }
    if (a.fin_counters) {
      // every group of this channel chunk has left its row: the last one to arrive adds them in bn_reduce_finalize_kernel's order
      __shared__ unsigned fin_flag;
      if (mx_last_arriver(a.fin_counters + chunk_id, gridDim.x, &fin_flag))
        bn_bwd_reduce_finalize_32(a.part, (int)gridDim.x, a.C, c0, a.fin, reinterpret_cast<double*>(td));
    }
  }
}

// dW[i] += sum_g part[g][i]: one workgroup = 16 consecutive elements x 16 row lanes (lane l adds rows l, l+16, ... with four
// loads in flight), the 16 lane sums are added in lane order: one owner per element, no atomics, same bits every run.
__global__ __launch_bounds__(256) void dw_parts_reduce_kernel(const float* __restrict__ part, int P, int n, float* __restrict__ dW) {
  __shared__ float red[16][17];
  const int el = threadIdx.x & 15, gl = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n) {
    int g = gl;
    for (; g + 48 < P; g += 64) {
      s0 += part[(long)g * n + i]; s1 += part[(long)(g + 16) * n + i]; s2 += part[(long)(g + 32) * n + i]; s3 += part[(long)(g + 48) * n + i];
    }
    for (; g < P; g += 16) s0 += part[(long)g * n + i];
  }
  red[gl][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (gl == 0 && i < n) {
    float t = red[0][el];
#pragma unroll
    for (int k = 1; k < 16; ++k) t += red[k][el];
    dW[i] += t;
  }
}

static void launch_dw_parts_reduce(const float* part, int P, int n, float* dW, hipStream_t st) {
  hipLaunchKernelGGL(dw_parts_reduce_kernel, dim3(cdiv(n, 16)), dim3(256), 0, st, part, P, n, dW);
}

// tile shape of the fused backward: 1 = 14 x 28 (PX 7), 0 = 8 x 16 (PX 8), 3 = 8 x 28 (3x3 only), 2 = 7 x 28 (a knob: MX_DW_FUSED_TILE=2).  The large tile wherever it covers the image with
// fewer staged elements per valid output (its halo factor is lower, its quantisation coarser); MX_DW_FUSED_TILE forces one.
static int dw_fused_shape(int H, int Wd, int K) {
  static const int forced = getenv("MX_DW_FUSED_TILE") ? atoi(getenv("MX_DW_FUSED_TILE")) : -1;
  if (forced == 0 || forced == 1 || forced == 2 || (forced == 3 && K == 3)) return forced;
  // measured on MI355X (tools/microbench.py dwfused, profiles/r03_dwfused_tiles.txt): 5x5 gains 13-19 % on every B7 layer
  // (2.3-2.6 -> 2.8-2.9 TB/s); 3x3, which ran 3 workgroups per CU on the small tile, loses 9-15 % at 112 / 224 pixels and is
  // level at 28: the large tile is taken for 5x5 only
  // (round 4, after the rewrite: 3x3 takes 8 x 28 row-staged tiles - three workgroups per CU like the small tile, 1.71x instead of 1.41x
  // staged pixels per output but scalar row addressing and 7-pixel threads - where the image is a multiple of 28 wide: 3840 x 28 x 28
  // 368 -> 346 us, 960 x 28 x 28 108 -> 99, level at 112 / 224; profiles/r04_knob_sweep.txt)
  if (K != 5) return (Wd % 28 == 0) ? 3 : 0;
  const double small = (double)cdiv(H, 8) * cdiv(Wd, 16) * (8 + K - 1) * (16 + K - 1);
  const double large = (double)cdiv(H, 14) * cdiv(Wd, 28) * (14 + K - 1) * (28 + K - 1);
  return large < 0.9 * small ? 1 : 0;
}

// groups = partial rows written; gpp > 0: XCD-aware launch (a plane = one sample x one channel chunk is cut into gpp groups that do
// not straddle samples, all on one XCD); gpp = 0: plain 2-D grid, a group is tpb consecutive tiles of the (sample, tile) sequence
static void dw_fused_geom(int N, int H, int Wd, int C, int K, int* tiles_x, int* tiles_y, int* tpb, int* groups, int* gpp) {
  const int shape = dw_fused_shape(H, Wd, K);
  *tiles_x = cdiv(Wd, shape ? 28 : 16); *tiles_y = cdiv(H, shape == 2 ? 7 : shape == 3 ? 8 : shape ? 14 : 8);
  const int ntile = (*tiles_x) * (*tiles_y);
  long ntiles = (long)N * ntile;
  int chunks = cdiv(C, CB);
  // workgroups per launch ~ this target (tuning override MX_DW_GROUPS).  The 5x5 kernel (2 workgroups per CU, heavy
  // per-workgroup prologue / partial-row epilogue) wants few long-lived workgroups: 11.0 -> 9.8 ms per step at 1024
  // instead of 4096; the 3x3 kernel (3 per CU) wants the opposite: 6.8 ms at 4096, 8.0 ms at 1024.
  static const long override_target = getenv("MX_DW_GROUPS") ? atol(getenv("MX_DW_GROUPS")) : 0;
  // (round 4, rewritten kernel: the 14 x 28 / 5x5 form is level or 2-5 % faster at 512 - one workgroup per slot of its 2 per CU)
  const long group_target = override_target > 0 ? override_target : (shape == 3 ? 4096 : shape == 2 ? 768 : shape ? 512 : K == 5 ? 1024 : 4096);
  long g = group_target / chunks;
  if (g < 1) g = 1;
  if (g > ntiles) g = ntiles;
  *tpb = (int)((ntiles + g - 1) / g);
  *groups = cdiv(ntiles, *tpb);
  *gpp = 0;
  static const int xcd_on = getenv("MX_DW_XCD") ? atoi(getenv("MX_DW_XCD")) : 0;      // XCD-aware ids: built on PMC evidence of 1.4-1.6x fabric-side over-read, measured neutral (profiles/r03_dw_xcd.txt): the Infinity Cache serves the halo
  if (xcd_on && *tpb < ntile && (long)chunks * N >= 64) {
    *gpp = cdiv(ntile, *tpb);
    *groups = N * (*gpp);
  }
}

// ---------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------
static int dw_check(const DwArgs& a, int K, int S, const char* who) {
  MX_CHECK_ARG(K == 3 || K == 5, "%s: kernel %d unsupported (3 or 5)", who, K);
  MX_CHECK_ARG(S == 1 || S == 2, "%s: stride %d unsupported (1 or 2)", who, S);
  MX_CHECK_ARG(a.N > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.C % 4 == 0, "%s: bad extents N=%d H=%d W=%d C=%d", who, a.N, a.H, a.W, a.C);
  MX_CHECK_ARG(a.Ho > 0 && a.Wo > 0 && (a.Ho - 1) * S + K - a.pad <= a.H + K && a.pad >= 0 && a.pad < K,
               "%s: inconsistent output size Ho=%d Wo=%d pad=%d", who, a.Ho, a.Wo, a.pad);
  MX_CHECK_ARG((a.sc == nullptr) == (a.sh == nullptr), "%s: prologue scale/shift come together", who);
  return MX_OK;
}

#define DW_DISPATCH(KERNEL_S1, KERNEL_S2, K, S, grid, st, a)                        \
  do {                                                                              \
    if (K == 3 && S == 1) hipLaunchKernelGGL((KERNEL_S1(3)), grid, dim3(256), 0, st, a); \
    else if (K == 5 && S == 1) hipLaunchKernelGGL((KERNEL_S1(5)), grid, dim3(256), 0, st, a); \
    else if (K == 3 && S == 2) hipLaunchKernelGGL((KERNEL_S2(3)), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((KERNEL_S2(5)), grid, dim3(256), 0, st, a);             \
  } while (0)

#define FWD_S1(k) dw_fwd_kernel<k, 1, 8, 16, 4>
#define FWD_S1W(k) dw_fwd_kernel<k, 1, 8, 28, 7>
#define FWD_S2(k) dw_fwd_kernel<k, 2, 8, 8, 2>
#define BWW_S1(k) dw_bwd_weight_kernel<k, 1, 8, 16, 4>
#define BWW_S2(k) dw_bwd_weight_kernel<k, 2, 8, 8, 2>
#define BWD_S1(k) dw_bwd_data_kernel<k, 1, 8, 16>
#define BWD_S2(k) dw_bwd_data_kernel<k, 2, 8, 16>

extern "C" {

// tile groups of the forward: ~MX_DWF_GROUPS workgroups per launch (default 4096), each walking tpb consecutive tiles
// 1: the stride-1 forward takes its 8 x 28 tiles (images a multiple of 28 pixels wide: 224 / 112 / 56 / 28 of B7 at 448; MX_DWF_WIDE=0: never)
static int dw_fwd_wide(int Wo, int S) {
  static const int on = getenv("MX_DWF_WIDE") ? atoi(getenv("MX_DWF_WIDE")) : 1;
  return (on && S == 1 && Wo % 28 == 0) ? 1 : 0;
}

static void dw_fwd_geom(int N, int Ho, int Wo, int C, int S, bool pooled, int* tiles_x, int* tiles_y, int* tpb, int* groups, int* gpp) {
  *tiles_x = cdiv(Wo, dw_fwd_wide(Wo, S) ? 28 : S == 1 ? 16 : 8); *tiles_y = cdiv(Ho, 8);
  const int ntile = (*tiles_x) * (*tiles_y);
  const long ntiles = (long)N * ntile;
  const int chunks = cdiv(C, CB);
  static const long target = getenv("MX_DWF_GROUPS") ? atol(getenv("MX_DWF_GROUPS")) : 4096;
  long g = target / chunks;
  if (g < 1) g = 1;
  if (g > ntiles) g = ntiles;
  if (pooled) {
    // groups do not straddle samples: gpp per sample, at least one
    long per = g / N;
    if (per < 1) per = 1;
    if (per > ntile) per = ntile;
    if ((long)N * chunks > MX_WS_COUNTERS) per = 1;      // more (sample, chunk) pairs than arrival counters: one group per sample, no hand-off
    *tpb = cdiv(ntile, per);
    *gpp = cdiv(ntile, *tpb);
    *groups = N * (*gpp);
  } else {
    *tpb = (int)((ntiles + g - 1) / g);
    *groups = cdiv(ntiles, *tpb);
    *gpp = 0;
  }
}

// Y = dwconv(act(X)); act = swish(scale*x+shift) if scale != null.
// number of partial-statistics rows mx_dwconv_fwd writes
int mx_dwconv_fwd_parts(int N, int Ho, int Wo, int C, int S) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (S != 1 && S != 2)) return MX_EARG;
  int tx, ty, tpb, groups, gpp;
  dw_fwd_geom(N, Ho, Wo, C, S, false, &tx, &ty, &tpb, &groups, &gpp);
  return groups;
}

// bytes of scratch mx_dwconv_fwd needs when it also produces the SE squeeze sums (`pooled`); 0 = none
long mx_dwconv_fwd_ws(int N, int Ho, int Wo, int C, int S) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (S != 1 && S != 2)) return MX_EARG;
  int tx, ty, tpb, groups, gpp;
  dw_fwd_geom(N, Ho, Wo, C, S, true, &tx, &ty, &tpb, &groups, &gpp);
  if (gpp <= 1) return 0;
  return MX_WS_COUNTER_BYTES + (long)gpp * N * C * 4;
}

int mx_dwconv_fwd(const float* X, const float* scale, const float* shift, const float* W, float* Y, float* stats,
                  const float* pool_scale, const float* pool_shift, float* pooled, void* ws, long ws_bytes, int N,
                  int H, int Wd, int C, int K, int S, int pad_lo, int Ho, int Wo, void* stream) {
  DwArgs a{};
  a.x = X; a.sc = scale; a.sh = shift; a.w = W; a.y = Y; a.stats = stats;
  a.ps = pool_scale; a.pb = pool_shift; a.pooled = pooled;
  MX_CHECK_ARG(!pooled || (pool_scale && pool_shift && !stats), "dwconv_fwd: pooled needs pool_scale/pool_shift and excludes stats");
  a.N = N; a.H = H; a.W = Wd; a.Ho = Ho; a.Wo = Wo; a.C = C; a.pad = pad_lo;
  MX_CHECK_ARG(X && W && Y, "dwconv_fwd: null pointer");
  if (int e = dw_check(a, K, S, "dwconv_fwd")) return e;
  int groups;
  dw_fwd_geom(N, Ho, Wo, C, S, pooled != nullptr, &a.tiles_x, &a.tiles_y, &a.tiles_per_block, &groups, &a.gpp);
  dim3 grid(groups, cdiv(C, CB), 1);
  if (pooled && a.gpp > 1) {
    const long need = mx_dwconv_fwd_ws(N, Ho, Wo, C, S);
    MX_CHECK_ARG(ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0, "dwconv_fwd: pooled needs %ld bytes of scratch (mx_dwconv_fwd_ws)", need);
    MX_CHECK_ARG((long)N * grid.y <= MX_WS_COUNTERS, "dwconv_fwd: N=%d x %d channel chunks exceed the %d arrival counters", N, (int)grid.y, MX_WS_COUNTERS);
    a.counters = reinterpret_cast<unsigned*>(ws);
    a.poolpart = reinterpret_cast<float*>((char*)ws + MX_WS_COUNTER_BYTES);
  }
  if (dw_fwd_wide(Wo, S)) DW_DISPATCH(FWD_S1W, FWD_S2, K, S, grid, (hipStream_t)stream, a);
  else DW_DISPATCH(FWD_S1, FWD_S2, K, S, grid, (hipStream_t)stream, a);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// dX = dwconv^T(dY) (+ residual).  dX is the gradient w.r.t. the *activated* input.
int mx_dwconv_bwd_data(const float* dY, const float* W, const float* residual, float* dX, int N, int H, int Wd, int C,
                       int K, int S, int pad_lo, int Ho, int Wo, void* stream) {
  DwArgs a{};
  a.x = dY; a.w = W; a.res = residual; a.y = dX;
  a.N = N; a.H = H; a.W = Wd; a.Ho = Ho; a.Wo = Wo; a.C = C; a.pad = pad_lo;
  MX_CHECK_ARG(dY && W && dX, "dwconv_bwd_data: null pointer");
  if (int e = dw_check(a, K, S, "dwconv_bwd_data")) return e;
  a.tiles_x = cdiv(Wd, 16); a.tiles_y = cdiv(H, 8);
  dim3 grid(a.tiles_x * a.tiles_y, cdiv(C, CB), N);
  DW_DISPATCH(BWD_S1, BWD_S2, K, S, grid, (hipStream_t)stream, a);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// dW[C,1,K,K] += sum dY * act(X)
static void dw_bww_geom(int N, int Ho, int Wo, int C, int S, int* tiles_x, int* tiles_y, int* tpb, int* rows) {
  const int TH = 8, TW = (S == 1) ? 16 : 8;
  *tiles_x = cdiv(Wo, TW); *tiles_y = cdiv(Ho, TH);
  long ntiles = (long)N * (*tiles_x) * (*tiles_y);
  int chunks = cdiv(C, CB);
  long groups = 2048 / chunks;
  if (groups < 1) groups = 1;
  if (groups > ntiles) groups = ntiles;
  *tpb = (int)((ntiles + groups - 1) / groups);
  *rows = cdiv(ntiles, *tpb);
}

// number of partial rows [C*K*K] mx_dwconv_bwd_weight writes into dw_scratch
int mx_dwconv_bwd_weight_parts(int N, int Ho, int Wo, int C, int S) {
  if (N <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (S != 1 && S != 2)) return MX_EARG;
  int tx, ty, tpb, rows;
  dw_bww_geom(N, Ho, Wo, C, S, &tx, &ty, &tpb, &rows);
  return rows;
}

// dw_scratch: [mx_dwconv_bwd_weight_parts][C*K*K] floats (per-workgroup partial rows, added in a fixed order by a second kernel)
int mx_dwconv_bwd_weight(const float* X, const float* scale, const float* shift, const float* dY, float* dW, float* dw_scratch,
                         int N, int H, int Wd, int C, int K, int S, int pad_lo, int Ho, int Wo, void* stream) {
  DwArgs a{};
  a.x = X; a.sc = scale; a.sh = shift; a.dy = dY; a.y = dW; a.stats = dw_scratch;
  a.N = N; a.H = H; a.W = Wd; a.Ho = Ho; a.Wo = Wo; a.C = C; a.pad = pad_lo;
  MX_CHECK_ARG(X && dY && dW && dw_scratch, "dwconv_bwd_weight: null pointer");
  if (int e = dw_check(a, K, S, "dwconv_bwd_weight")) return e;
  int rows;
  dw_bww_geom(N, Ho, Wo, C, S, &a.tiles_x, &a.tiles_y, &a.tiles_per_block, &rows);
  dim3 grid(rows, cdiv(C, CB), 1);
  DW_DISPATCH(BWW_S1, BWW_S2, K, S, grid, (hipStream_t)stream, a);
  MX_LAUNCH_CHECK();
  launch_dw_parts_reduce(dw_scratch, rows, C * K * K, dW, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// number of partial rows mx_dwconv_bwd_fused writes (BN0 sums [rows][2][C] and dW scratch [rows][C*K*K])
int mx_dwconv_bwd_fused_parts(int N, int H, int Wd, int C, int K) {
  if (N <= 0 || H <= 0 || Wd <= 0 || C <= 0 || (K != 3 && K != 5)) return MX_EARG;
  int tx, ty, tpb, groups, gpp;
  dw_fused_geom(N, H, Wd, C, K, &tx, &ty, &tpb, &groups, &gpp);
  return groups;
}

// Stride-1 backward of [BN0+SiLU] -> dwconv -> BN1 -> SiLU -> SE gate fused (see dw_bwd_fused_kernel):
//   gX = dwconv^T(dd) [* swish'(a0*X+b0)] [+ residual];  dW += sum dd*act(X);  part = BN0 backward partial sums (a0 != NULL)
static int dw_bwd_fused_impl(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                            const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                            const float* W, const float* residual, float* gX, float* dW, float* dw_scratch, float* part, int N, int H,
                            int Wd, int C, int K, int pad_lo, void* stream, unsigned* fin_counters, const BnBwdFin* fin);

int mx_dwconv_bwd_fused(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                        const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                        const float* W, const float* residual, float* gX, float* dW, float* dw_scratch, float* part, int N, int H,
                        int Wd, int C, int K, int pad_lo, void* stream) {
  return dw_bwd_fused_impl(dA, D, gate, add, a1, b1, c1, c2, c3, X, a0, b0, W, residual, gX, dW, dw_scratch, part, N, H, Wd, C, K, pad_lo,
                           stream, nullptr, nullptr);
}

// The same with the BatchNorm-0 backward statistics FINISHED in the kernel (a0 / b0 / part required): the last workgroup of every
// 32-channel chunk adds the partial rows in mx_bn_bwd_finalize's order and writes dgamma / dbeta (+=) and the coefficients o1 o2 o3 of
// dX = o1*g + o2*X + o3 - what a separate mx_bn_bwd_finalize(part, ...) launch would have left, bit for bit.  ws: zeroed scratch whose
// first 64 KB are arrival counters (left zero), as for mx_pool_sum.  Returns MX_EARG for geometries it does not take (the XCD-grouped
// grid): call mx_dwconv_bwd_fused + mx_bn_bwd_finalize then.
int mx_dwconv_bwd_fused_bn0(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                            const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                            const float* W, float* gX, float* dW, float* dw_scratch, float* part, int N, int H, int Wd, int C, int K,
                            int pad_lo, void* ws, long ws_bytes, double count, const float* gamma, const float* mean, const float* rstd,
                            int training, float* dgamma, float* dbeta, float* o1, float* o2, float* o3, void* stream) {
  MX_CHECK_ARG(a0 && b0 && part, "dwconv_bwd_fused_bn0: the BatchNorm-0 form only (a0, b0, part)");
  MX_CHECK_ARG(ws && ws_bytes >= MX_WS_COUNTER_BYTES && ((uintptr_t)ws & 15) == 0, "dwconv_bwd_fused_bn0: scratch with %d bytes of counters required", MX_WS_COUNTER_BYTES);
  MX_CHECK_ARG(gamma && mean && rstd && dgamma && dbeta && o1 && o2 && o3 && count > 0, "dwconv_bwd_fused_bn0: null pointer");
  MX_CHECK_ARG(cdiv(C, CB) <= MX_WS_COUNTERS, "dwconv_bwd_fused_bn0: too many channel chunks");
  BnBwdFin fin{count, gamma, mean, rstd, training, dgamma, dbeta, o1, o2, o3};
  return dw_bwd_fused_impl(dA, D, gate, add, a1, b1, c1, c2, c3, X, a0, b0, W, nullptr, gX, dW, dw_scratch, part, N, H, Wd, C, K, pad_lo,
                           stream, reinterpret_cast<unsigned*>(ws), &fin);
}

static int dw_bwd_fused_impl(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                            const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                            const float* W, const float* residual, float* gX, float* dW, float* dw_scratch, float* part, int N, int H,
                            int Wd, int C, int K, int pad_lo, void* stream, unsigned* fin_counters, const BnBwdFin* fin) {
  MX_CHECK_ARG(dA && D && gate && add && a1 && b1 && c1 && c2 && c3 && X && W && gX && dw_scratch, "dwconv_bwd_fused: null pointer");
  MX_CHECK_ARG((a0 == nullptr) == (b0 == nullptr), "dwconv_bwd_fused: a0/b0 come together");
  MX_CHECK_ARG(!a0 || part, "dwconv_bwd_fused: BN0 present -> part required");
  MX_CHECK_ARG((K == 3 || K == 5) && pad_lo == (K - 1) / 2, "dwconv_bwd_fused: stride 1 with symmetric pad only (K=%d pad=%d)", K, pad_lo);
  MX_CHECK_ARG(N > 0 && H > 0 && Wd > 0 && C > 0 && C % 4 == 0, "dwconv_bwd_fused: bad extents");
  DwFusedArgs a{};
  a.dA = dA; a.d = D; a.gate = gate; a.add = add; a.a1 = a1; a.b1 = b1; a.c1 = c1; a.c2 = c2; a.c3 = c3;
  a.x = X; a.a0 = a0; a.b0 = b0; a.w = W; a.res = residual; a.gx = gX; a.dwpart = dw_scratch; a.part = a0 ? part : nullptr;
  a.N = N; a.H = H; a.W = Wd; a.C = C; a.pad = pad_lo;
  int groups;
  dw_fused_geom(N, H, Wd, C, K, &a.tiles_x, &a.tiles_y, &a.tiles_per_block, &groups, &a.gpp);
  dim3 grid(groups, cdiv(C, CB), 1);
  a.chunks = grid.y;
  if (a.gpp > 0) {
    a.xcd = 1;
    grid = dim3((unsigned)(cdiv((long)a.chunks * N, 8) * 8 * a.gpp), 1, 1);
  }
  if (fin_counters) {
    MX_CHECK_ARG(!a.xcd, "dwconv_bwd_fused_bn0: not with the XCD-grouped grid (MX_DW_XCD)");
    a.fin_counters = fin_counters; a.fin = *fin;
  }
  const int shape = dw_fused_shape(H, Wd, K);
  if (shape == 3) {
    hipLaunchKernelGGL((dw_bwd_fused_kernel<3, 8, 28, 7>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (shape == 2) {
    if (K == 3) hipLaunchKernelGGL((dw_bwd_fused_kernel<3, 7, 28, 7>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dw_bwd_fused_kernel<5, 7, 28, 7>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else if (shape) {
    static bool big_lds = false;                         // 14 x 28 tiles: 72 KB (5x5) / 60 KB (3x3) of static LDS
    (void)big_lds;
    if (K == 3) hipLaunchKernelGGL((dw_bwd_fused_kernel<3, 14, 28, 7>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dw_bwd_fused_kernel<5, 14, 28, 7>), grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    if (K == 3) hipLaunchKernelGGL((dw_bwd_fused_kernel<3, 8, 16, 8>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((dw_bwd_fused_kernel<5, 8, 16, 8>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  MX_LAUNCH_CHECK();
  if (dW) {                        // dW == null: the caller adds the partial rows later (mx_dw_parts_reduce), e.g. off the critical path
    launch_dw_parts_reduce(dw_scratch, groups, C * K * K, dW, (hipStream_t)stream);
    MX_LAUNCH_CHECK();
  }
  return MX_OK;
}

// dW[n] += sum over the P partial rows part[P][n] in row order: the second half of mx_dwconv_bwd_fused when it was called with dW == null
int mx_dw_parts_reduce(const float* part, int P, int n, float* dW, void* stream) {
  MX_CHECK_ARG(part && dW && P > 0 && n > 0, "dw_parts_reduce: bad arguments");
  launch_dw_parts_reduce(part, P, n, dW, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
