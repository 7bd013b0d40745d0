// BatchNorm statistics / affine / backward, SiLU and SE-gate elementwise work on NHWC
// activation matrices [rows, C] (rows = N*H*W), HBM-bound streaming kernels.
//
// Reference semantics: nn.BatchNorm2d(momentum=0.01, eps=1e-3) in train and eval mode
// (src/efficientnet_pytorch/model.py:45,53,64,132), MemoryEfficientSwish (utils.py:36-52),
// the SE gate (model.py:81-84) and drop_connect + skip (model.py:90-93, utils.py:82-91).
//
// All per-channel reductions use one skeleton: a block owns a contiguous slab of rows, threads
// are laid out (row-in-pass, float4-column) so every wave reads whole contiguous lines, partial
// sums live in registers, are combined through LDS and leave the block as one fp64 (or fp32)
// atomic per column.
#include "common.h"

// ---------------------------------------------------------------------------
// column-reduction skeleton
// ---------------------------------------------------------------------------
struct ColGeom {
  int rows, C, c4;        // matrix extents, C/4
  int tcols;              // float4 columns handled by one block (<= 256)
  int rpp;                // rows per pass = 256 / tcols (>=1)
  int rows_per_block;
  int rps;                // rows per sample
};

static ColGeom col_geom(long rows, int C, int rps, long rows_limit /*rows that one block may span at most*/,
                        int groups = 1 /*independent row ranges (samples) sharing the grid*/, int planes = 1) {
  ColGeom g;
  g.rows = (int)rows; g.C = C; g.c4 = C / 4; g.rps = rps;
  // float4 columns per workgroup: the width that wastes the fewest of the 256 threads (both on the padded last column
  // chunk and on 256 % tcols).  With a fixed 256, C = 1344 (336 float4) ran its second chunk at 31 % and C = 2304 its
  // third at 25 % - the two most common widths of the network.
  {
    const int cand[] = {g.c4 <= 256 ? g.c4 : 256, 256, 192, 128, 96, 64, 48, 32};
    double best = -1.0;
    g.tcols = cand[0];
    for (int tc : cand) {
      if (tc > g.c4) continue;
      const double util = (double)g.c4 / (double)(cdiv(g.c4, tc) * tc) * (double)(256 / tc * tc) / 256.0;
      // per-sample reductions (groups > 1) take the NARROWEST of equally good widths: more column chunks mean fewer row
      // blocks per sample, and the last arriver adds a sample's row blocks one after the other (C = 960 as one 240-wide
      // chunk cut every sample into 32 row blocks: a 160-load chain at the end of se_bn1_pool, 84 us against 40)
      if (groups > 1 ? util > best - 1e-9 : util > best + 1e-9) { best = util; g.tcols = tc; }
    }
  }
  g.rpp = 256 / g.tcols;
  int colchunks = cdiv(g.c4, g.tcols);
  // workgroups per reduction launch (tuning override MX_COLREDUCE_BLOCKS), swept twice on one box, ms per step:
  // 256: 153.3  512: 150.1/150.5  1024: 147.7/147.9  2048: 149.4/150.0  4096: 150.3/151.7
  static const long block_target = getenv("MX_COLREDUCE_BLOCKS") ? atol(getenv("MX_COLREDUCE_BLOCKS")) : 1024;
  // (the five-plane SE / BN1 pass carries 10 loads per row and thread and a five-fold final sum: half as many workgroups suit it
  //  better - tools/microbench.py pool, MX_COLREDUCE_BLOCKS 1024 / 512 / 256)
  long target_blocks = (planes == 5 ? block_target / 2 : block_target) / ((long)colchunks * groups);
  if (target_blocks < 1) target_blocks = 1;
  long rpb = (rows_limit + target_blocks - 1) / target_blocks;
  if (rpb < 4L * g.rpp) rpb = 4L * g.rpp;
  rpb = (rpb + g.rpp - 1) / g.rpp * g.rpp;
  if (rpb > rows_limit) rpb = rows_limit;
  g.rows_per_block = (int)rpb;
  return g;
}

// F::eval(r, c, v0, v1): contributes two float4 partials for element block (row r, cols c..c+3)
// sum of P partial float4s, `stride` floats apart, in index order (four loads in flight, one fixed association)
__device__ __forceinline__ float4 ordered_sum4(const float* p, int P, long stride) {
  float4 t = ld4(p);
  int i = 1;
  for (; i + 3 < P; i += 4) {
    const float4 a = ld4(p + i * stride), b = ld4(p + (i + 1) * stride), c = ld4(p + (i + 2) * stride), d = ld4(p + (i + 3) * stride);
    t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w;
    t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
    t.x += c.x; t.y += c.y; t.z += c.z; t.w += c.w;
    t.x += d.x; t.y += d.y; t.z += d.z; t.w += d.w;
  }
  for (; i < P; ++i) { const float4 a = ld4(p + i * stride); t.x += a.x; t.y += a.y; t.z += a.z; t.w += a.w; }
  return t;
}

// PER_SAMPLE: out0[sample][C] = the per-sample column sums.  A sample's rows are cut into gridDim.x row blocks; each
// block leaves its partial in `scratch` [gridDim.x][samples][C] and the last one to arrive adds them in block order
// (mx_last_arriver; counters [samples][gridDim.y]): same bits every run, no zero-filled output, no atomics.
template <class F, int NOUT, bool PER_SAMPLE, typename OutT>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, ColGeom g, OutT* out0, OutT* out1, float* scratch, unsigned* counters) {
  __shared__ float4 sm[2][256];
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x;
  const int used = g.tcols * g.rpp;
  const int tc = tid % g.tcols, tr = tid / g.tcols;
  const int c4 = blockIdx.y * g.tcols + tc;
  long r0, r1;
  int sample = 0;
  if (PER_SAMPLE) {
    sample = blockIdx.z;
    r0 = (long)sample * g.rps + (long)blockIdx.x * g.rows_per_block;
    r1 = min((long)(sample + 1) * g.rps, r0 + g.rows_per_block);
  } else {
    r0 = (long)blockIdx.x * g.rows_per_block;
    r1 = min((long)g.rows, r0 + g.rows_per_block);
  }
  float4 a0 = make_float4(0, 0, 0, 0), a1 = make_float4(0, 0, 0, 0);
  if (tid < used && c4 < g.c4) {
    long r = r0 + tr;
    const long st = g.rpp;
    float4 b0 = make_float4(0, 0, 0, 0), b1 = b0, c0 = b0, c1 = b0, d0 = b0, d1 = b0;
    if constexpr (F::kPipelined) {
      // The SE squeeze (one tensor, two transcendentals per element): 2.25 workgroups per CU at C = 2304 cannot hide a round trip behind
      // the evaluation of the previous eight rows, so the requests for rows r + 8 st .. r + 15 st leave BEFORE rows r .. r + 7 st are
      // evaluated (78 -> ~50 us on the 231 MB tensors of stage 6; round 5).  Same accumulators in the same order as the plain loop below.
      if (f.G == nullptr && r + 7 * st < r1) {
        const int c = 4 * c4;
        float4 cur[8], nxt[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = ld4(f.X + (r + i * st) * f.C + c);
        for (;;) {
          const bool more = r + 15 * st < r1;
          if (more) {
#pragma unroll
            for (int i = 0; i < 8; ++i) nxt[i] = ld4(f.X + (r + (8 + i) * st) * f.C + c);
          }
          __builtin_amdgcn_sched_barrier(0);
#define MX_POOL_ACC(s, v) { const float4 y = f.value(v, c); s.x += y.x; s.y += y.y; s.z += y.z; s.w += y.w; }
          MX_POOL_ACC(a0, cur[0]) MX_POOL_ACC(b0, cur[1]) MX_POOL_ACC(c0, cur[2]) MX_POOL_ACC(d0, cur[3])
          MX_POOL_ACC(a0, cur[4]) MX_POOL_ACC(b0, cur[5]) MX_POOL_ACC(c0, cur[6]) MX_POOL_ACC(d0, cur[7])
#undef MX_POOL_ACC
          r += 8 * st;
          if (!more) break;
#pragma unroll
          for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
        }
      }
    }
    if (NOUT == 1) {        // (eight rows for the two-tensor reductions as well: measured 0.1-0.2 ms per step slower, profiles/r04_knob_sweep.txt)
      for (; r + 7 * st < r1; r += 8 * st) {     // one-tensor reductions (pooling): eight independent rows in flight
        f.eval(r, 4 * c4, a0, a1);
        f.eval(r + st, 4 * c4, b0, b1);
        f.eval(r + 2 * st, 4 * c4, c0, c1);
        f.eval(r + 3 * st, 4 * c4, d0, d1);
        f.eval(r + 4 * st, 4 * c4, a0, a1);
        f.eval(r + 5 * st, 4 * c4, b0, b1);
        f.eval(r + 6 * st, 4 * c4, c0, c1);
        f.eval(r + 7 * st, 4 * c4, d0, d1);
      }
    }
    for (; r + 3 * st < r1; r += 4 * st) {       // four independent rows in flight
      f.eval(r, 4 * c4, a0, a1);
      f.eval(r + st, 4 * c4, b0, b1);
      f.eval(r + 2 * st, 4 * c4, c0, c1);
      f.eval(r + 3 * st, 4 * c4, d0, d1);
    }
    for (; r < r1; r += st) f.eval(r, 4 * c4, a0, a1);
    a0.x += b0.x + c0.x + d0.x; a0.y += b0.y + c0.y + d0.y; a0.z += b0.z + c0.z + d0.z; a0.w += b0.w + c0.w + d0.w;
    a1.x += b1.x + c1.x + d1.x; a1.y += b1.y + c1.y + d1.y; a1.z += b1.z + c1.z + d1.z; a1.w += b1.w + c1.w + d1.w;
  }
  sm[0][tid] = a0;
  sm[1][tid] = a1;
  __syncthreads();
  const bool owner = tid < g.tcols && c4 < g.c4;
  float4 s0 = make_float4(0, 0, 0, 0), s1 = s0;
  if (owner) {
    s0 = sm[0][tid]; s1 = sm[1][tid];
    for (int i = 1; i < g.rpp; ++i) {
      float4 t0 = sm[0][tid + i * g.tcols], t1 = sm[1][tid + i * g.tcols];
      s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w;
      s1.x += t1.x; s1.y += t1.y; s1.z += t1.z; s1.w += t1.w;
    }
    if (!PER_SAMPLE) {
      // one partial row per row-block: part[blockIdx.x][2][C]; summed in fp64 by the finalise kernel.
      // (contended fp64 atomics on 2C addresses ran at ~1/14 of the atomic rate and were not reproducible)
      OutT* p0 = out0 + (long)blockIdx.x * 2 * g.C + 4 * c4;
      p0[0] = s0.x; p0[1] = s0.y; p0[2] = s0.z; p0[3] = s0.w;
      if (NOUT > 1) { OutT* p1 = p0 + g.C; p1[0] = s1.x; p1[1] = s1.y; p1[2] = s1.z; p1[3] = s1.w; }
    }
  }
  if (PER_SAMPLE) {
    float* o = reinterpret_cast<float*>(out0) + (long)sample * g.C + 4 * c4;
    if (gridDim.x == 1) {
      if (owner) st4(o, s0);
      return;
    }
    const long plane = (long)gridDim.z * g.C;
    if (owner) mx_st4_wt(scratch + blockIdx.x * plane + (long)sample * g.C + 4 * c4, s0);
    if (!mx_last_arriver(counters + sample * gridDim.y + blockIdx.y, gridDim.x, &last_flag)) return;
    if (owner) st4(o, ordered_sum4(scratch + (long)sample * g.C + 4 * c4, gridDim.x, plane));
  }
}

static int colreduce_parts(long rows, int C) {
  ColGeom g = col_geom(rows, C, 1, rows, 1);
  return cdiv(rows, g.rows_per_block);
}

// geometry of the per-sample reductions (mx_pool_sum, mx_se_bn1_pool): falls back to one row block per sample when the
// launch would need more arrival counters than the scratch header holds
static ColGeom pool_geom(long rows, int C, int rps, dim3* grid, int planes = 1) {
  const int N = (int)(rows / rps);
  ColGeom g = col_geom(rows, C, rps, rps, N, planes);
  const int colchunks = cdiv(g.c4, g.tcols);
  if ((long)N * colchunks > MX_WS_COUNTERS) g.rows_per_block = rps;
  *grid = dim3(cdiv(rps, g.rows_per_block), colchunks, N);
  return g;
}

static long pool_ws_bytes(long rows, int C, int rps, int planes) {
  dim3 grid;
  pool_geom(rows, C, rps, &grid, planes);
  if (grid.x == 1) return 0;
  return MX_WS_COUNTER_BYTES + (long)grid.x * planes * (rows / rps) * C * 4;
}

template <class F, int NOUT, bool PER_SAMPLE, typename OutT>
static void launch_colreduce(const F& f, long rows, int C, int rps, OutT* o0, OutT* o1, hipStream_t st, void* ws = nullptr) {
  if (PER_SAMPLE) {
    dim3 grid;
    ColGeom g = pool_geom(rows, C, rps, &grid);
    hipLaunchKernelGGL((colreduce_kernel<F, NOUT, PER_SAMPLE, OutT>), grid, dim3(256), 0, st, f, g, o0, o1,
                       ws ? reinterpret_cast<float*>((char*)ws + MX_WS_COUNTER_BYTES) : nullptr, reinterpret_cast<unsigned*>(ws));
    return;
  }
  ColGeom g = col_geom(rows, C, rps, rows, 1);
  dim3 grid(cdiv(rows, g.rows_per_block), cdiv(g.c4, g.tcols), 1);
  hipLaunchKernelGGL((colreduce_kernel<F, NOUT, PER_SAMPLE, OutT>), grid, dim3(256), 0, st, f, g, o0, o1, (float*)nullptr, (unsigned*)nullptr);
}

// ---------------------------------------------------------------------------
// effective upstream gradient of a BN output, shared by the reduce and the apply pass:
//   g = G[r,c]
//   if rs   : g *= rs[n]                       (drop_connect scale of the sample, utils.py:90)
//   if gate : g  = g * gate[n,c] + add[n,c]    (SE gate and the pooled-path gradient, model.py:82-84)
//   if act  : g *= swish'(a[c] * X[r,c] + b[c])  (SiLU backward recomputed from the saved pre-BN tensor)
// ---------------------------------------------------------------------------
struct GEff {
  const float* G; const float* X;
  const float* rs; const float* gate; const float* add;
  const float* a; const float* b;
  int C, rps;
  __device__ __forceinline__ void get(long r, int c, float4& g, float4& x) const {
    get_n(r, (int)((unsigned)r / (unsigned)rps), c, g, x);   // (32-bit: a 64-bit division is ~60 instructions per float4; rows < 2^31)
  }
  // the same with the row's sample index n = r / rps supplied (the streaming kernels carry it along instead of dividing)
  __device__ __forceinline__ void get_n(long r, int n, int c, float4& g, float4& x) const {
    g = ld4(G + r * C + c);
    x = ld4(X + r * C + c);
    if (rs) { float s = rs[n]; g.x *= s; g.y *= s; g.z *= s; g.w *= s; }
    if (gate) {
      float4 gt = ld4(gate + (long)n * C + c), ad = ld4(add + (long)n * C + c);
      g.x = g.x * gt.x + ad.x; g.y = g.y * gt.y + ad.y; g.z = g.z * gt.z + ad.z; g.w = g.w * gt.w + ad.w;
    }
    if (a) {
      float4 aa = ld4(a + c), bb = ld4(b + c);
      g.x *= swish_gradf_(aa.x * x.x + bb.x); g.y *= swish_gradf_(aa.y * x.y + bb.y);
      g.z *= swish_gradf_(aa.z * x.z + bb.z); g.w *= swish_gradf_(aa.w * x.w + bb.w);
    }
  }
};

struct FBwdReduce {      // sum g, sum g*x
  static constexpr bool kPipelined = false;
  GEff e;
  __device__ __forceinline__ void eval(long r, int c, float4& s0, float4& s1) const {
    float4 g, x;
    e.get(r, c, g, x);
    s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
    s1.x += g.x * x.x; s1.y += g.y * x.y; s1.z += g.z * x.z; s1.w += g.w * x.w;
  }
};

struct FStats {          // sum x, sum x^2
  static constexpr bool kPipelined = false;
  const float* X; int C;
  __device__ __forceinline__ void eval(long r, int c, float4& s0, float4& s1) const {
    float4 x = ld4(X + r * C + c);
    s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
    s1.x += x.x * x.x; s1.y += x.y * x.y; s1.z += x.z * x.z; s1.w += x.w * x.w;
  }
};

struct FPool {           // per-sample: sum act(x) ; optionally sum g*act(x)
  static constexpr bool kPipelined = true;      // colreduce_kernel requests the next eight rows before it evaluates the current eight (G == nullptr)
  const float* X; const float* G; const float* a; const float* b; int C; int act;
  __device__ __forceinline__ float4 value(float4 x, int c) const {
    if (a) {
      float4 aa = ld4(a + c), bb = ld4(b + c);
      x.x = aa.x * x.x + bb.x; x.y = aa.y * x.y + bb.y; x.z = aa.z * x.z + bb.z; x.w = aa.w * x.w + bb.w;
    }
    if (act) { x.x = swishf_(x.x); x.y = swishf_(x.y); x.z = swishf_(x.z); x.w = swishf_(x.w); }
    return x;
  }
  __device__ __forceinline__ void eval(long r, int c, float4& s0, float4& s1) const {
    float4 x = value(ld4(X + r * C + c), c);
    if (G) {
      float4 g = ld4(G + r * C + c);
      x.x *= g.x; x.y *= g.y; x.z *= g.z; x.w *= g.w;
    }
    s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
  }
};

// ---------------------------------------------------------------------------
// SE / BN1 backward in ONE pass over (dA, d_raw): per (sample, channel)
//   out[0] = sum dA * act            (gradient of the SE gate, model.py:84)
//   out[1] = sum dA * s'(z)          out[2] = sum s'(z)
//   out[3] = sum dA * s'(z) * x      out[4] = sum s'(z) * x          z = a*x + b, act = swish(z), x = d_raw
// The BatchNorm-1 backward sums for g = (dA*gate + add) * s'(z) are then sum_n gate*out[1] + add*out[2] (and with x),
// which needs `add` (the SE backward) but no second pass over the tensors.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void se_bn1_pool_kernel(const float* G, const float* X, const float* a, const float* b, ColGeom g,
                                                          float* out /*[5][N][C]*/, long plane, float* scratch /*[gridDim.x][5][N][C]*/,
                                                          unsigned* counters) {
  __shared__ float4 sm[5][256];
  __shared__ unsigned last_flag;
  const int tid = threadIdx.x;
  const int used = g.tcols * g.rpp;
  const int tc = tid % g.tcols, tr = tid / g.tcols;
  const int c4 = blockIdx.y * g.tcols + tc;
  const int sample = blockIdx.z;
  const long r0 = (long)sample * g.rps + (long)blockIdx.x * g.rows_per_block;
  const long r1 = min((long)(sample + 1) * g.rps, r0 + g.rows_per_block);
  float4 acc[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc[i] = make_float4(0, 0, 0, 0);
  if (tid < used && c4 < g.c4) {
    const int c = 4 * c4;
    const float4 aa = ld4(a + c), bb = ld4(b + c);
#define SE1(f, x, ga)                                          \
      {                                                           \
        float z = aa.f * x.f + bb.f, sg = sigmoidf_(z);           \
        float act = z * sg, sp = sg * (1.f + z * (1.f - sg));     \
        acc[0].f += ga.f * act; acc[1].f += ga.f * sp; acc[2].f += sp; \
        acc[3].f += ga.f * sp * x.f; acc[4].f += sp * x.f;        \
      }
    long r = r0 + tr;
    for (; r + 3 * g.rpp < r1; r += 4 * g.rpp) {   // four rows (eight loads) in flight
      const float4 x0 = ld4(X + r * g.C + c), g0 = ld4(G + r * g.C + c);
      const float4 x1 = ld4(X + (r + g.rpp) * g.C + c), g1 = ld4(G + (r + g.rpp) * g.C + c);
      const float4 x2 = ld4(X + (r + 2 * g.rpp) * g.C + c), g2 = ld4(G + (r + 2 * g.rpp) * g.C + c);
      const float4 x3 = ld4(X + (r + 3 * g.rpp) * g.C + c), g3 = ld4(G + (r + 3 * g.rpp) * g.C + c);
      SE1(x, x0, g0) SE1(y, x0, g0) SE1(z, x0, g0) SE1(w, x0, g0)
      SE1(x, x1, g1) SE1(y, x1, g1) SE1(z, x1, g1) SE1(w, x1, g1)
      SE1(x, x2, g2) SE1(y, x2, g2) SE1(z, x2, g2) SE1(w, x2, g2)
      SE1(x, x3, g3) SE1(y, x3, g3) SE1(z, x3, g3) SE1(w, x3, g3)
    }
    for (; r + g.rpp < r1; r += 2 * g.rpp) {       // two rows (four loads) in flight
      const float4 x0 = ld4(X + r * g.C + c), g0 = ld4(G + r * g.C + c);
      const float4 x1 = ld4(X + (r + g.rpp) * g.C + c), g1 = ld4(G + (r + g.rpp) * g.C + c);
      SE1(x, x0, g0) SE1(y, x0, g0) SE1(z, x0, g0) SE1(w, x0, g0)
      SE1(x, x1, g1) SE1(y, x1, g1) SE1(z, x1, g1) SE1(w, x1, g1)
    }
    for (; r < r1; r += g.rpp) {
      const float4 x0 = ld4(X + r * g.C + c), g0 = ld4(G + r * g.C + c);
      SE1(x, x0, g0) SE1(y, x0, g0) SE1(z, x0, g0) SE1(w, x0, g0)
    }
#undef SE1
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) sm[i][tid] = acc[i];
  __syncthreads();
  const bool owner = tid < g.tcols && c4 < g.c4;
  const long off = (long)sample * g.C + 4 * c4;
  // the row blocks of a sample are joined in block order by the last one to arrive (see colreduce_kernel): no atomics
  float* dst = gridDim.x == 1 ? out : scratch + (long)blockIdx.x * 5 * plane;
  if (owner) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float4 s0 = sm[i][tid];
      for (int k = 1; k < g.rpp; ++k) { float4 t0 = sm[i][tid + k * g.tcols]; s0.x += t0.x; s0.y += t0.y; s0.z += t0.z; s0.w += t0.w; }
      if (gridDim.x == 1) st4(dst + i * plane + off, s0);
      else mx_st4_wt(dst + i * plane + off, s0);
    }
  }
  if (gridDim.x == 1) return;
  if (!mx_last_arriver(counters + sample * gridDim.y + blockIdx.y, gridDim.x, &last_flag)) return;
  if (owner) {
#pragma unroll
    for (int i = 0; i < 5; ++i) st4(out + i * plane + off, ordered_sum4(scratch + i * plane + off, gridDim.x, 5 * plane));
  }
}

// ---------------------------------------------------------------------------
// per-channel finalisation kernels (tiny)
// ---------------------------------------------------------------------------
// Two-level reduction of the partial rows part[P][2][C]: level 1 (this kernel, grid = channel chunks x row slices)
// sums a slice of rows in fp64 into acc[slice][2C] (plain stores); level 2, the finalise kernel, adds the <= 64 slices in
// slice order.  One block = 32 channels x 8 row lanes.  (The first version joined the slices with fp64 atomics.)
constexpr int BN_MAX_SLICES = 64;
__global__ __launch_bounds__(256) void bn_parts_reduce_kernel(const float* part, int P, int C, int rows_per_slice, double* acc) {
  __shared__ double sh[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int p0 = blockIdx.y * rows_per_slice, p1 = min(P, p0 + rows_per_slice);
  double a = 0.0, b = 0.0;
  if (c < C) {
    int p = p0 + rl;
    for (; p + 24 < p1; p += 32) {     // four rows in flight per thread
      float a0 = part[(long)p * 2 * C + c], b0 = part[(long)p * 2 * C + C + c];
      float a1 = part[(long)(p + 8) * 2 * C + c], b1 = part[(long)(p + 8) * 2 * C + C + c];
      float a2 = part[(long)(p + 16) * 2 * C + c], b2 = part[(long)(p + 16) * 2 * C + C + c];
      float a3 = part[(long)(p + 24) * 2 * C + c], b3 = part[(long)(p + 24) * 2 * C + C + c];
      a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; p < p1; p += 8) { a += (double)part[(long)p * 2 * C + c]; b += (double)part[(long)p * 2 * C + C + c]; }
  }
  sh[0][rl][cl] = a; sh[1][rl][cl] = b;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  for (int i = 1; i < 8; ++i) { a += sh[0][i][cl]; b += sh[1][i][cl]; }
  acc[(long)blockIdx.y * 2 * C + c] = a;
  acc[(long)blockIdx.y * 2 * C + C + c] = b;
}

// returns the number of slices written to acc[slices][2C]
static int launch_parts_reduce(const float* part, int P, int C, double* acc, hipStream_t st) {
  int slices = P / 32;
  if (slices < 1) slices = 1;
  if (slices > BN_MAX_SLICES) slices = BN_MAX_SLICES;
  int rps = (P + slices - 1) / slices;
  slices = (P + rps - 1) / rps;
  hipLaunchKernelGGL(bn_parts_reduce_kernel, dim3(cdiv(C, 32), slices), dim3(256), 0, st, part, P, C, rps, acc);
  return slices;
}

struct BnFwdFin {
  double count; const float* gamma; const float* beta; float* rmean; float* rvar; float momentum, eps; int training;
  float* scale; float* shift; float* mean_out; float* rstd_out;
};

__device__ __forceinline__ void bn_fwd_finalize_one(int c, double st0, double st1, double count, const float* gamma,
                                                    const float* beta, float* rmean, float* rvar, float momentum, float eps,
                                                    int training, float* scale, float* shift, float* mean_out, float* rstd_out) {
  float mean, rstd;
  if (training) {
    double m = st0 / count;
    double var = st1 / count - m * m;
    if (var < 0) var = 0;
    mean = (float)m;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    // running stats: unbiased variance (torch.nn.functional.batch_norm)
    double unb = count > 1 ? var * count / (count - 1) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
  } else {
    mean = rmean[c];
    rstd = 1.0f / sqrtf(rvar[c] + eps);
  }
  float sc = gamma[c] * rstd;
  scale[c] = sc;
  shift[c] = beta[c] - mean * sc;
  mean_out[c] = mean;
  rstd_out[c] = rstd;
}

// second level of the two-level reduction: 32 channels x 8 slice lanes per workgroup (lane l adds slices l, l+8, ...), the 8
// lane sums are added in lane order
__device__ __forceinline__ bool slices_sum(const double* acc, int slices, int C, double& s0, double& s1) {
  __shared__ double sh[2][8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  s0 = s1 = 0.0;
  if (c < C)
    for (int i = sl; i < slices; i += 8) { s0 += acc[(long)i * 2 * C + c]; s1 += acc[(long)i * 2 * C + C + c]; }
  sh[0][sl][cl] = s0; sh[1][sl][cl] = s1;
  __syncthreads();
  if (sl != 0 || c >= C) return false;
  for (int i = 1; i < 8; ++i) { s0 += sh[0][i][cl]; s1 += sh[1][i][cl]; }
  return true;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* acc, int slices, int C, double count, const float* gamma,
                                   const float* beta, float* rmean, float* rvar, float momentum, float eps, int training,
                                   float* scale, float* shift, float* mean_out, float* rstd_out) {
  double st0, st1;
  if (!slices_sum(acc, training ? slices : 0, C, st0, st1)) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  bn_fwd_finalize_one(c, st0, st1, count, gamma, beta, rmean, rvar, momentum, eps, training, scale, shift, mean_out, rstd_out);
}


__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* acc, int slices, int C, double count, const float* gamma,
                                       const float* mean, const float* rstd, int training, float* dgamma, float* dbeta,
                                       float* c1, float* c2, float* c3) {
  double s0, s1;
  if (!slices_sum(acc, slices, C, s0, s1)) return;
  const int c = blockIdx.x * 32 + (threadIdx.x & 31);
  bn_bwd_finalize_one(c, s0, s1, count, gamma, mean, rstd, training, dgamma, dbeta, c1, c2, c3);
}

// Up to FUSED_FINALIZE_MAX_ROWS partial rows: reduction and finalisation in ONE launch (one workgroup = 32 channels x 8 row
// lanes walks all rows).  The memset + reduce + finalise triple costs three ~5 us launches per BatchNorm per direction:
// 324 BatchNorm finalisations per B7 step.
static const int FUSED_FINALIZE_MAX_ROWS = getenv("MX_BN_FUSED_ROWS") ? atoi(getenv("MX_BN_FUSED_ROWS")) : 1024;
template <bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* part, int P, int C, BnFwdFin f, BnBwdFin b) {
  __shared__ double sh[2][8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) bn_parts_lane_sum(part, P, C, c, rl, s0, s1);
  sh[0][rl][cl] = s0; sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  for (int i = 1; i < 8; ++i) { s0 += sh[0][i][cl]; s1 += sh[1][i][cl]; }
  if (BWD) bn_bwd_finalize_one(c, s0, s1, b.count, b.gamma, b.mean, b.rstd, b.training, b.dgamma, b.dbeta, b.c1, b.c2, b.c3);
  else bn_fwd_finalize_one(c, s0, s1, f.count, f.gamma, f.beta, f.rmean, f.rvar, f.momentum, f.eps, 1, f.scale, f.shift, f.mean_out, f.rstd_out);
}

// The pooled-path gradient of the SE block, the BN1 backward sums and their finalisation in one launch:
//   add[n,c]  = inv_hw * sum_j gh[n,j] W1[j,c]                (d loss / d squeeze input, model.py:82-83 backward; gh from
//               mx_se_bwd) - owned per (n, c), j in ascending order: no atomics across squeeze slices
//   sums      = sum_n gate*S1 + add*S2, sum_n gate*S3 + add*S4 (fp64, n ascending per lane, lanes in order)  -> bn_bwd_finalize_one
// One workgroup = 64 channels x NL sample lanes (NL = 16: 1024 threads; the kernel is latency-bound - a chain of SQ dependent
// W1 loads per sample group - so more lanes, not more work per lane: 23.5 -> 12 us at C = 2304); gh [N][SQ] is staged in LDS.
constexpr int B1_NL = 16;
// STAGED = false: gh is read from global memory (batches whose [N][SQ] table outgrows 64 KB of LDS: B7 above 76 samples per
// GPU); same sums in the same order, so the result does not depend on which variant ran.
template <bool STAGED>
__global__ __launch_bounds__(64 * B1_NL) void bn1_sums_finalize_kernel(const float* pooled /*[5][N][C]*/, const float* gate, const float* gh,
                                                                      const float* W1, float inv_hw, float* add, int N, int C, int SQ,
                                                                      BnBwdFin b) {
  extern __shared__ float smem_b1[];                  // STAGED: [N][SQ], then 2 x [NL][64] doubles; else the doubles only
  const int cl = threadIdx.x & 63, nl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const float* ghs = STAGED ? smem_b1 : gh;
  if (STAGED) {
    for (int i = threadIdx.x; i < N * SQ; i += 64 * B1_NL) smem_b1[i] = gh[i];
    __syncthreads();
  }
  const long plane = (long)N * C;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    for (int n0 = nl; n0 < N; n0 += 2 * B1_NL) {      // this lane's samples n0 and n0 + NL: two accumulators per W1 load
      const int n1 = n0 + B1_NL;
      float a0 = 0.f, a1 = 0.f;
      const float* g0 = ghs + n0 * SQ;
      const float* g1 = ghs + (n1 < N ? n1 : n0) * SQ;      // (a1 is dropped below when n1 is past the batch)
      int j = 0;
      for (; j + 32 <= SQ; j += 32) {                        // 32 rows of W1 requested before the first is used: SQ / 32 round trips
        float w[32];                                         // to L2 instead of SQ / 4 (the compiler's own unrolling): 11.7 -> ~7 us at SQ = 96
#pragma unroll
        for (int u = 0; u < 32; ++u) w[u] = W1[(long)(j + u) * C + c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 32; ++u) { a0 += g0[j + u] * w[u]; a1 += g1[j + u] * w[u]; }
      }
      for (; j + 8 <= SQ; j += 8) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) w[u] = W1[(long)(j + u) * C + c];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a0 += g0[j + u] * w[u]; a1 += g1[j + u] * w[u]; }
      }
      for (; j < SQ; ++j) {
        const float w = W1[(long)j * C + c];
        a0 += g0[j] * w;
        a1 += g1[j] * w;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int n = k ? n1 : n0;
        if (n < N) {
          const long i = (long)n * C + c;
          const float ad = (k ? a1 : a0) * inv_hw;
          add[i] = ad;
          s0 += (double)gate[i] * pooled[plane + i] + (double)ad * pooled[2 * plane + i];
          s1 += (double)gate[i] * pooled[3 * plane + i] + (double)ad * pooled[4 * plane + i];
        }
      }
    }
  }
  double* red = reinterpret_cast<double*>(smem_b1 + (STAGED ? ((N * SQ + 1) & ~1) : 0));
  __syncthreads();
  red[nl * 64 + cl] = s0; red[B1_NL * 64 + nl * 64 + cl] = s1;
  __syncthreads();
  if (nl != 0 || c >= C) return;
  s0 = red[cl]; s1 = red[B1_NL * 64 + cl];
  for (int l = 1; l < B1_NL; ++l) { s0 += red[l * 64 + cl]; s1 += red[B1_NL * 64 + l * 64 + cl]; }
  bn_bwd_finalize_one(c, (double)(float)s0, (double)(float)s1, b.count, b.gamma, b.mean, b.rstd, b.training, b.dgamma, b.dbeta,
                      b.c1, b.c2, b.c3);
}

// ---------------------------------------------------------------------------
// streaming elementwise kernels
// Index walk of the grid-stride streaming kernels over [rows][C / 4] float4s: (row, float4 column, sample) are carried from one
// element to the next by additions and one compare each - the 64-bit divisions i / c4n and r / rps cost ~100 instructions per float4,
// more than the rest of the loop body, and these kernels run beside the VALU-bound weight-gradient GEMMs of the side stream.
struct RowWalk {
  long r; int c4, n, rem;                 // row, float4 column, sample, row within the sample
  int dr, dc, dn, drem, c4n, rps;
  __device__ __forceinline__ RowWalk(long i0, long stride, int c4n_, int rps_) : c4n(c4n_), rps(rps_) {
    r = i0 / c4n; c4 = (int)(i0 - r * c4n);
    n = (int)(r / rps); rem = (int)(r - (long)n * rps);
    const long sr = stride / c4n;
    dr = (int)sr; dc = (int)(stride - sr * c4n);
    dn = dr / rps; drem = dr - dn * rps;
  }
  __device__ __forceinline__ void next() {
    c4 += dc; r += dr; n += dn; rem += drem;
    if (c4 >= c4n) { c4 -= c4n; ++r; ++rem; }
    if (rem >= rps) { rem -= rps; ++n; }
  }
};

// ---------------------------------------------------------------------------
// out = (sc[c]*P + sh[c]) [swish] [* gate[n,c]] [* rs[n]] [+ R]
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* P, const float* sc, const float* sh, const float* rs,
                                                       const float* R, const float* gate, float* out, long total4, int C,
                                                       int rps, int act) {
  RowWalk w(blockIdx.x * 256L + threadIdx.x, (long)gridDim.x * 256, C / 4, rps);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256, w.next()) {
    const int c = w.c4 * 4;
    float4 v = ld4(P + i * 4);
    float4 a = ld4(sc + c), b = ld4(sh + c);
    v.x = a.x * v.x + b.x; v.y = a.y * v.y + b.y; v.z = a.z * v.z + b.z; v.w = a.w * v.w + b.w;
    if (act) { v.x = swishf_(v.x); v.y = swishf_(v.y); v.z = swishf_(v.z); v.w = swishf_(v.w); }
    if (gate) { float4 g = ld4(gate + (long)w.n * C + c); v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w; }
    if (rs) { float s = rs[w.n]; v.x *= s; v.y *= s; v.z *= s; v.w *= s; }
    if (R) { float4 q = ld4(R + i * 4); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    st4(out + i * 4, v);
  }
}

// out = c1[c]*g_eff + c2[c]*X + c3[c]   (BatchNorm backward, data gradient)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(GEff e, const float* c1, const float* c2, const float* c3,
                                                           float* out, long total4) {
  RowWalk w(blockIdx.x * 256L + threadIdx.x, (long)gridDim.x * 256, e.C / 4, e.rps);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256, w.next()) {
    const int c = w.c4 * 4;
    float4 g, x;
    e.get_n(w.r, w.n, c, g, x);
    float4 k1 = ld4(c1 + c), k2 = ld4(c2 + c), k3 = ld4(c3 + c);
    float4 o;
    o.x = k1.x * g.x + k2.x * x.x + k3.x; o.y = k1.y * g.y + k2.y * x.y + k3.y;
    o.z = k1.z * g.z + k2.z * x.z + k3.z; o.w = k1.w * g.w + k2.w * x.w + k3.w;
    st4(out + i * 4, o);
  }
}

static int grid_for(long total4) {
  // grid cap of the grid-stride streaming kernels (tuning override MX_STREAM_BLOCKS); swept 2048 / 4096 / 8192 twice:
  // 148.6, 149.4 / 147.9, 148.1 / 148.9, 149.4 ms per step
  static const long cap = getenv("MX_STREAM_BLOCKS") ? atol(getenv("MX_STREAM_BLOCKS")) : 4096;
  long b = (total4 + 255) / 256;
  return (int)(b < cap ? (b < 1 ? 1 : b) : cap);
}

// Inference constants of one MBConv block in ONE launch (was ~17 elementwise launches per block, ~900 per network: the
// no-grad eval forward of train_mcl.py:205-206 refolds every step, because the optimizer has just moved the weights).
struct FoldArgs {
  const float *We, *g0, *b0, *m0, *v0, *g1, *b1, *m1, *v1, *Wp, *g2, *b2, *m2, *v2;
  float eps0, eps1, eps2;
  int Cin, Cexp, Cout;
  float *We_f, *be, *s1, *t1, *Wp_f, *bp;
};
__global__ __launch_bounds__(256) void fold_block_kernel(FoldArgs a) {
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
  if (a.We) {
    for (long i = tid; i < (long)a.Cexp * a.Cin; i += nth) {
      const int c = (int)(i / a.Cin);
      a.We_f[i] = a.We[i] * (a.g0[c] * rsqrtf(a.v0[c] + a.eps0));
    }
    for (long c = tid; c < a.Cexp; c += nth) {
      const float s = a.g0[c] * rsqrtf(a.v0[c] + a.eps0);
      a.be[c] = a.b0[c] - s * a.m0[c];
    }
  }
  for (long c = tid; c < a.Cexp; c += nth) {
    const float s = a.g1[c] * rsqrtf(a.v1[c] + a.eps1);
    a.s1[c] = s; a.t1[c] = a.b1[c] - s * a.m1[c];
  }
  for (long i = tid; i < (long)a.Cout * a.Cexp; i += nth) {
    const int c = (int)(i / a.Cexp);
    a.Wp_f[i] = a.Wp[i] * (a.g2[c] * rsqrtf(a.v2[c] + a.eps2));
  }
  for (long c = tid; c < a.Cout; c += nth) {
    const float s = a.g2[c] * rsqrtf(a.v2[c] + a.eps2);
    a.bp[c] = a.b2[c] - s * a.m2[c];
  }
}

extern "C" {

int mx_fold_block(const float* We, const float* g0, const float* b0, const float* m0, const float* v0, float eps0,
                  const float* g1, const float* b1, const float* m1, const float* v1, float eps1,
                  const float* Wp, const float* g2, const float* b2, const float* m2, const float* v2, float eps2,
                  int Cin, int Cexp, int Cout, float* We_f, float* be, float* s1, float* t1, float* Wp_f, float* bp, void* stream) {
  MX_CHECK_ARG(Cin > 0 && Cexp > 0 && Cout > 0, "fold_block: bad extents");
  MX_CHECK_ARG(!We || (g0 && b0 && m0 && v0 && We_f && be), "fold_block: expand weight without its BatchNorm / outputs");
  MX_CHECK_ARG(g1 && b1 && m1 && v1 && s1 && t1 && Wp && g2 && b2 && m2 && v2 && Wp_f && bp, "fold_block: null pointer");
  FoldArgs a{We, g0, b0, m0, v0, g1, b1, m1, v1, Wp, g2, b2, m2, v2, eps0, eps1, eps2, Cin, Cexp, Cout, We_f, be, s1, t1, Wp_f, bp};
  long work = (long)Cexp * (Cin > Cout ? Cin : Cout);
  int grid = (int)((work + 1023) / 1024);
  if (grid > 1024) grid = 1024;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(fold_block_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_colreduce_parts(long rows, int C) {
  if (rows <= 0 || C <= 0 || C % 4) return MX_EARG;
  return colreduce_parts(rows, C);
}

int mx_colstats(const float* X, long rows, int C, float* part, void* stream) {
  MX_CHECK_ARG(X && part && rows > 0 && C > 0 && C % 4 == 0, "colstats: bad args rows=%ld C=%d", rows, C);
  FStats f{X, C};
  launch_colreduce<FStats, 2, false, float>(f, rows, C, 1, part, part, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bn_finalize(const float* part, int P, int C, double count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                   float* mean, float* rstd, double* acc, void* stream) {
  MX_CHECK_ARG(C > 0 && gamma && beta && running_mean && running_var && scale && shift && mean && rstd,
               "bn_finalize: null argument");
  MX_CHECK_ARG(!training || (part && P > 0 && count > 0 && acc), "bn_finalize: training needs partial statistics, count and acc[64][2C]");
  if (training && P <= FUSED_FINALIZE_MAX_ROWS) {
    BnFwdFin f{count, gamma, beta, running_mean, running_var, momentum, eps, 1, scale, shift, mean, rstd};
    hipLaunchKernelGGL(bn_reduce_finalize_kernel<false>, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, part, P, C, f, BnBwdFin{});
    MX_LAUNCH_CHECK();
    return MX_OK;
  }
  int slices = 0;
  if (training) slices = launch_parts_reduce(part, P, C, acc, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, acc, slices, C, count, gamma,
                     beta, running_mean, running_var, momentum, eps, training, scale, shift, mean, rstd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bn_apply(const float* P, const float* scale, const float* shift, const float* row_scale, const float* residual,
                const float* gate, float* out, long rows, int C, int rows_per_sample, int act, void* stream) {
  MX_CHECK_ARG(P && scale && shift && out && rows > 0 && C % 4 == 0 && rows_per_sample > 0, "bn_apply: bad args");
  long total4 = rows * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, P, scale, shift,
                     row_scale, residual, gate, out, total4, C, rows_per_sample, act);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// sums[2C] += (sum g_eff, sum g_eff * X)
int mx_bn_bwd_reduce(const float* G, const float* X, const float* row_scale, const float* gate, const float* gate_add,
                     const float* act_scale, const float* act_shift, long rows, int C, int rows_per_sample,
                     float* part, void* stream) {
  MX_CHECK_ARG(G && X && part && rows > 0 && C % 4 == 0 && rows_per_sample > 0, "bn_bwd_reduce: bad args");
  MX_CHECK_ARG((gate == nullptr) == (gate_add == nullptr), "bn_bwd_reduce: gate and gate_add come together");
  MX_CHECK_ARG((act_scale == nullptr) == (act_shift == nullptr), "bn_bwd_reduce: act scale/shift come together");
  FBwdReduce f{GEff{G, X, row_scale, gate, gate_add, act_scale, act_shift, C, rows_per_sample}};
  launch_colreduce<FBwdReduce, 2, false, float>(f, rows, C, rows_per_sample, part, part, (hipStream_t)stream);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bn_bwd_finalize(const float* part, int P, int C, double count, const float* gamma, const float* mean, const float* rstd,
                       int training, float* dgamma, float* dbeta, float* c1, float* c2, float* c3, double* acc, void* stream) {
  MX_CHECK_ARG(part && P > 0 && gamma && mean && rstd && dgamma && dbeta && c1 && c2 && c3 && acc && C > 0 && count > 0,
               "bn_bwd_finalize: bad args");
  if (P <= FUSED_FINALIZE_MAX_ROWS) {
    BnBwdFin b{count, gamma, mean, rstd, training, dgamma, dbeta, c1, c2, c3};
    hipLaunchKernelGGL(bn_reduce_finalize_kernel<true>, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, part, P, C, BnFwdFin{}, b);
    MX_LAUNCH_CHECK();
    return MX_OK;
  }
  const int slices = launch_parts_reduce(part, P, C, acc, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, acc, slices, C, count, gamma,
                     mean, rstd, training, dgamma, dbeta, c1, c2, c3);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bn_bwd_apply(const float* G, const float* X, const float* row_scale, const float* gate, const float* gate_add,
                    const float* act_scale, const float* act_shift, const float* c1, const float* c2, const float* c3,
                    float* out, long rows, int C, int rows_per_sample, void* stream) {
  MX_CHECK_ARG(G && X && out && c1 && c2 && c3 && rows > 0 && C % 4 == 0 && rows_per_sample > 0, "bn_bwd_apply: bad args");
  GEff e{G, X, row_scale, gate, gate_add, act_scale, act_shift, C, rows_per_sample};
  long total4 = rows * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, e, c1, c2, c3, out,
                     total4);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// bytes of scratch mx_pool_sum (planes = 1) / mx_se_bn1_pool (planes = 5) need; 0 = one row block per sample, no scratch
long mx_pool_ws(long rows, int C, int rows_per_sample, int planes) {
  if (rows <= 0 || C <= 0 || C % 4 || rows_per_sample <= 0 || rows % rows_per_sample || (planes != 1 && planes != 5)) return MX_EARG;
  return pool_ws_bytes(rows, C, rows_per_sample, planes);
}

// out5[5][N][C] = the five per-(sample, channel) sums described at se_bn1_pool_kernel (overwritten, deterministic)
int mx_se_bn1_pool(const float* dA, const float* X, const float* scale, const float* shift, long rows, int C, int rows_per_sample,
                   float* out5, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(dA && X && scale && shift && out5 && rows > 0 && C % 4 == 0 && rows_per_sample > 0 && rows % rows_per_sample == 0,
               "se_bn1_pool: bad args");
  const int N = (int)(rows / rows_per_sample);
  dim3 grid;
  ColGeom g = pool_geom(rows, C, rows_per_sample, &grid, 5);
  const long need = pool_ws_bytes(rows, C, rows_per_sample, 5);
  MX_CHECK_ARG(need == 0 || (ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0), "se_bn1_pool: workspace of %ld bytes required (mx_pool_ws)", need);
  hipLaunchKernelGGL(se_bn1_pool_kernel, grid, dim3(256), 0, (hipStream_t)stream, dA, X, scale, shift, g, out5, (long)N * C,
                     need ? reinterpret_cast<float*>((char*)ws + MX_WS_COUNTER_BYTES) : nullptr, reinterpret_cast<unsigned*>(ws));
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bn1_sums_finalize(const float* pooled5, const float* gate, const float* gh, const float* W1, float inv_hw, float* add,
                         int N, int C, int SQ, double count, const float* gamma, const float* mean, const float* rstd, int training,
                         float* dgamma, float* dbeta, float* c1, float* c2, float* c3, void* stream) {
  MX_CHECK_ARG(pooled5 && gate && gh && W1 && add && N > 0 && C > 0 && SQ > 0 && count > 0, "bn1_sums_finalize: bad args");
  MX_CHECK_ARG(gamma && mean && rstd && dgamma && dbeta && c1 && c2 && c3, "bn1_sums_finalize: null pointer");
  const size_t redb = 2 * B1_NL * 64 * sizeof(double);
  const size_t shb = (size_t)((N * SQ + 1) & ~1) * sizeof(float) + redb;
  BnBwdFin b{count, gamma, mean, rstd, training, dgamma, dbeta, c1, c2, c3};
  if (shb <= 64 * 1024)
    hipLaunchKernelGGL(bn1_sums_finalize_kernel<true>, dim3(cdiv(C, 64)), dim3(64 * B1_NL), shb, (hipStream_t)stream, pooled5, gate, gh, W1,
                       inv_hw, add, N, C, SQ, b);
  else      // the table does not fit the default dynamic-LDS limit: read it from global memory (L2-resident, N*SQ*4 bytes)
    hipLaunchKernelGGL(bn1_sums_finalize_kernel<false>, dim3(cdiv(C, 64)), dim3(64 * B1_NL), redb, (hipStream_t)stream, pooled5, gate, gh, W1,
                       inv_hw, add, N, C, SQ, b);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// out[n,c] = sum_hw f(X[n,hw,c]) with f = [affine a,b] [swish] [* G] (overwritten, deterministic); used for the SE squeeze
// (model.py:82), the global average pool of the head (MuSCLe.py:240) and their backward reductions.
int mx_pool_sum(const float* X, const float* G, const float* scale, const float* shift, int act, long rows, int C,
                int rows_per_sample, float* out, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(X && out && rows > 0 && C % 4 == 0 && rows_per_sample > 0 && rows % rows_per_sample == 0,
               "pool_sum: bad args rows=%ld C=%d rps=%d", rows, C, rows_per_sample);
  const long need = pool_ws_bytes(rows, C, rows_per_sample, 1);
  MX_CHECK_ARG(need == 0 || (ws && ws_bytes >= need && ((uintptr_t)ws & 15) == 0), "pool_sum: workspace of %ld bytes required (mx_pool_ws)", need);
  FPool f{X, G, scale, shift, C, act};
  launch_colreduce<FPool, 1, true, float>(f, rows, C, rows_per_sample, out, out, (hipStream_t)stream, need ? ws : nullptr);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
