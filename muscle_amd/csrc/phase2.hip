// Phase-2 kernels of the MCL step (epochs >= 8 / >= 12): cam_maxnorm, PixPro, channel L2 normalisation,
// dynamic crop extraction and the Sinkhorn-EMD matching.
//
// Reference: train_mcl.py:21-28 (cam_maxnorm), src/loss_multilabel.py:93-105 (PixPro),
// F.normalize(dim=1) at train_mcl.py:218-219, src/torchutils.py:217-291 (get_dynamic_crops: window crop,
// bilinear(align_corners=True) resize, 4x4 average pool), src/loss_multilabel.py:201-257,287-326 (EMD.dynamic_matching:
// cost 1 - x.y, weights, 10 log-domain Sinkhorn iterations, min-score pair re-evaluated with gradient).
// Crop features are packed as [pixels, 24] fp32 rows (21 classes + 3 zeros) so a cost entry is six float4 FMAs;
// the cost matrix itself is never stored.
#include "common.h"

constexpr int FP = 24;   // padded feature width

// ---------------------------------------------------------------------------
// cam_maxnorm: y = relu((relu(x) - mn - 1e-6) / (mx - mn + 1e-6)), mn/mx over HW per (n,k)
// stats[nk] = {mn, mx, argmin, argmax} (indices stored as float bit patterns of ints)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxnorm_fwd_kernel(const float* x, float* y, float* stats, long HW) {
  __shared__ float smn[4], smx[4];
  __shared__ int simn[4], simx[4];
  const long nk = blockIdx.x;
  const float* xp = x + nk * HW;
  float mn = INFINITY, mx = -INFINITY;
  int imn = 0x7fffffff, imx = 0x7fffffff;
  for (long i = threadIdx.x; i < HW; i += 256) {
    float v = fmaxf(xp[i], 0.f);
    if (v < mn) { mn = v; imn = (int)i; }
    if (v > mx) { mx = v; imx = (int)i; }
  }
  // wave reduce keeping the lowest index on ties
  for (int o = 32; o > 0; o >>= 1) {
    float omn = __shfl_xor(mn, o, 64), omx = __shfl_xor(mx, o, 64);
    int oimn = __shfl_xor(imn, o, 64), oimx = __shfl_xor(imx, o, 64);
    if (omn < mn || (omn == mn && oimn < imn)) { mn = omn; imn = oimn; }
    if (omx > mx || (omx == mx && oimx < imx)) { mx = omx; imx = oimx; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { smn[wave] = mn; smx[wave] = mx; simn[wave] = imn; simx[wave] = imx; }
  __syncthreads();
  mn = smn[0]; mx = smx[0]; imn = simn[0]; imx = simx[0];
  for (int w = 1; w < 4; ++w) {
    if (smn[w] < mn || (smn[w] == mn && simn[w] < imn)) { mn = smn[w]; imn = simn[w]; }
    if (smx[w] > mx || (smx[w] == mx && simx[w] < imx)) { mx = smx[w]; imx = simx[w]; }
  }
  const float inv = 1.f / (mx - mn + 1e-6f);
  for (long i = threadIdx.x; i < HW; i += 256) y[nk * HW + i] = fmaxf((fmaxf(xp[i], 0.f) - mn - 1e-6f) * inv, 0.f);
  if (threadIdx.x == 0) {
    stats[nk * 4 + 0] = mn; stats[nk * 4 + 1] = mx;
    stats[nk * 4 + 2] = __int_as_float(imn); stats[nk * 4 + 3] = __int_as_float(imx);
  }
}

__global__ __launch_bounds__(256) void maxnorm_bwd_kernel(const float* x, const float* stats, const float* gy, float* gx, long HW) {
  __shared__ float red[2][4];
  const long nk = blockIdx.x;
  const float* xp = x + nk * HW;
  const float* gp = gy + nk * HW;
  const float mn = stats[nk * 4], mx = stats[nk * 4 + 1];
  const int imn = __float_as_int(stats[nk * 4 + 2]), imx = __float_as_int(stats[nk * 4 + 3]);
  const float d = mx - mn + 1e-6f, inv = 1.f / d;
  float s1 = 0.f, s2 = 0.f;   // sum g_u/d , sum g_u*(r-mn-1e-6)/d^2
  for (long i = threadIdx.x; i < HW; i += 256) {
    float r = fmaxf(xp[i], 0.f);
    float num = r - mn - 1e-6f;
    float gu = (num > 0.f) ? gp[i] : 0.f;
    s1 += gu * inv;
    s2 += gu * num * inv * inv;
    gx[nk * HW + i] = (xp[i] > 0.f) ? gu * inv : 0.f;
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    s2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    float gmn = -s1 + s2, gmx = -s2;     // d/d mn, d/d mx
    if (xp[imn] > 0.f) gx[nk * HW + imn] += gmn;
    if (xp[imx] > 0.f) gx[nk * HW + imx] += gmx;
  }
}

// ---------------------------------------------------------------------------
// PixPro: loss = 1 - mean_n mean_window cos(m*f1, m*f2) over the channel dim; g1 = d loss / d f1
// g1 must be zero-filled.  The per-pixel cosines (|cos| <= 1) are summed per sample as 64-bit fixed-point integers (x 2^32):
// integer atomics are associative, so the loss has the same bits every run; pixpro_finish_kernel forms 1 - mean_n mean_window.
// ---------------------------------------------------------------------------
#define PIXPRO_FIX 4294967296.0f      /* 2^32 */
__global__ __launch_bounds__(256) void pixpro_kernel(const float* f1, const float* f2, const float* mask, const long* c1,
                                                     const long* c2, unsigned long long* ssum, float* g1, int N, int K, int H, int W) {
  const int n = blockIdx.y;
  const long h1 = c1[n * 4], w1 = c1[n * 4 + 1], hl = c1[n * 4 + 2], wl = c1[n * 4 + 3];
  const long h2 = c2[n * 4], w2 = c2[n * 4 + 1];
  const long HW = (long)H * W, cnt = hl * wl;
  const float scale = 1.f / ((float)N * (float)cnt);
  float acc = 0.f;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < cnt; p += (long)gridDim.x * 256) {
    long i = p / wl, j = p % wl;
    const float* a = f1 + (long)n * K * HW + (h1 + i) * W + (w1 + j);
    const float* b = f2 + (long)n * K * HW + (h2 + i) * W + (w2 + j);
    float av[32], bv[32];
    float dot = 0.f, na = 0.f, nb = 0.f;
    for (int k = 0; k < K; ++k) {
      float m = mask ? mask[n * K + k] : 1.f;
      av[k] = a[k * HW] * m; bv[k] = b[k * HW] * m;
      dot += av[k] * bv[k]; na += av[k] * av[k]; nb += bv[k] * bv[k];
    }
    na = sqrtf(na); nb = sqrtf(nb);
    float nac = fmaxf(na, 1e-8f), nbc = fmaxf(nb, 1e-8f);
    float cs = dot / (nac * nbc);
    acc += cs;
    float* g = g1 + (long)n * K * HW + (h1 + i) * W + (w1 + j);
    for (int k = 0; k < K; ++k) {
      float m = mask ? mask[n * K + k] : 1.f;
      float gv = bv[k] / (nac * nbc);
      if (na > 1e-8f) gv -= cs * av[k] / (na * na);
      g[k * HW] = -gv * scale * m;
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(ssum + n, (unsigned long long)(long long)(acc * PIXPRO_FIX));
}

__global__ void pixpro_finish_kernel(const unsigned long long* ssum, const long* c1, int N, float* loss) {
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += 64) {
    const double cnt = (double)c1[n * 4 + 2] * (double)c1[n * 4 + 3];
    if (cnt > 0) acc += (double)(long long)ssum[n] / (double)PIXPRO_FIX / cnt;
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) loss[0] = (float)(1.0 - acc / N);
}

// ---------------------------------------------------------------------------
// F.normalize(x, dim=1) on NCHW: y = x / max(||x||_2, 1e-12); bwd: gx = (gy - y (y.gy)) / max(||x||, eps)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chan_l2norm_kernel(const float* x, const float* gy, float* out, int N, int K, long HW, int bwd) {
  const long total = (long)N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    const float* xb = x + n * K * HW + p;
    float v[32];
    float s = 0.f;
    for (int k = 0; k < K; ++k) { v[k] = xb[k * HW]; s += v[k] * v[k]; }
    float nr = fmaxf(sqrtf(s), 1e-12f), inv = 1.f / nr;
    float* ob = out + n * K * HW + p;
    if (!bwd) {
      for (int k = 0; k < K; ++k) ob[k * HW] = v[k] * inv;
    } else {
      const float* gb = gy + n * K * HW + p;
      float g[32], dot = 0.f;
      for (int k = 0; k < K; ++k) { g[k] = gb[k * HW]; dot += g[k] * v[k] * inv; }
      for (int k = 0; k < K; ++k) ob[k * HW] = (g[k] - v[k] * inv * dot) * inv;
    }
  }
}

// ---------------------------------------------------------------------------
// crops.  One table row (8 ints) per crop: {sample, y0, x0, lh, lw, rh, rw, out_offset(pixels)}:
// out[off + (r*rw + c), k] = bilinear_align_corners(src[sample, k, y0:y0+lh, x0:x0+lw] -> (rh, rw))
// (rh == lh and rw == lw is the identity crop used for the second view)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void bil_coord2(int d, int in, int out, int& i0, int& i1, float& w1) {
  float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  float s = scale * d;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - i0;
}

// grid (crops, 3): workgroup (crop, channel group of 8) - a crop per workgroup left the chip a third full at ~900 crops
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* src, const int* table, float* out, int K, int H, int W) {
  const int* t = table + blockIdx.x * 8;
  const int kbeg = blockIdx.y * (FP / 3), kend = kbeg + FP / 3;
  const int n = t[0], y0 = t[1], x0 = t[2], lh = t[3], lw = t[4], rh = t[5], rw = t[6], off = t[7];
  const long HW = (long)H * W;
  for (int p = threadIdx.x; p < rh * rw; p += 256) {
    int r = p / rw, c = p % rw;
    int ya, yb, xa, xb;
    float wy, wx;
    bil_coord2(r, lh, rh, ya, yb, wy);
    bil_coord2(c, lw, rw, xa, xb, wx);
    const float* b = src + (long)n * K * HW;
    float* o = out + ((long)off + p) * FP;
    for (int k = kbeg; k < kend; ++k) {
      float v = 0.f;
      if (k < K) {
        const float* q = b + k * HW;
        float a00 = q[(long)(y0 + ya) * W + x0 + xa], a01 = q[(long)(y0 + ya) * W + x0 + xb];
        float a10 = q[(long)(y0 + yb) * W + x0 + xa], a11 = q[(long)(y0 + yb) * W + x0 + xb];
        v = (1.f - wy) * ((1.f - wx) * a00 + wx * a01) + wy * ((1.f - wx) * a10 + wx * a11);
      }
      o[k] = v;
    }
  }
}

// adjoint: gsrc[n,k,...] += W^T gout[off ...] for every crop of the table.  Crops overlap and a bilinear footprint is shared by
// neighbouring output pixels, so the scatter form needs atomics (order-dependent bits).  Gather form instead: workgroup
// (sample n, class group kg) walks the crops of ITS sample in table order; for one crop every thread owns source pixels of
// the window and adds, over the few output pixels whose footprint names it (<= 4 x 4 when the resize shrinks, which it does
// here: torchutils.py:254-262), weight x gradient for its KG classes; a barrier separates crops.  One adder per element,
// fixed order: same bits every run.
constexpr int CROP_KG = 7;     // classes per workgroup (21 = 3 x 7)
__device__ __forceinline__ void adj_range(int si, int in, int out, int& lo, int& hi) {
  // output indices d whose bil_coord2(d) can name source index si: scale*d in (si - 1, si + 1); one index of margin each side
  if (out <= 1 || in <= 1) { lo = 0; hi = out - 1; return; }
  const float inv = (float)(out - 1) / (float)(in - 1);
  lo = max(0, (int)floorf((float)(si - 1) * inv) - 1);
  hi = min(out - 1, (int)ceilf((float)(si + 1) * inv) + 1);
}
__global__ __launch_bounds__(256) void crop_resize_bwd_kernel(const float* gout_all, const int* table, int ncrops, float* gsrc, int K,
                                                              int H, int W) {
  const int n = blockIdx.x, k0 = blockIdx.y * CROP_KG;
  const long HW = (long)H * W;
  // grid.z cuts the image into bands of rows: workgroup (n, kg, band) owns the band's source pixels (96 workgroups for a batch
  // of 32 left the chip idle: 680 us).  An element still has one adder, and still meets its crops in table order.
  const int bh = (H + gridDim.z - 1) / gridDim.z, by0 = blockIdx.z * bh, by1 = min(H, by0 + bh);
  for (int ci = 0; ci < ncrops; ++ci) {
    const int* trow = table + ci * 8;
    if (trow[0] != n) continue;                   // (uniform over the workgroup)
    const int y0 = trow[1], x0 = trow[2], lh = trow[3], lw = trow[4], rh = trow[5], rw = trow[6];
    const int s_lo = max(0, by0 - y0), s_hi = min(lh, by1 - y0);      // window rows inside this band
    if (s_lo >= s_hi) continue;                   // (uniform: the barrier below is skipped by the whole workgroup)
    const float* gout = gout_all + (long)trow[7] * FP;
    for (int sp = threadIdx.x; sp < (s_hi - s_lo) * lw; sp += 256) {
      const int sy = s_lo + sp / lw, sx = sp % lw;
      int rlo, rhi, clo, chi;
      adj_range(sy, lh, rh, rlo, rhi);
      adj_range(sx, lw, rw, clo, chi);
      float acc[CROP_KG];
#pragma unroll
      for (int j = 0; j < CROP_KG; ++j) acc[j] = 0.f;
      for (int r = rlo; r <= rhi; ++r) {
        int ya, yb; float wy;
        bil_coord2(r, lh, rh, ya, yb, wy);
        const float wyy = (ya == sy ? 1.f - wy : 0.f) + (yb == sy ? wy : 0.f);     // (both terms when ya == yb, as the forward)
        if (wyy == 0.f) continue;
        for (int c = clo; c <= chi; ++c) {
          int xa, xb; float wx;
          bil_coord2(c, lw, rw, xa, xb, wx);
          const float wxx = (xa == sx ? 1.f - wx : 0.f) + (xb == sx ? wx : 0.f);
          if (wxx == 0.f) continue;
          const float* g = gout + (long)(r * rw + c) * FP + k0;
          const float wgt = wyy * wxx;
#pragma unroll
          for (int j = 0; j < CROP_KG; ++j) acc[j] += wgt * g[j];
        }
      }
#pragma unroll
      for (int j = 0; j < CROP_KG; ++j)
        if (k0 + j < K && acc[j] != 0.f) gsrc[((long)n * K + k0 + j) * HW + (long)(y0 + sy) * W + x0 + sx] += acc[j];
    }
    __syncthreads();                              // the next crop of this sample may touch the same pixels
  }
}

// 4x4 average pool (stride 4, floor) over packed crops.  table row (4 ints): {in_off, h, w, out_off}
__global__ __launch_bounds__(256) void avgpool4_kernel(const float* in, const int* table, float* out, int bwd) {
  const int* t = table + blockIdx.x * 4;
  const int ioff = t[0], h = t[1], w = t[2], ooff = t[3];
  const int ph = h / 4, pw = w / 4;
  if (!bwd) {
    for (int i = threadIdx.x + 256 * blockIdx.y; i < ph * pw * FP; i += 256 * gridDim.y) {
      int k = i % FP, p = i / FP;
      int r = p / pw, c = p % pw;
      float s = 0.f;
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) s += in[((long)ioff + (4 * r + a) * w + 4 * c + b) * FP + k];
      out[((long)ooff + p) * FP + k] = s * (1.f / 16.f);
    }
  } else {   // reads the pooled crop's gradient (at out_off), writes the unpooled crop's gradient (at in_off)
    for (int i = threadIdx.x + 256 * blockIdx.y; i < h * w * FP; i += 256 * gridDim.y) {
      int k = i % FP, p = i / FP;
      int r = p / w, c = p % w;
      float v = 0.f;
      if (r / 4 < ph && c / 4 < pw) v = in[((long)ooff + (r / 4) * pw + c / 4) * FP + k] * (1.f / 16.f);
      out[((long)ioff + p) * FP + k] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Sinkhorn EMD.  pair table row (6 ints): {x_off, n1, y_off, n2, sample, reserved}
// ---------------------------------------------------------------------------
constexpr int EMD_MAXP = 1024;   // max pixels per crop
constexpr float EMD_REG = 0.1f;
constexpr int EMD_ITERS = 10;

struct F24 { float4 a, b, c, d, e, f; };
__device__ __forceinline__ F24 ld24(const float* p) {
  F24 r; r.a = ld4(p); r.b = ld4(p + 4); r.c = ld4(p + 8); r.d = ld4(p + 12); r.e = ld4(p + 16); r.f = ld4(p + 20); return r;
}
__device__ __forceinline__ float dot4(float4 x, float4 y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
__device__ __forceinline__ float dot24(const F24& x, const float* y) {
  return dot4(x.a, ld4(y)) + dot4(x.b, ld4(y + 4)) + dot4(x.c, ld4(y + 8)) + dot4(x.d, ld4(y + 12)) + dot4(x.e, ld4(y + 16)) +
         dot4(x.f, ld4(y + 20));
}

// Work decomposition.  A sweep gives every "owner" (a row i for the u update, a column j for the v update) the log-sum-exp of
// its line of M.  One owner per thread leaves most of a 1024-thread workgroup idle (a 49 x 272 pair has 49 rows) and makes the
// line loop the critical path, so the line of owner o is cut into S = 2^k <= 16 contiguous slices, S * n_own <= 1024: thread
// (s, o) = threadIdx / n_own, threadIdx % n_own reduces slice s, the partials meet in LDS (pm / ps, indexed by threadIdx) and
// thread o joins them in slice order.  Neighbouring lanes share s, so the other side's vector is an LDS broadcast read.
constexpr int EMD_T = 1024;       // threads per workgroup
constexpr int EMD_MAXS = 16;
static_assert(EMD_MAXP <= EMD_T, "one owner per thread");

struct EmdSlice { int idx, s, S, lo, hi; };
__device__ __forceinline__ EmdSlice emd_slice(int n_own, int n_other) {
  int S = 1;
  while (S < EMD_MAXS && 2 * S * n_own <= EMD_T) S *= 2;
  EmdSlice e;
  e.S = S;
  e.s = threadIdx.x / n_own; e.idx = threadIdx.x - e.s * n_own;
  const int chunk = (n_other + S - 1) / S;
  e.lo = e.s * chunk; e.hi = min(n_other, e.lo + chunk);
  if (e.s >= S) { e.idx = -1; e.lo = e.hi = 0; }
  return e;
}

// (m, s): max and sum exp(M - m) of M_oj = (a_o . b_j - 1 + pa_o + pb_j) / reg over j in [lo, hi), four entries at a time
__device__ __forceinline__ void emd_lse_part(const F24& ao, float pao, const float* bs, const float* pb, int lo, int hi, float& m,
                                             float& s) {
  const float ir = 1.f / EMD_REG;
  m = -INFINITY; s = 0.f;
  int j = lo;
  for (; j + 4 <= hi; j += 4) {
    const float M0 = (dot24(ao, bs + j * FP) - 1.f + pao + pb[j]) * ir;
    const float M1 = (dot24(ao, bs + (j + 1) * FP) - 1.f + pao + pb[j + 1]) * ir;
    const float M2 = (dot24(ao, bs + (j + 2) * FP) - 1.f + pao + pb[j + 2]) * ir;
    const float M3 = (dot24(ao, bs + (j + 3) * FP) - 1.f + pao + pb[j + 3]) * ir;
    const float mn = fmaxf(fmaxf(fmaxf(M0, M1), fmaxf(M2, M3)), m);
    s = s * __expf(m - mn) + ((__expf(M0 - mn) + __expf(M1 - mn)) + (__expf(M2 - mn) + __expf(M3 - mn)));
    m = mn;
  }
  for (; j < hi; ++j) {
    const float M = (dot24(ao, bs + j * FP) - 1.f + pao + pb[j]) * ir;
    const float mn = fmaxf(M, m);
    s = s * __expf(m - mn) + __expf(M - mn);
    m = mn;
  }
}

// log-sum-exp of owner o's whole line from its S slice partials, slices in order
__device__ __forceinline__ float emd_lse_join(const float* pm, const float* ps, int o, int n_own, int S) {
  float m = -INFINITY;
  for (int s = 0; s < S; ++s) m = fmaxf(m, pm[s * n_own + o]);
  float t = 0.f;
  for (int s = 0; s < S; ++s) {
    const float ms = pm[s * n_own + o];
    if (ms > -INFINITY) t += ps[s * n_own + o] * __expf(ms - m);
  }
  return m + __logf(t);
}

// shared layout: xs[n1*24], ys[n2*24], u[n1], v[n2], lmu[n1], lnu[n2], tmp[n1], pm[1024], ps[1024]  (+ gradient: cl[n2], ub[n1], vb[n2])
// XG ("x in global"): the launch holds a pair whose two crops do not fit the 160 KB together (two 28 x 28 crops need 150 KB of
// features alone).  Then only y is staged and x is read where it lies - every lane of a wave asks for the same row, one cache
// line per request - so that any pair of crops up to EMD_MAXP pixels each runs, at a lower rate.
struct EmdLds { float *xs, *ys, *u, *v, *lmu, *lnu, *tmp, *pm, *ps, *cl, *ub, *vb; };
template <bool XG>
__device__ __forceinline__ EmdLds emd_carve(float* sh, int n1, int n2) {
  EmdLds L;
  L.xs = sh; L.ys = L.xs + (XG ? 0 : n1 * FP); L.u = L.ys + n2 * FP; L.v = L.u + n1; L.lmu = L.v + n2; L.lnu = L.lmu + n1; L.tmp = L.lnu + n2;
  L.pm = L.tmp + n1; L.ps = L.pm + EMD_T; L.cl = L.ps + EMD_T; L.ub = L.cl + n2; L.vb = L.ub + n1;
  return L;
}

// One Jacobi iteration: both updates use the same (u, v) (loss_multilabel.py:215-217).
__device__ __forceinline__ void emd_iterate(const EmdLds& L, const float* xs, const EmdSlice& r, const EmdSlice& c, int n1, int n2) {
  const int tid = threadIdx.x;
  // rows: u_i' = reg*(lmu_i - LSE_j M_ij) + u_i
  if (r.idx >= 0) {
    const F24 xi = ld24(xs + r.idx * FP);
    float m, s;
    emd_lse_part(xi, L.u[r.idx], L.ys, L.v, r.lo, r.hi, m, s);
    L.pm[tid] = m; L.ps[tid] = s;
  }
  __syncthreads();
  if (tid < n1) L.tmp[tid] = EMD_REG * (L.lmu[tid] - emd_lse_join(L.pm, L.ps, tid, n1, r.S)) + L.u[tid];
  __syncthreads();
  // columns (use OLD u): v_j' = reg*(lnu_j - LSE_i M_ij) + v_j
  if (c.idx >= 0) {
    const F24 yj = ld24(L.ys + c.idx * FP);
    float m, s;
    emd_lse_part(yj, L.v[c.idx], xs, L.u, c.lo, c.hi, m, s);
    L.pm[tid] = m; L.ps[tid] = s;
  }
  __syncthreads();
  if (tid < n2) L.v[tid] = EMD_REG * (L.lnu[tid] - emd_lse_join(L.pm, L.ps, tid, n2, c.S)) + L.v[tid];
  if (tid < n1) L.u[tid] = L.tmp[tid];
  __syncthreads();
}

template <bool XG>
__device__ __forceinline__ void emd_setup(const float* X, const float* Y, const EmdLds& L, const float* xs, int n1, int n2) {
  if (!XG) for (int i = threadIdx.x; i < n1 * FP; i += EMD_T) L.xs[i] = X[i];
  for (int i = threadIdx.x; i < n2 * FP; i += EMD_T) L.ys[i] = Y[i];
  __syncthreads();
  // weights (get_weight_vector, :250-257): mu_i = x_i . mean(y), nu_j = y_j . mean(x); pm[0..23], pm[24..47] hold the means
  if (threadIdx.x < 2 * FP) {
    int k = threadIdx.x % FP;
    const float* src = (threadIdx.x < FP) ? L.ys : xs;
    int n = (threadIdx.x < FP) ? n2 : n1;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += src[i * FP + k];
    L.pm[threadIdx.x] = s / n;
  }
  __syncthreads();
  float my[FP], mx[FP];
  for (int k = 0; k < FP; ++k) { my[k] = L.pm[k]; mx[k] = L.pm[FP + k]; }
  __syncthreads();
  for (int i = threadIdx.x; i < n1; i += EMD_T) {
    float s = 0.f;
    for (int k = 0; k < FP; ++k) s += xs[i * FP + k] * my[k];
    L.lmu[i] = __logf(s + 1e-6f);
    L.u[i] = 0.f;
  }
  for (int j = threadIdx.x; j < n2; j += EMD_T) {
    float s = 0.f;
    for (int k = 0; k < FP; ++k) s += L.ys[j * FP + k] * mx[k];
    L.lnu[j] = __logf(s + 1e-6f);
    L.v[j] = 0.f;
  }
  __syncthreads();
}

// sum_ij exp(M_ij) * C_ij / (n1*n2): slice partials, then waves in order
__device__ __forceinline__ float emd_distance(const EmdLds& L, const float* xs, const EmdSlice& r, int n1, int n2) {
  const float ir = 1.f / EMD_REG;
  float acc = 0.f;
  if (r.idx >= 0) {
    const F24 xi = ld24(xs + r.idx * FP);
    const float ui = L.u[r.idx];
    for (int j = r.lo; j < r.hi; ++j) {
      float c = 1.f - dot24(xi, L.ys + j * FP);
      acc += __expf((-c + ui + L.v[j]) * ir) * c;
    }
  }
  acc = wave_sum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) L.pm[threadIdx.x >> 6] = acc;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < EMD_T / 64; ++w) t += L.pm[w];
  return t / ((float)n1 * (float)n2);
}

// One workgroup per pair, in table order: the host sorts the table by descending n1 * n2, so the long pairs start first and
// the short ones fill the tail.  traj (optional): (u_t, v_t) of the 11 Sinkhorn states, which mx_emd_grad differentiates.
template <bool XG>
__global__ __launch_bounds__(EMD_T) void emd_score_kernel(const float* feat, const int* pairs, float* score, float* traj,
                                                          long traj_stride) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int* t = pairs + blockIdx.x * 6;
  const int n1 = t[1], n2 = t[3], nn = n1 + n2;
  const EmdLds L = emd_carve<XG>(sh, n1, n2);
  const EmdSlice r = emd_slice(n1, n2), c = emd_slice(n2, n1);
  const float* xs = XG ? feat + (long)t[0] * FP : L.xs;
  emd_setup<XG>(feat + (long)t[0] * FP, feat + (long)t[2] * FP, L, xs, n1, n2);
  float* tr = traj ? traj + blockIdx.x * traj_stride : nullptr;
  for (int it = 0; it <= EMD_ITERS; ++it) {
    if (tr) {
      if (threadIdx.x < n1) tr[it * nn + threadIdx.x] = L.u[threadIdx.x];
      if (threadIdx.x < n2) tr[it * nn + n1 + threadIdx.x] = L.v[threadIdx.x];
    }
    if (it < EMD_ITERS) emd_iterate(L, xs, r, c, n1, n2);
  }
  float d = emd_distance(L, xs, r, n1, n2);
  if (threadIdx.x == 0) score[blockIdx.x] = d;
}

// best[s] = the minimal-score pair of sample s, ties to the lower rank (column 5 = the pair's position in the reference's
// enumeration order: stable sort semantics, :318); loss += score/ns.  Pairs are spread over the threads; the per-sample
// minimum is an LDS atomicMin on the 64-bit key (order-preserving score bits, rank) - a minimum does not depend on arrival order.
__device__ __forceinline__ unsigned emd_ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ __launch_bounds__(1024) void emd_best_kernel(const float* score, const int* pairs, int npairs, int nsamples, int* best, float* loss) {
  __shared__ unsigned long long key[1024];
  __shared__ float bsv[1024];
  const int tid = threadIdx.x;
  key[tid] = ~0ull;
  bsv[tid] = 0.f;
  if (tid < nsamples) best[tid] = -1;
  __syncthreads();
  for (int p = tid; p < npairs; p += 1024) {
    const int s = pairs[p * 6 + 4];
    if (s >= 0 && s < nsamples)
      atomicMin(&key[s], ((unsigned long long)emd_ord(score[p]) << 32) | (unsigned)pairs[p * 6 + 5]);
  }
  __syncthreads();
  for (int p = tid; p < npairs; p += 1024) {
    const int s = pairs[p * 6 + 4];
    if (s >= 0 && s < nsamples && key[s] == (((unsigned long long)emd_ord(score[p]) << 32) | (unsigned)pairs[p * 6 + 5])) {
      best[s] = p;                                // ranks are unique within a table: one writer per sample
      bsv[s] = score[p] / nsamples;
    }
  }
  __syncthreads();
  if (tid == 0) {                                 // one adder, sample order (was a float atomic per sample)
    float t = 0.f;
    for (int i = 0; i < nsamples; ++i) t += bsv[i];
    loss[0] += t;
  }
}

// backward through the 10 iterations for the best pair of each sample: gx[best crop1 pixels, 24] = d dist / d x * gscale.
// traj: the states mx_emd_scores recorded, row `pair` holds (EMD_ITERS+1) * (n1 + n2) floats of (u_t, v_t).
// Same slice decomposition as the forward; every thread keeps the 24 gradient terms of its (row, slice) in registers through
// all stages and the slices of a row are added in order at the end (in the LDS that held x).
template <bool XG>
__global__ __launch_bounds__(EMD_T) void emd_grad_kernel(const float* feat, const int* pairs, const int* best, const float* traj,
                                                         long traj_stride, const float* gup, float gscale, float* gx) {
  extern __shared__ __attribute__((aligned(16))) float sh[];
  const int bi = best[blockIdx.x];
  if (bi < 0) return;
  if (gup) gscale *= gup[0];
  const int* t = pairs + bi * 6;
  const int n1 = t[1], n2 = t[3], nn = n1 + n2, tid = threadIdx.x;
  const EmdLds L = emd_carve<XG>(sh, n1, n2);
  const EmdSlice r = emd_slice(n1, n2), c = emd_slice(n2, n1);
  const float* tr = traj + bi * traj_stride;
  const float* xs = XG ? feat + (long)t[0] * FP : L.xs;
  emd_setup<XG>(feat + (long)t[0] * FP, feat + (long)t[2] * FP, L, xs, n1, n2);
  const float ir = 1.f / EMD_REG, inv12 = 1.f / ((float)n1 * (float)n2);
  float gxl[FP];
#pragma unroll
  for (int k = 0; k < FP; ++k) gxl[k] = 0.f;
  // final stage, on the last state: Mbar_ij = pi_ij * Cd_ij / (n1 n2); Cbar += -Mbar/reg (the explicit Cd factor is detached);
  // ubar_i = sum_j Mbar/reg, vbar_j = sum_i Mbar/reg
  if (tid < n1) L.u[tid] = tr[EMD_ITERS * nn + tid];
  if (tid < n2) L.v[tid] = tr[EMD_ITERS * nn + n1 + tid];
  __syncthreads();
  {
    float us = 0.f, vs = 0.f;
    if (r.idx >= 0) {
      const F24 xi = ld24(xs + r.idx * FP);
      const float ui = L.u[r.idx];
      for (int j = r.lo; j < r.hi; ++j) {
        const float* y = L.ys + j * FP;
        const float cst = 1.f - dot24(xi, y);
        const float mb = __expf((-cst + ui + L.v[j]) * ir) * cst * inv12 * ir;   // Mbar / reg
        us += mb;
        // Cbar_ij = -mb ; dC/dx_i = -y_j  ->  gx_i += mb * y_j
#pragma unroll
        for (int k = 0; k < FP; ++k) gxl[k] += mb * y[k];
      }
    }
    if (c.idx >= 0) {
      const F24 yj = ld24(L.ys + c.idx * FP);
      const float vj = L.v[c.idx];
      for (int i = c.lo; i < c.hi; ++i) {
        const float cst = 1.f - dot24(yj, xs + i * FP);
        vs += __expf((-cst + L.u[i] + vj) * ir) * cst * inv12 * ir;
      }
    }
    L.pm[tid] = us; L.ps[tid] = vs;
  }
  for (int it = EMD_ITERS - 1; it >= 0; --it) {
    __syncthreads();                              // the sweeps of the stage above are done: its u, v, tmp, cl, ub, vb are free
    // join the slice partials of the stage above into the adjoints this stage consumes, and stage this stage's state:
    // u, v = state `it`; tmp_i = rowLSE_i = lmu_i - (u'_i - u_i)/reg ; cl_j = colLSE_j = lnu_j - (v'_j - v_j)/reg
    const float* ut = tr + it * nn;
    const float* un = ut + nn;
    if (tid < n1) {
      float ubn = 0.f;
      for (int s = 0; s < r.S; ++s) ubn += L.pm[s * n1 + tid];
      const float a = ut[tid];
      L.ub[tid] = ubn; L.u[tid] = a; L.tmp[tid] = L.lmu[tid] - (un[tid] - a) * ir;
    }
    if (tid < n2) {
      float vbn = 0.f;
      for (int s = 0; s < c.S; ++s) vbn += L.ps[s * n2 + tid];
      const float a = ut[n1 + tid];
      L.vb[tid] = vbn; L.v[tid] = a; L.cl[tid] = L.lnu[tid] - (un[n1 + tid] - a) * ir;
    }
    __syncthreads();                              // pm / ps are read, the stage's vectors are in place
    // P_ij = exp(M_ij - rowLSE_i), Q_ij = exp(M_ij - colLSE_j)
    // Cbar_ij += ub'_i P_ij + vb'_j Q_ij  ->  gx_i -= (...) y_j ; ub_i = -sum_j vb'_j Q_ij ; vb_j = -sum_i ub'_i P_ij
    float nu_ = 0.f, acc = 0.f;
    if (r.idx >= 0) {
      const F24 xi = ld24(xs + r.idx * FP);
      const float ui = L.u[r.idx], rl = L.tmp[r.idx], ubi = L.ub[r.idx];
      for (int j = r.lo; j < r.hi; ++j) {
        const float* y = L.ys + j * FP;
        const float M = (dot24(xi, y) - 1.f + ui + L.v[j]) * ir;
        const float P = __expf(M - rl);
        const float vq = L.vb[j] * __expf(M - L.cl[j]);
        const float cb = ubi * P + vq;
        nu_ -= vq;
#pragma unroll
        for (int k = 0; k < FP; ++k) gxl[k] -= cb * y[k];
      }
    }
    if (c.idx >= 0) {
      const F24 yj = ld24(L.ys + c.idx * FP);
      const float vj = L.v[c.idx];
      for (int i = c.lo; i < c.hi; ++i) {
        const float M = (dot24(yj, xs + i * FP) - 1.f + L.u[i] + vj) * ir;
        acc -= L.ub[i] * __expf(M - L.tmp[i]);
      }
    }
    L.pm[tid] = nu_; L.ps[tid] = acc;
  }
  // slices of a row, in order: into the LDS that held x, or (XG) straight into the output rows - one workgroup owns them
  float* out = gx + (long)t[0] * FP;
  __syncthreads();
  for (int s = 0; s < r.S; ++s) {
    if (r.idx >= 0 && r.s == s) {
      float* d = (XG ? out : L.xs) + r.idx * FP;
#pragma unroll
      for (int k = 0; k < FP; ++k) {
        if (XG) {      // read-modify-write of global memory between waves: device-scope accesses, past the CU's L1
          const float prev = s ? __hip_atomic_load(d + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
          __hip_atomic_store(d + k, prev + gxl[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          d[k] = (s ? d[k] : 0.f) + gxl[k];
        }
      }
    }
    if (XG) __threadfence();
    __syncthreads();
  }
  if (XG) {
    for (int e = tid; e < n1 * FP; e += EMD_T)
      out[e] = __hip_atomic_load(out + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * gscale;
  } else {
    for (int e = tid; e < n1 * FP; e += EMD_T) out[e] = L.xs[e] * gscale;
  }
}

static int gs(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

extern "C" {

int mx_maxnorm(const float* x, const float* gy, float* out, float* stats, int NK, long HW, int bwd, void* stream) {
  MX_CHECK_ARG(x && out && stats && NK > 0 && HW > 0 && HW < 0x7fffffff && (!bwd || gy), "maxnorm: bad args");
  if (!bwd) hipLaunchKernelGGL(maxnorm_fwd_kernel, dim3(NK), dim3(256), 0, (hipStream_t)stream, x, out, stats, HW);
  else hipLaunchKernelGGL(maxnorm_bwd_kernel, dim3(NK), dim3(256), 0, (hipStream_t)stream, x, stats, gy, out, HW);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_pixpro(const float* f1, const float* f2, const float* mask, const long* coord1, const long* coord2, float* loss, float* g1,
              int N, int K, int H, int W, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(f1 && f2 && coord1 && coord2 && loss && g1 && N > 0 && K > 0 && K <= 32 && H > 0 && W > 0, "pixpro: bad args");
  MX_CHECK_ARG(ws && ws_bytes >= (long)N * 8 && ((uintptr_t)ws & 7) == 0, "pixpro: %d bytes of scratch required", N * 8);
  hipMemsetAsync(ws, 0, (size_t)N * 8, (hipStream_t)stream);
  hipLaunchKernelGGL(pixpro_kernel, dim3(cdiv((long)H * W, 256), N), dim3(256), 0, (hipStream_t)stream, f1, f2, mask, coord1, coord2,
                     (unsigned long long*)ws, g1, N, K, H, W);
  hipLaunchKernelGGL(pixpro_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned long long*)ws, coord1, N, loss);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_chan_l2norm(const float* x, const float* gy, float* out, int N, int K, long HW, int bwd, void* stream) {
  MX_CHECK_ARG(x && out && N > 0 && K > 0 && K <= 32 && HW > 0 && (!bwd || gy), "chan_l2norm: bad args");
  hipLaunchKernelGGL(chan_l2norm_kernel, dim3(gs((long)N * HW)), dim3(256), 0, (hipStream_t)stream, x, gy, out, N, K, HW, bwd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_crop_resize(const float* src, const int* table, int ncrops, float* out, int K, int H, int W, void* stream) {
  MX_CHECK_ARG(src && table && out && ncrops > 0 && K > 0 && K <= FP, "crop_resize: bad args");
  hipLaunchKernelGGL(crop_resize_kernel, dim3(ncrops, 3), dim3(256), 0, (hipStream_t)stream, src, table, out, K, H, W);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_crop_resize_bwd(const float* gout, const int* table, int ncrops, float* gsrc, int nsamples, int K, int H, int W, void* stream) {
  MX_CHECK_ARG(gout && table && gsrc && ncrops > 0 && nsamples > 0 && K > 0 && K <= FP, "crop_resize_bwd: bad args");
  // samples = 1 + the largest sample index of the table is not known here: the grid covers `nsamples` given by the caller
  hipLaunchKernelGGL(crop_resize_bwd_kernel, dim3(nsamples, cdiv(K, CROP_KG), H >= 64 ? 8 : 1), dim3(256), 0, (hipStream_t)stream, gout, table,
                     ncrops, gsrc, K, H, W);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_avgpool4(const float* in, const int* table, int ncrops, float* out, int bwd, void* stream) {
  MX_CHECK_ARG(in && table && out && ncrops > 0, "avgpool4: bad args");
  hipLaunchKernelGGL(avgpool4_kernel, dim3(ncrops, 8), dim3(256), 0, (hipStream_t)stream, in, table, out, bwd);   // 8 slices of a crop's elements
  MX_LAUNCH_CHECK();
  return MX_OK;
}

static size_t emd_lds(int maxn1, int maxn2, int grad, int xg) {
  size_t f = (size_t)((xg ? 0 : maxn1) + maxn2) * FP + 2 * (size_t)(maxn1 + maxn2) + (size_t)maxn1 + 2 * EMD_T;
  if (grad) f += (size_t)(maxn1 + 2 * maxn2);
  return f * sizeof(float);
}

int mx_emd_scores(const float* feat, const int* pairs, int npairs, int maxn1, int maxn2, float* score, float* traj, void* stream) {
  MX_CHECK_ARG(feat && pairs && score && npairs > 0, "emd_scores: bad args");
  MX_CHECK_ARG(maxn1 > 0 && maxn2 > 0 && maxn1 <= EMD_MAXP && maxn2 <= EMD_MAXP, "emd_scores: crop larger than %d pixels", EMD_MAXP);
  const int xg = emd_lds(maxn1, maxn2, 0, 0) > 160 * 1024;        // the largest pair does not fit: x stays in global memory
  size_t sh = emd_lds(maxn1, maxn2, 0, xg);
  MX_CHECK_ARG(sh <= 160 * 1024, "emd_scores: LDS need %zu exceeds 160 KiB", sh);
  const long stride = (long)(EMD_ITERS + 1) * (maxn1 + maxn2);
  if (xg) {
    hipFuncSetAttribute((const void*)emd_score_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(emd_score_kernel<true>, dim3(npairs), dim3(EMD_T), sh, (hipStream_t)stream, feat, pairs, score, traj, stride);
  } else {
    if (sh > 48 * 1024) hipFuncSetAttribute((const void*)emd_score_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(emd_score_kernel<false>, dim3(npairs), dim3(EMD_T), sh, (hipStream_t)stream, feat, pairs, score, traj, stride);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_emd_best(const float* score, const int* pairs, int npairs, int nsamples, int* best, float* loss, void* stream) {
  MX_CHECK_ARG(score && pairs && best && loss && npairs > 0 && nsamples > 0 && nsamples <= 1024, "emd_best: bad args (nsamples <= 1024)");
  hipLaunchKernelGGL(emd_best_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, pairs, npairs, nsamples, best, loss);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_emd_grad(const float* feat, const int* pairs, const int* best, int nsamples, int maxn1, int maxn2, const float* traj,
                const float* gup, float gscale, float* gx, void* stream) {
  MX_CHECK_ARG(feat && pairs && best && traj && gx && nsamples > 0, "emd_grad: bad args");
  MX_CHECK_ARG(maxn1 > 0 && maxn2 > 0 && maxn1 <= EMD_MAXP && maxn2 <= EMD_MAXP, "emd_grad: crop larger than %d pixels", EMD_MAXP);
  const int xg = emd_lds(maxn1, maxn2, 1, 0) > 160 * 1024;
  size_t sh = emd_lds(maxn1, maxn2, 1, xg);
  MX_CHECK_ARG(sh <= 160 * 1024, "emd_grad: LDS need %zu exceeds 160 KiB", sh);
  const long stride = (long)(EMD_ITERS + 1) * (maxn1 + maxn2);
  if (xg) {
    hipFuncSetAttribute((const void*)emd_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(emd_grad_kernel<true>, dim3(nsamples), dim3(EMD_T), sh, (hipStream_t)stream, feat, pairs, best, traj, stride, gup,
                       gscale, gx);
  } else {
    if (sh > 48 * 1024) hipFuncSetAttribute((const void*)emd_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
    hipLaunchKernelGGL(emd_grad_kernel<false>, dim3(nsamples), dim3(EMD_T), sh, (hipStream_t)stream, feat, pairs, best, traj, stride, gup,
                       gscale, gx);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
