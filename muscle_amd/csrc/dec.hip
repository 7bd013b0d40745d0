// Decoder-mode (config 4) kernels: BiFPN support ops, cross entropy against the arg-max pseudo-label, gradient
// clipping, and the BEACON FieldLoss.
//
// Reference: src/MuSCLe.py:30-58 (_BIFPN_Layer: 3x3/s2 average pools and bilinear resizes between 1x1 convs),
// train_muscle.py:189-202 (argmax pseudo-label, CrossEntropyLoss, clip_grad_norm_(9)), src/edge.py:25-89,175-440
// (Mix_fg + 5x5 Sobel, OrientQuantize, FieldLoss.in_out_div / bifilter / loss_constructor).
#include "common.h"

static int gs(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

// ---------------------------------------------------------------------------
// F.avg_pool2d(k=3, s=2, p=1) (count_include_pad) on NHWC and its adjoint
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool3s2_kernel(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, int bwd) {
  const int c4n = C / 4;
  if (!bwd) {
    const long total = (long)N * Ho * Wo * c4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      int c = (int)(i % c4n) * 4;
      long p = i / c4n;
      int ox = (int)(p % Wo); long q = p / Wo;
      int oy = (int)(q % Ho), n = (int)(q / Ho);
      float4 s = make_float4(0, 0, 0, 0);
      for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
          int iy = 2 * oy + dy, ix = 2 * ox + dx;
          if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            float4 v = ld4(x + (((long)n * H + iy) * W + ix) * C + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
          }
        }
      const float k = 1.f / 9.f;
      st4(y + p * C + c, make_float4(s.x * k, s.y * k, s.z * k, s.w * k));
    }
  } else {   // x = grad of the pooled map [N,Ho,Wo,C], y = grad of the input [N,H,W,C]
    const long total = (long)N * H * W * c4n;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      int c = (int)(i % c4n) * 4;
      long p = i / c4n;
      int ix = (int)(p % W); long q = p / W;
      int iy = (int)(q % H), n = (int)(q / H);
      float4 s = make_float4(0, 0, 0, 0);
      for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy)          // 2*oy-1 <= iy <= 2*oy+1
        for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox)
          if (oy < Ho && ox < Wo) {
            float4 v = ld4(x + (((long)n * Ho + oy) * Wo + ox) * C + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
          }
      const float k = 1.f / 9.f;
      st4(y + p * C + c, make_float4(s.x * k, s.y * k, s.z * k, s.w * k));
    }
  }
}

__device__ __forceinline__ void bc(int d, int in, int out, int& i0, int& i1, float& w1) {
  float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  float s = scale * d;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - i0;
}

// destination indices whose bilinear footprint (bc) can name source index si: scale*d in (si - 1, si + 1), one index of margin
__device__ __forceinline__ void adj_rng(int si, int in, int out, int& lo, int& hi) {
  if (out <= 1 || in <= 1) { lo = 0; hi = out - 1; return; }
  const float inv = (float)(out - 1) / (float)(in - 1);
  lo = max(0, (int)floorf((float)(si - 1) * inv) - 1);
  hi = min(out - 1, (int)ceilf((float)(si + 1) * inv) + 1);
}

// adjoint of mx_resize_nhwc (no relu): gsrc[N,Hs,Ws,C] += W^T gdst[N,Hd,Wd,C].  Gather form: a thread owns four channels of one
// SOURCE pixel and adds weight x gradient over the destination pixels that name it, rows then columns ascending - one adder per
// element, same bits every run (the scatter form needed float atomics).
__global__ __launch_bounds__(256) void resize_nhwc_bwd_kernel(const float* gdst, float* gsrc, int N, int Hs, int Ws, int C, int Hd, int Wd) {
  const int c4n = C / 4;
  const long total = (long)N * Hs * Ws * c4n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % c4n) * 4;
    long p = i / c4n;
    const int sx = (int)(p % Ws); long q = p / Ws;
    const int sy = (int)(q % Hs), n = (int)(q / Hs);
    int ylo, yhi, xlo, xhi;
    adj_rng(sy, Hs, Hd, ylo, yhi);
    adj_rng(sx, Ws, Wd, xlo, xhi);
    float4 acc = make_float4(0, 0, 0, 0);
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1; float wy;
      bc(y, Hs, Hd, y0, y1, wy);
      const float wyy = (y0 == sy ? 1.f - wy : 0.f) + ((y1 == sy && y1 != y0) ? wy : 0.f);
      if (wyy == 0.f) continue;
      for (int x = xlo; x <= xhi; ++x) {
        int x0, x1; float wx;
        bc(x, Ws, Wd, x0, x1, wx);
        const float wxx = (x0 == sx ? 1.f - wx : 0.f) + ((x1 == sx && x1 != x0) ? wx : 0.f);
        if (wxx == 0.f) continue;
        const float4 g = ld4(gdst + (((long)n * Hd + y) * Wd + x) * C + c);
        const float w = wyy * wxx;
        acc.x += w * g.x; acc.y += w * g.y; acc.z += w * g.z; acc.w += w * g.w;
      }
    }
    float* o = gsrc + p * C + c;
    float4 v = ld4(o);
    v.x += acc.x; v.y += acc.y; v.z += acc.z; v.w += acc.w;
    st4(o, v);
  }
}


// ---------------------------------------------------------------------------
// CrossEntropyLoss(seg [N,K,HW], argmax_k mask [N,K,HW]) mean over N*HW, and its gradient
// ---------------------------------------------------------------------------
// forward: the per-pixel terms (>= 0) are summed as 64-bit fixed-point integers (x 2^32: associative, same bits every run);
// ce_finish_kernel adds the mean to loss[0]
#define CE_FIX 4294967296.0f
__global__ void ce_finish_kernel(const unsigned long long* acc, double inv, float* loss) {
  loss[0] += (float)((double)acc[0] / (double)CE_FIX * inv);
}
__global__ __launch_bounds__(256) void ce_argmax_kernel(const float* seg, const float* mask, const float* gup, unsigned long long* lacc, float* gseg,
                                                        int N, int K, long HW, int bwd) {
  const long total = (long)N * HW;
  const float inv = 1.f / (float)total;
  float acc = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    const float* s = seg + n * K * HW + p;
    const float* m = mask + n * K * HW + p;
    int t = 0; float mb = m[0], mx = s[0];
    for (int k = 1; k < K; ++k) { float v = m[k * HW]; if (v > mb) { mb = v; t = k; } mx = fmaxf(mx, s[k * HW]); }
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += __expf(s[k * HW] - mx);
    float lse = mx + __logf(se);
    if (!bwd) acc += lse - s[t * HW];
    else {
      float g = gup[0] * inv;
      float* o = gseg + n * K * HW + p;
      for (int k = 0; k < K; ++k) o[k * HW] = g * (__expf(s[k * HW] - lse) - (k == t ? 1.f : 0.f));
    }
  }
  if (!bwd) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && acc != 0.f) atomicAdd(lacc, (unsigned long long)(acc * CE_FIX));
  }
  (void)inv;
}

// ---------------------------------------------------------------------------
// clip_grad_norm_: sq[0] += sum x^2 (fp64); then x *= min(1, max_norm / (sqrt(sq) + 1e-6))
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqsum_kernel(const float* x, long n, double* sq /*[gridDim.x]*/) {
  __shared__ double sh[4];
  double a = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { double v = x[i]; a += v * v; }
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) sq[blockIdx.x] = ((sh[0] + sh[1]) + sh[2]) + sh[3];      // one partial per workgroup (was an fp64 atomic)
}
// total = the nparts partials added in a fixed order (thread t: t, t + 256, ...; then the 256 thread sums in thread order)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* x, long n, const double* sq, int nparts, float max_norm, float* norm_out) {
  __shared__ double sh[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += sq[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x == 0) { double t = 0.0; for (int i = 0; i < 256; ++i) t += sh[i]; sh[0] = t; }
  __syncthreads();
  const float total = (float)sqrt(sh[0]);
  const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = total;
  if (coef >= 1.0f) return;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= coef;
}

// ---------------------------------------------------------------------------
// FieldLoss stage 1: p = softmax(beta*seg)[1:], per-class 5x5 Sobel (kernel entries 1e-6 for "zero", edge.py:38-44),
// magnitude sqrt(gx^2+gy^2+1e-8), orientation quantised to 8 sectors of 3.1416/8 (edge.py:65-89).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void field_prob_kernel(const float* seg, float beta, float* prob, int N, int K, long HW) {
  const long total = (long)N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    const float* s = seg + n * K * HW + p;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, s[k * HW] * beta);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(s[k * HW] * beta - mx);
    for (int k = 1; k < K; ++k) prob[n * (K - 1) * HW + (k - 1) * HW + p] = expf(s[k * HW] * beta - mx) / se;
  }
}

__constant__ float kGx[25] = {2.0f, 1.0f, 1e-6f, -1.0f, -2.0f, 3.0f, 2.0f, 1e-6f, -2.0f, -3.0f, 4.0f, 3.0f, 0.0f, -3.0f, -4.0f,
                              3.0f, 2.0f, 1e-6f, -2.0f, -3.0f, 2.0f, 1.0f, 1e-6f, -1.0f, -2.0f};
__constant__ float kGy[25] = {2.0f, 3.0f, 4.0f, 3.0f, 2.0f, 1.0f, 2.0f, 3.0f, 2.0f, 1.0f, 1e-6f, 1e-6f, 1e-6f, 1e-6f, 1e-6f,
                              -1.0f, -2.0f, -3.0f, -2.0f, -1.0f, -2.0f, -3.0f, -4.0f, -3.0f, -2.0f};

__device__ __forceinline__ int quantize_orient(float gx, float gy) {
  const float div = 3.1416f / 8.f;
  float o = atan2f(gy, gx);
  if (3 * div > o && o >= div) return 0;
  if (5 * div > o && o >= 3 * div) return 1;
  if (7 * div > o && o >= 5 * div) return 2;
  if ((8 * div > o && o >= 7 * div) || (-7 * div > o && o >= -8 * div)) return 3;
  if (-5 * div > o && o >= -7 * div) return 4;
  if (-3 * div > o && o >= -5 * div) return 5;
  if (-1 * div > o && o >= -3 * div) return 6;
  return 7;
}

// mag [N,F,HW], orient u8 [N,F,HW], mx [N,F] (float bits, atomicMax; zero-filled), edge_fg [N,HW] = sum_f mag
__global__ __launch_bounds__(256) void field_edge_kernel(const float* prob, const float* lab, float* mag, unsigned char* orient,
                                                         unsigned* mx, float* edge_fg, int N, int F, int H, int W) {
  // Per-(sample, class) maximum of the magnitude: every pixel used to atomicMax the same global word (200 704 pixels x the
  // labelled classes per sample: 16.4 ms of contended atomics at [16,20,448,448]).  Now: wave maximum -> one LDS atomicMax
  // per wave -> one global atomicMax per workgroup and class (magnitudes are positive: the float bits order like uints).
  __shared__ unsigned smax[64];
  __shared__ long sn0;
  const long HW = (long)H * W, total = (long)N * HW;
  const float m_zero = sqrtf(1e-8f);                       // an unlabelled class: gx = gy = 0 exactly
  const unsigned char o_zero = (unsigned char)quantize_orient(0.f, 0.f);
  const long stride = (long)gridDim.x * 256;
  const long iters = (total + stride - 1) / stride;        // uniform trip count: the body has barriers
  for (long it = 0; it < iters; ++it) {
    const long i = it * stride + blockIdx.x * 256L + threadIdx.x;
    const bool act = i < total;
    const long n = act ? i / HW : -1, p = act ? i - n * HW : 0;
    if (threadIdx.x == 0) sn0 = n;
    if (threadIdx.x < 64) smax[threadIdx.x] = 0u;
    __syncthreads();
    const long n0 = sn0;
    const int y = (int)(p / W), x = (int)(p % W);
    float esum = 0.f;
    for (int f = 0; f < F; ++f) {
      const float l = act ? lab[n * F + f] : 0.f;
      float m = m_zero;
      unsigned char o = o_zero;
      if (l != 0.f) {
        float gx = 0.f, gy = 0.f;
        const float* pp = prob + (n * F + f) * HW;
        for (int ky = 0; ky < 5; ++ky)
          for (int kx = 0; kx < 5; ++kx) {
            int iy = y + ky - 2, ix = x + kx - 2;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
              float v = pp[(long)iy * W + ix];
              gx += kGx[ky * 5 + kx] * v; gy += kGy[ky * 5 + kx] * v;
            }
          }
        gx *= l; gy *= l;
        m = sqrtf(gx * gx + gy * gy + 1e-8f);
        o = (unsigned char)quantize_orient(gx, gy);
      }
      if (act) {
        mag[(n * F + f) * HW + p] = m;
        orient[(n * F + f) * HW + p] = o;
      }
      esum += m;
      // maximum over the labelled pixels of this class
      const bool same = act && n == n0 && l != 0.f;
      if (__any(same)) {                                    // (skips the 17 unlabelled classes of a sample)
        const float wm = wave_max(same ? m : 0.f);
        if ((threadIdx.x & 63) == 0 && wm > 0.f && f < 64) atomicMax(&smax[f], __float_as_uint(wm));
      }
      if (act && l != 0.f && (n != n0 || f >= 64)) atomicMax(mx + n * F + f, __float_as_uint(m));   // workgroup straddles two samples
    }
    if (act) edge_fg[i] = esum;
    __syncthreads();
    if (threadIdx.x < 64 && threadIdx.x < F && n0 >= 0 && smax[threadIdx.x] != 0u) atomicMax(mx + n0 * F + threadIdx.x, smax[threadIdx.x]);
    __syncthreads();
  }
}

// stage 2: one workgroup per labelled (sample, class) slot: boundary pixels (mag >= 0.8 max, max > 1) in row-major
// order -> out / in index lists by FieldLoss.in_out_div (edge.py:196-229, orient+1 in [1,8]) with margins removed.
// slots rows: {n, f}; lists [S][HW] ints; counts [S][3] = {n_out, n_in, n_pos}
__global__ __launch_bounds__(256) void field_select_kernel(const float* mag, const unsigned char* orient, const unsigned* mx,
                                                           const int* slots, int step, int* out_list, int* in_list, int* counts,
                                                           int F, int H, int W) {
  __shared__ int wsum[2][4];
  __shared__ int base[3];
  const int s = blockIdx.x, n = slots[2 * s], f = slots[2 * s + 1];
  const long HW = (long)H * W;
  const float m = __uint_as_float(mx[n * F + f]);
  const float* mg = mag + ((long)n * F + f) * HW;
  const unsigned char* og = orient + ((long)n * F + f) * HW;
  if (threadIdx.x < 3) base[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int npos = 0;
  for (long p0 = 0; p0 < HW; p0 += 256) {
    long p = p0 + threadIdx.x;
    bool pos = false;
    float oi = 0.f, ii = 0.f;
    if (p < HW) {
      pos = (mg[p] >= 0.8f * m) && (m > 1.f);
      if (pos) {
        int o = og[p] + 1;
        float ind = (float)p;
        // outs = ind + (-step)^(1+(o<4)) * ((o%4==0)*w) + (-1)^(1+o) * ((o==2)|(o==6))
        float a = (o % 4 == 0) ? (float)W * ((o < 4) ? (float)(step * step) : (float)(-step)) : 0.f;
        float b = (o == 2 || o == 6) ? (((1 + o) & 1) ? -1.f : 1.f) : 0.f;
        oi = ind + a + b;
        // ins = ind + (-step)^(o<4) * ((o%4==0)*w) + (-1)^o * ((o==2)|(o==6))
        float a2 = (o % 4 == 0) ? (float)W * ((o < 4) ? (float)(-step) : 1.f) : 0.f;
        float b2 = (o == 2 || o == 6) ? ((o & 1) ? -1.f : 1.f) : 0.f;
        ii = ind + a2 + b2;
      }
    }
    auto keep = [&](float v) {
      float r = fmodf(v, (float)(W - 1));
      if (r < 0.f) r += (float)(W - 1);
      return r != 0.f && r != 1.f && v > 0.f && v < (float)(HW - 1);
    };
    bool ko = pos && keep(oi), ki = pos && keep(ii);
    unsigned long long bo = __ballot(ko), bi = __ballot(ki);
    unsigned long long lower = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int po = __popcll(bo & lower), pi = __popcll(bi & lower);
    if (lane == 0) { wsum[0][wave] = __popcll(bo); wsum[1][wave] = __popcll(bi); }
    npos += pos ? 1 : 0;
    __syncthreads();
    int offo = base[0], offi = base[1];
    for (int w2 = 0; w2 < wave; ++w2) { offo += wsum[0][w2]; offi += wsum[1][w2]; }
    if (ko) out_list[(long)s * HW + offo + po] = (int)oi;
    if (ki) in_list[(long)s * HW + offi + pi] = (int)ii;
    __syncthreads();
    if (threadIdx.x == 0) {
      base[0] += wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
      base[1] += wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
    }
    __syncthreads();
  }
  npos = (int)wave_sum((float)npos);
  if (lane == 0) atomicAdd(&base[2], npos);
  __syncthreads();
  if (threadIdx.x == 0) { counts[3 * s] = base[0]; counts[3 * s + 1] = base[1]; counts[3 * s + 2] = base[2]; }
}

// stage 3: softmax features at the sampled points.  pts rows {n, pixel}; one wave per point.
// dense: mode 0 = full-res NCHW [N,CH,H,W]; mode 1 = low-res NHWC [N,h,w,CH] upsampled on the fly (align_corners)
__global__ __launch_bounds__(256) void field_gather_kernel(const float* dense, int mode, int h, int w, const float* mask, const int* pts,
                                                           int npts, float* feat, float* mfeat, int CH, int K, int ML, int H, int W) {
  const int lane = threadIdx.x & 63;
  const long HW = (long)H * W;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < npts; q += gridDim.x * 4) {
    const int n = pts[2 * q], p = pts[2 * q + 1];
    const int y = p / W, x = p % W;
    int y0 = 0, y1 = 0, x0 = 0, x1 = 0; float wy = 0.f, wx = 0.f;
    if (mode == 1) { bc(y, h, H, y0, y1, wy); bc(x, w, W, x0, x1, wx); }
    float v[16];   // CH <= 1024
    float mxv = -INFINITY;
    int cnt = 0;
    for (int c = lane; c < CH; c += 64, ++cnt) {
      float val;
      if (mode == 0) val = dense[((long)n * CH + c) * HW + p];
      else {
        const float* b = dense + (long)n * h * w * CH + c;
        val = (1.f - wy) * ((1.f - wx) * b[((long)y0 * w + x0) * CH] + wx * b[((long)y0 * w + x1) * CH]) +
              wy * ((1.f - wx) * b[((long)y1 * w + x0) * CH] + wx * b[((long)y1 * w + x1) * CH]);
      }
      v[cnt] = val; mxv = fmaxf(mxv, val);
    }
    mxv = wave_max(mxv);
    float se = 0.f; cnt = 0;
    for (int c = lane; c < CH; c += 64, ++cnt) { v[cnt] = expf(v[cnt] - mxv); se += v[cnt]; }
    se = wave_sum(se); cnt = 0;
    for (int c = lane; c < CH; c += 64, ++cnt) feat[(long)q * CH + c] = v[cnt] / se;
    // mask softmax over K classes (lanes < K), padded to ML columns with zeros
    float mv = (lane < K) ? mask[((long)n * K + lane) * HW + p] : -INFINITY;
    float mm = wave_max(mv);
    float me = (lane < K) ? expf(mv - mm) : 0.f;
    float ms = wave_sum(me);
    if (lane < ML) mfeat[(long)q * ML + lane] = (lane < K) ? me / ms : 0.f;
  }
}

// stage 5: per slot, from sim [k,k] (outs x ins) and sim_m: the eight FP/FN/TP/TN terms (edge.py:231-261,330-347),
// loss += sum / nsamples, gsim = d loss / d sim.  One workgroup per slot, k <= 256.
__global__ void field_loss_sum_kernel(const float* slot_loss, int nslots, float* loss) {     // slots in order, one adder
  float t = 0.f;
  for (int i = 0; i < nslots; ++i) t += slot_loss[i];
  loss[0] += t;
}
__global__ __launch_bounds__(256) void field_terms_kernel(const float* sim, const float* simm, int k, float inv_n, float* slot_loss,
                                                          float* gsim) {
  __shared__ float rm[256], cm[256], rmm[256], cmm[256];
  __shared__ float tot[2];
  __shared__ int cnt[2][4];
  __shared__ float ssum[2][4];
  const int s = blockIdx.x, t = threadIdx.x;
  const float* S = sim + (long)s * k * k;
  const float* M = simm + (long)s * k * k;
  if (t < k) {
    float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
    for (int j = 0; j < k; ++j) { a += S[t * k + j]; b += S[j * k + t]; c += M[t * k + j]; d += M[j * k + t]; }
    rm[t] = a / k; cm[t] = b / k; rmm[t] = c / k; cmm[t] = d / k;
  }
  __syncthreads();
  if (t == 0) {
    float a = 0.f, c = 0.f;
    for (int i = 0; i < k; ++i) { a += rm[i]; c += rmm[i]; }
    tot[0] = a / k; tot[1] = c / k;
  }
  __syncthreads();
  // class of row t (dim=1) and of column t (dim=0): 0 FP, 1 FN, 2 TP, 3 TN
  int cls[2] = {0, 0};
  __shared__ unsigned char clsm[2][256];
  if (t < k) {
    for (int d = 0; d < 2; ++d) {
      bool sm = (d == 0 ? rmm[t] : cmm[t]) > tot[1];
      bool sd = (d == 0 ? rm[t] : cm[t]) > tot[0];
      cls[d] = (sm && !sd) ? 0 : ((!sm && sd) ? 1 : ((!sm && !sd) ? 2 : 3));
      clsm[d][t] = (unsigned char)cls[d];
    }
  }
  __syncthreads();
  if (t < 8) {                                   // thread (d, c): count and sum of class c along dimension d, rows in order
    const int d = t / 4, c = t % 4;
    int n_ = 0;
    float a = 0.f;
    for (int i = 0; i < k; ++i)
      if (clsm[d][i] == c) { ++n_; a += (d == 0 ? rm[i] : cm[i]); }
    cnt[d][c] = n_; ssum[d][c] = a;
  }
  __syncthreads();
  if (t == 0) {
    float l = 0.f;
    for (int d = 0; d < 2; ++d)
      for (int c = 0; c < 4; ++c)
        if (cnt[d][c] > 0) l += ((c == 0 || c == 3) ? -1.f : 1.f) * ssum[d][c] / cnt[d][c];
    slot_loss[s] = l * inv_n;
  }
  // gsim[i][j] = sign(cls_row i)/(k*cnt_row) + sign(cls_col j)/(k*cnt_col), scaled by inv_n
  __shared__ float rw[256], cw[256];
  if (t < k) {
    rw[t] = ((cls[0] == 0 || cls[0] == 3) ? -1.f : 1.f) / ((float)k * cnt[0][cls[0]]) * inv_n;
    cw[t] = ((cls[1] == 0 || cls[1] == 3) ? -1.f : 1.f) / ((float)k * cnt[1][cls[1]]) * inv_n;
  }
  __syncthreads();
  for (int i = t; i < k * k; i += 256) gsim[(long)s * k * k + i] = rw[i / k] + cw[i % k];
}

// stage 6: softmax backward at the out points and scatter of the gradient into the dense feature gradient.
// gfeat [npts, CH] = d loss / d softmaxed features (out points only); feat the forward softmax.
// Two forms.  (a) mode 1 (the training step: features interpolated from the low-resolution NHWC map): field_point_grad_kernel
// writes the per-point gradient rows gpt [npts, CH]; field_tile_gather_kernel - one workgroup per (sample, 8 x 8 tile of the
// low-resolution map), thread = channel - walks ALL points in list order and adds the bilinear corner weights of the points
// that fall into its tile into an LDS tile (a thread only ever touches its own channel column), then adds the tile to gdense:
// one adder per element, points in order: same bits every run.  (b) mode 0 (materialised full-resolution features, the public
// unfused path): the round-1 scatter with float atomics (last-bit noise), kept for that path only.
__global__ __launch_bounds__(256) void field_point_grad_kernel(const float* feat, const float* gfeat, int npts, const float* gup,
                                                               float* gpt, int CH) {
  const int lane = threadIdx.x & 63;
  const float gs_ = gup ? gup[0] : 1.f;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < npts; q += gridDim.x * 4) {
    float dot = 0.f;
    for (int c = lane; c < CH; c += 64) dot += gfeat[(long)q * CH + c] * feat[(long)q * CH + c];
    dot = wave_sum(dot);
    for (int c = lane; c < CH; c += 64) gpt[(long)q * CH + c] = feat[(long)q * CH + c] * (gfeat[(long)q * CH + c] - dot) * gs_;
  }
}

constexpr int FT = 8;      // tile edge (cells of the low-resolution map)
__global__ __launch_bounds__(256) void field_tile_gather_kernel(const float* gpt, const int* pts, int npts, int h, int w, float* gdense,
                                                                int CH, int H, int W) {
  extern __shared__ float lacc[];                  // [FT*FT][CH]
  const int tiles_x = (w + FT - 1) / FT;
  const int ty0 = (blockIdx.x / tiles_x) * FT, tx0 = (blockIdx.x % tiles_x) * FT, n = blockIdx.y;
  for (int i = threadIdx.x; i < FT * FT * CH; i += 256) lacc[i] = 0.f;
  __syncthreads();
  bool any = false;
  for (int q = 0; q < npts; ++q) {
    if (pts[2 * q] != n) continue;                 // (uniform: every thread reads the same words)
    const int p = pts[2 * q + 1];
    int y0, y1, x0, x1; float wy, wx;
    bc(p / W, h, H, y0, y1, wy); bc(p % W, w, W, x0, x1, wx);
    if (y1 < ty0 || y0 >= ty0 + FT || x1 < tx0 || x0 >= tx0 + FT) continue;
    any = true;
    const float w00 = (1.f - wy) * (1.f - wx), w01 = (x1 != x0) ? (1.f - wy) * wx : 0.f;
    const float w10 = (y1 != y0) ? wy * (1.f - wx) : 0.f, w11 = (y1 != y0 && x1 != x0) ? wy * wx : 0.f;
    for (int c = threadIdx.x; c < CH; c += 256) {
      const float g = gpt[(long)q * CH + c];
      auto add = [&](int yy, int xx, float wgt) {
        if (wgt != 0.f && yy >= ty0 && yy < ty0 + FT && xx >= tx0 && xx < tx0 + FT) lacc[((yy - ty0) * FT + (xx - tx0)) * CH + c] += wgt * g;
      };
      add(y0, x0, w00); add(y0, x1, w01); add(y1, x0, w10); add(y1, x1, w11);
    }
  }
  if (!any) return;                                // (uniform)
  __syncthreads();
  for (int i = threadIdx.x; i < FT * FT * CH; i += 256) {
    const int c = i % CH, cell = i / CH, yy = ty0 + cell / FT, xx = tx0 + cell % FT;
    if (yy < h && xx < w && lacc[i] != 0.f) gdense[(((long)n * h + yy) * w + xx) * CH + c] += lacc[i];
  }
}

__global__ __launch_bounds__(256) void field_scatter_kernel(const float* feat, const float* gfeat, const int* pts, int npts, int mode,
                                                            int h, int w, const float* gup, float* gdense, int CH, int H, int W) {
  const int lane = threadIdx.x & 63;
  const long HW = (long)H * W;
  const float gs_ = gup ? gup[0] : 1.f;
  for (int q = blockIdx.x * 4 + (threadIdx.x >> 6); q < npts; q += gridDim.x * 4) {
    const int n = pts[2 * q], p = pts[2 * q + 1];
    float dot = 0.f;
    for (int c = lane; c < CH; c += 64) dot += gfeat[(long)q * CH + c] * feat[(long)q * CH + c];
    dot = wave_sum(dot);
    for (int c = lane; c < CH; c += 64) {
      float g = feat[(long)q * CH + c] * (gfeat[(long)q * CH + c] - dot) * gs_;
      unsafeAtomicAdd(gdense + ((long)n * CH + c) * HW + p, g);
    }
  }
}

extern "C" {

int mx_avgpool3s2(const float* x, float* y, int N, int H, int W, int C, int bwd, void* stream) {
  MX_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C % 4 == 0, "avgpool3s2: bad args");
  int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  long tot = bwd ? (long)N * H * W * (C / 4) : (long)N * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(avgpool3s2_kernel, dim3(gs(tot)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo, bwd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_resize_nhwc_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int C, int Hd, int Wd, void* stream) {
  MX_CHECK_ARG(gdst && gsrc && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0, "resize_nhwc_bwd: bad args");
  MX_CHECK_ARG(C % 4 == 0 && (((uintptr_t)gdst | (uintptr_t)gsrc) & 15) == 0,
               "resize_nhwc_bwd: C=%d must be a multiple of 4 and the tensors 16-byte aligned (the gather moves float4 channel groups)", C);
  hipLaunchKernelGGL(resize_nhwc_bwd_kernel, dim3(gs((long)N * Hs * Ws * (C / 4))), dim3(256), 0, (hipStream_t)stream, gdst, gsrc, N, Hs,
                     Ws, C, Hd, Wd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_ce_argmax(const float* seg, const float* mask, const float* gup, float* loss, float* gseg, int N, int K, long HW, int bwd,
                 void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(seg && mask && N > 0 && K > 1 && HW > 0 && (bwd ? (gup && gseg) : (loss != nullptr)), "ce_argmax: bad args");
  MX_CHECK_ARG(bwd || (ws && ws_bytes >= 8 && ((uintptr_t)ws & 7) == 0), "ce_argmax: forward needs 8 bytes of scratch");
  hipStream_t st = (hipStream_t)stream;
  if (!bwd) hipMemsetAsync(ws, 0, 8, st);
  hipLaunchKernelGGL(ce_argmax_kernel, dim3(gs((long)N * HW)), dim3(256), 0, st, seg, mask, gup, (unsigned long long*)ws, gseg, N, K, HW, bwd);
  if (!bwd) hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(1), 0, st, (const unsigned long long*)ws, 1.0 / ((double)N * (double)HW), loss);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// sq_scratch: 2048 doubles (one partial square sum per workgroup, added in a fixed order)
int mx_clip_grad_norm(float* grads, long n, float max_norm, double* sq_scratch, float* norm_out, void* stream) {
  MX_CHECK_ARG(grads && sq_scratch && n > 0 && max_norm > 0, "clip_grad_norm: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int nparts = gs(n) > 2048 ? 2048 : gs(n);
  hipLaunchKernelGGL(sqsum_kernel, dim3(nparts), dim3(256), 0, st, grads, n, sq_scratch);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(gs(n) > 4096 ? 4096 : gs(n)), dim3(256), 0, st, grads, n, sq_scratch, nparts, max_norm, norm_out);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_field_edges(const float* seg, const float* lab_fg, float beta, float* prob, float* mag, unsigned char* orient, unsigned* mx,
                   float* edge_fg, int N, int K, int H, int W, void* stream) {
  MX_CHECK_ARG(seg && lab_fg && prob && mag && orient && mx && edge_fg && N > 0 && K > 1 && H > 0 && W > 0, "field_edges: bad args");
  hipStream_t st = (hipStream_t)stream;
  long HW = (long)H * W;
  hipMemsetAsync(mx, 0, sizeof(unsigned) * N * (K - 1), st);
  hipLaunchKernelGGL(field_prob_kernel, dim3(gs((long)N * HW)), dim3(256), 0, st, seg, beta, prob, N, K, HW);
  hipLaunchKernelGGL(field_edge_kernel, dim3(gs((long)N * HW)), dim3(256), 0, st, prob, lab_fg, mag, orient, mx, edge_fg, N, K - 1, H, W);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_field_select(const float* mag, const unsigned char* orient, const unsigned* mx, const int* slots, int nslots, int step,
                    int* out_list, int* in_list, int* counts, int F, int H, int W, void* stream) {
  MX_CHECK_ARG(mag && orient && mx && slots && out_list && in_list && counts && nslots > 0 && W > 2, "field_select: bad args");
  hipLaunchKernelGGL(field_select_kernel, dim3(nslots), dim3(256), 0, (hipStream_t)stream, mag, orient, mx, slots, step, out_list,
                     in_list, counts, F, H, W);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_field_gather(const float* dense, int mode, int h, int w, const float* mask, const int* pts, int npts, float* feat, float* mfeat,
                    int CH, int K, int ML, int H, int W, void* stream) {
  MX_CHECK_ARG(dense && mask && pts && feat && mfeat && npts > 0 && CH > 0 && CH <= 1024 && K <= 64 && ML >= K && ML <= 64,
               "field_gather: bad args");
  hipLaunchKernelGGL(field_gather_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, (hipStream_t)stream, dense, mode, h, w, mask, pts, npts,
                     feat, mfeat, CH, K, ML, H, W);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// slot_loss: nslots floats of scratch (the per-slot terms, added to loss[0] in slot order)
int mx_field_terms(const float* sim, const float* simm, int nslots, int k, float inv_n, float* loss, float* gsim, float* slot_loss,
                   void* stream) {
  MX_CHECK_ARG(sim && simm && loss && gsim && slot_loss && nslots > 0 && k > 0 && k <= 256, "field_terms: bad args (k <= 256)");
  hipLaunchKernelGGL(field_terms_kernel, dim3(nslots), dim3(256), 0, (hipStream_t)stream, sim, simm, k, inv_n, slot_loss, gsim);
  hipLaunchKernelGGL(field_loss_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (const float*)slot_loss, nslots, loss);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// gpt: npts*CH floats of scratch (mode 1 only: the per-point gradient rows; may be NULL for mode 0); nsamples = N of gdense
int mx_field_scatter(const float* feat, const float* gfeat, const int* pts, int npts, int mode, int h, int w, const float* gup,
                     float* gdense, float* gpt, int nsamples, int CH, int H, int W, void* stream) {
  MX_CHECK_ARG(feat && gfeat && pts && gdense && npts > 0 && CH > 0 && nsamples > 0, "field_scatter: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1) {
    MX_CHECK_ARG(gpt != nullptr && h > 0 && w > 0, "field_scatter: mode 1 needs gpt [npts, CH]");
    const size_t sh = (size_t)FT * FT * CH * sizeof(float);
    MX_CHECK_ARG(sh <= 160 * 1024, "field_scatter: %d channels do not fit the LDS tile", CH);
    static bool big = false;
    if (sh > 64 * 1024 && !big) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&field_tile_gather_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      big = true;
    }
    hipLaunchKernelGGL(field_point_grad_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, st, feat, gfeat, npts, gup, gpt, CH);
    hipLaunchKernelGGL(field_tile_gather_kernel, dim3(cdiv(h, FT) * cdiv(w, FT), nsamples), dim3(256), sh, st, (const float*)gpt, pts, npts, h, w,
                       gdense, CH, H, W);
  } else {
    hipLaunchKernelGGL(field_scatter_kernel, dim3(cdiv(npts, 4)), dim3(256), 0, st, feat, gfeat, pts, npts, mode, h, w, gup, gdense, CH, H, W);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
