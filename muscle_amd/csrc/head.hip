// CAM head + PCM (pixel-correlation module) support kernels.
//
// Reference: MuSCLe.forward(cam='cam'/'pix') src/MuSCLe.py:237-279 and MuSCLe.PCM :213-223.
// The contractions (fc, CAM 1x1, fuse 1x1, f^T f, cam*aff) run on the MFMA GEMM (gemm.hip); this file
// holds what sits between them: bilinear(align_corners=True) resampling, the per-pixel L2
// normalisation, the affinity column normalisation and their adjoints.  Low-resolution head tensors
// are NHWC with the class dimension padded to a leading dimension that is a multiple of 4.
#include "common.h"
#include <hip/hip_fp16.h>

// align_corners=True source coordinate (torch upsample_bilinear2d): src = dst * (in-1)/(out-1)
__device__ __forceinline__ void bil_coord(int d, int in, int out, int& i0, int& i1, float& w1) {
  float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  float s = scale * d;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - i0;
}

// dst[n,y,x,coff + c] = [relu] bilinear(src[n,:,:,c])   NHWC -> NHWC (channel slice of a wider tensor)
__global__ __launch_bounds__(256) void resize_nhwc_kernel(const float* src, float* dst, int N, int Hs, int Ws, int C, int Hd,
                                                          int Wd, int ldd, int coff, int relu) {
  const int c4n = C / 4;
  const long total = (long)N * Hd * Wd * c4n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int c = (int)(i % c4n) * 4;
    long p = i / c4n;
    int x = (int)(p % Wd);
    long q = p / Wd;
    int y = (int)(q % Hd), n = (int)(q / Hd);
    int y0, y1, x0, x1;
    float wy, wx;
    bil_coord(y, Hs, Hd, y0, y1, wy);
    bil_coord(x, Ws, Wd, x0, x1, wx);
    const float* b = src + (long)n * Hs * Ws * C + c;
    float4 a00 = ld4(b + ((long)y0 * Ws + x0) * C), a01 = ld4(b + ((long)y0 * Ws + x1) * C);
    float4 a10 = ld4(b + ((long)y1 * Ws + x0) * C), a11 = ld4(b + ((long)y1 * Ws + x1) * C);
    float4 o;
#define BIL(f) o.f = (1.f - wy) * ((1.f - wx) * a00.f + wx * a01.f) + wy * ((1.f - wx) * a10.f + wx * a11.f)
    BIL(x); BIL(y); BIL(z); BIL(w);
#undef BIL
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(dst + p * ldd + coff + c, o);
  }
}

// dst[n,k,Y,X] (NCHW, K classes) = bilinear(src[n,:,:,k]) with src NHWC of leading dimension lds
__global__ __launch_bounds__(256) void upsample_to_nchw_kernel(const float* src, float* dst, int N, int Hs, int Ws, int lds,
                                                               int K, int Hd, int Wd) {
  const long total = (long)N * K * Hd * Wd;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int x = (int)(i % Wd);
    long q = i / Wd;
    int y = (int)(q % Hd);
    q /= Hd;
    int k = (int)(q % K), n = (int)(q / K);
    int y0, y1, x0, x1;
    float wy, wx;
    bil_coord(y, Hs, Hd, y0, y1, wy);
    bil_coord(x, Ws, Wd, x0, x1, wx);
    const float* b = src + (long)n * Hs * Ws * lds + k;
    float a00 = b[((long)y0 * Ws + x0) * lds], a01 = b[((long)y0 * Ws + x1) * lds];
    float a10 = b[((long)y1 * Ws + x0) * lds], a11 = b[((long)y1 * Ws + x1) * lds];
    dst[i] = (1.f - wy) * ((1.f - wx) * a00 + wx * a01) + wy * ((1.f - wx) * a10 + wx * a11);
  }
}

// adjoint of the above: gsrc[n,sy,sx,k] (+)= sum over destination pixels of weight * gdst[n,k,Y,X].
// One block per (n,k): separable, first along X into LDS [Hd][Ws], then along Y.
__global__ __launch_bounds__(256) void upsample_to_nchw_bwd_kernel(const float* gdst, float* gsrc, int N, int Hs, int Ws, int lds,
                                                                   int K, int Hd, int Wd, int accumulate) {
  extern __shared__ float t[];   // [Hd][Ws]
  const int nk = blockIdx.x, n = nk / K, k = nk % K;
  const float* g = gdst + (long)nk * Hd * Wd;
  const float sx = (Wd > 1) ? (float)(Ws - 1) / (float)(Wd - 1) : 0.f;
  const float sy = (Hd > 1) ? (float)(Hs - 1) / (float)(Hd - 1) : 0.f;
  // stage 1: gather along X: t[y][sxi] = sum over the destination columns whose footprint names source column sxi (weight 1-f when
  // floor(src) == sxi, f when it is sxi-1).  One owner per element, ascending x: same bits every run (it was an LDS-atomic scatter).
  for (int o = threadIdx.x; o < Hd * Ws; o += 256) {
    const int sxi = o % Ws, y = o / Ws;
    int xlo = 0, xhi = Wd - 1;
    if (sx > 0.f) {
      xlo = max(0, (int)floorf((float)(sxi - 1) / sx) - 1);
      xhi = min(Wd - 1, (int)ceilf((float)(sxi + 1) / sx) + 1);
    }
    float acc = 0.f;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1;
      float wx;
      bil_coord(x, Ws, Wd, x0, x1, wx);
      const float v = g[y * Wd + x];
      if (x0 == sxi) acc += (1.f - wx) * v;
      if (x1 == sxi && x1 != x0) acc += wx * v;
    }
    t[o] = acc;
  }
  __syncthreads();
  (void)sx;
  // stage 2: gather along Y: source row i receives from dst rows with floor(src) == i (w 1-f) or i-1 (w f)
  for (int o = threadIdx.x; o < Hs * Ws; o += 256) {
    int sxi = o % Ws, syi = o / Ws;
    float acc = 0.f;
    // only destination rows whose source coordinate y*sy lies in (syi - 1, syi + 1) can name row syi (one row of margin
    // each side for the rounding of bil_coord); the rest of the 448 rows contributed exact zeros
    int ylo = 0, yhi = Hd - 1;
    if (sy > 0.f) {
      ylo = max(0, (int)floorf((float)(syi - 1) / sy) - 1);
      yhi = min(Hd - 1, (int)ceilf((float)(syi + 1) / sy) + 1);
    }
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1;
      float wy;
      bil_coord(y, Hs, Hd, y0, y1, wy);
      if (y0 == syi) acc += (1.f - wy) * t[y * Ws + sxi];
      if (y1 == syi && y1 != y0) acc += wy * t[y * Ws + sxi];
    }
    float* d = gsrc + (((long)n * Hs + syi) * Ws + sxi) * lds + k;
    *d = accumulate ? (*d + acc) : acc;
  }
}

// rows [R, C]: y = x / (||x||_2 + eps); saves the norm.  One wave per row.
__global__ __launch_bounds__(256) void row_l2norm_kernel(const float* x, float* y, float* nrm, long R, int C, float eps) {
  const int lane = threadIdx.x & 63;
  for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
    const float* p = x + r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += p[c] * p[c];
    s = sqrtf(wave_sum(s));
    float inv = 1.f / (s + eps);
    for (int c = lane; c < C; c += 64) y[r * C + c] = p[c] * inv;
    if (lane == 0) nrm[r] = s;
  }
}

// backward of y = x/(n+eps): gx = gy/(n+eps) - (x/n) * (gy.x)/(n+eps)^2
__global__ __launch_bounds__(256) void row_l2norm_bwd_kernel(const float* x, const float* nrm, const float* gy, float* gx, long R,
                                                             int C, float eps) {
  const int lane = threadIdx.x & 63;
  for (long r = blockIdx.x * 4L + (threadIdx.x >> 6); r < R; r += (long)gridDim.x * 4) {
    const float* p = x + r * C;
    const float* g = gy + r * C;
    float d = 0.f;
    for (int c = lane; c < C; c += 64) d += p[c] * g[c];
    d = wave_sum(d);
    float n = nrm[r], inv = 1.f / (n + eps);
    float k = (n > 0.f) ? d * inv * inv / n : 0.f;
    for (int c = lane; c < C; c += 64) gx[r * C + c] = g[c] * inv - p[c] * k;
  }
}

// PCM tail.  T[b,j,0..L) = aff * camx with camx[:, :, K] == 1, so T[b,j,K] is the affinity row sum.
//   fwd : rv[b,j,k] = T[b,j,k] / (T[b,j,K] + eps)   (k < K; padding columns -> 0)
//   bwd : gT[b,j,k] = grv[b,j,k]*r ; gT[b,j,K] = -sum_k grv*T*r^2 ; r = 1/(T[b,j,K]+eps)
__global__ __launch_bounds__(256) void pcm_norm_kernel(const float* T, const float* grv, float* out, long rows, int L, int K,
                                                       float eps, int bwd) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += (long)gridDim.x * 256) {
    const float* t = T + r * L;
    float inv = 1.f / (t[K] + eps);
    if (!bwd) {
      for (int k = 0; k < L; ++k) out[r * L + k] = (k < K) ? t[k] * inv : 0.f;
    } else {
      const float* g = grv + r * L;
      float acc = 0.f;
      for (int k = 0; k < K; ++k) { out[r * L + k] = g[k] * inv; acc += g[k] * t[k]; }
      out[r * L + K] = -acc * inv * inv;
      for (int k = K + 1; k < L; ++k) out[r * L + k] = 0.f;
    }
  }
}

// G2[b,i,j] = (gaff[b,i,j] + gaff[b,j,i]) * (aff[b,i,j] > 0): gradient of relu(f f^T) w.r.t. the symmetric product
__global__ __launch_bounds__(256) void sym_relu_grad_kernel(const float* gaff, const float* aff, float* out, int B, int n, int ld) {
  const long total = (long)B * n * ld;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int j = (int)(i % ld);
    long q = i / ld;
    int ii = (int)(q % n);
    long b = q / n;
    float v = 0.f;
    if (j < n && aff[i] > 0.f) v = gaff[i] + gaff[(b * n + j) * ld + ii];
    out[i] = v;
  }
}

// small elementwise helpers on flat fp32 arrays
//  op 0: out = alpha*a                 op 1: out = a + alpha*b
//  op 2: out = (y > 0) ? a (+ b) : 0   (relu backward; b optional, y given as third operand)
__global__ __launch_bounds__(256) void ew_kernel(int op, const float* a, const float* b, const float* y, float alpha, float* out,
                                                 long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v;
    if (op == 0) v = alpha * a[i];
    else if (op == 1) v = a[i] + alpha * b[i];
    else v = (y[i] > 0.f) ? (a[i] + (b ? b[i] : 0.f)) : 0.f;
    out[i] = v;
  }
}

// X[n, hw, c] += alpha * v[n, c]   (backward of the global average pool)
__global__ __launch_bounds__(256) void bcast_add_kernel(float* X, const float* v, float alpha, long rows, int C, int rps) {
  const int c4n = C / 4;
  const long total = rows * c4n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long r = i / c4n;
    int c = (int)(i - r * c4n) * 4;
    float4 x = ld4(X + i * 4), a = ld4(v + (r / rps) * C + c);
    x.x += alpha * a.x; x.y += alpha * a.y; x.z += alpha * a.z; x.w += alpha * a.w;
    st4(X + i * 4, x);
  }
}


// ---------------------------------------------------------------------------
// infer_mcl.py:124-148 for one forward pass of the multi-scale / flip list, fused: the model's own
// F.interpolate(align_corners=True) from the 1/16 map to the pass's input size (Hs x Ws), the script's cv2.resize
// (bilinear, half-pixel centres, edge clamp) from there to the original image size (H x W), the un-flip of the odd
// passes and the running sum over passes, for the K-1 foreground channels.  src: one sample, NHWC [h,w,lds], channel 0
// is the background.  acc[k-1,Y,X] += value.  Neither the [21,Hs,Ws] nor the [H,W,21] intermediate exists.
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bil_lr(const float* b, int h, int w, int lds, int Hs, int Ws, int yy, int xx) {
  int y0, y1, x0, x1;
  float wy, wx;
  bil_coord(yy, h, Hs, y0, y1, wy);
  bil_coord(xx, w, Ws, x0, x1, wx);
  float a00 = b[((long)y0 * w + x0) * lds], a01 = b[((long)y0 * w + x1) * lds];
  float a10 = b[((long)y1 * w + x0) * lds], a11 = b[((long)y1 * w + x1) * lds];
  return (1.f - wy) * ((1.f - wx) * a00 + wx * a01) + wy * ((1.f - wx) * a10 + wx * a11);
}

// half-pixel source coordinate (cv2.resize INTER_LINEAR / torch align_corners=False): src = (dst+0.5)*in/out - 0.5, >= 0
__device__ __forceinline__ void hp_coord(int d, int in, int out, int& i0, int& i1, float& w1) {
  float s = ((float)d + 0.5f) * ((float)in / (float)out) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - i0;
}

__global__ __launch_bounds__(256) void infer_accum_kernel(const float* src, float* acc, int h, int w, int lds, int K, int Hs,
                                                          int Ws, int H, int W, int flip) {
  const long total = (long)(K - 1) * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int X = (int)(i % W);
    long q = i / W;
    int Y = (int)(q % H), k = (int)(q / H) + 1;
    int Xr = flip ? (W - 1 - X) : X;            // np.flip(axis=1) after the resize
    int y0, y1, x0, x1;
    float wy, wx;
    hp_coord(Y, Hs, H, y0, y1, wy);
    hp_coord(Xr, Ws, W, x0, x1, wx);
    const float* b = src + k;
    float a00 = bil_lr(b, h, w, lds, Hs, Ws, y0, x0), a01 = bil_lr(b, h, w, lds, Hs, Ws, y0, x1);
    float a10 = bil_lr(b, h, w, lds, Hs, Ws, y1, x0), a11 = bil_lr(b, h, w, lds, Hs, Ws, y1, x1);
    acc[i] += (1.f - wy) * ((1.f - wx) * a00 + wx * a01) + wy * ((1.f - wx) * a10 + wx * a11);
  }
}

// infer_mcl.py:153-158 per channel (one workgroup each): clamp at 0, min / max over the image, zero what is below
// min + 1e-6, then (v - min - 1e-6) / (max - min + 1e-6).  In place.
__global__ __launch_bounds__(256) void infer_norm_kernel(float* acc, long HW) {
  __shared__ float smx[4], smn[4];
  float* a = acc + (long)blockIdx.x * HW;
  float mx = -3.4e38f, mn = 3.4e38f;
  for (long i = threadIdx.x; i < HW; i += 256) {
    float v = fmaxf(a[i], 0.f);
    mx = fmaxf(mx, v);
    mn = fminf(mn, v);
  }
  mx = wave_max(mx);
  mn = wave_min(mn);
  if ((threadIdx.x & 63) == 0) { smx[threadIdx.x >> 6] = mx; smn[threadIdx.x >> 6] = mn; }
  __syncthreads();
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  const float lo = mn + 1e-6f, den = mx - mn + 1e-6f;
  for (long i = threadIdx.x; i < HW; i += 256) {
    float v = fmaxf(a[i], 0.f);
    if (v < lo) v = 0.f;
    a[i] = (v - mn - 1e-6f) / den;
  }
}


// ---------------------------------------------------------------------------
// Per-epoch rapid evaluation (train_mcl.py:286-318 + src/evaluation.py:19-52): for every threshold t,
//   predict = argmax_k [t, half(pred_1*label_1), ..., half(pred_{K-1}*label_{K-1})]   (first maximum wins, as np.argmax)
// and, over pixels with gt < 255:  P[predict]++, T[gt]++, TP[gt] += (predict == gt).
// pred: one image, [K,H,W] fp32 (already cam_maxnorm'ed); label: [K] (entry 0 unused); gt: uint8 [H,W].
// counts: int64 [nt][K][3] = (TP, P, T), accumulated across images.  The values go through fp16 exactly as the script's
// np.half files do.  Per workgroup the counts are kept in LDS (nt*K*3 ints) and flushed once.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void eval_confusion_kernel(const float* pred, const float* label, const unsigned char* gt,
                                                             const float* thr, int nt, int K, long HW, long long* counts) {
  extern __shared__ int lc[];                   // [nt][K][3]
  for (int i = threadIdx.x; i < nt * K * 3; i += 256) lc[i] = 0;
  __syncthreads();
  for (long p = blockIdx.x * 256L + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
    const int g = gt[p];
    if (g >= 255) continue;
    float best = -1.f;                          // values are >= 0; strict > keeps the first maximum
    int bk = 0;
    for (int k = 1; k < K; ++k) {
      float v = __half2float(__float2half_rn(pred[(long)k * HW + p] * label[k]));
      if (v > best) { best = v; bk = k; }
    }
    for (int t = 0; t < nt; ++t) {
      const int pr = (thr[t] >= best) ? 0 : bk;  // channel 0 holds the threshold and precedes every other channel
      atomicAdd(&lc[(t * K + pr) * 3 + 1], 1);
      if (g < K) {
        atomicAdd(&lc[(t * K + g) * 3 + 2], 1);
        if (pr == g) atomicAdd(&lc[(t * K + g) * 3 + 0], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt * K * 3; i += 256)
    if (lc[i]) atomicAdd((unsigned long long*)&counts[i], (unsigned long long)lc[i]);
}

static int gs(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

extern "C" {

int mx_resize_nhwc(const float* src, float* dst, int N, int Hs, int Ws, int C, int Hd, int Wd, int ldd, int coff, int relu,
                   void* stream) {
  MX_CHECK_ARG(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C % 4 == 0 && ldd % 4 == 0 && coff % 4 == 0 &&
                   coff + C <= ldd, "resize_nhwc: bad args C=%d ldd=%d coff=%d", C, ldd, coff);
  hipLaunchKernelGGL(resize_nhwc_kernel, dim3(gs((long)N * Hd * Wd * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                     Hs, Ws, C, Hd, Wd, ldd, coff, relu);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_upsample_to_nchw(const float* src, float* dst, int N, int Hs, int Ws, int lds, int K, int Hd, int Wd, void* stream) {
  MX_CHECK_ARG(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && K > 0 && K <= lds, "upsample_to_nchw: bad args");
  hipLaunchKernelGGL(upsample_to_nchw_kernel, dim3(gs((long)N * K * Hd * Wd)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                     Hs, Ws, lds, K, Hd, Wd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_upsample_to_nchw_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int lds, int K, int Hd, int Wd,
                            int accumulate, void* stream) {
  MX_CHECK_ARG(gdst && gsrc && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && K > 0 && K <= lds, "upsample_to_nchw_bwd: bad args");
  size_t sh = (size_t)Hd * Ws * sizeof(float);
  MX_CHECK_ARG(sh <= 160 * 1024, "upsample_to_nchw_bwd: Hd*Ws=%d exceeds LDS", Hd * Ws);
  if (sh > 48 * 1024)
    hipFuncSetAttribute((const void*)upsample_to_nchw_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
  hipLaunchKernelGGL(upsample_to_nchw_bwd_kernel, dim3(N * K), dim3(256), sh, (hipStream_t)stream, gdst, gsrc, N, Hs, Ws, lds,
                     K, Hd, Wd, accumulate);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_row_l2norm(const float* x, float* y, float* nrm, long R, int C, float eps, void* stream) {
  MX_CHECK_ARG(x && y && nrm && R > 0 && C > 0, "row_l2norm: bad args");
  hipLaunchKernelGGL(row_l2norm_kernel, dim3(gs(R * 64)), dim3(256), 0, (hipStream_t)stream, x, y, nrm, R, C, eps);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_row_l2norm_bwd(const float* x, const float* nrm, const float* gy, float* gx, long R, int C, float eps, void* stream) {
  MX_CHECK_ARG(x && nrm && gy && gx && R > 0 && C > 0, "row_l2norm_bwd: bad args");
  hipLaunchKernelGGL(row_l2norm_bwd_kernel, dim3(gs(R * 64)), dim3(256), 0, (hipStream_t)stream, x, nrm, gy, gx, R, C, eps);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_pcm_norm(const float* T, const float* grv, float* out, long rows, int L, int K, float eps, int bwd, void* stream) {
  MX_CHECK_ARG(T && out && rows > 0 && K > 0 && K < L && (!bwd || grv), "pcm_norm: bad args");
  hipLaunchKernelGGL(pcm_norm_kernel, dim3(gs(rows)), dim3(256), 0, (hipStream_t)stream, T, grv, out, rows, L, K, eps, bwd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_sym_relu_grad(const float* gaff, const float* aff, float* out, int B, int n, int ld, void* stream) {
  MX_CHECK_ARG(gaff && aff && out && B > 0 && n > 0 && ld >= n, "sym_relu_grad: bad args");
  hipLaunchKernelGGL(sym_relu_grad_kernel, dim3(gs((long)B * n * ld)), dim3(256), 0, (hipStream_t)stream, gaff, aff, out, B, n, ld);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_ew(int op, const float* a, const float* b, const float* y, float alpha, float* out, long n, void* stream) {
  MX_CHECK_ARG(a && out && n > 0 && op >= 0 && op <= 2, "ew: bad args");
  MX_CHECK_ARG(op != 1 || b, "ew: op 1 needs b");
  MX_CHECK_ARG(op != 2 || y, "ew: op 2 needs y");
  hipLaunchKernelGGL(ew_kernel, dim3(gs(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, y, alpha, out, n);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_bcast_add(float* X, const float* v, float alpha, long rows, int C, int rows_per_sample, void* stream) {
  MX_CHECK_ARG(X && v && rows > 0 && C % 4 == 0 && rows_per_sample > 0, "bcast_add: bad args");
  hipLaunchKernelGGL(bcast_add_kernel, dim3(gs(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, X, v, alpha, rows, C,
                     rows_per_sample);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_infer_accum(const float* src, float* acc, int h, int w, int lds, int K, int Hs, int Ws, int H, int W, int flip, void* stream) {
  MX_CHECK_ARG(src && acc && h > 0 && w > 0 && K > 1 && K <= lds && Hs > 0 && Ws > 0 && H > 0 && W > 0, "infer_accum: bad args");
  hipLaunchKernelGGL(infer_accum_kernel, dim3(gs((long)(K - 1) * H * W)), dim3(256), 0, (hipStream_t)stream, src, acc, h, w, lds, K,
                     Hs, Ws, H, W, flip);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_infer_norm(float* acc, int channels, long HW, void* stream) {
  MX_CHECK_ARG(acc && channels > 0 && HW > 0, "infer_norm: bad args");
  hipLaunchKernelGGL(infer_norm_kernel, dim3(channels), dim3(256), 0, (hipStream_t)stream, acc, HW);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_eval_confusion(const float* pred, const float* label, const unsigned char* gt, const float* thresholds, int nt, int K, int H,
                      int W, long long* counts, void* stream) {
  MX_CHECK_ARG(pred && label && gt && thresholds && counts && nt > 0 && nt <= 64 && K >= 2 && K <= 256 && H > 0 && W > 0,
               "eval_confusion: bad args");
  const long HW = (long)H * W;
  int blocks = (int)((HW + 256 * 8 - 1) / (256 * 8));
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(eval_confusion_kernel, dim3(blocks), dim3(256), sizeof(int) * nt * K * 3, (hipStream_t)stream, pred, label, gt,
                     thresholds, nt, K, HW, counts);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
