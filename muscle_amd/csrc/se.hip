// Squeeze-and-Excitation excitation path (the two 1x1 convs on the pooled [N,C] vector) and
// the stem's im2col.
//
// Reference: MBConvBlock.forward, src/efficientnet_pytorch/model.py:81-84:
//   s = avgpool(x); h = se_reduce(s) (+bias); r = swish(h); e = se_expand(r) (+bias); x = sigmoid(e) * x
// The pooling itself is mx_pool_sum (bn.hip); here the pooled sums arrive as [N,C] and are scaled by
// 1/HW.  One workgroup per sample: the matrices are tiny (C <= 3840, squeeze <= 160), the job is
// latency-bound, so wave-shuffle dot products straight from L2 are enough.
#include "common.h"

constexpr int SQ_MAX = 256;

// gate[n,c] = sigmoid(W2[c,:] . swish(W1 s + b1) + b2[c]);  saves s (mean) and h (pre-activation)
__global__ __launch_bounds__(256) void se_fwd_kernel(const float* pooled, float inv_hw, const float* W1, const float* b1,
                                                     const float* W2, const float* b2, float* s_out, float* h_out,
                                                     float* gate, int C, int SQ) {
  __shared__ float r[SQ_MAX];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* ps = pooled + (long)n * C;
  for (int c = tid; c < C; c += 256) s_out[(long)n * C + c] = ps[c] * inv_hw;
  for (int j = wave; j < SQ; j += 4) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc += W1[(long)j * C + c] * (ps[c] * inv_hw);
    acc = wave_sum(acc);
    if (lane == 0) {
      float h = acc + b1[j];
      h_out[(long)n * SQ + j] = h;
      r[j] = swishf_(h);
    }
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float acc = b2[c];
    for (int j = 0; j < SQ; ++j) acc += W2[(long)c * SQ + j] * r[j];
    gate[(long)n * C + c] = sigmoidf_(acc);
  }
}

// backward of the excitation path for one sample.
//   in : ggate[n,c] = sum_hw dA[n,hw,c] * act[n,hw,c]   (gradient w.r.t. the gate)
//   out: add[n,c]   = (dL/ds)[n,c] / HW                 (pooled-path gradient, broadcast over hw)
//        dW1,db1,dW2,db2 += ...
__global__ __launch_bounds__(256) void se_bwd_kernel(const float* ggate, const float* gate, const float* s, const float* h,
                                                     const float* W1, const float* W2, float inv_hw, float* add,
                                                     float* dW1, float* db1, float* dW2, float* db2, int C, int SQ) {
  __shared__ float r[SQ_MAX], gh[SQ_MAX];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* gg = ggate + (long)n * C;
  const float* gt = gate + (long)n * C;
  const float* sn = s + (long)n * C;
  for (int j = tid; j < SQ; j += 256) r[j] = swishf_(h[(long)n * SQ + j]);
  __syncthreads();
  // g_e[c] = ggate*gate*(1-gate); dW2[c,j] += g_e[c]*r[j]; db2[c] += g_e[c]
  for (int c = tid; c < C; c += 256) {
    float g = gt[c];
    float ge = gg[c] * g * (1.f - g);
    unsafeAtomicAdd(db2 + c, ge);
    for (int j = 0; j < SQ; ++j) unsafeAtomicAdd(dW2 + (long)c * SQ + j, ge * r[j]);
  }
  // g_r[j] = sum_c g_e[c] * W2[c,j];  g_h = g_r * swish'(h)
  for (int j = wave; j < SQ; j += 4) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) {
      float g = gt[c];
      acc += gg[c] * g * (1.f - g) * W2[(long)c * SQ + j];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      float v = acc * swish_gradf_(h[(long)n * SQ + j]);
      gh[j] = v;
      unsafeAtomicAdd(db1 + j, v);
    }
  }
  __syncthreads();
  // dW1[j,c] += g_h[j]*s[c];  g_s[c] = sum_j g_h[j]*W1[j,c]
  for (int c = tid; c < C; c += 256) {
    float sc = sn[c], acc = 0.f;
    for (int j = 0; j < SQ; ++j) {
      float v = gh[j];
      acc += v * W1[(long)j * C + c];
      unsafeAtomicAdd(dW1 + (long)j * C + c, v * sc);
    }
    add[(long)n * C + c] = acc * inv_hw;
  }
}

// stem patches: out[(n,oy,ox), t] = img[n, ci, oy*2-pad+ky, ox*2-pad+kx], t = ci*9+ky*3+kx, row length 28
// (27 taps + one zero so rows stay 16-byte aligned).  The stem convolution (model.py:131,175; 3x3, stride 2,
// static same padding) then runs on the MFMA GEMM as [R,28] x [C0,28]^T.
__global__ __launch_bounds__(256) void stem_im2col_kernel(const float* img, float* out, int N, int H, int W, int Ho, int Wo,
                                                          int pad) {
  const long total = (long)N * Ho * Wo;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    int ox = (int)(p % Wo);
    long t = p / Wo;
    int oy = (int)(t % Ho), n = (int)(t / Ho);
    float v[28];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          int iy = oy * 2 - pad + ky, ix = ox * 2 - pad + kx;
          v[ci * 9 + ky * 3 + kx] =
              (iy >= 0 && iy < H && ix >= 0 && ix < W) ? img[(((long)n * 3 + ci) * H + iy) * W + ix] : 0.f;
        }
    v[27] = 0.f;
    float* o = out + p * 28;
#pragma unroll
    for (int j = 0; j < 7; ++j) st4(o + 4 * j, make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]));
  }
}

extern "C" {

int mx_se_fwd(const float* pooled_sum, float inv_hw, const float* W1, const float* b1, const float* W2, const float* b2,
              float* s, float* h, float* gate, int N, int C, int SQ, void* stream) {
  MX_CHECK_ARG(pooled_sum && W1 && b1 && W2 && b2 && s && h && gate, "se_fwd: null pointer");
  MX_CHECK_ARG(N > 0 && C > 0 && SQ > 0 && SQ <= SQ_MAX, "se_fwd: bad extents N=%d C=%d SQ=%d", N, C, SQ);
  hipLaunchKernelGGL(se_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, pooled_sum, inv_hw, W1, b1, W2, b2, s, h,
                     gate, C, SQ);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_se_bwd(const float* ggate, const float* gate, const float* s, const float* h, const float* W1, const float* W2,
              float inv_hw, float* add, float* dW1, float* db1, float* dW2, float* db2, int N, int C, int SQ,
              void* stream) {
  MX_CHECK_ARG(ggate && gate && s && h && W1 && W2 && add && dW1 && db1 && dW2 && db2, "se_bwd: null pointer");
  MX_CHECK_ARG(N > 0 && C > 0 && SQ > 0 && SQ <= SQ_MAX, "se_bwd: bad extents N=%d C=%d SQ=%d", N, C, SQ);
  hipLaunchKernelGGL(se_bwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, ggate, gate, s, h, W1, W2, inv_hw, add,
                     dW1, db1, dW2, db2, C, SQ);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_stem_im2col(const float* img, float* out, int N, int H, int W, int Ho, int Wo, int pad_lo, void* stream) {
  MX_CHECK_ARG(img && out && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && pad_lo >= 0 && pad_lo < 3, "stem_im2col: bad args");
  long total = (long)N * Ho * Wo;
  long b = (total + 255) / 256;
  hipLaunchKernelGGL(stem_im2col_kernel, dim3((int)(b < 8192 ? b : 8192)), dim3(256), 0, (hipStream_t)stream, img, out, N,
                     H, W, Ho, Wo, pad_lo);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
