// Squeeze-and-Excitation excitation path (the two 1x1 convs on the pooled [N,C] vector) and
// the stem's im2col.
//
// Reference: MBConvBlock.forward, src/efficientnet_pytorch/model.py:81-84:
//   s = avgpool(x); h = se_reduce(s) (+bias); r = swish(h); e = se_expand(r) (+bias); x = sigmoid(e) * x
// The pooling itself is mx_pool_sum (bn.hip); here the pooled sums arrive as [N,C] and are scaled by
// 1/HW.  The matrices are tiny (C <= 3840, squeeze <= 160) and the job is latency-bound, so the work is cut
// into (sample, output) pieces that fill the chip, with wave-shuffle dot products straight from L2; weight
// gradients are owned per channel (no atomics).
#include "common.h"

constexpr int SQ_MAX = 256;

// ---- forward: two launches so the grid has >> 256 workgroups even for N = 32 ------------------------
// h[n,j] = W1[j,:] . s[n,:] + b1[j]   (one wave per (n,j); s = pooled_sum * inv_hw, also written out)
__global__ __launch_bounds__(256) void se_fwd_reduce_kernel(const float* pooled, float inv_hw, const float* W1, const float* b1,
                                                            float* s_out, float* h_out, int C, int SQ) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.y * 4 + wave;
  const float* ps = pooled + (long)n * C;
  if (blockIdx.y == 0)
    for (int c = threadIdx.x; c < C; c += 256) s_out[(long)n * C + c] = ps[c] * inv_hw;
  if (j >= SQ) return;
  float acc = 0.f;
  const float* w = W1 + (long)j * C;
  if ((C & 3) == 0 && (((uintptr_t)W1 | (uintptr_t)pooled) & 15) == 0) {
    // 16-byte loads, two independent chains (C = 2304: 9 wave-wide load pairs instead of 36; 18.8 -> ~10 us on the 28 x 28 stages)
    float4 a0 = make_float4(0, 0, 0, 0), a1 = a0;
    int c = 4 * lane;
    for (; c + 256 < C; c += 512) {
      const float4 w0 = ld4(w + c), p0 = ld4(ps + c), w1 = ld4(w + c + 256), p1 = ld4(ps + c + 256);
      a0.x += w0.x * p0.x; a0.y += w0.y * p0.y; a0.z += w0.z * p0.z; a0.w += w0.w * p0.w;
      a1.x += w1.x * p1.x; a1.y += w1.y * p1.y; a1.z += w1.z * p1.z; a1.w += w1.w * p1.w;
    }
    if (c < C) {
      const float4 w0 = ld4(w + c), p0 = ld4(ps + c);
      a0.x += w0.x * p0.x; a0.y += w0.y * p0.y; a0.z += w0.z * p0.z; a0.w += w0.w * p0.w;
    }
    acc = ((a0.x + a1.x) + (a0.y + a1.y)) + ((a0.z + a1.z) + (a0.w + a1.w));
  } else {
    for (int c = lane; c < C; c += 64) acc += w[c] * ps[c];
  }
  acc = wave_sum(acc) * inv_hw;
  if (lane == 0) h_out[(long)n * SQ + j] = acc + b1[j];
}

// gate[n,c] = sigmoid(W2[c,:] . swish(h[n,:]) + b2[c])
__global__ __launch_bounds__(256) void se_fwd_expand_kernel(const float* h, const float* W2, const float* b2, float* gate, int C,
                                                            int SQ) {
  __shared__ float r[SQ_MAX];
  const int n = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  for (int j = threadIdx.x; j < SQ; j += 256) r[j] = swishf_(h[(long)n * SQ + j]);
  __syncthreads();
  if (c >= C) return;
  float acc = b2[c];
  const float* w = W2 + (long)c * SQ;
  if ((SQ & 3) == 0 && ((uintptr_t)W2 & 15) == 0) {          // a row of W2 in 16-byte pieces (every B7 width), two chains
    float a1 = 0.f;
    int j = 0;
    for (; j + 4 < SQ; j += 8) {
      const float4 w0 = ld4(w + j), w1 = ld4(w + j + 4);
      acc += (w0.x * r[j] + w0.y * r[j + 1]) + (w0.z * r[j + 2] + w0.w * r[j + 3]);
      a1 += (w1.x * r[j + 4] + w1.y * r[j + 5]) + (w1.z * r[j + 6] + w1.w * r[j + 7]);
    }
    if (j < SQ) { const float4 w0 = ld4(w + j); acc += (w0.x * r[j] + w0.y * r[j + 1]) + (w0.z * r[j + 2] + w0.w * r[j + 3]); }
    acc += a1;
  } else {
    for (int j = 0; j < SQ; ++j) acc += w[j] * r[j];
  }
  gate[(long)n * C + c] = sigmoidf_(acc);
}

// ---- backward --------------------------------------------------------------------------------------
// (A) g_r[n,j] = sum_c ge[n,c] W2[c,j] with ge = ggate*gate*(1-gate);  gh[n,j] = g_r * swish'(h)   (db1 leaves kernel B).
//     One workgroup of 1024 threads per (n, sixteen consecutive j): W2 is [C,SQ] row-major, a thread's load is one 16-byte piece of
//     row c and four neighbouring lanes read 64 consecutive bytes of it; 256 channel lanes share the C rows (9 independent
//     loads per thread at C = 2304 - round 4's form, one wave per four j, walked all 36 in one lane: 14 us of load latency on the
//     backward chain of every block).  Sums: lanes of a wave by xor-shuffle, the 16 waves in ascending order - the same bits every run.
//     SQ is a multiple of 4 for every EfficientNet-B7 width.
__global__ __launch_bounds__(1024) void se_bwd_a_kernel(const float* ggate, const float* gate, const float* h, const float* W2,
                                                        float* gh, int C, int SQ) {
  __shared__ float red[16][16];
  const int n = blockIdx.x, quad = threadIdx.x & 3, cl = threadIdx.x >> 2, wave = threadIdx.x >> 6;
  const int j = (blockIdx.y * 4 + quad) * 4;
  const float* gg = ggate + (long)n * C;
  const float* gt = gate + (long)n * C;
  float4 acc = make_float4(0, 0, 0, 0);
  if (j < SQ) {
#pragma unroll 4
    for (int c = cl; c < C; c += 256) {
      const float g = gt[c];
      const float ge = gg[c] * g * (1.f - g);
      const float4 w = ld4(W2 + (long)c * SQ + j);
      acc.x += ge * w.x; acc.y += ge * w.y; acc.z += ge * w.z; acc.w += ge * w.w;
    }
  }
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) {
    acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off); acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
  }
  if ((threadIdx.x & 63) < 4) {
    red[wave][quad * 4 + 0] = acc.x; red[wave][quad * 4 + 1] = acc.y; red[wave][quad * 4 + 2] = acc.z; red[wave][quad * 4 + 3] = acc.w;
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const int jj = blockIdx.y * 16 + threadIdx.x;
    if (jj < SQ) {
      float a = red[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < 16; ++w) a += red[w][threadIdx.x];
      gh[(long)n * SQ + jj] = a * swish_gradf_(h[(long)n * SQ + jj]);
    }
  }
}

// the same with one squeeze unit per wave, for widths whose SQ is not a multiple of 4 (B0 ... B6)
__global__ __launch_bounds__(256) void se_bwd_a1_kernel(const float* ggate, const float* gate, const float* h, const float* W2,
                                                       float* gh, int C, int SQ) {
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.y * 4 + wave;
  if (j >= SQ) return;
  const float* gg = ggate + (long)n * C;
  const float* gt = gate + (long)n * C;
  float acc = 0.f;
  for (int c = lane; c < C; c += 64) {
    float g = gt[c];
    acc += gg[c] * g * (1.f - g) * W2[(long)c * SQ + j];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    float v = acc * swish_gradf_(h[(long)n * SQ + j]);
    gh[(long)n * SQ + j] = v;
  }
}

// (B) thread = channel c, block row = chunk of SE_JB squeeze units: dW2[c,j] += sum_n ge[n,c] r[n,j];
//     dW1[j,c] += sum_n gh[n,j] s[n,c]; db2 by chunk 0; db1[j] += sum_n gh[n,j] by the first threads of channel block 0.
//     Everything is owned by one thread and summed over n in ascending order: no atomics, same bits every run.
//     (The pooled-path gradient add[n,c] = inv_hw * sum_j gh[n,j] W1[j,c] used to leave here through fp32 atomics across
//     the squeeze chunks; it is now formed, per (n, c) in one thread, by mx_bn1_sums_finalize, its first consumer.)
constexpr int SE_NB = 32, SE_JB = 4;
__global__ __launch_bounds__(256) void se_bwd_b_kernel(const float* ggate, const float* gate, const float* s, const float* h,
                                                       const float* gh, float* dW1, float* db1,
                                                       float* dW2, float* db2, int N, int C, int SQ) {
  extern __shared__ float sh[];          // r[N][SE_JB], ghs[N][SE_JB]
  float* r = sh;
  float* ghs = sh + (long)N * SE_JB;
  const int j0 = blockIdx.y * SE_JB, jn = min(SE_JB, SQ - j0);
  for (int i = threadIdx.x; i < N * SE_JB; i += 256) {
    int n = i / SE_JB, jj = i % SE_JB;
    bool ok = jj < jn;
    r[i] = ok ? swishf_(h[(long)n * SQ + j0 + jj]) : 0.f;
    ghs[i] = ok ? gh[(long)n * SQ + j0 + jj] : 0.f;
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < jn) {
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += ghs[n * SE_JB + threadIdx.x];
    db1[j0 + threadIdx.x] += a;
  }
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  for (int n0 = 0; n0 < N; n0 += SE_NB) {
    float ge[SE_NB], sv[SE_NB];
    float sb = 0.f;
#pragma unroll
    for (int n = 0; n < SE_NB; ++n) {
      ge[n] = sv[n] = 0.f;
      if (n0 + n < N) {
        float g = gate[(long)(n0 + n) * C + c];
        ge[n] = ggate[(long)(n0 + n) * C + c] * g * (1.f - g);
        sv[n] = s[(long)(n0 + n) * C + c];
        sb += ge[n];
      }
    }
    if (blockIdx.y == 0) db2[c] += sb;
    if (jn == SE_JB && (SQ & 3) == 0 && ((uintptr_t)dW2 & 15) == 0) {
      // a full chunk: the four dW2[c, j0..j0+3] of this channel are ONE 16-byte read-modify-write (they were four 4-byte ones, 384 bytes
      // apart between neighbouring lanes), and the gradient rows are requested before the sums are formed
      float4* p2 = reinterpret_cast<float4*>(dW2 + (long)c * SQ + j0);
      float4 o2 = *p2;
      float o1[SE_JB];
#pragma unroll
      for (int jj = 0; jj < SE_JB; ++jj) o1[jj] = dW1[(long)(j0 + jj) * C + c];
      float a2[SE_JB], a1[SE_JB];
#pragma unroll
      for (int jj = 0; jj < SE_JB; ++jj) {
        a2[jj] = 0.f; a1[jj] = 0.f;
#pragma unroll
        for (int n = 0; n < SE_NB; ++n) {
          const int nn = (n0 + n < N) ? n0 + n : 0;     // ge/sv are 0 beyond N
          a2[jj] += ge[n] * r[nn * SE_JB + jj];
          a1[jj] += ghs[nn * SE_JB + jj] * sv[n];
        }
      }
      o2.x += a2[0]; o2.y += a2[1]; o2.z += a2[2]; o2.w += a2[3];
      *p2 = o2;
#pragma unroll
      for (int jj = 0; jj < SE_JB; ++jj) dW1[(long)(j0 + jj) * C + c] = o1[jj] + a1[jj];
      continue;
    }
    for (int jj = 0; jj < jn; ++jj) {
      float a2 = 0.f, a1 = 0.f;
#pragma unroll
      for (int n = 0; n < SE_NB; ++n) {
        const int nn = (n0 + n < N) ? n0 + n : 0;     // ge/sv are 0 beyond N
        a2 += ge[n] * r[nn * SE_JB + jj];
        a1 += ghs[nn * SE_JB + jj] * sv[n];
      }
      dW2[(long)c * SQ + j0 + jj] += a2;
      dW1[(long)(j0 + jj) * C + c] += a1;
    }
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
  hipLaunchKernelGGL(se_fwd_reduce_kernel, dim3(N, cdiv(SQ, 4)), dim3(256), 0, (hipStream_t)stream, pooled_sum, inv_hw, W1, b1,
                     s, h, C, SQ);
  hipLaunchKernelGGL(se_fwd_expand_kernel, dim3(N, cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, h, W2, b2, gate, C, SQ);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_se_bwd_gh(const float* ggate, const float* gate, const float* h, const float* W2, float* gh, int N, int C, int SQ,
                 void* stream) {
  MX_CHECK_ARG(ggate && gate && h && W2 && gh, "se_bwd_gh: null pointer");
  MX_CHECK_ARG(N > 0 && C > 0 && SQ > 0 && SQ <= SQ_MAX, "se_bwd_gh: bad extents N=%d C=%d SQ=%d", N, C, SQ);
  if (SQ % 4 == 0 && ((uintptr_t)W2 & 15) == 0)
    hipLaunchKernelGGL(se_bwd_a_kernel, dim3(N, cdiv(SQ, 16)), dim3(1024), 0, (hipStream_t)stream, ggate, gate, h, W2, gh, C, SQ);
  else
    hipLaunchKernelGGL(se_bwd_a1_kernel, dim3(N, cdiv(SQ, 4)), dim3(256), 0, (hipStream_t)stream, ggate, gate, h, W2, gh, C, SQ);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_se_bwd_params(const float* ggate, const float* gate, const float* s, const float* h, const float* gh, float* dW1, float* db1,
                     float* dW2, float* db2, int N, int C, int SQ, void* stream) {
  MX_CHECK_ARG(ggate && gate && s && h && gh && dW1 && db1 && dW2 && db2, "se_bwd_params: null pointer");
  MX_CHECK_ARG(N > 0 && C > 0 && SQ > 0 && SQ <= SQ_MAX, "se_bwd_params: bad extents N=%d C=%d SQ=%d", N, C, SQ);
  size_t shb = (size_t)2 * N * SE_JB * sizeof(float);
  MX_CHECK_ARG(shb <= 48 * 1024, "se_bwd_params: N=%d too large for LDS staging", N);
  hipLaunchKernelGGL(se_bwd_b_kernel, dim3(cdiv(C, 256), cdiv(SQ, SE_JB)), dim3(256), shb, (hipStream_t)stream, ggate, gate, s, h, gh,
                     dW1, db1, dW2, db2, N, C, SQ);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_se_bwd(const float* ggate, const float* gate, const float* s, const float* h, const float* W2,
              float* dW1, float* db1, float* dW2, float* db2, float* gh, int N, int C,
              int SQ, void* stream) {
  MX_CHECK_ARG(dW1 && db1 && dW2 && db2 && s, "se_bwd: null pointer");
  MX_CHECK_ARG(N > 0 && (size_t)2 * N * SE_JB * sizeof(float) <= 48 * 1024, "se_bwd: N=%d too large for LDS staging", N);
  int rc = mx_se_bwd_gh(ggate, gate, h, W2, gh, N, C, SQ, stream);
  if (rc != MX_OK) return rc;
  return mx_se_bwd_params(ggate, gate, s, h, gh, dW1, db1, dW2, db2, N, C, SQ, stream);
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
