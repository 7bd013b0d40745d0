// Loss kernels of the MCL step (phase 1): classification losses, image-level contrast (IMC),
// cam_softmaxnorm and the equivariant-regularisation (ER) top-k loss; plus the fused Adam update.
//
// Reference: src/loss_multilabel.py:24-33 (pairwise), :36-66 (IMC), :68-91 (focal),
// nn.MultiLabelSoftMarginLoss (train_mcl.py:146,181), train_mcl.py:30-36 (cam_softmaxnorm),
// train_mcl.py:178,185-188 (ER expression), torch.optim.Adam with weight_decay (train_mcl.py:134).
#include "common.h"

// ---------------------------------------------------------------------------
// small [N, C] classification losses.  mode 0 focal(p,y); 1 soft-margin(x,y); 2 pairwise(p,y);
// 3 sigmoid forward (out = sigmoid(x)); 4 sigmoid backward (out = g * s * (1-s), x = s, y = g)
// loss: focal / softmargin -> scalar loss[0] (overwritten); pairwise -> loss[n]
// grad: d loss / d input for a unit upstream gradient (pairwise: per-sample loss_n)
// ---------------------------------------------------------------------------
// modes 0 / 1 (scalar losses): ONE workgroup, a wave per sample (strided), per-sample terms parked in LDS and added in
// sample order by one thread - the loss has the same bits every run (the per-sample workgroups used to meet in an atomic)
constexpr int CLS_MAXN = 1024;
__global__ __launch_bounds__(1024) void cls_scalar_loss_kernel(int mode, const float* x, int ldx, const float* y, int ldy, float* loss,
                                                               float* grad, int ldg, int N, int C) {
  __shared__ float ln[CLS_MAXN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  for (int n = wave; n < N; n += NW) {
    const float* xr = x + (long)n * ldx;
    const float* yr = y + (long)n * ldy;
    float acc = 0.f;
    if (mode == 0) {
      for (int c = lane; c < C; c += 64) {
        float p = xr[c], t = yr[c];
        float pt = t * p + (1.f - t) * (1.f - p);
        float om = 1.f - pt, lg = logf(pt + 1e-9f);
        acc += -0.5f * om * om * lg;
        float dpt = -0.5f * (-2.f * om * lg + om * om / (pt + 1e-9f));
        grad[(long)n * ldg + c] = dpt * (2.f * t - 1.f) / N;
      }
      acc = wave_sum(acc) / N;
    } else {
      for (int c = lane; c < C; c += 64) {
        float v = xr[c], t = yr[c];
        // log sigmoid(v) = min(v,0) - log1p(exp(-|v|))
        float ls = fminf(v, 0.f) - log1pf(__expf(-fabsf(v)));
        float lsn = fminf(-v, 0.f) - log1pf(__expf(-fabsf(v)));
        acc += -(t * ls + (1.f - t) * lsn);
        grad[(long)n * ldg + c] = (sigmoidf_(v) - t) / ((float)C * N);
      }
      acc = wave_sum(acc) / ((float)C * N);
    }
    if (lane == 0) ln[n] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int n = 0; n < N; ++n) t += ln[n];
    loss[0] = t;
  }
}

__global__ __launch_bounds__(64) void cls_loss_kernel(int mode, const float* x, int ldx, const float* y, int ldy, float* loss,
                                                      float* grad, int ldg, int N, int C) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (long)n * ldx;
  const float* yr = y + (long)n * ldy;
  if (mode == 3) { for (int c = lane; c < C; c += 64) grad[(long)n * ldg + c] = sigmoidf_(xr[c]); return; }
  if (mode == 4) { for (int c = lane; c < C; c += 64) { float s = xr[c]; grad[(long)n * ldg + c] = yr[c] * s * (1.f - s); } return; }
  {
    float sn = 0.f, sp = 0.f;
    for (int c = lane; c < C; c += 64) {
      float p = xr[c], t = yr[c];
      float pos = (t == 0.f) ? 0.f : p, neg = (t == 1.f) ? 0.f : p;
      sn += __expf(neg);
      sp += __expf(-pos);
    }
    sn = wave_sum(sn); sp = wave_sum(sp);
    float cc = (float)C * C;
    float S = sn * sp / cc;
    if (lane == 0) loss[n] = logf(1.f + S);
    float k = 1.f / (1.f + S) / cc;
    for (int c = lane; c < C; c += 64) {
      float p = xr[c], t = yr[c];
      float g = 0.f;
      if (t != 1.f) g += __expf(p) * sp;      // neg_c = p active
      if (t != 0.f) g -= __expf(-p) * sn;     // pos_c = p active
      grad[(long)n * ldg + c] = g * k;
    }
  }
}

// ---------------------------------------------------------------------------
// IMC (loss_multilabel.py:36-66) in one workgroup: N <= 64 samples.
// out[0] = loss, out[1] = number of valid anchor rows; gemb = d loss / d emb.
// ws: N*D floats (normalised embeddings) + N floats (norms)
// ---------------------------------------------------------------------------
constexpr int IMC_MAXN = 64;
__global__ __launch_bounds__(1024) void imc_kernel(const float* emb, const float* label, int N, int D, int L, float* out,
                                                  float* gemb, float* ws) {
  __shared__ float Wm[IMC_MAXN][IMC_MAXN + 1];   // first S, then the symmetric pair weights
  __shared__ unsigned char Pm[IMC_MAXN][IMC_MAXN], Gm[IMC_MAXN][IMC_MAXN];
  __shared__ float nrm[IMC_MAXN], rowk1[IMC_MAXN], rowk2[IMC_MAXN];
  __shared__ float rowloss[IMC_MAXN];            // per anchor row, added in row order by one thread (was an LDS atomic)
  __shared__ unsigned char rowvalid[IMC_MAXN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NW = blockDim.x >> 6, NT = blockDim.x;          // one workgroup of 16 waves: every phase strides over waves / threads
  float* E = ws;
  for (int i = wave; i < N; i += NW) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) { float v = emb[(long)i * D + d]; s += v * v; }
    s = sqrtf(wave_sum(s));
    float dn = fmaxf(s, 1e-6f);
    for (int d = lane; d < D; d += 64) E[(long)i * D + d] = emb[(long)i * D + d] / dn;
    if (lane == 0) nrm[i] = s;
  }
  __syncthreads();
  for (int p = wave; p < N * N; p += NW) {
    int i = p / N, j = p % N;
    if (j <= i) { if (lane == 0) { Wm[i][j] = 0.f; Pm[i][j] = 0; Gm[i][j] = 0; } continue; }
    float dot = 0.f;
    for (int d = lane; d < D; d += 64) dot += E[(long)i * D + d] * E[(long)j * D + d];
    dot = wave_sum(dot);
    int same = 1, inter = 0;
    for (int c = lane; c < L; c += 64) {
      float a = label[(long)i * L + c], b = label[(long)j * L + c];
      if (a != b) same = 0;
      inter += ((long)a & (long)b) ? 1 : 0;
    }
    same = __all(same);
    inter = (int)wave_sum((float)inter);
    if (lane == 0) { Wm[i][j] = __expf(dot / 0.1f); Pm[i][j] = same; Gm[i][j] = (inter == 0); }
  }
  __syncthreads();
  if (tid < N) {
    int i = tid;
    float sp = 1e-6f, sn = 1e-6f;
    int vp = 0, vn = 0;
    for (int j = i + 1; j < N; ++j) {
      if (Pm[i][j]) { sp += Wm[i][j]; ++vp; }
      if (Gm[i][j]) { sn += Wm[i][j]; ++vn; }
    }
    bool valid = vp >= 1 && vn >= 1 && vn > vp;
    rowloss[i] = valid ? -logf(sp / (sp + sn)) / N : 0.f;
    rowvalid[i] = valid;
    if (valid) {
      rowk1[i] = (-1.f / sp + 1.f / (sp + sn)) / N;   // d/d sp
      rowk2[i] = (1.f / (sp + sn)) / N;               // d/d sn
    } else {
      rowk1[i] = rowk2[i] = 0.f;
    }
  }
  __syncthreads();
  // pair weights dL/d(dot_ij), upper triangle
  for (int p = tid; p < N * N; p += NT) {
    int i = p / N, j = p % N;
    if (j > i) {
      float w = (Pm[i][j] ? rowk1[i] : 0.f) + (Gm[i][j] ? rowk2[i] : 0.f);
      Wm[i][j] = w * Wm[i][j] / 0.1f;
    }
  }
  __syncthreads();
  for (int p = tid; p < N * N; p += NT) {
    int i = p / N, j = p % N;
    if (j < i) Wm[i][j] = Wm[j][i];
  }
  __syncthreads();
  // g_e[i] = sum_j Wsym[i][j] * e_j ; then back through the normalisation
  for (int i = wave; i < N; i += NW) {
    float dotge = 0.f;
    float gloc[16];   // D <= 1024
    int cnt = 0;
    for (int d = lane; d < D; d += 64, ++cnt) {
      float g = 0.f;
      for (int j = 0; j < N; ++j) g += Wm[i][j] * E[(long)j * D + d];
      gloc[cnt] = g;
      dotge += g * E[(long)i * D + d];
    }
    dotge = wave_sum(dotge);
    float s = nrm[i];
    cnt = 0;
    for (int d = lane; d < D; d += 64, ++cnt) {
      float g = gloc[cnt];
      float v = (s > 1e-6f) ? (g - E[(long)i * D + d] * dotge) / s : g / 1e-6f;
      gemb[(long)i * D + d] = v;
    }
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f, v = 0.f;
    for (int i = 0; i < N; ++i)
      if (rowvalid[i]) { l += rowloss[i]; v += 1.f; }
    out[0] = l; out[1] = v;
  }
}

// ---------------------------------------------------------------------------
// IMC on the matrix cores (round 4; loss_multilabel.py:36-66: s = exp(e_i . e_j / 0.1) over all pairs is the dense
// N x N x D contraction E E^T).  Two launches of ceil(N/16) workgroups, one per block of 16 anchor rows:
//   imc_gram_kernel: row norms; wave w computes the 16 x 16 tile (block, w) of emb emb^T on v_mfma_f32_16x16x4_f32 (one
//     16-byte load per operand, lane and four MFMAs); epilogue in registers: cosine, exp, the P / G pair masks from the
//     labels, the four row sums over j > i (DPP row reduction, then the waves' shares added in wave order), per-row loss and
//     its two derivatives.  Leaves s, the masks, the normalised embeddings of its rows and the per-row terms in the workspace.
//   imc_grad_kernel: g_e[i] = sum_j Wsym[i][j] e_j as a second MFMA product ([16 x N] x [N x D], the tiles of a row block
//     spread over the four waves, kept in registers), then back through the normalisation; workgroup 0 adds the blocks' losses
//     in block order.
// Every sum has a fixed order: bit-reproducible.  D % 16 == 0 (else the one-workgroup kernel above runs).
// workspace (floats): E [N][D] | nrm [N] | S [N][N] | F [N][N] | k1 [N] | k2 [N] | block loss, valid [2][4]
// ---------------------------------------------------------------------------
typedef float lf32x4 __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ float imc_row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}

struct ImcWs { float *E, *nrm, *S, *F, *k1, *k2, *blk; };
static __host__ __device__ __forceinline__ ImcWs imc_ws(float* ws, int N, int D) {
  ImcWs w;
  w.E = ws; w.nrm = w.E + (long)N * D; w.S = w.nrm + N; w.F = w.S + N * N; w.k1 = w.F + N * N; w.k2 = w.k1 + N; w.blk = w.k2 + N;
  return w;
}

__global__ __launch_bounds__(256) void imc_gram_kernel(const float* __restrict__ emb, const float* __restrict__ label, int N, int D, int L,
                                                       float* ws) {
  __shared__ float nrm_s[IMC_MAXN];
  __shared__ float lab_s[IMC_MAXN * 32];
  __shared__ float rs[4][16][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * 16;
  const ImcWs w = imc_ws(ws, N, D);
  for (int i = tid; i < N * L; i += 256) lab_s[(i / L) * 32 + i % L] = label[i];
  for (int i = wave; i < N; i += 4) {
    float s = 0.f;
    for (int d = 4 * lane; d < D; d += 256) { const float4 v = ld4(emb + (long)i * D + d); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    s = sqrtf(wave_sum(s));
    if (lane == 0) nrm_s[i] = s;
    if (i >= i0 && i < i0 + 16) {
      const float dn = fmaxf(s, 1e-6f);
      for (int d = 4 * lane; d < D; d += 256) {
        float4 v = ld4(emb + (long)i * D + d);
        v.x /= dn; v.y /= dn; v.z /= dn; v.w /= dn;
        st4(w.E + (long)i * D + d, v);
      }
      if (lane == 0) w.nrm[i] = s;
    }
  }
  __syncthreads();
  float sp[4] = {0.f, 0.f, 0.f, 0.f}, sn[4] = {0.f, 0.f, 0.f, 0.f}, vp[4] = {0.f, 0.f, 0.f, 0.f}, vn[4] = {0.f, 0.f, 0.f, 0.f};
  if (16 * wave < N) {
    const int ra = min(i0 + l15, N - 1), rb = min(16 * wave + l15, N - 1);
    const float* pa = emb + (long)ra * D + 4 * g;
    const float* pb = emb + (long)rb * D + 4 * g;
    lf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int k = 0; k < D; k += 16) {
      const float4 a4 = ld4(pa + k), b4 = ld4(pb + k);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    const int j = 16 * wave + l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 4 * g + r;                            // acc[r] = emb_i . emb_j
      if (i < N && j < N) {
        const float cosv = acc[r] / (fmaxf(nrm_s[i], 1e-6f) * fmaxf(nrm_s[j], 1e-6f));
        const float s = __expf(cosv / 0.1f);
        int same = 1, inter = 0;
        for (int c = 0; c < L; ++c) {
          const float a = lab_s[i * 32 + c], b = lab_s[j * 32 + c];
          if (a != b) same = 0;
          inter += ((long)a & (long)b) ? 1 : 0;
        }
        const int P = same, G = inter == 0;
        w.S[i * N + j] = s;
        w.F[i * N + j] = (float)(P | (G << 1));
        if (j > i) {
          if (P) { sp[r] += s; vp[r] += 1.f; }
          if (G) { sn[r] += s; vn[r] += 1.f; }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = imc_row16_sum(sp[r]), b = imc_row16_sum(sn[r]), c = imc_row16_sum(vp[r]), d = imc_row16_sum(vn[r]);
    if (l15 == 0) { rs[wave][4 * g + r][0] = a; rs[wave][4 * g + r][1] = b; rs[wave][4 * g + r][2] = c; rs[wave][4 * g + r][3] = d; }
  }
  __syncthreads();
  __shared__ float rl[16], rv[16];
  if (tid < 16) {
    const int i = i0 + tid;
    float a = 1e-6f, b = 1e-6f, c = 0.f, d = 0.f;
    for (int wv = 0; wv < 4; ++wv) { a += rs[wv][tid][0]; b += rs[wv][tid][1]; c += rs[wv][tid][2]; d += rs[wv][tid][3]; }
    const bool valid = i < N && c >= 1.f && d >= 1.f && d > c;
    rl[tid] = valid ? -logf(a / (a + b)) / N : 0.f;
    rv[tid] = valid ? 1.f : 0.f;
    if (i < N) {
      w.k1[i] = valid ? (-1.f / a + 1.f / (a + b)) / N : 0.f;   // d loss / d sp
      w.k2[i] = valid ? (1.f / (a + b)) / N : 0.f;              // d loss / d sn
    }
  }
  __syncthreads();
  if (tid == 0) {
    float l = 0.f, v = 0.f;
    for (int t = 0; t < 16; ++t) { l += rl[t]; v += rv[t]; }
    w.blk[blockIdx.x] = l; w.blk[4 + blockIdx.x] = v;
  }
}

__global__ __launch_bounds__(256) void imc_grad_kernel(int N, int D, float* ws, float* __restrict__ out, float* __restrict__ gemb) {
  __shared__ float Wm[16][IMC_MAXN + 4];
  __shared__ float k1s[IMC_MAXN], k2s[IMC_MAXN];
  __shared__ float dg[4][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * 16, NB = gridDim.x;
  const ImcWs w = imc_ws(ws, N, D);
  if (blockIdx.x == 0 && tid == 0) {
    float l = 0.f, v = 0.f;
    for (int b = 0; b < NB; ++b) { l += w.blk[b]; v += w.blk[4 + b]; }
    out[0] = l; out[1] = v;
  }
  if (tid < N) { k1s[tid] = w.k1[tid]; k2s[tid] = w.k2[tid]; }
  __syncthreads();
  const int Np = (N + 3) & ~3;
  for (int p = tid; p < 16 * IMC_MAXN; p += 256) {
    const int r = p / IMC_MAXN, j = p % IMC_MAXN, i = i0 + r;
    float v = 0.f;
    if (i < N && j < N && j != i) {
      const int f = (int)w.F[i * N + j], mn = min(i, j);
      v = ((f & 1 ? k1s[mn] : 0.f) + (f & 2 ? k2s[mn] : 0.f)) * w.S[i * N + j] / 0.1f;
    }
    Wm[r][j] = v;
  }
  __syncthreads();
  // g_e tiles of this wave: columns 16 t .. 16 t + 15 for t = wave, wave + 4, ...; acc[r] = g_e[i0 + 4 g + r][16 t + l15]
  constexpr int MAXT = 1024 / 16 / 4;
  lf32x4 acc[MAXT];
  const int nt = D / 16;
  float dot[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < MAXT; ++u) {
    acc[u] = lf32x4{0.f, 0.f, 0.f, 0.f};
    const int t = wave + 4 * u;
    if (t < nt) {
      for (int k = 0; k < Np; k += 4) {
        const int j = min(k + g, N - 1);                          // Wm is zero beyond N
        acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wm[l15][k + g], w.E[(long)j * D + 16 * t + l15], acc[u], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = min(i0 + 4 * g + r, N - 1);
        dot[r] += acc[u][r] * w.E[(long)i * D + 16 * t + l15];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a = imc_row16_sum(dot[r]);
    if (l15 == 0) dg[wave][4 * g + r] = a;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rr = 4 * g + r, i = i0 + rr;
    if (i >= N) continue;
    const float dotge = dg[0][rr] + dg[1][rr] + dg[2][rr] + dg[3][rr];
    const float s = w.nrm[i];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int t = wave + 4 * u;
      if (t < nt) {
        const int d = 16 * t + l15;
        const float ge = acc[u][r];
        gemb[(long)i * D + d] = (s > 1e-6f) ? (ge - w.E[(long)i * D + d] * dotge) / s : ge / 1e-6f;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// cam_softmaxnorm (train_mcl.py:30-36) on NCHW [N, K, H, W]: fg = softmax over channels 1..K-1,
// bg = 1 - max fg.  One thread per pixel; channel reads are plane-strided, x-contiguous.
// ---------------------------------------------------------------------------
constexpr int KMAX = 32;

__device__ __forceinline__ void softmaxnorm_px(const float* x, long plane, int K, float* o, int* amax) {
  float mx = -INFINITY;
  for (int k = 1; k < K; ++k) { o[k] = x[k * plane]; mx = fmaxf(mx, o[k]); }
  float s = 0.f;
  for (int k = 1; k < K; ++k) { o[k] = __expf(o[k] - mx); s += o[k]; }
  float inv = 1.f / s, best = -1.f;
  int bi = 1;
  for (int k = 1; k < K; ++k) { o[k] *= inv; if (o[k] > best) { best = o[k]; bi = k; } }
  o[0] = 1.f - best;
  *amax = bi;
}

__global__ __launch_bounds__(256) void softmaxnorm_kernel(const float* x, const float* gy, float* out, int N, int K, long HW,
                                                          int bwd) {
  const long total = (long)N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    const float* xb = x + n * K * HW + p;
    float o[KMAX];
    int am;
    softmaxnorm_px(xb, HW, K, o, &am);
    float* ob = out + n * K * HW + p;
    if (!bwd) {
      for (int k = 0; k < K; ++k) ob[k * HW] = o[k];
    } else {
      const float* gb = gy + n * K * HW + p;
      float g[KMAX];
      float dot = 0.f;
      for (int k = 1; k < K; ++k) g[k] = gb[k * HW];
      g[am] -= gb[0];
      for (int k = 1; k < K; ++k) dot += g[k] * o[k];
      ob[0] = 0.f;
      for (int k = 1; k < K; ++k) ob[k * HW] = o[k] * (g[k] - dot);
    }
  }
}

// ---------------------------------------------------------------------------
// ER loss: d[n, k, p] = | sm(cams)[k] - sm(sgcs)[k] | * m[n,k]; top-k mean per row by exact radix select.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void er_diff_kernel(const float* cams, const float* sgcs, const float* lwb, float* d, int N,
                                                      int K, long HW) {
  const long total = (long)N * HW;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    float a[KMAX], b[KMAX];
    int t0, t1;
    softmaxnorm_px(cams + n * K * HW + p, HW, K, a, &t0);
    softmaxnorm_px(sgcs + n * K * HW + p, HW, K, b, &t1);
    for (int k = 0; k < K; ++k) d[n * K * HW + k * HW + p] = fabsf(a[k] - b[k]) * lwb[n * K + k];
  }
}

constexpr int RBINS = 2048;
// Bin sums are kept in 64-bit FIXED POINT (value * 2^36, values are |softmax difference| * mask <= 1; rows of < 2^24
// elements): integer addition is associative, so the histogram - and with it the loss - has the same bits whatever order
// the workgroups and their LDS / global atomics land in; fp32 atomics moved the last bits of loss_er from run to run.
// 2^-36 per element is far below the fp32 round-off of the sum it replaces.
#define ER_FIX_SCALE 68719476736.0f       /* 2^36 */
#define ER_FIX_INV (1.0 / 68719476736.0)
typedef unsigned long long er_fix_t;
__device__ __forceinline__ er_fix_t er_fix(float v) { return (er_fix_t)(v * ER_FIX_SCALE); }
// histogram of one radix digit over the elements whose higher digits equal prefix[n]
__global__ __launch_bounds__(256) void er_hist_kernel(const float* d, long row_len, int shift, int nbits, unsigned himask,
                                                      const unsigned* prefix, unsigned* hcnt, er_fix_t* hsum) {
  __shared__ unsigned lc[RBINS];
  __shared__ er_fix_t ls[RBINS];
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < RBINS; i += 256) { lc[i] = 0; ls[i] = 0; }
  __syncthreads();
  const unsigned pf = prefix[n], dm = (1u << nbits) - 1u;
  const float* row = d + n * row_len;
  const long per = (row_len + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per, end = min(row_len, beg + per);
  for (long i = beg + threadIdx.x; i < end; i += 256) {
    float v = row[i];
    unsigned key = __float_as_uint(v);
    if (key != 0u && (key & himask) == pf) {     // exact zeros (masked channels) cannot change the top-k sum
      unsigned b = (key >> shift) & dm;
      atomicAdd(&lc[b], 1u);
      atomicAdd(&ls[b], er_fix(v));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RBINS; i += 256)
    if (lc[i]) { atomicAdd(&hcnt[n * RBINS + i], lc[i]); atomicAdd(&hsum[n * RBINS + i], ls[i]); }
}

// per row: pick the digit of the k-th largest; state = {krem (still to take), prefix, sum_gt, cnt_eq}.
// One wave per row: lane l owns the l-th chunk of bins counted from the top; a wave scan of the chunk counts finds the
// chunk in which the running count reaches k, and only that lane walks its bins.  (One thread per row walking up to
// 2048 bins through dependent global loads took 340 us per pass, three passes per step.)
__global__ __launch_bounds__(64) void er_scan_kernel(const unsigned* hcnt, const er_fix_t* hsum, int shift, int nbits, unsigned* krem,
                                                     unsigned* prefix, er_fix_t* sum_gt, unsigned* cnt_eq, int N) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n >= N) return;
  const int nb = 1 << nbits, per = nb / 64;          // nbits is 10 or 11
  const unsigned* hc = hcnt + (long)n * RBINS;
  const er_fix_t* hs = hsum + (long)n * RBINS;
  const int top = nb - 1 - lane * per;               // this lane's bins: top, top-1, ..., top-per+1
  unsigned cl = 0;
  er_fix_t sl = 0;
  for (int i = 0; i < per; ++i) { cl += hc[top - i]; sl += hs[top - i]; }
  unsigned incl = cl;                                // inclusive scan over lanes (from the top bin down)
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  const unsigned k0 = krem[n];
  const unsigned long long m = __ballot(incl >= k0);
  const int L = m ? (__ffsll((long long)m) - 1) : 63;  // no chunk reaches k: the walk ends at bin 0, which lane 63 owns
  er_fix_t s_above = lane < L ? sl : 0;                // integer wave sum: exact
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s_above += (er_fix_t)__shfl_xor((long long)s_above, o, 64);
  const unsigned k_above = __shfl(incl - cl, L, 64);
  if (lane != L) return;
  unsigned k = k0 - k_above;
  er_fix_t s = sum_gt[n] + s_above;
  int b = top;
  for (; b > 0 && b > top - per; --b) {
    unsigned c = hc[b];
    if (c >= k) break;
    k -= c;
    s += hs[b];
  }
  if (b == top - per) b = top - per + 1;             // cannot happen when incl >= k0; keeps b inside the chunk
  krem[n] = k;
  sum_gt[n] = s;
  prefix[n] |= ((unsigned)b) << shift;
  cnt_eq[n] = hc[b];
}

// loss = sum_n (sum_gt[n] + krem[n] * tau[n]) / (N * k)
// k_dev (optional, device int): the top-k count read at run time (hipGraph replays: k = int(0.2 * sum(labels) * H * W) changes
// with every batch, train_mcl.py:178,188)
__global__ void er_krem_init_kernel(unsigned* krem, int N, const int* k_dev) {
  for (int n = threadIdx.x; n < N; n += 64) krem[n] = (unsigned)k_dev[0];
}

__global__ void er_final_kernel(const unsigned* krem, const unsigned* prefix, const er_fix_t* sum_gt, int N, float inv_nk, float* loss,
                                const int* k_dev) {
  if (k_dev) inv_nk = 1.0f / ((float)N * (float)k_dev[0]);
  double acc = 0.0;
  for (int n = threadIdx.x; n < N; n += 64) acc += (double)sum_gt[n] * ER_FIX_INV + (double)krem[n] * (double)__uint_as_float(prefix[n]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (threadIdx.x == 0) loss[0] = (float)(acc * (double)inv_nk);
}

// gradient w.r.t. raw sgcs: selected elements get -sign(c - s) * m / (N k) (ties at tau share krem/cnt_eq),
// pulled back through cam_softmaxnorm.
__global__ __launch_bounds__(256) void er_bwd_kernel(const float* cams, const float* sgcs, const float* lwb, const unsigned* prefix,
                                                     const unsigned* krem, const unsigned* cnt_eq, const float* gup, float gscale,
                                                     float* gsg, int N, int K, long HW) {
  const long total = (long)N * HW;
  if (gup) gscale *= gup[0];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    long n = i / HW, p = i - n * HW;
    float a[KMAX], b[KMAX], g[KMAX];
    int t0, am;
    softmaxnorm_px(cams + n * K * HW + p, HW, K, a, &t0);
    softmaxnorm_px(sgcs + n * K * HW + p, HW, K, b, &am);
    const unsigned tk = prefix[n];
    const float tiew = cnt_eq[n] ? (float)krem[n] / (float)cnt_eq[n] : 0.f;
    for (int k = 0; k < K; ++k) {
      float m = lwb[n * K + k];
      float df = a[k] - b[k];
      unsigned key = __float_as_uint(fabsf(df) * m);
      float w = (key > tk) ? 1.f : ((key == tk) ? tiew : 0.f);
      float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
      g[k] = -sg * m * w * gscale;
    }
    float dot = 0.f;
    g[am] -= g[0];
    for (int k = 1; k < K; ++k) dot += g[k] * b[k];
    float* ob = gsg + n * K * HW + p;
    ob[0] = 0.f;
    for (int k = 1; k < K; ++k) ob[k * HW] = b[k] * (g[k] - dot);
  }
}

// ---------------------------------------------------------------------------
// Adam, L2 weight decay folded into the gradient (torch.optim.Adam, not AdamW), flat arrays
// ---------------------------------------------------------------------------
// dyn (optional, device): {lr, bias_corr1, sqrt_bias_corr2} read at run time instead of the launch arguments, so that a
// captured hipGraph of the step keeps advancing Adam's step count and follows the LR scheduler when it is replayed
__global__ __launch_bounds__(256) void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2s, const float* dyn) {
  if (dyn) { lr = dyn[0]; bc1 = dyn[1]; bc2s = dyn[2]; }
  for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
    if (i + 3 < n) {
      float4 pp = ld4(p + i), gg = ld4(g + i), mm = ld4(m + i), vv = ld4(v + i);
#define ADAM1(f)                                            \
  {                                                         \
    float gr = gg.f + wd * pp.f;                            \
    mm.f = b1 * mm.f + (1.f - b1) * gr;                     \
    vv.f = b2 * vv.f + (1.f - b2) * gr * gr;                \
    pp.f -= (lr / bc1) * mm.f / (sqrtf(vv.f) / bc2s + eps); \
  }
      ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
      st4(p + i, pp); st4(m + i, mm); st4(v + i, vv);
    } else {
      for (long j = i; j < n; ++j) {
        float gr = g[j] + wd * p[j];
        m[j] = b1 * m[j] + (1.f - b1) * gr;
        v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
        p[j] -= (lr / bc1) * m[j] / (sqrtf(v[j]) / bc2s + eps);
      }
    }
  }
}

static int gs(long n) { long b = (n + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

extern "C" {

int mx_cls_loss(int mode, const float* x, int ldx, const float* y, int ldy, float* loss, float* grad, int ldg, int N, int C,
                void* stream) {
  MX_CHECK_ARG(mode >= 0 && mode <= 4 && x && y && grad && N > 0 && C > 0, "cls_loss: bad args");
  MX_CHECK_ARG(mode >= 3 || loss, "cls_loss: loss output required");
  if (mode <= 1) {
    MX_CHECK_ARG(N <= CLS_MAXN, "cls_loss: N=%d (<= %d)", N, CLS_MAXN);
    hipLaunchKernelGGL(cls_scalar_loss_kernel, dim3(1), dim3(N >= 16 ? 1024 : 64 * N), 0, (hipStream_t)stream, mode, x, ldx, y, ldy, loss,
                       grad, ldg, N, C);
  } else {
    hipLaunchKernelGGL(cls_loss_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, mode, x, ldx, y, ldy, loss, grad, ldg, N, C);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_imc(const float* emb, const float* label, int N, int D, int L, float* out2, float* gemb, float* workspace, void* stream) {
  MX_CHECK_ARG(emb && label && out2 && gemb && workspace, "imc: null pointer");
  MX_CHECK_ARG(N > 0 && N <= IMC_MAXN && D > 0 && D <= 1024 && L > 0, "imc: N=%d (<=64) D=%d (<=1024) L=%d", N, D, L);
  static const int mfma = getenv("MX_IMC_MFMA") ? atoi(getenv("MX_IMC_MFMA")) : 1;
  if (mfma && D % 16 == 0 && L <= 32 && (((uintptr_t)emb | (uintptr_t)workspace) & 15) == 0) {
    const int nb = cdiv(N, 16);
    hipLaunchKernelGGL(imc_gram_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, emb, label, N, D, L, workspace);
    MX_LAUNCH_CHECK();
    hipLaunchKernelGGL(imc_grad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, N, D, workspace, out2, gemb);
  } else {
    hipLaunchKernelGGL(imc_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, emb, label, N, D, L, out2, gemb, workspace);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_softmaxnorm(const float* x, const float* gy, float* out, int N, int K, long HW, int bwd, void* stream) {
  MX_CHECK_ARG(x && out && N > 0 && K >= 2 && K <= KMAX && HW > 0 && (!bwd || gy), "softmaxnorm: bad args K=%d", K);
  hipLaunchKernelGGL(softmaxnorm_kernel, dim3(gs((long)N * HW)), dim3(256), 0, (hipStream_t)stream, x, gy, out, N, K, HW, bwd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// ER forward.  workspace layout (caller-allocated, zero-filled): d [N*K*HW] floats, then
//   state: krem[N] u32, prefix[N] u32, sum_gt[N] u64 (fixed point), cnt_eq[N] u32, hcnt[N*2048] u32, hsum[N*2048] u64
int mx_er_fwd(const float* cams, const float* sgcs, const float* lwb, int N, int K, long HW, long k, float* d, unsigned* krem,
              unsigned* prefix, unsigned long long* sum_gt, unsigned* cnt_eq, unsigned* hcnt, unsigned long long* hsum, float* loss, void* stream) {
  MX_CHECK_ARG(cams && sgcs && lwb && d && krem && prefix && sum_gt && cnt_eq && hcnt && hsum && loss, "er_fwd: null pointer");
  MX_CHECK_ARG(N > 0 && K >= 2 && K <= KMAX && HW > 0, "er_fwd: bad extents");
  MX_CHECK_ARG(k >= 1 && k <= (long)K * HW, "er_fwd: k=%ld out of range for rows of %ld (torch.topk would raise)", k, (long)K * HW);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(er_diff_kernel, dim3(gs((long)N * HW)), dim3(256), 0, st, cams, sgcs, lwb, d, N, K, HW);
  // krem <- k  (prefix, sum_gt zeroed by the caller)
  {
    unsigned kk = (unsigned)k;
    hipMemsetD32Async((hipDeviceptr_t)krem, (int)kk, N, st);
  }
  const long row_len = (long)K * HW;
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  const unsigned himask[3] = {0u, 0xFFE00000u, 0xFFFFFC00u};
  int chunks = (int)((row_len + 16383) / 16384);
  if (chunks > 512) chunks = 512;
  for (int ps = 0; ps < 3; ++ps) {
    hipMemsetAsync(hcnt, 0, sizeof(unsigned) * N * RBINS, st);
    hipMemsetAsync(hsum, 0, sizeof(er_fix_t) * N * RBINS, st);
    hipLaunchKernelGGL(er_hist_kernel, dim3(chunks, N), dim3(256), 0, st, d, row_len, shifts[ps], bits[ps], himask[ps], prefix,
                       hcnt, hsum);
    hipLaunchKernelGGL(er_scan_kernel, dim3(N), dim3(64), 0, st, hcnt, hsum, shifts[ps], bits[ps], krem, prefix, sum_gt,
                       cnt_eq, N);
  }
  hipLaunchKernelGGL(er_final_kernel, dim3(1), dim3(64), 0, st, krem, prefix, sum_gt, N, 1.0f / ((float)N * (float)k), loss, (const int*)nullptr);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_er_bwd(const float* cams, const float* sgcs, const float* lwb, const unsigned* prefix, const unsigned* krem,
              const unsigned* cnt_eq, const float* gup, float gscale, float* gsgcs, int N, int K, long HW, void* stream) {
  MX_CHECK_ARG(cams && sgcs && lwb && prefix && krem && cnt_eq && gsgcs && N > 0 && K >= 2 && K <= KMAX && HW > 0, "er_bwd: bad args");
  hipLaunchKernelGGL(er_bwd_kernel, dim3(gs((long)N * HW)), dim3(256), 0, (hipStream_t)stream, cams, sgcs, lwb, prefix, krem,
                     cnt_eq, gup, gscale, gsgcs, N, K, HW);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
            float weight_decay, float bias_corr1, float sqrt_bias_corr2, const float* dyn, void* stream) {
  MX_CHECK_ARG(p && g && m && v && n > 0, "adam: bad args");
  MX_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(gs((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bias_corr1, sqrt_bias_corr2, dyn);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// ER loss straight from the low-resolution maps (no full-resolution tensor is ever written):
// cams_full / sgcs_full (MuSCLe.py:256-257, bilinear align_corners) are recomputed per pixel from the
// [N,h,w,L] NHWC low-res CAM / SGC, then cam_softmaxnorm, mask, |diff| and the radix select as above.
// Exact zeros (masked channels: ~88 % of all elements) are never histogrammed: they cannot change the sum, and
// 64 lanes hitting LDS bin 0 serialised the pass.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void lr_coord(int d, int in, int out, int& i0, int& i1, float& w1) {
  float scale = (out > 1) ? (float)(in - 1) / (float)(out - 1) : 0.f;
  float s = scale * d;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  w1 = s - i0;
}

// fills a[K], b[K] with the softmaxnorm'ed upsampled cam / sgc at (n, Y, X); returns argmax of b's foreground
__device__ __forceinline__ int lr_pixel(const float* cam, const float* sgc, int n, int Y, int X, int h, int w, int L, int K, int H,
                                        int W, float* a, float* b, float& wy, float& wx, int& y0, int& y1, int& x0, int& x1) {
  lr_coord(Y, h, H, y0, y1, wy);
  lr_coord(X, w, W, x0, x1, wx);
  const long base = (long)n * h * w * L;
  const float* c00 = cam + base + ((long)y0 * w + x0) * L; const float* c01 = cam + base + ((long)y0 * w + x1) * L;
  const float* c10 = cam + base + ((long)y1 * w + x0) * L; const float* c11 = cam + base + ((long)y1 * w + x1) * L;
  const float* s00 = sgc + base + ((long)y0 * w + x0) * L; const float* s01 = sgc + base + ((long)y0 * w + x1) * L;
  const float* s10 = sgc + base + ((long)y1 * w + x0) * L; const float* s11 = sgc + base + ((long)y1 * w + x1) * L;
  float ma = -INFINITY, mb = -INFINITY;
  // cells are L floats apart with L % 4 == 0 (checked by the callers): 16-byte loads, four classes per step; the
  // interpolation of each class is the same expression as before (bit-identical)
  const float u00 = (1.f - wx), u01 = wx, v0 = (1.f - wy), v1 = wy;
  for (int k4 = 0; k4 < K; k4 += 4) {
    const float4 p00 = ld4(c00 + k4), p01 = ld4(c01 + k4), p10 = ld4(c10 + k4), p11 = ld4(c11 + k4);
    const float4 q00 = ld4(s00 + k4), q01 = ld4(s01 + k4), q10 = ld4(s10 + k4), q11 = ld4(s11 + k4);
    const float av[4] = {v0 * (u00 * p00.x + u01 * p01.x) + v1 * (u00 * p10.x + u01 * p11.x),
                         v0 * (u00 * p00.y + u01 * p01.y) + v1 * (u00 * p10.y + u01 * p11.y),
                         v0 * (u00 * p00.z + u01 * p01.z) + v1 * (u00 * p10.z + u01 * p11.z),
                         v0 * (u00 * p00.w + u01 * p01.w) + v1 * (u00 * p10.w + u01 * p11.w)};
    const float bv[4] = {v0 * (u00 * q00.x + u01 * q01.x) + v1 * (u00 * q10.x + u01 * q11.x),
                         v0 * (u00 * q00.y + u01 * q01.y) + v1 * (u00 * q10.y + u01 * q11.y),
                         v0 * (u00 * q00.z + u01 * q01.z) + v1 * (u00 * q10.z + u01 * q11.z),
                         v0 * (u00 * q00.w + u01 * q01.w) + v1 * (u00 * q10.w + u01 * q11.w)};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k4 + j;
      if (k >= 1 && k < K) { a[k] = av[j]; b[k] = bv[j]; ma = fmaxf(ma, av[j]); mb = fmaxf(mb, bv[j]); }
    }
    // at most two chunks (16 loads) in flight: hoisting all six chunks' loads costs 190 registers and the callers' occupancy
    if ((k4 & 4) != 0) __builtin_amdgcn_sched_barrier(0);
  }
  float sa = 0.f, sb = 0.f;
  for (int k = 1; k < K; ++k) { a[k] = __expf(a[k] - ma); sa += a[k]; b[k] = __expf(b[k] - mb); sb += b[k]; }
  float ia = 1.f / sa, ib = 1.f / sb, besta = -1.f, bestb = -1.f;
  int am = 1;
  for (int k = 1; k < K; ++k) {
    a[k] *= ia; b[k] *= ib;
    besta = fmaxf(besta, a[k]);
    if (b[k] > bestb) { bestb = b[k]; am = k; }
  }
  a[0] = 1.f - besta; b[0] = 1.f - bestb;
  return am;
}

__global__ __launch_bounds__(256) void er_lr_hist_kernel(const float* cam, const float* sgc, const float* lwb, int h, int w, int L,
                                                         int K, int H, int W, int shift, int nbits, unsigned himask,
                                                         const unsigned* prefix, unsigned* hcnt, er_fix_t* hsum, float* vals) {
  __shared__ unsigned lc[RBINS];
  __shared__ er_fix_t ls[RBINS];
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < RBINS; i += 256) { lc[i] = 0; ls[i] = 0; }
  __syncthreads();
  const unsigned pf = prefix[n], dm = (1u << nbits) - 1u;
  const long HW = (long)H * W;
  const long per = (HW + gridDim.x - 1) / gridDim.x;
  const long beg = blockIdx.x * per, end = min(HW, beg + per);
  for (long p = beg + threadIdx.x; p < end; p += 256) {
    float a[KMAX], b[KMAX], wy, wx;
    int y0, y1, x0, x1;
    lr_pixel(cam, sgc, n, (int)(p / W), (int)(p % W), h, w, L, K, H, W, a, b, wy, wx, y0, y1, x0, x1);
    for (int k = 0; k < K; ++k) {
      float m = lwb[n * K + k];
      if (m == 0.f) continue;
      float v = fabsf(a[k] - b[k]) * m;
      // first pass of the select: park the value (planes of the ACTIVE classes only, ~12 % of [N,K,H,W]) so that the two
      // later digit passes read 4 bytes per value instead of re-evaluating the upsample and both softmaxes per pixel
      if (vals) vals[((long)n * K + k) * HW + p] = v;
      unsigned key = __float_as_uint(v);
      if (key != 0u && (key & himask) == pf) {
        unsigned bb = (key >> shift) & dm;
        atomicAdd(&lc[bb], 1u);
        atomicAdd(&ls[bb], er_fix(v));
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RBINS; i += 256)
    if (lc[i]) { atomicAdd(&hcnt[n * RBINS + i], lc[i]); atomicAdd(&hsum[n * RBINS + i], ls[i]); }
}

// digit passes 2 and 3 from the parked values: planes of the active classes of sample blockIdx.y, pixel chunk blockIdx.x
__global__ __launch_bounds__(256) void er_vals_hist_kernel(const float* vals, const float* lwb, int K, long HW, int shift, int nbits,
                                                           unsigned himask, const unsigned* prefix, unsigned* hcnt, er_fix_t* hsum) {
  __shared__ unsigned lc[RBINS];
  __shared__ er_fix_t ls[RBINS];
  const int n = blockIdx.y;
  for (int i = threadIdx.x; i < RBINS; i += 256) { lc[i] = 0; ls[i] = 0; }
  __syncthreads();
  const unsigned pf = prefix[n], dm = (1u << nbits) - 1u;
  const long per = ((HW + gridDim.x - 1) / gridDim.x + 3) & ~3L;        // multiples of 4 pixels: 16-byte loads
  const long beg = blockIdx.x * per, end = min(HW, beg + per);
  for (int k = 0; k < K; ++k) {
    if (lwb[n * K + k] == 0.f) continue;
    const float* row = vals + ((long)n * K + k) * HW;
    for (long i = beg + 4 * threadIdx.x; i < end; i += 1024) {
      float v4[4];
      if (i + 3 < end && (HW & 3) == 0) { const float4 q = ld4(row + i); v4[0] = q.x; v4[1] = q.y; v4[2] = q.z; v4[3] = q.w; }
      else { for (int e = 0; e < 4; ++e) v4[e] = (i + e < end) ? row[i + e] : 0.f; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned key = __float_as_uint(v4[e]);
        if (key != 0u && (key & himask) == pf) {
          const unsigned bb = (key >> shift) & dm;
          atomicAdd(&lc[bb], 1u);
          atomicAdd(&ls[bb], er_fix(v4[e]));
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < RBINS; i += 256)
    if (lc[i]) { atomicAdd(&hcnt[n * RBINS + i], lc[i]); atomicAdd(&hsum[n * RBINS + i], ls[i]); }
}

// gradient w.r.t. the low-res SGC: one workgroup per low-res cell gathers from the full-res pixels it feeds
__global__ __launch_bounds__(256) void er_lr_bwd_kernel(const float* cam, const float* sgc, const float* lwb, const unsigned* prefix,
                                                        const unsigned* krem, const unsigned* cnt_eq, const float* gup,
                                                        float gscale, const int* k_dev, float* gsgc, int h, int w, int L, int K, int H, int W) {
  __shared__ float red[4][KMAX];
  const int cell = blockIdx.x, n = blockIdx.y;
  const int cy = cell / w, cx = cell % w;
  if (k_dev) gscale = 1.0f / ((float)gridDim.y * (float)k_dev[0]);
  if (gup) gscale *= gup[0];
  // full-res rows / cols whose interpolation touches (cy, cx): source coordinate in (cy-1, cy+1)
  const float sy = (H > 1) ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = (W > 1) ? (float)(w - 1) / (float)(W - 1) : 0.f;
  int Ylo = (sy > 0.f) ? (int)floorf((cy - 1) / sy) : 0, Yhi = (sy > 0.f) ? (int)ceilf((cy + 1) / sy) : H - 1;
  int Xlo = (sx > 0.f) ? (int)floorf((cx - 1) / sx) : 0, Xhi = (sx > 0.f) ? (int)ceilf((cx + 1) / sx) : W - 1;
  Ylo = max(Ylo, 0); Yhi = min(Yhi, H - 1); Xlo = max(Xlo, 0); Xhi = min(Xhi, W - 1);
  const int nx = Xhi - Xlo + 1, cnt = (Yhi - Ylo + 1) * nx;
  const unsigned tk = prefix[n];
  const float tiew = cnt_eq[n] ? (float)krem[n] / (float)cnt_eq[n] : 0.f;
  float acc[KMAX];
  for (int k = 0; k < K; ++k) acc[k] = 0.f;
  for (int i = threadIdx.x; i < cnt; i += 256) {
    const int Y = Ylo + i / nx, X = Xlo + i % nx;
    float a[KMAX], b[KMAX], g[KMAX], wy, wx;
    int y0, y1, x0, x1;
    const int am = lr_pixel(cam, sgc, n, Y, X, h, w, L, K, H, W, a, b, wy, wx, y0, y1, x0, x1);
    float wgt = 0.f;     // bilinear weight of this cell for this pixel
    if (y0 == cy) wgt += (x0 == cx ? (1.f - wy) * (1.f - wx) : 0.f) + ((x1 == cx && x1 != x0) ? (1.f - wy) * wx : 0.f);
    if (y1 == cy && y1 != y0) wgt += (x0 == cx ? wy * (1.f - wx) : 0.f) + ((x1 == cx && x1 != x0) ? wy * wx : 0.f);
    if (wgt == 0.f) continue;
    for (int k = 0; k < K; ++k) {
      float m = lwb[n * K + k];
      float df = a[k] - b[k];
      unsigned key = __float_as_uint(fabsf(df) * m);
      float ww = (key > tk) ? 1.f : ((key == tk) ? tiew : 0.f);
      float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
      g[k] = -sg * m * ww * gscale;
    }
    g[am] -= g[0];
    float dot = 0.f;
    for (int k = 1; k < K; ++k) dot += g[k] * b[k];
    for (int k = 1; k < K; ++k) acc[k] += wgt * b[k] * (g[k] - dot);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 1; k < K; ++k) {
    float v = wave_sum(acc[k]);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < L) {
    int k = threadIdx.x;
    float v = (k >= 1 && k < K) ? red[0][k] + red[1][k] + red[2][k] + red[3][k] : 0.f;
    gsgc[((long)n * h * w + cell) * L + k] = v;
  }
}

// The same gradient with every full-resolution pixel evaluated ONCE (the gather kernel above evaluates each pixel for
// each of the up to four low-res cells it feeds).  One workgroup = a band of `ty` rows x 256 columns, thread = column.
// ty is chosen so that a band spans less than one low-res row spacing: its pixels touch at most three low-res rows, which are
// three register slots per class; the column direction goes through LDS atomics, then one global atomic per touched
// (cell, class).  Both levels add 64-bit FIXED-POINT integers (the unscaled per-pixel terms are O(1): value * 2^40), so
// the sums do not depend on the order the atomics land in; er_lr_bwd_finish_kernel converts and applies the common factor
// 1 / (N k) * upstream.  gacc must be zero-filled.
#define ER_BWD_FIX 1099511627776.0f        /* 2^40 */
template <int KT>
__global__ __launch_bounds__(256) void er_lr_bwd_band_kernel(const float* cam, const float* sgc, const float* lwb,
                                                             const unsigned* prefix, const unsigned* krem, const unsigned* cnt_eq,
                                                             unsigned long long* gacc, int h, int w, int L,
                                                             int H, int W, int ty) {
  extern __shared__ unsigned long long lacc[];                 // [3][ncx][KT]
  const int n = blockIdx.z, Y0 = blockIdx.y * ty, X = blockIdx.x * 256 + threadIdx.x;
  const int Y1 = min(H, Y0 + ty);
  int cb, t1, xb, xe, t2;
  float tw;
  lr_coord(Y0, h, H, cb, t1, tw);
  lr_coord(blockIdx.x * 256, w, W, xb, t2, tw);
  lr_coord(min(W - 1, blockIdx.x * 256 + 255), w, W, t1, xe, tw);
  const int ncx = xe - xb + 1;
  for (int i = threadIdx.x; i < 3 * ncx * KT; i += 256) lacc[i] = 0ull;
  __syncthreads();
  float S[3][KT];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < KT; ++k) S[r][k] = 0.f;
  int x0 = 0, x1 = 0;
  float wx = 0.f;
  if (X < W) {
    const unsigned tk = prefix[n];
    const float tiew = cnt_eq[n] ? (float)krem[n] / (float)cnt_eq[n] : 0.f;
    float m[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) m[k] = lwb[n * KT + k];
    for (int Y = Y0; Y < Y1; ++Y) {
      float a[KMAX], b[KMAX], g[KT], wy;
      int y0, y1;
      const int am = lr_pixel(cam, sgc, n, Y, X, h, w, L, KT, H, W, a, b, wy, wx, y0, y1, x0, x1);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        float df = a[k] - b[k];
        unsigned key = __float_as_uint(fabsf(df) * m[k]);
        float ww = (key > tk) ? 1.f : ((key == tk) ? tiew : 0.f);
        float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
        g[k] = -sg * m[k] * ww;
      }
      float g0 = g[0];
#pragma unroll
      for (int k = 1; k < KT; ++k) g[k] -= (k == am) ? g0 : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int k = 1; k < KT; ++k) dot += g[k] * b[k];
      // rows y0 and y1 (= y0 or y0 + 1) receive this pixel; y0 - cb is 0 or 1 for the whole band row (a band is shorter
      // than one low-res row spacing) and uniform over the workgroup: branch once per pixel row, not per class
      const int s0 = y0 - cb;
      const float w0 = 1.f - wy, w1 = (y1 != y0) ? wy : 0.f;
      if (s0 == 0) {
#pragma unroll
        for (int k = 1; k < KT; ++k) { const float v = b[k] * (g[k] - dot); S[0][k] += w0 * v; S[1][k] += w1 * v; }
      } else if (s0 == 1) {
#pragma unroll
        for (int k = 1; k < KT; ++k) { const float v = b[k] * (g[k] - dot); S[1][k] += w0 * v; S[2][k] += w1 * v; }
      } else {
#pragma unroll
        for (int k = 1; k < KT; ++k) { const float v = b[k] * (g[k] - dot); S[2][k] += (w0 + w1) * v; }
      }
    }
    const float u0 = 1.f - wx, u1 = (x1 != x0) ? wx : 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int k = 1; k < KT; ++k) {
        const float v = S[r][k];
        if (v != 0.f) {
          atomicAdd(&lacc[(r * ncx + (x0 - xb)) * KT + k], (unsigned long long)(long long)(u0 * v * ER_BWD_FIX));
          if (u1 != 0.f) atomicAdd(&lacc[(r * ncx + (x1 - xb)) * KT + k], (unsigned long long)(long long)(u1 * v * ER_BWD_FIX));
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * ncx * KT; i += 256) {
    const unsigned long long v = lacc[i];
    if (v == 0ull) continue;
    const int k = i % KT, cx = (i / KT) % ncx, r = i / (KT * ncx);
    const int cy = cb + r;
    if (cy < h) atomicAdd(&gacc[(((long)n * h + cy) * w + xb + cx) * L + k], v);
  }
}

// gsgc = fixed-point sums * 2^-40 * gscale * upstream
__global__ __launch_bounds__(256) void er_lr_bwd_finish_kernel(const unsigned long long* gacc, const float* gup, float gscale, const int* k_dev,
                                                               int N, float* gsgc, long total) {
  if (k_dev) gscale = 1.0f / ((float)N * (float)k_dev[0]);
  if (gup) gscale *= gup[0];
  const double f = (double)gscale / (double)ER_BWD_FIX;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) gsgc[i] = (float)((double)(long long)gacc[i] * f);
}

extern "C" {

// ER loss from the low-res NHWC maps cam/sgc [N,h,w,L] for an H x W image (train_mcl.py:175-188 + MuSCLe.py:256-257 fused).
// state buffers as mx_er_fwd (prefix and sum_gt zeroed by the caller); nothing of size H*W is allocated.
int mx_er_lr_fwd(const float* cam, const float* sgc, const float* lwb, int N, int h, int w, int L, int K, int H, int W, long k,
                 const int* k_dev, unsigned* krem, unsigned* prefix, unsigned long long* sum_gt, unsigned* cnt_eq, unsigned* hcnt,
                 unsigned long long* hsum, float* vals, float* loss, void* stream) {
  MX_CHECK_ARG(cam && sgc && lwb && krem && prefix && sum_gt && cnt_eq && hcnt && hsum && loss, "er_lr_fwd: null pointer");
  MX_CHECK_ARG(N > 0 && h > 0 && w > 0 && K >= 2 && K <= KMAX && K <= L && L % 4 == 0 && H > 0 && W > 0, "er_lr_fwd: bad extents (L must be a multiple of 4)");
  MX_CHECK_ARG((((uintptr_t)cam | (uintptr_t)sgc) & 15) == 0, "er_lr_fwd: maps must be 16-byte aligned");
  MX_CHECK_ARG(k_dev || (k >= 1 && k <= (long)K * H * W), "er_lr_fwd: k=%ld out of range for rows of %ld (torch.topk would raise)", k, (long)K * H * W);
  hipStream_t st = (hipStream_t)stream;
  if (k_dev) hipLaunchKernelGGL(er_krem_init_kernel, dim3(1), dim3(64), 0, st, krem, N, k_dev);
  else hipMemsetD32Async((hipDeviceptr_t)krem, (int)(unsigned)k, N, st);
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
  const unsigned himask[3] = {0u, 0xFFE00000u, 0xFFFFFC00u};
  long HW = (long)H * W;
  static const int chunk_px = getenv("MX_ER_CHUNK") ? atoi(getenv("MX_ER_CHUNK")) : 1024;   // pixels per workgroup; 4096 / 2048 / 1024 / 512: 0.81 / 0.69 / 0.65 / 0.69 ms (N=16), 1.25 / 1.17 / 1.16 / 1.27 (N=32)
  int chunks = (int)((HW + chunk_px - 1) / chunk_px);
  if (chunks > 256) chunks = 256;
  for (int ps = 0; ps < 3; ++ps) {
    hipMemsetAsync(hcnt, 0, sizeof(unsigned) * N * RBINS, st);
    hipMemsetAsync(hsum, 0, sizeof(er_fix_t) * N * RBINS, st);
    if (ps == 0 || !vals)
      hipLaunchKernelGGL(er_lr_hist_kernel, dim3(chunks, N), dim3(256), 0, st, cam, sgc, lwb, h, w, L, K, H, W, shifts[ps], bits[ps],
                         himask[ps], prefix, hcnt, hsum, ps == 0 ? vals : (float*)nullptr);
    else
      hipLaunchKernelGGL(er_vals_hist_kernel, dim3(chunks, N), dim3(256), 0, st, (const float*)vals, lwb, K, HW, shifts[ps], bits[ps],
                         himask[ps], prefix, hcnt, hsum);
    hipLaunchKernelGGL(er_scan_kernel, dim3(N), dim3(64), 0, st, hcnt, hsum, shifts[ps], bits[ps], krem, prefix, sum_gt,
                       cnt_eq, N);
  }
  hipLaunchKernelGGL(er_final_kernel, dim3(1), dim3(64), 0, st, krem, prefix, sum_gt, N, 1.0f / ((float)N * (float)(k > 0 ? k : 1)), loss, k_dev);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// bytes of scratch mx_er_lr_bwd needs (64-bit accumulators of the 21-class band kernel; 0 for other class counts)
long mx_er_lr_bwd_ws(int N, int h, int w, int L, int K) {
  if (N <= 0 || h <= 0 || w <= 0 || L <= 0) return MX_EARG;
  return K == 21 ? (long)N * h * w * L * 8 : 0;
}

int mx_er_lr_bwd(const float* cam, const float* sgc, const float* lwb, const unsigned* prefix, const unsigned* krem,
                 const unsigned* cnt_eq, const float* gup, float gscale, const int* k_dev, float* gsgc, int N, int h, int w, int L,
                 int K, int H, int W, void* ws, long ws_bytes, void* stream) {
  MX_CHECK_ARG(cam && sgc && lwb && prefix && krem && cnt_eq && gsgc, "er_lr_bwd: null pointer");
  MX_CHECK_ARG(N > 0 && h > 0 && w > 0 && K >= 2 && K <= KMAX && K <= L && L % 4 == 0 && L <= 256 && H > 0 && W > 0, "er_lr_bwd: bad extents (L must be a multiple of 4)");
  MX_CHECK_ARG((((uintptr_t)cam | (uintptr_t)sgc) & 15) == 0, "er_lr_bwd: maps must be 16-byte aligned");
  if (K == 21) {
    const long total = (long)N * h * w * L;
    MX_CHECK_ARG(ws && ws_bytes >= total * 8 && ((uintptr_t)ws & 7) == 0, "er_lr_bwd: %ld bytes of scratch required (mx_er_lr_bwd_ws)", total * 8);
    // rows per band: the largest count that keeps a band within one low-res row spacing (at most three low-res rows touched)
    int ty = (h > 1) ? (H - 1) / (h - 1) : H;
    if (ty < 1) ty = 1;
    if (ty > 32) ty = 32;
    const double sx = (W > 1) ? (double)(w - 1) / (double)(W - 1) : 0.0;   // widest low-res column span of a 256-column segment
    int xe0 = (int)(sx * 255.0) + 3;
    if (xe0 > w) xe0 = w;
    const size_t sh = (size_t)3 * (xe0 + 1) * 21 * sizeof(unsigned long long);
    MX_CHECK_ARG(sh <= 64 * 1024, "er_lr_bwd: low-res span per segment too wide (%d columns)", xe0);
    hipMemsetAsync(ws, 0, (size_t)total * 8, (hipStream_t)stream);
    hipLaunchKernelGGL(er_lr_bwd_band_kernel<21>, dim3(cdiv(W, 256), cdiv(H, ty), N), dim3(256), sh, (hipStream_t)stream, cam, sgc,
                       lwb, prefix, krem, cnt_eq, (unsigned long long*)ws, h, w, L, H, W, ty);
    long nb = (total + 255) / 256;
    hipLaunchKernelGGL(er_lr_bwd_finish_kernel, dim3((int)(nb < 2048 ? nb : 2048)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)ws, gup, gscale, k_dev, N, gsgc, total);
  } else {
    hipLaunchKernelGGL(er_lr_bwd_kernel, dim3(h * w, N), dim3(256), 0, (hipStream_t)stream, cam, sgc, lwb, prefix, krem, cnt_eq, gup,
                       gscale, k_dev, gsgc, h, w, L, K, H, W);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
