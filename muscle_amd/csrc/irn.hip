// IRN random-walk propagation support kernels (src/indexing.py:77-142 as used by infer_irn.py:76).
// The matrix powers and the final cam x transition product run on the MFMA GEMM (mx_bgemm); this file builds the dense
// affinity matrix straight from the edge map (the reference goes through flat index tables, index_select, max_pool2d,
// a sparse COO tensor assembled on the CPU and .to_dense().cuda()) and turns it into the column-stochastic matrix.
#include "common.h"
#include <hip/hip_fp16.h>

// padded edge map of propagate_to_edge (:124): radius columns left/right and radius rows at the bottom, value 1.0
__device__ __forceinline__ float irn_edge_padded(const float* edge, int h, int w, int radius, int y, int x) {
  const int xi = x - radius;
  return (y >= 0 && y < h && xi >= 0 && xi < w) ? edge[y * w + xi] : 1.0f;
}

// dense[f][t] = dense[t][f] = 1 - max over the straight path f -> t of the padded edge map, for every source pixel f and
// every search direction whose destination t lies inside the image (edge_to_affinity :77-93 + affinity_sparse2dense
// :96-113 + the crop :131-133).  The diagonal is 1 (indices_id, :106).  dense is n4 x n4 with leading dimension ld,
// zero-filled by the caller; rows/columns n..n4-1 are isolated vertices with a unit diagonal.
__global__ __launch_bounds__(256) void irn_affinity_kernel(const float* edge, int h, int w, int radius, const int* pcoord,
                                                           const int* poff, const int* plen, int nd, float* dense, int ld,
                                                           int n4) {
  const int n = h * w;
  const long total = (long)n * nd;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total + n4; i += (long)gridDim.x * 256) {
    if (i >= total) {                       // diagonal
      const long v = i - total;
      dense[v * ld + v] = 1.0f;
      continue;
    }
    const int d = (int)(i / n), f = (int)(i - (long)d * n);
    const int fy = f / w, fx = f - fy * w;
    const int* pc = pcoord + 2 * poff[d];
    const int ty = fy + pc[0], tx = fx + pc[1];           // destination = first (farthest) path pixel
    if (ty < 0 || ty >= h || tx < 0 || tx >= w) continue;
    float mx = -INFINITY;
    for (int l = 0; l < plen[d]; ++l) mx = fmaxf(mx, irn_edge_padded(edge, h, w, radius, fy + pc[2 * l], fx + radius + pc[2 * l + 1]));
    const float aff = 1.0f - mx;
    const int t = ty * w + tx;
    dense[(long)f * ld + t] = aff;
    dense[(long)t * ld + f] = aff;
  }
}

// in place: dense = dense^beta; colsum[j] = sum_i dense[i][j]     (to_transition_matrix :116-118)
// one workgroup = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void irn_pow_colsum_kernel(float* dense, int n4, int ld, float beta, float* colsum) {
  __shared__ float sh[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + cx;
  const int ib = (beta > 0.f && beta <= 64.f && beta == floorf(beta)) ? (int)beta : 0;
  float s = 0.f;
  if (j < n4) {
    for (int i = ry; i < n4; i += 4) {
      float v = dense[(long)i * ld + j];
      if (ib > 0) {                          // integral exponent (the script's beta=10): exact repeated multiplication
        float r = 1.f, b = v;
        for (int e = ib; e; e >>= 1) { if (e & 1) r *= b; b *= b; }
        v = r;
      } else {
        v = (v == 0.f) ? 0.f : powf(v, beta);
      }
      dense[(long)i * ld + j] = v;
      s += v;
    }
  }
  sh[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && j < n4) colsum[j] = sh[0][cx] + sh[1][cx] + sh[2][cx] + sh[3][cx];
}

__global__ __launch_bounds__(256) void irn_col_scale_kernel(float* dense, int n4, int ld, const float* colsum) {
  const long total = (long)n4 * n4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / n4;
    const int j = (int)(i - r * n4);
    dense[r * ld + j] = dense[r * ld + j] / colsum[j];
  }
}


// infer_irn.py:78-94: rw_up = interpolate(rw, x4, bilinear, align_corners=False)[:, :H, :W]; rw_up /= max(rw_up);
// label = argmax over [bg_thres, rw_up_1 .. rw_up_C] (first maximum wins).
__device__ __forceinline__ float irn_up4(const float* m, int h, int w, int Y, int X) {
  float sy = ((float)Y + 0.5f) * 0.25f - 0.5f, sx = ((float)X + 0.5f) * 0.25f - 0.5f;
  if (sy < 0.f) sy = 0.f;
  if (sx < 0.f) sx = 0.f;
  int y0 = (int)sy, x0 = (int)sx;
  if (y0 > h - 1) y0 = h - 1;
  if (x0 > w - 1) x0 = w - 1;
  const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  const float wy = sy - y0, wx = sx - x0;
  return (1.f - wy) * ((1.f - wx) * m[y0 * w + x0] + wx * m[y0 * w + x1]) + wy * ((1.f - wx) * m[y1 * w + x0] + wx * m[y1 * w + x1]);
}

__global__ __launch_bounds__(256) void irn_up_max_kernel(const float* rw, int C, int h, int w, int H, int W, unsigned* mx_bits) {
  float m = 0.f;
  const long total = (long)C * H * W;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int X = (int)(i % W), Y = (int)((i / W) % H), c = (int)(i / ((long)W * H));
    m = fmaxf(m, irn_up4(rw + (long)c * h * w, h, w, Y, X));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(mx_bits, __float_as_uint(m));      // values are >= 0: bit order == value order
}

__global__ __launch_bounds__(256) void irn_label_kernel(const float* rw, int C, int h, int w, int H, int W, const unsigned* mx_bits,
                                                        float bg_thres, unsigned char* label, __half* soft /*[H,W,C+1] or null*/) {
  const float mx = __uint_as_float(mx_bits[0]);
  const long total = (long)H * W;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const int X = (int)(p % W), Y = (int)(p / W);
    float best = bg_thres;
    int bk = 0;
    if (soft) soft[p * (C + 1)] = __float2half_rn(bg_thres);
    for (int c = 0; c < C; ++c) {
      const float v = irn_up4(rw + (long)c * h * w, h, w, Y, X) / mx;
      if (soft) soft[p * (C + 1) + c + 1] = __float2half_rn(v);
      if (v > best) { best = v; bk = c + 1; }
    }
    label[p] = (unsigned char)bk;
  }
}

extern "C" {

int mx_irn_affinity(const float* edge, int h, int w, int radius, const int* pcoord, const int* poff, const int* plen, int nd,
                    float* dense, int ld, int n4, void* stream) {
  MX_CHECK_ARG(edge && pcoord && poff && plen && dense && h > 0 && w > 0 && radius > 0 && nd > 0, "irn_affinity: bad args");
  MX_CHECK_ARG(n4 >= h * w && ld >= n4, "irn_affinity: dense must be at least h*w square (n4=%d, ld=%d)", n4, ld);
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(dense, 0, sizeof(float) * (size_t)n4 * ld, st);
  long total = (long)h * w * nd + n4;
  long blocks = (total + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(irn_affinity_kernel, dim3((unsigned)blocks), dim3(256), 0, st, edge, h, w, radius, pcoord, poff, plen, nd, dense,
                     ld, n4);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_irn_transition(float* dense, int n4, int ld, float beta, float* colsum, void* stream) {
  MX_CHECK_ARG(dense && colsum && n4 > 0 && ld >= n4, "irn_transition: bad args");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(irn_pow_colsum_kernel, dim3(cdiv(n4, 64)), dim3(256), 0, st, dense, n4, ld, beta, colsum);
  long blocks = ((long)n4 * n4 + 255) / 256;
  if (blocks > 65535) blocks = 65535;
  hipLaunchKernelGGL(irn_col_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dense, n4, ld, colsum);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

int mx_irn_finish(const float* rw, int C, int h, int w, int H, int W, float bg_thres, unsigned* max_scratch, unsigned char* label,
                  void* soft_half, void* stream) {
  MX_CHECK_ARG(rw && max_scratch && label && C > 0 && C < 255 && h > 0 && w > 0 && H > 0 && W > 0 && H <= 4 * h && W <= 4 * w,
               "irn_finish: bad args (the label map is the top-left HxW crop of the 4x upsampled maps)");
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(max_scratch, 0, sizeof(unsigned), st);
  long b1 = ((long)C * H * W + 255) / 256, b2 = ((long)H * W + 255) / 256;
  if (b1 > 4096) b1 = 4096;
  if (b2 > 4096) b2 = 4096;
  hipLaunchKernelGGL(irn_up_max_kernel, dim3((unsigned)b1), dim3(256), 0, st, rw, C, h, w, H, W, max_scratch);
  hipLaunchKernelGGL(irn_label_kernel, dim3((unsigned)b2), dim3(256), 0, st, rw, C, h, w, H, W, max_scratch, bg_thres, label,
                     (__half*)soft_half);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
