// Input stage of the training loop (reference: src/data.py:215-332, src/imutils.py:143-181,
// :376-388): what the reference's DataLoader workers do per image in numpy - color_norm ((x/255 - mean)/std in float64),
// RandomCrop's placement into a zero container, HWC -> CHW, and the .float() cast of train_mcl.py:163-165 - as one
// batched kernel over uint8 crops that the host only has to decode, flip / resize with PIL and cut.  The bytes crossing
// PCIe are uint8 (0.9 MB per image instead of 2.4 MB fp32 + 2 x 1.2 MB fp64) and the fp32 tensors are written once,
// straight into the batch tensors the loop body reads.  Bit-exact with the numpy expressions: same fp64 operations in
// the same order, one rounding to fp32.
#include "common.h"

struct InputJob { int src_off, sh, sw, top, left, erase_yx, erase_hw, sstride; };   // uint8 HWC crop [sh, sw, 3] placed at (top, left);
// erase_yx = y | x << 16, erase_hw = h | w << 16 of RandomErasing's box in the OUTPUT tensor (0 = none);
// sstride = row stride of the source in pixels (0 = sw: a packed crop; > sw: a crop inside a larger, jittered image)

__global__ __launch_bounds__(256) void input_stage_kernel(const unsigned char* __restrict__ src, const InputJob* __restrict__ jobs,
                                                          float* __restrict__ dst, int Hd, int Wd) {
  const InputJob jb = jobs[blockIdx.y];
  const long plane = (long)Hd * Wd;
  float* out = dst + (long)blockIdx.y * 3 * plane;
  const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
  const int ey = jb.erase_yx & 0xffff, ex = (jb.erase_yx >> 16) & 0xffff, eh = jb.erase_hw & 0xffff, ew = (jb.erase_hw >> 16) & 0xffff;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < plane; p += (long)gridDim.x * 256) {
    const int y = (int)(p / Wd), x = (int)(p - (long)y * Wd);
    const int sy = y - jb.top, sx = x - jb.left;
    float v[3] = {0.f, 0.f, 0.f};                         // RandomCrop's container is zero outside the pasted crop
    if (sy >= 0 && sy < jb.sh && sx >= 0 && sx < jb.sw) {
      const unsigned char* px = src + jb.src_off + ((long)sy * (jb.sstride > 0 ? jb.sstride : jb.sw) + sx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)(((double)px[c] / 255.0 - mean[c]) / stdv[c]);   // imutils.py:383-388
    }
    if (y >= ey && y < ey + eh && x >= ex && x < ex + ew) v[0] = v[1] = v[2] = 0.f;   // RandomErasing(value=0), train_mcl.py:114
    out[p] = v[0]; out[plane + p] = v[1]; out[2 * plane + p] = v[2];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ColorJitter on the device: torchvision's PIL backend (functional_pil.adjust_brightness / _contrast / _saturation / _hue),
// i.e. Pillow's C routines, restated bit for bit on uint8 RGB images in HBM.  What Pillow computes (checked against
// Pillow 12.2 itself: all 16.7 M colours through both HSV conversions, the three blends on images and noise):
//   blend(d, x, alpha)   ImagingBlend: t = (float)d + alpha * (float)(x - d) in SINGLE precision, mul and add rounded
//                        separately; 0 <= alpha <= 1: (UINT8)t, else clamp to [0, 255] then truncate
//   brightness           blend(0, x, f);   saturation  blend(L(x), x, f), L = (19595 r + 38470 g + 7471 b + 0x8000) >> 16
//   contrast             blend(m, x, f), m = int(sum(L) / count + 0.5) over the WHOLE image (double division)
//   hue                  rgb2hsv_row / hsv2rgb of Convert.c ("following colorsys.py": float values, double constants),
//                        H += shift (uint8 wrap)
// One job = one image; an image's four adjustments run in its own random order: launch `pos` applies order[pos] of every
// image in place (each Pillow call returns a uint8 image, so rounding to uint8 between the steps is part of the result).
struct JitterJob { int off, h, w, order; float fb, fc, fs; int hue_add; };     // order: 4 nibbles, 0 b, 1 c, 2 s, 3 h, 15 none

__device__ __forceinline__ int jit_L(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// HIP's __fmul_rn / __fadd_rn are plain * and + in the headers and hipcc contracts them into one FMA (also under
// `#pragma clang fp contract(off)`: measured), whose single rounding differs from Pillow's two in a few thousand pixels per
// image (128 + 1.7f * -70 = 8.9999967 fused, 9.0 unfused).  An empty asm pins the rounded product in a register.
__device__ __forceinline__ float jit_pin(float v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ double jit_pin(double v) { asm volatile("" : "+v"(v)); return v; }

__device__ __forceinline__ unsigned char jit_blend(int d, int x, float alpha, bool inter) {
  const float t = (float)d + jit_pin(alpha * (float)(x - d));
  if (inter) return (unsigned char)t;
  if (t <= 0.0f) return 0;
  if (t >= 255.0f) return 255;
  return (unsigned char)t;
}

__device__ __forceinline__ void jit_hue(unsigned char& r8, unsigned char& g8, unsigned char& b8, int add) {
  // rgb2hsv_row
  const int r = r8, g = g8, b = b8;
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = __fsub_rn(bc, gc);
    else if (g == maxc) h = (float)__dsub_rn(__dadd_rn(2.0, (double)rc), (double)bc);
    else h = (float)__dsub_rn(__dadd_rn(4.0, (double)gc), (double)rc);
    h = (float)fmod(__dadd_rn(__ddiv_rn((double)h, 6.0), 1.0), 1.0);
    uh = min(255, max(0, (int)__dmul_rn((double)h, 255.0)));
    us = min(255, max(0, (int)__dmul_rn((double)s, 255.0)));
  }
  uh = (uh + add) & 0xFF;                                   // np_h += np.uint8(hue_factor * 255), uint8 wrap
  // hsv2rgb
  if (us == 0) { r8 = g8 = b8 = (unsigned char)uv; return; }
  const double x6 = __ddiv_rn(__dmul_rn((double)(float)uh, 6.0), 255.0);
  const int i = (int)floor(x6);
  const float f = (float)__dsub_rn(x6, (double)(float)i);
  const float fs = (float)__ddiv_rn((double)(float)us, 255.0);
  const double vf = (double)(float)uv;
  const int p = (int)round(vf * jit_pin(1.0 - (double)fs));
  const int q = (int)round(vf * jit_pin(1.0 - jit_pin((double)fs * (double)f)));
  const int t = (int)round(vf * jit_pin(1.0 - jit_pin((double)fs * jit_pin(1.0 - (double)f))));
  const unsigned char up = (unsigned char)min(255, max(0, p)), uq = (unsigned char)min(255, max(0, q)), ut = (unsigned char)min(255, max(0, t));
  const unsigned char v = (unsigned char)uv;
  switch (i % 6) {
    case 0: r8 = v; g8 = ut; b8 = up; break;
    case 1: r8 = uq; g8 = v; b8 = up; break;
    case 2: r8 = up; g8 = v; b8 = ut; break;
    case 3: r8 = up; g8 = uq; b8 = v; break;
    case 4: r8 = ut; g8 = up; b8 = v; break;
    default: r8 = v; g8 = up; b8 = uq; break;
  }
}

// sums[job] += sum of L over the image, for the jobs whose adjustment at `pos` is contrast (exact: 64-bit integers)
__global__ __launch_bounds__(256) void jitter_lsum_kernel(const unsigned char* __restrict__ src, const JitterJob* __restrict__ jobs,
                                                          unsigned long long* __restrict__ sums, int pos) {
  const JitterJob jb = jobs[blockIdx.y];
  if (((jb.order >> (4 * pos)) & 15) != 1) return;
  __shared__ unsigned long long part[4];
  const long n = (long)jb.h * jb.w;
  unsigned long long acc = 0;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    const unsigned char* px = src + jb.off + p * 3;
    acc += (unsigned long long)jit_L(px[0], px[1], px[2]);
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sums + blockIdx.y, part[0] + part[1] + part[2] + part[3]);
}

__global__ __launch_bounds__(256) void jitter_apply_kernel(unsigned char* __restrict__ src, const JitterJob* __restrict__ jobs,
                                                           const unsigned long long* __restrict__ sums, int pos) {
  const JitterJob jb = jobs[blockIdx.y];
  const int op = (jb.order >> (4 * pos)) & 15;
  if (op > 3) return;
  const long n = (long)jb.h * jb.w;
  const float alpha = op == 0 ? jb.fb : op == 1 ? jb.fc : jb.fs;
  const bool inter = alpha >= 0.0f && alpha <= 1.0f;
  int mean = 0;
  if (op == 1) mean = (int)(__dadd_rn(__ddiv_rn((double)sums[blockIdx.y], (double)n), 0.5));   // int(stat.mean[0] + 0.5)
  for (long p = blockIdx.x * 256L + threadIdx.x; p < n; p += (long)gridDim.x * 256) {
    unsigned char* px = src + jb.off + p * 3;
    unsigned char r = px[0], g = px[1], b = px[2];
    if (op == 0) { r = jit_blend(0, r, alpha, inter); g = jit_blend(0, g, alpha, inter); b = jit_blend(0, b, alpha, inter); }
    else if (op == 1) { r = jit_blend(mean, r, alpha, inter); g = jit_blend(mean, g, alpha, inter); b = jit_blend(mean, b, alpha, inter); }
    else if (op == 2) { const int L = jit_L(r, g, b); r = jit_blend(L, r, alpha, inter); g = jit_blend(L, g, alpha, inter); b = jit_blend(L, b, alpha, inter); }
    else jit_hue(r, g, b, jb.hue_add);
    px[0] = r; px[1] = g; px[2] = b;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// PIL.Image.resize (RandomResizeLong's bicubic, get_views' bilinear 448 x 448; imutils.py:127-141, data.py:274-276) on the
// device: Pillow's ImagingResample for 8-bit images is integer arithmetic - per output column / row a window [xmin, xmin +
// n) of the input and n coefficients in 22-bit fixed point (computed on the host in double exactly as precompute_coeffs +
// normalize_coeffs_8bpc do), ss = 2^21 + sum in[x] * k[x], out = clip8(ss >> 22) - horizontal pass into a temporary 8-bit
// image, then the vertical pass.  Bit-exact with Pillow (tests/test_input_path.py).
// Job: {src_off, Hin, Win, tmp_off, dst_off, Wout, Hout, tab_off}; table at tab_off (int32 words): ksize_h, ksize_v,
// bounds_h[Wout][2], kk_h[Wout][ksize_h], bounds_v[Hout][2], kk_v[Hout][ksize_v].
struct ResampleJob { int src_off, hin, win, tmp_off, dst_off, wout, hout, tab_off; };

__device__ __forceinline__ unsigned char rs_clip8(int v) { v >>= 22; return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void resample_h_kernel(const unsigned char* __restrict__ src, const ResampleJob* __restrict__ jobs,
                                                         const int* __restrict__ tabs, unsigned char* __restrict__ tmp) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int* tab = tabs + jb.tab_off;
  const int ks = tab[0];
  const int* bounds = tab + 2;
  const int* kk = bounds + 2 * jb.wout;
  const long total = (long)jb.hin * jb.wout;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const int y = (int)(p / jb.wout), xx = (int)(p - (long)y * jb.wout);
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ks;
    const unsigned char* in = src + jb.src_off + ((long)y * jb.win + xmin) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int x = 0; x < n; ++x) { const int c = k[x]; s0 += in[3 * x] * c; s1 += in[3 * x + 1] * c; s2 += in[3 * x + 2] * c; }
    unsigned char* o = tmp + jb.tmp_off + p * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}

__global__ __launch_bounds__(256) void resample_v_kernel(const unsigned char* __restrict__ tmp, const ResampleJob* __restrict__ jobs,
                                                         const int* __restrict__ tabs, unsigned char* __restrict__ dst) {
  const ResampleJob jb = jobs[blockIdx.y];
  const int* tab = tabs + jb.tab_off;
  const int ksh = tab[0], ks = tab[1];
  const int* bounds = tab + 2 + 2 * jb.wout + (long)jb.wout * ksh;
  const int* kk = bounds + 2 * jb.hout;
  const long total = (long)jb.hout * jb.wout;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < total; p += (long)gridDim.x * 256) {
    const int yy = (int)(p / jb.wout), x = (int)(p - (long)yy * jb.wout);
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ks;
    const unsigned char* in = tmp + jb.tmp_off + ((long)ymin * jb.wout + x) * 3;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    for (int y = 0; y < n; ++y) { const int c = k[y]; const unsigned char* q = in + (long)y * jb.wout * 3; s0 += q[0] * c; s1 += q[1] * c; s2 += q[2] * c; }
    unsigned char* o = dst + jb.dst_off + p * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}

extern "C" {

// dst[n, 3, Hd, Wd] (fp32, fully written) <- color_norm(src crop n) placed at (top, left), zeros elsewhere.
// src: packed uint8 HWC crops; jobs: n x 8 int32 {src_off, sh, sw, top, left, erase y|x<<16, erase h|w<<16, source row stride or 0}; both on the device.
int mx_input_stage(const unsigned char* src, const int* jobs, float* dst, int n, int Hd, int Wd, void* stream) {
  MX_CHECK_ARG(src && jobs && dst, "input_stage: null pointer");
  MX_CHECK_ARG(n > 0 && Hd > 0 && Wd > 0, "input_stage: bad extents n=%d Hd=%d Wd=%d", n, Hd, Wd);
  const long plane = (long)Hd * Wd;
  int bx = cdiv(plane, 256);
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(input_stage_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, src, (const InputJob*)jobs, dst, Hd, Wd);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// ColorJitter (torchvision PIL backend = Pillow's C routines) in place on n uint8 HWC images inside `src`.
// jobs: n x 8 int32/float32 {byte offset, h, w, order (nibble per position: 0 brightness, 1 contrast, 2 saturation, 3 hue,
// 15 none), brightness, contrast, saturation factor (float), hue shift 0..255}; sums: n uint64 scratch.  Nine launches.
int mx_color_jitter(unsigned char* src, const int* jobs, unsigned long long* sums, int n, int max_pixels, void* stream) {
  MX_CHECK_ARG(src && jobs && sums && n > 0 && max_pixels > 0, "color_jitter: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int bx = cdiv(max_pixels, 256 * 8);
  if (bx < 1) bx = 1;
  if (bx > 256) bx = 256;
  for (int pos = 0; pos < 4; ++pos) {
    hipMemsetAsync(sums, 0, sizeof(unsigned long long) * n, st);
    hipLaunchKernelGGL(jitter_lsum_kernel, dim3(bx, n), dim3(256), 0, st, src, (const JitterJob*)jobs, sums, pos);
    hipLaunchKernelGGL(jitter_apply_kernel, dim3(bx, n), dim3(256), 0, st, src, (const JitterJob*)jobs, sums, pos);
  }
  MX_LAUNCH_CHECK();
  return MX_OK;
}

// Pillow's 8-bit two-pass resample of n uint8 HWC images: src -> (horizontal) tmp -> (vertical) dst; jobs n x 8 int32
// {src_off, Hin, Win, tmp_off, dst_off, Wout, Hout, tab_off (words)}; tabs: the coefficient tables (see above).
int mx_resample(const unsigned char* src, const int* jobs, const int* tabs, unsigned char* tmp, unsigned char* dst, int n,
                int max_pixels, void* stream) {
  MX_CHECK_ARG(src && jobs && tabs && tmp && dst && n > 0 && max_pixels > 0, "resample: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  int bx = cdiv(max_pixels, 256 * 4);
  if (bx < 1) bx = 1;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(resample_h_kernel, dim3(bx, n), dim3(256), 0, st, src, (const ResampleJob*)jobs, tabs, tmp);
  hipLaunchKernelGGL(resample_v_kernel, dim3(bx, n), dim3(256), 0, st, (const unsigned char*)tmp, (const ResampleJob*)jobs, tabs, dst);
  MX_LAUNCH_CHECK();
  return MX_OK;
}

}  // extern "C"
