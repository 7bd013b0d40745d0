// Input stage of the training loop (reference: src/data.py:215-332, src/imutils.py:143-181,
// :376-388): what the reference's DataLoader workers do per image in numpy - color_norm ((x/255 - mean)/std in float64),
// RandomCrop's placement into a zero container, HWC -> CHW, and the .float() cast of train_mcl.py:163-165 - as one
// batched kernel over uint8 crops that the host only has to decode, flip / resize with PIL and cut.  The bytes crossing
// PCIe are uint8 (0.9 MB per image instead of 2.4 MB fp32 + 2 x 1.2 MB fp64) and the fp32 tensors are written once,
// straight into the batch tensors the loop body reads.  Bit-exact with the numpy expressions: same fp64 operations in
// the same order, one rounding to fp32.
#include "common.h"

struct InputJob { int src_off, sh, sw, top, left, erase_yx, erase_hw, pad2; };   // uint8 HWC crop [sh, sw, 3] placed at (top, left);
// erase_yx = y | x << 16, erase_hw = h | w << 16 of RandomErasing's box in the OUTPUT tensor (0 = none)

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
      const unsigned char* px = src + jb.src_off + ((long)sy * jb.sw + sx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) v[c] = (float)(((double)px[c] / 255.0 - mean[c]) / stdv[c]);   // imutils.py:383-388
    }
    if (y >= ey && y < ey + eh && x >= ex && x < ex + ew) v[0] = v[1] = v[2] = 0.f;   // RandomErasing(value=0), train_mcl.py:114
    out[p] = v[0]; out[plane + p] = v[1]; out[2 * plane + p] = v[2];
  }
}

extern "C" {

// dst[n, 3, Hd, Wd] (fp32, fully written) <- color_norm(src crop n) placed at (top, left), zeros elsewhere.
// src: packed uint8 HWC crops; jobs: n x 8 int32 {src_off, sh, sw, top, left, erase y|x<<16, erase h|w<<16, 0}; both on the device.
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

}  // extern "C"
