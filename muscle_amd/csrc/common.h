// Shared device/host helpers for libmuscle_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MX_OK 0
#define MX_EARG (-1)

extern "C" const char* mx_last_error(void);
void mx_set_error(const char* fmt, ...);

#define MX_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      mx_set_error(__VA_ARGS__);           \
      return MX_EARG;                      \
    }                                      \
  } while (0)

#define MX_LAUNCH_CHECK()                                     \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) {                                  \
      mx_set_error("launch failed: %s", hipGetErrorString(e__)); \
      return (int)e__;                                        \
    }                                                         \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// dW[e] += sum_g part[g][e] in a fixed order (wgrad.hip); n = elements per partial matrix, a multiple of 4
void mx_launch_parts_reduce(const float* part, int groups, int n, float* dW, hipStream_t st);

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
// v_exp_f32 + v_rcp_f32 (both <= 1 ulp): ~1e-7 relative, far inside the stated fp32 tolerance, and ~4x fewer
// instructions than the IEEE-exact division sequence hipcc emits for 1.0f / x
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
// d/dx [x*sigmoid(x)] = s * (1 + x * (1 - s))   (src/efficientnet_pytorch/utils.py:44-47)
__device__ __forceinline__ float swish_gradf_(float x) {
  float s = sigmoidf_(x);
  return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------
// Ordered cross-workgroup reductions without a second launch ("last arriver finishes").
// Every workgroup of a group stores its partial result to global scratch with mx_st_wt / mx_st4_wt, then calls
// mx_last_arriver(): exactly one caller per group - the one that arrives last - gets true; it then reads ALL partials of
// the group (plain loads) and adds them in index order, so the result does not depend on which workgroup came last
// (fp32 atomics gave sums whose last bits moved from run to run).
// Visibility on gfx950 (8 XCDs with private L2s, per-CU L1s never refreshed by other CUs; cdna_hip_programming.md,
// Guideline 16, counter form): the partials are stored WRITE-THROUGH (sc1), every storing wave drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds to the counter with an agent-scope atomic, and the
// lane that sees the last ticket runs ONE agent-scope acquire (drops this CU's stale L1 lines) before the workgroup's plain
// loads.  No release fence anywhere: __threadfence() in every thread (buffer_wbl2 + buffer_inv per wave) made these
// HBM-bound reductions 3x slower.
// `counter` must be 0 before the launch and is 0 again afterwards (the last arriver resets it).
// Scratch layout used by the entry points that take `ws`: [MX_WS_COUNTER_BYTES of counters][partials].
// ---------------------------------------------------------------------------
#define MX_WS_COUNTER_BYTES 65536
#define MX_WS_COUNTERS (MX_WS_COUNTER_BYTES / 4)

__device__ __forceinline__ void mx_st_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mx_st4_wt(float* p, float4 v) {
  mx_st_wt(p, v.x); mx_st_wt(p + 1, v.y); mx_st_wt(p + 2, v.z); mx_st_wt(p + 3, v.w);
}

__device__ __forceinline__ bool mx_last_arriver(unsigned* counter, unsigned total, unsigned* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its write-through partials have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned last = (prev == total - 1u) ? 1u : 0u;
    if (last) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // everyone has arrived: ready for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier lets the others load
    }
    *lds_flag = last;
  }
  __syncthreads();
  return *lds_flag != 0u;
}

// BatchNorm backward finalisation of one channel from its two sums (shared by bn.hip and by the producers that finish their own
// statistics, dwconv.hip): dgamma / dbeta (+=) and the coefficients of dX = c1*g + c2*X + c3.
struct BnBwdFin {
  double count; const float* gamma; const float* mean; const float* rstd; int training;
  float* dgamma; float* dbeta; float* c1; float* c2; float* c3;
};

__device__ __forceinline__ void bn_bwd_finalize_one(int c, double sg, double sgx, double count, const float* gamma,
                                                    const float* mean, const float* rstd, int training, float* dgamma,
                                                    float* dbeta, float* c1, float* c2, float* c3) {
  double m = mean[c], r = rstd[c], gm = gamma[c];
  double dg = r * (sgx - m * sg);
  dgamma[c] += (float)dg;
  dbeta[c] += (float)sg;
  if (training) {
    double k = gm * r * r * (dg / count);
    c1[c] = (float)(gm * r);
    c2[c] = (float)(-k);
    c3[c] = (float)(-gm * r * (sg / count) + k * m);
  } else {
    c1[c] = (float)(gm * r);
    c2[c] = 0.f;
    c3[c] = 0.f;
  }
}

// The reduction of bn_reduce_finalize_kernel<true> (bn.hip) for 32 channels starting at c0, by ONE workgroup of 256 threads: 8 row
// lanes walk the P partial rows part[P][2][C] in the same order and association, so a producer that finishes its own statistics
// (the last workgroup to arrive) leaves the same bits as the separate launch did.  `sh` = 2 x 8 x 32 doubles of LDS.
// One row lane's share of the partial rows of a channel: rows rl, rl + 8, ... in batches of four (the summation order every BatchNorm
// finalisation has used since round 2: ((a0 + a1) + (a2 + a3)) per batch, batches in ascending order, fp64).  U batches are REQUESTED before
// the first is added - the kernels that call this are one short dependent chain of loads each (7-9 us for 196 rows at one batch per round
// trip, 244 launches per B7 step), and the order of the additions, hence every bit of the result, does not depend on U.
template <int U>
__device__ __forceinline__ void bn_parts_batches(const float* part, int P, int C, int c, int& p, double& s0, double& s1) {
  for (; p + 24 + 32 * (U - 1) < P; p += 32 * U) {
    float a[U][4], b[U][4];
    const unsigned o = (unsigned)p * 2u * (unsigned)C + (unsigned)c;          // (the partial rows are far below 2^31 bytes: P <= 1024 rows of 2C)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned q = o + (unsigned)(32 * u + 8 * k) * 2u * (unsigned)C;
        a[u][k] = part[q]; b[u][k] = part[q + (unsigned)C];
      }
    __builtin_amdgcn_sched_barrier(0);                 // every request before the first addition (the scheduler interleaves them otherwise)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s0 += ((double)a[u][0] + (double)a[u][1]) + ((double)a[u][2] + (double)a[u][3]);
      s1 += ((double)b[u][0] + (double)b[u][1]) + ((double)b[u][2] + (double)b[u][3]);
    }
  }
}
__device__ __forceinline__ void bn_parts_lane_sum(const float* part, int P, int C, int c, int rl, double& s0, double& s1) {
  int p = rl;
  bn_parts_batches<6>(part, P, C, c, p, s0, s1);
  bn_parts_batches<2>(part, P, C, c, p, s0, s1);
  bn_parts_batches<1>(part, P, C, c, p, s0, s1);
  float ta[3], tb[3];                                  // the last rows (fewer than four per lane): requested together from clamped rows, no branches
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const unsigned q = (unsigned)min(p + 8 * k, P - 1) * 2u * (unsigned)C + (unsigned)c;
    ta[k] = part[q]; tb[k] = part[q + (unsigned)C];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const bool on = p + 8 * k < P;
    const double n0 = s0 + (double)ta[k], n1 = s1 + (double)tb[k];
    s0 = on ? n0 : s0; s1 = on ? n1 : s1;
  }
}

__device__ __forceinline__ void bn_bwd_reduce_finalize_32(const float* part, int P, int C, int c0, const BnBwdFin& b, double* sh) {
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = c0 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) bn_parts_lane_sum(part, P, C, c, rl, s0, s1);
  sh[rl * 32 + cl] = s0; sh[256 + rl * 32 + cl] = s1;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  for (int i = 1; i < 8; ++i) { s0 += sh[i * 32 + cl]; s1 += sh[256 + i * 32 + cl]; }
  bn_bwd_finalize_one(c, s0, s1, b.count, b.gamma, b.mean, b.rstd, b.training, b.dgamma, b.dbeta, b.c1, b.c2, b.c3);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Operand prologue descriptor used by the GEMM and elementwise kernels.
// A "matrix" is [rows, cols] row-major fp32 with cols == channels (NHWC activations).
//   mode 0 PLAIN : v = p[r,c]
//   mode 1 BNACT : v = swish(c1[c]*p[r,c] + c2[c]) * (rowp ? rowp[(r/rps)*cols + c] : 1)
//                  (BN scale/shift + SiLU + optional SE gate folded into the consumer's load)
//   mode 2 AFFINE: v = c1[c]*p[r,c] + c2[c]
struct MxOperand {
  const float* p;
  const float* c1;
  const float* c2;
  const float* rowp;
  int mode;
  int rps;  // rows per sample
};
enum { MX_PLAIN = 0, MX_BNACT = 1, MX_AFFINE = 2, MX_BNBWD = 3 };
// mode 3 BNBWD (GEMM A operand only): v = c1[c]*p[r,c] + c2[c]*rowp[r,c] + c3[c] with c1 = coefficient table [3][cols]
//   (c2, c3 = c1 + cols, + 2 cols) and rowp = the SECOND tensor [rows, same ld]: the BatchNorm backward apply
//   dX = c1*g + c2*x + c3 folded into the consumer GEMM's load; the GEMM also materialises v (GemmArgs::a_out)

__device__ __forceinline__ float4 mx_apply(const MxOperand& o, float4 v, long r, int c, int cols) {
  if (o.mode == MX_PLAIN) return v;
  float4 a = ld4(o.c1 + c), b = ld4(o.c2 + c);
  v.x = a.x * v.x + b.x; v.y = a.y * v.y + b.y; v.z = a.z * v.z + b.z; v.w = a.w * v.w + b.w;
  if (o.mode == MX_AFFINE) return v;
  v.x = swishf_(v.x); v.y = swishf_(v.y); v.z = swishf_(v.z); v.w = swishf_(v.w);
  if (o.rowp) {
    float4 g = ld4(o.rowp + (r / o.rps) * (long)cols + c);
    v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
  }
  return v;
}
