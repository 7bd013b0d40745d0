/* libmuscle_hip — C ABI of the MI355X (gfx950) kernels behind the MCL / MuSCLe training hot path.
 *
 * The reference (SCoulY/MuSCLe) is pure PyTorch and has no FFI of its own; what each entry point
 * replaces is the ATen/cuDNN op sequence at the cited reference line (paths relative to the
 * reference repository).  The drop-in boundary a user sees is the Python surface in muscle_amd/
 * (same class and function names as src/__init__.py:1-6); this header is the boundary *under* it.
 *
 * Conventions
 *   - every activation is NHWC fp32, viewed as a row-major matrix [rows = N*H*W, C]; C % 4 == 0 and
 *     every pointer 16-byte aligned (the kernels use 16-byte accesses);
 *   - weights keep the reference's state_dict layouts (conv [Cout, Cin/groups, k, k], linear [out, in]);
 *   - the caller owns all memory; the library never allocates, never retains a pointer past return;
 *   - calls only enqueue work on `stream` (a hipStream_t passed as void*), never synchronise;
 *   - return 0 on success, a negative value for an argument error detected before launch, a positive
 *     hipError_t otherwise; mx_last_error() returns a thread-local message;
 *   - "+=" outputs accumulate into what the caller hands in (parameter gradients); every element has ONE adder.
 *   - the training step is run-to-run deterministic: no result is joined through floating-point atomics.  Reductions
 *     that span workgroups use one of three schemes: (i) per-channel BatchNorm statistics: producers write one partial
 *     row per workgroup into `part[P][2][C]` (P from the matching mx_*_parts() helper, no zeroing needed) and the finalise
 *     entry points add the P rows in fp64 in a fixed order (two levels above 1024 rows: <= 64 row slices into the caller's
 *     acc[64][2C] scratch, then one thread per channel); (ii) "last arriver finishes": workgroups park partials in the
 *     caller's `ws` scratch and the last one to arrive adds them in index order (mx_pool_sum, mx_se_bn1_pool,
 *     mx_dwconv_fwd's squeeze); (iii) 64-bit fixed-point integer atomics where the terms are bounded (ER loss).
 *   - `ws` scratch (entry points that take `void* ws, long ws_bytes`; size from the matching mx_*_ws() helper, 0 = not
 *     needed): 16-byte aligned; for scheme (ii) its first 64 KB hold arrival counters that must be ZERO on entry and are
 *     zero again when the launch has drained, so one zero-initialised buffer per stream can be reused for every call.
 *   - an operand "mode" selects the prologue applied while the operand is loaded:
 *        0 PLAIN   v = x
 *        1 BNACT   v = swish(scale[c]*x + shift[c]) * (gate ? gate[n,c] : 1)     n = row / rows_per_sample
 *        2 AFFINE  v = scale[c]*x + shift[c]
 */
#ifndef MUSCLE_HIP_H
#define MUSCLE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

int mx_version(void);
int mx_abi_hash(void);        /* hash of this header the library was built against (muscle_amd/_lib.py checks it) */
const char* mx_last_error(void);

/* ---- pointwise (1x1) convolutions on fp32 MFMA: model.py:44,63,77,86 ------------------------------ */

/* C[M,N] = A'[M,K] * W[N,K]^T (+bias[N]) (+residual[M,ldc]) (relu); stats (optional) = partial column
 * (sum, sum^2) rows part[mx_pw_fwd_parts(M,N,K)][2][N] of C. */
int mx_pw_fwd_parts(int M, int N, int K);
int mx_pw_fwd(const float* A, int a_mode, const float* a_scale, const float* a_shift, const float* a_gate,
              int rows_per_sample, const float* W, float* C, int M, int K, int N, int lda, int ldc,
              const float* bias, const float* residual, int relu, float* stats, void* stream);

/* Arithmetic of the forward / data-gradient / weight-gradient GEMMs (process-wide; initial value from MX_GEMM_SPLIT, default 1).
 * 0: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) for every GEMM.
 * 1: (default) "split" arithmetic for the MFMA-bound shapes: each fp32 operand is split exactly into three bf16 terms
 *    (x = h + m + l), six of the nine cross products (each exact in fp32) are accumulated in fp32 on the bf16 matrix pipe; the
 *    dropped terms are <= 3 * 2^-24 of a product - one fp32 rounding.  Operands, accumulation and results are fp32; the error
 *    against fp64 equals the fp32-MFMA kernels'; 2.67x the matrix rate.  The other shapes stay on exact-fp32 MFMA.
 * 2: split arithmetic for every NT GEMM (tests). */
int mx_set_gemm_mode(int mode);
int mx_get_gemm_mode(void);
/* Which kernel takes the split-arithmetic weight gradients with plain operands (backward of model.py:77,86), and an optional fixed
 * number of row groups for them (process-wide; for tests and measurement).  kernel: 0 = wgrad_split_kernel (rounds 3-4), 1 = the
 * single-stream pipelined kernel, 2 = (default; initial value from MX_WGRAD_PIPE) the wave-specialised persistent kernel of round 5,
 * -1 = leave as is.  groups: > 0 fixes the row groups of every such launch (same groups => the three kernels give the same bits:
 * same tiles, same MFMA order per group, same fixed-order sum of the groups), 0 = back to the planner's choice, -1 = leave as is. */
int mx_set_wgrad_kernel(int kernel, int groups);
int mx_get_wgrad_kernel(void);
/* 1 if, in the current mode, this GEMM runs in split arithmetic on the bf16 pipe (kind 0: mx_pw_fwd / data gradient
 * C[M,N] = A[M,K] W[N,K]^T; kind 1: weight gradient dW[M=Co,N=Ci] over K=R rows), else 0 - for measurement code. */
int mx_gemm_uses_split(int kind, int M, int N, int K);

/* Second-generation split kernel (round 4): the weight of a 1x1 convolution (model.py:44,63: `_expand_conv`, `_project_conv`) is
 * split ONCE per optimizer step into three bf16 planes laid out as the kernel's LDS image (bytes: mx_pw_planes_bytes; K % 32 == 0,
 * else MX_EARG = no image for this shape), and mx_pw_fwd_planes runs C[M,N] = A[M,K] * W[N,K]^T (+bias) (+residual) (relu)
 * (+stats as mx_pw_fwd) on a PLAIN fp32 A against that image - same split arithmetic as mode 1 of mx_set_gemm_mode.
 * mx_pw_planes_batch: n matrices in one launch; table = DEVICE array of n rows of 5 longs {src W[N][K] fp32, dst image, N, K,
 * first_tile}, first_tile = running sum of mx_pw_planes_tiles, total_tiles = the full sum.
 * mx_pw_fwd_uses_planes: 1 if, in the current mode, a GEMM of this shape should take mx_pw_fwd_planes (else call mx_pw_fwd). */
long mx_pw_planes_bytes(int N, int K);
int mx_pw_planes_tiles(int N, int K);
int mx_pw_planes_batch(const long* table, int n, int total_tiles, void* stream);
int mx_pw_fwd_uses_planes(int M, int K, int N);
int mx_pw_fwd_planes(const float* A, const void* Wplanes, float* C, int M, int K, int N, int lda, int ldc,
                     const float* bias, const float* residual, int relu, float* stats, void* stream);

/* dst[cols,rows] = src[rows,cols]^T (conv weights): the data gradient runs as mx_pw_fwd against the transposed weight. */
int mx_transpose(const float* src, float* dst, int rows, int cols, void* stream);
/* n such transposes in one launch (every 1x1 conv weight of the network, once per step).  table: DEVICE array of n rows of 5 longs
 * {src, dst, rows, cols, first_tile}; first_tile = running sum of ceil(rows/32)*ceil(cols/32), total_tiles = the full sum */
int mx_transpose_batch(const long* table, int n, int total_tiles, void* stream);

/* dX[M,N] = G[M,K] * W[K,N] (+residual): data gradient of the 1x1 conv with weight W[K=Cout, N=Cin]. */
int mx_pw_dgrad(const float* G, const float* W, float* dX, int M, int K, int N, int ldg, int ldx,
                const float* residual, void* stream);

/* mx_pw_dgrad against Wt = W^T with the BatchNorm backward apply (mx_bn_bwd_apply, plain form) folded into the operand load:
 * dZ = coef[0][k]*G + coef[1][k]*X + coef[2][k] (coef = the [3][K] block mx_bn_bwd_finalize writes), dX = dZ * Wt^T (+ residual);
 * dZ [M, ldg] is also WRITTEN, for the weight gradient.  dZ must not alias G or X. */
int mx_pw_dgrad_bnbwd(const float* G, const float* X, const float* coef, const float* Wt, float* dX, float* dZ, int M, int K, int N,
                      int ldg, int ldx, const float* residual, void* stream);

/* The project convolution's forward GEMM (model.py:83-86) with its operand prologue in the second-generation split kernel:
 * C[M,N] = (swish(scale[k]*A[m,k] + shift[k]) * gate[m / rows_per_sample, k]) * W[N,K]^T, W given as its pre-split image; statistics as
 * mx_pw_fwd.  K % 32 == 0.  mx_pw_fwd_act_uses_planes: 1 where the engine should take it (narrow outputs of the HBM-bound stages). */
int mx_pw_fwd_act_uses_planes(int M, int K, int N);
int mx_pw_fwd_planes_act(const float* A, const float* scale, const float* shift, const float* gate, int rows_per_sample,
                         const void* Wplanes, float* C, int M, int K, int N, int lda, int ldc, float* stats, void* stream);

/* The BatchNorm-0 backward apply dZ = c1*G + c2*X + c3 (model.py:45 backward; coef = [3][K] as mx_bn_bwd_finalize leaves it) folded into
 * BOTH consumers of dZ in split arithmetic, so that dZ is never written (round 4):
 *   mx_pw_dgrad_bnbwd_planes: dX[M,N] = dZ[M,K] * Wt[N,K]^T (+residual), Wt given as its pre-split image (mx_pw_planes_batch);
 *   mx_pw_wgrad_tile_bnbwd:   dW[Co,Ci] += dZ[R,Co]^T X[R,Ci] (G, G2 = the BatchNorm input, [R, ldg]); scratch = mx_pw_wgrad_tile_ws(R,Co,Ci,0);
 *                             shapes: mx_pw_wgrad_tile_bnbwd_ok (1 = the split-arithmetic tiled kernel takes it in the current mode). */
int mx_pw_dgrad_bnbwd_planes(const float* G, const float* X, const float* coef, const void* WtPlanes, float* dX, int M, int K, int N,
                             int ldg, int ldx, const float* residual, void* stream);
int mx_pw_wgrad_tile_bnbwd_ok(int R, int Co, int Ci);
/* the same fold in the small-output weight-gradient kernel (stages 1-2: 192 x 32, 288 x 48; HBM-bound, so dropping the pass that
 * wrote dZ is a net saving there); scratch = mx_pw_wgrad_small_ws(R,Co,Ci,0) */
int mx_pw_wgrad_small_bnbwd_ok(int R, int Co, int Ci);
int mx_pw_wgrad_small_bnbwd(const float* G, const float* G2, const float* coef, const float* X, float* dW, int R, int Co, int Ci,
                            int ldg, int ldx, void* ws, long ws_bytes, void* stream);
int mx_pw_wgrad_tile_bnbwd(const float* G, const float* G2, const float* coef, const float* X, float* dW, int R, int Co, int Ci,
                           int ldg, int ldx, void* ws, long ws_bytes, void* stream);
/* Round 5: the fold on the weight-gradient side ONLY.  The weight gradient of the expand convolution (backward of model.py:77) runs
 * first; the loader waves of its wave-specialised kernel form dZ = c1*G + c2*G2 + c3 and also STORE it (dz [R, ldg], not aliasing G or
 * G2), and the data gradient then reads that materialised dZ like before: the apply pass mx_bn_bwd_apply (2 reads + 1 write of the
 * Cexp-wide tensors) is not launched, and no matrix wave carries a second operand stream.  _ok: 1 where the kernel takes the shape. */
int mx_pw_wgrad_tile_bnbwd_dz_ok(int R, int Co, int Ci, int ldg, int ldx);
int mx_pw_wgrad_tile_bnbwd_dz(const float* G, const float* G2, const float* coef, const float* X, float* dW, float* dz, int R, int Co,
                              int Ci, int ldg, int ldx, void* ws, long ws_bytes, void* stream);

/* dW[Co,Ci] += G[R,Co]^T * X'[R,Ci]: weight gradient, general kernel (the shapes the two below do not take).  The R pixel
 * rows are split into slices; with more than one slice each adds into its own partial matrix in `ws` (mx_pw_wgrad_ws bytes)
 * and the partial matrices are added to dW in slice order. */
long mx_pw_wgrad_ws(int R, int Co, int Ci);
int mx_pw_wgrad(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                void* ws, long ws_bytes, void* stream);

/* The same weight gradient for SMALL outputs and long reductions (Co*Ci <= 40960, R >= 65536: the first four stages): one
 * workgroup owns the whole output over a contiguous range of rows, partial matrices are added in a fixed order -
 * deterministic, both operands read once.  mx_pw_wgrad_small_ws: bytes of scratch needed, 0 = shape not taken. */
long mx_pw_wgrad_small_ws(int R, int Co, int Ci, int x_mode);
int mx_pw_wgrad_small(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                      const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                      void* ws, long ws_bytes, void* stream);

/* The same weight gradient for LARGE outputs (Co*Ci >= 16384): dW cut into 128/64-wide tiles, the rows into groups, partial
 * tiles per group added in a fixed order - deterministic, no atomics; workgroup ids are XCD-aware so that a row slab crosses
 * the fabric once.  mx_pw_wgrad_tile_ws: bytes of scratch needed, 0 = shape not taken.
 * Kernels behind it (same results contract, chosen per shape and arithmetic): split arithmetic, plain operands - one persistent
 * workgroup per CU of 4 MFMA waves + 4 loader waves (wgrad_split_ws_kernel; mx_set_wgrad_kernel selects the earlier forms);
 * exact-fp32 arithmetic (mx_set_gemm_mode(0)), plain operands, 128 x 128 tiles - the same workgroup with loader waves that only move
 * rows by LDS-DMA and v_mfma_f32_32x32x2 MFMA waves (wgrad_f32_ws_kernel); operand prologues and the other tile sizes - the first
 * split kernel / the tiled fp32 kernel.  One fp32 accumulation chain is at most 1568 rows in the wave-specialised forms. */
long mx_pw_wgrad_tile_ws(int R, int Co, int Ci, int x_mode);
int mx_pw_wgrad_tile(const float* G, const float* X, int x_mode, const float* x_scale, const float* x_shift,
                     const float* x_gate, int rows_per_sample, float* dW, int R, int Co, int Ci, int ldg, int ldx,
                     void* ws, long ws_bytes, void* stream);

/* batched plain GEMM for the PCM head (MuSCLe.py:213-223): layout 0: C=A*B^T (B [N,K]); 1: C=A*B (B [K,N]). */
int mx_bgemm(int layout, const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             long sa, long sb, long sc, int batch, int relu, void* stream);

/* ---- BatchNorm2d (train / eval), SiLU, SE gate, drop_connect + skip: model.py:45-94, utils.py:36-91 -- */

/* part[mx_colreduce_parts(rows,C)][2][C] = partial (sum x, sum x^2) per channel */
int mx_colreduce_parts(long rows, int C);
int mx_colstats(const float* X, long rows, int C, float* part, void* stream);

/* training: batch mean / biased var from the P partial rows (summed in fp64), running stats updated with
 * `momentum` (unbiased var); eval: running stats.  Writes scale = gamma*rstd, shift = beta - mean*scale, mean, rstd. */
int mx_bn_finalize(const float* part, int P, int C, double count, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, float momentum, float eps, int training, float* scale, float* shift,
                   float* mean, float* rstd, double* acc /* [64][2C] */, void* stream);

/* Inference constants of one MBConv block, one launch (eval-mode BatchNorm, model.py:72-92 under model.eval()):
 * s = gamma * rsqrt(running_var + eps), t = beta - s * running_mean for each of the block's BatchNorms {g,b,m,v};
 * We_f[Cexp,Cin] = We * s0[:,None], be = t0 (We == NULL: the block has no expand conv); s1, t1 = BN1's affine;
 * Wp_f[Cout,Cexp] = Wp * s2[:,None], bp = t2 */
int mx_fold_block(const float* We, const float* g0, const float* b0, const float* m0, const float* v0, float eps0,
                  const float* g1, const float* b1, const float* m1, const float* v1, float eps1,
                  const float* Wp, const float* g2, const float* b2, const float* m2, const float* v2, float eps2,
                  int Cin, int Cexp, int Cout, float* We_f, float* be, float* s1, float* t1, float* Wp_f, float* bp, void* stream);

/* out = (scale[c]*P + shift[c]) [swish if act] [* gate[n,c]] [* row_scale[n]] [+ residual]
 * (BN2 + drop_connect + skip; with act + gate: the activated, SE-gated project-conv input, model.py:78-84) */
int mx_bn_apply(const float* P, const float* scale, const float* shift, const float* row_scale, const float* residual,
                const float* gate, float* out, long rows, int C, int rows_per_sample, int act, void* stream);

/* effective upstream gradient  g = G [* row_scale[n]] ; [g = g*gate[n,c] + gate_add[n,c]] ;
 * [g *= swish'(act_scale[c]*X + act_shift[c])] ;  part[mx_colreduce_parts(rows,C)][2][C] = partial (sum g, sum g*X) */
int mx_bn_bwd_reduce(const float* G, const float* X, const float* row_scale, const float* gate, const float* gate_add,
                     const float* act_scale, const float* act_shift, long rows, int C, int rows_per_sample,
                     float* part, void* stream);

/* dgamma += rstd*(sum gx - mean*sum g); dbeta += sum g; (c1,c2,c3) so that dX = c1*g + c2*X + c3 */
int mx_bn_bwd_finalize(const float* part, int P, int C, double count, const float* gamma, const float* mean, const float* rstd,
                       int training, float* dgamma, float* dbeta, float* c1, float* c2, float* c3, double* acc,
                       void* stream);

/* out = c1[c]*g + c2[c]*X + c3[c] with g as in mx_bn_bwd_reduce (out may alias G) */
int mx_bn_bwd_apply(const float* G, const float* X, const float* row_scale, const float* gate, const float* gate_add,
                    const float* act_scale, const float* act_shift, const float* c1, const float* c2, const float* c3,
                    float* out, long rows, int C, int rows_per_sample, void* stream);

/* out[n,c] = sum_hw f(X[n,hw,c]),  f = [scale*x+shift] [swish if act] [* G]: SE squeeze (model.py:82),
 * head GAP (MuSCLe.py:240) and the gate-gradient reduction of their backward.  Overwrites `out`; ws: mx_pool_ws(.., 1). */
long mx_pool_ws(long rows, int C, int rows_per_sample, int planes);
int mx_pool_sum(const float* X, const float* G, const float* scale, const float* shift, int act, long rows, int C,
                int rows_per_sample, float* out, void* ws, long ws_bytes, void* stream);

/* SE + BN1 backward reductions in one pass over (dA, X = d_raw): out5[5][N][C] = per-(sample,channel) sums of
 * {dA*act, dA*s', s', dA*s'*X, s'*X} with z = scale*X+shift, act = swish(z), s' = swish'(z); out5[0] is dL/dgate
 * (model.py:84).  Overwrites out5; ws: mx_pool_ws(.., 5). */
int mx_se_bn1_pool(const float* dA, const float* X, const float* scale, const float* shift, long rows, int C, int rows_per_sample,
                   float* out5, void* ws, long ws_bytes, void* stream);
/* From those sums, in one launch and without a second pass over the tensors:
 *   add[n,c] = inv_hw * sum_j gh[n,j] W1[j,c]  (WRITTEN: the gradient reaching d_raw through the squeeze path, model.py:82-83;
 *              gh [N,SQ] from mx_se_bwd, W1 = _se_reduce.weight [SQ,C]),
 *   the BatchNorm-1 backward sums for g = (dA*gate + add)*s' and mx_bn_bwd_finalize on them: dgamma/dbeta += and the
 *   coefficients c1, c2, c3 of dX = c1*g + c2*X + c3. */
int mx_bn1_sums_finalize(const float* pooled5, const float* gate, const float* gh, const float* W1, float inv_hw, float* add,
                         int N, int C, int SQ, double count, const float* gamma, const float* mean, const float* rstd, int training,
                         float* dgamma, float* dbeta, float* c1, float* c2, float* c3, void* stream);

/* ---- depthwise k x k convolution, k in {3,5}, stride in {1,2}: model.py:50-52,78; utils.py:122-145 ---- */

/* Y = dwconv(act(X)), act = swish(scale*x+shift) when scale != NULL; stats (optional, training) = partial (sum Y, sum Y^2)
 * rows part[mx_dwconv_fwd_parts(N,Ho,Wo,S)][2][C]; pooled (optional, inference, excludes stats):
 * pooled[n][c] = sum_hw swish(pool_scale[c]*Y + pool_shift[c]), the SE squeeze of model.py:81-82 under eval-mode BN1
 * (ws: mx_dwconv_fwd_ws bytes, only read when pooled is given) */
int mx_dwconv_fwd_parts(int N, int Ho, int Wo, int C, int S);
long mx_dwconv_fwd_ws(int N, int Ho, int Wo, int C, int S);
int mx_dwconv_fwd(const float* X, const float* scale, const float* shift, const float* W, float* Y, float* stats,
                  const float* pool_scale, const float* pool_shift, float* pooled, void* ws, long ws_bytes, int N,
                  int H, int Wd, int C, int K, int S, int pad_lo, int Ho, int Wo, void* stream);

/* dX = dwconv^T(dY) (+residual): gradient w.r.t. the activated input */
int mx_dwconv_bwd_data(const float* dY, const float* W, const float* residual, float* dX, int N, int H, int Wd, int C,
                       int K, int S, int pad_lo, int Ho, int Wo, void* stream);

/* dW[C,1,K,K] += sum dY * act(X).  dw_scratch: [mx_dwconv_bwd_weight_parts(N,Ho,Wo,C,S)][C*K*K] floats for per-workgroup
 * partial rows (added in a fixed order by a second kernel). */
int mx_dwconv_bwd_weight_parts(int N, int Ho, int Wo, int C, int S);
int mx_dwconv_bwd_weight(const float* X, const float* scale, const float* shift, const float* dY, float* dW, float* dw_scratch,
                         int N, int H, int Wd, int C, int K, int S, int pad_lo, int Ho, int Wo, void* stream);

/* Stride-1 backward of [BN0+SiLU] -> dwconv -> BN1 -> SiLU -> SE gate in one pass: the BatchNorm-1 data gradient
 * dd = c1*g + c2*D + c3 with g = (dA*gate + add)*swish'(a1*D + b1) is formed while staging (never stored);
 * gX = dwconv^T(dd) [* swish'(a0*X+b0) when a0] [+ residual]; dW += sum dd*act(X); when a0 is given,
 * part[mx_dwconv_bwd_fused_parts()][2][C] = BatchNorm-0 backward partial sums (sum g, sum g*X) of the written gX.
 * dw_scratch[mx_dwconv_bwd_fused_parts()][C*K*K] is workspace (per-workgroup dW rows, summed into dW by a second launch; with
 * dW == NULL that launch is left to the caller: mx_dw_parts_reduce(dw_scratch, parts, C*K*K, dW) on any stream ordered behind this call -
 * nothing consumes a depthwise weight gradient before the optimizer, so it need not sit on the backward's critical path). */
int mx_dwconv_bwd_fused_parts(int N, int H, int Wd, int C, int K);
int mx_dwconv_bwd_fused(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                        const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                        const float* W, const float* residual, float* gX, float* dW, float* dw_scratch, float* part, int N, int H,
                        int Wd, int C, int K, int pad_lo, void* stream);

/* mx_dwconv_bwd_fused with the BatchNorm-0 backward statistics FINISHED in the kernel (a0 / b0 / part required): the last workgroup of
 * every 32-channel chunk adds the partial rows in mx_bn_bwd_finalize's order and writes dgamma / dbeta (+=) and the coefficients o1 o2 o3
 * of dX = o1*g + o2*X + o3 - bit for bit what mx_bn_bwd_finalize(part, ...) would have left, without its launch.  ws: zeroed scratch
 * whose first 64 KB are arrival counters (left zero by the call), 16-byte aligned. */
int mx_dwconv_bwd_fused_bn0(const float* dA, const float* D, const float* gate, const float* add, const float* a1, const float* b1,
                            const float* c1, const float* c2, const float* c3, const float* X, const float* a0, const float* b0,
                            const float* W, float* gX, float* dW, float* dw_scratch, float* part, int N, int H, int Wd, int C, int K,
                            int pad_lo, void* ws, long ws_bytes, double count, const float* gamma, const float* mean, const float* rstd,
                            int training, float* dgamma, float* dbeta, float* o1, float* o2, float* o3, void* stream);

int mx_dw_parts_reduce(const float* part, int P, int n, float* dW, void* stream);

/* ---- SE excitation (model.py:83-84) and stem patches (model.py:131,175) -------------------------------- */

/* s = pooled_sum*inv_hw; h = W1 s + b1; gate = sigmoid(W2 swish(h) + b2) */
int mx_se_fwd(const float* pooled_sum, float inv_hw, const float* W1, const float* b1, const float* W2, const float* b2,
              float* s, float* h, float* gate, int N, int C, int SQ, void* stream);

/* given ggate[n,c] = dL/dgate: gh[n,j] = dL/d(se_reduce output) [N,SQ] (WRITTEN; mx_bn1_sums_finalize turns it into the
 * pooled-path gradient `add`); dW1,db1,dW2,db2 += (each element summed over the samples by one thread, in order) */
int mx_se_bwd(const float* ggate, const float* gate, const float* s, const float* h, const float* W2,
              float* dW1, float* db1, float* dW2, float* db2, float* gh, int N, int C,
              int SQ, void* stream);

/* The two halves of mx_se_bwd as separate launches: gh (consumed by mx_bn1_sums_finalize on the data-gradient chain) and the four
 * parameter gradients (no consumer before the optimizer: the engine queues them beside the weight-gradient GEMMs). */
int mx_se_bwd_gh(const float* ggate, const float* gate, const float* h, const float* W2, float* gh, int N, int C, int SQ,
                 void* stream);
int mx_se_bwd_params(const float* ggate, const float* gate, const float* s, const float* h, const float* gh, float* dW1, float* db1,
                     float* dW2, float* db2, int N, int C, int SQ, void* stream);

/* out[(n,oy,ox), ci*9+ky*3+kx] = img[n,ci,oy*2-pad+ky,ox*2-pad+kx] (NCHW image), rows of 28 floats (27 + 0) */
int mx_stem_im2col(const float* img, float* out, int N, int H, int W, int Ho, int Wo, int pad_lo, void* stream);

/* ---- CAM head + PCM support: MuSCLe.py:213-223,237-279 ---------------------------------------------------- */

/* dst[n,y,x,coff+c] = [relu] bilinear_align_corners(src[n,:,:,c]); NHWC -> channel slice of an NHWC tensor of width ldd */
int mx_resize_nhwc(const float* src, float* dst, int N, int Hs, int Ws, int C, int Hd, int Wd, int ldd, int coff, int relu,
                   void* stream);

/* dst[n,k,Y,X] (NCHW) = bilinear_align_corners(src[n,:,:,k]); src NHWC with leading dimension lds (MuSCLe.py:256-257) */
int mx_upsample_to_nchw(const float* src, float* dst, int N, int Hs, int Ws, int lds, int K, int Hd, int Wd, void* stream);

/* ---- infer_mcl.py post-processing, one forward pass of the multi-scale / flip list:
 * acc[k-1,Y,X] += resize_halfpixel( upsample_align_corners(src[:,:,k], Hs x Ws), H x W )[Y, flip ? W-1-X : X]  for k = 1..K-1
 * (infer_mcl.py:124-148: model upsample, cv2.resize to the original size, np.flip of the odd passes, sum over passes).
 * src: ONE sample, NHWC [h,w,lds], channel 0 = background. */
int mx_infer_accum(const float* src, float* acc, int h, int w, int lds, int K, int Hs, int Ws, int H, int W, int flip, void* stream);
/* in place per channel over [channels][HW] (infer_mcl.py:153-158 / :161-166): v = max(v,0); v[v < min+1e-6] = 0;
 * v = (v - min - 1e-6) / (max - min + 1e-6) */
int mx_infer_norm(float* acc, int channels, long HW, void* stream);

/* ---- input stage (src/data.py:215-332, src/imutils.py:143-181,376-388) ------------------------------
 * dst[n,3,Hd,Wd] (fp32, fully written) = RandomCrop container of color_norm(uint8 HWC crop n) at (top,left), CHW, zeros
 * elsewhere; src = packed crops, jobs = n x 8 int32 {src_off, sh, sw, top, left, ey | ex<<16, eh | ew<<16, source row
 * stride in pixels or 0 = sw}, both on the
 * device; the e* box (0 = none) is RandomErasing(value=0) of train_mcl.py:114 in output coordinates.
 * Bit-exact with the numpy expressions (fp64 (x/255 - mean)/std, one rounding to fp32). */
int mx_input_stage(const unsigned char* src, const int* jobs, float* dst, int n, int Hd, int Wd, void* stream);

/* transforms.ColorJitter (train_mcl.py:108, src/data.py:223; torchvision 0.9.0 PIL backend = Pillow's ImagingBlend, rgb2l,
 * rgb2hsv_row / hsv2rgb) in place on n uint8 HWC images inside src, bit-exact with Pillow.  jobs: n x 8 words {byte
 * offset, h, w, order (one nibble per position: 0 brightness, 1 contrast, 2 saturation, 3 hue, 15 none), brightness,
 * contrast, saturation factors (float32), hue shift 0..255}; sums: n uint64 of scratch. */
int mx_color_jitter(unsigned char* src, const int* jobs, unsigned long long* sums, int n, int max_pixels, void* stream);

/* PIL.Image.resize (imutils.RandomResizeLong's bicubic, src/imutils.py:127-141; get_views' bilinear 448x448,
 * src/data.py:274-276) = Pillow's 8-bit ImagingResample: horizontal pass src -> tmp, vertical pass tmp -> dst, 22-bit
 * fixed-point coefficient tables computed by the host as Pillow's precompute_coeffs does.  jobs: n x 8 int32 {src_off, Hin,
 * Win, tmp_off, dst_off, Wout, Hout, tab_off}; tabs (int32): per job ksize_h, ksize_v, bounds_h[Wout][2], kk_h[Wout][ksize_h],
 * bounds_v[Hout][2], kk_v[Hout][ksize_v].  Bit-exact with Pillow. */
int mx_resample(const unsigned char* src, const int* jobs, const int* tabs, unsigned char* tmp, unsigned char* dst, int n,
                int max_pixels, void* stream);

/* ---- IRN random-walk propagation (src/indexing.py:77-142 as called by infer_irn.py:76).
 * mx_irn_affinity: dense[n4][ld] (zero-filled here) <- symmetric affinity 1 - max(edge along the straight path) for every
 *   pixel pair joined by one of the nd search directions, unit diagonal; edge [h,w]; pcoord = int32 (dy,dx) pairs of all
 *   paths back to back (farthest pixel first = the destination), poff[d] / plen[d] = first pair and length of path d.
 * mx_irn_transition: dense <- dense^beta / column sums (to_transition_matrix :116-118); colsum[n4] is workspace.
 * The matrix powers and the final product are mx_bgemm calls. */
int mx_irn_affinity(const float* edge, int h, int w, int radius, const int* pcoord, const int* poff, const int* plen, int nd,
                    float* dense, int ld, int n4, void* stream);
int mx_irn_transition(float* dense, int n4, int ld, float beta, float* colsum, void* stream);
/* infer_irn.py:78-94 after the walk: rw [C,h,w] -> 4x bilinear (half-pixel) upsample, top-left HxW crop, / global max,
 * label[H,W] (uint8) = argmax over [bg_thres, maps] (first maximum wins); soft_half (optional, fp16 [H,W,C+1]) = the
 * values themselves as the script's --soft_output writes them.  max_scratch: one uint32. */
int mx_irn_finish(const float* rw, int C, int h, int w, int H, int W, float bg_thres, unsigned* max_scratch, unsigned char* label,
                  void* soft_half, void* stream);

/* ---- per-epoch rapid evaluation (train_mcl.py:286-318 + src/evaluation.py:19-52), one image:
 * for each threshold t: predict = argmax_k [t, half(pred_k*label_k)] (first maximum wins); over pixels with gt < 255:
 * counts[t][k][0..2] += (TP, P, T).  pred [K,H,W] fp32 (cam_maxnorm'ed), label [K], gt uint8 [H,W], counts int64 [nt][K][3]. */
int mx_eval_confusion(const float* pred, const float* label, const unsigned char* gt, const float* thresholds, int nt, int K, int H,
                      int W, long long* counts, void* stream);

/* adjoint of mx_upsample_to_nchw: gsrc (=|+=) W^T gdst */
int mx_upsample_to_nchw_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int lds, int K, int Hd, int Wd,
                            int accumulate, void* stream);

/* y = x / (||x||_2 + eps) per row, nrm saved (MuSCLe.py:218) and its backward */
int mx_row_l2norm(const float* x, float* y, float* nrm, long R, int C, float eps, void* stream);
int mx_row_l2norm_bwd(const float* x, const float* nrm, const float* gy, float* gx, long R, int C, float eps, void* stream);

/* PCM tail on T = aff * [cam | 1]: fwd rv = T[:, :K] / (T[:, K] + eps) (MuSCLe.py:221-222); bwd gives dL/dT */
int mx_pcm_norm(const float* T, const float* grv, float* out, long rows, int L, int K, float eps, int bwd, void* stream);

/* out[b,i,j] = (gaff[b,i,j] + gaff[b,j,i]) * (aff[b,i,j] > 0): gradient through relu(f^T f) (MuSCLe.py:220);
 * matrices are [b, n, ld] with ld >= n (padding columns -> 0) */
int mx_sym_relu_grad(const float* gaff, const float* aff, float* out, int B, int n, int ld, void* stream);

/* flat elementwise: op 0 out = alpha*a; op 1 out = a + alpha*b; op 2 out = (y > 0) ? a (+ b) : 0 */
int mx_ew(int op, const float* a, const float* b, const float* y, float alpha, float* out, long n, void* stream);

/* X[n,hw,c] += alpha * v[n,c] (backward of the global average pool, MuSCLe.py:240) */
int mx_bcast_add(float* X, const float* v, float alpha, long rows, int C, int rows_per_sample, void* stream);

/* ---- losses of the MCL step and the optimiser ---------------------------------------------------------------- */

/* [N,C] classification losses with d loss / d input for unit upstream gradient:
 * mode 0 focal(p,y) gamma 2 alpha .5 (loss_multilabel.py:68-91) -> loss[0] = ; 1 MultiLabelSoftMargin(x,y)
 * (train_mcl.py:146) -> loss[0] = ; 2 Log_Sum_Exp_Pairwise(p,y) (loss_multilabel.py:24-33) -> loss[n];
 * 3 grad = sigmoid(x); 4 grad = y * x * (1 - x) (sigmoid backward, x = sigmoid output, y = upstream) */
int mx_cls_loss(int mode, const float* x, int ldx, const float* y, int ldy, float* loss, float* grad, int ldg, int N, int C,
                void* stream);

/* image_level_contrast (loss_multilabel.py:36-66): out2 = {loss, #valid anchor rows}; gemb = d loss / d emb;
 * workspace: N*D + 2*N*N + 4*N + 16 floats, 16-byte aligned.  N <= 64, D <= 1024.  The pair matrix exp(e_i . e_j / 0.1) (a dense
 * N x N x D contraction) and the gradient's [N x N] x [N x D] product run on v_mfma_f32_16x16x4_f32 in ceil(N/16) workgroups
 * when D % 16 == 0 (one VALU workgroup otherwise). */
int mx_imc(const float* emb, const float* label, int N, int D, int L, float* out2, float* gemb, float* workspace, void* stream);

/* cam_softmaxnorm (train_mcl.py:30-36) on NCHW [N,K,HW]; bwd != 0: out = d/dx given gy */
int mx_softmaxnorm(const float* x, const float* gy, float* out, int N, int K, long HW, int bwd, void* stream);

/* ER loss (train_mcl.py:185-188): mean over rows of the top-k of |softmaxnorm(cams)-softmaxnorm(sgcs)|*lwb, exact
 * 3-pass radix select.  d [N*K*HW] scratch; krem/prefix/sum_gt/cnt_eq [N] and hcnt/hsum [N*2048] state
 * (prefix and sum_gt zeroed by the caller) are kept for mx_er_bwd.  sum_gt and hsum are 64-bit fixed-point sums
 * (value * 2^36; the terms are <= 1): integer atomics, so the loss has the same bits every run. */
int mx_er_fwd(const float* cams, const float* sgcs, const float* lwb, int N, int K, long HW, long k, float* d, unsigned* krem,
              unsigned* prefix, unsigned long long* sum_gt, unsigned* cnt_eq, unsigned* hcnt, unsigned long long* hsum, float* loss,
              void* stream);
/* gsgcs = d loss / d raw_sgcs * gscale * (gup ? gup[0] : 1): gup is the upstream gradient as a device scalar */
int mx_er_bwd(const float* cams, const float* sgcs, const float* lwb, const unsigned* prefix, const unsigned* krem,
              const unsigned* cnt_eq, const float* gup, float gscale, float* gsgcs, int N, int K, long HW, void* stream);

/* The same ER loss computed straight from the low-resolution NHWC maps cam/sgc [N,h,w,L] (MuSCLe.py:256-257 fused in):
 * no H*W-sized tensor is read or written.  gsgc [N,h,w,L] = d loss / d sgc_lowres. */
/* k_dev (optional, device int32): the top-k count read at run time instead of k / gscale's 1/(N k) - for hipGraph replays
 * of the step, where k = int(0.2 * sum(labels) * H * W) (train_mcl.py:178,188) changes with every batch */
int mx_er_lr_fwd(const float* cam, const float* sgc, const float* lwb, int N, int h, int w, int L, int K, int H, int W, long k,
                 const int* k_dev, unsigned* krem, unsigned* prefix, unsigned long long* sum_gt, unsigned* cnt_eq, unsigned* hcnt,
                 unsigned long long* hsum, float* vals /* optional scratch [N*K*H*W]: see below */, float* loss, void* stream);
/* vals: with it the first digit pass parks |diff|*mask for the planes of the labelled classes (the rest of the buffer is
 * never touched) and the two later passes histogram those 4-byte values instead of re-evaluating upsample + softmaxes per
 * pixel; NULL = three evaluating passes, nothing of size H*W allocated.  Same result bit for bit. */
/* ws: mx_er_lr_bwd_ws bytes (64-bit fixed-point accumulators of the band kernel; plain scratch, no counter header) */
long mx_er_lr_bwd_ws(int N, int h, int w, int L, int K);
int mx_er_lr_bwd(const float* cam, const float* sgc, const float* lwb, const unsigned* prefix, const unsigned* krem,
                 const unsigned* cnt_eq, const float* gup, float gscale, const int* k_dev, float* gsgc, int N, int h, int w, int L,
                 int K, int H, int W, void* ws, long ws_bytes, void* stream);

/* torch.optim.Adam(weight_decay) update on flat arrays (train_mcl.py:134,199,229) */
/* dyn (optional, device float[3] = {lr, bias_corr1, sqrt_bias_corr2}): read at run time instead of the arguments (hipGraph replays) */
int mx_adam(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
            float weight_decay, float bias_corr1, float sqrt_bias_corr2, const float* dyn, void* stream);

/* ---- phase 2 (epochs >= 8 / >= 12): cam_maxnorm, PixPro, crops, Sinkhorn EMD ------------------------------------ */

/* cam_maxnorm (train_mcl.py:21-28) per (n,k) plane of HW elements; stats[nk] = {min, max, argmin, argmax};
 * bwd != 0: out = d/dx given gy (gradient also flows through the min / max elements, as torch's does) */
int mx_maxnorm(const float* x, const float* gy, float* out, float* stats, int NK, long HW, int bwd, void* stream);

/* PixPro (loss_multilabel.py:93-105) on NCHW maps (optionally multiplied by mask[n,k], train_mcl.py:209); coords are
 * int64 [N,4] = (h0,w0,hl,wl).  loss[0] is overwritten; g1 must hold zeros on entry; g1 = d loss / d f1.  ws: N*8 bytes (the
 * per-sample cosine sums are 64-bit fixed-point integers: same bits every run). */
int mx_pixpro(const float* f1, const float* f2, const float* mask, const long* coord1, const long* coord2, float* loss, float* g1,
              int N, int K, int H, int W, void* ws, long ws_bytes, void* stream);

/* F.normalize(x, dim=1) on NCHW (train_mcl.py:218-219) and its backward */
int mx_chan_l2norm(const float* x, const float* gy, float* out, int N, int K, long HW, int bwd, void* stream);

/* torchutils.get_dynamic_crops (torchutils.py:217-291) pieces.  table rows of 8 ints {sample,y0,x0,lh,lw,rh,rw,out_off}:
 * out[out_off + r*rw + c, 0..23] = bilinear_align_corners(src[sample,:,y0:y0+lh,x0:x0+lw] -> (rh,rw)), 21 classes + 0 pad */
int mx_crop_resize(const float* src, const int* table, int ncrops, float* out, int K, int H, int W, void* stream);
/* adjoint: gsrc[nsamples,K,H,W] += ...; ordered gather (one adder per element, crops of a sample in table order) */
int mx_crop_resize_bwd(const float* gout, const int* table, int ncrops, float* gsrc, int nsamples, int K, int H, int W, void* stream);
/* 4x4/stride-4 average pool over packed crops; table rows of 4 ints {in_off,h,w,out_off}; bwd: in = pooled grad */
int mx_avgpool4(const float* in, const int* table, int ncrops, float* out, int bwd, void* stream);

/* EMD.dynamic_matching (loss_multilabel.py:287-326).  pairs rows of 6 ints {x_off,n1,y_off,n2,sample,rank}: one workgroup per
 * row, rows in any order (longest first balances the chip); rank = the pair's position in the reference's enumeration.
 * scores: Sinkhorn distance of every pair; traj (optional): npairs * 11*(maxn1+maxn2) floats that receive the 11 states
 *         (u_t, v_t) of every pair, which mx_emd_grad differentiates;
 * best:   minimal-score pair (row index) per sample, ties to the lower rank (= the first minimal pair of the reference's stable
 *         sort, :318); loss += mean of their scores;
 * grad:   gx[crop-1 pixels of each best pair] = d loss / d x (through the 10 iterations), scaled by gscale*(gup?gup[0]:1);
 *         traj = what mx_emd_scores recorded for the same table */
int mx_emd_scores(const float* feat, const int* pairs, int npairs, int maxn1, int maxn2, float* score, float* traj, void* stream);
int mx_emd_best(const float* score, const int* pairs, int npairs, int nsamples, int* best, float* loss, void* stream);
int mx_emd_grad(const float* feat, const int* pairs, const int* best, int nsamples, int maxn1, int maxn2, const float* traj,
                const float* gup, float gscale, float* gx, void* stream);

/* ---- decoder mode / config 4: BiFPN support, CE, gradient clipping, BEACON FieldLoss ------------------------------ */

/* F.avg_pool2d(k=3,s=2,p=1) on NHWC (MuSCLe.py:51,54); bwd: x = pooled gradient, y = input gradient */
int mx_avgpool3s2(const float* x, float* y, int N, int H, int W, int C, int bwd, void* stream);
/* adjoint of mx_resize_nhwc (no relu): gsrc += W^T gdst */
int mx_resize_nhwc_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int C, int Hd, int Wd, void* stream);   /* ordered gather; C % 4 == 0 */
/* CrossEntropyLoss(seg [N,K,HW], argmax_k mask) (train_muscle.py:189-191): loss[0] += mean; bwd: gseg = gup[0]*dL/dseg */
int mx_ce_argmax(const float* seg, const float* mask, const float* gup, float* loss, float* gseg, int N, int K, long HW, int bwd,
                 void* ws /* forward: 8 bytes, the fixed-point sum */, long ws_bytes, void* stream);
/* clip_grad_norm_(max_norm, 2) on a flat gradient arena (train_muscle.py:202); norm_out (optional) = total norm;
 * sq_scratch: 2048 doubles (per-workgroup partial square sums, added in a fixed order) */
int mx_clip_grad_norm(float* grads, long n, float max_norm, double* sq_scratch, float* norm_out, void* stream);
/* FieldLoss stage 1 (edge.py:423-440,45-89): softmax(beta*seg)[1:], 5x5 Sobel per labelled class, magnitude,
 * 8-way orientation, per-(n,class) max (float bits), edge_fg = sum over classes */
int mx_field_edges(const float* seg, const float* lab_fg, float beta, float* prob, float* mag, unsigned char* orient, unsigned* mx,
                   float* edge_fg, int N, int K, int H, int W, void* stream);
/* stage 2 (edge.py:196-229,372-375): per slot {n,class}: ordered out / in point lists [S][H*W], counts [S][3] */
int mx_field_select(const float* mag, const unsigned char* orient, const unsigned* mx, const int* slots, int nslots, int step,
                    int* out_list, int* in_list, int* counts, int F, int H, int W, void* stream);
/* stage 3: channel-softmaxed dense features (mode 0 full-res NCHW, mode 1 low-res NHWC upsampled on the fly) and
 * class-softmaxed mask at the sampled points pts {n, pixel} */
int mx_field_gather(const float* dense, int mode, int h, int w, const float* mask, const int* pts, int npts, float* feat, float* mfeat,
                    int CH, int K, int ML, int H, int W, void* stream);
/* stage 5 (edge.py:231-261,330-347): loss += terms/nsamples, gsim = d loss / d sim, per slot of k x k similarities */
int mx_field_terms(const float* sim, const float* simm, int nslots, int k, float inv_n, float* loss, float* gsim,
                   float* slot_loss /* nslots floats of scratch */, void* stream);
/* stage 6: softmax backward at the out points, scattered (+=) into the dense-feature gradient */
int mx_field_scatter(const float* feat, const float* gfeat, const int* pts, int npts, int mode, int h, int w, const float* gup,
                     float* gdense, float* gpt /* mode 1: npts*CH floats of scratch */, int nsamples, int CH, int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSCLE_HIP_H */
