/* C ABI of libspnet_hip.so -- the MI355X (gfx950) kernels of the SPNet detection hot path.
 *
 * The reference (drscotthawley/SPNet) has no native code and no FFI: its hot path runs inside
 * Keras 2.1.3 / TensorFlow 1.14 / numpy.  Each entry point below names the reference call site whose
 * arithmetic it replaces (paths relative to the reference root).  A maintainer binds these with ctypes
 * (see INTEGRATION.md and spnet_amd/_lib.py).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 data unless noted; tensors are NHWC, row-major
 *   - pointers passed to vectorised kernels must be 16-byte aligned, channel counts multiples of 4
 *     unless the entry says otherwise
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, they never
 *     synchronise, allocate or free, so a sequence of calls can be captured in a hipGraph
 *   - return value: 0 on success, otherwise a hipError_t (hipErrorInvalidValue for shape/alignment
 *     violations)
 */
#ifndef SPNET_HIP_H
#define SPNET_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- dense contractions (MFMA fp32) ---------------------------------------------------------- */
/* C[M,N] = sum_k A(m,k) B(k,n) (+ bias[n]).  Replaces the pointwise half of SeparableConv2D, the
 * 1x1/stride-2 residual Conv2D and Dense('FinalOutput')
 * (spnet/models.py:357-359, 388), forward, data-gradient and weight-gradient forms.
 * a_major / b_major: 0 = reduction index contiguous (A[m*lda+k], B[n*ldb+k]),
 *                    1 = output index contiguous    (A[k*lda+m], B[k*ldb+n]).
 * Supported pairs: (0,1) forward, (0,0) dgrad, (1,1) wgrad.
 * split_k: 0 = choose automatically, >1 = that many K slices summed deterministically through
 * `workspace` (>= split_k*M*N floats).  tile (BM x BN): 0 auto (cost model over ids 1-8), 1 128x128, 2 128x64, 3 64x64,
 * 4 32x128, 5 96x96, 6 96x64, 7 64x128, 8 128x96, 9 32x64, 10 32x32 (9 and 10: few-row problems, chosen only through
 * spnet_amd/gemm_tiles.json, where tools/autotune_gemm.py measured them faster inside the step). */
int spnet_gemm_f32(const float* A, int a_major, int lda, const float* B, int b_major, int ldb, float* C,
                   int ldc, int M, int N, int K, int split_k, float* workspace, long ws_floats,
                   const float* bias, int tile, void* stream);
/* ---- fp32 GEMM on the bf16 matrix cores by operand splitting (csrc/gemm_bf16x3.hip): the forward, data-gradient and
 * weight-gradient GEMMs of the pointwise convolutions (keras SeparableConv2D pointwise step / 1x1 Conv2D;
 * spnet/models.py:346-359).  Every fp32 operand is the exact sum of three bf16 pieces; six bf16 MFMAs with fp32 accumulation
 * per product block; error against float64 no larger than spnet_gemm_f32's, not the same bits.
 * "planes" of a matrix X[R][K]: 3 * spnet_bf16x3_plane_elems(R, K) bf16 (high, middle, low piece) in 1-KiB pieces of 16 rows
 * x 32 columns (layout: csrc/x3t.h), 16-byte aligned, allocated ZEROED (pad rows / columns are never written), at most 2 GiB.
 *   spnet_split_bf16x3           planes of a Keras pointwise kernel W[K][N] in the forward form (rows = output channels)
 *   spnet_split_rows_bf16x3      planes of an fp32 matrix A[R][K] (row stride lda)
 *   spnet_split_bf16x3_batched   all splits of a step in one launch, job = {src, planes, K, N, sn, sk} as six 64-bit words in
 *                                device memory, element (row n, column k) = src[n*sn + k*sk] (forward form of W[cin][cout]:
 *                                K = cin, N = cout, sn 1, sk cout; data gradient dX = dY W^T: K = cout, N = cin, sn cout,
 *                                sk 1); max_elems = the largest plane_elems(N, K) of the batch
 *   spnet_gemm_bf16x3_fwd        C[M][N] = A B^T, A [M][K] fp32 split inside the kernel (lda % 4 == 0, K % 4 == 0, 16-byte
 *                                aligned), B = planes of [N][K]
 *   spnet_gemm_bf16x3_pp         the same with A given as planes of [M][K] (written by the producing kernel:
 *                                spnet_dwconv3x3_*_x3, spnet_bn_bwd_*_x3): no split, no stage registers, LDS-DMA only
 *   _colstats / colstats != NULL BatchNorm column sums of C as [*stat_rows][2][N] partial rows, *stat_rows = ceil(M/96)
 *                                (stat_rows: HOST int)
 *   spnet_gemm_bf16x3_wgrad_batched  nbatch weight gradients dW[cin][cout] = z^T dy of ONE shape in one launch, from the
 *                                planes of z [M][cin] and dy [M][cout]; jobs = DEVICE array of {z planes, dy planes, dst}
 *                                (three 64-bit words each); ksplit == 1: dst = dW; ksplit > 1: dst = ksplit slabs of
 *                                cin*cout floats in slice order (add them with spnet_reduce_slabs); deterministic.
 *                                spnet_gemm_bf16x3_wgrad_ksplit: the slice count this library would choose. */
long spnet_bf16x3_kp(int K);
long spnet_bf16x3_plane_elems(long R, int K);
int spnet_split_bf16x3(const float* W, void* planes, int K, int N, void* stream);
int spnet_split_rows_bf16x3(const float* A, long lda, void* planes, long R, int K, void* stream);
int spnet_split_bf16x3_batched(const void* jobs, int njobs, long max_elems, void* stream);
int spnet_gemm_bf16x3_fwd(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K, void* stream);
int spnet_gemm_bf16x3_fwd_colstats(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                                   float* colstats, int* stat_rows, void* stream);
int spnet_gemm_bf16x3_pp(const void* a_planes, const void* b_planes, float* C, int ldc, int M, int N, int K, float* colstats,
                         int* stat_rows, void* stream);
long spnet_gemm_bf16x3_wgrad_ksplit(int cin, int cout, int M, int nbatch);
/* The data-gradient GEMM of a SeparableConv2D's pointwise step with the backward of its DEPTHWISE step fused into the
 * epilogue (keras SeparableConv2D inside Xception's middle / exit flow; spnet/models.py:357-359): dz = dy W^T stays in LDS.
 * dy_planes = planes of dL/d(pointwise output) [B*H*W][cout], w_planes = planes of W as stored ([cin][cout]); the plane must
 * satisfy spnet_gemm_bf16x3_dwbwd_ok(H, W, cin) (192 % (H*W) == 0, W <= 16: a 192-pixel tile holds whole images).  Every
 * other argument and every output is spnet_dwconv3x3_tiled_bwd's (x_fwd = the depthwise input, w = its [3][3][cin] kernel,
 * dx bit-identical to the two-launch path; the dw / BatchNorm partial sums in spnet_gemm_bf16x3_dwbwd_rows(B*H*W) rows).
 * spnet_gemm_bf16x3_dwbwd_ok returns 1 | 0 (a predicate, not a launch status). */
/* Inference counterpart: the FORWARD pointwise GEMM (z_planes [B*H*W][cin] x w_planes in the forward form) with, in its
 * epilogue, the folded BatchNormalization (scale_shift[2*cout] or NULL) + ReLU (relu_next) of its output and the depthwise
 * 3x3 of the NEXT SeparableConv2D (w_next [3][3][cout]); out_planes = planes of [B*H*W][cout], bit-identical to
 * spnet_gemm_bf16x3_pp followed by spnet_dwconv3x3_stream_fwd_x3 -- the pointwise output never reaches HBM (keras Xception
 * middle flow, sepconv -> BN -> relu -> sepconv chains; spnet/models.py:357-359; predict_spnet.py:84-87). */
int spnet_gemm_bf16x3_pp_dwfwd(const void* z_planes, const void* w_planes, int B, int H, int W, int cin, int cout,
                               const float* scale_shift, int relu_next, const float* w_next, void* out_planes, void* stream);
long spnet_gemm_bf16x3_dwbwd_rows(long M);
long spnet_gemm_bf16x3_dwbwd_ok(int H, int W, int cin);
int spnet_gemm_bf16x3_pp_dwbwd(const void* dy_planes, const void* w_planes, int B, int H, int W, int cin, int cout,
                               const float* x_fwd, const float* w, float* dx, int relu_in, const float* add, float* partial,
                               const float* in_scale, const float* in_shift, const float* bn_mean, const float* bn_invstd,
                               float* bn_partial, const float* bn_x, void* stream);
int spnet_gemm_bf16x3_wgrad_batched(const void* jobs, int nbatch, int cin, int cout, int M, int ksplit, void* stream);
/* C += A B, formed in the epilogue (no K split): a data gradient added onto what another consumer of the same tensor
 * has already left in C (the branch convolutions of an inception block; call site spnet/models.py:357-359). */
int spnet_gemm_f32_accumulate(const float* A, int a_major, int lda, const float* B, int b_major, int ldb, float* C,
                              int ldc, int M, int N, int K, int tile, void* stream);

/* ---- fp32 MFMA contractions with fused epilogues / operand forms (product path) ------------------------------------ */
/* spnet_gemm_f32's contraction (no split-K, no bias) that also emits BatchNorm column statistics of C from the
 * accumulators: colstats[rows][2][N] per row-tile (sum, sum of squares), *stat_rows (HOST int) = rows.
 * colstats must hold ceil(M/32)*2*N floats. */
int spnet_gemm_f32_colstats(const float* A, int a_major, int lda, const float* B, int b_major, int ldb,
                            float* C, int ldc, int M, int N, int K, int tile, float* colstats,
                            int* stat_rows, void* stream);
/* nbatch independent problems of ONE shape in one launch, no K split.  A0/B0/C0: operands of problem 0;
 * offsets: DEVICE array {A_b - A0, B_b - B0, C_b - C0} in floats (multiples of 4), b = 0..nbatch-1
 * (the engine's deferred middle-flow weight gradients). */
int spnet_gemm_f32_batched(const float* A0, const float* B0, float* C0, const long long* offsets, int nbatch,
                           int a_major, int lda, int b_major, int ldb, int ldc, int M, int N, int K, int tile,
                           void* stream);
/* ... with every problem's K axis cut into `ksplit` slices: fp32 slabs in `workspace` (nbatch * ksplit * M * N floats),
 * summed per problem in slice order by one more launch (deterministic).  ldc == N.  For many same-shaped weight
 * gradients whose outputs are too small to fill the chip even side by side (the repeated blocks of keras
 * InceptionResNetV2 at batch 16; call site spnet/models.py:357-359).  spnet_gemm_batched_ksplit: the slice count (and
 * tile id) this library would choose for such a batch; 1 = no split. */
long spnet_gemm_batched_ksplit(int M, int N, int K, int nbatch, int* tile_out);
int spnet_gemm_f32_batched_splitk(const float* A0, const float* B0, float* C0, const long long* offsets, int nbatch,
                                  int a_major, int lda, int b_major, int ldb, int ldc, int M, int N, int K, int tile,
                                  int ksplit, float* workspace, long ws_floats, void* stream);

/* dX[M,N] = dY[M,K] B[K,N] where dY = a[k]*g + b[k]*yp + c[k] is the output of a BatchNormalization backward
 * (keras BatchNormalization behind every SeparableConv2D / Conv2D of Xception), blended from the incoming gradient
 * g and the saved pre-normalisation tensor yp while the A tile is staged: dY never makes a pass of its own.
 * coef = [a | b | c], cld floats each (from spnet_bn_bwd_coeffs*; zero beyond channel K-1; cld % 32 == 0,
 * cld >= K rounded up to 32).  B is W^T in the output-major form (spnet_transpose_batched).  dy_out (or NULL)
 * receives dY once, for the layer's weight-gradient GEMM. */
int spnet_gemm_f32_bnblend(const float* g, const float* yp, const float* coef, int cld, int lda, const float* B,
                           int ldb, float* C, int ldc, int M, int N, int K, int tile, float* dy_out, void* stream);

/* block1_conv2 of keras Xception (3x3 VALID stride 1) as implicit GEMMs that gather their operand tiles from the NHWC tensors (no patch matrix):
 * x [B][H][W][cin], w / dw HWIO [3][3][cin][cout], y / dy [B][H-2][W-2][cout].  (cin, cout) = (32, 64).
 * wgrad splits the pixels over workgroups: workspace >= spnet_conv3x3_wgrad_ws() floats. */
int spnet_conv3x3_fwd(const float* x, const float* w, float* y, int B, int H, int W, int cin, int cout,
                      void* stream);
int spnet_conv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int cin, int cout,
                        void* stream);
long spnet_conv3x3_wgrad_ws(int B, int H, int W, int cin, int cout);
int spnet_conv3x3_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int cin, int cout,
                        float* workspace, long ws_floats, void* stream);
/* njobs matrix transposes in one launch: dst_j[c][r] = src_j[r][c]; jobs: DEVICE array of {src, dst, R, C}
 * (four 64-bit words per job).  The engine keeps W^T of the pointwise kernels whose data-gradient product
 * dX = dY W^T (keras SeparableConv2D / Conv2D(1x1), call site spnet/models.py:357-359) runs through
 * spnet_gemm_f32_bnblend, which takes B in the forward (output-major) form; refreshed once per optimizer step. */
int spnet_transpose_batched(const void* jobs, int njobs, int max_rows, int max_cols, void* stream);
/* out[M][ldc] = sum over nslab slabs of M*N floats, in slab order (the second stage of every K split). */
int spnet_reduce_slabs(const float* ws, int nslab, int M, int N, float* out, int ldc, void* stream);
/* Even-pixel gather / scatter-add: TF 'same' 1x1 stride-2 residual convs of Xception blocks 2,3,4,13. */
int spnet_gather_s2(const float* x, float* xs, int B, int H, int W, int C, void* stream);
int spnet_scatter_add_s2(const float* dxs, float* dx, int B, int H, int W, int C, void* stream);

/* ---- depthwise 3x3 / SAME (34 stride-1 layers per Xception forward: keras SeparableConv2D depthwise step;
 *      MobileNet's DepthwiseConv2D, stride 1 | 2).  w is [3][3][C]. ---- */
int spnet_reduce_rows(const float* in, int P, int L, float* out, void* stream);
/* ... with 32*L floats of scratch: many rows (a Conv2D bias gradient: keras InceptionResNetV2's block convs,
 * spnet/models.py:357-359) are folded in 32 parallel slices first. */
int spnet_reduce_rows_ws(const float* in, int P, int L, float* out, float* scratch, long scratch_floats, void* stream);
/* njobs independent row reductions in one launch; jobs: DEVICE array of {in, out, P, L} (four 64-bit words
 * per job), max_L = largest L.  (All depthwise weight gradients of a step: spnet_dwconv3x3_tiled_bwd with
 * dw == NULL leaves its partial sums in the workspace.) */
int spnet_reduce_rows_batched(const void* jobs, int njobs, int max_L, void* stream);
/* Strided depthwise 3x3 / TF-SAME (keras.applications.mobilenet DepthwiseConv2D, strides 1 or 2; spnet/models.py:346-355).
 * op 0: out = y [B][ceil(H/s)][ceil(W/s)][C] from a = x, b = w;  op 1: out = dx [B][H][W][C] from a = dy, b = w;
 * op 2: out = dw [3][3][C] from a = x, b = dy (workspace: spnet_dwconv3x3_strided_ws floats).  H, W: input extent. */
long spnet_dwconv3x3_strided_ws(int B, int H, int W, int C, int stride);
int spnet_dwconv3x3_strided(int op, const float* a, const float* b, float* out, int B, int H, int W, int C, int stride,
                            float* workspace, void* stream);
/* LDS-tiled forms used by the engine: forward, and the FUSED backward (data + weight gradient in one
 * pass over x and dy).  workspace: spnet_dwconv3x3_tiled_bwd_ws(B,H,W,C) floats. */
int spnet_dwconv3x3_tiled_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C,
                              int relu_in, const float* in_scale, const float* in_shift, void* stream);
/* The same with the PRODUCER BatchNorm's training-forward finalize folded into the prologue (one dependent launch less
 * per SeparableConv2D -> BatchNormalization -> SeparableConv2D chain inside keras Xception; spnet/models.py:357-359):
 * partial [rows][2][C] = column sums of x left by spnet_gemm_f32_colstats (rows <= 128), M = pixels they were taken
 * over; the kernel applies relu?(x*scale + shift) with the coefficients it derives (bit-identical to
 * spnet_bn_finalize_fwd), writes save_mean / save_invstd / scale_shift[2C] and updates the moving statistics. */
int spnet_dwconv3x3_tiled_fwd_bnfin(const float* x, const float* w, float* y, int B, int H, int W, int C, int relu_in,
                                    const float* partial, int rows, long M, const float* gamma, const float* beta,
                                    float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                                    float* scale_shift, float eps, float momentum, void* stream);
/* The three forward forms with the output written as the bf16x3 planes of the [B*H*W][C] matrix (see the bf16x3 GEMM
 * block above; y_planes: zeroed allocation of 3 * spnet_bf16x3_plane_elems(B*H*W, C) bf16): the depthwise step of a
 * SeparableConv2D hands its result to the pointwise GEMMs (spnet_gemm_bf16x3_pp forward, spnet_gemm_bf16x3_wgrad_batched)
 * in the form they read, 6 bytes per element, no fp32 copy.  Same arithmetic as the fp32 forms. */
int spnet_dwconv3x3_tiled_fwd_x3(const float* x, const float* w, void* y_planes, int B, int H, int W, int C,
                                 int relu_in, const float* in_scale, const float* in_shift, void* stream);
int spnet_dwconv3x3_tiled_fwd_bnfin_x3(const float* x, const float* w, void* y_planes, int B, int H, int W, int C, int relu_in,
                                       const float* partial, int rows, long M, const float* gamma, const float* beta,
                                       float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                                       float* scale_shift, float eps, float momentum, void* stream);
int spnet_dwconv3x3_stream_fwd_x3(const float* x, const float* w, void* y_planes, int B, int H, int W, int C, int relu_in,
                                  const float* in_scale, const float* in_shift, int rows_per_seg, void* stream);
long spnet_dwconv3x3_tiled_bwd_ws(int B, int H, int W, int C);
long spnet_dwconv3x3_tiled_rows(int B, int H, int W, int C);
/* in_scale/in_shift (or NULL): the producer BatchNorm's affine applied on load (x_fwd is then the PRE-BN
 * tensor); bn_partial (or NULL): also emit that BatchNorm's backward sums [rows][2][C]; bn_x (or NULL =
 * x_fwd): the pre-BN tensor those sums refer to when x_fwd is not it (a block output BN(yp)+residual).
 * dw == NULL: the [rows][9][C] weight-gradient partial sums stay in `workspace` for a later reduction. */
int spnet_dwconv3x3_tiled_bwd(const float* dy, const float* x_fwd, const float* w, float* dx, float* dw,
                              int B, int H, int W, int C, int relu_in, const float* add, float* workspace,
                              const float* in_scale, const float* in_shift, const float* bn_mean,
                              const float* bn_invstd, float* bn_partial, const float* bn_x, void* stream);
/* The same two operations as STREAMING kernels for planes large enough to be pure bandwidth (entry flow at batch 32,
 * every plane of the batch-128 inference plan; call site spnet/models.py:357-359): a wave owns 8 columns x 32 channels
 * and marches down its rows with its loads 3-4 rows ahead in registers; no LDS tile, no barrier, no row halo.
 * rows_per_seg: rows per wave (<= 0: the library's choice); the backward's partial buffers have
 * spnet_dwconv3x3_stream_rows() rows.  H*W*C*4 < 2^31 per image (32-bit buffer offsets).  y and dx are bit-identical
 * to the tiled kernels' (same fmaf order).  spnet_dwconv3x3_prefers_stream: 1 where the streaming form is the faster
 * one for this plane (the engine asks once per layer). */
long spnet_dwconv3x3_prefers_stream(int B, int H, int W, int C, int backward);
long spnet_dwconv3x3_stream_rows(int B, int H, int W, int C, int rows_per_seg);
long spnet_dwconv3x3_stream_bwd_ws(int B, int H, int W, int C, int rows_per_seg);
int spnet_dwconv3x3_stream_fwd(const float* x, const float* w, float* y, int B, int H, int W, int C, int relu_in,
                               const float* in_scale, const float* in_shift, int rows_per_seg, void* stream);
int spnet_dwconv3x3_stream_bwd(const float* dy, const float* x_fwd, const float* w, float* dx, float* dw, int B, int H,
                               int W, int C, int relu_in, const float* add, float* workspace, const float* in_scale,
                               const float* in_shift, const float* bn_mean, const float* bn_invstd, float* bn_partial,
                               const float* bn_x, int rows_per_seg, void* stream);

/* ---- BatchNormalization(axis=-1, momentum .99, eps 1e-3) (spnet/models.py:326-336 + 40 in Xception) --- */
/* act: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 ReLU6 (MobileNet) fused behind the affine; residual (or NULL) added last;
 * res_bcast=1 (C<=4 only): residual holds one value per pixel, broadcast over the channels (the
 * stem's skip connection from the 1-channel input, spnet/models.py:337). */
long spnet_bn_ws(long M, int C);
int spnet_bn_fwd_train(const float* x, long M, int C, const float* gamma, const float* beta,
                       float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                       float* scale_shift, int act, const float* residual, int res_bcast, float* y,
                       float eps, float momentum, float* workspace, void* stream);
int spnet_bn_fwd_infer(const float* x, long M, int C, const float* gamma, const float* beta,
                       const float* moving_mean, const float* moving_var, float* scale_shift, int act,
                       const float* residual, int res_bcast, float* y, float eps, void* stream);
int spnet_bn_bwd(const float* x, const float* dy, long M, int C, const float* gamma, const float* beta,
                 const float* save_mean, const float* save_invstd, int act, float* dx, float* dgamma,
                 float* dbeta, float* coeffs, float* workspace, void* stream);
/* Split forms for fused pipelines: statistics arrive as partial[P][2][C] from a GEMM epilogue
 * (spnet_gemm_f32_colstats) or from the fused depthwise backward.  The three finalize entries CONSUME `partial` when
 * P >= 1024: a first stage leaves 64 slice sums (hi, lo float pairs) in the buffer's own first rows, so the partial rows
 * cannot be read again afterwards (a second consumer must take its copy first). */
int spnet_bn_finalize_fwd(float* partial, int P, long M, int C, const float* gamma, const float* beta,
                          float* moving_mean, float* moving_var, float* save_mean, float* save_invstd,
                          float* scale_shift, float eps, float momentum, void* stream);
int spnet_bn_infer_coeffs(int C, const float* gamma, const float* beta, const float* moving_mean,
                          const float* moving_var, float* scale_shift, float eps, void* stream);
int spnet_bn_apply(const float* x, long M, int C, const float* scale_shift, int act, const float* residual,
                   int res_bcast, float* y, void* stream);
/* spnet_bn_finalize_fwd + spnet_bn_apply (no broadcast residual) as ONE launch while P <= 128 partial rows -- the closing
 * BatchNormalization + Add of a keras Xception middle block in training (spnet/models.py:357-359); results identical. */
int spnet_bn_finalize_apply(float* partial, int P, const float* x, long M, int C, const float* gamma,
                            const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                            float* save_invstd, float* scale_shift, int act, const float* residual, float* y, float eps,
                            float momentum, void* stream);
/* The `_ld` forms of the three forward entries write y with a row stride of ldy floats (>= C, multiples of 4): the output
 * is a column block of a wider tensor -- a conv2d_bn branch of keras InceptionResNetV2 written straight into the buffer
 * of its Concatenate (call site spnet/models.py:357-359), so that no copy pass follows. */
int spnet_bn_finalize_apply_ld(float* partial, int P, const float* x, long M, int C, const float* gamma,
                               const float* beta, float* moving_mean, float* moving_var, float* save_mean,
                               float* save_invstd, float* scale_shift, int act, const float* residual, float* y, long ldy,
                               float eps, float momentum, void* stream);
int spnet_bn_fwd_train_ld(const float* x, long M, int C, const float* gamma, const float* beta, float* moving_mean,
                          float* moving_var, float* save_mean, float* save_invstd, float* scale_shift, int act,
                          const float* residual, int res_bcast, float* y, long ldy, float eps, float momentum,
                          float* workspace, void* stream);
int spnet_bn_fwd_infer_ld(const float* x, long M, int C, const float* gamma, const float* beta, const float* moving_mean,
                          const float* moving_var, float* scale_shift, int act, const float* residual, int res_bcast,
                          float* y, long ldy, float eps, void* stream);
int spnet_bn_bwd_from_partials(const float* x, const float* dy, long M, int C, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd, int P,
                               const float* partial, float* dx, float* dgamma, float* dbeta, float* coeffs,
                               void* stream);
/* spnet_bn_bwd / spnet_bn_bwd_from_partials with dx written as the bf16x3 planes of the [M][C] matrix (dx_planes: zeroed
 * allocation of 3 * spnet_bf16x3_plane_elems(M, C) bf16; C % 4 == 0): the gradient a BatchNormalization hands to the
 * pointwise convolution in front of it is read by that layer's data-gradient and weight-gradient GEMMs only
 * (spnet_gemm_bf16x3_pp, spnet_gemm_bf16x3_wgrad_batched), in this form.  Same arithmetic as the fp32 forms. */
int spnet_bn_bwd_x3(const float* x, const float* dy, long M, int C, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, int act, void* dx_planes, float* dgamma,
                    float* dbeta, float* coeffs, float* workspace, void* stream);
int spnet_bn_bwd_from_partials_x3(const float* x, const float* dy, long M, int C, const float* gamma,
                                  const float* beta, const float* save_mean, const float* save_invstd, int P,
                                  const float* partial, void* dx_planes, float* dgamma, float* dbeta, float* coeffs,
                                  void* stream);

/* Reduction half of the backward only (dgamma, dbeta, blend coefficients for spnet_gemm_f32_bnblend). */
int spnet_bn_bwd_coeffs_from_partials(int P, const float* partial, long M, int C, const float* gamma,
                                      const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                      float* coef, int cld, void* stream);
int spnet_bn_bwd_coeffs(const float* x, const float* dy, long M, int C, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef,
                        int cld, float* workspace, void* stream);

/* ---- pooling --------------------------------------------------------------------------------- */
/* MaxPooling2D(3, strides 2, 'same') + residual add (Xception blocks 2,3,4,13); idx4 packs the argmax
 * tap of 4 channels per uint32 (B*OH*OW*C/4 words) for the backward pass. */
/* x_ss / r_ss (or NULL): [scale|shift] of the BatchNorm producing x / residual, applied on load. */
int spnet_maxpool3x3s2_add_fwd(const float* x, const float* residual, float* y, uint32_t* idx4, int B,
                               int H, int W, int C, const float* x_ss, const float* r_ss, void* stream);
int spnet_maxpool3x3s2_bwd(const float* dy, const uint32_t* idx4, float* dx, int B, int H, int W, int C,
                           void* stream);
/* The same, also emitting the backward sums (sum dx, sum dx*xhat) of the BatchNorm whose un-materialised output was
 * pooled -- yp = its pre-normalisation tensor [B,H,W,C], mean / invstd as saved by its forward -- as
 * partial[rows][2][C], rows = spnet_maxpool3x3s2_bwd_rows(B,H,W,C,max_rows) <= max_rows: the gradient path of block{2,3,4,13}_pool +
 * block*_sepconv2_bn in keras.applications.Xception (call site spnet/models.py:357-359) without a reduction pass. */
long spnet_maxpool3x3s2_bwd_rows(int B, int H, int W, int C, int max_rows);
int spnet_maxpool3x3s2_bwd_bnsums(const float* dy, const uint32_t* idx4, float* dx, int B, int H, int W, int C,
                                  const float* yp, const float* mean, const float* invstd, float* partial, int rows,
                                  void* stream);
/* AveragePooling2D(2) of the stem (spnet/models.py:323,337); any C. */
int spnet_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int spnet_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);

/* ---- keras InceptionResNetV2 (cf.basemodel = 'InceptionResNetV2', spnet/models.py:357-359) ------------------------ */
/* MaxPooling2D(3, strides=2, 'valid') (stem, mixed_6a, mixed_7a); idx4 as above. */
int spnet_maxpool3x3s2_valid_fwd(const float* x, float* y, uint32_t* idx4, int B, int H, int W, int C, void* stream);
int spnet_maxpool3x3s2_valid_bwd(const float* dy, const uint32_t* idx4, float* dx, int B, int H, int W, int C, void* stream);
/* AveragePooling2D(3, strides=1, 'same') (mixed_5b; TF averages over the in-image entries); backward = 1: in = dy, out = dx. */
int spnet_avgpool3x3s1_same(const float* in, float* out, int B, int H, int W, int C, int backward, void* stream);
/* Patch matrix of a KH x KW / stride s / 'same' | 'valid' convolution and its adjoint: backward = 0: out = col
 * [B*OH*OW][KH*KW*C] from in = x; backward = 1: out = dx [B][H][W][C] from in = dcol.  The convolution itself is
 * spnet_gemm_f32 on col and the flattened HWIO kernel (1x1 convs skip the patch matrix). */
int spnet_patches(const float* in, float* out, int B, int H, int W, int C, int KH, int KW, int stride, int same, int backward,
                  void* stream);
/* The forward gather of an input whose pixels are ldx floats apart (a column block of a wider tensor). */
int spnet_patches_ld(const float* in, long ldx, float* out, int B, int H, int W, int C, int KH, int KW, int stride, int same,
                     void* stream);
/* The forward convolution itself on the tuned GEMM kernel (gemm.hip, AG = 1; no patch matrix written or read): the A tile of K step (tap, 32 channels) is the plain
 * K-major fetch from a shifted base with the out-of-image rows zeroed, in the pipeline slots of the matrix form -- no
 * patch matrix, no gather launch, the tile ids / autotuned table of spnet_gemm_f32.  x pixels ldx floats apart (ldx >= C:
 * a column block of a wider tensor), Wk = the HWIO kernel as [KH*KW*C][Cout], y [B*OH*OW][ldy]; C % 32 == 0, stride
 * 1 | 2, TF SAME (same != 0) or VALID; bias / colstats + stat_rows optional (as spnet_gemm_f32_colstats).  Bit-identical
 * to spnet_patches(_ld) + spnet_gemm_f32(_colstats) on the same tile (keras Conv2D k_h x k_w of Inception-ResNet-v2,
 * spnet/models.py:357-359). */
int spnet_conv_gemm_f32(const float* x, long ldx, const float* Wk, float* y, int ldy, int B, int H, int W, int C, int Cout,
                        int KH, int KW, int stride, int same, const float* bias, int tile, float* colstats,
                        int* stat_rows, void* stream);
/* Gradient producers that also leave the BatchNorm-backward sums (sum g, sum g*xhat) of the conv2d_bn layer whose output
 * y = relu(BN(yp)) they differentiate, g masked with y > 0 when relu != 0: partial[rows][2][C], rows <=
 * spnet_grad_bnsums_rows(pixels, max_rows).  spnet_patches_bwd_bnsums = spnet_patches(backward = 1) + mask + sums (the
 * consumer is a k x k convolution); spnet_copy_cols_bnsums = one branch of a Concatenate backward (strided column block
 * -> dense) + mask + sums.  The layer's BatchNorm backward then runs spnet_bn_bwd_from_partials with no activation
 * (keras InceptionResNetV2 conv2d_bn chains; call site spnet/models.py:357-359). */
long spnet_grad_bnsums_rows(long npix, int max_rows);
int spnet_patches_bwd_bnsums(const float* dcol, float* dx, int B, int H, int W, int C, int KH, int KW, int stride, int same,
                             const float* y, const float* yp, const float* mean, const float* invstd, int relu,
                             float* partial, int rows, void* stream);
int spnet_copy_cols_bnsums(const float* src, int lds, float* dst, long M, int C, const float* y, long ldy, const float* yp,
                           const float* mean, const float* invstd, int relu, float* partial, int rows, void* stream);
/* `_ld` forms: row strides for dx (lddx / ldd), yp (ldyp) and the partial rows (ldp floats per half-row) -- the layer may be one
 * member of a group of sibling 1x1 convolutions that share a concatenated pre-normalisation tensor, gradient buffer and
 * partial rows (column blocks of [.][Ct] tensors; IRv2Backbone runs such a group as one GEMM + one BatchNormalization). */
int spnet_patches_bwd_bnsums_ld(const float* dcol, float* dx, long lddx, int B, int H, int W, int C, int KH, int KW, int stride,
                                int same, const float* y, long ldy, const float* yp, long ldyp, const float* mean,
                                const float* invstd, int relu, float* partial, long ldp, int rows, void* stream);
int spnet_copy_cols_bnsums_ld(const float* src, int lds, float* dst, long ldd, long M, int C, const float* y, long ldy,
                              const float* yp, long ldyp, const float* mean, const float* invstd, int relu, float* partial,
                              long ldp, int rows, void* stream);
/* Many strided column-block copies in one launch: jobs (device memory) = njobs x {src, dst, rows, cols, lds, ldd} as six
 * 64-bit words; max_elems = the largest rows*cols (sizes the grid).  The kernels of sibling convolutions gathered into
 * their concatenated GEMM operand after every optimizer step. */
int spnet_copy_cols_batched(const void* jobs, int njobs, long max_elems, void* stream);
/* inception_resnet_block: y = x + scale*up (+ ReLU); backward: dx = g*(y>0 if relu), dup = scale*dx. */
int spnet_resadd(const float* x, const float* up, float* y, long n, float scale, int relu, void* stream);
int spnet_resadd_bwd(const float* y, const float* g, float* dx, float* dup, long n, float scale, int relu, void* stream);
/* dst[r*ldd + c] (+)= src[r*lds + c] for c < cols: Concatenate's channel blocks and gradient accumulation. */
int spnet_copy_cols(const float* src, int lds, float* dst, int ldd, long rows, int cols, int accumulate, void* stream);

/* ---- small-channel direct 3x3 convs: stem conv2d_1..3 (models.py:321,330,335), block1_conv1 ---- */
/* op 0 fwd (a=x,b=w,out=y) | 1 bwd-data (a=dy,b=w,out=dx) | 2 bwd-weight (a=x,b=dy,out=dw).
 * (cin,cout,stride,same) in {(1,3,1,1), (3,3,1,1), (3,32,2,0)}; w is HWIO. */
int spnet_conv3x3_small(int op, int cin, int cout, int stride, int same, const float* a, const float* b,
                        float* out, int B, int H, int W, float* workspace, long ws_floats, void* stream);

/* The stem's first three layers fused (spnet/models.py:321-323, 337): conv2d_1 (1 -> 3, 3x3 'same', no bias) +
 * AveragePooling2D(2), and the skip connection's AveragePooling2D(2) of the input frame.
 * op 0: a = w [3][3][1][3]; out = p1 [B][H/2][W/2][3], out2 = px [B][H/2][W/2].
 * op 2: a = dp1 (gradient of p1); out = dw [3][3][1][3]; workspace >= 512*27 floats.  (No data gradient: x is the input.) */
int spnet_stem_head(int op, const float* x, const float* a, float* out, float* out2, int B, int H, int W,
                    float* workspace, long ws_floats, void* stream);

/* ---- loss / decode ---------------------------------------------------------------------------- */
/* custom_loss + my_loss terms + d/dy_pred (spnet/models.py:557-633).  loss_out[6] =
 * (center,size,angle,noobj,class,total); parts = B*5 floats scratch; grad may be NULL. */
int spnet_ellipse_loss(const float* y_true, const float* y_pred, float* grad, float* parts,
                       float* loss_out, int B, int ncols, int hybrid, void* stream);
/* SelectiveSigmoid (spnet/models.py:277-298) / the sigmoid columns of the 'compound' head (models.py:379-386,
 * after InterleaveColumns the sigmoid outputs sit at columns start::step).  backward == 0: y[:, start::step] =
 * sigmoid(.) in place; backward == 1: grad[:, start::step] *= y (1 - y), y being the post-sigmoid output. */
int spnet_selective_sigmoid(float* y, float* grad, int B, int ncols, int start, int step, int backward, void* stream);
/* denorm_Y + cleanup_antinode_vars angle (spnet/utils.py:186-188,56-64; evaluate_spnet.py:70-73):
 * out[B][ncols/8][7] = (cx,cy,a,b,angle_deg,noobj,rings). */
int spnet_decode(const float* y_norm, const float* means, const float* ranges, float* out, int B,
                 int ncols, int sigmoid_noobj, void* stream);

/* ---- evaluation metric (spnet/diagnostics.py:64-161: compute_iou / calc_map) ----------------------------- */
/* Raster IoU of npairs (predicted, true) ellipse rows [npairs][8] (denormalised cx,cy,a,b,cos2t,sin2t,noobj,
 * rings) on an nx x ny canvas; iou[i] = -1 where the true slot is empty or both rasters are. */
int spnet_ellipse_iou(const float* yp, const float* yt, long npairs, int nx, int ny, double* iou, void* stream);

/* Count metrics (spnet/diagnostics.py:13-59: calc_errors) over [N][ncols] de-normalised grids: counts[7] (int, zeroed
 * by the caller) = ring_miscounts, ring_truecounts, total_obj, false_obj_pos, false_obj_neg, true_obj_pos,
 * true_obj_neg; pix_err[N] = centre error of each row's first predictor. */
int spnet_calc_errors(const float* yp, const float* yt, long N, int ncols, int* counts, float* pix_err, void* stream);

/* ---- optimizer ---------------------------------------------------------------------------------- */
/* Keras Adam + l2 on the first l2_n elements (spnet/models.py:494, 47-71).  sq_scratch >= 2048 floats.
 * mask (or NULL): n floats, 0 = frozen element (layer.trainable=False, spnet/models.py:361-373).
 * lr_t_dev (or NULL): device float overriding lr_t, so a captured hipGraph of the step can be replayed. */
int spnet_adam_step(float* p, const float* g, float* m, float* v, long n, long l2_n, float lr_t,
                    float beta1, float beta2, float eps, float l2, float grad_scale, const float* mask,
                    float* sq_scratch, float* l2_loss_out, const float* lr_t_dev, void* stream);
/* The same step over a RANGE of the flat buffers (pointers to the range's first element, n % 4 == 0, l2_n = how many of
 * its leading elements are l2-regularised): the Dense head's range is updated as soon as its gradient is final, underneath
 * the backbone's backward; the rest when backward has ended.  Leaves spnet_adam_parts(n) sum-of-squares partials in
 * sq_partial; spnet_adam_l2_sum folds the partials of every range of a step into l2_loss_out[0] = l2 * sum(w^2). */
long spnet_adam_parts(long n);
int spnet_adam_part(float* p, const float* g, float* m, float* v, long n, long l2_n, float lr_t, float beta1, float beta2,
                    float eps, float l2, float grad_scale, const float* mask, float* sq_partial, const float* lr_t_dev,
                    void* stream);
int spnet_adam_l2_sum(const float* sq_partial, int count, float l2, float* l2_loss_out, void* stream);

/* ---- input codec on the device (spnet/utils.py:340-342, load_X_one_proc) ------------------------------------------ */
/* uint8 grey levels -> float32 network input, dst = (src / 255 - 0.5) * 2 with numpy's float32 roundings (bit-identical
 * to the host conversion); n pixels, pointers 16-byte aligned.  Lets frames cross PCIe as bytes. */
int spnet_u8_to_input(const unsigned char* src, float* dst, long n, void* stream);
/* Batch assembly: dst[i][0..L) = src[index[i]][0..L), n rows of L floats (16 bytes per lane when L % 4 == 0 and both
 * buffers are 16-byte aligned, 4 otherwise), index int32 | int64 on the device (idx_bytes 4 | 8), clamped to [0, src_rows) -- the minibatch gather Keras' fit does on the host
 * (train_spnet.py:75,81: model.fit(X_train, Y_train, batch_size, shuffle=True)). */
int spnet_gather_rows(const float* src, long src_rows, const void* index, int idx_bytes, float* dst, int n, long L,
                      void* stream);

/* ---- augmentation (spnet/callbacks.py:272-341, spnet/augmentation.py) ---------------------------- */
/* per-frame min/max -> mm[N][2]; scratch: N*32 floats */
int spnet_minmax(const float* x, int N, long hw, float* mm, float* scratch, void* stream);
int spnet_cutout(const float* src, const int* src_index, float* dst, int N, int H, int W,
                 const int* rects, const float* vals, const int* nrect, void* stream);
int spnet_saltpepper(float* x, int N, int H, int W, const int* coords, int n_salt, int n_pepper,
                     const int* flag, const float* mm, void* stream);
/* cv2.GaussianBlur(img, (k,k), 0), k = ksize[n] in {0 = copy, 3, 5, 7} per frame (fixed small-kernel table, reflect-101
 * borders): what blur_inplace (spnet/augmentation.py:66-70) computes and then discards.  dst != src. */
int spnet_gaussian_blur(const float* src, float* dst, int N, int H, int W, const int* ksize, void* stream);
int spnet_warp_affine(const float* src, float* dst, int N, int H, int W, int C, const float* minv,
                      void* stream);
/* cv2.warpAffine as OpenCV computes it for 8-bit images (INTER_LINEAR, zero border; rotate_image / translate_image,
 * spnet/augmentation.py:193-194, 232): 1/32-pixel fixed-point coordinates, 15-bit integer bilinear weights.  src / dst
 * hold 8-bit values as fp32; xrow [N][H][2] = (X0, Y0), xcol [N][W][2] = (adelta, bdelta): the host-rounded row and
 * column terms of the INVERTED matrix, scaled by 1024 (+ round_delta 16 in X0 / Y0). */
int spnet_warp_affine_fixed(const float* src, float* dst, int N, int H, int W, int C, const int* xrow, const int* xcol,
                            void* stream);
/* Dropout(0.1) of the stem (spnet/models.py:340); same call with dy regenerates the mask in backward. */
int spnet_dropout(const float* x, float* y, long n, unsigned seed, float rate, const unsigned* seed_dev,
                  void* stream);   /* seed_dev (or NULL): device uint32 overriding seed (hipGraph replay) */

/* ---- synthetic input (gen_fake_espi.py:60-279 without the band-pass mix-up) -------------------------------- */
/* Rasterises N fake-ESPI frames in device memory from host-drawn parameters: waves [N][5] = amp, wavelength,
 * thickness, slope, spacing (draw_waves :60-80); nodes [N][7][8] = cx, cy, a, b, angle_deg, rings, start, valid
 * (draw_antinodes :145-206, draw_rings :101-114); nnode [N].  noise != 0 adds the clipped N(40,40) noise and the
 * 50 % pixel dropout (gen_images :255-263).  out_f (or NULL): [N][H][W] network input in [-1,1]; out_u8 (or NULL):
 * the uint8 frame. */
int spnet_fake_espi(const float* waves, const float* nodes, const int* nnode, int N, int H, int W, unsigned seed,
                    int noise, float* out_f, unsigned char* out_u8, void* stream);

#ifdef __cplusplus
}
#endif
#endif
