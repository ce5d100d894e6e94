// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3"): the forward, data-gradient and (round 5) weight-gradient
// GEMMs of the pointwise (1x1) convolutions (keras SeparableConv2D's pointwise step and the residual 1x1 convolutions of
// Xception, the pointwise layers of MobileNet; call site spnet/models.py:346-359).
//
// Every fp32 operand is the exact sum of three bf16 numbers, x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// 3 x 8 significant bits = the 24 of fp32), and a product a*b is the sum of nine piece products.  Six of them --
//   ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
// -- carry everything down to 2^-24 of the product (the dropped am*bl, al*bm, al*bl are <= 2^-23.4 relative in sum), each
// piece product is exact in fp32 (8 x 8 bits) and v_mfma_f32_16x16x32_bf16 accumulates in fp32: six bf16 MFMAs (6 x 16
// cycles for a 16x16x32 block) in place of eight fp32 MFMAs (8 x 32 cycles) for a result whose error against float64 is no
// larger than that of the k-ordered fmaf chain of spnet_gemm_f32 (tests/test_kernels_gpu.py), but it is not that chain's bits.
//
// Operand layout (round 5): bf16 planes in 1-KiB pieces, x3t.h.  Round 4's kernel took A as fp32 and split it while staging
// the tile; its time was set by neither the split nor the matrix pipe but by the L2 -> LDS path (knock-out builds,
// tools/x3pp_probe.py: the operand traffic of a 96 x 96 tile pair alone 1.6-1.8 us per K step against 0.86 us of MFMAs,
// because a wave's load touched 16 rows x 64 bytes -- half lines).  With pieces a wave instruction moves eight full lines
// straight into LDS (LDS-DMA: no stage registers, no ds_write, no split in the GEMM at all), and the producers of the
// activations write the planes themselves (dwconv.hip, bn.hip): 6144 x 728 x 728 42-45 -> 31-33 us isolated.
//
//   gemm_bf16x3_pp_kernel      C[M][N] = A planes x B planes^T (forward: A = z, B = split of W^T; data gradient: A = dy,
//                              B = split of W as stored), BatchNorm column sums in the epilogue on request
//   gemm_bf16x3_wgrad_kernel   dW[cin][cout] = sum over pixels z^T dy from the SAME two plane sets, fragments by transposing
//                              LDS reads (ds_read_b64_tr_b16); batched over layers, optional deterministic K split
//   gemm_bf16x3_fwd_kernel     round 4's operand form (A fp32, split in the kernel) on the new B layout, for A operands whose
//                              producer does not write planes (strided residual convolutions, MobileNet)
//   split kernels              W (either operand form) or any fp32 matrix -> planes, one wave per piece
#include "x3t.h"
#include <type_traits>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) bf16x4* lds_b64_t;

#define X3_BM 96
#define X3_BN 96
#define X3_LDR 32            // LDS row stride in bf16 (64 bytes): four 16-byte chunks, XOR-swizzled with s = -(row / 4) mod 4.
                             // A ds_read_b128 is served in four groups of 16 lanes that are NOT consecutive -- {0-3, 12-15,
                             // 20-27}, ... (MI355X_MICROARCH.md, LDS): a group holds rows 0-3, 12-15 at chunk kg and rows 4-11
                             // at chunk kg ^ 1, and rows that are 4 apart share banks, so s(0-3), s(12-15), 1 ^ s(4-7),
                             // 1 ^ s(8-11) must differ: s = 0, 3, 2, 1 by row quad.
#define X3_SWZ(ROW_, CHUNK_) ((((CHUNK_) ^ (0 - ((ROW_) >> 2))) & 3) * 8)
#define X3_PLANE (X3_BM * X3_LDR)
#define X3_STEP (6 * X3_PLANE)          // bf16 elements of one K step in LDS: three A planes, three B planes (36 KB)
#define X3_PIECES 36
#define X3_PER 9                        // pieces per wave and K step
#define X3_NDMA X3_PER                  // (LDS-DMA instructions per wave and K step of the kernel being compiled: the pipeline
                                        // macros below count them; the 192-row kernel redefines it)

// ------------------------------------------------------------------------------------------------ split kernels
// Job j = {src, planes, K, N, sn, sk} (six 64-bit words, device memory): element (n, k) = src[n * sn + k * sk] becomes
// element (row n, column k) of the planes.  Forward form of a Keras pointwise kernel [cin][cout]: K = cin, N = cout, sn = 1,
// sk = cout; data-gradient form (dX = dY W^T): K = cout, N = cin, sn = cout, sk = 1; an activation matrix [R][K]: N = R,
// sn = lda, sk = 1.  One wave per piece: 64 lanes x 16 bytes = the piece's 1 KiB, written contiguously.
__device__ __forceinline__ void x3_split_job(const float* __restrict__ W, unsigned short* __restrict__ planes, int K, long N,
                                             long sn, long sk) {
  const int nk = (K + 31) / 32;
  const long npieces = ((N + 15) / 16) * nk, ps = npieces * 512;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane >> 2, chunk = ((lane & 3) ^ (0 - (r16 >> 2))) & 3;
  const bool vec = sk == 1 && !(sn & 3) && !((uintptr_t)W & 15);
  for (long piece = (long)blockIdx.x * 4 + wave; piece < npieces; piece += (long)gridDim.x * 4) {
    const long n = (piece / nk) * 16 + r16;
    const int k0 = (int)(piece % nk) * 32 + chunk * 8;
    float v[8];
    if (vec && n < N && k0 + 8 <= K) {
      const float4 a = *reinterpret_cast<const float4*>(W + n * sn + k0), b = *reinterpret_cast<const float4*>(W + n * sn + k0 + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (n < N && k0 + e < K) ? W[n * sn + (long)(k0 + e) * sk] : 0.f;
    }
    uint4 h, m, l;
    x3_split_pk(v[0], v[1], h.x, m.x, l.x);
    x3_split_pk(v[2], v[3], h.y, m.y, l.y);
    x3_split_pk(v[4], v[5], h.z, m.z, l.z);
    x3_split_pk(v[6], v[7], h.w, m.w, l.w);
    unsigned short* dst = planes + piece * 512 + lane * 8;
    *reinterpret_cast<uint4*>(dst) = h;
    *reinterpret_cast<uint4*>(dst + ps) = m;
    *reinterpret_cast<uint4*>(dst + 2 * ps) = l;
  }
}
__global__ __launch_bounds__(256) void split_bf16x3_tiled_kernel(const long long* __restrict__ jobs) {
  const long long* jb = jobs + 6 * blockIdx.y;
  x3_split_job(reinterpret_cast<const float*>(jb[0]), reinterpret_cast<unsigned short*>(jb[1]), (int)jb[2], jb[3], jb[4], jb[5]);
}
__global__ __launch_bounds__(256) void split_bf16x3_one_kernel(const float* __restrict__ W, unsigned short* __restrict__ planes,
                                                               int K, long N, long sn, long sk) {
  x3_split_job(W, planes, K, N, sn, sk);
}

// ------------------------------------------------------------------------------------------------ shared pieces
// LDS-DMA of one K step: 36 pieces of 1 KiB (A planes 0..2 x 6 row groups, then B likewise); wave w moves pieces w, w + 4,
// ...: pieces 0-15 are A's, 20-35 B's for every wave, 16-19 (i == 4) depend on the wave -- a scalar select, no branch (a
// branch per piece would cut the K step into basic blocks and nothing could be scheduled between the MFMAs).
#define X3_DMA(SOFF_, KSTRIDE_A_, KSTRIDE_B_, KS_, BUF_)                                                       \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * X3_STEP;                                                     \
    _Pragma("unroll") for (int i = 0; i < X3_PER; ++i) {                                                       \
      const int c = wave + 4 * i;                                                                              \
      const bool is_a = i < 4 ? true : (i > 4 ? false : (c < 18));                                             \
      const __amdgpu_buffer_rsrc_t rs_ = is_a ? rs_a : rs_b;                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(base_ + c * 512), 16, lane * 16,               \
                                               SOFF_[i] + (KS_) * (is_a ? (KSTRIDE_A_) : (KSTRIDE_B_)), 0, 0); \
    }                                                                                                          \
  } while (0)
// smallest terms first: (l,h) (m,m) (h,l), then (m,h) (h,m), then (h,h); the nine tiles of a term back to back, so that
// consecutive MFMAs never wait for each other's accumulator
#define X3_MFMA(FA_, FB_)                                                                                      \
  _Pragma("unroll") for (int t = 0; t < 6; ++t)                                                                \
  _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                \
  _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA_[i][TA[t]], FB_[j][TB[t]], acc[i][j], 0, 0, 0)
// One K step of the software pipeline: at the barrier the LDS-DMA of step KS + 1 (issued a whole step earlier) has landed
// and every wave has its fragments of step KS in registers; then the DMA of step KS + 2 goes into the buffer step KS was
// read out of, the fragments of step KS + 1 are read into the other register set and the 54 MFMAs of step KS run, the
// three kinds interleaved: one DMA per 6 MFMAs, one fragment read per 3 (two per 3 for the transposing reads).
#define X3_BODY(KS_, CA_, CB_, NA_, NB_, DMA_, RD_, READ_, RPG_)                                               \
  do {                                                                                                         \
    __syncthreads(); /* vmcnt(0) lgkmcnt(0) + barrier */                                                       \
    if (DMA_) X3_ISSUE((KS_) + 2, (KS_));                                                                      \
    if (RD_) READ_((KS_) + 1, NA_, NB_);                                                                       \
    X3_MFMA(CA_, CB_);                                                                                         \
    if ((DMA_) && (RD_)) {                                                                                     \
      _Pragma("unroll") for (int g = 0; g < 18; ++g) {                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        if (g % 2 == 0 && g / 2 < X3_NDMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, RPG_, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
#define X3_PIPELINE(NSTEPS_, READ_, RPG_)                                                                      \
  do {                                                                                                         \
    bf16x8 xa[3][3], xb[3][3], ya[3][3], yb[3][3];                                                             \
    X3_ISSUE(0, 0);                                                                                            \
    if ((NSTEPS_) > 1) {                                                                                       \
      X3_ISSUE(1, 1);                                                                                          \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(X3_NDMA) : "memory");                                            \
    } else {                                                                                                   \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    }                                                                                                          \
    __builtin_amdgcn_s_barrier();                                                                              \
    READ_(0, xa, xb);                                                                                          \
    int ks = 0;                                                                                                \
    for (; ks + 3 < (NSTEPS_); ks += 2) {                                                                      \
      X3_BODY(ks, xa, xb, ya, yb, true, true, READ_, RPG_);                                                    \
      X3_BODY(ks + 1, ya, yb, xa, xb, true, true, READ_, RPG_);                                                \
    }                                                                                                          \
    for (; ks < (NSTEPS_); ks += 2) {                                                                          \
      if (ks + 2 < (NSTEPS_)) X3_BODY(ks, xa, xb, ya, yb, true, true, READ_, RPG_);                            \
      else if (ks + 1 < (NSTEPS_)) X3_BODY(ks, xa, xb, ya, yb, false, true, READ_, RPG_);                      \
      else X3_BODY(ks, xa, xb, ya, yb, false, false, READ_, RPG_);                                             \
      if (ks + 1 < (NSTEPS_)) {                                                                                \
        if (ks + 3 < (NSTEPS_)) X3_BODY(ks + 1, ya, yb, xa, xb, true, true, READ_, RPG_);                      \
        else if (ks + 2 < (NSTEPS_)) X3_BODY(ks + 1, ya, yb, xa, xb, false, true, READ_, RPG_);                \
        else X3_BODY(ks + 1, ya, yb, xa, xb, false, false, READ_, RPG_);                                       \
      }                                                                                                        \
    }                                                                                                          \
  } while (0)

// C/D map of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
#define X3_STORE_C(C_, LDC_, ROW0_, COL0_, MROWS_, NCOLS_)                                                     \
  _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                \
  _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                              \
    const int row = (ROW0_) + wm * 48 + i * 16 + kg * 4 + r;                                                   \
    if (row < (MROWS_)) {                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                          \
        const int col = (COL0_) + wn * 48 + j * 16 + p16;                                                      \
        if (col < (NCOLS_)) (C_)[(long)row * (LDC_) + col] = acc[i][j][r];                                     \
      }                                                                                                        \
    }                                                                                                          \
  }

// BatchNorm column sums of this 96-row tile (sum, sum of squares per output column), as spnet_gemm_f32_colstats leaves
// them: colstats[tile row][2][N].  Per lane over its 12 rows, then over the four row groups of the wave (lanes 16
// apart), then over the two waves that share the columns (through LDS: the stage buffers are idle now); fixed order.
#define X3_COLSTATS()                                                                                          \
  do {                                                                                                         \
    __syncthreads();                                                                                           \
    float* sred = reinterpret_cast<float*>(smem); /* [2 sums][2 wm][96 columns] */                             \
    _Pragma("unroll") for (int j = 0; j < 3; ++j) {                                                            \
      float sv = 0.f, qv = 0.f;                                                                                \
      _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                            \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                          \
        const int row = m0 + wm * 48 + i * 16 + kg * 4 + r;                                                    \
        const float v = row < M ? acc[i][j][r] : 0.f;                                                          \
        sv += v;                                                                                               \
        qv = fmaf(v, v, qv);                                                                                   \
      }                                                                                                        \
      sv += __shfl_xor(sv, 16, 64); qv += __shfl_xor(qv, 16, 64);                                              \
      sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64);                                              \
      if (lane < 16) {                                                                                         \
        const int cl = wn * 48 + j * 16 + p16;                                                                 \
        sred[(0 * 2 + wm) * X3_BN + cl] = sv;                                                                  \
        sred[(1 * 2 + wm) * X3_BN + cl] = qv;                                                                  \
      }                                                                                                        \
    }                                                                                                          \
    __syncthreads();                                                                                           \
    if (tid < 2 * X3_BN) {                                                                                     \
      const int q = tid / X3_BN, cl = tid % X3_BN, col = n0 + cl;                                              \
      if (col < N) colstats[((long)tm * 2 + q) * N + col] = sred[(q * 2 + 0) * X3_BN + cl] + sred[(q * 2 + 1) * X3_BN + cl]; \
    }                                                                                                          \
  } while (0)

// ------------------------------------------------------------------------------------------------ planes x planes
// C[M][N] = A x B^T, A = planes of [M][K], B = planes of [N][K] (a_ps / b_ps: elements per plane).  96 x 96 tile, 2 x 2 waves
// of 48 x 48 (3 x 3 MFMA tiles), K step 32, two LDS buffers of 36 KB (two workgroups per CU), two fragment register sets.
// Row groups past the operand's last are clamped (they only reach masked outputs).
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_pp_kernel(const unsigned short* __restrict__ Ap, long a_ps,
                                                                const unsigned short* __restrict__ Bp, long b_ps,
                                                                float* __restrict__ C, int ldc, int M, int N, int nk, int tiles_n,
                                                                float* __restrict__ colstats) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * X3_STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * X3_BM, n0 = tn * X3_BN;
  const int rg_a = (M + 15) / 16, rg_b = (N + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)(3 * a_ps * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)(3 * b_ps * 2), 0x00020000);
  int soff[X3_PER];
#pragma unroll
  for (int i = 0; i < X3_PER; ++i) {
    const int c = wave + 4 * i;
    soff[i] = c < 18 ? (int)(((c / 6) * a_ps + (long)min(m0 / 16 + c % 6, rg_a - 1) * nk * 512) * 2)
                     : (int)((((c - 18) / 6) * b_ps + (long)min(n0 / 16 + (c - 18) % 6, rg_b - 1) * nk * 512) * 2);
  }
#define X3_ISSUE(KS_, BUF_) X3_DMA(soff, 1024, 1024, KS_, BUF_)
  f32x4v acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  const int a_off = (wm * 48 + p16) * X3_LDR + X3_SWZ(p16, kg);
  const int b_off = (3 * X3_BM + wn * 48 + p16) * X3_LDR + X3_SWZ(p16, kg);
#define X3_READ_ROWS(KS_, FA_, FB_)                                                                            \
  do {                                                                                                         \
    const unsigned short* base_ = smem + ((KS_) & 1) * X3_STEP;                                                \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                            \
      FA_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + a_off + (p * X3_BM + t * 16) * X3_LDR);             \
      FB_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + b_off + (p * X3_BN + t * 16) * X3_LDR);             \
    }                                                                                                          \
  } while (0)
  X3_PIPELINE(nk, X3_READ_ROWS, 1);
#undef X3_ISSUE
  X3_STORE_C(C, ldc, m0, n0, M, N);
  if (colstats) X3_COLSTATS();
}

// ------------------------------------------------------------------------------------------------ data gradient + depthwise backward
// The data-gradient GEMM of a separable convolution's pointwise step with the FUSED BACKWARD OF ITS DEPTHWISE STEP in the
// epilogue (round 5): dz = dy W^T never reaches HBM and the depthwise-backward launch disappears.
// A workgroup owns 192 pixels x 96 channels of dz -- whole images of the plane (12 x 16: one, 6 x 8: four), so the 3 x 3
// neighbourhoods need no halo from another workgroup and, channels being independent, nothing is recomputed per column
// tile.  Main loop: gemm_bf16x3_pp_kernel with a 192-row A tile (8 waves of 48 x 48, one workgroup per CU, 2 x 54 KB of
// LDS, 7 LDS-DMA pieces per wave and K step).  Epilogue: the accumulators go to LDS as an fp32 tile [192][100]; a thread
// then owns one channel quad and 12 pixels (a column of a 12-row image, or two columns of 6-row ones), fetches x (the
// depthwise input: for the ReLU mask and the tap sums), the residual-branch gradient `add` and bn_x up front (the GEMM's
// registers are dead: 36 loads in flight), and computes exactly what dw3x3_tile_bwd_kernel (dwconv.hip) computes:
//   dx[q] = (sum_d dz[q-d] k[d]) * relu'(x*scale + shift) + add        (its fmaf order: dx is bit-identical)
//   dk[d] partial sums over the tile's pixels, the producer BatchNorm's (sum dx, sum dx*xhat) partial sums
// one partial row per workgroup row (ceil(M / 192) rows, [rows][9][C] and [rows][2][C]), reduced over the 16 pixel groups
// through LDS in a fixed order.  Planes: 12 x 16 and 6 x 8 (Xception's middle and exit flow at 384 x 512 frames).
#undef X3_NDMA
#define X3_NDMA 7
#define FB_BM 192
#define FB_STEP (3 * (FB_BM + X3_BN) * X3_LDR)      // bf16 elements of one K step: 54 KB
#define FB_LD 100                                   // fp32 row stride of the dz tile in LDS
// MODE 1 (inference): the FORWARD GEMM of a pointwise convolution with, in its epilogue, the BatchNorm affine + ReLU of its
// output and the depthwise step of the NEXT separable convolution -- the tile holds whole images, so
// z'[q] = sum_d relu(C[q+d] * scale + shift) k'[d] comes straight out of the C tile in LDS and is written as the bf16x3
// planes the next pointwise GEMM reads: the pre-normalisation tensor never reaches HBM, the next unit's depthwise launch
// disappears (wt = that depthwise kernel, in_scale / in_shift = this layer's folded BatchNorm, relu_in = the ReLU in
// front of the next depthwise; dw3x3_stream_fwd_kernel's fmaf order: the planes are bit-identical to the two launches').
template <int MODE, bool ADD, bool BNX>
__global__ __launch_bounds__(512, 2) void gemm_bf16x3_pp_dwbwd_kernel(
    const unsigned short* __restrict__ Ap, long a_ps, const unsigned short* __restrict__ Bp, long b_ps, int M, int N, int nk,
    int tiles_n, int H, int W, const float* __restrict__ x, const float* __restrict__ wt, float* __restrict__ dx, int relu_in,
    const float* __restrict__ add, float* __restrict__ partial, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd,
    float* __restrict__ bn_partial, const float* __restrict__ bn_x, unsigned short* __restrict__ out_planes, long out_ps) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * FB_STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * FB_BM, n0 = tn * X3_BN;
  const int rg_a = (M + 15) / 16, rg_b = (N + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)(3 * a_ps * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)(3 * b_ps * 2), 0x00020000);
  // 54 pieces per K step: A planes 0..2 x 12 row groups (0..35), B planes x 6 (36..53); wave w moves pieces w + 8 i
  // (i = 0..6; the two surplus slots repeat piece 53: same bytes): pieces w + 8 i are A's for i <= 3, B's for i >= 5
  int soff[X3_NDMA];
#pragma unroll
  for (int i = 0; i < X3_NDMA; ++i) {
    const int c = min(wave + 8 * i, 53);
    soff[i] = c < 36 ? (int)(((c / 12) * a_ps + (long)min(m0 / 16 + c % 12, rg_a - 1) * nk * 512) * 2)
                     : (int)((((c - 36) / 6) * b_ps + (long)min(n0 / 16 + (c - 36) % 6, rg_b - 1) * nk * 512) * 2);
  }
#define X3_ISSUE(KS_, BUF_)                                                                                    \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * FB_STEP;                                                     \
    _Pragma("unroll") for (int i = 0; i < X3_NDMA; ++i) {                                                      \
      const int c = min(wave + 8 * i, 53);                                                                     \
      const bool is_a = i < 4 ? true : (i > 4 ? false : (c < 36));                                             \
      const __amdgpu_buffer_rsrc_t rs_ = is_a ? rs_a : rs_b;                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(base_ + c * 512), 16, lane * 16,               \
                                               soff[i] + (KS_) * 1024, 0, 0);                                  \
    }                                                                                                          \
  } while (0)
  f32x4v acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  const int a_off = (wm * 48 + p16) * X3_LDR + X3_SWZ(p16, kg);
  const int b_off = (3 * FB_BM + wn * 48 + p16) * X3_LDR + X3_SWZ(p16, kg);
#define FB_READ_ROWS(KS_, FA_, FB_)                                                                            \
  do {                                                                                                         \
    const unsigned short* base_ = smem + ((KS_) & 1) * FB_STEP;                                                \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                            \
      FA_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + a_off + (p * FB_BM + t * 16) * X3_LDR);             \
      FB_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + b_off + (p * X3_BN + t * 16) * X3_LDR);             \
    }                                                                                                          \
  } while (0)
  X3_PIPELINE(nk, FB_READ_ROWS, 1);
#undef X3_ISSUE

  // ---- epilogue: dz tile -> LDS, then the depthwise backward over it
#ifdef FB_SKIP_EPILOGUE       // (diagnostic build, tools/dwfuse_bwd_time.py: the main loop alone; results wrong)
  {
    float keep = 0.f;       // (every accumulator stays live: a dead one would take its MFMAs with it)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) keep += acc[i][j][r];
    if (keep == 1.2345f) dx[0] = keep;
    return;
  }
#endif
  if constexpr (MODE == 1) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 zero = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    float* ct = reinterpret_cast<float*>(smem);
    float scv[3], shv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int col = min(n0 + wn * 48 + j * 16 + p16, N - 1);
      scv[j] = in_scale ? in_scale[col] : 1.f;
      shv[j] = in_scale ? in_shift[col] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float v = acc[i][j][r];
          if (in_scale) v = fmaf(v, scv[j], shv[j]);
          if (relu_in) v = fmaxf(v, 0.f);
          ct[(wm * 48 + i * 16 + kg * 4 + r) * FB_LD + wn * 48 + j * 16 + p16] = v;
        }
    const int cq = tid & 31, pg = tid >> 5;
    const int c = n0 + cq * 4;
    const bool active = cq < 24 && c < N;
    const int nk_out = (N + 31) >> 5;
    auto walk_fwd = [&](auto ph_tag, auto pw_tag) {
      constexpr int PH = decltype(ph_tag)::value, PW = decltype(pw_tag)::value, HW = PH * PW;
      constexpr int WALKS = (FB_BM / PH) / 16;
      v4 k[9];
      const int cc = active ? c : 0;
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) k[tp] = *reinterpret_cast<const v4*>(wt + (long)tp * N + cc);
      __syncthreads();                                  // the activated tile is complete
#define FB_FMA(D_, A_, B_) D_ = v4{fmaf(A_.x, B_.x, D_.x), fmaf(A_.y, B_.y, D_.y), fmaf(A_.z, B_.z, D_.z), fmaf(A_.w, B_.w, D_.w)}
#pragma unroll
      for (int wk = 0; wk < WALKS; ++wk) {
        const int col = pg + 16 * wk, img = col / PW, w = col % PW;
        const int base = img * HW + w;
        const bool ok = active && m0 + img * HW < M;
        const float* col0 = ct + base * FB_LD + cq * 4;
        const int ol = w > 0 ? -FB_LD : 0, orr = w < PW - 1 ? FB_LD : 0;
        const float fl = w > 0 ? 1.f : 0.f, fr = w < PW - 1 ? 1.f : 0.f;
        v4 a_prev = zero, a_cur = zero;
        // input row r feeds output rows r + 1 (taps 0-2), r (3-5), r - 1 (6-8): dw3x3_stream_fwd_kernel's march, zero rows
        // above and below the image included
#pragma unroll
        for (int r = -1; r <= PH; ++r) {
          v4 v0 = zero, v1 = zero, v2 = zero;
          if (r >= 0 && r < PH) {
            const float* rp = col0 + r * PW * FB_LD;
            v0 = *reinterpret_cast<const v4*>(rp + ol) * fl;
            v1 = *reinterpret_cast<const v4*>(rp);
            v2 = *reinterpret_cast<const v4*>(rp + orr) * fr;
          }
          FB_FMA(a_prev, v0, k[6]); FB_FMA(a_prev, v1, k[7]); FB_FMA(a_prev, v2, k[8]);
          FB_FMA(a_cur, v0, k[3]); FB_FMA(a_cur, v1, k[4]); FB_FMA(a_cur, v2, k[5]);
          v4 a_next = v0 * k[0];
          FB_FMA(a_next, v1, k[1]); FB_FMA(a_next, v2, k[2]);
          const int t = r - 1;
          if (t >= 0 && t < PH && ok)
            x3t_store4(out_planes, out_ps, x3t_off(m0 + base + t * PW, c, nk_out), a_prev.x, a_prev.y, a_prev.z, a_prev.w);
          a_prev = a_cur;
          a_cur = a_next;
        }
      }
#undef FB_FMA
    };
    if (H == 12) walk_fwd(std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
    else walk_fwd(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
    return;
  }
  __syncthreads();
  float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        ct[(wm * 48 + i * 16 + kg * 4 + r) * FB_LD + wn * 48 + j * 16 + p16] = acc[i][j][r];
  const int cq = tid & 31, pg = tid >> 5;               // channel quad of the tile (24 of 32 lanes work), pixel group
  const int c = n0 + cq * 4;                            // first channel of the quad
  const bool active = cq < 24 && c < N;
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 zero = {0.f, 0.f, 0.f, 0.f};
  v4 accw[9], bsg = zero, bsgx = zero;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) accw[tp] = zero;
  // A thread walks down image columns: 12 pixels in all (one column of a 12 x 16 image, or one column each of two 6 x 8
  // images) -- the plane is a template parameter, so every index below is a constant or a shift, the 3 x 3 window slides
  // (3 LDS reads per pixel) and nothing branches: the epilogue runs at 2 waves per SIMD and is instruction-bound otherwise.
  auto walk_columns = [&](auto ph_tag, auto pw_tag) {
    constexpr int PH = decltype(ph_tag)::value, PW = decltype(pw_tag)::value, HW = PH * PW;
    constexpr int WALKS = (FB_BM / PH) / 16;
    static_assert(PH * WALKS == 12, "12 pixels per thread");
    v4 xq[12], aq[ADD ? 12 : 1], bq[BNX ? 12 : 1];
    int base[WALKS];
    bool ok[WALKS], lft[WALKS], rgt[WALKS];
#pragma unroll
    for (int wk = 0; wk < WALKS; ++wk) {
      const int col = pg + 16 * wk, img = col / PW, w = col % PW;
      base[wk] = img * HW + w;                          // tile-local pixel of row 0 of this column
      ok[wk] = active && m0 + img * HW < M;             // (M is a multiple of H*W: an image is inside or outside as a whole)
      lft[wk] = w > 0;
      rgt[wk] = w < PW - 1;
#pragma unroll
      for (int h = 0; h < PH; ++h) {
        const long o = (long)(m0 + (ok[wk] ? base[wk] + h * PW : 0)) * N + (active ? c : 0);
        xq[wk * PH + h] = *reinterpret_cast<const v4*>(x + o);
        if (ADD) aq[wk * PH + h] = *reinterpret_cast<const v4*>(add + o);
        if (BNX) bq[wk * PH + h] = *reinterpret_cast<const v4*>(bn_x + o);
      }
    }
    v4 k[9], sc = {1.f, 1.f, 1.f, 1.f}, sh = zero, mu = zero, is = zero;
    const int cc = active ? c : 0;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) k[tp] = *reinterpret_cast<const v4*>(wt + (long)tp * N + cc);
    const bool affine = in_scale != nullptr;
    if (affine) { sc = *reinterpret_cast<const v4*>(in_scale + cc); sh = *reinterpret_cast<const v4*>(in_shift + cc); }
    if (bn_partial) { mu = *reinterpret_cast<const v4*>(bn_mean + cc); is = *reinterpret_cast<const v4*>(bn_invstd + cc); }
    __syncthreads();                                    // the dz tile is complete
#define FB_FMA(D_, A_, B_) D_ = v4{fmaf(A_.x, B_.x, D_.x), fmaf(A_.y, B_.y, D_.y), fmaf(A_.z, B_.z, D_.z), fmaf(A_.w, B_.w, D_.w)}
#pragma unroll
    for (int wk = 0; wk < WALKS; ++wk) {
      const float* col0 = ct + base[wk] * FB_LD + cq * 4;       // row 0 of the column; the neighbours 1 pixel = FB_LD floats away
      const int ol = lft[wk] ? -FB_LD : 0, orr = rgt[wk] ? FB_LD : 0;
      const float fl = lft[wk] ? 1.f : 0.f, fr = rgt[wk] ? 1.f : 0.f;
      v4 m0_ = zero, m1 = zero, m2 = zero;                        // row h - 1 (zeros above the image)
      v4 c0 = *reinterpret_cast<const v4*>(col0 + ol) * fl, c1 = *reinterpret_cast<const v4*>(col0),
         c2 = *reinterpret_cast<const v4*>(col0 + orr) * fr;
#pragma unroll
      for (int h = 0; h < PH; ++h) {
        v4 n0_ = zero, n1 = zero, n2 = zero;                      // row h + 1 (zeros below the image)
        if (h + 1 < PH) {
          const float* rn = col0 + (h + 1) * PW * FB_LD;
          n0_ = *reinterpret_cast<const v4*>(rn + ol) * fl;
          n1 = *reinterpret_cast<const v4*>(rn);
          n2 = *reinterpret_cast<const v4*>(rn + orr) * fr;
        }
        const v4 raw = xq[wk * PH + h];
        v4 a = raw;
        if (affine) a = v4{fmaf(raw.x, sc.x, sh.x), fmaf(raw.y, sc.y, sh.y), fmaf(raw.z, sc.z, sh.z), fmaf(raw.w, sc.w, sh.w)};
        const v4 xin = relu_in ? v4{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)} : a;
        v4 res = n2 * k[0];                                       // (the tile kernel's order: dx comes out bit-identical)
        FB_FMA(res, n1, k[1]); FB_FMA(res, n0_, k[2]);
        FB_FMA(res, c2, k[3]); FB_FMA(res, c1, k[4]); FB_FMA(res, c0, k[5]);
        FB_FMA(res, m2, k[6]); FB_FMA(res, m1, k[7]); FB_FMA(res, m0_, k[8]);
        if (relu_in) res = v4{a.x > 0.f ? res.x : 0.f, a.y > 0.f ? res.y : 0.f, a.z > 0.f ? res.z : 0.f, a.w > 0.f ? res.w : 0.f};
        if (ADD) res += aq[wk * PH + h];
        if (ok[wk]) {
          FB_FMA(accw[0], xin, n2); FB_FMA(accw[1], xin, n1); FB_FMA(accw[2], xin, n0_);
          FB_FMA(accw[3], xin, c2); FB_FMA(accw[4], xin, c1); FB_FMA(accw[5], xin, c0);
          FB_FMA(accw[6], xin, m2); FB_FMA(accw[7], xin, m1); FB_FMA(accw[8], xin, m0_);
          *reinterpret_cast<v4*>(dx + (long)(m0 + base[wk] + h * PW) * N + c) = res;
          if (bn_partial) {                             // res = dL/d(BN output of the producer); xhat from the pre-BN value
            const v4 pre = BNX ? bq[wk * PH + h] : raw;
            bsg += res;
            bsgx = v4{fmaf(res.x, (pre.x - mu.x) * is.x, bsgx.x), fmaf(res.y, (pre.y - mu.y) * is.y, bsgx.y),
                      fmaf(res.z, (pre.z - mu.z) * is.z, bsgx.z), fmaf(res.w, (pre.w - mu.w) * is.w, bsgx.w)};
          }
        }
        m0_ = c0; m1 = c1; m2 = c2;
        c0 = n0_; c1 = n1; c2 = n2;
      }
    }
  };
  if (H == 12) walk_columns(std::integral_constant<int, 12>{}, std::integral_constant<int, 16>{});
  else walk_columns(std::integral_constant<int, 6>{}, std::integral_constant<int, 8>{});
  // the 9 tap sums (+ the 2 BatchNorm sums) over the 16 pixel groups of each channel quad, fixed order
  __syncthreads();
  v4* red = reinterpret_cast<v4*>(smem);                // [11][16 groups][32 quads]
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) red[(tp * 16 + pg) * 32 + cq] = accw[tp];
  red[(9 * 16 + pg) * 32 + cq] = bsg;
  red[(10 * 16 + pg) * 32 + cq] = bsgx;
  __syncthreads();
  for (int q = tid; q < 11 * 32; q += 512) {
    const int tp = q / 32, ll = q % 32, co = n0 + ll * 4;
    if (ll < 24 && co < N && (tp < 9 || bn_partial)) {
      v4 sum = red[(tp * 16) * 32 + ll];
      for (int g = 1; g < 16; ++g) sum += red[(tp * 16 + g) * 32 + ll];
      if (tp < 9) *reinterpret_cast<v4*>(partial + ((long)tm * 9 + tp) * N + co) = sum;
      else *reinterpret_cast<v4*>(bn_partial + ((long)tm * 2 + (tp - 9)) * N + co) = sum;
    }
  }
}
#undef X3_NDMA
#define X3_NDMA X3_PER

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[cin][cout] = sum over pixels m of z[m][cin] dy[m][cout], both operands the planes of [M][cin] / [M][cout] (the planes
// the forward / data-gradient launches read row-wise).  A 96 x 96 tile of dW per workgroup, contraction steps of 32 pixels =
// two 16-row groups: a step's LDS image holds, per operand and plane, 3 channel chunks x 2 row groups = 6 pieces (the same
// 36 KB).  An MFMA operand needs 8 consecutive PIXELS of one channel per lane -- a column of the image -- which
// ds_read_b64_tr_b16 delivers (4 rows x 16 columns per 16 lanes, transposed; two reads per fragment; conflict-free under
// the piece's swizzle: the two 16-lane groups of a half wave read rows 8 apart = the two 32-byte halves of the same
// 64-byte segments).  Job b = {z planes, dy planes, dst} (three 64-bit words).  Slice s of `ksplit` covers steps
// [s * per, (s + 1) * per) and writes dst + s * cin * cout (slabs for spnet_reduce_slabs) -- deterministic either way.
// One-dimensional grid of ksplit x nbatch x tiles workgroups, XCD-aware: the workgroups the dispatcher deals to one XCD
// (id % 8) get a CONTIGUOUS range of (slice, job, tile) triples, tile fastest -- every tile of a job's slice streams the
// same z and dy pixels, so they must share one L2.  (Round 5, PMC: with the remap applied per job -- eight tiles of each
// job on every XCD -- each XCD fetched all of dy and the launch read 4.7 x its algorithmic bytes past the L2, 6 TB/s.)
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_wgrad_kernel(const long long* __restrict__ jobs, int cin, int cout, int M,
                                                                   int tiles_n, int tiles, int nbatch, int steps_per_slice) {
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * X3_STEP];
  const int wid = xcd_remap(blockIdx.x, gridDim.x);
  const int lid = wid % tiles, job = (wid / tiles) % nbatch, slice = wid / (tiles * nbatch);
  const long long* jb = jobs + 3 * job;
  const unsigned short* __restrict__ Zp = reinterpret_cast<const unsigned short*>(jb[0]);
  const unsigned short* __restrict__ Gp = reinterpret_cast<const unsigned short*>(jb[1]);
  float* __restrict__ dst = reinterpret_cast<float*>(jb[2]) + (long)slice * cin * cout;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * X3_BM, n0 = tn * X3_BN;             // first cin / cout of the tile
  const int nkz = (cin + 31) / 32, nkg = (cout + 31) / 32, rgm = (M + 15) / 16;
  const long z_ps = (long)rgm * nkz * 512, g_ps = (long)rgm * nkg * 512;
  const int nsteps_all = (M + 31) / 32;
  const int s_beg = slice * steps_per_slice, nsteps = min(steps_per_slice, nsteps_all - s_beg);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Zp, 0, (int)(3 * z_ps * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Gp, 0, (int)(3 * g_ps * 2), 0x00020000);
  // piece c of an operand: plane c / 6, channel chunk (c % 6) / 2, row group half c % 2
  int soff[X3_PER];
#pragma unroll
  for (int i = 0; i < X3_PER; ++i) {
    const int c = wave + 4 * i;
    const int q = c < 18 ? c : c - 18;
    const int plane = q / 6, j = (q % 6) / 2, half = q % 2;
    soff[i] = c < 18 ? (int)((plane * z_ps + ((long)(2 * s_beg + half) * nkz + min(m0 / 32 + j, nkz - 1)) * 512) * 2)
                     : (int)((plane * g_ps + ((long)(2 * s_beg + half) * nkg + min(n0 / 32 + j, nkg - 1)) * 512) * 2);
  }
  // (M % 32 == 16: the last step's second row group does not exist -- its pieces come from beyond the descriptor's range:
  // zeros)
  const bool odd_tail = (rgm & 1) && s_beg + nsteps == nsteps_all;
  const int kst_a = nkz * 2048, kst_b = nkg * 2048;       // bytes from one step's piece to the next: two row groups
#define X3_ISSUE(KS_, BUF_)                                                                                    \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * X3_STEP;                                                     \
    const bool tail_ = odd_tail && (KS_) == nsteps - 1;                                                        \
    _Pragma("unroll") for (int i = 0; i < X3_PER; ++i) {                                                       \
      const int c = wave + 4 * i;                                                                              \
      const bool is_a = i < 4 ? true : (i > 4 ? false : (c < 18));                                             \
      const __amdgpu_buffer_rsrc_t rs_ = is_a ? rs_a : rs_b;                                                   \
      const int so_ = (tail_ && (c & 1)) ? 0x7ffffff0 : soff[i] + (KS_) * (is_a ? kst_a : kst_b);              \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(base_ + c * 512), 16, lane * 16, so_, 0, 0);   \
    }                                                                                                          \
  } while (0)
  f32x4v acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  // transposing fragment reads: lane 4q + p of a 16-lane group supplies the address of row q, columns 4p .. 4p + 3 of a
  // 4 x 16 block and receives column (lane % 16) of the four rows.  Pixels 8 kg .. 8 kg + 7 of the step: row group half
  // kg / 2, rows 8 (kg & 1) + {0..3} and + {4..7}; channel tile ct of the wave: chunk piece ct / 2, columns 16 (ct % 2) ...
  const int tq = p16 >> 2, tp = p16 & 3;
  int zaddr[3][2], gaddr[3][2];                           // element offsets inside a K step's image, plane 0
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int r16 = 8 * (kg & 1) + 4 * h2 + tq;
      const int ctz = 3 * wm + t, ctg = 3 * wn + t;
      const int cz = 2 * (ctz & 1) + (tp >> 1), cg = 2 * (ctg & 1) + (tp >> 1);
      zaddr[t][h2] = ((ctz >> 1) * 2 + (kg >> 1)) * 512 + r16 * 32 + X3_SWZ(r16, cz) + (tp & 1) * 4;
      gaddr[t][h2] = 18 * 512 + ((ctg >> 1) * 2 + (kg >> 1)) * 512 + r16 * 32 + X3_SWZ(r16, cg) + (tp & 1) * 4;
    }
#define X3_TR(PTR_) __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_b64_t)(PTR_))
#define X3_READ_COLS(KS_, FA_, FB_)                                                                            \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((KS_) & 1) * X3_STEP;                                                      \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                            \
      const bf16x4 a0_ = X3_TR(base_ + p * 6 * 512 + zaddr[t][0]), a1_ = X3_TR(base_ + p * 6 * 512 + zaddr[t][1]);   \
      const bf16x4 b0_ = X3_TR(base_ + p * 6 * 512 + gaddr[t][0]), b1_ = X3_TR(base_ + p * 6 * 512 + gaddr[t][1]);   \
      FA_[t][p] = __builtin_shufflevector(a0_, a1_, 0, 1, 2, 3, 4, 5, 6, 7);                                   \
      FB_[t][p] = __builtin_shufflevector(b0_, b1_, 0, 1, 2, 3, 4, 5, 6, 7);                                   \
    }                                                                                                          \
  } while (0)
  X3_PIPELINE(nsteps, X3_READ_COLS, 2);
#undef X3_ISSUE
  X3_STORE_C(dst, cout, m0, n0, cin, cout);
}

// ------------------------------------------------------------------------------------------------ fp32 A, split in the kernel
// Round 4's kernel on the new B layout: A [M][K] fp32 (activations whose producer does not write planes), split on the fly
// while the tile is staged; B = planes.  Two register sets so that a K step is in flight for a whole iteration, its split
// and LDS stores issued in the shadow of the MFMAs, one barrier per step.
// B slot s (0 .. 1151) = (plane s / 384, piece (s % 384) / 64, 16-byte position s % 64): a thread owns slots tid + 256 i,
// a wave's load = 1 KiB of contiguous memory = a piece, stored to LDS at the same position.
#define X3_BSRC(S_) (Bp + ((S_) / 384) * plane_stride + ((long)min(n0 / 16 + ((S_) % 384) / 64, rg_b - 1) * nk + ks_) * 512 + ((S_) % 64) * 8)
#define X3_BDST(S_) (base_ + (3 + (S_) / 384) * X3_PLANE + ((S_) % 384) * 8)

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_fwd_kernel(const float* __restrict__ A, int lda,
                                                                 const unsigned short* __restrict__ Bp,
                                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                                 int tiles_n, float* __restrict__ colstats) {
  // [buffer][A planes 0..2 | B planes 0..2][row][X3_LDR]
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * X3_STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * X3_BM, n0 = tn * X3_BN;
  const int nk = (K + 31) / 32, rg_b = (N + 15) / 16;
  const long plane_stride = (long)rg_b * nk * 512;

  // global fetch slots.  A: 96 rows x 8 float4 = 768 slots, 3 per thread (row = s / 8, k quad = s % 8).
  // two register sets (x, y): a K step stays in flight for a whole iteration before it is split and stored
  float4 ax0, ax1, ax2, ay0, ay1, ay2;
  uint4 bx0, bx1, bx2, bx3, bx4, by0, by1, by2, by3, by4;
  const int ar0 = min(m0 + tid / 8, M - 1), ar1 = min(m0 + (tid + 256) / 8, M - 1), ar2 = min(m0 + (tid + 512) / 8, M - 1);
  const int akq = (tid % 8) * 4;                   // (256 % 8 == 0: the same k quad for the three slots)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int s4 = tid < 128 ? tid + 1024 : tid;     // slot 4 exists for half the threads: the others re-read their slot 0
#define X3_FETCH(KS_, S_)                                                                                      \
  do {                                                                                                         \
    const int ks_ = (KS_);                                                                                     \
    const int kk_ = min(ks_ * 32 + akq, K - 4); /* always a load from global memory, zeroed in the stage */    \
    a##S_##0 = *reinterpret_cast<const float4*>(A + (long)ar0 * lda + kk_);                                    \
    a##S_##1 = *reinterpret_cast<const float4*>(A + (long)ar1 * lda + kk_);                                    \
    a##S_##2 = *reinterpret_cast<const float4*>(A + (long)ar2 * lda + kk_);                                    \
    b##S_##0 = *reinterpret_cast<const uint4*>(X3_BSRC(tid));                                                  \
    b##S_##1 = *reinterpret_cast<const uint4*>(X3_BSRC(tid + 256));                                            \
    b##S_##2 = *reinterpret_cast<const uint4*>(X3_BSRC(tid + 512));                                            \
    b##S_##3 = *reinterpret_cast<const uint4*>(X3_BSRC(tid + 768));                                            \
    b##S_##4 = *reinterpret_cast<const uint4*>(X3_BSRC(s4));                                                   \
  } while (0)
#define X3_SPLIT_STORE(AV_, S_)                                                                                \
  do {                                                                                                         \
    unsigned h0_, m0_, l0_, h1_, m1_, l1_;                                                                     \
    const float4 v_ = kok_ ? (AV_) : zero4;                                                                    \
    x3_split_pk(v_.x, v_.y, h0_, m0_, l0_);                                                                    \
    x3_split_pk(v_.z, v_.w, h1_, m1_, l1_);                                                                    \
    const int o_ = ((S_) / 8) * X3_LDR + X3_SWZ((S_) / 8, akq / 8) + (akq & 4);                                \
    *reinterpret_cast<uint2*>(base_ + 0 * X3_PLANE + o_) = make_uint2(h0_, h1_);                               \
    *reinterpret_cast<uint2*>(base_ + 1 * X3_PLANE + o_) = make_uint2(m0_, m1_);                               \
    *reinterpret_cast<uint2*>(base_ + 2 * X3_PLANE + o_) = make_uint2(l0_, l1_);                               \
  } while (0)
#define X3_STAGE(KS_, S_, ALLK_)                                                                               \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((KS_) & 1) * X3_STEP;                                                      \
    const bool kok_ = (ALLK_) || (KS_) * 32 + akq < K; /* (the zeroing is for the last K step alone) */        \
    X3_SPLIT_STORE(a##S_##0, tid);                                                                             \
    X3_SPLIT_STORE(a##S_##1, tid + 256);                                                                       \
    X3_SPLIT_STORE(a##S_##2, tid + 512);                                                                       \
    *reinterpret_cast<uint4*>(X3_BDST(tid)) = b##S_##0;                                                        \
    *reinterpret_cast<uint4*>(X3_BDST(tid + 256)) = b##S_##1;                                                  \
    *reinterpret_cast<uint4*>(X3_BDST(tid + 512)) = b##S_##2;                                                  \
    *reinterpret_cast<uint4*>(X3_BDST(tid + 768)) = b##S_##3;                                                  \
    *reinterpret_cast<uint4*>(X3_BDST(s4)) = b##S_##4; /* threads >= 128 repeat their slot 0: no branch */     \
  } while (0)

  f32x4v acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // Pipeline: K step ks + 1 sits in one register set and ks + 2 is in flight into the other while step ks is multiplied;
  // the split and the LDS stores of ks + 1 are issued between the MFMAs of step ks, then ks + 3 is fetched into the set
  // just emptied.
  X3_FETCH(0, x);
  X3_STAGE(0, x, false);
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
// The split and the stage stores of K step ks + 1 are spread over all 54 MFMAs of step ks, two vector instructions per
// MFMA: 8 cycles of MFMA issue + 2 x 4 fill the 16 cycles an MFMA executes (MI355X_MICROARCH.md, issue costs), where four
// per MFMA behind the last 27 alone stretched those gaps to 24 (round 4: 47.4 -> 46.0 us on 6144 x 728 x 728).  The K tail
// is zeroed in the last K step only, which the steady loop never stages (12 v_cndmask per K step less).
#define X3_FBODY(KS_, S_, STEADY_)                                                                             \
    if ((STEADY_) || (KS_) + 1 < nk) {                                                                         \
      X3_STAGE((KS_) + 1, S_, STEADY_);                                                                        \
      X3_MFMA(af, bfr);                                                                                        \
      _Pragma("unroll") for (int g = 0; g < 54; ++g) {                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                     \
        if (g % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                     \
      }                                                                                                        \
      if ((STEADY_) || (KS_) + 3 < nk) X3_FETCH((KS_) + 3, S_);                                                \
    } else {                                                                                                   \
      X3_MFMA(af, bfr);                                                                                        \
    }
#define X3_ITER(KS_, S_, STEADY_)                                                                              \
  do {                                                                                                         \
    const unsigned short* base = smem + ((KS_) & 1) * X3_STEP;                                                 \
    bf16x8 af[3][3], bfr[3][3]; /* [tile][plane] */                                                            \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                            \
      af[t][p] = *reinterpret_cast<const bf16x8*>(base + p * X3_PLANE + (wm * 48 + t * 16 + p16) * X3_LDR + X3_SWZ(p16, kg));        \
      bfr[t][p] = *reinterpret_cast<const bf16x8*>(base + (3 + p) * X3_PLANE + (wn * 48 + t * 16 + p16) * X3_LDR + X3_SWZ(p16, kg)); \
    }                                                                                                          \
    X3_FBODY(KS_, S_, STEADY_);                                                                                \
    __syncthreads();                                                                                           \
  } while (0)
  // The steady state is a loop of its own with nothing conditional in it: the wait for a register set's loads is then
  // counted against the eight younger loads of the other set (s_waitcnt vmcnt(15) ... (8)).  With the fetch behind
  // "if (ks + 3 < nk)" the compiler has to assume the younger loads were never issued and waits for them as well
  // (vmcnt(7) ... (0)): a prefetch distance of one K step instead of two (-2 % on the 6144 x 728 x 728 launches).
  int ks = 0;
  if (nk > 4) {
    X3_FETCH(1, x);
    X3_FETCH(2, y);
    __syncthreads();
    for (; ks + 4 < nk; ks += 2) {
      X3_ITER(ks, x, true);
      X3_ITER(ks + 1, y, true);
    }
  } else {
    if (nk > 1) X3_FETCH(1, x);
    if (nk > 2) X3_FETCH(2, y);
    __syncthreads();
  }
  for (; ks < nk; ks += 2) {
    X3_ITER(ks, x, false);
    if (ks + 1 < nk) X3_ITER(ks + 1, y, false);
  }
  X3_STORE_C(C, ldc, m0, n0, M, N);
  if (colstats) X3_COLSTATS();
}

// ------------------------------------------------------------------------------------------------ entry points
extern "C" long spnet_bf16x3_kp(int K) { return (long)(K + 31) / 32 * 32; }
// bf16 elements of ONE plane of an [R][K] matrix (a plane set = 3 of them, 16-byte aligned, allocated zeroed)
extern "C" long spnet_bf16x3_plane_elems(long R, int K) { return x3t_plane_elems(R, K); }

extern "C" int spnet_split_bf16x3_batched(const void* jobs, int njobs, long max_elems, void* stream) {
  if (!jobs || njobs < 1 || max_elems < 1) return (int)hipErrorInvalidValue;
  long gx = (max_elems / 512 + 3) / 4;             // pieces of the largest job, four per workgroup
  if (gx > 2048) gx = 2048;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(split_bf16x3_tiled_kernel, dim3((unsigned)gx, njobs), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs));
  SPNET_RETURN_LAUNCH_STATUS();
}

static int x3_split_one(const float* W, void* planes, int K, long N, long sn, long sk, void* stream) {
  if (!W || !planes || K < 1 || N < 1 || (((uintptr_t)planes) & 15)) return (int)hipErrorInvalidValue;
  long gx = (x3t_plane_elems(N, K) / 512 + 3) / 4;
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(split_bf16x3_one_kernel, dim3((unsigned)gx), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<unsigned short*>(planes), K, N, sn, sk);
  SPNET_RETURN_LAUNCH_STATUS();
}
// planes of a Keras pointwise kernel W[K][N] in the forward operand form (rows = output channels)
extern "C" int spnet_split_bf16x3(const float* W, void* planes, int K, int N, void* stream) {
  return x3_split_one(W, planes, K, N, 1, N, stream);
}
// planes of an fp32 matrix A[R][K] (row stride lda): what a producer kernel with the planes output writes directly
extern "C" int spnet_split_rows_bf16x3(const float* A, long lda, void* planes, long R, int K, void* stream) {
  return x3_split_one(A, planes, K, R, lda, 1, stream);
}

static bool x3_planes_ok(const void* p, long R, int K) {
  return p && !(((uintptr_t)p) & 15) && 3 * x3t_plane_elems(R, K) * 2 < (1L << 31);     // 32-bit descriptor offsets
}

// C[M][N] = A[M][K] * B, A fp32 (lda % 4 == 0, K % 4 == 0, 16-byte aligned), B given as the planes of [N][K].
static int x3_launch(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K, float* colstats,
                     int* stat_rows, void* stream) {
  if (!A || !C || M < 1 || N < 1 || K < 4 || (lda & 3) || (K & 3)) return (int)hipErrorInvalidValue;
  if ((((uintptr_t)A) & 15) || !x3_planes_ok(planes, N, K)) return (int)hipErrorInvalidValue;
  const int tm = spnet_cdiv(M, X3_BM), tn = spnet_cdiv(N, X3_BN);
  if (stat_rows) *stat_rows = tm;
  hipLaunchKernelGGL(gemm_bf16x3_fwd_kernel, dim3(tm * tn), dim3(256), 0, (hipStream_t)stream, A, lda,
                     reinterpret_cast<const unsigned short*>(planes), C, ldc, M, N, K, tn, colstats);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_gemm_bf16x3_fwd(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K,
                                     void* stream) {
  return x3_launch(A, lda, planes, C, ldc, M, N, K, nullptr, nullptr, stream);
}

// The same with the BatchNorm column sums of C in the epilogue: colstats[*stat_rows][2][N], *stat_rows = ceil(M / 96)
// (the layout spnet_gemm_f32_colstats leaves; the consumers take the row count as an argument).
extern "C" int spnet_gemm_bf16x3_fwd_colstats(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N,
                                              int K, float* colstats, int* stat_rows, void* stream) {
  if (!colstats || !stat_rows) return (int)hipErrorInvalidValue;
  return x3_launch(A, lda, planes, C, ldc, M, N, K, colstats, stat_rows, stream);
}

// C[M][N] = A B^T with both operands as planes (A: [M][K], B: [N][K]); colstats / stat_rows NULL or as above.
extern "C" int spnet_gemm_bf16x3_pp(const void* a_planes, const void* b_planes, float* C, int ldc, int M, int N, int K,
                                    float* colstats, int* stat_rows, void* stream) {
  if (!C || M < 1 || N < 1 || K < 1 || ldc < N || (colstats && !stat_rows)) return (int)hipErrorInvalidValue;
  if (!x3_planes_ok(a_planes, M, K) || !x3_planes_ok(b_planes, N, K)) return (int)hipErrorInvalidValue;
  const int tm = spnet_cdiv(M, X3_BM), tn = spnet_cdiv(N, X3_BN);
  if (stat_rows) *stat_rows = tm;
  hipLaunchKernelGGL(gemm_bf16x3_pp_kernel, dim3(tm * tn), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned short*>(a_planes), x3t_plane_elems(M, K),
                     reinterpret_cast<const unsigned short*>(b_planes), x3t_plane_elems(N, K), C, ldc, M, N, (K + 31) / 32, tn,
                     colstats);
  SPNET_RETURN_LAUNCH_STATUS();
}

// Data-gradient GEMM of a pointwise convolution + the fused backward of the depthwise convolution in front of it
// (gemm_bf16x3_pp_dwbwd_kernel): dy_planes = planes of dL/d(pointwise output) [M][cout], w_planes = planes of W as stored
// ([cin][cout]: the data-gradient form), M = B*H*W pixels of an H x W plane with 192 % (H*W) == 0 and W <= 16 (a 192-row tile
// holds whole images); the remaining arguments and every output are spnet_dwconv3x3_tiled_bwd's (x_fwd = the depthwise
// input, w = its [3][3][cin] kernel, dx, partial rows for dw / the producer BatchNorm's sums) with
// spnet_gemm_bf16x3_dwbwd_rows(M) partial rows.  dz is never written.
extern "C" long spnet_gemm_bf16x3_dwbwd_rows(long M) { return (M + FB_BM - 1) / FB_BM; }
extern "C" long spnet_gemm_bf16x3_dwbwd_ok(int H, int W, int cin) {      // the planes of Xception's middle and exit flow
  return (((H == 12 && W == 16) || (H == 6 && W == 8)) && !(cin & 3)) ? 1 : 0;
}
extern "C" int spnet_gemm_bf16x3_pp_dwbwd(const void* dy_planes, const void* w_planes, int B, int H, int W, int cin, int cout,
                                          const float* x_fwd, const float* w, float* dx, int relu_in, const float* add,
                                          float* partial, const float* in_scale, const float* in_shift, const float* bn_mean,
                                          const float* bn_invstd, float* bn_partial, const float* bn_x, void* stream) {
  const long M = (long)B * H * W;
  if (!x_fwd || !w || !dx || !partial || B < 1 || !spnet_gemm_bf16x3_dwbwd_ok(H, W, cin)) return (int)hipErrorInvalidValue;
  if (bn_partial && (!bn_mean || !bn_invstd)) return (int)hipErrorInvalidValue;
  if (!x3_planes_ok(dy_planes, M, cout) || !x3_planes_ok(w_planes, cin, cout)) return (int)hipErrorInvalidValue;
  if ((((uintptr_t)x_fwd) | ((uintptr_t)dx) | ((uintptr_t)add) | ((uintptr_t)bn_x) | ((uintptr_t)w)) & 15) return (int)hipErrorInvalidValue;
  const int tm = spnet_cdiv(M, FB_BM), tn = spnet_cdiv(cin, X3_BN);
#define FB_LAUNCH(ADD_, BNX_)                                                                                  \
  hipLaunchKernelGGL((gemm_bf16x3_pp_dwbwd_kernel<0, ADD_, BNX_>), dim3(tm * tn), dim3(512), 0, (hipStream_t)stream,                \
                     reinterpret_cast<const unsigned short*>(dy_planes), x3t_plane_elems(M, cout),                              \
                     reinterpret_cast<const unsigned short*>(w_planes), x3t_plane_elems(cin, cout), (int)M, cin, (cout + 31) / 32, tn, \
                     H, W, x_fwd, w, dx, relu_in, add, partial, in_scale, in_shift, bn_mean, bn_invstd, bn_partial, bn_x,        \
                     (unsigned short*)nullptr, 0L)
  if (add && bn_x) FB_LAUNCH(true, true);
  else if (add) FB_LAUNCH(true, false);
  else if (bn_x) FB_LAUNCH(false, true);
  else FB_LAUNCH(false, false);
#undef FB_LAUNCH
  SPNET_RETURN_LAUNCH_STATUS();
}

// Inference: the forward GEMM of a pointwise convolution (z_planes = planes of its input [B*H*W][cin], w_planes = planes of
// W in the forward form) with the folded BatchNorm (scale_shift[2 * cout] or NULL) + ReLU (relu_next) of its output and the
// depthwise 3x3 of the NEXT separable convolution (w_next [3][3][cout]) in the epilogue; the result goes out as the planes
// of [B*H*W][cout] that the next pointwise GEMM reads.  Same planes, bit for bit, as spnet_gemm_bf16x3_pp followed by
// spnet_dwconv3x3_stream_fwd_x3; the pointwise output itself is never written.  Planes: spnet_gemm_bf16x3_dwbwd_ok.
extern "C" int spnet_gemm_bf16x3_pp_dwfwd(const void* z_planes, const void* w_planes, int B, int H, int W, int cin, int cout,
                                          const float* scale_shift, int relu_next, const float* w_next, void* out_planes,
                                          void* stream) {
  const long M = (long)B * H * W;
  if (!w_next || B < 1 || !spnet_gemm_bf16x3_dwbwd_ok(H, W, cout)) return (int)hipErrorInvalidValue;
  if (!x3_planes_ok(z_planes, M, cin) || !x3_planes_ok(w_planes, cout, cin) || !x3_planes_ok(out_planes, M, cout))
    return (int)hipErrorInvalidValue;
  if ((((uintptr_t)w_next) | ((uintptr_t)scale_shift)) & 15) return (int)hipErrorInvalidValue;
  const int tm = spnet_cdiv(M, FB_BM), tn = spnet_cdiv(cout, X3_BN);
  hipLaunchKernelGGL((gemm_bf16x3_pp_dwbwd_kernel<1, false, false>), dim3(tm * tn), dim3(512), 0, (hipStream_t)stream,
                     reinterpret_cast<const unsigned short*>(z_planes), x3t_plane_elems(M, cin),
                     reinterpret_cast<const unsigned short*>(w_planes), x3t_plane_elems(cout, cin), (int)M, cout, (cin + 31) / 32, tn,
                     H, W, (const float*)nullptr, w_next, (float*)nullptr, relu_next, (const float*)nullptr, (float*)nullptr,
                     scale_shift, scale_shift ? scale_shift + cout : nullptr, (const float*)nullptr, (const float*)nullptr,
                     (float*)nullptr, (const float*)nullptr, reinterpret_cast<unsigned short*>(out_planes), x3t_plane_elems(M, cout));
  SPNET_RETURN_LAUNCH_STATUS();
}

// K slices this library would cut a batch of weight gradients into (>= 1): enough workgroups for two per CU, at least 24
// contraction steps per slice.
extern "C" long spnet_gemm_bf16x3_wgrad_ksplit(int cin, int cout, int M, int nbatch) {
  const long tiles = (long)spnet_cdiv(cin, X3_BM) * spnet_cdiv(cout, X3_BN) * (nbatch < 1 ? 1 : nbatch);
  const int steps = (M + 31) / 32;
  long s = (512 + tiles - 1) / tiles;
  if (s > steps / 24) s = steps / 24;
  if (s < 1) s = 1;
  const int per = (int)((steps + s - 1) / s);
  return (steps + per - 1) / per;
}

// nbatch weight gradients of ONE shape in one launch: jobs = DEVICE array of {z planes ([M][cin]), dy planes ([M][cout]),
// dst} (three 64-bit words per problem).  ksplit == 1: dst = dW [cin][cout]; ksplit > 1: dst = ksplit slabs of cin * cout
// floats, slice order (spnet_reduce_slabs adds them in that order).
extern "C" int spnet_gemm_bf16x3_wgrad_batched(const void* jobs, int nbatch, int cin, int cout, int M, int ksplit,
                                               void* stream) {
  if (!jobs || nbatch < 1 || cin < 1 || cout < 1 || M < 1 || ksplit < 1) return (int)hipErrorInvalidValue;
  if (3 * x3t_plane_elems(M, cin) * 2 >= (1L << 31) || 3 * x3t_plane_elems(M, cout) * 2 >= (1L << 31))
    return (int)hipErrorInvalidValue;
  const int steps = (M + 31) / 32;
  if (ksplit > steps) return (int)hipErrorInvalidValue;
  const int per = (steps + ksplit - 1) / ksplit;
  if ((long)per * (ksplit - 1) >= steps) return (int)hipErrorInvalidValue;       // (an empty last slice)
  const int tm = spnet_cdiv(cin, X3_BM), tn = spnet_cdiv(cout, X3_BN);
  if ((long)tm * tn * nbatch * ksplit >= (1L << 31)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gemm_bf16x3_wgrad_kernel, dim3(tm * tn * nbatch * ksplit), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs), cin, cout, M, tn, tm * tn, nbatch, per);
  SPNET_RETURN_LAUNCH_STATUS();
}
