// fp32 GEMM on the bf16 matrix cores by operand splitting ("bf16x3"): the forward and data-gradient GEMMs of the pointwise
// (1x1) convolutions with >= 256 output columns (keras SeparableConv2D's pointwise step and the residual 1x1 convolutions of
// Xception, the pointwise layers of MobileNet; call site spnet/models.py:346-359) since round 4.
//
// Every fp32 operand is the exact sum of three bf16 numbers, x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// 3 x 8 significant bits = the 24 of fp32), and a product a*b is the sum of nine piece products.  Six of them --
//   ah*bh + (ah*bm + am*bh) + (ah*bl + am*bm + al*bh)
// -- carry everything down to 2^-24 of the product (the dropped am*bl, al*bm, al*bl are <= 2^-23.4 relative in sum), each
// piece product is exact in fp32 (8 x 8 bits) and v_mfma_f32_16x16x32_bf16 accumulates in fp32: six bf16 MFMAs (6 x 16
// cycles for a 16x16x32 block) in place of eight fp32 MFMAs (8 x 32 cycles) -- 2.67x fewer matrix-pipe cycles for a result
// whose error is of the size of ONE fp32 rounding per product: against float64 its error is no larger than that of the
// k-ordered fmaf chain of spnet_gemm_f32 (tests/test_kernels_gpu.py), but it is not that chain's bits.  Rounds 2-3 kept it
// as a probe; round 4 measured it in the whole train step (x1.085) and ran the whole GPU suite through it at the tolerances
// set from the exact kernels' own errors, and made it the product path for these two operand forms (Engine(pointwise=
// "f32") keeps the exact chain; weight gradients, blended data gradients, k x k convolutions and the Dense head are fp32
// MFMA kernels as before).  `dtype` of the bench line says so.
//
// Operand forms: A [M][K] fp32 (activations or gradients, split on the fly while the tile is staged), B given as three bf16
// planes in K-major order [3][N][Kp] (weights: split ONCE per optimizer step by spnet_split_bf16x3_batched, Kp = K rounded
// up to 32, zero padded), C [M][N] fp32; forward: B element (n, k) = W[k][n]; data gradient dX = dY W^T: (n, k) = W[n][k].
// 96x96 tile per workgroup (2 x 2 waves of 48x48 = 3x3 MFMA tiles), K step 32 = one MFMA depth,
// two LDS buffers with XOR-swizzled 16-byte chunks (conflict-free fragment reads without padding: 72 KB, two workgroups
// per CU), two register sets so that a K step is in flight for a whole iteration, its split and LDS stores issued in
// the shadow of the MFMAs (sched_group_barrier), one barrier per step; optional BatchNorm column sums in the epilogue.
//
// Measured (MI355X): 6144 x 728 x 728 in the train step 49.6 us against the exact kernel's 67.7 (x0.73), 24576 x 728 x 728 in
// the predict plan 174 against 225 us; shapes with 128 output columns lose (a 96-wide tile wastes a third of its second
// column tile: 168 against 147 us on 372000 x 128 x 128) and stay on the exact kernel.  Knock-out builds: MFMAs + fragment
// reads alone 32 us, fetch + split + stage alone 32 us, and the two do not overlap -- the 96x96 tile pulls 30 KB per K step
// through the vector memory path (A as fp32 + B as three planes = 10 B per operand element; 353 MB per launch, 6.6 TB/s
// from L2 / Infinity Cache), 1.7x the exact kernel's bytes, so the memory path and not the matrix pipe (16.5 us of MFMA at
// peak) sets the time.
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f2bf(float x) {           // round to nearest even; inputs are finite
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = f2bf(x);
  const float r1 = x - bf2f(h);          // exact
  m = f2bf(r1);
  const float r2 = r1 - bf2f(m);         // exact
  l = f2bf(r2);
}

// W [K][N] fp32 (Keras pointwise kernel [cin][cout]) -> planes[p][n][k] bf16, p = 0 (high) .. 2 (low), k < Kp
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float* __restrict__ W, unsigned short* __restrict__ planes,
                                                           int K, int N, int Kp) {
  const long total = (long)N * Kp;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % Kp), n = (int)(i / Kp);
    unsigned short h = 0, m = 0, l = 0;
    if (k < K) split3(W[(long)k * N + n], h, m, l);
    planes[i] = h;
    planes[total + i] = m;
    planes[2 * total + i] = l;
  }
}

// Two fp32 -> two bf16 (round to nearest even) in one v_cvt_pk_bf16_f32; the pieces of a pair come back as floats by a
// shift / a mask.  5.5 VALU operations per element for the three pieces (the scalar form above takes ~25).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  f32x2_t v = {a, b};
  bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return __builtin_bit_cast(unsigned, r);
}
// (x0, x1) -> packed (h, m, l) pairs
__device__ __forceinline__ void split3_pk(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = cvt_pk(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xFFFF0000u);
  l = cvt_pk(s0, s1);
}

#define X3_BM 96
#define X3_BN 96
#define X3_LDR 32            // LDS row stride in bf16: no padding (72 KB for two buffers, two workgroups per CU); the four 16-byte chunks of a
                             // row are XOR-swizzled with s = -(row / 4) mod 4.  A ds_read_b128 is served in four groups of 16
                             // lanes that are NOT consecutive -- {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS): a group
                             // holds rows 0-3, 12-15 at chunk kg and rows 4-11 at chunk kg ^ 1, and rows that are 4 apart
                             // share banks, so s(0-3), s(12-15), 1 ^ s(4-7), 1 ^ s(8-11) must differ: s = 0, 3, 2, 1 by row
                             // quad.  (s = row / 4, the first form, read 2-way conflicted exactly like no swizzle at all.)
#define X3_SWZ(ROW_, CHUNK_) ((((CHUNK_) ^ (0 - ((ROW_) >> 2))) & 3) * 8)
#define X3_PLANE (X3_BM * X3_LDR)

// B slot s (0 .. 1151) = (plane s / 384, row (s % 384) / 4, 16-byte chunk s % 4); a thread owns slots tid + 256 i.
// (five named registers and macros over them: an array of these ends up in scratch memory, lambdas or not)
#define X3_BSRC(S_) (Bp + ((S_) / 384) * plane_stride + (long)min(n0 + ((S_) % 384) / 4, N - 1) * Kp + k0_ + ((S_) % 4) * 8)
#define X3_BDST(S_) (base_ + (3 + (S_) / 384) * X3_PLANE + (((S_) % 384) / 4) * X3_LDR + X3_SWZ(((S_) % 384) / 4, (S_) % 4))

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_fwd_kernel(const float* __restrict__ A, int lda,
                                                                 const unsigned short* __restrict__ Bp, int Kp,
                                                                 float* __restrict__ C, int ldc, int M, int N, int K,
                                                                 int tiles_n, float* __restrict__ colstats) {
  // [buffer][A planes 0..2 | B planes 0..2][row][X3_LDR]
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 6 * X3_PLANE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * X3_BM, n0 = tn * X3_BN;
  const long plane_stride = (long)N * Kp;
  const int nk = Kp / 32;

  // global fetch slots.  A: 96 rows x 8 float4 = 768 slots, 3 per thread (row = s / 8, k quad = s % 8).
  // B: 3 planes x 96 rows x 4 sixteen-byte chunks = 1152 slots, 4.5 per thread (plane = s / 384, row = (s % 384) / 4).
  // two register sets (x, y): a K step stays in flight for a whole iteration before it is split and stored
  float4 ax0, ax1, ax2, ay0, ay1, ay2;
  uint4 bx0, bx1, bx2, bx3, bx4, by0, by1, by2, by3, by4;
  const int ar0 = min(m0 + tid / 8, M - 1), ar1 = min(m0 + (tid + 256) / 8, M - 1), ar2 = min(m0 + (tid + 512) / 8, M - 1);
  const int akq = (tid % 8) * 4;                   // (256 % 8 == 0: the same k quad for the three slots)
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const int s4 = tid < 128 ? tid + 1024 : tid;     // slot 4 exists for half the threads: the others re-read their slot 0
#define X3_FETCH(KS_, S_)                                                                                      \
  do {                                                                                                         \
    const int k0_ = (KS_) * 32;                                                                                \
    const int kk_ = min(k0_ + akq, K - 4); /* always a load from global memory, zeroed in the stage */         \
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
    split3_pk(v_.x, v_.y, h0_, m0_, l0_);                                                                      \
    split3_pk(v_.z, v_.w, h1_, m1_, l1_);                                                                      \
    const int o_ = ((S_) / 8) * X3_LDR + X3_SWZ((S_) / 8, akq / 8) + (akq & 4);                                \
    *reinterpret_cast<uint2*>(base_ + 0 * X3_PLANE + o_) = make_uint2(h0_, h1_);                               \
    *reinterpret_cast<uint2*>(base_ + 1 * X3_PLANE + o_) = make_uint2(m0_, m1_);                               \
    *reinterpret_cast<uint2*>(base_ + 2 * X3_PLANE + o_) = make_uint2(l0_, l1_);                               \
  } while (0)
#define X3_STAGE(KS_, S_, ALLK_)                                                                                     \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((KS_) & 1) * 6 * X3_PLANE;                                                 \
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
  // smallest terms first: (l,h) (m,m) (h,l), then (m,h) (h,m), then (h,h); the nine tiles of a term back to back, so
  // that consecutive MFMAs never wait for each other's accumulator
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
#define X3_TERMS(T0_, T1_)                                                                                     \
  _Pragma("unroll") for (int t = (T0_); t < (T1_); ++t)                                                        \
  _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                \
  _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][TA[t]], bfr[j][TB[t]], acc[i][j], 0, 0, 0)
// The split and the stage stores of K step ks + 1 are spread over all 54 MFMAs of step ks, two vector instructions per
// MFMA: 8 cycles of MFMA issue + 2 x 4 fill the 16 cycles an MFMA executes (MI355X_MICROARCH.md, issue costs), where four
// per MFMA behind the last 27 alone stretched those gaps to 24 (round 4: 47.4 -> 46.0 us on 6144 x 728 x 728).  The K tail
// is zeroed in the last K step only, which the steady loop never stages (12 v_cndmask per K step less).
#define X3_BODY(KS_, S_, STEADY_)                                                                              \
    if ((STEADY_) || (KS_) + 1 < nk) {                                                                         \
      X3_STAGE((KS_) + 1, S_, STEADY_);                                                        \
      X3_TERMS(0, 6);                                                                          \
      _Pragma("unroll") for (int g = 0; g < 54; ++g) {                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                                     \
        if (g % 3 == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                     \
      }                                                                                                        \
      if ((STEADY_) || (KS_) + 3 < nk) X3_FETCH((KS_) + 3, S_);                                \
    } else {                                                                                                   \
      X3_TERMS(0, 6);                                                                          \
    }
#define X3_ITER(KS_, S_, STEADY_)                                                                              \
  do {                                                                                                         \
    const unsigned short* base = smem + ((KS_) & 1) * 6 * X3_PLANE;                                            \
    bf16x8 af[3][3], bfr[3][3]; /* [tile][plane] */                                                            \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                            \
      af[t][p] = *reinterpret_cast<const bf16x8*>(base + p * X3_PLANE + (wm * 48 + t * 16 + p16) * X3_LDR + X3_SWZ(p16, kg));        \
      bfr[t][p] = *reinterpret_cast<const bf16x8*>(base + (3 + p) * X3_PLANE + (wn * 48 + t * 16 + p16) * X3_LDR + X3_SWZ(p16, kg)); \
    }                                                                                                          \
    X3_BODY(KS_, S_, STEADY_);                                                                 \
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

  // C/D map of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + wm * 48 + i * 16 + kg * 4 + r;
      if (row < M) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int col = n0 + wn * 48 + j * 16 + p16;
          if (col < N) C[(long)row * ldc + col] = acc[i][j][r];
        }
      }
    }
  // BatchNorm column sums of this 96-row tile (sum, sum of squares per output column), as spnet_gemm_f32_colstats leaves
  // them: colstats[tile row][2][N].  Per lane over its 12 rows, then over the four row groups of the wave (lanes 16
  // apart), then over the two waves that share the columns (through LDS: the stage buffers are idle now); fixed order.
  if (colstats) {
    float* sred = reinterpret_cast<float*>(smem);      // [2 sums][2 wm][96 columns]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * 48 + i * 16 + kg * 4 + r;
          const float v = row < M ? acc[i][j][r] : 0.f;
          sv += v;
          qv = fmaf(v, v, qv);
        }
      sv += __shfl_xor(sv, 16, 64); qv += __shfl_xor(qv, 16, 64);
      sv += __shfl_xor(sv, 32, 64); qv += __shfl_xor(qv, 32, 64);
      if (lane < 16) {
        const int cl = wn * 48 + j * 16 + p16;
        sred[(0 * 2 + wm) * X3_BN + cl] = sv;
        sred[(1 * 2 + wm) * X3_BN + cl] = qv;
      }
    }
    __syncthreads();
    if (tid < 2 * X3_BN) {
      const int q = tid / X3_BN, cl = tid % X3_BN, col = n0 + cl;
      if (col < N) colstats[((long)tm * 2 + q) * N + col] = sred[(q * 2 + 0) * X3_BN + cl] + sred[(q * 2 + 1) * X3_BN + cl];
    }
  }
}

// All weight splits of a step in one launch: job j = {W, planes, K, N, sn, sk} (six 64-bit words, device memory): the
// B operand element (n, k) is W[n * sn + k * sk] -- forward form of a Keras pointwise kernel [cin][cout]: K = cin, N = cout,
// sn = 1, sk = cout; data-gradient form (dX = dY W^T): K = cout, N = cin, sn = cout, sk = 1.
__global__ __launch_bounds__(256) void split_bf16x3_batched_kernel(const long long* __restrict__ jobs) {
  __shared__ float tile[32][33];
  const long long* jb = jobs + 6 * blockIdx.y;
  const float* __restrict__ W = reinterpret_cast<const float*>(jb[0]);
  unsigned short* __restrict__ planes = reinterpret_cast<unsigned short*>(jb[1]);
  const int K = (int)jb[2], N = (int)jb[3];
  const long sn = jb[4], sk = jb[5];
  const int Kp = (K + 31) / 32 * 32;
  const long total = (long)N * Kp;
  if (sk == 1) {            // k is the contiguous axis of the source as well: straight through, coalesced both ways
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
      const int k = (int)(i % Kp), n = (int)(i / Kp);
      unsigned short h = 0, m = 0, l = 0;
      if (k < K) split3(W[n * sn + k], h, m, l);
      planes[i] = h;
      planes[total + i] = m;
      planes[2 * total + i] = l;
    }
    return;
  }
  // n is the contiguous axis of the source (the forward form of a [cin][cout] kernel): 32 x 32 tiles through LDS, read
  // along n, written along k
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int tk = Kp / 32, tn = (N + 31) / 32;
  for (int t = blockIdx.x; t < tk * tn; t += gridDim.x) {
    const int k0 = (t % tk) * 32, n0 = (t / tk) * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + ty + 8 * j, n = n0 + tx;
      tile[ty + 8 * j][tx] = (k < K && n < N) ? W[n * sn + k * sk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + ty + 8 * j, k = k0 + tx;
      if (n < N) {
        unsigned short h, m, l;
        split3(tile[tx][ty + 8 * j], h, m, l);
        const long i = (long)n * Kp + k;
        planes[i] = h;
        planes[total + i] = m;
        planes[2 * total + i] = l;
      }
    }
    __syncthreads();
  }
}

extern "C" int spnet_split_bf16x3_batched(const void* jobs, int njobs, long max_elems, void* stream) {
  if (!jobs || njobs < 1 || max_elems < 1) return (int)hipErrorInvalidValue;
  long gx = (max_elems + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(split_bf16x3_batched_kernel, dim3((unsigned)gx, njobs), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const long long*>(jobs));
  SPNET_RETURN_LAUNCH_STATUS();
}

// planes: 3 * N * Kp bf16 (Kp = K rounded up to 32)
extern "C" long spnet_bf16x3_kp(int K) { return (long)(K + 31) / 32 * 32; }

extern "C" int spnet_split_bf16x3(const float* W, void* planes, int K, int N, void* stream) {
  if (!W || !planes || K < 1 || N < 1 || (((uintptr_t)planes) & 15)) return (int)hipErrorInvalidValue;
  const int Kp = (int)spnet_bf16x3_kp(K);
  hipLaunchKernelGGL(split_bf16x3_kernel, dim3(spnet_ew_grid((long)N * Kp, 256)), dim3(256), 0, (hipStream_t)stream, W,
                     reinterpret_cast<unsigned short*>(planes), K, N, Kp);
  SPNET_RETURN_LAUNCH_STATUS();
}

// C[M][N] = A[M][K] * W[K][N], W given as the planes of spnet_split_bf16x3.  lda % 4 == 0, A 16-byte aligned.
static int x3_launch(const float* A, int lda, const void* planes, float* C, int ldc, int M, int N, int K, float* colstats,
                     int* stat_rows, void* stream) {
  if (!A || !planes || !C || M < 1 || N < 1 || K < 1 || (lda & 3) || (K & 3)) return (int)hipErrorInvalidValue;
  if ((((uintptr_t)A) | ((uintptr_t)planes)) & 15) return (int)hipErrorInvalidValue;
  const int Kp = (int)spnet_bf16x3_kp(K);
  const int tm = spnet_cdiv(M, X3_BM), tn = spnet_cdiv(N, X3_BN);
  if (stat_rows) *stat_rows = tm;
  hipLaunchKernelGGL(gemm_bf16x3_fwd_kernel, dim3(tm * tn), dim3(256), 0, (hipStream_t)stream, A, lda,
                     reinterpret_cast<const unsigned short*>(planes), Kp, C, ldc, M, N, K, tn, colstats);
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
