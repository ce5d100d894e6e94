// fp32 MFMA GEMM for the pointwise (1x1) convolutions, the stride-2 residual convs, block1_conv2
// (through im2col) and the Dense head:  C[M,N] = sum_k A(m,k) * B(k,n)
//
// Replaces the TensorFlow/cuDNN kernels behind keras SeparableConv2D's pointwise step, Conv2D(1x1)
// and Dense (call sites spnet/models.py:357-359, 388).  Arithmetic is exact f32: the
// v_mfma_f32_16x16x4_f32 instruction is a k-ordered fmaf chain (one rounding per product), so the
// result equals a plain f32 dot product evaluated in k order.
//
// Operand layouts ("major" = which index is contiguous in memory):
//   A  K_MAJOR   : A(m,k) = A[m*lda + k]      (activations [pixels][Cin]: forward, dgrad)
//      OUT_MAJOR : A(m,k) = A[k*lda + m]      (wgrad: A = X^T, read without a transpose pass)
//   B  OUT_MAJOR : B(k,n) = B[k*ldb + n]      (weights [Cin][Cout]: forward; dY in wgrad)
//      K_MAJOR   : B(k,n) = B[n*ldb + k]      (dgrad: B = W^T read in place)
// Each operand's LDS image keeps its own major (see TileStage): operand fetches are conflict-free
// ds_read_b32 and no staging pass transposes.
//
// Split-K: grid.z slices write fp32 slabs to a workspace, a second kernel sums them in slice order
// (deterministic; no float atomics).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SP_K_MAJOR = 0, SP_OUT_MAJOR = 1 };

// One operand tile (BR output rows/cols x BK reduction steps) staged global -> registers -> LDS.
// The LDS image keeps the operand's own major, so no staging pass transposes anything:
//   K_MAJOR   image [r][k], leading dim BK+2 : the 16 rows x 2 k of a 32-lane fetch group land on
//             banks (2r + k) % 32, all distinct; rows are 8-byte aligned -> two ds_write_b64 per float4
//   OUT_MAJOR image [k][r], leading dim BR+16: 16 consecutive r per k, the next k 16 banks further
template <int BR, int BK, int MAJ>
struct TileStage {
  static constexpr int TOTAL = BR * BK / 4;            // float4 per tile
  static constexpr int NV = (TOTAL + 255) / 256;       // float4 per thread
  static constexpr int LD = (MAJ == SP_K_MAJOR) ? (BK + 2) : (BR + 16);
  static constexpr int SIZE = (MAJ == SP_K_MAJOR) ? BR * LD : BK * LD;
  float4 v[NV];

  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int r0, int R, int k0,
                                       int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((TOTAL % 256 == 0) || f < TOTAL) {
        if (MAJ == SP_OUT_MAJOR) {
          const int k = f / (BR / 4), r4 = f % (BR / 4);
          const int gk = k0 + k, gr = r0 + r4 * 4;
          if (gk < kend && gr < R) val = *reinterpret_cast<const float4*>(P + (long)gk * ld + gr);
        } else {
          const int r = f / (BK / 4), kq = f % (BK / 4);
          const int gr = r0 + r, gk = k0 + kq * 4;
          if (gr < R && gk < kend) val = *reinterpret_cast<const float4*>(P + (long)gr * ld + gk);
        }
      }
      v[i] = val;
    }
  }

  __device__ __forceinline__ void store(float* __restrict__ S, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      if ((TOTAL % 256 == 0) || f < TOTAL) {
        if (MAJ == SP_OUT_MAJOR) {
          const int k = f / (BR / 4), r4 = f % (BR / 4);
          *reinterpret_cast<float4*>(S + k * LD + r4 * 4) = v[i];
        } else {
          const int r = f / (BK / 4), kq = f % (BK / 4);
          float2* s = reinterpret_cast<float2*>(S + r * LD + kq * 4);
          s[0] = make_float2(v[i].x, v[i].y);
          s[1] = make_float2(v[i].z, v[i].w);
        }
      }
    }
  }

  // element (row r of the tile, reduction index k of the tile)
  static __device__ __forceinline__ float fetch(const float* __restrict__ S, int r, int k) {
#if defined(SP_ABLATE) && SP_ABLATE == 1
    return (float)(r + k) * 1e-3f;     // ablation build: no LDS operand reads
#else
    return (MAJ == SP_K_MAJOR) ? S[r * LD + k] : S[k * LD + r];
#endif
  }
};

template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], float* __restrict__ smem,
                                              float* __restrict__ C, int ldc, int M, int N, int m0, int n0,
                                              int tm, int z, long slab_stride, const float* __restrict__ bias,
                                              float* __restrict__ colstats, int tid, int lane, int wm, int wn) {
  // Optional BatchNorm statistics of the output tile: per column sum and sum of squares over this
  // workgroup's BM rows -> colstats[tm][2][N] (rows past M hold zeros and contribute nothing).
  // Fixed reduction order: registers (i, r) -> lanes (xor 16, 32) -> waves (wm order) through LDS.
  if (colstats) {
    float* sred = smem;                       // [2][WM][BN], the staging buffers are idle now
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + wm * (TM * 16) + i * 16 + 4 * (lane >> 4) + r;
          const float v = row < M ? acc[i][j][r] : 0.f;
          sv += v;
          qv = fmaf(v, v, qv);
        }
      sv += __shfl_xor(sv, 16, 64);
      qv += __shfl_xor(qv, 16, 64);
      sv += __shfl_xor(sv, 32, 64);
      qv += __shfl_xor(qv, 32, 64);
      if (lane < 16) {
        const int cl = wn * (TN * 16) + j * 16 + lane;
        sred[(0 * WM + wm) * BN + cl] = sv;
        sred[(1 * WM + wm) * BN + cl] = qv;
      }
    }
    __syncthreads();
    for (int c = tid; c < 2 * BN; c += 256) {
      const int q = c / BN, cl = c % BN;
      const int col = n0 + cl;
      if (col < N) {
        float t = sred[(q * WM) * BN + cl];
#pragma unroll
        for (int w = 1; w < WM; ++w) t += sred[(q * WM + w) * BN + cl];
        colstats[((long)tm * 2 + q) * N + col] = t;
      }
    }
  }

  float* Cz = C + (long)z * slab_stride;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (TN * 16) + j * 16 + (lane & 15);
      const int rbase = m0 + wm * (TM * 16) + i * 16 + 4 * (lane >> 4);
      if (col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = rbase + r;
          if (row < M) Cz[(long)row * ldc + col] = acc[i][j][r] + bv;
        }
      }
    }
  }
}

// 4 waves (WM x WN), each wave owns a (BM/WM) x (BN/WN) block built from 16x16 MFMA tiles
// (v_mfma_f32_16x16x4_f32: A[l&15][k=l>>4], B[k=l>>4][l&15], D col=l&15,row=4*(l>>4)+reg).
template <int BM, int BN, int BK, int WM, int WN, int AMAJ, int BMAJ>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda,
                                                       const float* __restrict__ B, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N,
                                                       int K, int k_chunk, long slab_stride,
                                                       int tiles_m, int tiles_n, int nsplit,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ colstats) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(TM >= 1 && TN >= 1 && (BM % (WM * 16)) == 0 && (BN % (WN * 16)) == 0, "wave tile");
  typedef TileStage<BM, BK, AMAJ> SA;
  typedef TileStage<BN, BK, BMAJ> SB;
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nblk = tiles_m * tiles_n * nsplit;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int tn = lid % tiles_n;
  lid /= tiles_n;
  const int tm = lid % tiles_m;
  const int z = lid / tiles_m;

  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = z * k_chunk;
  const int kend = min(K, kbeg + k_chunk);
  const int nt = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  SA sa;
  SB sb;

  if (nt > 0) {
    sa.load(A, lda, m0, M, kbeg, kend, tid);
    sb.load(B, ldb, n0, N, kbeg, kend, tid);
    sa.store(smem, tid);
    sb.store(smem + SA::SIZE, tid);
  }
  __syncthreads();

  const int arow = wm * (TM * 16) + (lane & 15);
  const int bcol = wn * (TN * 16) + (lane & 15);
  const int kq = lane >> 4;

  for (int t = 0; t < nt; ++t) {
    const float* cur = smem + (t & 1) * STAGE;
    float* nxt = smem + ((t + 1) & 1) * STAGE;
    const bool more = (t + 1 < nt);
#if !(defined(SP_ABLATE) && SP_ABLATE == 2)
    if (more) {
      sa.load(A, lda, m0, M, kbeg + (t + 1) * BK, kend, tid);
      sb.load(B, ldb, n0, N, kbeg + (t + 1) * BK, kend, tid);
    }
#endif
    const float* as = cur;
    const float* bs = cur + SA::SIZE;
    // The next tile's registers are written to the idle LDS buffer in the MIDDLE of the MFMA stream:
    // nobody reads that buffer during this iteration, so its ds_write latency (and the vmcnt wait in
    // front of it) hides under the second half of the MFMAs instead of sitting in front of the barrier.
#ifndef SP_STORE_AT
#define SP_STORE_AT (BK / 2)
#endif
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
#if !(defined(SP_ABLATE) && SP_ABLATE == 3)
      if (kk == SP_STORE_AT && more) {
        sa.store(nxt, tid);
        sb.store(nxt + SA::SIZE, tid);
      }
#endif
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = SA::fetch(as, arow + i * 16, kk + kq);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = SB::fetch(bs, bcol + j * 16, kk + kq);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (SP_STORE_AT >= BK && more) {
      sa.store(nxt, tid);
      sb.store(nxt + SA::SIZE, tid);
    }
#if !(defined(SP_ABLATE) && SP_ABLATE == 3)
    __syncthreads();
#endif
  }

  gemm_epilogue<BM, BN, WM, WN, TM, TN>(acc, smem, C, ldc, M, N, m0, n0, tm, z, slab_stride, bias, colstats, tid, lane,
                                        wm, wn);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (global_load_lds_dwordx4): tiles go HBM/L2 -> LDS without passing through VGPRs, so
// the K loop carries no staging registers, no ds_write and no VALU for them; the loads of tile t+1 are in
// flight underneath the MFMAs of tile t and are retired by the vmcnt(0) of the closing barrier.
// A wave-instruction writes 64 x 16 B = 1 KiB of CONSECUTIVE LDS, so the image is addressed as a linear
// sequence of 16-byte chunks: chunk c -> (row c / CPR, column 4*(c % CPR)); lanes that fall into a row's
// padding fetch a dummy element.  No bounds handling exists in this path, therefore:
//   * K % BK == 0 (BK = 32, or 28 for the 728-channel layers) -- checked by the host,
//   * rows / columns past M / N are CLAMPED to the last valid one; their results are masked by the
//     epilogue (and by the column statistics).
// K-major rows are 36 floats apart (16-B aligned; 2-way bank conflict on the 16x2 operand fetch).
// ------------------------------------------------------------------------------------------------
template <int BR, int BK, int MAJ>
struct TileDma {
  static constexpr int WIDTH = (MAJ == SP_K_MAJOR) ? BK : BR;     // valid floats per LDS row
  static constexpr int ROWS = (MAJ == SP_K_MAJOR) ? BR : BK;
  static constexpr int LD = (MAJ == SP_K_MAJOR) ? 36 : (BR + 16);
  static constexpr int CPR = LD / 4;                               // 16-byte chunks per LDS row
  static constexpr int NCH = ROWS * CPR;
  static constexpr int NINSTR = (NCH + 63) / 64;                   // wave-instructions per tile
  static constexpr int SIZE = NINSTR * 256;                        // floats (whole wave-instructions)
  static_assert(BK <= 32, "K-major rows hold at most 32 reduction steps");

  static __device__ __forceinline__ void issue(float* __restrict__ S, const float* __restrict__ P, int ld,
                                               int r0, int R, int k0, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < (NINSTR + 3) / 4; ++i) {
      const int ins = wave + i * 4;                                // wave-uniform
      if (ins < NINSTR) {
        const int c = ins * 64 + lane;
        const int row = c / CPR, col = (c % CPR) * 4;
        const float* g = P;                                        // padding lanes: any valid address
        if (col < WIDTH && row < ROWS) {
          if (MAJ == SP_K_MAJOR) g = P + (long)min(r0 + row, R - 1) * ld + (k0 + col);
          else g = P + (long)(k0 + row) * ld + min(r0 + col, R - 4);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(S + ins * 256), 16, 0, 0);
      }
    }
  }
  static __device__ __forceinline__ float fetch(const float* __restrict__ S, int r, int k) {
    return (MAJ == SP_K_MAJOR) ? S[r * LD + k] : S[k * LD + r];
  }
};

template <int BM, int BN, int BK, int WM, int WN, int AMAJ, int BMAJ>
__global__ __launch_bounds__(256) void gemm_f32_dma_kernel(const float* __restrict__ A, int lda,
                                                           const float* __restrict__ B, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int N,
                                                           int K, int k_chunk, long slab_stride,
                                                           int tiles_m, int tiles_n, int nsplit,
                                                           const float* __restrict__ bias,
                                                           float* __restrict__ colstats) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  typedef TileDma<BM, BK, AMAJ> SA;
  typedef TileDma<BN, BK, BMAJ> SB;
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nblk = tiles_m * tiles_n * nsplit;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int tn = lid % tiles_n;
  lid /= tiles_n;
  const int tm = lid % tiles_m;
  const int z = lid / tiles_m;

  const int m0 = tm * BM, n0 = tn * BN;
  const int kbeg = z * k_chunk;
  const int kend = min(K, kbeg + k_chunk);
  const int nt = (kend - kbeg) / BK;               // exact: K and k_chunk are multiples of BK

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  if (nt > 0) {
    SA::issue(smem, A, lda, m0, M, kbeg, wave, lane);
    SB::issue(smem + SA::SIZE, B, ldb, n0, N, kbeg, wave, lane);
  }
  __syncthreads();                                  // vmcnt(0) + barrier: tile 0 has landed

  const int arow = wm * (TM * 16) + (lane & 15);
  const int bcol = wn * (TN * 16) + (lane & 15);
  const int kq = lane >> 4;

  for (int t = 0; t < nt; ++t) {
    const float* cur = smem + (t & 1) * STAGE;
    float* nxt = smem + ((t + 1) & 1) * STAGE;
    if (t + 1 < nt) {                               // the idle buffer was last read before the previous barrier
      SA::issue(nxt, A, lda, m0, M, kbeg + (t + 1) * BK, wave, lane);
      SB::issue(nxt + SA::SIZE, B, ldb, n0, N, kbeg + (t + 1) * BK, wave, lane);
    }
    const float* as = cur;
    const float* bs = cur + SA::SIZE;
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = SA::fetch(as, arow + i * 16, kk + kq);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = SB::fetch(bs, bcol + j * 16, kk + kq);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                // retires this wave's DMA (vmcnt(0)) and publishes it
  }
  gemm_epilogue<BM, BN, WM, WN, TM, TN>(acc, smem, C, ldc, M, N, m0, n0, tm, z, slab_stride, bias, colstats, tid, lane,
                                        wm, wn);
}

// out[row*ldc + col] = sum_z ws[z*M*N + row*N + col] (+ bias[col]); N % 4 == 0, ldc % 4 == 0.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, int nslab,
                                                           int M, int N, float* __restrict__ out,
                                                           int ldc, const float* __restrict__ bias) {
  const long total4 = (long)M * N / 4;
  const long mn = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const int row = (int)(e / N), col = (int)(e % N);
    float4 s = *reinterpret_cast<const float4*>(ws + e);
    for (int zz = 1; zz < nslab; ++zz) {
      const float4 t = *reinterpret_cast<const float4*>(ws + (long)zz * mn + e);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + col);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    *reinterpret_cast<float4*>(out + (long)row * ldc + col) = s;
  }
}

#ifndef SP_BK
#define SP_BK 32
#endif

// bk_dma: 32 / 28 -> LDS-DMA kernel with that K tile; 0 -> register-staged kernel (any K % 4 == 0)
template <int BM, int BN, int WM, int WN>
static int launch_tile(const float* A, int amaj, int lda, const float* B, int bmaj, int ldb, float* C,
                       int ldc, int M, int N, int K, int nsplit, int k_chunk, long slab_stride,
                       const float* bias, float* colstats, int bk_dma, hipStream_t st) {
  const int tm = spnet_cdiv(M, BM), tn = spnet_cdiv(N, BN);
  dim3 grid(tm * tn * nsplit), block(256);
#define SP_ARGS A, lda, B, ldb, C, ldc, M, N, K, k_chunk, slab_stride, tm, tn, nsplit, bias, colstats
#define SP_LAUNCH(KERNEL, BKV, AM, BMJ) \
  hipLaunchKernelGGL((KERNEL<BM, BN, BKV, WM, WN, AM, BMJ>), grid, block, 0, st, SP_ARGS)
#define SP_FORMS(KERNEL, BKV)                                                                             \
  if (amaj == SP_K_MAJOR && bmaj == SP_OUT_MAJOR) SP_LAUNCH(KERNEL, BKV, SP_K_MAJOR, SP_OUT_MAJOR);       \
  else if (amaj == SP_K_MAJOR && bmaj == SP_K_MAJOR) SP_LAUNCH(KERNEL, BKV, SP_K_MAJOR, SP_K_MAJOR);      \
  else if (amaj == SP_OUT_MAJOR && bmaj == SP_OUT_MAJOR) SP_LAUNCH(KERNEL, BKV, SP_OUT_MAJOR, SP_OUT_MAJOR); \
  else return (int)hipErrorInvalidValue
  if (bk_dma == 32) { SP_FORMS(gemm_f32_dma_kernel, 32); }
  else if (bk_dma == 28) { SP_FORMS(gemm_f32_dma_kernel, 28); }
  else { SP_FORMS(gemm_f32_kernel, SP_BK); }
#undef SP_FORMS
#undef SP_LAUNCH
#undef SP_ARGS
  return 0;
}

// Tile ids: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 32x128, 5 = 96x96
#define SP_NTILES 5
static void tile_dims(int tile, int* bm, int* bn) {
  switch (tile) {
    case 1: *bm = 128; *bn = 128; break;
    case 2: *bm = 128; *bn = 64; break;
    case 3: *bm = 64; *bn = 64; break;
    case 5: *bm = 96; *bn = 96; break;
    default: *bm = 32; *bn = 128; break;
  }
}

// Automatic K split of one tile shape: enough workgroups for ~2 per CU, slices at least 4 K-tiles deep.
static int auto_split(long tiles, int M, int N, int K, bool have_ws, long ws_floats) {
  if (tiles >= 512 || !have_ws) return 1;
  long want = (512 + tiles - 1) / tiles;
  long maxk = K / (SP_BK * 4);
  if (maxk < 1) maxk = 1;
  if (want > maxk) want = maxk;
  const long fit = ws_floats / ((long)M * N);
  if (want > fit) want = fit;
  return want > 1 ? (int)want : 1;
}

// cost ~ rounds over the 256 CUs x work per workgroup / tile efficiency.  The 96x96 tile exists for the
// network's dominant shape, M = batch*12*16 = 6144 rows x 728 channels: 64 x 8 = 512 tiles = exactly two
// per CU, where 128x128 leaves 288 tiles (1.125 rounds) and 64x64 pays twice the LDS traffic per FLOP.
static int pick_tile(int M, int N, int K, int split_k, bool have_ws, long ws_floats) {
  if (M <= 32) return 4;
  const int cand[4] = {1, 5, 2, 3};
  const double eff[4] = {1.00, 0.97, 0.90, 0.85};
  int best = 1;
  double best_cost = 1e300;
  for (int c = 0; c < 4; ++c) {
    int bm, bn;
    tile_dims(cand[c], &bm, &bn);
    const long tiles = (long)spnet_cdiv(M, bm) * spnet_cdiv(N, bn);
    const int ns = split_k > 0 ? split_k : auto_split(tiles, M, N, K, have_ws, ws_floats);
    const double blocks = (double)tiles * ns;
    double rounds = (double)(long)((blocks + 255.0) / 256.0);
    const double cost = rounds * bm * bn * ((double)K / ns + 64.0) / eff[c];
    if (cost < best_cost) { best_cost = cost; best = cand[c]; }
  }
  return best;
}

static int gemm_impl(const float* A, int a_major, int lda, const float* B, int b_major, int ldb, float* C,
                     int ldc, int M, int N, int K, int split_k, float* workspace, long ws_floats,
                     const float* bias, int tile, float* colstats, int* stat_rows, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (colstats) split_k = 1;   // statistics are taken from complete dot products
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  if ((lda & 3) || (ldb & 3) || (N & 3) || (ldc & 3)) return (int)hipErrorInvalidValue;
  if ((a_major == SP_K_MAJOR || b_major == SP_K_MAJOR) && (K & 3)) return (int)hipErrorInvalidValue;
  if (a_major == SP_OUT_MAJOR && (M & 3)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return (int)hipErrorInvalidValue;
  if (tile <= 0 || tile > SP_NTILES) tile = pick_tile(M, N, K, split_k, workspace != nullptr, ws_floats);
  int bm, bn;
  tile_dims(tile, &bm, &bn);
  const long tiles = (long)spnet_cdiv(M, bm) * spnet_cdiv(N, bn);
  const int BK = SP_BK;
  int nsplit = split_k;
  if (nsplit <= 0) nsplit = auto_split(tiles, M, N, K, workspace != nullptr, ws_floats);
  // LDS-DMA path: needs K to be a whole number of K tiles (no bounds handling) and >= 4 rows/columns to
  // clamp to; otherwise the register-staged kernel (any K % 4 == 0) is used.
  int bk_dma = 0;
#ifndef SP_NO_DMA
  if (M >= 4 && N >= 4 && tile != 4) {
    if (K % 32 == 0) bk_dma = 32;
    else if (K % 28 == 0) bk_dma = 28;
  }
#endif
  const int bkt = bk_dma ? bk_dma : BK;
  int k_chunk = ((K + nsplit - 1) / nsplit + bkt - 1) / bkt * bkt;
  nsplit = (K + k_chunk - 1) / k_chunk;
  float* out = C;
  int out_ld = ldc;
  long slab = 0;
  const float* kbias = bias;
  if (nsplit > 1) {
    if (!workspace || (long)nsplit * M * N > ws_floats) return (int)hipErrorInvalidValue;
    if (((uintptr_t)workspace) & 15) return (int)hipErrorInvalidValue;
    out = workspace;
    out_ld = N;
    slab = (long)M * N;
    kbias = nullptr;
  }
  if (stat_rows) *stat_rows = spnet_cdiv(M, bm);
  int rc;
  switch (tile) {
    case 1: rc = launch_tile<128, 128, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, bk_dma, st); break;
    case 2: rc = launch_tile<128, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, bk_dma, st); break;
    case 3: rc = launch_tile<64, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, bk_dma, st); break;
    case 5: rc = launch_tile<96, 96, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, bk_dma, st); break;
    default: rc = launch_tile<32, 128, 1, 4>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, bk_dma, st); break;
  }
  if (rc) return rc;
  if (nsplit > 1) {
    const long total4 = (long)M * N / 4;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(spnet_ew_grid(total4, 256)), dim3(256), 0, st,
                       workspace, nsplit, M, N, C, ldc, bias);
  }
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_gemm_f32(const float* A, int a_major, int lda, const float* B, int b_major,
                              int ldb, float* C, int ldc, int M, int N, int K, int split_k,
                              float* workspace, long ws_floats, const float* bias, int tile,
                              void* stream) {
  return gemm_impl(A, a_major, lda, B, b_major, ldb, C, ldc, M, N, K, split_k, workspace, ws_floats, bias,
                   tile, nullptr, nullptr, stream);
}

// Forward-form GEMM that also emits BatchNorm column statistics of C: colstats[rows][2][N] holds per
// row-tile partial (sum, sum of squares); *stat_rows (host) receives `rows`.  colstats must hold
// ceil(M/32)*2*N floats (an upper bound for every tile shape).
extern "C" int spnet_gemm_f32_colstats(const float* A, int a_major, int lda, const float* B, int b_major,
                                       int ldb, float* C, int ldc, int M, int N, int K, int tile,
                                       float* colstats, int* stat_rows, void* stream) {
  if (!colstats || !stat_rows) return (int)hipErrorInvalidValue;
  return gemm_impl(A, a_major, lda, B, b_major, ldb, C, ldc, M, N, K, 1, nullptr, 0, nullptr, tile, colstats,
                   stat_rows, stream);
}

// ------------------------------------------------------------------------------------------------
// im2col / col2im for block1_conv2 (3x3, stride 1, VALID, Cin % 4 == 0) and the stride-2 row
// gather / scatter of the 1x1 residual convs (TF SAME on a 1x1/s2 conv samples the even pixels).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const float* __restrict__ x,
                                                        float* __restrict__ col, int Bn, int H, int W,
                                                        int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)Bn * OH * OW * 9 * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int tap = (int)(t % 9);
    t /= 9;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    const int kh = tap / 3, kw = tap % 3;
    const float4 v = *reinterpret_cast<const float4*>(
        x + (((long)b * H + oh + kh) * W + ow + kw) * C + c4 * 4);
    *reinterpret_cast<float4*>(col + i * 4) = v;
  }
}

__global__ __launch_bounds__(256) void col2im3x3_kernel(const float* __restrict__ dcol,
                                                        float* __restrict__ dx, int Bn, int H, int W,
                                                        int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)Bn * H * W * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    const int b = (int)(t / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int oh = h - kh;
      if (oh < 0 || oh >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ow = w - kw;
        if (ow < 0 || ow >= OW) continue;
        const float4 v = *reinterpret_cast<const float4*>(
            dcol + (((long)b * OH + oh) * OW + ow) * (9L * C) + (kh * 3 + kw) * C + c4 * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dx + i * 4) = s;
  }
}

__global__ __launch_bounds__(256) void gather_s2_kernel(const float* __restrict__ x,
                                                        float* __restrict__ xs, int Bn, int H, int W,
                                                        int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)Bn * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    *reinterpret_cast<float4*>(xs + i * 4) = *reinterpret_cast<const float4*>(
        x + (((long)b * H + 2 * oh) * W + 2 * ow) * C + c4 * 4);
  }
}

// dx[b, 2oh, 2ow, :] += dxs[b, oh, ow, :]   (each destination touched by exactly one thread)
__global__ __launch_bounds__(256) void scatter_add_s2_kernel(const float* __restrict__ dxs,
                                                             float* __restrict__ dx, int Bn, int H,
                                                             int W, int C, int OH, int OW) {
  const int c4n = C / 4;
  const long total = (long)Bn * OH * OW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int ow = (int)(t % OW);
    t /= OW;
    const int oh = (int)(t % OH);
    const int b = (int)(t / OH);
    float4* d = reinterpret_cast<float4*>(dx + (((long)b * H + 2 * oh) * W + 2 * ow) * C + c4 * 4);
    const float4 g = *reinterpret_cast<const float4*>(dxs + i * 4);
    float4 v = *d;
    v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
    *d = v;
  }
}

extern "C" int spnet_im2col3x3(const float* x, float* col, int B, int H, int W, int C, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const int OH = H - 2, OW = W - 2;
  const long total = (long)B * OH * OW * 9 * (C / 4);
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, col, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_col2im3x3(const float* dcol, float* dx, int B, int H, int W, int C, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const int OH = H - 2, OW = W - 2;
  const long total = (long)B * H * W * (C / 4);
  hipLaunchKernelGGL(col2im3x3_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dcol, dx, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_gather_s2(const float* x, float* xs, int B, int H, int W, int C, void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(gather_s2_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, xs, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_scatter_add_s2(const float* dxs, float* dx, int B, int H, int W, int C,
                                    void* stream) {
  if (C & 3) return (int)hipErrorInvalidValue;
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const long total = (long)B * OH * OW * (C / 4);
  hipLaunchKernelGGL(scatter_add_s2_kernel, dim3(spnet_ew_grid(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, dxs, dx, B, H, W, C, OH, OW);
  SPNET_RETURN_LAUNCH_STATUS();
}
