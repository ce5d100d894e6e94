// Probe (round 5): the bf16x3 GEMM with BOTH operands given as three bf16 planes ("planes x planes").
//   C[M][N] (fp32) = sum over the six piece products of A_pa[M][Kp] * B_pb[N][Kp]^T, the term and tile order of
//   gemm_bf16x3_fwd_kernel (csrc/gemm_bf16x3.hip), so the result is that kernel's bit for bit when A's planes are the
//   round-to-nearest-even split of its fp32 A.
// What it removes from the product kernel: the in-kernel split of A (2.8 vector instructions per MFMA), the stage
// registers (two sets of 3 float4 + 5 uint4) and every ds_write -- both operands travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB = 16 rows x 64 B per wave instruction), the XOR swizzle of the 16-byte chunks applied on
// the SOURCE address (cdna_hip_programming.md rule 21).  What it adds: 6 instead of 4 bytes per A element.
// Variants (template): 0 = one K step ahead, wait + barrier per step; 1 = fragments of the next K step read before the
// barrier... (see x3pp_probe.py for the table).
#include "common.h"

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

#define PP_BM 96
#define PP_BN 96
#define PP_LDR 32
#define PP_PLANE (PP_BM * PP_LDR)
#define PP_SWZ(ROW_, CHUNK_) ((((CHUNK_) ^ (0 - ((ROW_) >> 2))) & 3) * 8)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// A planes: Ap + p * a_ps + row * lda (elements), B planes likewise.  Kp % 32 == 0, the pad columns of both are zero.
// WM x WN waves of 48 x 48 (3 x 3 MFMA tiles of 16 x 16 x 32): <2, 2> = the product kernel's 96 x 96 tile, two workgroups
// per CU; <2, 4> = 96 x 192 under one 512-thread workgroup per CU (one A tile for twice the columns).
// SPREAD: the LDS-DMA issues of the next K step spread between the MFMAs instead of ahead of the fragment reads.
template <int WM, int WN, int SPREAD>
__global__ __launch_bounds__(WM * WN * 64, 2) void x3pp_kernel(const unsigned short* __restrict__ Ap, long a_ps, int lda,
                                                      const unsigned short* __restrict__ Bp, long b_ps, int ldb,
                                                      float* __restrict__ C, int ldc, int M, int N, int Kp, int tiles_n) {
  constexpr int BM = 48 * WM, BN = 48 * WN, NW = WM * WN;
  constexpr int STEP = 3 * (BM + BN) * PP_LDR;            // bf16 elements per K step
  constexpr int PIECES = 3 * (BM + BN) / 16, PA = 3 * BM / 16, PER = (PIECES + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = Kp / 32;

  // LDS-DMA pieces of a K step: 1 KiB = 16 rows x 64 B each, A planes first; wave w moves pieces w, w + NW, ...
  // lane -> (row lane / 4 of the group, 16-byte chunk position lane % 4); the chunk that belongs there is XOR-swizzled.
  const int prow = lane >> 2;
  const int pchunk = ((lane & 3) ^ (0 - (lane >> 4))) & 3;
  const unsigned short* src[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = min(wave + NW * i, PIECES - 1);
    if (c < PA) {
      const int plane = c / (BM / 16), row = (c % (BM / 16)) * 16 + prow;
      src[i] = Ap + plane * a_ps + (long)min(m0 + row, M - 1) * lda + pchunk * 8;
    } else {
      const int plane = (c - PA) / (BN / 16), row = ((c - PA) % (BN / 16)) * 16 + prow;
      src[i] = Bp + plane * b_ps + (long)min(n0 + row, N - 1) * ldb + pchunk * 8;
    }
  }
#define PP_ISSUE(KS_, BUF_)                                                                                    \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * STEP;                                                        \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                          \
      const int c = min(wave + NW * i, PIECES - 1); /* (a surplus wave repeats the last piece: same bytes) */  \
        __builtin_amdgcn_global_load_lds((const void*)(src[i] + (KS_) * 32), (lds_ptr_t)(base_ + c * 512), 16, 0, 0); \
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

  PP_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (!SPREAD && more) PP_ISSUE(ks + 1, ks + 1);
    const unsigned short* base = smem + (ks & 1) * STEP;
    bf16x8 af[3][3], bfr[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[t][p] = *reinterpret_cast<const bf16x8*>(base + (p * BM + wm * 48 + t * 16 + p16) * PP_LDR + PP_SWZ(p16, kg));
        bfr[t][p] = *reinterpret_cast<const bf16x8*>(base + (3 * BM + p * BN + wn * 48 + t * 16 + p16) * PP_LDR + PP_SWZ(p16, kg));
      }
    if (SPREAD) PP_ISSUE(min(ks + 1, nk - 1), ks + 1);   // (last step: a tile nobody reads, no branch)
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][TA[t]], bfr[j][TB[t]], acc[i][j], 0, 0, 0);
    if (SPREAD) {
#pragma unroll
      for (int g = 0; g < PER; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // (register-only MFMAs otherwise sink below the barrier: rule 18)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

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
}

// Knock-out builds of the simple form (timing only, results wrong): KO bit 0 = no LDS-DMA inside the loop, bit 1 = no
// barrier, bit 2 = fragments read once ahead of the loop, bit 3 = no MFMAs (fragments kept alive), bit 4 = B pieces only.
template <int KO>
__global__ __launch_bounds__(256, 2) void x3pp_ko_kernel(const unsigned short* __restrict__ Ap, long a_ps, int lda,
                                                         const unsigned short* __restrict__ Bp, long b_ps, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int N, int Kp, int tiles_n) {
  constexpr int WM = 2, WN = 2, BM = 96, BN = 96, NW = 4;
  constexpr int STEP = 3 * (BM + BN) * PP_LDR;
  constexpr int PIECES = 36, PA = 18, PER = 9;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = Kp / 32;
  const int prow = lane >> 2;
  const int pchunk = ((lane & 3) ^ (0 - (lane >> 4))) & 3;
  const unsigned short* src[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = min(wave + NW * i, PIECES - 1);
    if (c < PA) {
      const int plane = c / (BM / 16), row = (c % (BM / 16)) * 16 + prow;
      src[i] = Ap + plane * a_ps + (long)min(m0 + row, M - 1) * lda + pchunk * 8;
    } else {
      const int plane = (c - PA) / (BN / 16), row = ((c - PA) % (BN / 16)) * 16 + prow;
      src[i] = Bp + plane * b_ps + (long)min(n0 + row, N - 1) * ldb + pchunk * 8;
    }
  }
  f32x4v acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  PP_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 af[3][3], bfr[3][3];
#define KO_READ(KS_)                                                                                           \
  do {                                                                                                         \
    const unsigned short* base = smem + ((KS_) & 1) * STEP;                                                    \
    _Pragma("unroll") for (int t = 0; t < 3; ++t)                                                              \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                            \
      af[t][p] = *reinterpret_cast<const bf16x8*>(base + (p * BM + wm * 48 + t * 16 + p16) * PP_LDR + PP_SWZ(p16, kg));              \
      bfr[t][p] = *reinterpret_cast<const bf16x8*>(base + (3 * BM + p * BN + wn * 48 + t * 16 + p16) * PP_LDR + PP_SWZ(p16, kg));    \
    }                                                                                                          \
  } while (0)
  if (KO & 4) KO_READ(0);
  for (int ks = 0; ks < nk; ++ks) {
    if (!(KO & 1) && ks + 1 < nk) {
      unsigned short* base_ = smem + ((ks + 1) & 1) * STEP;
#pragma unroll
      for (int i = (KO & 16) ? 5 : 0; i < PER; ++i) {
        const int c = min(wave + NW * i, PIECES - 1);
        __builtin_amdgcn_global_load_lds((const void*)(src[i] + (ks + 1) * 32), (lds_ptr_t)(base_ + c * 512), 16, 0, 0);
      }
    }
    if (!(KO & 4)) KO_READ(ks);
    if (!(KO & 8)) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][TA[t]], bfr[j][TB[t]], acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p) asm volatile("" ::"v"(af[t][p]), "v"(bfr[t][p]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(KO & 2)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
  }
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
}

// Software-pipelined form: the fragments of K step ks + 1 are read from LDS into a second register set WHILE the MFMAs
// of step ks run, and the LDS-DMA of step ks + 2 is issued into the buffer step ks has just been read out of -- one
// wait + barrier at the top of a step (the DMA it waits for was issued a whole step earlier, the fragment reads it waits
// for likewise), nothing between the MFMAs but issue slots.
// TILED: the planes are stored the way the LDS image wants them -- [plane][group of 16 rows][K step][1 KiB piece], the
// XOR swizzle already applied inside the piece -- so a wave's LDS-DMA instruction reads 1 KiB of CONTIGUOUS global memory
// (eight full 128-byte lines) instead of 16 rows x 64 bytes (half lines, each line touched again by the next K step).
// PIPE = 0: the simple loop (DMA one step ahead, fragments read after the barrier).
template <int WM, int WN, int TILED, int PIPE>
__global__ __launch_bounds__(WM * WN * 64, 2) void x3pp2_kernel(const unsigned short* __restrict__ Ap, long a_ps, int lda,
                                                       const unsigned short* __restrict__ Bp, long b_ps, int ldb,
                                                       float* __restrict__ C, int ldc, int M, int N, int Kp, int tiles_n) {
// (the body is compiled in the device pass only: in the host pass hipcc cannot instantiate the buffer LDS-DMA builtin
// with template-dependent arguments, says nothing, and leaves the kernel's launch stub undefined)
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 48 * WM, BN = 48 * WN, NW = WM * WN;
  constexpr int STEP = 3 * (BM + BN) * PP_LDR;
  constexpr int PIECES = 3 * (BM + BN) / 16, PA = 3 * BM / 16, PER = (PIECES + NW - 1) / NW;
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * STEP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = Kp / 32;
  // LDS-DMA through buffer descriptors (one per operand, all three planes): the lane part of a piece's address is ONE
  // 32-bit offset per operand (row lane / 4 of the 16-row group, swizzled 16-byte chunk), the piece part (plane, row
  // group, K step) is scalar.  Rows past the operand's last read the next plane's rows (finite values that only reach
  // masked outputs) or, past the third plane, zeros (the descriptor's range check): no clamps, no 64-bit pointers.
  const int prow = lane >> 2;
  const int pchunk = ((lane & 3) ^ (0 - (lane >> 4))) & 3;
  const int voff_a = TILED ? lane * 16 : (prow * lda + pchunk * 8) * 2, voff_b = TILED ? lane * 16 : (prow * ldb + pchunk * 8) * 2;
  const int rg_a = (M + 15) / 16, rg_b = (N + 15) / 16;       // (TILED: a_ps = rg_a * nk * 512 elements)
  constexpr int KSTRIDE = TILED ? 1024 : 64;                  // bytes from one K step's piece to the next
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)((TILED ? 3 * a_ps : 2 * a_ps + (long)M * lda) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)((TILED ? 3 * b_ps : 2 * b_ps + (long)N * ldb) * 2), 0x00020000);
  int soff[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = min(wave + NW * i, PIECES - 1);
    if (TILED)
      soff[i] = c < PA ? (int)((c / (BM / 16)) * a_ps + (long)min(m0 / 16 + c % (BM / 16), rg_a - 1) * nk * 512) * 2
                       : (int)(((c - PA) / (BN / 16)) * b_ps + (long)min(n0 / 16 + (c - PA) % (BN / 16), rg_b - 1) * nk * 512) * 2;
    else
      soff[i] = c < PA ? (int)((c / (BM / 16)) * a_ps + (long)(m0 + (c % (BM / 16)) * 16) * lda) * 2
                       : (int)(((c - PA) / (BN / 16)) * b_ps + (long)(n0 + ((c - PA) % (BN / 16)) * 16) * ldb) * 2;
  }
// (which operand a piece belongs to is a compile-time fact for all but one piece per wave: a branch per piece would
// cut the K step into basic blocks and nothing could be scheduled between the MFMAs)
#define PP2_ISSUE(KS_, BUF_)                                                                                   \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * STEP;                                                        \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                          \
      const int c = min(wave + NW * i, PIECES - 1);                                                            \
      const bool is_a = (NW * i + NW - 1 < PA) ? true : (NW * i >= PA) ? false : (c < PA);                     \
      const __amdgpu_buffer_rsrc_t rs_ = is_a ? rs_a : rs_b;                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(base_ + c * 512), 16, is_a ? voff_a : voff_b, \
                                               soff[i] + (KS_) * KSTRIDE, 0, 0);                               \
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
  const int a_off = (wm * 48 + p16) * PP_LDR + PP_SWZ(p16, kg);                 // + (p * BM + t * 16) * PP_LDR
  const int b_off = (3 * BM + wn * 48 + p16) * PP_LDR + PP_SWZ(p16, kg);        // + (p * BN + t * 16) * PP_LDR
#define PP2_READ(KS_, FA_, FB_)                                                                                \
  do {                                                                                                         \
    const unsigned short* base_ = smem + ((KS_) & 1) * STEP;                                                   \
    _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                              \
    _Pragma("unroll") for (int t = 0; t < 3; ++t) {                                                            \
      FA_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + a_off + (p * BM + t * 16) * PP_LDR);                \
      FB_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + b_off + (p * BN + t * 16) * PP_LDR);                \
    }                                                                                                          \
  } while (0)
#define PP2_MFMA(FA_, FB_)                                                                                     \
  _Pragma("unroll") for (int t = 0; t < 6; ++t)                                                                \
  _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                                \
  _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                                \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA_[i][TA[t]], FB_[j][TB[t]], acc[i][j], 0, 0, 0)
// steady state: DMA of step KS + 2, fragment reads of step KS + 1, MFMAs of step KS, spread: 3 MFMAs per fragment read,
// one DMA piece per 6 MFMAs
#define PP2_BODY(KS_, CA_, CB_, NA_, NB_, DMA_, RD_)                                                           \
  do {                                                                                                         \
    __syncthreads(); /* vmcnt(0): DMA of KS + 1 landed; lgkmcnt(0): own reads of KS done; every wave past both */ \
    if (DMA_) PP2_ISSUE((KS_) + 2, (KS_));                                                                      \
    if (RD_) PP2_READ((KS_) + 1, NA_, NB_);                                                                    \
    PP2_MFMA(CA_, CB_);                                                                                        \
    if ((DMA_) && (RD_)) {                                                                                     \
      _Pragma("unroll") for (int g = 0; g < 18; ++g) {                                                         \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        if (g % 2 == 0 && g / 2 < PER) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)

  if (!PIPE) {
    bf16x8 fa[3][3], fb[3][3];
    PP2_ISSUE(0, 0);
    for (int ks = 0; ks < nk; ++ks) {
      __syncthreads();                       // vmcnt(0): the DMA of step ks has landed; every wave is done with step ks - 1
      PP2_ISSUE(min(ks + 1, nk - 1), ks + 1);   // (last step: a tile nobody reads, no branch)
      PP2_READ(ks, fa, fb);
      PP2_MFMA(fa, fb);
      _Pragma("unroll") for (int g = 0; g < PER; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  bf16x8 xa[3][3], xb[3][3], ya[3][3], yb[3][3];
  PP2_ISSUE(0, 0);
  if (nk > 1) {
    PP2_ISSUE(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  PP2_READ(0, xa, xb);
  int ks = 0;
  for (; ks + 3 < nk; ks += 2) {
    PP2_BODY(ks, xa, xb, ya, yb, true, true);
    PP2_BODY(ks + 1, ya, yb, xa, xb, true, true);
  }
  for (; ks < nk; ks += 2) {
    if (ks + 2 < nk) PP2_BODY(ks, xa, xb, ya, yb, true, true);
    else if (ks + 1 < nk) PP2_BODY(ks, xa, xb, ya, yb, false, true);
    else PP2_BODY(ks, xa, xb, ya, yb, false, false);
    if (ks + 1 < nk) {
      if (ks + 3 < nk) PP2_BODY(ks + 1, ya, yb, xa, xb, true, true);
      else if (ks + 2 < nk) PP2_BODY(ks + 1, ya, yb, xa, xb, false, true);
      else PP2_BODY(ks + 1, ya, yb, xa, xb, false, false);
    }
  }
  }

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
#endif
}

// Bigger WAVE tiles (round 5, second half): per K step a CU needs as many LDS cycles (fragment reads 8 waves x 18 KB + the
// DMA's 72 KB at 128 B per clock = 1,728 clocks) as MFMA cycles (108 per SIMD x 16 = 1,728): both pipes at ~67 % is what
// 1.29 us per step is.  The MFMA work is fixed; the fragment traffic is not: a wave that owns TI x TJ MFMA tiles reads
// (TI + TJ) x 3 KiB per step for TI x TJ x 6 MFMAs.  3 x 3 (48 x 48): 18 KB for 54; 6 x 3 (96 x 48): 27 KB for 108.
// Tiled planes only; one wave per SIMD when the workgroup has four waves and OCC = 1 (accumulators in AGPRs, two fragment
// register sets in VGPRs).
template <int TI, int TJ, int WM, int WN, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC) void x3pp3_kernel(const unsigned short* __restrict__ Ap, long a_ps,
                                                         const unsigned short* __restrict__ Bp, long b_ps,
                                                         float* __restrict__ C, int ldc, int M, int N, int Kp, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 16 * TI * WM, BN = 16 * TJ * WN, NW = WM * WN;
  constexpr int STEP = 3 * (BM + BN) * PP_LDR;
  constexpr int PIECES = 3 * (BM + BN) / 16, PA = 3 * BM / 16, PER = (PIECES + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem3[];
  unsigned short* smem = smem3;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = lid % tiles_n, tm = lid / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = Kp / 32;
  const int rg_a = (M + 15) / 16, rg_b = (N + 15) / 16;
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)(3 * a_ps * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)Bp, 0, (int)(3 * b_ps * 2), 0x00020000);
  int soff[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = min(wave + NW * i, PIECES - 1);
    soff[i] = c < PA ? (int)((c / (BM / 16)) * a_ps + (long)min(m0 / 16 + c % (BM / 16), rg_a - 1) * nk * 512) * 2
                     : (int)(((c - PA) / (BN / 16)) * b_ps + (long)min(n0 / 16 + (c - PA) % (BN / 16), rg_b - 1) * nk * 512) * 2;
  }
#define PP3_ISSUE(KS_, BUF_)                                                                                   \
  do {                                                                                                         \
    unsigned short* base_ = smem + ((BUF_) & 1) * STEP;                                                        \
    _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                                          \
      const int c = min(wave + NW * i, PIECES - 1);                                                            \
      const bool is_a = (NW * i + NW - 1 < PA) ? true : (NW * i >= PA) ? false : (c < PA);                     \
      const __amdgpu_buffer_rsrc_t rs_ = is_a ? rs_a : rs_b;                                                   \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (lds_ptr_t)(base_ + c * 512), 16, lane * 16,               \
                                               soff[i] + (KS_) * 1024, 0, 0);                                  \
    }                                                                                                          \
  } while (0)
  f32x4v acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  const int p16 = lane & 15, kg = lane >> 4;
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  const int a_off = (wm * 16 * TI + p16) * PP_LDR + PP_SWZ(p16, kg);
  const int b_off = (3 * BM + wn * 16 * TJ + p16) * PP_LDR + PP_SWZ(p16, kg);
#define PP3_READ(KS_, FA_, FB_)                                                                                \
  do {                                                                                                         \
    const unsigned short* base_ = smem + ((KS_) & 1) * STEP;                                                   \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                            \
      _Pragma("unroll") for (int t = 0; t < TI; ++t)                                                           \
        FA_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + a_off + (p * BM + t * 16) * PP_LDR);              \
      _Pragma("unroll") for (int t = 0; t < TJ; ++t)                                                           \
        FB_[t][p] = *reinterpret_cast<const bf16x8*>(base_ + b_off + (p * BN + t * 16) * PP_LDR);              \
    }                                                                                                          \
  } while (0)
#define PP3_MFMA(FA_, FB_)                                                                                     \
  _Pragma("unroll") for (int t = 0; t < 6; ++t)                                                                \
  _Pragma("unroll") for (int i = 0; i < TI; ++i)                                                               \
  _Pragma("unroll") for (int j = 0; j < TJ; ++j)                                                               \
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FA_[i][TA[t]], FB_[j][TB[t]], acc[i][j], 0, 0, 0)
  constexpr int NMF = 6 * TI * TJ, NRD = 3 * (TI + TJ);
  constexpr int PER_RD = NMF / NRD;                      // MFMAs per fragment read
#define PP3_BODY(KS_, CA_, CB_, NA_, NB_, DMA_, RD_)                                                           \
  do {                                                                                                         \
    __syncthreads();                                                                                           \
    if (DMA_) PP3_ISSUE((KS_) + 2, (KS_));                                                                      \
    if (RD_) PP3_READ((KS_) + 1, NA_, NB_);                                                                    \
    PP3_MFMA(CA_, CB_);                                                                                        \
    if ((DMA_) && (RD_)) {                                                                                     \
      _Pragma("unroll") for (int g = 0; g < NRD; ++g) {                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        if (g % 2 == 0 && g / 2 < PER) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                      \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                     \
        __builtin_amdgcn_sched_group_barrier(0x008, PER_RD - 2, 0);                                            \
      }                                                                                                        \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
  } while (0)
  bf16x8 xa[TI][3], xb[TJ][3], ya[TI][3], yb[TJ][3];
  PP3_ISSUE(0, 0);
  if (nk > 1) {
    PP3_ISSUE(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  PP3_READ(0, xa, xb);
  int ks = 0;
  for (; ks + 3 < nk; ks += 2) {
    PP3_BODY(ks, xa, xb, ya, yb, true, true);
    PP3_BODY(ks + 1, ya, yb, xa, xb, true, true);
  }
  for (; ks < nk; ks += 2) {
    if (ks + 2 < nk) PP3_BODY(ks, xa, xb, ya, yb, true, true);
    else if (ks + 1 < nk) PP3_BODY(ks, xa, xb, ya, yb, false, true);
    else PP3_BODY(ks, xa, xb, ya, yb, false, false);
    if (ks + 1 < nk) {
      if (ks + 3 < nk) PP3_BODY(ks + 1, ya, yb, xa, xb, true, true);
      else if (ks + 2 < nk) PP3_BODY(ks + 1, ya, yb, xa, xb, false, true);
      else PP3_BODY(ks + 1, ya, yb, xa, xb, false, false);
    }
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + (wm * TI + i) * 16 + kg * 4 + r;
      if (row < M) {
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int col = n0 + (wn * TJ + j) * 16 + p16;
          if (col < N) C[(long)row * ldc + col] = acc[i][j][r];
        }
      }
    }
#endif
}

// The product kernel (csrc/gemm_bf16x3.hip: fp32 A split in the kernel) with B read from TILED planes: same registers,
// same stores, but a wave's B load reads 1 KiB of contiguous memory.
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
#define X3_BSRC(S_) (Bp + ((S_) / 384) * plane_stride + ((long)min(n0 / 16 + ((S_) % 384) / 64, rg_b - 1) * nk + (KS_X)) * 512 + ((S_) % 64) * 8)
#define X3_BDST(S_) (base_ + (3 + (S_) / 384) * X3_PLANE + ((S_) % 384) * 8)

__global__ __launch_bounds__(256, 2) void x3ab_kernel(const float* __restrict__ A, int lda,
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
  const int nk = Kp / 32, rg_b = (N + 15) / 16;
  const long plane_stride = (long)rg_b * nk * 512;

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
    const int k0_ = (KS_) * 32;  const int KS_X = (KS_);                                                       \
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


// planes [3][R][Kp] -> tiled planes [3][ceil(R / 16)][Kp / 32][16 rows x 32 bf16, chunks swizzled]; rows past R are zero
__global__ __launch_bounds__(256) void x3pp_retile_kernel(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst,
                                                          int R, int Kp) {
  const int nk = Kp / 32, rgs = (R + 15) / 16;
  const long per_plane = (long)rgs * nk * 512, total = 3 * per_plane / 8;        // 16-byte chunks
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long e = i * 8;
    const int p = (int)(e / per_plane);
    const long q = e % per_plane;
    const int rg = (int)(q / (nk * 512)), ks = (int)((q / 512) % nk), pos = (int)(q % 512);
    const int row16 = pos / 32, cpos = (pos % 32) / 8;
    const int chunk = (cpos ^ (0 - (row16 >> 2))) & 3;
    const int r = rg * 16 + row16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < R) v = *reinterpret_cast<const uint4*>(src + ((long)p * R + r) * Kp + ks * 32 + chunk * 8);
    *reinterpret_cast<uint4*>(dst + e) = v;
  }
}
extern "C" int probe_x3pp_retile(const void* src, void* dst, int R, int Kp, void* stream) {
  hipLaunchKernelGGL(x3pp_retile_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)src,
                     (unsigned short*)dst, R, Kp);
  return (int)hipGetLastError();
}

extern "C" int probe_x3pp(int variant, const void* Ap, long a_ps, int lda, const void* Bp, long b_ps, int ldb, float* C, int ldc,
                          int M, int N, int Kp, void* stream) {
  if (!Ap || !Bp || !C || M < 1 || N < 1 || Kp < 32 || (Kp & 31) || (variant != 20 && (lda & 7)) || (ldb & 7) || (a_ps & 7) || (b_ps & 7))
    return (int)hipErrorInvalidValue;
  if ((((uintptr_t)Ap) | ((uintptr_t)Bp)) & 15) return (int)hipErrorInvalidValue;
  const unsigned short* a = reinterpret_cast<const unsigned short*>(Ap);
  const unsigned short* b = reinterpret_cast<const unsigned short*>(Bp);
  hipStream_t st = (hipStream_t)stream;
#define PP_LAUNCH(WM_, WN_, SP_)                                                                               \
  do {                                                                                                         \
    const int tm = spnet_cdiv(M, 48 * WM_), tn = spnet_cdiv(N, 48 * WN_);                                      \
    hipLaunchKernelGGL((x3pp_kernel<WM_, WN_, SP_>), dim3(tm * tn), dim3(64 * WM_ * WN_), 0, st, a, a_ps, lda, b, b_ps, ldb, \
                       C, ldc, M, N, Kp, tn);                                                                  \
  } while (0)
#define PP2_LAUNCH(WM_, WN_, TI_, PI_)                                                                         \
  do {                                                                                                         \
    const int tm = spnet_cdiv(M, 48 * WM_), tn = spnet_cdiv(N, 48 * WN_);                                      \
    hipLaunchKernelGGL((x3pp2_kernel<WM_, WN_, TI_, PI_>), dim3(tm * tn), dim3(64 * WM_ * WN_), 0, st, a, a_ps, lda, b, b_ps, ldb,     \
                       C, ldc, M, N, Kp, tn);                                                                  \
  } while (0)
  switch (variant) {
    case 0: PP_LAUNCH(2, 2, 0); break;
    case 1: PP_LAUNCH(2, 2, 1); break;
    case 2: PP_LAUNCH(2, 4, 0); break;
    case 3: PP_LAUNCH(2, 4, 1); break;
    case 4: PP_LAUNCH(4, 2, 0); break;
    case 5: PP2_LAUNCH(2, 2, 0, 1); break;
#define KO_LAUNCH(KO_)                                                                                         \
  do {                                                                                                         \
    const int tm = spnet_cdiv(M, 96), tn = spnet_cdiv(N, 96);                                                  \
    hipLaunchKernelGGL((x3pp_ko_kernel<KO_>), dim3(tm * tn), dim3(256), 0, st, a, a_ps, lda, b, b_ps, ldb, C, ldc, M, N, Kp, tn); \
  } while (0)
    case 100: KO_LAUNCH(0); break;
    case 101: KO_LAUNCH(1); break;       // no DMA
    case 103: KO_LAUNCH(3); break;       // no DMA, no barrier
    case 107: KO_LAUNCH(7); break;       // MFMAs alone
    case 108: KO_LAUNCH(8); break;       // DMA + reads + barrier, no MFMA
    case 112: KO_LAUNCH(12); break;      // DMA + barrier alone
    case 116: KO_LAUNCH(16); break;      // B pieces only
    case 115: KO_LAUNCH(15); break;      // empty loop
    case 6: PP2_LAUNCH(2, 4, 0, 1); break;
    case 7: PP2_LAUNCH(4, 2, 0, 1); break;
    case 8: PP2_LAUNCH(2, 2, 0, 0); break;
    case 20: {          // A = fp32 [M][K] (lda = K in elements), B = tiled planes
      const int tm = spnet_cdiv(M, 96), tn = spnet_cdiv(N, 96);
      hipLaunchKernelGGL(x3ab_kernel, dim3(tm * tn), dim3(256), 0, st, (const float*)Ap, lda, b, Kp, C, ldc, M, N, lda, tn, (float*)nullptr);
    } break;
#define PP3_LAUNCH(TI_, TJ_, WM_, WN_, OCC_)                                                                   \
  do {                                                                                                         \
    constexpr int bm_ = 16 * TI_ * WM_, bn_ = 16 * TJ_ * WN_;                                                  \
    constexpr int lds_ = 2 * 3 * (bm_ + bn_) * PP_LDR * 2;                                                     \
    const int tm = spnet_cdiv(M, bm_), tn = spnet_cdiv(N, bn_);                                                \
    static bool once_ = false;                                                                                 \
    if (!once_) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)x3pp3_kernel<TI_, TJ_, WM_, WN_, OCC_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_); \
      once_ = true;                                                                                            \
    }                                                                                                          \
    hipLaunchKernelGGL((x3pp3_kernel<TI_, TJ_, WM_, WN_, OCC_>), dim3(tm * tn), dim3(64 * WM_ * WN_), lds_, st, a, a_ps, b, b_ps, \
                       C, ldc, M, N, Kp, tn);                                                                  \
  } while (0)
    case 30: PP3_LAUNCH(3, 3, 2, 2, 2); break;   // = v11 (check of the generalisation)
    case 31: PP3_LAUNCH(6, 3, 2, 2, 1); break;   // 192 x 96, four waves of 96 x 48, one per SIMD
    case 32: PP3_LAUNCH(6, 3, 1, 2, 1); break;   // 96 x 96, two waves of 96 x 48, two workgroups per CU (one wave per SIMD)
    case 33: PP3_LAUNCH(3, 6, 2, 2, 1); break;   // 96 x 192, four waves of 48 x 96
    case 34: PP3_LAUNCH(6, 4, 2, 2, 1); break;   // 192 x 128, four waves of 96 x 64
    case 35: PP3_LAUNCH(4, 4, 2, 2, 1); break;   // 128 x 128, four waves of 64 x 64
    case 36: PP3_LAUNCH(3, 3, 4, 2, 1); break;   // 192 x 96, eight waves of 48 x 48 (= v15)
    case 10: PP2_LAUNCH(2, 2, 1, 0); break;      // tiled planes, simple loop
    case 11: PP2_LAUNCH(2, 2, 1, 1); break;      // tiled planes, pipelined
    case 12: PP2_LAUNCH(2, 4, 1, 0); break;
    case 13: PP2_LAUNCH(2, 4, 1, 1); break;
    case 14: PP2_LAUNCH(4, 2, 1, 0); break;
    case 15: PP2_LAUNCH(4, 2, 1, 1); break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}
