// fp32 MFMA GEMM for the pointwise (1x1) convolutions, the stride-2 residual convs and the Dense head
// (block1_conv2 runs the same tile machinery as implicit GEMMs in conv_gemm.hip):  C[M,N] = sum_k A(m,k) * B(k,n)
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
// Each operand's LDS image keeps its own major (see TileStage): no staging pass transposes, and operand
// fetches are conflict-free wide ds_reads (one read feeds up to four MFMAs).
//
// Split-K: grid.z slices write fp32 slabs to a workspace, a second kernel sums them in slice order
// (deterministic; no float atomics).
#include "gemm_tile.h"

#ifndef SP_BK
#define SP_BK 32
#endif
#ifndef SP_PIPE
#define SP_PIPE 1        // 0: the compiler-scheduled main loop everywhere (A/B measurements)
#endif
#ifndef SP_SPLIT_BELOW_TILES
#define SP_SPLIT_BELOW_TILES 256
#define SP_SPLIT_MIN_K 2048
#endif
#ifndef SP_PIPE_MIN_TILES
#define SP_PIPE_MIN_TILES 32
#endif

// 4 waves (WM x WN), each wave owns a (BM/WM) x (BN/WN) block built from 16x16 MFMA tiles
// (v_mfma_f32_16x16x4_f32: D col = l&15, row = 4*(l>>4)+reg).  Tile t+1 travels global -> registers while
// tile t is multiplied, and is written to the idle LDS buffer in the MIDDLE of the MFMA stream (nobody
// reads that buffer during this iteration), so one barrier per K tile suffices.
// AX = 1: the A operand is the blend a[k]*A_ + b[k]*X2_ + c[k] of two tensors (TileStage XF): the BatchNorm-backward
// output dy built on the fly from the incoming gradient g (= A_) and the saved pre-normalisation tensor yp (= X2_),
// coef = [a|b|c] with cld floats each.  dy_out (or NULL): the blended operand is also written to memory once (the
// weight-gradient GEMM of the same layer reads it): K tile t of a row tile is written by the workgroup of column
// tile t % tiles_n, so the extra stores are spread evenly over the workgroups that stage that row tile.
// (blended kernels of the small tiles are held to 3 workgroups per CU like their plain counterparts: the middle
// flow's 768-workgroup launches are exactly one resident wave of 3 per CU)


// N x { one MFMA, PER instructions of class MASK } for the instruction scheduler
template <int N, int MASK, int PER>
__device__ __forceinline__ void sp_interleave() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(MASK, PER, 0);
  }
}

// AG = 1: the A operand is not a matrix in memory but the patch matrix of a convolution, gathered from the NHWC tensor
// while the tile is staged (implicit GEMM): A[m][k] = X[b][oh*s + kh - pt][ow*s + kw - pl][c] with m = (b, oh, ow) and
// k = (kh*KW + kw)*C + c, zero outside the image.  C is a multiple of the K tile (32), so a K tile lies inside ONE tap: the
// tile's fetch is the plain K-major fetch from a shifted base pointer (the row offsets are those of the tap-(0,0) pixels,
// computed once) with the rows whose tap falls outside the image zeroed -- TileStage::load_gather, the same loads in the
// same pipeline slots as the matrix form, no patch matrix, no gather launch.  k runs in the order of the patch matrix,
// so the result has the bits of spnet_patches + spnet_gemm_f32 on the same tile.
struct ConvGeom {
  int H, W, C, KW, stride, pt, pl, OH, OW;      // C == 0: not a convolution
};

// PIPE = 1: the explicitly software-pipelined main loop (below) instead of the compiler-scheduled one.  It needs four
// K tiles of lead-in and pays from about 32 K tiles per workgroup (the weight gradients, the exit flow): measured per
// shape with tools/gemm_table.py, +3..8 % there, -10..20 % on the 2-8-tile GEMMs of the entry flow, equal at 23 tiles.
template <int BM, int BN, int BK, int WM, int WN, int AMAJ, int BMAJ, int AX = 0, int PIPE = 0, int AG = 0>
__global__ __launch_bounds__(256, (AX && BM * BN <= 96 * 64) ? 3 : 2) void gemm_f32_kernel(const float* __restrict__ A_, int lda,
                                                       const float* __restrict__ B_, int ldb,
                                                       float* __restrict__ C_, int ldc, int M, int N,
                                                       int K, int k_chunk, long slab_stride,
                                                       int tiles_m, int tiles_n, int nsplit,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ colstats,
                                                       const long long* __restrict__ batch,
                                                       const float* __restrict__ X2,
                                                       const float* __restrict__ coef, int cld,
                                                       float* __restrict__ dy_out, int kslices, int accumulate,
                                                       const ConvGeom cg) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(!AG || (!AX && AMAJ == SP_K_MAJOR), "gathered A operands are plain and K-major");
  static_assert(!AX || AMAJ == SP_K_MAJOR, "blended A operands are K-major");
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(TM >= 1 && TN >= 1 && (BM % (WM * 16)) == 0 && (BN % (WN * 16)) == 0, "wave tile");
  typedef TileStage<BM, BK, AMAJ, TM, 0, AX> SA;
  typedef TileStage<BN, BK, BMAJ, TN, (BMAJ == SP_K_MAJOR)> SB;
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  constexpr int NCH = BK / 16;
  // (+ a two-slot ring of blend coefficients: [a | b | c] x BK floats per K tile)
  constexpr int CFS = 3 * BK;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE + (AX ? 2 * CFS : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nblk = tiles_m * tiles_n * nsplit;
  int lid = xcd_remap(blockIdx.x, nblk);
  const int gwg = blockIdx.x;
  const int tn = lid % tiles_n;
  lid /= tiles_n;
  const int tm = lid % tiles_m;
  const int z = lid / tiles_m;

  const int m0 = tm * BM, n0 = tn * BN;
  // batch != NULL: grid.z-like index z selects one of nsplit / kslices independent problems of identical shape
  // (operand pointers from the table {A0,B0,C0,A1,B1,C1,...}) and one of its `kslices` K slices; otherwise z is the
  // K slice.  (the table holds element OFFSETS from problem 0's operands, which arrive as kernel arguments: pointers
  // read from memory would be generic and turn every operand fetch into a flat_load, which also counts on
  // lgkmcnt and serialises with the LDS traffic)
  // Batched problems with a K split (kslices > 1) write fp32 slabs: C_ is then the slab workspace, slab z.
  const int bz = batch ? z / kslices : 0;
  const int ks = batch ? z - bz * kslices : z;
  const float* __restrict__ A = batch ? A_ + batch[3 * bz] : A_;
  const float* __restrict__ B = batch ? B_ + batch[3 * bz + 1] : B_;
  float* __restrict__ C = (batch && kslices == 1) ? C_ + batch[3 * bz + 2] : C_;
  const int kbeg = ks * k_chunk;
  const int kend = min(K, kbeg + k_chunk);
  const int nt = (kend - kbeg + BK - 1) / BK;
  const int zs = batch ? (kslices > 1 ? z : 0) : z;     // slab index

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  static_assert(!PIPE || (!AX && TM * TN < 16), "no pipelined form of the blended kernels and of the 128x128 tiles "
                                                "(a second fragment set does not fit their register budget)");
  // Two register stages: while tile t is multiplied, tile t+1 waits in one stage (it is written to the
  // idle LDS buffer in the middle of the MFMA stream) and tile t+2 is already being fetched into the
  // other, so a global load has one and a half K tiles of MFMA time to arrive.
  SA sa[2];
  SB sb[2];
  sa[0].init(lda, m0, M, tid);
  sb[0].init(ldb, n0, N, tid);
#pragma unroll
  for (int i = 0; i < SA::NV; ++i) sa[1].off[i] = sa[0].off[i];
#pragma unroll
  for (int i = 0; i < SB::NV; ++i) sb[1].off[i] = sb[0].off[i];
  // Gathered A: per slot, the element offset of the slot's output pixel at tap (0,0) without padding (lda = the pixel
  // stride of X) and that pixel's input row / column, from which a tap's validity follows.
  int goff[AG ? SA::NV : 1], gh[AG ? SA::NV : 1], gw[AG ? SA::NV : 1];
  if constexpr (AG) {
    static_assert(SA::TOTAL % 256 == 0, "whole float4 slots per thread");
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) {
      const int f = tid + i * 256;
      const int r = min(m0 + f / (BK / 4), M - 1);
      const int ow = r % cg.OW, t2 = r / cg.OW;
      const int oh = t2 % cg.OH, b = t2 / cg.OH;
      gh[i] = oh * cg.stride;
      gw[i] = ow * cg.stride;
      goff[i] = ((b * cg.H + gh[i]) * cg.W + gw[i]) * lda + (f % (BK / 4)) * 4;
    }
  }
  // The K tiles of a problem are fetched in order (prologue: tiles 0, 1 (, 2, 3); step t: tile t+2 or t+4), so the tap
  // of the next tile is running state -- a few scalar adds per tile; dividing k0 by C and KW on every fetch costs as
  // much as the gather launch it replaces.
  int g_c0 = 0, g_kw = 0, g_dh = -cg.pt, g_dw = -cg.pl;
  auto gather_a = [&](auto& stA, int k0) {          // the next K tile of the patch matrix (inside one tap)
    if constexpr (AG) {
      const int shift = (g_dh * cg.W + g_dw) * lda + g_c0;
      unsigned ok = 0;
#pragma unroll
      for (int i = 0; i < SA::NV; ++i)
        ok |= (unsigned)((unsigned)(gh[i] + g_dh) < (unsigned)cg.H && (unsigned)(gw[i] + g_dw) < (unsigned)cg.W) << i;
      stA.load_gather(A, goff, shift, k0 < kend ? ok : 0u);
      g_c0 += BK;
      if (g_c0 == cg.C) {
        g_c0 = 0;
        ++g_kw;
        ++g_dw;
        if (g_kw == cg.KW) {
          g_kw = 0;
          g_dw = -cg.pl;
          ++g_dh;
        }
      }
    }
  };

  // Blended A operand: tile t+1 waits in its register stage as TWO raw tensors (v = g, w = yp); it is blended -- and,
  // by the column tile whose turn it is, written out as dy -- at the START of step t, before tile t+2's fetch
  // reuses the one shared w set.  The blend coefficients of a K tile (3 x BK floats) travel like the tile itself:
  // fetched by 3*BK/4 threads two tiles ahead, parked in a two-slot LDS ring in the middle of the step, read back
  // (three broadcast ds_read_b128 per thread) one barrier later -- a global load right in front of the blend
  // would put an L2 round trip on every wave's critical path once per K tile.
  float4 wsh[AX ? SA::NV : 1];
  float4 cfreg = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int CFT = 3 * BK / 4;                      // coefficient float4 per K tile
  float* cfs = smem + 2 * STAGE;
  auto cf_fetch = [&](int k0) {                        // this thread's share of tile k0's coefficients
    if constexpr (AX)
      if (tid < CFT) cfreg = *reinterpret_cast<const float4*>(coef + (tid / (BK / 4)) * cld + k0 + (tid % (BK / 4)) * 4);
  };
  auto cf_park = [&](int slot) {
    if constexpr (AX)
      if (tid < CFT) *reinterpret_cast<float4*>(cfs + slot * CFS + tid * 4) = cfreg;
  };
  auto fetch = [&](auto& stA, auto& stB, int k0, auto& wreg) {      // any tile: predicated
    if constexpr (AX) stA.load2(A, X2, lda, m0, M, k0, kend, tid, wreg);
    else if constexpr (AG) gather_a(stA, k0);
    else stA.load(A, lda, m0, M, k0, kend, tid);
    stB.load(B, ldb, n0, N, k0, kend, tid);
  };
  auto blend = [&](auto& stA, int k0, auto& wreg, const float4 a, const float4 b, const float4 c) {
    if constexpr (AX) {
      stA.xform(a, b, c, wreg);
      if (dy_out && ((k0 - kbeg) / BK) % tiles_n == tn) stA.store_global(dy_out, lda, m0, M, k0, kend, tid);
    }
  };
  if (nt > 0) {
    if constexpr (AX) {
      float4 w0[SA::NV];
      const int kq4 = (tid % (BK / 4)) * 4;
      fetch(sa[0], sb[0], kbeg, w0);
      if (nt > 1) {
        fetch(sa[1], sb[1], kbeg + BK, wsh);
        cf_fetch(kbeg + BK);
      }
      blend(sa[0], kbeg, w0, *reinterpret_cast<const float4*>(coef + kbeg + kq4),
            *reinterpret_cast<const float4*>(coef + cld + kbeg + kq4),
            *reinterpret_cast<const float4*>(coef + 2 * cld + kbeg + kq4));
      if (nt > 1) cf_park(1);
    } else {
      fetch(sa[0], sb[0], kbeg, wsh);
      if (nt > 1) fetch(sa[1], sb[1], kbeg + BK, wsh);
    }
    sa[0].store(smem, tid);
    sb[0].store(smem + SA::SIZE, tid);
    if constexpr (PIPE) {                          // tiles 0 and 1 in LDS, tiles 2 and 3 on their way into the stages
      if (nt > 1) {
        sa[1].store(smem + STAGE, tid);
        sb[1].store(smem + STAGE + SA::SIZE, tid);
      }
      if (nt > 2) fetch(sa[0], sb[0], kbeg + 2 * BK, wsh);
      if (nt > 3) fetch(sa[1], sb[1], kbeg + 3 * BK, wsh);
    }
  }
  __syncthreads();

  // PAR = t & 1: tile t sits in LDS buffer PAR, tile t+1 in register stage 1-PAR, stage PAR is free.
  // FULL: tile t+2 exists and lies completely below kend -> its fetch is branch-free, the whole step is one
  // basic block and the compiler interleaves loads, LDS writes and MFMAs.
  const long aoff = (AMAJ == SP_K_MAJOR ? (long)kbeg : (long)kbeg * lda) + (PIPE ? 4 : 2) * SA::kstep(lda);
  const long boff = (BMAJ == SP_K_MAJOR ? (long)kbeg : (long)kbeg * ldb) + (PIPE ? 4 : 2) * SB::kstep(ldb);
  const float* Ak = A + aoff;
  const float* Bk = B + boff;
  const float* Xk = AX ? X2 + aoff : nullptr;      // second tensor of the blended operand
  // Plain (AX = 0) kernels: the main loop is an explicit software pipeline over the two 16-deep chunks of a K tile.
  // Per step t (LDS buffer cur = t & 1 holds tile t; two register stages, stage t & 1 holds tile t+2):
  //   first half    fragments of chunk 1 (cur) -> set 1  |  MFMAs of chunk 0 on set 0
  //   barrier       (every wave has its last fragments of cur; tile t+1 is complete in nxt)
  //   second half   fragments of chunk 0 of tile t+1 (nxt) -> set 0 | stage (tile t+2) -> cur | global loads of tile
  //                 t+4 -> the same stage  |  MFMAs of chunk 1 on set 1
  // Every LDS fragment read is issued a chunk (>= 16 MFMAs) ahead of its first use, the stage is written to LDS TWO
  // steps after its global loads were issued (measured with the loads, the stage stores or the fragment reads knocked
  // out: with ONE step of cover the waves stall 0.5 us per step in front of the stage stores, waiting for
  // first-touch lines of the A operand) and a full step before the barrier that publishes it, so a wave
  // reaches the barrier with nothing outstanding and leaves it with a chunk of MFMAs ready to issue.  The
  // sched_group_barrier sequences pin that order in the branch-free FULL step and spread the LDS / global
  // instructions one per MFMA (an MFMA occupies the SIMD's issue port for 8 of its 32 cycles): left alone, hipcc
  // issues every fragment read directly in front of its first MFMA and sinks the global loads to the end of a step.
  static_assert(NCH == 2, "the pipelined step is written for two chunks per K tile");
  float fa[2][TM][4], fb[2][TN][4];
  constexpr int NMF = TM * TN * 4;                   // MFMAs per chunk
  constexpr int NLD = SA::NV + SB::NV;               // global loads = LDS stores per tile and thread
  constexpr int NFR = SA::NFR + SB::NFR;             // fragment reads per chunk (before ds_read2 merging)
  constexpr int PER = (NFR + 2 * NLD <= NMF) ? 1 : 2;   // LDS / global instructions per MFMA in the interleave
  auto frd0 = [&](const float* buf, int c) {
    SA::frags(buf, wm * (TM * 16), lane, c, fa[0]);
    SB::frags(buf + SA::SIZE, wn * (TN * 16), lane, c, fb[0]);
  };
  auto frd1 = [&](const float* buf, int c) {
    SA::frags(buf, wm * (TM * 16), lane, c, fa[1]);
    SB::frags(buf + SA::SIZE, wn * (TN * 16), lane, c, fb[1]);
  };
  auto mma0 = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0][i][q], fb[0][j][q], acc[i][j], 0, 0, 0);
  };
  auto mma1 = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1][i][q], fb[1][j][q], acc[i][j], 0, 0, 0);
  };
  auto pstep = [&](auto par, auto full, int t) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FULL = decltype(full)::value;   // tile t+4 exists and lies completely below kend
    float* cur = smem + PAR * STAGE;
    const float* nxt = smem + (1 - PAR) * STAGE;
    frd1(cur, 1);
    mma0();
    if (FULL) {
      sp_interleave<(NFR + PER - 1) / PER, 0x100, PER>();
      __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      __builtin_amdgcn_sched_barrier(0);           // nothing crosses: chunk 0 is multiplied in front of the barrier
    }
    __syncthreads();
    if (FULL) __builtin_amdgcn_sched_barrier(0);
    if (FULL || t + 1 < nt) frd0(nxt, 0);
    if (FULL || t + 2 < nt) {
      sa[PAR].store(cur, tid);
      sb[PAR].store(cur + SA::SIZE, tid);
    }
    if (FULL) {
      if constexpr (AG) gather_a(sa[PAR], kbeg + (t + 4) * BK);
      else sa[PAR].load_full(Ak);
      sb[PAR].load_full(Bk);
      Ak += SA::kstep(lda);
      Bk += SB::kstep(ldb);
    } else if (t + 4 < nt) {
      fetch(sa[PAR], sb[PAR], kbeg + (t + 4) * BK, wsh);
    }
    mma1();
    if (FULL) {
      sp_interleave<(NFR + PER - 1) / PER, 0x100, PER>();
      sp_interleave<(NLD + PER - 1) / PER, 0x200, PER>();
      sp_interleave<(NLD + PER - 1) / PER, 0x020, PER>();
      __builtin_amdgcn_sched_group_barrier(0x008, NMF, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto step = [&](auto par, auto full, int t) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FULL = decltype(full)::value;
    const float* cur = smem + PAR * STAGE;
    float* nxt = smem + (1 - PAR) * STAGE;
    const bool more = FULL || (t + 1 < nt);
    if constexpr (AX) {
      if (more) {
        const float4* cs = reinterpret_cast<const float4*>(cfs + (1 - PAR) * CFS) + tid % (BK / 4);
        blend(sa[1 - PAR], kbeg + (t + 1) * BK, wsh, cs[0], cs[BK / 4], cs[2 * (BK / 4)]);
      }
      if (FULL || t + 2 < nt) cf_fetch(kbeg + (t + 2) * BK);
    }
    if (FULL) {
      if constexpr (AX) sa[PAR].load_full2(Ak, Xk, wsh);
      else if constexpr (AG) gather_a(sa[PAR], kbeg + (t + 2) * BK);
      else sa[PAR].load_full(Ak);
      sb[PAR].load_full(Bk);
      Ak += SA::kstep(lda);
      Bk += SB::kstep(ldb);
      if constexpr (AX) Xk += SA::kstep(lda);
    } else if (t + 2 < nt) {
      fetch(sa[PAR], sb[PAR], kbeg + (t + 2) * BK, wsh);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      if (c == NCH / 2 && more) {
        sa[1 - PAR].store(nxt, tid);
        sb[1 - PAR].store(nxt + SA::SIZE, tid);
        if (FULL || t + 2 < nt) cf_park(PAR);          // coefficients of tile t+2 (slot parity of t+2 = PAR)
      }
      float a[TM][4], b[TN][4];
      SA::frags(cur, wm * (TM * 16), lane, c, a);
      SB::frags(cur + SA::SIZE, wn * (TN * 16), lane, c, b);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> P0;
  typedef std::integral_constant<int, 1> P1;
  const int nfull = (kend - kbeg) / BK;            // complete K tiles of this slice
  int t = 0;
  if constexpr (PIPE)
    if (nt > 0) frd0(smem, 0);                     // fragments of tile 0, chunk 0
  if constexpr (!PIPE) {
    for (; t + 3 < nfull; t += 2) {                // tiles t+2 and t+3 are complete
      step(P0(), std::true_type(), t);
      step(P1(), std::true_type(), t + 1);
    }
    for (; t < nt; t += 2) {
      step(P0(), std::false_type(), t);
      if (t + 1 < nt) step(P1(), std::false_type(), t + 1);
    }
  } else {
    for (; t + 5 < nfull; t += 2) {                // tiles t+4 and t+5 are complete
      pstep(P0(), std::true_type(), t);
      pstep(P1(), std::true_type(), t + 1);
    }
    for (; t < nt; t += 2) {
      pstep(P0(), std::false_type(), t);
      if (t + 1 < nt) pstep(P1(), std::false_type(), t + 1);
    }
  }

  gemm_epilogue<SA, SB, BM, BN, WM, WN, TM, TN>(acc, smem, C, ldc, M, N, m0, n0, tm, zs, slab_stride, bias, colstats,
                                                tid, lane, wm, wn, accumulate);
}

// out[row*ldc + col] = sum_z ws[z*M*N + row*N + col] (+ bias[col]); N % 4 == 0, ldc % 4 == 0.
// Workgroup = (256 / sl) output float4 x sl slab lanes (sl = 1..16, a power of two chosen from the slab count):
// lane g adds slabs g, g+sl, g+2sl, ... (up to 4 loads in flight), then the sl lane sums are added in lane
// order -- a fixed order, and a K split into hundreds of slabs (the weight gradients of the entry flow:
// K = 372,000 pixels) is no longer one long serial chain per output element.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ ws, int nslab,
                                                           int M, int N, float* __restrict__ out,
                                                           int ldc, const float* __restrict__ bias, int sl) {
  __shared__ float4 red[256];
  const long total4 = (long)M * N / 4;
  const long mn = (long)M * N;
  const int per = 256 / sl;                         // outputs per workgroup
  const int lane = threadIdx.x % per, g = threadIdx.x / per;
  const long i = (long)blockIdx.x * per + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total4) {
#pragma unroll 4
    for (int zz = g; zz < nslab; zz += sl) {
      const float4 t = *reinterpret_cast<const float4*>(ws + (long)zz * mn + i * 4);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[g * per + lane] = s;
  __syncthreads();
  if (g == 0 && i < total4) {
    float4 t = red[lane];
    for (int k = 1; k < sl; ++k) {
      const float4 v = red[k * per + lane];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    const long e = i * 4;
    const int row = (int)(e / N), col = (int)(e % N);
    if (bias) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + col);
      t.x += bv.x; t.y += bv.y; t.z += bv.z; t.w += bv.w;
    }
    *reinterpret_cast<float4*>(out + (long)row * ldc + col) = t;
  }
}

static void launch_reduce_slabs(const float* ws, int nslab, int M, int N, float* out, int ldc, const float* bias,
                                hipStream_t st) {
  const long total4 = (long)M * N / 4;
  int sl = 1;
  while (sl < 16 && sl * 4 <= nslab) sl <<= 1;      // >= 4 slabs per lane before another lane pays off
  const int per = 256 / sl;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((total4 + per - 1) / per)), dim3(256), 0, st, ws, nslab, M, N,
                     out, ldc, bias, sl);
}

// out[M][ldc] = sum of nslab slabs of M*N floats (slice order); shared with conv_gemm.hip
extern "C" int spnet_reduce_slabs(const float* ws, int nslab, int M, int N, float* out, int ldc, void* stream) {
  if ((N & 3) || (ldc & 3) || nslab < 1) return (int)hipErrorInvalidValue;
  const long total4 = (long)M * N / 4;
  launch_reduce_slabs(ws, nslab, M, N, out, ldc, nullptr, (hipStream_t)stream);
  SPNET_RETURN_LAUNCH_STATUS();
}


// xf: 0 = plain operands, 1 = blended A (forward form only)
template <int BM, int BN, int WM, int WN>
static int launch_tile(const float* A, int amaj, int lda, const float* B, int bmaj, int ldb, float* C,
                       int ldc, int M, int N, int K, int nsplit, int k_chunk, long slab_stride,
                       const float* bias, float* colstats, const long long* batch, hipStream_t st,
                       int xf = 0, const float* X2 = nullptr, const float* coef = nullptr, int cld = 0,
                       float* dy_out = nullptr, int kslices = 1, int accumulate = 0, const ConvGeom* cgp = nullptr) {
  const ConvGeom cg = cgp ? *cgp : ConvGeom{0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int tm = spnet_cdiv(M, BM), tn = spnet_cdiv(N, BN);
  dim3 grid(tm * tn * nsplit), block(256);
  // the pipelined main loop from 32 K tiles per workgroup (see the kernel's header comment)
  constexpr bool CAN_PIPE = SP_PIPE && (BM / WM / 16) * (BN / WN / 16) < 16;
  const bool pipe = CAN_PIPE && !xf && spnet_cdiv(K < k_chunk ? K : k_chunk, SP_BK) >= SP_PIPE_MIN_TILES;
#define SP_ARGS A, lda, B, ldb, C, ldc, M, N, K, k_chunk, slab_stride, tm, tn, nsplit, bias, colstats, batch, X2, coef, cld, dy_out, kslices, accumulate, cg
#define SP_LAUNCH(BKV, AM, BMJ, AXV, PV) \
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BKV, WM, WN, AM, BMJ, AXV, PV>), grid, block, 0, st, SP_ARGS)
#define SP_LAUNCH_P(AM, BMJ)                                          \
  do {                                                                \
    if constexpr (CAN_PIPE) {                                         \
      if (pipe) SP_LAUNCH(SP_BK, AM, BMJ, 0, 1);                      \
      else SP_LAUNCH(SP_BK, AM, BMJ, 0, 0);                           \
    } else {                                                          \
      SP_LAUNCH(SP_BK, AM, BMJ, 0, 0);                                \
    }                                                                 \
  } while (0)
  if (cg.C) {          // gathered A (implicit-GEMM convolution): forward operand form only
    if (xf || amaj != SP_K_MAJOR || bmaj != SP_OUT_MAJOR) return (int)hipErrorInvalidValue;
    if constexpr (CAN_PIPE) {
      if (pipe) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, SP_BK, WM, WN, SP_K_MAJOR, SP_OUT_MAJOR, 0, 1, 1>), grid, block, 0, st, SP_ARGS);
      else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, SP_BK, WM, WN, SP_K_MAJOR, SP_OUT_MAJOR, 0, 0, 1>), grid, block, 0, st, SP_ARGS);
    } else {
      hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, SP_BK, WM, WN, SP_K_MAJOR, SP_OUT_MAJOR, 0, 0, 1>), grid, block, 0, st, SP_ARGS);
    }
  } else if (xf == 1) {
    if (amaj == SP_K_MAJOR && bmaj == SP_OUT_MAJOR) SP_LAUNCH(SP_BK, SP_K_MAJOR, SP_OUT_MAJOR, 1, 0);
    else return (int)hipErrorInvalidValue;
  } else if (amaj == SP_K_MAJOR && bmaj == SP_OUT_MAJOR) SP_LAUNCH_P(SP_K_MAJOR, SP_OUT_MAJOR);
  else if (amaj == SP_K_MAJOR && bmaj == SP_K_MAJOR) SP_LAUNCH_P(SP_K_MAJOR, SP_K_MAJOR);
  else if (amaj == SP_OUT_MAJOR && bmaj == SP_OUT_MAJOR) SP_LAUNCH_P(SP_OUT_MAJOR, SP_OUT_MAJOR);
  else return (int)hipErrorInvalidValue;
#undef SP_LAUNCH_P
#undef SP_LAUNCH
#undef SP_ARGS
  return 0;
}

// Tile ids: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 32x128, 5 = 96x96, 6 = 96x64, 7 = 64x128, 8 = 128x96,
// 9 = 32x64 (few-row problems: M = 384 ... 4,096 rows of Inception-ResNet-v2 at batch 16, where 64-row tiles leave CUs empty),
// 10 = 32x32 (the same problems with few columns as well: twice the workgroups, half the MFMA chain per K tile).
// 9 and 10 are never chosen by the cost model: spnet_amd/gemm_tiles.json names the shapes where they win in the step.
#define SP_NTILES 10
static void tile_dims(int tile, int* bm, int* bn) {
  switch (tile) {
    case 1: *bm = 128; *bn = 128; break;
    case 2: *bm = 128; *bn = 64; break;
    case 3: *bm = 64; *bn = 64; break;
    case 5: *bm = 96; *bn = 96; break;
    case 6: *bm = 96; *bn = 64; break;
    case 7: *bm = 64; *bn = 128; break;
    case 8: *bm = 128; *bn = 96; break;
    case 9: *bm = 32; *bn = 64; break;
    case 10: *bm = 32; *bn = 32; break;
    default: *bm = 32; *bn = 128; break;
  }
}

// Automatic K split of one tile shape: enough workgroups for ~2 per CU, slices at least 4 K-tiles deep -- but only
// while the unsplit launch leaves CUs empty: from one workgroup per CU on, the slabs and their reduce launch cost more
// than the second workgroup buys (exit-flow shapes, tile / split sweep: 1536x1024x728 32.8 -> 26.7 us, 1536x1536x1024
// 51.5 -> 43.9, 1536x1536x2048 in the data-gradient form 87.8 -> 74.4), and not below 64 K tiles, where even a launch
// that fills three quarters of the CUs beats its split form (1536x728x1024, data-gradient form: 35.9 -> 29.3 us).
// (weight-gradient form: few output tiles and a long pixel axis by nature; the original rule -- split below two
// workgroups per CU -- measures better there: 1024x1536x1536 53.2 vs 56.7 us)
static int auto_split(long tiles, int M, int N, int K, bool have_ws, long ws_floats, int form) {
  if (!have_ws) return 1;
  if (form == 2 ? tiles >= 512 : (tiles >= SP_SPLIT_BELOW_TILES || K < SP_SPLIT_MIN_K)) return 1;
  long want = (512 + tiles - 1) / tiles;
  long maxk = K / (SP_BK * 4);
  if (maxk < 1) maxk = 1;
  if (want > maxk) want = maxk;
  const long fit = ws_floats / ((long)M * N);
  if (want > fit) want = fit;
  return want > 1 ? (int)want : 1;
}

// Tile choice: cost ~ (workgroups on the busiest CU) x (work per workgroup + a fixed prologue/epilogue worth
// K0 reduction steps) / (measured efficiency of that tile's main loop for this operand form), plus the
// slab traffic of a K split.  The efficiencies were fitted on MI355X to the network's GEMM shapes
// (tools/gemm_sweep.py; every shape of the 512x384 batch-32 step gets its measured-fastest tile) but the
// model itself is shape-agnostic.  Forms: 0 = K/OUT (forward), 1 = K/K (dgrad), 2 = OUT/OUT (wgrad).
static int pick_tile(int form, int M, int N, int K, int split_k, bool have_ws, long ws_floats, int nbatch = 1) {
  if (M <= 32) return 4;
  const int cand[7] = {1, 2, 3, 5, 6, 7, 8};
  static const double eff[3][7] = {{1.00, 0.95, 0.93, 0.93, 0.97, 1.00, 0.93},
                                   {0.95, 0.93, 0.97, 0.97, 1.00, 0.90, 0.99},
                                   {0.80, 0.85, 0.95, 1.00, 0.90, 0.85, 0.80}};
  int best = 1;
  double best_cost = 1e300;
  // (tile 9, 32x64, is never chosen here: it is selected per shape from measurements -- spnet_amd/gemm_tiles.json)
  for (int c = 0; c < 7; ++c) {
    int bm, bn;
    tile_dims(cand[c], &bm, &bn);
    const long tiles = (long)spnet_cdiv(M, bm) * spnet_cdiv(N, bn);
    int ns = split_k > 0 ? split_k : auto_split(tiles, M, N, K, have_ws, ws_floats, form);
    const int kc = spnet_cdiv(spnet_cdiv(K, ns), SP_BK) * SP_BK;
    ns = spnet_cdiv(K, kc);
    const double rounds = (double)((tiles * ns * nbatch + 255) / 256) / nbatch;
    double cost = rounds * bm * bn * ((double)kc + 128.0) / eff[form][c];
    if (ns > 1) cost += 2.0 * ns * (double)M * N / 256.0;
    if (cost < best_cost) { best_cost = cost; best = cand[c]; }
  }
  return best;
}

// out_j[i] = sum over the kslices slabs of problem j, slice order (deterministic); out_j = C0 + offsets[3j + 2].
__global__ __launch_bounds__(256) void reduce_slabs_batched_kernel(const float* __restrict__ ws, int kslices, long mn,
                                                                   float* __restrict__ C0,
                                                                   const long long* __restrict__ offsets) {
  const int j = blockIdx.y;
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= mn) return;
  const float* p = ws + (long)j * kslices * mn + i;
  float4 s = *reinterpret_cast<const float4*>(p);
  for (int k = 1; k < kslices; ++k) {
    const float4 t = *reinterpret_cast<const float4*>(p + (long)k * mn);
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  *reinterpret_cast<float4*>(C0 + offsets[3 * j + 2] + i) = s;
}

static int gemm_impl(const float* A, int a_major, int lda, const float* B, int b_major, int ldb, float* C,
                     int ldc, int M, int N, int K, int split_k, float* workspace, long ws_floats,
                     const float* bias, int tile, float* colstats, int* stat_rows, void* stream,
                     const long long* batch = nullptr, int nbatch = 0, int xf = 0, const float* X2 = nullptr,
                     const float* coef = nullptr, int cld = 0, float* dy_out = nullptr, int accumulate = 0,
                     const ConvGeom* cg = nullptr) {
  hipStream_t st = (hipStream_t)stream;
  if (cg && (batch || xf || accumulate || a_major != SP_K_MAJOR || b_major != SP_OUT_MAJOR || split_k != 1))
    return (int)hipErrorInvalidValue;     // gathered A: one whole forward-form problem
  if (accumulate) {                  // C += A B in the epilogue: whole dot products only (no slabs), one problem
    if (batch || colstats || xf) return (int)hipErrorInvalidValue;
    split_k = 1;
  }
  if (batch) {                       // nbatch whole problems side by side; a K split only when the caller asks for one
    if (nbatch < 1 || bias || colstats) return (int)hipErrorInvalidValue;
    if (split_k <= 1 || !workspace) {
      split_k = 1;
      workspace = nullptr;
      ws_floats = 0;
    } else if (ldc != N) {
      return (int)hipErrorInvalidValue;          // the slab reduce writes dense M x N outputs
    }
  }
  if (colstats) split_k = 1;   // statistics are taken from complete dot products
  if (M <= 0 || N <= 0 || K <= 0) return (int)hipErrorInvalidValue;
  if ((lda & 3) || (ldb & 3) || (N & 3) || (ldc & 3)) return (int)hipErrorInvalidValue;
  if ((a_major == SP_K_MAJOR || b_major == SP_K_MAJOR) && (K & 3)) return (int)hipErrorInvalidValue;
  if (a_major == SP_OUT_MAJOR && (M & 3)) return (int)hipErrorInvalidValue;
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) return (int)hipErrorInvalidValue;
  if (xf) {   // blended A operand: second tensor + [a|b|c] coefficients, zero-padded to whole K tiles; no K split
    if (xf != 1 || batch || !X2 || !coef || (((uintptr_t)X2 | (uintptr_t)coef | (uintptr_t)dy_out) & 15) || (cld & 3))
      return (int)hipErrorInvalidValue;
    if (cld < spnet_cdiv(K, SP_BK) * SP_BK) return (int)hipErrorInvalidValue;
    split_k = 1;
  }
  const bool auto_tile = (tile <= 0 || tile > SP_NTILES);
  const int form = (a_major == SP_OUT_MAJOR) ? 2 : (b_major == SP_K_MAJOR ? 1 : 0);
  if (tile <= 0 || tile > SP_NTILES)
    tile = pick_tile(form, M, N, K, split_k, workspace != nullptr, ws_floats, batch ? nbatch : 1);
  if (xf && auto_tile && tile == 1) tile = 8;     // the blended 128x128 kernel does not fit the register file
  int bm, bn;
  tile_dims(tile, &bm, &bn);
  const long tiles = (long)spnet_cdiv(M, bm) * spnet_cdiv(N, bn);
  const int BK = SP_BK;
  int nsplit = split_k;
  if (nsplit <= 0) nsplit = auto_split(tiles, M, N, K, workspace != nullptr, ws_floats, form);
  const int bkt = BK;
  int k_chunk = ((K + nsplit - 1) / nsplit + bkt - 1) / bkt * bkt;
  nsplit = (K + k_chunk - 1) / k_chunk;
  float* out = C;
  int out_ld = ldc;
  long slab = 0;
  const float* kbias = bias;
  if (nsplit > 1) {
    if (!workspace || (long)nsplit * (batch ? nbatch : 1) * M * N > ws_floats) return (int)hipErrorInvalidValue;
    if (((uintptr_t)workspace) & 15) return (int)hipErrorInvalidValue;
    out = workspace;
    out_ld = N;
    slab = (long)M * N;
    kbias = nullptr;
  }
  if (stat_rows) *stat_rows = spnet_cdiv(M, bm);
  const int kslices = batch ? nsplit : 1;
  float* C0 = C;
  if (batch) nsplit = nbatch * kslices;      // the kernel's slice index selects (problem, K slice)
  int rc;
  switch (tile) {
    case 1: rc = launch_tile<128, 128, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 2: rc = launch_tile<128, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 3: rc = launch_tile<64, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 5: rc = launch_tile<96, 96, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 6: rc = launch_tile<96, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 7: rc = launch_tile<64, 128, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 8: rc = launch_tile<128, 96, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 9: rc = launch_tile<32, 64, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    case 10: rc = launch_tile<32, 32, 2, 2>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
    default: rc = launch_tile<32, 128, 1, 4>(A, a_major, lda, B, b_major, ldb, out, out_ld, M, N, K, nsplit, k_chunk, slab, kbias, colstats, batch, st, xf, X2, coef, cld, dy_out, kslices, accumulate, cg); break;
  }
  if (rc) return rc;
  if (nsplit > 1 && !batch) {
    const long total4 = (long)M * N / 4;
    launch_reduce_slabs(workspace, nsplit, M, N, C, ldc, bias, st);
  }
  if (batch && kslices > 1) {
    const long mn = (long)M * N;
    hipLaunchKernelGGL(reduce_slabs_batched_kernel, dim3((unsigned)((mn / 4 + 255) / 256), nbatch), dim3(256), 0, st,
                       workspace, kslices, mn, C0, batch);
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

// C += A B (no K split: the sum is formed in the epilogue, read-modify-write of C by the workgroup that owns the tile):
// a data gradient added straight onto the gradient another consumer of the same tensor has already left in C, instead
// of a GEMM into a scratch tensor + an accumulation pass (the branch convolutions of an inception block all read the
// block input).
extern "C" int spnet_gemm_f32_accumulate(const float* A, int a_major, int lda, const float* B, int b_major, int ldb,
                                         float* C, int ldc, int M, int N, int K, int tile, void* stream) {
  return gemm_impl(A, a_major, lda, B, b_major, ldb, C, ldc, M, N, K, 1, nullptr, 0, nullptr, tile, nullptr, nullptr,
                   stream, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 1);
}

// nbatch independent problems of one shape in ONE launch (no K split).  A0/B0/C0: operands of problem 0;
// offsets (device memory): {A_b - A0, B_b - B0, C_b - C0} in floats for b = 0..nbatch-1 (multiples of 4).
// Used for the weight gradients of the eight middle-flow blocks, which individually are too small to fill
// the chip without a K split.
extern "C" int spnet_gemm_f32_batched(const float* A0, const float* B0, float* C0, const long long* offsets,
                                      int nbatch, int a_major, int lda, int b_major, int ldb, int ldc, int M,
                                      int N, int K, int tile, void* stream) {
  if (!offsets || !A0 || !B0 || !C0) return (int)hipErrorInvalidValue;
  return gemm_impl(A0, a_major, lda, B0, b_major, ldb, C0, ldc, M, N, K, 1, nullptr, 0, nullptr, tile, nullptr, nullptr,
                   stream, offsets, nbatch);
}

// The same with every problem's K axis cut into `ksplit` slices (fp32 slabs in `workspace`, nbatch * ksplit * M * N
// floats, summed per problem in slice order by one more launch): many same-shaped weight gradients whose outputs are
// too small to fill the chip even side by side -- the repeated blocks of Inception-ResNet-v2 at batch 16 (10 x block35,
// 20 x block17, 10 x block8: dW of 288x32 ... 2080x192 over K = 384 ... 9,744 pixels).  ldc == N.
// spnet_gemm_batched_ksplit: the slice count this library would choose (about three workgroups per CU, slices at
// least four K tiles deep) and the tile it goes with; 1 = no split.
extern "C" long spnet_gemm_batched_ksplit(int M, int N, int K, int nbatch, int* tile_out) {
  const int tile = N <= 64 ? 6 : 5;                  // 96x64 | 96x96
  int bm, bn;
  tile_dims(tile, &bm, &bn);
  const long tiles = (long)spnet_cdiv(M, bm) * spnet_cdiv(N, bn) * (nbatch < 1 ? 1 : nbatch);
  long want = (768 + tiles - 1) / tiles;
  long maxk = K / (SP_BK * 4);
  if (maxk < 1) maxk = 1;
  if (want > maxk) want = maxk;
  if (want > 64) want = 64;
  if (tile_out) *tile_out = tile;
  return want < 1 ? 1 : want;
}
extern "C" int spnet_gemm_f32_batched_splitk(const float* A0, const float* B0, float* C0, const long long* offsets,
                                             int nbatch, int a_major, int lda, int b_major, int ldb, int ldc, int M,
                                             int N, int K, int tile, int ksplit, float* workspace, long ws_floats,
                                             void* stream) {
  if (!offsets || !A0 || !B0 || !C0) return (int)hipErrorInvalidValue;
  if (ksplit > 1 && (!workspace || (((uintptr_t)workspace) & 15))) return (int)hipErrorInvalidValue;
  return gemm_impl(A0, a_major, lda, B0, b_major, ldb, C0, ldc, M, N, K, ksplit > 1 ? ksplit : 1, workspace, ws_floats,
                   nullptr, tile, nullptr, nullptr, stream, offsets, nbatch);
}

// dX[M,N] = dY[M,K] B[K,N] where dY is the BatchNorm-backward output dy = a[k]*g + b[k]*yp + c[k], blended from the
// incoming gradient g and the saved pre-normalisation tensor yp while the A tile is staged (forward operand form:
// B = W^T kept by the engine).  coef = [a | b | c], cld floats each (zero beyond channel K-1, cld a multiple of 32
// covering K).  dy_out (or NULL) receives dy itself, once, for the weight-gradient GEMM of the layer.
extern "C" int spnet_gemm_f32_bnblend(const float* g, const float* yp, const float* coef, int cld, int lda,
                                      const float* B, int ldb, float* C, int ldc, int M, int N, int K, int tile,
                                      float* dy_out, void* stream) {
  return gemm_impl(g, SP_K_MAJOR, lda, B, SP_OUT_MAJOR, ldb, C, ldc, M, N, K, 1, nullptr, 0, nullptr, tile, nullptr,
                   nullptr, stream, nullptr, 0, 1, yp, coef, cld, dy_out);
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

// The forward convolution of an NHWC tensor as an implicit GEMM on the same kernel: y[B*OH*OW][Cout] = patches(x) * Wk,
// Wk = the HWIO kernel read as [KH*KW*C][Cout], the patch matrix never written (AG = 1 above).  x pixels are ldx floats
// apart (ldx >= C: x may be a column block of a wider tensor); C % 32 == 0; stride 1 or 2; `same` = TF SAME padding, else
// VALID.  bias (or NULL) is added in the epilogue; colstats / stat_rows (or NULL / NULL) as spnet_gemm_f32_colstats.
// Bit-identical to spnet_patches + spnet_gemm_f32(_colstats) with the same tile.
extern "C" int spnet_conv_gemm_f32(const float* x, long ldx, const float* Wk, float* y, int ldy, int B, int H, int W,
                                   int C, int Cout, int KH, int KW, int stride, int same, const float* bias, int tile,
                                   float* colstats, int* stat_rows, void* stream) {
  if (!x || !Wk || !y || B < 1 || H < 1 || W < 1 || KH < 1 || KW < 1 || (stride != 1 && stride != 2)) return (int)hipErrorInvalidValue;
  if (C < SP_BK || (C % SP_BK) || (ldx & 3) || ldx < C || (!colstats) != (!stat_rows)) return (int)hipErrorInvalidValue;
  if ((long)B * H * W * ldx >= (1L << 31)) return (int)hipErrorInvalidValue;       // 32-bit element offsets in the gather
  // (the gathered fetch carries its tap as running state from k = 0 on: gemm_impl refuses a K split for it; taps beyond a
  // 16 x 16 window, an output row stride below Cout or a K of 2^31 are not convolutions this entry point serves)
  if (KH > 16 || KW > 16 || Cout < 4 || (Cout & 3) || ldy < Cout || (ldy & 3) || (long)KH * KW * C >= (1L << 31))
    return (int)hipErrorInvalidValue;
  ConvGeom cg;
  cg.H = H; cg.W = W; cg.C = C; cg.KW = KW; cg.stride = stride;
  if (same) {
    cg.OH = (H + stride - 1) / stride; cg.OW = (W + stride - 1) / stride;
    int th = (cg.OH - 1) * stride + KH - H, tw = (cg.OW - 1) * stride + KW - W;
    cg.pt = (th < 0 ? 0 : th) / 2; cg.pl = (tw < 0 ? 0 : tw) / 2;
  } else {
    cg.OH = (H - KH) / stride + 1; cg.OW = (W - KW) / stride + 1;
    cg.pt = cg.pl = 0;
  }
  if (cg.OH < 1 || cg.OW < 1) return (int)hipErrorInvalidValue;
  return gemm_impl(x, SP_K_MAJOR, (int)ldx, Wk, SP_OUT_MAJOR, Cout, y, ldy, B * cg.OH * cg.OW, Cout, KH * KW * C, 1, nullptr, 0,
                   bias, tile, colstats, stat_rows, stream, nullptr, 0, 0, nullptr, nullptr, 0, nullptr, 0, &cg);
}

// ------------------------------------------------------------------------------------------------
// Stride-2 row gather / scatter of the 1x1 residual convs (TF SAME on a 1x1/s2 conv samples the even pixels).
// ------------------------------------------------------------------------------------------------
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

// Transposes of many small matrices in one launch: job j writes dst_j[c][r] = src_j[r][c] for an R_j x C_j
// row-major matrix.  jobs (device memory) = njobs x {src pointer, dst pointer, R, C} as four 64-bit words.
// 32x32 tiles through a padded LDS tile: both the global read and the global write are 128-byte row segments.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const long long* __restrict__ jobs) {
  __shared__ float t[32][33];
  const long long* jb = jobs + 4 * blockIdx.z;
  const float* __restrict__ src = reinterpret_cast<const float*>(jb[0]);
  float* __restrict__ dst = reinterpret_cast<float*>(jb[1]);
  const int R = (int)jb[2], C = (int)jb[3];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  if (r0 >= R || c0 >= C) return;                    // whole workgroup leaves together
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    t[ty + 8 * i][tx] = (r < R && c < C) ? src[(long)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < C && r < R) dst[(long)c * R + r] = t[tx][ty + 8 * i];
  }
}

extern "C" int spnet_transpose_batched(const void* jobs, int njobs, int max_rows, int max_cols, void* stream) {
  if (!jobs || njobs < 1 || max_rows < 1 || max_cols < 1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((max_cols + 31) / 32, (max_rows + 31) / 32, njobs), dim3(256), 0,
                     (hipStream_t)stream, reinterpret_cast<const long long*>(jobs));
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
