// Implicit-GEMM passes of block1_conv2 (3x3, stride 1, VALID, 32 -> 64 channels at half resolution; keras
// Xception, call site spnet/models.py:357-359).  The im2col matrix of this layer is 9x its input (428 MB at
// batch 32, 512x384), so the passes that would have to WRITE or re-read it gather their operand tiles
// straight from the NHWC tensors instead:
//
//   dX   dx[b,ih,iw,ci] = sum_{kh,kw,co} dy[b,ih-kh,iw-kw,co] * w[kh,kw,ci,co]
//        as a GEMM with M = B*H*W input pixels, N = CIN, K = 9*COUT (tap-major): the A tile of K tile t
//        (tap = t / (COUT/32)) is the dy tensor shifted by that tap -- one loop-invariant offset per
//        staged row plus a wave-uniform shift per tile, with a 9-bit validity mask per row for the border
//        (rows whose tap falls outside dy read as zero).  No dcol matrix, no col2im pass.
//   Y    y[m, co] = sum_{tap,ci} x[in(m) + tap shift, ci] * w[tap, ci, co]: M = B*OH*OW output pixels, N = COUT,
//        K = 9*CIN; K tile t IS tap t (CIN = 32): the x tensor shifted by (kh*W + kw) pixels.
//   dW   dw[(tap,ci), co] = sum_p x[in(p) + tap shift, ci] * dy[p, co]: M = 9*CIN, N = COUT, K = pixels, split
//        over workgroups (fp32 slabs, summed in slice order by reduce_slabs); a staged float4 keeps its
//        (tap, ci) for the whole loop and walks the pixels incrementally (no divisions in the loop).
//
// Same MFMA tile machinery as gemm.hip (gemm_tile.h); arithmetic is the same k-ordered f32 fma chain.
#include "gemm_tile.h"

// bit i = row slot i has tap `tap` inside the image
template <int NV>
__device__ __forceinline__ unsigned tap_ok(const unsigned (&mask)[NV], int tap) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) m |= ((mask[i] >> tap) & 1u) << i;
  return m;
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_dgrad_kernel(const float* __restrict__ dy,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ dx, int Bn, int H, int W,
                                                               int OH, int OW) {
  constexpr int BM = 128, BN = CIN, BK = 32, WM = 4, WN = 1;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int KPT = COUT / BK;                    // K tiles per tap
  constexpr int KT = 9 * KPT;
  static_assert(CIN % 16 == 0 && COUT % BK == 0 && KT >= 2, "shape");
  typedef TileStage<BM, BK, SP_K_MAJOR, TM> SA;
  typedef TileStage<BN, BK, SP_K_MAJOR, TN, 1> SB;
  static_assert(SB::TOTAL % 256 == 0 && SA::TOTAL % 256 == 0, "whole float4 slots per thread");
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  constexpr int NCH = BK / 16;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const long M = (long)Bn * H * W;
  const int tm = blockIdx.x;
  const int m0 = tm * BM;

  // loop-invariant part of the A gather: this thread's rows (input pixels) and their tap validity
  int abase[SA::NV];
  unsigned amask[SA::NV];
  const int kq = tid % (BK / 4);
#pragma unroll
  for (int i = 0; i < SA::NV; ++i) {
    const int r = (tid + i * 256) / (BK / 4);
    long m = (long)m0 + r;
    if (m > M - 1) m = M - 1;                       // clamped rows only feed output rows >= M (never stored)
    const int iw = (int)(m % W);
    const long q = m / W;
    const int ih = (int)(q % H);
    const int b = (int)(q / H);
    abase[i] = (int)((((long)b * OH + ih) * OW + iw) * COUT) + kq * 4;
    unsigned mk = 0;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int oh = ih - tap / 3, ow = iw - tap % 3;
      if (oh >= 0 && oh < OH && ow >= 0 && ow < OW) mk |= 1u << tap;
    }
    amask[i] = mk;
  }
  const int boff[1] = {(tid / (BK / 4)) * COUT + kq * 4};     // B: one float4 per thread, row = input channel
  static_assert(SB::NV == 1, "one B slot per thread");

  // Register stages as four plain objects and the pipeline steps as macros: with the stages in an array
  // reached through lambda captures the compiler keeps them in scratch memory.
  SA sa0, sa1;
  SB sb0, sb1;
#define DG_FETCH(SA_, SB_, T_)                                                                              \
  do {                                                                                                      \
    const int tap_ = (T_) / KPT, half_ = (T_) % KPT;            /* wave-uniform */                          \
    const int shift_ = -((tap_ / 3) * OW + tap_ % 3) * COUT + half_ * BK;                                   \
    SA_.load_gather(dy, abase, shift_, tap_ok(amask, tap_));                                                \
    SB_.load_gather(w + (long)tap_ * (CIN * COUT) + half_ * BK, boff, 0, ~0u);                              \
  } while (0)
  // tile T in LDS buffer CUR_, tile T+1 in stage (SA_ST, SB_ST), tile T+2 fetched into (SA_LD, SB_LD)
#define DG_STEP(CUR_, SA_LD, SB_LD, SA_ST, SB_ST, T_, FETCH_, STORE_)                                       \
  do {                                                                                                      \
    if (FETCH_) DG_FETCH(SA_LD, SB_LD, (T_) + 2);                                                           \
    mma_tile_store<SA, SB, TM, TN, NCH, STORE_>(smem + (CUR_) * STAGE, wm * (TM * 16), wn * (TN * 16), lane, acc, SA_ST, \
                                                SB_ST, smem + (1 - (CUR_)) * STAGE, tid);                  \
    __syncthreads();                                                                                        \
  } while (0)

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  DG_FETCH(sa0, sb0, 0);
  DG_FETCH(sa1, sb1, 1);
  sa0.store(smem, tid);
  sb0.store(smem + SA::SIZE, tid);
  __syncthreads();

  static_assert(KT % 2 == 0 && KT >= 4, "step schedule");
  for (int t = 0; t < KT - 2; t += 2) {
    DG_STEP(0, sa0, sb0, sa1, sb1, t, true, true);
    DG_STEP(1, sa1, sb1, sa0, sb0, t + 1, true, true);
  }
  DG_STEP(0, sa0, sb0, sa1, sb1, KT - 2, false, true);          // nothing left to fetch
  DG_STEP(1, sa1, sb1, sa0, sb0, KT - 1, false, false);
#undef DG_STEP
#undef DG_FETCH

  gemm_epilogue<SA, SB, BM, BN, WM, WN, TM, TN>(acc, smem, dx, CIN, (int)M, CIN, m0, 0, tm, 0, 0, nullptr, nullptr, tid,
                                                lane, wm, wn);
}

template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w,
                                                             float* __restrict__ y, int Bn, int H, int W,
                                                             int OH, int OW) {
  constexpr int BM = 128, BN = COUT, BK = 32, WM = 2, WN = 2;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int KT = 9;
  static_assert(CIN == BK, "one K tile per tap");
  typedef TileStage<BM, BK, SP_K_MAJOR, TM> SA;
  typedef TileStage<BN, BK, SP_OUT_MAJOR, TN> SB;
  static_assert(SB::TOTAL % 256 == 0 && SA::TOTAL % 256 == 0, "whole float4 slots per thread");
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  constexpr int NCH = BK / 16;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const long M = (long)Bn * OH * OW;
  const int tm = blockIdx.x;
  const int m0 = tm * BM;

  int abase[SA::NV];
  const int kq = tid % (BK / 4);
#pragma unroll
  for (int i = 0; i < SA::NV; ++i) {
    const int r = (tid + i * 256) / (BK / 4);
    long m = (long)m0 + r;
    if (m > M - 1) m = M - 1;
    const int ow = (int)(m % OW);
    const long q = m / OW;
    const int oh = (int)(q % OH);
    const int b = (int)(q / OH);
    abase[i] = (int)((((long)b * H + oh) * W + ow) * CIN) + kq * 4;
  }
  int boff[SB::NV];
#pragma unroll
  for (int i = 0; i < SB::NV; ++i) {
    const int f = tid + i * 256;
    boff[i] = (f / (BN / 4)) * COUT + (f % (BN / 4)) * 4;
  }

  SA sa0, sa1;
  SB sb0, sb1;
#define FW_FETCH(SA_, SB_, T_)                                                                              \
  do {                                                                                                      \
    const float* xs_ = x + (((T_) / 3) * W + (T_) % 3) * CIN;   /* tap T: wave-uniform shift */             \
    const float* ws_ = w + (long)(T_) * (CIN * COUT);                                                       \
    SA_.load_gather(xs_, abase, 0, ~0u);                                                                    \
    SB_.load_gather(ws_, boff, 0, ~0u);                                                                     \
  } while (0)
#define FW_STEP(CUR_, SA_LD, SB_LD, SA_ST, SB_ST, T_, FETCH_, STORE_)                                       \
  do {                                                                                                      \
    if (FETCH_) FW_FETCH(SA_LD, SB_LD, (T_) + 2);                                                           \
    mma_tile_store<SA, SB, TM, TN, NCH, STORE_>(smem + (CUR_) * STAGE, wm * (TM * 16), wn * (TN * 16), lane, acc, SA_ST, \
                                                SB_ST, smem + (1 - (CUR_)) * STAGE, tid);                  \
    __syncthreads();                                                                                        \
  } while (0)

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  FW_FETCH(sa0, sb0, 0);
  FW_FETCH(sa1, sb1, 1);
  sa0.store(smem, tid);
  sb0.store(smem + SA::SIZE, tid);
  __syncthreads();

  // KT = 9: steps 0..6 fetch tile t+2, step 7 only stores tile 8, step 8 is the last
  static_assert(KT == 9, "step schedule below");
  for (int t = 0; t < 6; t += 2) {
    FW_STEP(0, sa0, sb0, sa1, sb1, t, true, true);
    FW_STEP(1, sa1, sb1, sa0, sb0, t + 1, true, true);
  }
  FW_STEP(0, sa0, sb0, sa1, sb1, 6, true, true);
  FW_STEP(1, sa1, sb1, sa0, sb0, 7, false, true);
  FW_STEP(0, sa0, sb0, sa1, sb1, 8, false, false);
#undef FW_STEP
#undef FW_FETCH

  gemm_epilogue<SA, SB, BM, BN, WM, WN, TM, TN>(acc, smem, y, COUT, (int)M, COUT, m0, 0, tm, 0, 0, nullptr, nullptr, tid,
                                                lane, wm, wn);
}

// dW walkers.  A: slot i reads x at (window origin of pixel a_p[i]) + its tap/channel offset, then moves BK
// pixels ahead in row-major output order; B: slot i reads row b_p[i] of dy.  Slots past `pend` are masked.
template <int NV, int BK, int CIN>
__device__ __forceinline__ void wg_walk(long (&a_p)[NV], int (&a_in)[NV], int (&a_ow)[NV], int (&a_oh)[NV],
                                        const int (&a_tapoff)[NV], long pend, int H, int W, int OH, int OW,
                                        int (&offs)[NV], unsigned& ok) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ok |= (a_p[i] < pend ? 1u : 0u) << i;
    offs[i] = a_in[i] + a_tapoff[i];
    a_p[i] += BK;
    int ow = a_ow[i] + BK, oh = a_oh[i], in = a_in[i] + BK * CIN;
    while (ow >= OW) {                                // row wraps (one at most when OW >= BK)
      ow -= OW;
      in += (W - OW) * CIN;
      if (++oh >= OH) { oh = 0; in += (H - OH) * W * CIN; }
    }
    a_ow[i] = ow; a_oh[i] = oh; a_in[i] = in;
  }
}
template <int NV, int BK, int COUT>
__device__ __forceinline__ void wg_rows(long (&b_p)[NV], const int (&b_off)[NV], long pend, int (&offs)[NV],
                                        unsigned& ok) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    ok |= (b_p[i] < pend ? 1u : 0u) << i;           // A is zero past pend; B only must stay in bounds
    offs[i] = (int)(b_p[i] * COUT) + b_off[i];
    b_p[i] += BK;
  }
}

// One workgroup = one 96-row block of dW (3 taps x 32 input channels) x all COUT columns x one slice of the
// pixels; slab z of the workspace receives its partial sum.
template <int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ dy,
                                                               float* __restrict__ slabs, int Bn, int H,
                                                               int W, int OH, int OW, int p_chunk) {
  constexpr int BM = 96, BN = COUT, BK = 32, WM = 2, WN = 2;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  static_assert(CIN == 32 && (9 * CIN) % BM == 0, "row blocks are whole taps");
  typedef TileStage<BM, BK, SP_OUT_MAJOR, TM> SA;
  typedef TileStage<BN, BK, SP_OUT_MAJOR, TN> SB;
  static_assert(SB::TOTAL % 256 == 0 && SA::TOTAL % 256 == 0, "whole float4 slots per thread");
  constexpr int STAGE = SA::SIZE + SB::SIZE;
  constexpr int NCH = BK / 16;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const long P = (long)Bn * OH * OW;                 // reduction length: output pixels
  const int tm = blockIdx.x % (9 * CIN / BM);
  const int z = blockIdx.x / (9 * CIN / BM);
  const int m0 = tm * BM;
  const long pbeg = (long)z * p_chunk;
  const long pend = pbeg + p_chunk < P ? pbeg + p_chunk : P;
  const int nt = (int)((pend - pbeg + BK - 1) / BK);

  // A slot i: pixel row k_i of the tile (walks +BK per tile), columns (tap, ci..ci+3) fixed
  int a_in[SA::NV], a_ow[SA::NV], a_oh[SA::NV], a_tapoff[SA::NV];
  long a_p[SA::NV];
#pragma unroll
  for (int i = 0; i < SA::NV; ++i) {
    const int f = tid + i * 256;
    const int k = f / (BM / 4), c = m0 + (f % (BM / 4)) * 4;
    const int tap = c / CIN, ci = c % CIN;
    a_tapoff[i] = ((tap / 3) * W + tap % 3) * CIN + ci;
    const long p = pbeg + k;
    a_p[i] = p;
    const long pc = p < P ? p : P - 1;
    const int ow = (int)(pc % OW);
    const long q = pc / OW;
    const int oh = (int)(q % OH);
    const int b = (int)(q / OH);
    a_ow[i] = ow;
    a_oh[i] = oh;
    a_in[i] = (int)((((long)b * H + oh) * W + ow) * CIN);       // element offset of the window's first pixel
  }
  int b_off[SB::NV];
  long b_p[SB::NV];
#pragma unroll
  for (int i = 0; i < SB::NV; ++i) {
    const int f = tid + i * 256;
    b_p[i] = pbeg + f / (BN / 4);
    b_off[i] = (f % (BN / 4)) * 4;
  }

  SA sa0, sa1;
  SB sb0, sb1;
  // fetch the NEXT tile in walking order (tiles are fetched strictly in sequence), then advance the walkers
#define WG_FETCH(SA_, SB_)                                                                                  \
  do {                                                                                                      \
    unsigned oka_ = 0, okb_ = 0;                                                                            \
    int offa_[SA::NV], offb_[SB::NV];                                                                       \
    wg_walk<SA::NV, BK, CIN>(a_p, a_in, a_ow, a_oh, a_tapoff, pend, H, W, OH, OW, offa_, oka_);             \
    wg_rows<SB::NV, BK, COUT>(b_p, b_off, pend, offb_, okb_);                                               \
    SA_.load_gather(x, offa_, 0, oka_);                                                                     \
    SB_.load_gather(dy, offb_, 0, okb_);                                                                    \
  } while (0)
  // Every step fetches and stores unconditionally: tiles past the slice are zeros (two extra fetches of
  // clamped addresses per workgroup buy a branch-free loop body).
#define WG_STEP(CUR_, SA_LD, SB_LD, SA_ST, SB_ST)                                                           \
  do {                                                                                                      \
    WG_FETCH(SA_LD, SB_LD);                                                                                 \
    mma_tile_store<SA, SB, TM, TN, NCH, true>(smem + (CUR_) * STAGE, wm * (TM * 16), wn * (TN * 16), lane, acc, SA_ST, \
                                              SB_ST, smem + (1 - (CUR_)) * STAGE, tid);                    \
    __syncthreads();                                                                                        \
  } while (0)

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  WG_FETCH(sa0, sb0);                                 // tile 0 (all zeros if the slice is empty)
  WG_FETCH(sa1, sb1);                                 // tile 1 (zeros past pend)
  sa0.store(smem, tid);
  sb0.store(smem + SA::SIZE, tid);
  __syncthreads();

  for (int t = 0; t < nt; t += 2) {
    WG_STEP(0, sa0, sb0, sa1, sb1);
    if (t + 1 < nt) WG_STEP(1, sa1, sb1, sa0, sb0);
  }
#undef WG_STEP
#undef WG_FETCH

  gemm_epilogue<SA, SB, BM, BN, WM, WN, TM, TN>(acc, smem, slabs, COUT, 9 * CIN, COUT, m0, 0, tm, z,
                                                (long)9 * CIN * COUT, nullptr, nullptr, tid, lane, wm, wn);
}

// dy [B][H-2][W-2][cout], w HWIO [3][3][cin][cout], dx [B][H][W][cin].  Supported: (cin, cout) = (32, 64).
extern "C" int spnet_conv3x3_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int cin,
                                   int cout, void* stream) {
  if (cin != 32 || cout != 64 || H < 3 || W < 3 || B < 1) return (int)hipErrorInvalidValue;
  if (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) return (int)hipErrorInvalidValue;
  const long M = (long)B * H * W;
  if (M * 64 >= (1L << 31)) return (int)hipErrorInvalidValue;     // 32-bit element offsets
  hipLaunchKernelGGL((conv3x3_dgrad_kernel<32, 64>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0,
                     (hipStream_t)stream, dy, w, dx, B, H, W, H - 2, W - 2);
  SPNET_RETURN_LAUNCH_STATUS();
}

// x [B][H][W][cin], w HWIO -> y [B][H-2][W-2][cout].  Supported: (cin, cout) = (32, 64).
extern "C" int spnet_conv3x3_fwd(const float* x, const float* w, float* y, int B, int H, int W, int cin,
                                 int cout, void* stream) {
  if (cin != 32 || cout != 64 || H < 3 || W < 3 || B < 1) return (int)hipErrorInvalidValue;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return (int)hipErrorInvalidValue;
  const long M = (long)B * (H - 2) * (W - 2);
  if ((long)B * H * W * 32 >= (1L << 31)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL((conv3x3_fwd_kernel<32, 64>), dim3((unsigned)((M + 127) / 128)), dim3(256), 0,
                     (hipStream_t)stream, x, w, y, B, H, W, H - 2, W - 2);
  SPNET_RETURN_LAUNCH_STATUS();
}

extern "C" int spnet_reduce_slabs(const float* ws, int nslab, int M, int N, float* out, int ldc, void* stream);   // gemm.hip

// Number of workspace floats spnet_conv3x3_wgrad needs for this geometry.
extern "C" long spnet_conv3x3_wgrad_ws(int B, int H, int W, int cin, int cout) {
  const long P = (long)B * (H - 2) * (W - 2);
  long ns = (P + 32 * 64 - 1) / (32 * 64);            // >= 64 K tiles per slice ...
  if (ns > 170) ns = 170;                             // ... and ~512 workgroups (3 row blocks each)
  if (ns < 1) ns = 1;
  return ns * 9 * cin * cout;
}

// x [B][H][W][cin], dy [B][H-2][W-2][cout] -> dw HWIO [3][3][cin][cout].  Supported: (cin, cout) = (32, 64).
extern "C" int spnet_conv3x3_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int cin,
                                   int cout, float* workspace, long ws_floats, void* stream) {
  if (cin != 32 || cout != 64 || H < 3 || W < 3 || B < 1) return (int)hipErrorInvalidValue;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) return (int)hipErrorInvalidValue;
  if ((long)B * H * W * 32 >= (1L << 31)) return (int)hipErrorInvalidValue;
  const long P = (long)B * (H - 2) * (W - 2);
  const long need = spnet_conv3x3_wgrad_ws(B, H, W, cin, cout);
  if (!workspace || ws_floats < need) return (int)hipErrorInvalidValue;
  int ns = (int)(need / (9 * cin * cout));
  const int p_chunk = (int)(((P + ns - 1) / ns + 31) / 32 * 32);
  ns = (int)((P + p_chunk - 1) / p_chunk);
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<32, 64>), dim3(3 * ns), dim3(256), 0, (hipStream_t)stream, x, dy,
                     workspace, B, H, W, H - 2, W - 2, p_chunk);
  return spnet_reduce_slabs(workspace, ns, 9 * cin, cout, dw, cout, stream);
}
