// Building blocks of the fp32 MFMA GEMM kernels (gemm.hip, conv_gemm.hip): the register/LDS staging of
// one operand tile, the wave's wide operand fetches, and the epilogue.
#pragma once
#include "common.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { SP_K_MAJOR = 0, SP_OUT_MAJOR = 1 };

// One operand tile (BR output rows/cols x BK reduction steps) staged global -> registers -> LDS, and the
// wave's MFMA operand fetches from it.  T = number of 16-wide MFMA tiles one wave owns along this operand.
// The LDS image keeps the operand's own major, so no staging pass transposes anything, and every operand
// fetch is a wide ds_read that feeds several MFMAs:
//
//   v_mfma_f32_16x16x4_f32 takes, in lane l, A[row l&15][k = l>>4] and B[k = l>>4][col l&15].  Which k a
//   lane group j = l>>4 supplies is free as long as A and B agree, and so is which output row/column the
//   16 lane positions of a tile stand for.  Within a chunk of 16 reduction steps, MFMA q (0..3) uses
//   k = 4j + q:
//     K_MAJOR   image [r][k], leading dim BK+8: lane (p = l&15, j) reads ONE float4 at [r0 + 16t + p][4j..4j+3]
//               = its operand for the 4 MFMAs of tile t (ds_read_b128, conflict-free for LD = 40).
//     OUT_MAJOR image [k][r]: lane (p, j) reads VW consecutive r at row k = 4j + q = its operand for MFMA q of
//               VW tiles at once; tile t = g*VW + u covers output index g*16*VW + VW*p + u (pos()), so a lane
//               ends up with VW CONSECUTIVE output columns and the epilogue stores them as one vector.
//               VW = 4 / 2 / 1 for T % 4 == 0 / T % 2 == 0 / odd T; the leading dim is chosen so that the
//               two k rows (4 apart) met inside one LDS lane group fall on disjoint banks.
//   PERM (K_MAJOR only): tile row x of a wave's 16T rows is stored at LDS row 16*(x % T) + x / T, so that lane
//               position p of MFMA tile t stands for output index T*p + t: a lane then holds T CONSECUTIVE
//               output columns (dgrad, where B = W^T is K-major) and the epilogue stores them as one vector.
//               The fetch pattern (and its bank behaviour) is unchanged; only the staging store permutes.
//   XF = 1:   the operand is not a tensor in memory but a per-channel affine blend of TWO tensors of one layout,
//               op = a[ch]*P + b[ch]*Q + c[ch]  (ch = the operand's channel index: k for a K-major A, the column for an
//               out-major B).  Both tensors are fetched with the same offsets and blended in registers just before the
//               staging store, so the blended tensor is never written to memory: this is how the BatchNorm backward
//               dy = k1*g + k2*xhat(yp) + k3 reaches the data-gradient and weight-gradient GEMMs (coef = [a|b|c], each
//               cld floats, zero beyond the last channel).
template <int BR, int BK, int MAJ, int T, int PERM = 0, int XF = 0>
struct TileStage {
  static constexpr int TOTAL = BR * BK / 4;            // float4 per tile
  static constexpr int NV = (TOTAL + 255) / 256;       // float4 per thread
  static constexpr int VW = (MAJ == SP_K_MAJOR) ? 1 : ((T % 4 == 0) ? 4 : ((T % 2 == 0) ? 2 : 1));
  static constexpr int VWO = (MAJ == SP_K_MAJOR) ? (PERM ? T : 1) : VW;   // consecutive outputs per lane
  static_assert(!PERM || MAJ == SP_K_MAJOR, "row permutation is for K-major images");
  static constexpr int LD = (MAJ == SP_K_MAJOR) ? (BK + 8) : (VW == 4 ? BR : (VW == 2 ? BR + 8 : BR + 4));
  static constexpr int SIZE = (MAJ == SP_K_MAJOR) ? BR * LD : BK * LD;
  static constexpr int NFR = (MAJ == SP_K_MAJOR) ? T : 4 * (T / VW);   // ds_read instructions of one frags() call
  static_assert(BR % 16 == 0 && BK % 16 == 0, "tile shape");
  float4 v[NV];
  unsigned off[NV];     // element offset of this thread's float4 slots at k0 = 0, rows clamped into range

  // Loop-invariant part of the addresses.  Rows/cols past R are clamped to the last valid one: what they
  // produce only reaches output rows/cols >= R, which the epilogue never stores or counts.
  __device__ __forceinline__ void init(int ld, int r0, int R, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      if (MAJ == SP_OUT_MAJOR) {
        const int k = f / (BR / 4), r4 = f % (BR / 4);
        off[i] = (unsigned)(k * ld + min(r0 + r4 * 4, R - 4));
      } else {
        const int r = f / (BK / 4), kq = f % (BK / 4);
        off[i] = (unsigned)(min(r0 + r, R - 1) * ld + kq * 4);
      }
    }
  }

  // A K tile that lies completely below kend: no predicates, no branches (Pk = operand advanced to k0).
  __device__ __forceinline__ void load_full(const float* __restrict__ Pk) {
    static_assert(TOTAL % 256 == 0, "whole float4 slots per thread");
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(Pk + off[i]);
  }
  // (the second tensor of a blended operand lives in a caller-owned register set w: ONE set serves both register
  // stages, because a stage is blended -- and its w values die -- before the next fetch is issued)
  __device__ __forceinline__ void load_full2(const float* __restrict__ Pk, const float* __restrict__ Qk,
                                             float4 (&w)[NV]) {
    static_assert(TOTAL % 256 == 0, "whole float4 slots per thread");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = *reinterpret_cast<const float4*>(Pk + off[i]);
      w[i] = *reinterpret_cast<const float4*>(Qk + off[i]);
    }
  }
  // v <- a*v + b*w + c with the per-channel coefficients of this thread's slots (K-major operands: every slot of a
  // thread covers the same four reduction steps kq*4..kq*4+3 of the tile, so ONE coefficient triple serves them all).
  __device__ __forceinline__ void xform(const float4 a, const float4 b, const float4 c, const float4 (&w)[NV]) {
    static_assert(MAJ == SP_K_MAJOR && 256 % (BK / 4) == 0, "K-major operands only");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x = fmaf(a.x, v[i].x, fmaf(b.x, w[i].x, c.x));
      v[i].y = fmaf(a.y, v[i].y, fmaf(b.y, w[i].y, c.y));
      v[i].z = fmaf(a.z, v[i].z, fmaf(b.z, w[i].z, c.z));
      v[i].w = fmaf(a.w, v[i].w, fmaf(b.w, w[i].w, c.w));
    }
  }
  // The (blended) tile as it stands in v[] -> out[row*ld + k] (K-major operands): only real rows / reduction steps.
  __device__ __forceinline__ void store_global(float* __restrict__ out, int ld, int r0, int R, int k0, int kend,
                                               int tid) const {
    static_assert(MAJ == SP_K_MAJOR, "K-major operands only");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      const int r = f / (BK / 4), kq = f % (BK / 4);
      // off[i] already is (row clamped into range)*ld + kq*4: a clamped (duplicate) row is simply not written
      if (((TOTAL % 256 == 0) || f < TOTAL) && r0 + r < R && k0 + kq * 4 < kend)
        *reinterpret_cast<float4*>(out + off[i] + k0) = v[i];
    }
  }
  // Gathered tile: slot i comes from base[offs[i] + shift], or is zero (and reads base[0]) when bit i of
  // `okmask` is clear.  (Kept as member functions with #pragma unroll: stage slots indexed from a loop that is
  // not fully unrolled would push the whole stage into scratch memory.)
  __device__ __forceinline__ void load_gather(const float* __restrict__ base, const int (&offs)[NV], int shift,
                                              unsigned okmask) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const bool ok = (okmask >> i) & 1u;
      float4 t = *reinterpret_cast<const float4*>(base + (ok ? offs[i] + shift : 0));
      t.x = ok ? t.x : 0.f; t.y = ok ? t.y : 0.f; t.z = ok ? t.z : 0.f; t.w = ok ? t.w : 0.f;
      v[i] = t;
    }
  }
  static __device__ __forceinline__ long kstep(int ld) { return (MAJ == SP_K_MAJOR) ? (long)BK : (long)BK * ld; }

  // Any K tile: elements past kend / past R read as zero.
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

  // Any K tile of a blended operand (both tensors, same predicates).
  __device__ __forceinline__ void load2(const float* __restrict__ P, const float* __restrict__ Q, int ld, int r0,
                                        int R, int k0, int kend, int tid, float4 (&w)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int f = tid + i * 256;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f), val2 = val;
      if ((TOTAL % 256 == 0) || f < TOTAL) {
        long o = -1;
        if (MAJ == SP_OUT_MAJOR) {
          const int k = f / (BR / 4), r4 = f % (BR / 4);
          const int gk = k0 + k, gr = r0 + r4 * 4;
          if (gk < kend && gr < R) o = (long)gk * ld + gr;
        } else {
          const int r = f / (BK / 4), kq = f % (BK / 4);
          const int gr = r0 + r, gk = k0 + kq * 4;
          if (gr < R && gk < kend) o = (long)gr * ld + gk;
        }
        if (o >= 0) {
          val = *reinterpret_cast<const float4*>(P + o);
          val2 = *reinterpret_cast<const float4*>(Q + o);
        }
      }
      v[i] = val;
      w[i] = val2;
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
          const int x = r % (16 * T);
          const int lr = PERM ? (r - x) + 16 * (x % T) + x / T : r;
          *reinterpret_cast<float4*>(S + lr * LD + kq * 4) = v[i];
        }
      }
    }
  }

  // Operand registers of one 16-step chunk (chunk c of the K tile): f[t][q] feeds MFMA q of tile t.
  // w0 = first row/col of this wave inside the tile.
  static __device__ __forceinline__ void frags(const float* __restrict__ S, int w0, int lane, int c,
                                               float (&f)[T][4]) {
    const int p = lane & 15, j = lane >> 4;
    if (MAJ == SP_K_MAJOR) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float4 x = *reinterpret_cast<const float4*>(S + (w0 + t * 16 + p) * LD + c * 16 + j * 4);
        f[t][0] = x.x; f[t][1] = x.y; f[t][2] = x.z; f[t][3] = x.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* row = S + (c * 16 + j * 4 + q) * LD + w0 + VW * p;
#pragma unroll
        for (int g = 0; g < T / VW; ++g) {
          if (VW == 4) {
            const float4 x = *reinterpret_cast<const float4*>(row + g * 64);
            f[g * VW + 0][q] = x.x; f[g * VW + 1 < T ? g * VW + 1 : 0][q] = x.y;
            f[g * VW + 2 < T ? g * VW + 2 : 0][q] = x.z; f[g * VW + 3 < T ? g * VW + 3 : 0][q] = x.w;
          } else if (VW == 2) {
            const float2 x = *reinterpret_cast<const float2*>(row + g * 32);
            f[g * VW + 0][q] = x.x; f[g * VW + 1 < T ? g * VW + 1 : 0][q] = x.y;
          } else {
            f[g][q] = row[g * 16];
          }
        }
      }
    }
  }

  // output index (inside the wave's T*16 rows/cols) that lane position p of MFMA tile t stands for
  static __device__ __forceinline__ int pos(int t, int p) {
    return (MAJ == SP_K_MAJOR) ? (PERM ? T * p + t : t * 16 + p) : (t / VW) * (16 * VW) + VW * p + (t % VW);
  }
};

// Writes the accumulators (+bias) to C / the split-K slab and, optionally, BatchNorm statistics of the
// output tile: per column sum and sum of squares over this workgroup's BM rows -> colstats[tm][2][N]
// (rows past M hold zeros and contribute nothing).  Fixed reduction order: registers (i, r) -> lanes
// (xor 16, 32) -> waves (wm order) through LDS.
template <class SA, class SB, int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[TM][TN], float* __restrict__ smem,
                                              float* __restrict__ C, int ldc, int M, int N, int m0, int n0,
                                              int tm, int z, long slab_stride, const float* __restrict__ bias,
                                              float* __restrict__ colstats, int tid, int lane, int wm, int wn,
                                              int accumulate = 0) {
  const int p = lane & 15, jq = lane >> 4;
  const int mw = m0 + wm * (TM * 16), nw = n0 + wn * (TN * 16);
  float* Cz = C + (long)z * slab_stride;
  constexpr int VW = SB::VWO;                // consecutive columns held by one lane
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = mw + SA::pos(i, 4 * jq + r);
      if (row < M) {
        float* crow = Cz + (long)row * ldc;
#pragma unroll
        for (int g = 0; g < TN / VW; ++g) {
          const int col = nw + SB::pos(g * VW, p);
          float o[VW];
#pragma unroll
          for (int u = 0; u < VW; ++u) o[u] = acc[i][g * VW + u][r] + (bias && col + u < N ? bias[col + u] : 0.f);
          if (accumulate) {   // C += A B: what is there is read with the vector the store will write
#pragma unroll
            for (int u = 0; u < VW; ++u)
              if (col + u < N) o[u] += crow[col + u];
          }
          // N % 4 == 0 and ldc % 4 == 0: an aligned 2- or 4-vector is inside or outside as a whole
          if (VW == 4) {
            if (col < N) *reinterpret_cast<float4*>(crow + col) = make_float4(o[0], o[VW > 1 ? 1 : 0], o[VW > 2 ? 2 : 0], o[VW > 3 ? 3 : 0]);
          } else if (VW == 2) {
            if (col < N) *reinterpret_cast<float2*>(crow + col) = make_float2(o[0], o[VW > 1 ? 1 : 0]);
          } else if (VW == 3) {   // 12-byte records, 4-byte aligned: one global_store_dwordx3 when wholly inside
            if (col + 2 < N) {
              *reinterpret_cast<float3*>(crow + col) = make_float3(o[0], o[VW > 1 ? 1 : 0], o[VW > 2 ? 2 : 0]);
            } else {
#pragma unroll
              for (int u = 0; u < VW; ++u)
                if (col + u < N) crow[col + u] = o[u];
            }
          } else {
#pragma unroll
            for (int u = 0; u < VW; ++u)
              if (col + u < N) crow[col + u] = o[u];
          }
        }
      }
    }
  }
  // BatchNorm column sums of this tile AFTER its stores have been issued: the shuffles, the LDS round trip and the
  // barrier below then run while the stores drain (they were in front of the stores until round 4: every wave held its
  // output back behind the barrier).  Same arithmetic, same order: bit-identical sums.
  if (colstats) {
    float* sred = smem;                       // [2][WM][BN], the staging buffers are idle now
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float sv = 0.f, qv = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mw + SA::pos(i, 4 * jq + r);
          const float v = row < M ? acc[i][j][r] : 0.f;
          sv += v;
          qv = fmaf(v, v, qv);
        }
      sv += __shfl_xor(sv, 16, 64);
      qv += __shfl_xor(qv, 16, 64);
      sv += __shfl_xor(sv, 32, 64);
      qv += __shfl_xor(qv, 32, 64);
      if (lane < 16) {
        const int cl = wn * (TN * 16) + SB::pos(j, p);
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

}

// All MFMAs of one K tile for one wave: NCH chunks of 16 reduction steps out of the LDS buffer `cur`
// (A image first, B image at +SA::SIZE).  In the MIDDLE of the MFMA stream the next tile's register stage
// (sa_next, sb_next) is written to the idle LDS buffer `nxt` (STORE = false: nothing left to write).
template <class SA, class SB, int TM, int TN, int NCH, bool STORE>
__device__ __forceinline__ void mma_tile_store(const float* __restrict__ cur, int wrow, int wcol, int lane,
                                               f32x4 (&acc)[TM][TN], const SA& sa_next, const SB& sb_next,
                                               float* __restrict__ nxt, int tid) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (STORE && c == NCH / 2) {
      sa_next.store(nxt, tid);
      sb_next.store(nxt + SA::SIZE, tid);
    }
    float a[TM][4], b[TN][4];
    SA::frags(cur, wrow, lane, c, a);
    SB::frags(cur + SA::SIZE, wcol, lane, c, b);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][q], b[j][q], acc[i][j], 0, 0, 0);
  }
}
