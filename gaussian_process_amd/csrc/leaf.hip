// Blocked leaf of the recursive Cholesky: one workgroup factors a diagonal block of order
// nb <= 128 and inverts the factor, with the O(nb^3) work on the fp64 matrix cores.
// Replaces the unblocked dpotf2 inside scipy.linalg.cholesky (regression.py:36,49) for the
// diagonal blocks, and produces the 128x128 block inverses every triangular solve multiplies by.
//
// The leaf is pure latency: at n = 65536 there are 512 of them on the critical path (and they ARE
// the critical path of the multi-GPU factorisation), so the design minimises the length of the
// dependent chain, not flops.  The block lives in LDS (stride 130 doubles: conflict-free MFMA
// fragment reads); a ragged block is padded with an identity, which factors and inverts to itself.
//
//   factorisation, 8 panels of 16 columns:
//     A  wave 0 factors the 16x16 diagonal tile and inverts the factor in registers, four columns at a
//        time: the 4x4 pivot block is broadcast by v_readlane and factored in uniform registers (1/sqrt by
//        v_rsq_f64 + one third-order correction), panel scaling and rank-4 updates are single MFMAs
//        (diag_tile below); no LDS, no barrier
//     B  the 16-row tiles below the diagonal tile: X <- X W_tile^T on the matrix cores, one tile per wave
//     C  trailing update T_ij -= P_i P_j^T on the lower 16x16 tiles, v_mfma_f64_16x16x4_f64,
//        tiles dealt round-robin to the 8 waves
//     three barriers per panel (24 in all, against one per column = 128 before)
//   inverse, in place over L, by recursive doubling  inv([A 0; C B]) = [A^-1 0; -B^-1 C A^-1  B^-1]:
//     level 0: the 8 diagonal tiles by substitution, one wave each
//     level 1-3: T = C A^-1 stays in the MFMA accumulator registers, whose layout IS the B-operand
//        fragment layout, so W = -B^-1 T needs no LDS round trip; one wave per tile column.  (The levels
//        are bound by the CU's four matrix pipes, 64 cycles per fp64 MFMA: spreading level 2 and 3 over all
//        8 waves, two per SIMD, measured the same 12 000 cycles.)
#include "gpx_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = GPX_NB;       // 128
constexpr int LD = NB + 2;       // 130: (l & 15) * LD + (l >> 4) hits 32 distinct 8-byte banks per half wave
constexpr int TS = 16;           // tile size = MFMA M = N
constexpr int NT = NB / TS;      // 8 tiles per side
constexpr int THREADS = 512;     // 8 waves

#ifdef GPX_LEAF_PROFILE   // tools/leaf_profile.hip: cycle stamps of the phases, thread 0
__device__ long long leaf_prof[64];
#define LEAF_TICK(k) do { if (threadIdx.x == 0) leaf_prof[k] = clock64(); } while (0)
#else
#define LEAF_TICK(k)
#endif

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to full double precision: hardware estimate (relative error ~2^-23) + one third-order
// correction step (error -> 5/16 e^3; measured 2.2e-16 against a host Cholesky, same as with two steps).
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
#ifndef GPX_LEAF_NR
#define GPX_LEAF_NR 1
#endif
#pragma unroll
  for (int it = 0; it < GPX_LEAF_NR; ++it) {
    const double e = fma(-p * r, r, 1.0);          // 1 - p r^2
    r = fma(r, e * fma(e, 0.375, 0.5), r);         // r (1 + e/2 + 3 e^2 / 8)
  }
  return r;
}

// acc += sign * A(tile ar, ac) * B(tile br, bc), all tiles in the LDS image S.
__device__ __forceinline__ v4d tile_mm(const double* S, int ar, int ac, int br, int bc, v4d acc, double sign,
                                       int lane) {
  const double* pa = S + (ar * TS + (lane & 15)) * LD + ac * TS + (lane >> 4);
  const double* pb = S + (br * TS + (lane >> 4)) * LD + bc * TS + (lane & 15);
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * pa[4 * s], pb[4 * s * LD], acc, 0, 0, 0);
  return acc;
}

// acc += sign * A(tile ar, ac) * T, T held in accumulator layout (register s = rows 4s + (lane >> 4)),
// which is exactly the B fragment of k-step s.
__device__ __forceinline__ v4d tile_mm_regb(const double* S, int ar, int ac, v4d t, v4d acc, double sign, int lane) {
  const double* pa = S + (ar * TS + (lane & 15)) * LD + ac * TS + (lane >> 4);
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * pa[4 * s], t[s], acc, 0, 0, 0);
  return acc;
}

__device__ __forceinline__ void tile_store(double* S, int tr, int tc, v4d v, int lane) {
  double* p = S + (tr * TS + (lane >> 4)) * LD + tc * TS + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) p[4 * r * LD] = v[r];
}

// One level of the recursive-doubling inverse.  The block starts at tile `base` and spans 2M
// tiles: A^-1 = tiles [base, base+M), B^-1 = [base+M, base+2M), C below-left.  Wave-uniform: this
// wave owns tile column jc of C.  Phase 1 (before the barrier) T_i = sum_{k>=jc} C_ik Ainv_k,jc;
// phase 2 (after it) W_i = -sum_{k<=i} Binv_ik T_k overwrites C_i,jc.
template <int M>
__device__ __forceinline__ void inv_level(double* S, int base, int jc, bool active, int lane) {
  v4d t[M];
  if (active) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      v4d acc = {0.0, 0.0, 0.0, 0.0};
      // k < jc multiplies a zero tile (Ainv is lower triangular and the tiles above the diagonal hold
      // zeros): the static trip count lets the LDS reads of all k be in flight together
#pragma unroll
      for (int k = 0; k < M; ++k) acc = tile_mm(S, base + M + i, base + k, base + k, base + jc, acc, 1.0, lane);
      t[i] = acc;
    }
  }
  if (M > 1) __syncthreads();   // every reader of C is done before anyone overwrites it
  if (active) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k <= i; ++k) acc = tile_mm_regb(S, base + M + i, base + M + k, t[k], acc, -1.0, lane);
      tile_store(S, base + M + i, base + jc, acc, lane);
    }
  }
}

// Inverse of one 16x16 lower-triangular tile by substitution, lane (l & 15) = column j of the inverse;
// Lt: the tile in LDS (row stride ldt), rp: reciprocals of its diagonal, W: 16 x WS output.
constexpr int WS = TS + 2;   // row stride of the tile inverses kept in LDS (conflict-free B-fragment reads)
__device__ __forceinline__ void tile_inverse(const double* Lt, int ldt, const double* rp, double* W, int lane) {
  const int j = lane & 15;
  double w[TS];
#pragma unroll
  for (int i = 0; i < TS; ++i) {
    double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) acc = fma(-Lt[i * ldt + k], w[k], acc);   // uniform address: LDS broadcast
    w[i] = acc * rp[i];
  }
  if (lane < TS) {
#pragma unroll
    for (int i = 0; i < TS; ++i) W[i * WS + j] = w[i];                    // w[i] = 0 for i < j
  }
}

// In place: lower triangle of S (L) -> L^-1, given the inverses of the 8 diagonal tiles in Wall
// (8 x 16 x WS).  Entered after a barrier that makes S and Wall visible; leaves a barrier-synchronised
// result.  Tiles above the diagonal must be (and stay) zero.
__device__ void invert_lower(double* S, const double* Wall) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  // level 0: the diagonal tiles are already inverted; put them in place
  for (int e = t; e < NT * TS * TS; e += THREADS) {
    const int d = e >> 8, i = (e >> 4) & 15, j = e & 15;
    S[(d * TS + i) * LD + d * TS + j] = Wall[(d * TS + i) * WS + j];
  }
  __syncthreads();
  LEAF_TICK(30);
  inv_level<1>(S, 2 * (wid & 3), 0, wid < 4, lane);                         // four 32-blocks
  __syncthreads();
  LEAF_TICK(31);
  inv_level<2>(S, 4 * ((wid >> 1) & 1), wid & 1, wid < 4, lane);            // two 64-blocks
  __syncthreads();
  LEAF_TICK(32);
  inv_level<4>(S, 0, wid & 3, wid < 4, lane);                               // the 128-block
  __syncthreads();
  LEAF_TICK(33);
}

// The 16x16 diagonal tile of panel p as a full symmetric matrix in accumulator layout (register r of lane l =
// element [4r + (l >> 4)][l & 15]); only its lower triangle is valid in S.
__device__ __forceinline__ v4d load_sym_tile(const double* S, int p0, int lane) {
  const int g = lane >> 4, j = lane & 15;
  v4d t;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * r + g;
    t[r] = S[(p0 + max(i, j)) * LD + p0 + min(i, j)];
  }
  return t;
}

// Phase A of panel p, one wave: factor the 16x16 diagonal tile T and invert the factor, four columns at a
// time, with the rank-4 work on the matrix core.  T arrives symmetric in accumulator layout, where register b
// (rows 4b..4b+3 = by symmetry columns 4b..4b+3) IS the A/B operand fragment of column block b.  Per block b:
//   1  the 4x4 diagonal block is broadcast (10 v_readlane pairs) and every lane factors it redundantly in
//      uniform registers; lane group g = l >> 4 then solves for column g of its inverse W_bb (4 values);
//   2  P^T = W_bb T[rows of b] and X_b = W_bb E[rows of b]: one MFMA each, with W_bb padded to 16x4 as the
//      A operand -- register 0 of the result is the scaled panel L[:, 4b + g] (rows of the inverse L^-1
//      resp.) already in fragment layout;
//   3  T -= P P^T and E -= P X_b: one MFMA each (E starts as the identity and ends as nothing: its rows
//      leave as X_b, the rows of L_tile^-1).  (Issuing the E half of block b inside block b + 1's pivot chain,
//      pinned with sched_barriers, measured 4 % slower per tile: left in program order.)
// Rows / columns of T and E in front of the current block fill with garbage and are never read again.
// The serial chain is 4 x (4 pivots + 2 MFMAs) instead of 16 pivots + 120 dependent v_readlane broadcasts
// (6000 -> 4100 cycles per tile, 8 tiles on the critical path of every leaf: about 150 dependent fp64 VALU
// instructions and four 64-cycle MFMAs per block).
__device__ __forceinline__ void diag_tile(double* S, double* Wall, int p, int* fail, int lane, v4d T) {
  const int p0 = p * TS;
  double* Wt = Wall + p * TS * WS;
  const int g = lane >> 4, i = lane & 15;
  v4d E;
#pragma unroll
  for (int r = 0; r < 4; ++r) E[r] = (4 * r + g == i) ? 1.0 : 0.0;
  const v4d zero = {0.0, 0.0, 0.0, 0.0};
  const double e0 = g == 0 ? 1.0 : 0.0, e1 = g == 1 ? 1.0 : 0.0, e2 = g == 2 ? 1.0 : 0.0, e3 = g == 3 ? 1.0 : 0.0;
  const double m0 = i == 0 ? 1.0 : 0.0, m1 = i == 1 ? 1.0 : 0.0, m2 = i == 2 ? 1.0 : 0.0, m3 = i == 3 ? 1.0 : 0.0;
  int bad = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const double tb = T[b];
    // element (row 4b + gr, column 4b + c) of the tile sits in lane 16 gr + 4b + c of register b
    const double a00 = readlane_f64(tb, 4 * b);
    const double a10 = readlane_f64(tb, 16 + 4 * b), a11 = readlane_f64(tb, 16 + 4 * b + 1);
    const double a20 = readlane_f64(tb, 32 + 4 * b), a21 = readlane_f64(tb, 32 + 4 * b + 1),
                 a22 = readlane_f64(tb, 32 + 4 * b + 2);
    const double a30 = readlane_f64(tb, 48 + 4 * b), a31 = readlane_f64(tb, 48 + 4 * b + 1),
                 a32 = readlane_f64(tb, 48 + 4 * b + 2), a33 = readlane_f64(tb, 48 + 4 * b + 3);
    const double r0 = rsqrt_nr(a00);
    const double l10 = a10 * r0, l20 = a20 * r0, l30 = a30 * r0;
    const double d1 = fma(-l10, l10, a11);
    const double r1 = rsqrt_nr(d1);
    const double l21 = fma(-l20, l10, a21) * r1, l31 = fma(-l30, l10, a31) * r1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, a22));
    const double r2 = rsqrt_nr(d2);
    const double l32 = fma(-l31, l21, fma(-l30, l20, a32)) * r2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, a33)));
    const double r3 = rsqrt_nr(d3);
    // one test per block on the common path: a non-positive (or NaN) pivot makes every later one NaN, and
    // fmin skips NaNs, so the minimum is the offending pivot (or NaN if the input was)
    if (!(fmin(fmin(a00, d1), fmin(d2, d3)) > 0.0) && bad == 0)
      bad = p0 + 4 * b + (!(a00 > 0.0) ? 1 : (!(d1 > 0.0) ? 2 : (!(d2 > 0.0) ? 3 : 4)));
    // column g of W_bb = L_bb^-1 by substitution (exact zeros above the diagonal)
    const double w0 = e0 * r0;
    const double w1 = fma(-l10, w0, e1) * r1;
    const double w2 = fma(-l21, w1, fma(-l20, w0, e2)) * r2;
    const double w3 = fma(-l32, w2, fma(-l31, w1, fma(-l30, w0, e3))) * r3;
    const double wsel = fma(m3, w3, fma(m2, w2, fma(m1, w1, m0 * w0)));   // W_bb[i][g], 16x4 padded; branch-free
    const double pcol = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, tb, zero, 0, 0, 0)[0];     // L[i][4b + g]
    const double xrow = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, E[b], zero, 0, 0, 0)[0];   // Linv[4b + g][i]
    if (b < 3) {
      T = __builtin_amdgcn_mfma_f64_16x16x4f64(-pcol, pcol, T, 0, 0, 0);
      E = __builtin_amdgcn_mfma_f64_16x16x4f64(-pcol, xrow, E, 0, 0, 0);
    }
    S[(p0 + i) * LD + p0 + 4 * b + g] = i >= 4 * b + g ? pcol : 0.0;
    Wt[(4 * b + g) * WS + i] = i <= 4 * b + g ? xrow : 0.0;
  }
  if (lane == 0 && bad != 0 && *fail == 0) *fail = bad;
}

// Factor the (padded) 128x128 SPD block in S in place (lower triangle; strict upper parts of the
// diagonal tiles are zeroed) and leave the inverses of the 8 diagonal 16x16 tiles in Wall.
// *fail = first non-positive pivot + 1 (0 = ok).  Per panel p of 16 columns:
//   B  the 16-row tiles below the diagonal tile, one per wave: X <- X W^T on the matrix cores
//      (A fragment X[l & 15][4s + (l >> 4)], B fragment W^T[4s + (l >> 4)][l & 15] = W[l & 15][4s + (l >> 4)])
//   C  trailing lower tiles (ti >= tj > p) -= P_ti P_tj^T.  Wave 0 takes the next diagonal tile and
//      goes straight on to phase A of panel p + 1 (the longest serial piece) while waves 1-7 share the
//      other tiles, <= 4 each, processed together: all LDS reads first, the MFMA chains interleaved,
//      then the writes.
// Two barriers per panel.
__device__ void factor_lower(double* S, double* Wall, int* fail) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  if (wid == 0) diag_tile(S, Wall, 0, fail, lane, load_sym_tile(S, 0, lane));
  __syncthreads();
  LEAF_TICK(2);
  for (int p = 0; p + 1 < NT; ++p) {
    const int p0 = p * TS;
    const double* Wt = Wall + p * TS * WS;
    {
      const int ti = p + 1 + wid;
      if (ti < NT) {
        const double* px = S + (ti * TS + (lane & 15)) * LD + p0 + (lane >> 4);
        const double* pw = Wt + (lane & 15) * WS + (lane >> 4);
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(px[4 * s4], pw[4 * s4], acc, 0, 0, 0);
        tile_store(S, ti, p, acc, lane);
      }
    }
    __syncthreads();
    LEAF_TICK(3 + 2 * p);
    if (wid == 0) {
      const int tn = p + 1;
      const double* pa = S + (tn * TS + (lane & 15)) * LD + p0 + (lane >> 4);
      v4d acc = load_sym_tile(S, tn * TS, lane);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * r], pa[4 * r], acc, 0, 0, 0);
      diag_tile(S, Wall, tn, fail, lane, acc);  // the updated tile goes on in registers, symmetric
    } else {
      const int m = NT - p - 1, ntile = m * (m + 1) / 2;   // tile 0 of the triangle is wave 0's
      double* pc[4];
      const double* pa[4];
      const double* pb[4];
      v4d acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = wid + 7 * q;
        if (c < ntile) {
          int row = 0;
          while ((row + 1) * (row + 2) / 2 <= c) ++row;   // wave-uniform, <= 7 steps
          const int ti = p + 1 + row, tj = p + 1 + c - row * (row + 1) / 2;
          pc[q] = S + (ti * TS + (lane >> 4)) * LD + tj * TS + (lane & 15);
          pa[q] = S + (ti * TS + (lane & 15)) * LD + p0 + (lane >> 4);
          pb[q] = S + (tj * TS + (lane & 15)) * LD + p0 + (lane >> 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[q][r] = pc[q][4 * r * LD];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (wid + 7 * q < ntile)
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[q][4 * r], pb[q][4 * r], acc[q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (wid + 7 * q < ntile) {
#pragma unroll
          for (int r = 0; r < 4; ++r) pc[q][4 * r * LD] = acc[q][r];
        }
    }
    __syncthreads();
    LEAF_TICK(4 + 2 * p);
  }
}

// S <- lower triangle of the nb x nb block at A, identity-padded to 128 x 128, zeros above the
// diagonal.  16-byte loads, all 16 of a thread in flight before the first LDS write (the loop it
// replaces took 15 us of dependent scalar loads).  Reads A[i][i+1] next to a diagonal element: inside
// the row because lda is even and >= n.
__device__ __forceinline__ void load_block(double* S, const double* A, long lda, int nb) {
  constexpr int PER = NB * NB / 2 / THREADS;   // 16 pairs per thread
  v2d v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    v[u] = (i < nb && j <= i) ? *reinterpret_cast<const v2d*>(A + (long)i * lda + j) : v2d{0.0, 0.0};
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    v2d w;
    w.x = (i < nb && j <= i) ? v[u].x : (i == j ? 1.0 : 0.0);
    w.y = (i < nb && j + 1 <= i) ? v[u].y : (i == j + 1 ? 1.0 : 0.0);
    *reinterpret_cast<v2d*>(S + i * LD + j) = w;
  }
}

// nb x nb block at A <- lower triangle of S, zeros above the diagonal.
__device__ __forceinline__ void store_block(const double* S, double* A, long lda, int nb) {
  constexpr int PER = NB * NB / 2 / THREADS;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    if (i >= nb || j >= nb) continue;
    v2d w = *reinterpret_cast<const v2d*>(S + i * LD + j);
    if (j > i) w.x = 0.0;
    if (j + 1 > i) w.y = 0.0;
    if (j + 1 < nb) *reinterpret_cast<v2d*>(A + (long)i * lda + j) = w;
    else A[(long)i * lda + j] = w.x;
  }
}

__device__ __forceinline__ void store_dinv(const double* S, double* dinv) {
  constexpr int PER = NB * NB / 2 / THREADS;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    v2d w = *reinterpret_cast<const v2d*>(S + i * LD + j);
    if (j > i) w.x = 0.0;
    if (j + 1 > i) w.y = 0.0;
    *reinterpret_cast<v2d*>(dinv + i * NB + j) = w;
  }
}

__global__ __launch_bounds__(THREADS) void leaf_blocked_kernel(int nb, double* A, long lda, double* dinv,
                                                               int* info, int global_off) {
  __shared__ double S[NB * LD];
  __shared__ double Wall[NT * TS * WS];
  __shared__ int fail;
  const int t = threadIdx.x;
  LEAF_TICK(0);
  if (t == 0) fail = 0;
  load_block(S, A, lda, nb);
  __syncthreads();
  LEAF_TICK(1);
  factor_lower(S, Wall, &fail);
  store_block(S, A, lda, nb);
  if (t == 0 && fail != 0 && fail <= nb && *info == 0) *info = global_off + fail;
  __syncthreads();
  LEAF_TICK(29);
  invert_lower(S, Wall);
  store_dinv(S, dinv);
  LEAF_TICK(34);
}

// dinv for an existing factor: one workgroup per diagonal block.
__global__ __launch_bounds__(THREADS) void leaf_blocked_inv_kernel(int n, const double* L, long ldl, double* dinv) {
  __shared__ double S[NB * LD];
  __shared__ double Wall[NT * TS * WS];
  __shared__ double rp[NB];
  const int t = threadIdx.x;
  const int off = blockIdx.x * NB;
  const int nb = min(NB, n - off);
  const double* A = L + (long)off * ldl + off;
  load_block(S, A, ldl, nb);
  if (t < NB) rp[t] = t < nb ? 1.0 / A[(long)t * ldl + t] : 1.0;
  __syncthreads();
  const int wid = t >> 6;
  tile_inverse(S + wid * TS * LD + wid * TS, LD, rp + wid * TS, Wall + wid * TS * WS, t & 63);
  __syncthreads();
  invert_lower(S, Wall);
  store_dinv(S, dinv + (size_t)blockIdx.x * NB * NB);
}

__global__ void copy_dinv_kernel(int nb, const double* dinv, double* L, long ldl) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nb * nb) return;
  const int i = e / nb, j = e - i * nb;
  L[(long)i * ldl + j] = dinv[i * NB + j];  // dinv already has zeros above the diagonal
}

__global__ void copy_dinv_batched_kernel(const double* dinv, double* L, long ldl) {
  const int blk = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0 .. 128 * 128 - 1
  const int i = e >> 7, j = e & 127;
  L[((long)blk * NB + i) * ldl + (long)blk * NB + j] = dinv[(size_t)blk * NB * NB + e];
}

}  // namespace

int gpx_leaf_potf2_inv(hipStream_t s, int nb, double* A, long lda, double* dinv_blk, int* info_dev,
                       int global_off) {
  GPX_REQUIRE(nb > 0 && nb <= NB, "leaf: nb out of range");
  hipLaunchKernelGGL(leaf_blocked_kernel, dim3(1), dim3(THREADS), 0, s, nb, A, lda, dinv_blk, info_dev, global_off);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_leaf_diag_inv_batched(hipStream_t s, int n, const double* L, long ldl, double* dinv) {
  const int nblk = (n + NB - 1) / NB;
  hipLaunchKernelGGL(leaf_blocked_inv_kernel, dim3(nblk), dim3(THREADS), 0, s, n, L, ldl, dinv);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_copy_dinv_to_diag(hipStream_t s, int nb, const double* dinv_blk, double* L, long ldl) {
  const int tot = nb * nb;
  hipLaunchKernelGGL(copy_dinv_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, nb, dinv_blk, L, ldl);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_copy_dinv_to_diag_batched(hipStream_t s, int nblk, const double* dinv, double* L, long ldl) {
  hipLaunchKernelGGL(copy_dinv_batched_kernel, dim3(NB * NB / 256, nblk), dim3(256), 0, s, dinv, L, ldl);
  GPX_HIP(hipGetLastError());
  return 0;
}
