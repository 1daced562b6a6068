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
//     A  wave 0 factors the 16x16 diagonal tile in registers (lane = row; pivots and multipliers
//        travel by v_readlane, no LDS, no barrier; 1/sqrt by v_rsq_f64 + 2 Newton steps)
//     B  one lane per row below the tile: x <- x L_tile^-T by substitution (multipliers are
//        uniform LDS broadcasts), <= 112 rows
//     C  trailing update T_ij -= P_i P_j^T on the lower 16x16 tiles, v_mfma_f64_16x16x4_f64,
//        tiles dealt round-robin to the 8 waves
//     three barriers per panel (24 in all, against one per column = 128 before)
//   inverse, in place over L, by recursive doubling  inv([A 0; C B]) = [A^-1 0; -B^-1 C A^-1  B^-1]:
//     level 0: the 8 diagonal tiles by substitution, one wave each
//     level 1-3: T = C A^-1 stays in the MFMA accumulator registers, whose layout IS the B-operand
//        fragment layout, so W = -B^-1 T needs no LDS round trip; one wave per tile column.
#include "gpx_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = GPX_NB;       // 128
constexpr int LD = NB + 2;       // 130: (l & 15) * LD + (l >> 4) hits 32 distinct 8-byte banks per half wave
constexpr int TS = 16;           // tile size = MFMA M = N
constexpr int NT = NB / TS;      // 8 tiles per side
constexpr int THREADS = 512;     // 8 waves

__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1/sqrt(p) to full double precision: hardware estimate + two Newton-Raphson steps.
__device__ __forceinline__ double rsqrt_nr(double p) {
  double r = __builtin_amdgcn_rsq(p);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = fma(-p * r, r, 1.0);          // 1 - p r^2
    r = fma(r, fma(e, 0.375 * e, 0.5 * e), r);     // r (1 + e/2 + 3 e^2 / 8)
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
      for (int k = jc; k < M; ++k) acc = tile_mm(S, base + M + i, base + k, base + k, base + jc, acc, 1.0, lane);
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

// In place: lower triangle of S (L, diagonal reciprocals in rp) -> L^-1.  Entered after a barrier
// that makes S and rp visible; leaves a barrier-synchronised result.  Strict upper parts of the
// diagonal tiles must be (and stay) zero; tiles above the diagonal are never touched.
__device__ void invert_lower(double* S, const double* rp) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  // level 0: diagonal tile `wid`, lane (l & 15) = column j of the tile's inverse
  {
    const int o = wid * TS, j = lane & 15;
    const double* Lt = S + o * LD + o;
    double w[TS];
#pragma unroll
    for (int i = 0; i < TS; ++i) {
      double acc = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) acc = fma(-Lt[i * LD + k], w[k], acc);   // uniform address: LDS broadcast
      w[i] = acc * rp[o + i];
    }
    if (lane < TS) {
#pragma unroll
      for (int i = 0; i < TS; ++i) S[(o + i) * LD + o + j] = w[i];          // w[i] = 0 for i < j
    }
  }
  __syncthreads();
  inv_level<1>(S, 2 * (wid & 3), 0, wid < 4, lane);                         // four 32-blocks
  __syncthreads();
  inv_level<2>(S, 4 * ((wid >> 1) & 1), wid & 1, wid < 4, lane);            // two 64-blocks
  __syncthreads();
  inv_level<4>(S, 0, wid & 3, wid < 4, lane);                               // the 128-block
  __syncthreads();
}

// Factor the (padded) 128x128 SPD block in S in place (lower triangle; strict upper parts of the
// diagonal tiles are zeroed).  rp[j] = 1 / L[j][j].  *fail = first non-positive pivot + 1 (0 = ok).
__device__ void factor_lower(double* S, double* rp, double* Ltile, int* fail) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  for (int p = 0; p < NT; ++p) {
    const int p0 = p * TS;
    // ---- A: diagonal tile, wave 0, lane (l & 15) = row
    if (wid == 0) {
      const int r = lane & 15;
      double a[TS], rpj[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) a[c] = S[(p0 + r) * LD + p0 + c];
      int bad = 0;
#pragma unroll
      for (int j = 0; j < TS; ++j) {
        const double d = readlane_f64(a[j], j);
        if (!(d > 0.0) && bad == 0) bad = p0 + j + 1;
        const double rs = rsqrt_nr(d);
        rpj[j] = rs;
        a[j] *= rs;                                  // row j: d / sqrt(d) = L_jj; rows below: L_ij
#pragma unroll
        for (int c = j + 1; c < TS; ++c) {
          const double lcj = readlane_f64(a[j], c);  // L[c][j]
          a[c] = fma(-a[j], lcj, a[c]);
        }
      }
      if (lane < TS) {
#pragma unroll
        for (int c = 0; c < TS; ++c) {
          const double v = c <= r ? a[c] : 0.0;
          S[(p0 + r) * LD + p0 + c] = v;
          Ltile[r * TS + c] = v;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < TS; ++j) rp[p0 + j] = rpj[j];
        if (bad != 0 && *fail == 0) *fail = bad;
      }
    }
    __syncthreads();
    // ---- B: rows below the tile, one lane per row: x <- x Ltile^-T
    const int below = NB - p0 - TS;
    if (t < below) {
      double* row = S + (p0 + TS + t) * LD + p0;
      double x[TS];
#pragma unroll
      for (int c = 0; c < TS; ++c) x[c] = row[c];
#pragma unroll
      for (int k = 0; k < TS; ++k) {
        x[k] *= rp[p0 + k];
#pragma unroll
        for (int c = k + 1; c < TS; ++c) x[c] = fma(-x[k], Ltile[c * TS + k], x[c]);
      }
#pragma unroll
      for (int c = 0; c < TS; ++c) row[c] = x[c];
    }
    __syncthreads();
    // ---- C: trailing lower tiles (ti >= tj > p) -= P_ti P_tj^T, dealt round-robin to the waves
    int cnt = 0;
    for (int ti = p + 1; ti < NT; ++ti) {
      for (int tj = p + 1; tj <= ti; ++tj, ++cnt) {
        if ((cnt & 7) != wid) continue;
        double* pc = S + (ti * TS + (lane >> 4)) * LD + tj * TS + (lane & 15);
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = pc[4 * r * LD];
        const double* pa = S + (ti * TS + (lane & 15)) * LD + p0 + (lane >> 4);
        const double* pb = S + (tj * TS + (lane & 15)) * LD + p0 + (lane >> 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-pa[4 * s], pb[4 * s], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) pc[4 * r * LD] = acc[r];
      }
    }
    __syncthreads();
  }
}

__device__ __forceinline__ void store_dinv(const double* S, double* dinv) {
  for (int e = threadIdx.x; e < NB * NB; e += THREADS) {
    const int i = e >> 7, j = e & 127;
    dinv[e] = j <= i ? S[i * LD + j] : 0.0;
  }
}

__global__ __launch_bounds__(THREADS) void leaf_blocked_kernel(int nb, double* A, long lda, double* dinv,
                                                               int* info, int global_off) {
  __shared__ double S[NB * LD];
  __shared__ double Ltile[TS * TS];
  __shared__ double rp[NB];
  __shared__ int fail;
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  for (int e = t; e < NB * NB; e += THREADS) {
    const int i = e >> 7, j = e & 127;
    S[i * LD + j] = (i < nb && j <= i) ? A[(long)i * lda + j] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  factor_lower(S, rp, Ltile, &fail);
  for (int e = t; e < nb * nb; e += THREADS) {
    const int i = e / nb, j = e - i * nb;
    A[(long)i * lda + j] = j <= i ? S[i * LD + j] : 0.0;
  }
  if (t == 0 && fail != 0 && fail <= nb && *info == 0) *info = global_off + fail;
  __syncthreads();
  invert_lower(S, rp);
  store_dinv(S, dinv);
}

// dinv for an existing factor: one workgroup per diagonal block.
__global__ __launch_bounds__(THREADS) void leaf_blocked_inv_kernel(int n, const double* L, long ldl, double* dinv) {
  __shared__ double S[NB * LD];
  __shared__ double rp[NB];
  const int t = threadIdx.x;
  const int off = blockIdx.x * NB;
  const int nb = min(NB, n - off);
  const double* A = L + (long)off * ldl + off;
  for (int e = t; e < NB * NB; e += THREADS) {
    const int i = e >> 7, j = e & 127;
    S[i * LD + j] = (i < nb && j <= i) ? A[(long)i * ldl + j] : (i == j ? 1.0 : 0.0);
  }
  if (t < NB) rp[t] = t < nb ? 1.0 / A[(long)t * ldl + t] : 1.0;
  __syncthreads();
  invert_lower(S, rp);
  store_dinv(S, dinv + (size_t)blockIdx.x * NB * NB);
}

__global__ void copy_dinv_kernel(int nb, const double* dinv, double* L, long ldl) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nb * nb) return;
  const int i = e / nb, j = e - i * nb;
  L[(long)i * ldl + j] = dinv[i * NB + j];  // dinv already has zeros above the diagonal
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
