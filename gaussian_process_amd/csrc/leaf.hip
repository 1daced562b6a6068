// Leaf of the recursive Cholesky: one workgroup factors a diagonal block of order nb <= 128 and
// also produces the block's inverse (used by the GEMM-shaped triangular solves).  This is the
// "panel factorisation with wavefront-level work" of the north star; it replaces the unblocked
// dpotf2 inside scipy.linalg.cholesky (regression.py:36,49).
//
// The leaf is pure latency (512 of them sit on the critical path at n = 65536), so both halves are
// organised around the length of the dependent chain per column, not around flops:
//   * potf2: the 128 x 128 block lives in REGISTERS (each of the 512 threads owns an 8 x 4 tile);
//     per column only the column itself crosses LDS (double-buffered), ONE barrier per column,
//     the pivot's reciprocal square root is a v_rsq_f64 + two Newton steps computed redundantly
//     by every thread instead of a sqrt and a divide.
//   * inverse: row-oriented forward substitution, 4 adjacent lanes per column of the inverse,
//     partial dot products combined with DPP shuffles, no workgroup barrier at all; the inverse
//     is written transposed into the (otherwise unused) upper triangle of the LDS image of L.
#include "gpx_internal.h"

namespace {

constexpr int NB = GPX_NB;      // 128
constexpr int LDS_LD = NB + 1;  // 129: odd stride -> row and column walks are conflict free
constexpr int LEAF_THREADS = 512;

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

// Right-looking Cholesky of the 128 x 128 block held in registers (thread (rb, cb) owns rows
// rb*8.., cols cb*4..).  On exit S (LDS, LDS_LD stride) holds L in its lower triangle incl. the
// diagonal, zeros elsewhere are NOT written; rdiag[j] = 1 / L[j][j].  *fail = first bad column + 1.
__device__ void potf2_regs(double (&a)[8][4], int nb, double* S, double* colbuf, double* rdiag, int* fail) {
  const int t = threadIdx.x;
  const int rb = t >> 5, cb = t & 31;
  const int row0 = rb * 8, col0 = cb * 4;
  const bool has_lower = col0 <= row0 + 7;
  if (cb == 0) {  // publish column 0
#pragma unroll
    for (int u = 0; u < 8; ++u) colbuf[row0 + u] = a[u][0];
  }
  for (int j = 0; j < nb; ++j) {
    const double* cbuf = colbuf + (j & 1) * NB;
    double* nbuf = colbuf + ((j + 1) & 1) * NB;
    __syncthreads();  // column j is in cbuf; everybody is done with nbuf (column j - 1)
    const double p = cbuf[j];
    double ti[8], ck[4];  // unconditional LDS reads (all in range), masks applied afterwards
#pragma unroll
    for (int u = 0; u < 8; ++u) ti[u] = cbuf[row0 + u];
#pragma unroll
    for (int w = 0; w < 4; ++w) ck[w] = cbuf[col0 + w];
    if (t == 0 && !(p > 0.0) && *fail == 0) *fail = j + 1;
    const double r = rsqrt_nr(p);
    double invp = r * r;
    invp = fma(invp, fma(-p, invp, 1.0), invp);  // one Newton step on 1/p
    if (cb == (j >> 2)) {  // column j of L: a_ij / sqrt(p); the diagonal is p / sqrt(p)
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = row0 + u;
        if (i >= j) S[i * LDS_LD + j] = ti[u] * r;
      }
      if (row0 <= j && j < row0 + 8) rdiag[j] = r;
    }
    const bool active = has_lower && row0 + 7 > j && col0 + 3 > j;
#pragma unroll
    for (int u = 0; u < 8; ++u) ti[u] = (row0 + u > j) ? ti[u] * invp : 0.0;
#pragma unroll
    for (int w = 0; w < 4; ++w) ck[w] = (col0 + w > j) ? ck[w] : 0.0;
    // look-ahead: the owners of column j + 1 update that column first and publish it, so the
    // next pivot's dependent chain starts while everybody's bulk update is still issuing
    const int jn = j + 1;
    if (jn < nb && cb == (jn >> 2)) {
      switch (jn & 3) {  // uniform
        case 0:
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u][0] = fma(-ti[u], ck[0], a[u][0]); nbuf[row0 + u] = a[u][0]; }
          ck[0] = 0.0;
          break;
        case 1:
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u][1] = fma(-ti[u], ck[1], a[u][1]); nbuf[row0 + u] = a[u][1]; }
          ck[1] = 0.0;
          break;
        case 2:
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u][2] = fma(-ti[u], ck[2], a[u][2]); nbuf[row0 + u] = a[u][2]; }
          ck[2] = 0.0;
          break;
        default:
#pragma unroll
          for (int u = 0; u < 8; ++u) { a[u][3] = fma(-ti[u], ck[3], a[u][3]); nbuf[row0 + u] = a[u][3]; }
          ck[3] = 0.0;
          break;
      }
    }
    if (active) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) a[u][w] = fma(-ti[u], ck[w], a[u][w]);
    }
  }
}

// Exchange within a quad of lanes with a DPP quad_perm move (no LDS round trip).
template <int QUAD_PERM>
__device__ __forceinline__ double quad_xor(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, QUAD_PERM, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, QUAD_PERM, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// In place: the inverse X of the lower-triangular L (lower triangle of S, diagonal reciprocals in
// rdiag) is written TRANSPOSED into the upper triangle of S: X[i][j] -> S[j][i], X[j][j] -> S[j][j].
// Must be entered after a barrier that makes S and rdiag visible; no barrier inside.
__device__ void trti2_rows(double* S, int nb, const double* rdiag) {
  const int t = threadIdx.x;
  const int j = t >> 2, sub = t & 3;  // 4 adjacent lanes share column j of X
  if (j >= nb) return;
  double* xrow = S + j * LDS_LD;      // X[p][j] lives at xrow[p]
  // every L[j][j] has been copied out already: its slot becomes X[j][j]
  if (sub == 0) xrow[j] = rdiag[j];
  for (int i = j + 1; i < nb; ++i) {
    const double* lrow = S + i * LDS_LD;
    double acc = 0.0;
    for (int p = j + sub; p < i; p += 16) {  // 4 independent LDS read pairs in flight per lane
      double l[4], x[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int pp = p + 4 * m;
        const int idx = pp < i ? pp : j;  // clamp: always a finished entry
        l[m] = lrow[idx];
        x[m] = xrow[idx];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) acc += (p + 4 * m < i) ? l[m] * x[m] : 0.0;
    }
    acc += quad_xor<0xB1>(acc);  // lanes ^ 1
    acc += quad_xor<0x4E>(acc);  // lanes ^ 2
    if (sub == 0) xrow[i] = -acc * rdiag[i];
  }
}

__device__ __forceinline__ void write_dinv(const double* S, int nb, double* dinv) {
  for (int e = threadIdx.x; e < NB * NB; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    double v = 0.0;
    if (i < nb && j < nb) {
      if (j <= i) v = S[j * LDS_LD + i];
    } else if (i == j) {
      v = 1.0;
    }
    dinv[e] = v;
  }
}

__global__ __launch_bounds__(LEAF_THREADS) void leaf_potf2_inv_kernel(int nb, double* A, long lda,
                                                                       double* dinv, int* info,
                                                                       int global_off) {
  __shared__ double S[NB * LDS_LD];
  __shared__ double colbuf[2 * NB];
  __shared__ double rdiag[NB];
  __shared__ int fail;
  const int t = threadIdx.x;
  const int rb = t >> 5, cb = t & 31;
  if (t == 0) fail = 0;
  double a[8][4];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int i = rb * 8 + u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int k = cb * 4 + w;
      a[u][w] = (i < nb && k <= i) ? A[(long)i * lda + k] : 0.0;
    }
  }
  __syncthreads();
  potf2_regs(a, nb, S, colbuf, rdiag, &fail);
  __syncthreads();
  // write L (zero the strict upper triangle inside the block)
  for (int e = t; e < nb * nb; e += LEAF_THREADS) {
    const int i = e / nb, j = e - i * nb;
    A[(long)i * lda + j] = (j <= i) ? S[i * LDS_LD + j] : 0.0;
  }
  if (t == 0 && fail != 0 && *info == 0) *info = global_off + fail;
  __syncthreads();
  trti2_rows(S, nb, rdiag);
  __syncthreads();
  write_dinv(S, nb, dinv);
}

// dinv for an existing factor: one workgroup per diagonal block.
__global__ __launch_bounds__(LEAF_THREADS) void leaf_diag_inv_kernel(int n, const double* L, long ldl,
                                                                      double* dinv) {
  __shared__ double S[NB * LDS_LD];
  __shared__ double rdiag[NB];
  const int t = threadIdx.x;
  const int off = blockIdx.x * NB;
  const int nb = min(NB, n - off);
  const double* A = L + (long)off * ldl + off;
  for (int e = t; e < nb * nb; e += LEAF_THREADS) {
    const int i = e / nb, j = e - i * nb;
    if (j <= i) S[i * LDS_LD + j] = A[(long)i * ldl + j];
  }
  if (t < nb) rdiag[t] = 1.0 / A[(long)t * ldl + t];
  __syncthreads();
  trti2_rows(S, nb, rdiag);
  __syncthreads();
  write_dinv(S, nb, dinv + (size_t)blockIdx.x * NB * NB);
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
  hipLaunchKernelGGL(leaf_potf2_inv_kernel, dim3(1), dim3(LEAF_THREADS), 0, s, nb, A, lda, dinv_blk,
                     info_dev, global_off);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_leaf_diag_inv_batched(hipStream_t s, int n, const double* L, long ldl, double* dinv) {
  const int nblk = (n + NB - 1) / NB;
  hipLaunchKernelGGL(leaf_diag_inv_kernel, dim3(nblk), dim3(LEAF_THREADS), 0, s, n, L, ldl, dinv);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_copy_dinv_to_diag(hipStream_t s, int nb, const double* dinv_blk, double* L, long ldl) {
  const int tot = nb * nb;
  hipLaunchKernelGGL(copy_dinv_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, nb, dinv_blk, L, ldl);
  GPX_HIP(hipGetLastError());
  return 0;
}
