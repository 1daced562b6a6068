// Leaf of the recursive Cholesky: one workgroup factors a diagonal block of order nb <= 128
// entirely in LDS and also produces the block's inverse (used by the GEMM-shaped triangular
// solves).  This is the "panel factorisation with wavefront-level work" of the north star; it
// replaces the unblocked dpotf2 inside scipy.linalg.cholesky (regression.py:36,49).
#include "gpx_internal.h"

namespace {

constexpr int NB = GPX_NB;      // 128
constexpr int LDS_LD = NB + 1;  // 129: odd stride -> column walks are conflict free
constexpr int LEAF_THREADS = 512;

// In-LDS right-looking Cholesky of S (nb x nb, lower).  Returns first failing column + 1 (or 0)
// in *fail (LDS int).  Two barriers per column.
__device__ void potf2_lds(double* S, int nb, int* fail) {
  const int t = threadIdx.x;
  for (int j = 0; j < nb; ++j) {
    const double piv = S[j * LDS_LD + j];
    const bool bad = !(piv > 0.0);
    if (bad && t == 0 && *fail == 0) *fail = j + 1;
    const double d = sqrt(piv);
    // scale column j below the diagonal
    for (int i = j + 1 + t; i < nb; i += LEAF_THREADS) S[i * LDS_LD + j] = S[i * LDS_LD + j] / d;
    __syncthreads();
    if (t == 0) S[j * LDS_LD + j] = d;  // nobody reads S[j][j] during the update below
    // trailing update: S[i][k] -= S[i][j] * S[k][j], j < k <= i
    const int rem = nb - j - 1;
    if (rem > 0) {
      // 16 threads per row chunk: thread -> (row slot, column lane)
      const int cl = t & 15, rs = t >> 4;  // 32 row slots
      for (int i = j + 1 + rs; i < nb; i += LEAF_THREADS / 16) {
        const double lij = S[i * LDS_LD + j];
        for (int k = j + 1 + cl; k <= i; k += 16) S[i * LDS_LD + k] -= lij * S[k * LDS_LD + j];
      }
    }
    __syncthreads();
  }
}

// In-LDS inverse of the lower-triangular S (nb x nb), in place, column by column from the right
// (LAPACK dtrti2 'L'): column j of the inverse = -inv(S22) * S[j+1:, j] / S[j][j], with inv(S22)
// already in place.  tmp: nb doubles of LDS.
__device__ void trti2_lds(double* S, int nb, double* tmp) {
  const int t = threadIdx.x;
  const int sub = t & 3, row_t = t >> 2;  // 4 threads cooperate on one row's dot product
  for (int j = nb - 1; j >= 0; --j) {
    const double ajj = 1.0 / S[j * LDS_LD + j];
    for (int i = j + 1 + t; i < nb; i += LEAF_THREADS) tmp[i] = S[i * LDS_LD + j];
    __syncthreads();
    // rows i > j: x_i = sum_{p=j+1..i} Sinv[i][p] * tmp[p]
    for (int i = j + 1 + row_t; i < nb; i += LEAF_THREADS / 4) {
      double acc = 0.0;
      for (int p = j + 1 + sub; p <= i; p += 4) acc += S[i * LDS_LD + p] * tmp[p];
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (sub == 0) S[i * LDS_LD + j] = -acc * ajj;
    }
    if (t == 0) S[j * LDS_LD + j] = ajj;
    __syncthreads();
  }
}

__global__ __launch_bounds__(LEAF_THREADS) void leaf_potf2_inv_kernel(int nb, double* A, long lda,
                                                                       double* dinv, int* info,
                                                                       int global_off) {
  __shared__ double S[NB * LDS_LD];
  __shared__ double tmp[NB];
  __shared__ int fail;
  const int t = threadIdx.x;
  if (t == 0) fail = 0;
  // load the lower triangle (pad to identity beyond nb so the inverse block is well defined)
  for (int e = t; e < NB * NB; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    double v = 0.0;
    if (i < nb && j < nb) {
      if (j <= i) v = A[(long)i * lda + j];
    } else if (i == j) {
      v = 1.0;
    }
    S[i * LDS_LD + j] = v;
  }
  __syncthreads();
  potf2_lds(S, nb, &fail);
  // write L (zero the strict upper triangle inside the block)
  for (int e = t; e < nb * nb; e += LEAF_THREADS) {
    const int i = e / nb, j = e - i * nb;
    A[(long)i * lda + j] = (j <= i) ? S[i * LDS_LD + j] : 0.0;
  }
  if (t == 0 && fail != 0 && *info == 0) *info = global_off + fail;
  __syncthreads();
  trti2_lds(S, NB, tmp);
  for (int e = t; e < NB * NB; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    dinv[e] = (j <= i) ? S[i * LDS_LD + j] : 0.0;
  }
}

// dinv for an existing factor: one workgroup per diagonal block.
__global__ __launch_bounds__(LEAF_THREADS) void leaf_diag_inv_kernel(int n, const double* L, long ldl,
                                                                      double* dinv) {
  __shared__ double S[NB * LDS_LD];
  __shared__ double tmp[NB];
  const int t = threadIdx.x;
  const int off = blockIdx.x * NB;
  const int nb = min(NB, n - off);
  const double* A = L + (long)off * ldl + off;
  for (int e = t; e < NB * NB; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    double v = 0.0;
    if (i < nb && j < nb) {
      if (j <= i) v = A[(long)i * ldl + j];
    } else if (i == j) {
      v = 1.0;
    }
    S[i * LDS_LD + j] = v;
  }
  __syncthreads();
  trti2_lds(S, NB, tmp);
  double* out = dinv + (size_t)blockIdx.x * NB * NB;
  for (int e = t; e < NB * NB; e += LEAF_THREADS) {
    const int i = e >> 7, j = e & 127;
    out[e] = (j <= i) ? S[i * LDS_LD + j] : 0.0;
  }
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
