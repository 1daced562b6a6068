// Triangular solves with ONE right-hand side (HBM-bound): x <- L^-1 x and x <- L^-T x.
//
// These are the vector solves of regression.py:50 (a = L^-1 y), regression.py:61 (the second half
// of cho_solve) and expectation_propagation.py:63,92.  The factor is read exactly once: the
// recursion  [L11 0; L21 L22]  ->  solve(L11), x2 -= L21 x1 (GEMV over all CUs), solve(L22)
// sends every off-diagonal element through one streaming GEMV, and the diagonal leaves (<= 1024
// rows) are solved by one workgroup from the 128x128 diagonal-block inverses gpx_potrf left in
// dinv (so the leaf is two small mat-vecs per block, no divisions, no scalar recurrence).
// Algorithmic bytes: 4 n^2 (lower triangle of L) + 64 KiB per diagonal block of dinv.
// All reductions have a fixed association order: results are bitwise reproducible.
#include <cstdlib>

#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = GPX_NB;
constexpr int LEAF_MAX = 1024;      // rows a leaf workgroup can hold
constexpr int LEAF_THREADS = 512;   // 8 waves
constexpr int XCHUNK = 2048;        // columns of x staged per pass in the GEMV

// ---- 16 lanes per row: a lane owns 8 consecutive columns (4 x 16-byte loads) of a 128-column
// segment; 4 rows per wave and pass.  seg_partial = the lane's share of the dot product.
struct Seg { v2d a0, a1, a2, a3; };

__device__ __forceinline__ Seg seg_load(const double* row128, int l16) {
  const v2d* p = reinterpret_cast<const v2d*>(row128 + l16 * 8);
  Seg s;
  s.a0 = p[0]; s.a1 = p[1]; s.a2 = p[2]; s.a3 = p[3];
  return s;
}

__device__ __forceinline__ double seg_partial(const Seg& s, const double* x128, int l16, double acc) {
  const double* x = x128 + l16 * 8;
  acc = fma(s.a0.x, x[0], acc);
  acc = fma(s.a0.y, x[1], acc);
  acc = fma(s.a1.x, x[2], acc);
  acc = fma(s.a1.y, x[3], acc);
  acc = fma(s.a2.x, x[4], acc);
  acc = fma(s.a2.y, x[5], acc);
  acc = fma(s.a3.x, x[6], acc);
  acc = fma(s.a3.y, x[7], acc);
  return acc;
}

// sum over the 16 lanes of a row group (every lane of the group gets the total)
__device__ __forceinline__ double group16_sum(double v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// ---- forward leaf: x <- L^-1 x for n <= LEAF_MAX rows, one workgroup --------------------------
__global__ __launch_bounds__(LEAF_THREADS) void trsv_leaf_fwd_kernel(int n, const double* L, long ldl,
                                                                     const double* dinv, double* x,
                                                                     long incx) {
  __shared__ double xs[LEAF_MAX];
  __shared__ double ys[NB];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = lane >> 4, l16 = lane & 15;
  for (int i = t; i < LEAF_MAX; i += LEAF_THREADS) xs[i] = i < n ? x[(long)i * incx] : 0.0;
  __syncthreads();
  const int nblk = (n + NB - 1) / NB;
  for (int b = 0; b < nblk; ++b) {
    const int b0 = b * NB;
    // x_b = dinv_b r_b   (dinv blocks are full 128x128, identity-padded past n)
    const double* D = dinv + (size_t)b * NB * NB;
    {
      Seg s[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) s[p] = seg_load(D + (p * 32 + wid * 4 + g) * NB, l16);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const double v = group16_sum(seg_partial(s[p], xs + b0, l16, 0.0));
        if (l16 == 0) ys[p * 32 + wid * 4 + g] = v;
      }
    }
    __syncthreads();
    if (t < NB) xs[b0 + t] = ys[t];
    __syncthreads();
    // r_j -= L[j][b0 : b0+128] x_b for the rows below, 32 rows per pass, 4 passes in flight
    for (int j0 = b0 + NB; j0 < n; j0 += 128) {
      Seg s[4];
      int row[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        row[p] = j0 + p * 32 + wid * 4 + g;
        s[p] = seg_load(L + (long)min(row[p], n - 1) * ldl + b0, l16);
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const double v = group16_sum(seg_partial(s[p], xs + b0, l16, 0.0));
        if (l16 == 0 && row[p] < n) xs[row[p]] -= v;
      }
    }
    __syncthreads();
  }
  for (int i = t; i < n; i += LEAF_THREADS) x[(long)i * incx] = xs[i];
}

// ---- backward leaf: x <- L^-T x ----------------------------------------------------------------
__global__ __launch_bounds__(LEAF_THREADS) void trsv_leaf_bwd_kernel(int n, const double* L, long ldl,
                                                                     const double* dinv, double* x,
                                                                     long incx) {
  __shared__ double xs[LEAF_MAX];
  __shared__ double part[LEAF_MAX];
  const int t = threadIdx.x;
  for (int i = t; i < LEAF_MAX; i += LEAF_THREADS) xs[i] = i < n ? x[(long)i * incx] : 0.0;
  __syncthreads();
  const int nblk = (n + NB - 1) / NB;
  for (int b = nblk - 1; b >= 0; --b) {
    const int b0 = b * NB;
    const int bw = min(NB, n - b0);
    // x_b = dinv_b^T r_b : thread (rg, cp) sums 16 rows of two columns, then 8 partials per column
    {
      const double* D = dinv + (size_t)b * NB * NB;
      const int cp = t & 63, rg = t >> 6;
      v2d acc = {0.0, 0.0};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = rg * 16 + i;
        const v2d a = *reinterpret_cast<const v2d*>(D + r * NB + 2 * cp);
        const double xv = xs[b0 + r];
        acc.x = fma(a.x, xv, acc.x);
        acc.y = fma(a.y, xv, acc.y);
      }
      part[rg * NB + 2 * cp] = acc.x;
      part[rg * NB + 2 * cp + 1] = acc.y;
    }
    __syncthreads();
    if (t < NB) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q) s += part[q * NB + t];
      xs[b0 + t] = s;
    }
    __syncthreads();
    // r_j -= sum_i L[b0+i][j] x_b[i] for the columns j < b0
    if (b0 > 0) {
      const int C = b0, pairs = C / 2;
      const int groups = b >= 8 ? 1 : 8 / b;          // row groups so that all 512 threads work
      const int rpg = (NB + groups - 1) / groups;
      for (int item = t; item < groups * pairs; item += LEAF_THREADS) {
        const int gq = item / pairs, p = item - gq * pairs;
        const int i1 = min(bw, (gq + 1) * rpg);
        v2d acc = {0.0, 0.0};
#pragma unroll 8
        for (int i = gq * rpg; i < i1; ++i) {
          const v2d a = *reinterpret_cast<const v2d*>(L + (long)(b0 + i) * ldl + 2 * p);
          const double xv = xs[b0 + i];
          acc.x = fma(a.x, xv, acc.x);
          acc.y = fma(a.y, xv, acc.y);
        }
        part[gq * C + 2 * p] = acc.x;
        part[gq * C + 2 * p + 1] = acc.y;
      }
      __syncthreads();
      for (int j = t; j < C; j += LEAF_THREADS) {
        double s = 0.0;
        for (int q = 0; q < groups; ++q) s += part[q * C + j];
        xs[j] -= s;
      }
      __syncthreads();
    }
  }
  for (int i = t; i < n; i += LEAF_THREADS) x[(long)i * incx] = xs[i];
}

// ---- x2 -= A x1, A n2 x n1 row-major (n1 a multiple of 128): 16 rows per workgroup -------------
__global__ __launch_bounds__(256) void trsv_gemvn_kernel(int n2, int n1, const double* A, long lda,
                                                         const double* x1, double* x2, long incx) {
  __shared__ double xs[XCHUNK];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = lane >> 4, l16 = lane & 15;
  const int r = blockIdx.x * 16 + wid * 4 + g;
  const double* row = A + (long)min(r, n2 - 1) * lda;
  double acc = 0.0;
  for (int c0 = 0; c0 < n1; c0 += XCHUNK) {
    const int cw = min(XCHUNK, n1 - c0);
    __syncthreads();
    for (int i = t; i < cw; i += 256) xs[i] = x1[(long)(c0 + i) * incx];
    __syncthreads();
    int p = 0;
    for (; p + 512 <= cw; p += 512) {
      Seg s[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] = seg_load(row + c0 + p + q * 128, l16);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = seg_partial(s[q], xs + p + q * 128, l16, acc);
    }
    for (; p < cw; p += 128) acc = seg_partial(seg_load(row + c0 + p, l16), xs + p, l16, acc);
  }
  acc = group16_sum(acc);
  if (l16 == 0 && r < n2) x2[(long)r * incx] -= acc;
}

// ---- x1 -= A^T x2: per row chunk partial column sums, then a fixed-order final pass ------------
__global__ __launch_bounds__(256) void trsv_gemvt_partial_kernel(int n2, int n1, const double* A, long lda,
                                                                 const double* x2, long incx, int rc,
                                                                 double* partials) {
  __shared__ double xs[XCHUNK];
  const int t = threadIdx.x;
  const int col = (blockIdx.x * 256 + t) * 2;
  const int i0 = blockIdx.y * rc, i1 = min(n2, i0 + rc);
  for (int i = t; i < i1 - i0; i += 256) xs[i] = x2[(long)(i0 + i) * incx];
  __syncthreads();
  if (col >= n1) return;
  const double* a = A + (long)i0 * lda + col;
  v2d acc = {0.0, 0.0};
#pragma unroll 8
  for (int i = 0; i < i1 - i0; ++i) {
    const v2d v = *reinterpret_cast<const v2d*>(a + (long)i * lda);
    const double xv = xs[i];
    acc.x = fma(v.x, xv, acc.x);
    acc.y = fma(v.y, xv, acc.y);
  }
  *reinterpret_cast<v2d*>(partials + (size_t)blockIdx.y * n1 + col) = acc;
}

__global__ __launch_bounds__(256) void trsv_gemvt_final_kernel(int nchunks, int n1, const double* partials,
                                                               double* x1, long incx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += partials[(size_t)c * n1 + j];
  x1[(long)j * incx] -= s;
}

int g_leaf = 0;  // rows per leaf; 0 = not yet initialised (GPX_TRSV_LEAF overrides, development only)

inline int leaf_rows() {
  if (g_leaf == 0) {
    g_leaf = 256;  // measured at n = 65536: 256 -> 7.3 / 11.0 ms (fwd / bwd), 512 -> 8.0 / 13.8, 1024 -> 10.1 / 22.2
    if (const char* e = getenv("GPX_TRSV_LEAF")) {
      const int v = atoi(e);
      if (v >= NB && v <= LEAF_MAX && v % NB == 0) g_leaf = v;
    }
  }
  return g_leaf;
}

inline int split_rows(int n) {
  int h = ((n / 2 + NB - 1) / NB) * NB;
  if (h >= n) h -= NB;
  return h;
}

int fwd(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* x, long incx) {
  if (n <= leaf_rows()) {
    hipLaunchKernelGGL(trsv_leaf_fwd_kernel, dim3(1), dim3(LEAF_THREADS), 0, s, n, L, ldl, dinv, x, incx);
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const int n1 = split_rows(n), n2 = n - n1;
  int rc = fwd(s, n1, L, ldl, dinv, x, incx);
  if (rc != 0) return rc;
  hipLaunchKernelGGL(trsv_gemvn_kernel, dim3((n2 + 15) / 16), dim3(256), 0, s, n2, n1, L + (long)n1 * ldl, ldl,
                     x, x + (long)n1 * incx, incx);
  GPX_HIP(hipGetLastError());
  return fwd(s, n2, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, x + (long)n1 * incx, incx);
}

int bwd(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* x, long incx,
        double* red, size_t red_elems) {
  if (n <= leaf_rows()) {
    hipLaunchKernelGGL(trsv_leaf_bwd_kernel, dim3(1), dim3(LEAF_THREADS), 0, s, n, L, ldl, dinv, x, incx);
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const int n1 = split_rows(n), n2 = n - n1;
  int rc = bwd(s, n2, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, x + (long)n1 * incx,
               incx, red, red_elems);
  if (rc != 0) return rc;
  // rows per chunk: as small as the partials buffer allows (more workgroups), at most XCHUNK
  int rows = NB;
  while ((size_t)((n2 + rows - 1) / rows) * n1 > red_elems && rows < XCHUNK) rows *= 2;
  const int nchunks = (n2 + rows - 1) / rows;
  if ((size_t)nchunks * n1 > red_elems) {
    gpx_set_error("trsv: reduction scratch too small for n=%d", n);
    return GPX_ENOMEM;
  }
  hipLaunchKernelGGL(trsv_gemvt_partial_kernel, dim3((n1 / 2 + 255) / 256, nchunks), dim3(256), 0, s, n2, n1,
                     L + (long)n1 * ldl, ldl, x + (long)n1 * incx, incx, rows, red);
  GPX_HIP(hipGetLastError());
  hipLaunchKernelGGL(trsv_gemvt_final_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, nchunks, n1, red, x, incx);
  GPX_HIP(hipGetLastError());
  return bwd(s, n1, L, ldl, dinv, x, incx, red, red_elems);
}

}  // namespace

int gpx_trsv_launch(hipStream_t s, int trans, int n, const double* L, long ldl, const double* dinv, double* x,
                    long incx, double* red, size_t red_elems) {
  if (n <= 0) return 0;
  return trans ? bwd(s, n, L, ldl, dinv, x, incx, red, red_elems) : fwd(s, n, L, ldl, dinv, x, incx);
}
