// Triangular solves with ONE right-hand side (HBM-bound): x <- L^-1 x and x <- L^-T x.
//
// These are the vector solves of regression.py:50 (a = L^-1 y), regression.py:61 (the second half
// of cho_solve) and expectation_propagation.py:63,92.  The factor is read exactly once: the
// recursion  [L11 0; L21 L22]  ->  solve(L11), x2 -= L21 x1 (GEMV over all CUs), solve(L22)
// sends every off-diagonal element through one streaming GEMV, and the diagonal leaves (256 rows)
// are solved by one workgroup from the 128x128 diagonal-block inverses gpx_potrf left in dinv (so
// the leaf is three small mat-vecs, no divisions, no scalar recurrence).
// The n / 256 dependent leaves are the critical chain (~9 us each, 2.4 ms at n = 65536) and come on
// top of the streaming part.  Running the streaming GEMVs on a second stream beside the chain was
// built and measured (blocks of 2048 rows, the GEMV of a block two blocks behind the chain): the
// leaves' loads then queue behind the saturated HBM, a leaf takes 21 us instead of 10, and the solve
// got slower (7.1 / 7.4 ms against 6.6 / 6.9 ms one after the other) -- not kept.
// Algorithmic bytes: 4 n^2 (lower triangle of L) + 64 KiB per diagonal block of dinv.
// All reductions have a fixed association order: results are bitwise reproducible.
#include <cstdlib>

#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int NB = GPX_NB;
constexpr int LEAF_ROWS = 256;      // rows of a leaf (two 128-blocks; 512 -> 8.0 / 13.8 ms, 1024 -> 10.1 / 22.2 ms at n = 65536 before the prefetching leaf)
constexpr int LEAF_THREADS = 512;   // 8 waves
constexpr int XCHUNK = 2048;        // columns of x staged per pass in the GEMV

// ---- 16 lanes per row of a 128-column segment; a lane owns the four 16-byte pieces at columns
// 2 l16 + 32 q, q = 0..3, so that ONE load instruction of the wave reads 256 contiguous bytes of each
// of its 4 rows (whole 128-byte lines; with 8 consecutive columns per lane an instruction touched
// every line of the segment for a quarter of its bytes, and the leaf ran at 21 us instead of 8).
// seg_partial = the lane's share of the dot product.
struct Seg { v2d a0, a1, a2, a3; };

__device__ __forceinline__ Seg seg_load(const double* row128, int l16) {
  const v2d* p = reinterpret_cast<const v2d*>(row128 + l16 * 2);
  Seg s;
  s.a0 = p[0]; s.a1 = p[16]; s.a2 = p[32]; s.a3 = p[48];
  return s;
}

__device__ __forceinline__ double seg_partial(const Seg& s, const double* x128, int l16, double acc) {
  const double* x = x128 + l16 * 2;
  acc = fma(s.a0.x, x[0], acc);
  acc = fma(s.a0.y, x[1], acc);
  acc = fma(s.a1.x, x[32], acc);
  acc = fma(s.a1.y, x[33], acc);
  acc = fma(s.a2.x, x[64], acc);
  acc = fma(s.a2.y, x[65], acc);
  acc = fma(s.a3.x, x[96], acc);
  acc = fma(s.a3.y, x[97], acc);
  return acc;
}

// sum over the 16 lanes of a row group (every lane of the group gets the total): the 16 lanes are one
// DPP row, so each step is a row rotation folded into two v_mov_dpp (no LDS crossbar round trip as
// with __shfl_xor; the forward leaf runs 12 of these reductions on its critical path)
template <int CTRL>
__device__ __forceinline__ double dpp_rot(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ double group16_sum(double v) {
  v += dpp_rot<0x128>(v);   // row_ror:8
  v += dpp_rot<0x124>(v);   // row_ror:4
  v += dpp_rot<0x122>(v);   // row_ror:2
  v += dpp_rot<0x121>(v);   // row_ror:1
  return v;
}

// ---- leaves: n <= 256 rows (at most two 128-blocks), one workgroup of 512 threads ----------------
// Everything a leaf reads from HBM -- dinv_0, the off-diagonal block L10 and dinv_1, 384 KiB -- is
// requested before the first use (96 doubles per thread in registers), so the leaf pays ONE memory
// latency instead of one per dependent step (forward 12.5 -> 9.4 us, backward 16 -> 8.4 us per leaf; 256
// of them form the critical chain of a solve at n = 65536).

// forward: x <- L^-1 x.  Lane (wave wid, group g, l16) owns the Seg columns (see above) of rows
// p * 32 + wid * 4 + g, p = 0..3, of each 128 x 128 block.
__global__ __launch_bounds__(LEAF_THREADS) void trsv_leaf_fwd_kernel(int n, const double* L, long ldl,
                                                                     const double* dinv, double* x,
                                                                     long incx) {
  __shared__ double xs[LEAF_ROWS];   // right-hand side
  __shared__ double xo[LEAF_ROWS];   // solution
  __shared__ double rs[NB];          // updated right-hand side of the second block
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = lane >> 4, l16 = lane & 15;
  const bool two = n > NB;
  Seg d0[4], l10[4], d1[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) d0[p] = seg_load(dinv + (p * 32 + wid * 4 + g) * NB, l16);
  if (two) {
#pragma unroll
    for (int p = 0; p < 4; ++p) l10[p] = seg_load(L + (long)min(NB + p * 32 + wid * 4 + g, n - 1) * ldl, l16);
#pragma unroll
    for (int p = 0; p < 4; ++p) d1[p] = seg_load(dinv + (size_t)NB * NB + (p * 32 + wid * 4 + g) * NB, l16);
  }
  if (t < LEAF_ROWS) xs[t] = t < n ? x[(long)t * incx] : 0.0;
  __syncthreads();
  // x_0 = dinv_0 r_0 (dinv blocks are full 128 x 128, identity-padded past n)
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const double v = group16_sum(seg_partial(d0[p], xs, l16, 0.0));
    if (l16 == 0) xo[p * 32 + wid * 4 + g] = v;
  }
  __syncthreads();
  if (two) {
    // r_1 -= L10 x_0, then x_1 = dinv_1 r_1
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = p * 32 + wid * 4 + g;
      const double v = group16_sum(seg_partial(l10[p], xo, l16, 0.0));
      if (l16 == 0) rs[r] = NB + r < n ? xs[NB + r] - v : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const double v = group16_sum(seg_partial(d1[p], rs, l16, 0.0));
      if (l16 == 0) xo[NB + p * 32 + wid * 4 + g] = v;
    }
    __syncthreads();
  }
  if (t < n) x[(long)t * incx] = xo[t];
}

// backward: x <- L^-T x.  Thread (rg = t / 64, cp = t % 64) owns rows rg * 16 .. + 15 of columns
// 2 cp, 2 cp + 1 of each block (a wave reads 1 KiB of one row per load); y = M^T v is 8 partial sums
// per column through LDS.
struct Blk16 { v2d a[16]; };

__device__ __forceinline__ Blk16 blk_load(const double* M, long ld, int rg, int cp, int row_max) {
  Blk16 b;
#pragma unroll
  for (int i = 0; i < 16; ++i)
    b.a[i] = *reinterpret_cast<const v2d*>(M + (long)min(rg * 16 + i, row_max) * ld + 2 * cp);
  return b;
}

// part[rg][0..127] <- this thread's share of M^T v (v in LDS, 128 entries)
__device__ __forceinline__ void blk_tmatvec(const Blk16& b, const double* v, double* part, int rg, int cp) {
  v2d acc = {0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const double xv = v[rg * 16 + i];
    acc.x = fma(b.a[i].x, xv, acc.x);
    acc.y = fma(b.a[i].y, xv, acc.y);
  }
  part[rg * NB + 2 * cp] = acc.x;
  part[rg * NB + 2 * cp + 1] = acc.y;
}

__device__ __forceinline__ double part_sum(const double* part, int j) {
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += part[q * NB + j];
  return s;
}

__global__ __launch_bounds__(LEAF_THREADS) void trsv_leaf_bwd_kernel(int n, const double* L, long ldl,
                                                                     const double* dinv, double* x,
                                                                     long incx) {
  __shared__ double xs[LEAF_ROWS];
  __shared__ double xo[LEAF_ROWS];
  __shared__ double part[8 * NB];
  const int t = threadIdx.x, cp = t & 63, rg = t >> 6;
  const bool two = n > NB;
  Blk16 d1, l10, d0;
  if (two) {
    d1 = blk_load(dinv + (size_t)NB * NB, NB, rg, cp, NB - 1);
    l10 = blk_load(L + (long)NB * ldl, ldl, rg, cp, n - 1 - NB);   // rows past n are clamped; their x is zero
  }
  d0 = blk_load(dinv, NB, rg, cp, NB - 1);
  if (t < LEAF_ROWS) xs[t] = t < n ? x[(long)t * incx] : 0.0;
  __syncthreads();
  if (two) {
    // x_1 = dinv_1^T r_1
    blk_tmatvec(d1, xs + NB, part, rg, cp);
    __syncthreads();
    if (t < NB) xo[NB + t] = part_sum(part, t);
    __syncthreads();
    // r_0 -= L10^T x_1
    blk_tmatvec(l10, xo + NB, part, rg, cp);
    __syncthreads();
    if (t < NB) xs[t] -= part_sum(part, t);
    __syncthreads();
  }
  // x_0 = dinv_0^T r_0
  blk_tmatvec(d0, xs, part, rg, cp);
  __syncthreads();
  if (t < NB) xo[t] = part_sum(part, t);
  __syncthreads();
  if (t < n) x[(long)t * incx] = xo[t];
}

// ---- x2 -= A x1, A n2 x n1 row-major (n1 a multiple of 128): 16 rows per workgroup -------------
__device__ __forceinline__ void gemvn_body(double* xs, int blk, int n2, int n1, const double* A, long lda,
                                           const double* x1, double* x2, long incx) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = lane >> 4, l16 = lane & 15;
  const int r = blk * 16 + wid * 4 + g;
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

__global__ __launch_bounds__(256) void trsv_gemvn_kernel(int n2, int n1, const double* A, long lda,
                                                         const double* x1, double* x2, long incx) {
  __shared__ double xs[XCHUNK];
  gemvn_body(xs, blockIdx.x, n2, n1, A, lda, x1, x2, incx);
}

// ---- x1 -= A^T x2 (A n2 x n1 row-major): column sums per row chunk, then a fixed-order final pass ----
// A workgroup owns a strip of 128 columns (a wave reads 1 KiB of one row per load) and a chunk of
// `rc` rows; its 4 waves take every 4th row, so a thread walks rc / 4 rows with the loads of 8 rows in
// flight (one thread per column pair walking all rc rows left the small GEMVs of the chain
// latency-bound: 18 us each).  The chunk's column sums go to partials[chunk][col].
// (Combining the chunks in the same launch -- last workgroup of a strip by ticket counter -- was
// measured and lost by 10x: a device-scope release / acquire on this multi-XCD part writes back and
// invalidates the XCD's L2 for every workgroup; the kernel boundary does it once.)
__global__ __launch_bounds__(256) void trsv_gemvt_partial_kernel(int n2, int n1, const double* A, long lda,
                                                                 const double* x2, long incx, int rc,
                                                                 double* partials) {
  __shared__ double xs[XCHUNK];
  __shared__ v2d red[4][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int chunk = blockIdx.y;
  const int col = blockIdx.x * NB + 2 * lane;
  const int i0 = chunk * rc, i1 = min(n2, i0 + rc);
  for (int i = t; i < i1 - i0; i += 256) xs[i] = x2[(long)(i0 + i) * incx];
  __syncthreads();
  const int cc = min(col, n1 - 2);   // n1 is a multiple of 128 on every path; the clamp keeps the loads in range anyway
  const double* a = A + (long)i0 * lda + cc;
  v2d acc = {0.0, 0.0};
#pragma unroll 8
  for (int i = w; i < i1 - i0; i += 4) {
    const v2d v = *reinterpret_cast<const v2d*>(a + (long)i * lda);
    const double xv = xs[i];
    acc.x = fma(v.x, xv, acc.x);
    acc.y = fma(v.y, xv, acc.y);
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && col < n1) {
    v2d s = red[0][lane];
    s += red[1][lane];
    s += red[2][lane];
    s += red[3][lane];
    *reinterpret_cast<v2d*>(partials + (size_t)chunk * n1 + col) = s;
  }
}

__global__ __launch_bounds__(256) void trsv_gemvt_final_kernel(int nchunks, int n1, const double* partials,
                                                               double* x1, long incx) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n1) return;
  double s = 0.0;
#pragma unroll 4
  for (int c = 0; c < nchunks; ++c) s += partials[(size_t)c * n1 + j];
  x1[(long)j * incx] -= s;
}

inline int split_rows(int n) {
  int h = ((n / 2 + NB - 1) / NB) * NB;
  if (h >= n) h -= NB;
  return h;
}

int launch_gemvn(hipStream_t s, int n2, int n1, const double* A, long lda, const double* x1, double* x2, long incx) {
  hipLaunchKernelGGL(trsv_gemvn_kernel, dim3((n2 + 15) / 16), dim3(256), 0, s, n2, n1, A, lda, x1, x2, incx);
  GPX_HIP(hipGetLastError());
  return 0;
}

// x1 -= A^T x2 (A n2 x n1): column sums per row chunk into `red`, then the fixed-order final pass
int launch_gemvt(hipStream_t s, int n2, int n1, const double* A, long lda, const double* x2, double* x1, long incx,
                 double* red, size_t red_elems) {
  // rows per chunk: enough workgroups to pull the block at HBM speed (a chunk of 128 rows x 128 columns is
  // 128 KiB), shorter chunks when the block is small, longer ones when the partials would not fit
  int rows = n2 * (long)n1 <= 1024L * 1024 ? 64 : NB;
  while ((size_t)((n2 + rows - 1) / rows) * n1 > red_elems && rows < XCHUNK) rows *= 2;
  const int nchunks = (n2 + rows - 1) / rows;
  if ((size_t)nchunks * n1 > red_elems) {
    gpx_set_error("trsv: reduction scratch too small (%d x %d block)", n2, n1);
    return GPX_ENOMEM;
  }
  hipLaunchKernelGGL(trsv_gemvt_partial_kernel, dim3((n1 + NB - 1) / NB, nchunks), dim3(256), 0, s, n2, n1, A, lda,
                     x2, incx, rows, red);
  GPX_HIP(hipGetLastError());
  hipLaunchKernelGGL(trsv_gemvt_final_kernel, dim3((n1 + 255) / 256), dim3(256), 0, s, nchunks, n1, red, x1, incx);
  GPX_HIP(hipGetLastError());
  return 0;
}

int fwd(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* x, long incx) {
  if (n <= LEAF_ROWS) {
    hipLaunchKernelGGL(trsv_leaf_fwd_kernel, dim3(1), dim3(LEAF_THREADS), 0, s, n, L, ldl, dinv, x, incx);
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const int n1 = split_rows(n), n2 = n - n1;
  int rc = fwd(s, n1, L, ldl, dinv, x, incx);
  if (rc != 0) return rc;
  rc = launch_gemvn(s, n2, n1, L + (long)n1 * ldl, ldl, x, x + (long)n1 * incx, incx);
  if (rc != 0) return rc;
  return fwd(s, n2, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, x + (long)n1 * incx, incx);
}

int bwd(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* x, long incx,
        double* red, size_t red_elems) {
  if (n <= LEAF_ROWS) {
    hipLaunchKernelGGL(trsv_leaf_bwd_kernel, dim3(1), dim3(LEAF_THREADS), 0, s, n, L, ldl, dinv, x, incx);
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const int n1 = split_rows(n), n2 = n - n1;
  int rc = bwd(s, n2, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB, x + (long)n1 * incx,
               incx, red, red_elems);
  if (rc != 0) return rc;
  rc = launch_gemvt(s, n2, n1, L + (long)n1 * ldl, ldl, x + (long)n1 * incx, x, incx, red, red_elems);
  if (rc != 0) return rc;
  return bwd(s, n1, L, ldl, dinv, x, incx, red, red_elems);
}

// ---- triangular MULTIPLIES with one vector: y = W x and y = W^T x, W lower triangular -------------
// Once the gradient pass has W = L^-1 (gpx_trtri), a = L^-1 y and alpha = L^-T a (regression.py:50,61) are
// two streaming passes over W's triangle with no dependent chain at all: 2 x 3.4 ms at n = 65536 against
// 6.3 + 6.9 ms for the two solves, and at n = 4096 2 x 15 us against 2 x 220 us (16 dependent leaves each).
// What lies above the diagonal of W (scratch of the inversion, later Ky^-1) is never multiplied in: masked by
// select, so NaN / Inf there are harmless.

// y = W x: 16 rows per workgroup, 16 lanes per row; the long rows (bottom of the matrix) are dispatched first.
__device__ __forceinline__ void trmv_n_body(double* xs, int blk, int nblk, int n, const double* W, long ldw,
                                            const double* x, double* y) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, g = lane >> 4, l16 = lane & 15;
  const int r0 = (nblk - 1 - blk) * 16;
  const int r = r0 + wid * 4 + g;
  const double* row = W + (long)min(r, n - 1) * ldw;
  const int d0 = (r0 / NB) * NB;   // first column of the diagonal 128-segment, the same for the 16 rows
  double acc = 0.0;
  for (int c0 = 0; c0 <= d0; c0 += XCHUNK) {
    const int cw = min(XCHUNK, d0 + NB - c0);   // a multiple of 128
    __syncthreads();
    for (int i = t; i < cw; i += 256) xs[i] = c0 + i < n ? x[c0 + i] : 0.0;
    __syncthreads();
    int p = 0;
    for (; c0 + p + 512 <= d0 && p + 512 <= cw; p += 512) {
      Seg s[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] = seg_load(row + c0 + p + q * 128, l16);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = seg_partial(s[q], xs + p + q * 128, l16, acc);
    }
    for (; c0 + p + 128 <= d0 && p + 128 <= cw; p += 128) acc = seg_partial(seg_load(row + c0 + p, l16), xs + p, l16, acc);
    if (c0 + p == d0 && p < cw) {   // the diagonal segment: columns <= r only, pair loads kept inside the row
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = d0 + 2 * l16 + 32 * q;
        const v2d v = *reinterpret_cast<const v2d*>(row + (col < n ? col : 0));
        acc = fma(col <= r ? v.x : 0.0, xs[p + 2 * l16 + 32 * q], acc);
        acc = fma(col + 1 <= r ? v.y : 0.0, xs[p + 2 * l16 + 32 * q + 1], acc);
      }
    }
  }
  acc = group16_sum(acc);
  if (l16 == 0 && r < n) y[r] = acc;
}

__global__ __launch_bounds__(256) void trmv_lower_n_kernel(int n, const double* W, long ldw, const double* x,
                                                           double* y) {
  __shared__ double xs[XCHUNK];
  trmv_n_body(xs, (int)blockIdx.x, (int)gridDim.x, n, W, ldw, x, y);
}

// y = W^T x: column sums per row chunk (as trsv_gemvt_partial_kernel), rows >= column only; chunks that lie
// entirely above a strip's diagonal block write nothing and the final pass starts at the first one that does.
// (col0, pitch: the launch covers the columns [col0, col0 + pitch) and its partials are pitch wide -- one band of
// columns at a time when n / rc chunks of n partials do not fit the scratch, n > 131008)
__global__ __launch_bounds__(256) void trmv_lower_t_partial_kernel(int n, int col0, int pitch, const double* W, long ldw,
                                                                   const double* x, int rc, double* partials) {
  __shared__ double xs[XCHUNK];
  __shared__ v2d red[4][64];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int chunk = blockIdx.y, strip0 = col0 + blockIdx.x * NB;
  const int col = strip0 + 2 * lane;
  const int i0 = chunk * rc, i1 = min(n, i0 + rc);
  if (i1 <= strip0) return;   // uniform
  for (int i = t; i < i1 - i0; i += 256) xs[i] = x[i0 + i];
  __syncthreads();
  const int cc = min(col, (n - 1) & ~1);   // pair loads stay inside the row (ldw is even and >= n)
  const double* a = W + (long)i0 * ldw + cc;
  v2d acc = {0.0, 0.0};
  if (i0 >= strip0 + NB) {   // uniform: the whole chunk is below the strip's diagonal block
#pragma unroll 8
    for (int i = w; i < i1 - i0; i += 4) {
      const v2d v = *reinterpret_cast<const v2d*>(a + (long)i * ldw);
      const double xv = xs[i];
      acc.x = fma(v.x, xv, acc.x);
      acc.y = fma(v.y, xv, acc.y);
    }
  } else {
#pragma unroll 8
    for (int i = w; i < i1 - i0; i += 4) {
      const v2d v = *reinterpret_cast<const v2d*>(a + (long)i * ldw);
      const double xv = xs[i];
      acc.x = fma(col <= i0 + i ? v.x : 0.0, xv, acc.x);
      acc.y = fma(col + 1 <= i0 + i ? v.y : 0.0, xv, acc.y);
    }
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && col < n) {
    v2d s = red[0][lane];
    s += red[1][lane];
    s += red[2][lane];
    s += red[3][lane];
    *reinterpret_cast<v2d*>(partials + (size_t)chunk * pitch + (col - col0)) = s;
  }
}

__global__ __launch_bounds__(256) void trmv_lower_t_final_kernel(int n, int col0, int pitch, int nchunks, int rc,
                                                                 const double* partials, double* y) {
  const int j = col0 + blockIdx.x * 256 + threadIdx.x;
  if (j >= n || j >= col0 + pitch) return;
  double s = 0.0;
  for (int c = ((j / NB) * NB) / rc; c < nchunks; ++c) s += partials[(size_t)c * pitch + (j - col0)];   // fixed order
  y[j] = s;
}

}  // namespace

int gpx_trmv_lower_launch(hipStream_t s, int trans, int n, const double* W, long ldw, const double* x, double* y,
                          double* red, size_t red_elems) {
  if (n <= 0) return 0;
  if (!trans) {
    hipLaunchKernelGGL(trmv_lower_n_kernel, dim3((n + 15) / 16), dim3(256), 0, s, n, W, ldw, x, y);
    GPX_HIP(hipGetLastError());
    return 0;
  }
  const int npad = (n + NB - 1) / NB * NB;
  int rows = n <= 1024 ? 64 : NB;
  while ((size_t)((n + rows - 1) / rows) * npad > red_elems && rows < XCHUNK) rows *= 2;
  const int nchunks = (n + rows - 1) / rows;
  // a chunk holds at most XCHUNK rows (its piece of x sits in LDS): beyond n ~ 131 000 the partials of all columns
  // no longer fit the scratch at once, and the columns go band by band (the same kernels, the same sums)
  size_t band = red_elems / nchunks / NB * NB;
  if (band == 0) {
    gpx_set_error("trmv: reduction scratch too small (n = %d)", n);
    return GPX_ENOMEM;
  }
  static const long band_cap = [] {
    const char* e = getenv("GPX_TRMV_BAND_COLS");   // development/test override: columns per band (multiple of 128)
    return e ? atol(e) / NB * NB : 0L;
  }();
  if (band_cap > 0 && (size_t)band_cap < band) band = (size_t)band_cap;
  if (band > (size_t)npad) band = npad;
  for (int c0 = 0; c0 < n; c0 += (int)band) {
    const int bw = (int)band < npad - c0 ? (int)band : npad - c0;
    hipLaunchKernelGGL(trmv_lower_t_partial_kernel, dim3(bw / NB, nchunks), dim3(256), 0, s, n, c0, bw, W, ldw, x, rows,
                       red);
    GPX_HIP(hipGetLastError());
    hipLaunchKernelGGL(trmv_lower_t_final_kernel, dim3((bw + 255) / 256), dim3(256), 0, s, n, c0, bw, nchunks, rows, red, y);
    GPX_HIP(hipGetLastError());
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// The same solves through EXPLICIT INVERSES of the 1024 x 1024 diagonal blocks.
//
// The recursion above is a chain of 2 n / 256 - 1 dependent launches (511 at n = 65536), each with 7-9 us of
// fixed cost: 6.3 / 6.9 ms for a solve that streams 17 GB (2.7 / 2.5 TB/s).  With W_t = L_tt^-1 for the diagonal
// blocks of order B = 1024 a solve is n / B steps of TWO streaming launches each,
//     forward:   x_t = W_t r_t,      r_{s > t} -= L[s, t] x_t            (row-major GEMV over all CUs)
//     backward:  x_t = W_t^T r_t,    r_{s < t} -= L[t, s]^T x_t          (column sums of a 1024-row block:
//                                                                         one workgroup per strip of columns, no partials)
// i.e. 128 launches: a small one that completes the next block's right-hand side, and one that forms x_next
// (64 workgroups) TOGETHER with the update of the rest, so the latency-bound triangular multiply hides under the
// streaming part.  The inverses are formed level by level from the 128-block inverses in dinv, all blocks
// of a level in two batched MFMA GEMM launches (as trtri_levels in gpx_api.hip, here on copies of the diagonal
// blocks: L itself is not touched): n B^2 / 3 flops, 7 launches -- 0.5 ms at n = 65536.  The ragged tail
// (n mod 1024 rows) goes through the recursion.  Fixed association order everywhere: bitwise reproducible.
// ---------------------------------------------------------------------------------------------------
namespace {

constexpr int BB = GPX_TRSV_BLOCK;

// ws block t (ld BB) <- lower 128-blocks of L's diagonal block t, its diagonal 128-blocks <- dinv
__global__ __launch_bounds__(256) void binv_init_kernel(const double* L, long ldl, const double* dinv, double* ws) {
  const int t = blockIdx.z;
  const int bi = blockIdx.y / (BB / NB), bj = blockIdx.y % (BB / NB);
  if (bj > bi) return;
  const int e = blockIdx.x * 256 + threadIdx.x;   // 0 .. 128 * 64 - 1: (row, column pair)
  const int i = e >> 6, j = (e & 63) * 2;
  double* dst = ws + (size_t)t * BB * BB + (size_t)(bi * NB + i) * BB + bj * NB + j;
  v2d v;
  if (bi == bj) v = *reinterpret_cast<const v2d*>(dinv + ((size_t)t * (BB / NB) + bi) * NB * NB + i * NB + j);
  else v = *reinterpret_cast<const v2d*>(L + ((long)t * BB + bi * NB + i) * ldl + (long)t * BB + bj * NB + j);
  *reinterpret_cast<v2d*>(dst) = v;
}

// dst[i] = src[i * inc] (gather) or dst[i * inc] = src[i] (scatter)
__global__ void vec_copy_kernel(int n, const double* src, long sinc, double* dst, long dinc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(long)i * dinc] = src[(long)i * sinc];
}

// Column sums of a block of at most XCHUNK rows, one workgroup per strip of SW columns, complete in the workgroup:
//   mode 0:  y[c] -= sum_r A[r][c] v[r]                 (r < rows)
//   mode 1:  y[c]  = sum_{r >= c} A[r][c] v[r]          (A lower triangular rows x rows, cols == rows)
// SW = 128: a wave reads 1 KiB of one row per load, 4 row lanes; SW = 64: 512 bytes per row, 8 row lanes (twice as
// many workgroups for the narrow blocks near the top of the matrix); SW = 16: one 128-byte line per row, 32 row
// lanes -- 64 workgroups for a single 1024 x 1024 block (the small launch on the chain).  8 rows in flight per thread.
template <int SW>
__device__ __forceinline__ void strip_body(double* vs, v2d* red, int blk, int rows, int cols, const double* A, long lda,
                                           const double* v, double* y, int mode) {
  constexpr int CPW = SW / 2, RL = 256 / CPW;   // column pairs per strip, row lanes
  const int t = threadIdx.x, cp = t % CPW, rl = t / CPW;
  const int c0 = blk * SW, c = c0 + 2 * cp;
  for (int i = t; i < rows; i += 256) vs[i] = v[i];
  __syncthreads();
  const int cc = min(c, cols - 2);            // cols is a multiple of SW on every path; the clamp keeps loads in range
  const double* a = A + cc;
  v2d acc = {0.0, 0.0};
  if (!mode) {
#pragma unroll 8
    for (int r = rl; r < rows; r += RL) {
      const v2d q = *reinterpret_cast<const v2d*>(a + (long)r * lda);
      const double x = vs[r];
      acc.x = fma(q.x, x, acc.x);
      acc.y = fma(q.y, x, acc.y);
    }
  } else {
    const int rbeg = (c0 / RL) * RL;          // lower: rows above the strip's first column contribute nothing
#pragma unroll 8
    for (int r = rbeg + rl; r < rows; r += RL) {
      const v2d q = *reinterpret_cast<const v2d*>(a + (long)r * lda);
      const double x = vs[r];
      acc.x = fma(c <= r ? q.x : 0.0, x, acc.x);
      acc.y = fma(c + 1 <= r ? q.y : 0.0, x, acc.y);
    }
  }
  red[rl * CPW + cp] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) {
    v2d sum = red[cp];
#pragma unroll
    for (int q = 1; q < RL; ++q) sum += red[q * CPW + cp];
    if (mode) {
      y[c] = sum.x;
      y[c + 1] = sum.y;
    } else {
      y[c] -= sum.x;
      y[c + 1] -= sum.y;
    }
  }
}

template <int SW>
__global__ __launch_bounds__(256) void gemvt_strip_kernel(int rows, int cols, const double* A, long lda, const double* v,
                                                           double* y, int mode) {
  __shared__ double vs[XCHUNK];
  __shared__ v2d red[256];
  strip_body<SW>(vs, red, (int)blockIdx.x, rows, cols, A, lda, v, y, mode);
}

// One chain step of the forward solve in ONE launch: workgroups [0, nb_tri) form x_next = W_next r_next (the block
// whose right-hand side the previous, small launch has just completed), the others subtract L[rows below, cur] x_cur
// from the rest of the right-hand side -- the triangular multiply (latency-bound, 64 workgroups) hides under the
// streaming update instead of standing between two launches.
__global__ __launch_bounds__(256) void trsv_fwd_step_kernel(int nb_tri, const double* Wn, const double* r_next, double* sol_next,
                                                            int n2, const double* A, long lda, const double* sol_cur,
                                                            double* r_rest) {
  __shared__ double xs[XCHUNK];
  const int b = blockIdx.x;
  if (b < nb_tri) trmv_n_body(xs, b, nb_tri, BB, Wn, (long)BB, r_next, sol_next);
  else gemvn_body(xs, b - nb_tri, n2, BB, A, lda, sol_cur, r_rest, 1L);
}

// The mirror image for the backward solve: workgroups [0, nb_tri) form x_prev = W_prev^T r_prev, the others subtract
// L[cur, columns left of prev]^T x_cur from the rest.
template <int SW>
__global__ __launch_bounds__(256) void trsv_bwd_step_kernel(int nb_tri, const double* Wp, const double* r_prev, double* sol_prev,
                                                            int cols_rest, const double* A, long lda, const double* sol_cur,
                                                            double* r_rest) {
  __shared__ double vs[XCHUNK];
  __shared__ v2d red[256];
  const int b = blockIdx.x;
  if (b < nb_tri) strip_body<SW>(vs, red, b, BB, BB, Wp, (long)BB, r_prev, sol_prev, 1);
  else strip_body<SW>(vs, red, b - nb_tri, BB, cols_rest, A, lda, sol_cur, r_rest, 0);
}

}  // namespace

size_t gpx_trsv_blockinv_elems(int n) {
  if (n < BB) return 0;
  const size_t npad = ((size_t)n + 63) / 64 * 64;
  return (size_t)(n / BB) * BB * BB + 2 * npad;
}

int gpx_trsv_blockinv_prepare(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* ws) {
  const int T = n / BB;   // full blocks only
  if (T == 0) return 0;
  hipLaunchKernelGGL(binv_init_kernel, dim3(NB * 64 / 256, (BB / NB) * (BB / NB), T), dim3(256), 0, s, L, ldl, dinv, ws);
  GPX_HIP(hipGetLastError());
  // W21 = -W22 (L21 W11) for all 2 sz-blocks of a level (inner batch) of all T diagonal blocks (outer batch); T^T is
  // kept in the block that mirrors L21 above the diagonal (see trtri_rec in gpx_api.hip)
  for (int sz = NB; sz < BB; sz *= 2) {
    const int inner = BB / (2 * sz);
    const long si = 2L * sz * (BB + 1), so = (long)BB * BB;
    double* L21 = ws + (long)sz * BB;
    double* L22 = L21 + sz;
    double* Tt = ws + sz;
    GemmDesc g1;
    g1.A = ws; g1.lda = BB; g1.a_lay = GPX_LAY_RC; g1.a_tri = 1;
    g1.B = L21; g1.ldb = BB; g1.b_lay = GPX_LAY_KC; g1.b_tri = 0;
    g1.Cin = nullptr; g1.ldcin = 0; g1.D = Tt; g1.ldd = BB;
    g1.M = g1.N = g1.K = sz; g1.alpha = 1.0; g1.beta = 0.0;
    g1.c_lower = 0; g1.diag_add = 0.0; g1.inplace = 0;
    g1.batch = inner * T; g1.batch_inner = inner;
    g1.stride_a = g1.stride_b = g1.stride_d = si;
    g1.stride2_a = g1.stride2_b = g1.stride2_d = so;
    int rc = gpx_gemm_launch(s, g1);
    if (rc != 0) return rc;
    GemmDesc g2 = g1;
    g2.A = L22; g2.a_lay = GPX_LAY_KC;
    g2.B = Tt; g2.b_lay = GPX_LAY_KC;
    g2.D = L21; g2.alpha = -1.0;
    rc = gpx_gemm_launch(s, g2);
    if (rc != 0) return rc;
  }
  return 0;
}

int gpx_trsv_blockinv_solve(hipStream_t s, int trans, int n, const double* L, long ldl, const double* dinv,
                            double* ws, double* x, long incx, double* red, size_t red_elems) {
  const int T = n / BB, nt = n - T * BB;   // full blocks, rows of the ragged tail
  const size_t npad = ((size_t)n + 63) / 64 * 64;
  double* r = ws + (size_t)T * BB * BB;     // running right-hand side
  double* sol = r + npad;                   // solution
  hipLaunchKernelGGL(vec_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, x, incx, r, 1L);
  const double* dinv_tail = dinv + (size_t)T * (BB / NB) * NB * NB;
  auto Wt = [&](int t) { return ws + (size_t)t * BB * BB; };
  auto Lblk = [&](int bi, int bj) { return L + (long)bi * BB * ldl + (long)bj * BB; };
  if (!trans) {
    if (T > 0) hipLaunchKernelGGL(trmv_lower_n_kernel, dim3(BB / 16), dim3(256), 0, s, BB, Wt(0), (long)BB, r, sol);
    for (int t = 0; t < T; ++t) {
      double* sol_t = sol + (long)t * BB;
      if (t + 1 < T) {
        // small launch: the next block's right-hand side; then x_next together with the update of everything below it
        hipLaunchKernelGGL(trsv_gemvn_kernel, dim3(BB / 16), dim3(256), 0, s, BB, BB, Lblk(t + 1, t), ldl, sol_t,
                           r + (long)(t + 1) * BB, 1L);
        const int n2 = n - (t + 2) * BB;
        hipLaunchKernelGGL(trsv_fwd_step_kernel, dim3(BB / 16 + (n2 + 15) / 16), dim3(256), 0, s, BB / 16, Wt(t + 1),
                           r + (long)(t + 1) * BB, sol + (long)(t + 1) * BB, n2, Lblk(t + 2, t), ldl, sol_t,
                           r + (long)(t + 2) * BB);
      } else if (nt > 0) {
        const int rc = launch_gemvn(s, nt, BB, Lblk(T, t), ldl, sol_t, r + (long)T * BB, 1L);
        if (rc != 0) return rc;
      }
    }
    if (nt > 0) {
      const int rc = fwd(s, nt, L + (long)T * BB * (ldl + 1), ldl, dinv_tail, r + (long)T * BB, 1L);
      if (rc != 0) return rc;
      hipLaunchKernelGGL(vec_copy_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, nt, r + (long)T * BB, 1L, sol + (long)T * BB, 1L);
    }
  } else {
    if (nt > 0) {
      const int rc = bwd(s, nt, L + (long)T * BB * (ldl + 1), ldl, dinv_tail, r + (long)T * BB, 1L, red, red_elems);
      if (rc != 0) return rc;
      hipLaunchKernelGGL(vec_copy_kernel, dim3((nt + 255) / 256), dim3(256), 0, s, nt, r + (long)T * BB, 1L, sol + (long)T * BB, 1L);
      // r[0 : T B] -= L[tail rows, 0 : T B]^T x_tail
      if (T > 0)
        hipLaunchKernelGGL((gemvt_strip_kernel<64>), dim3(T * BB / 64), dim3(256), 0, s, nt, T * BB, L + (long)T * BB * ldl, ldl,
                           sol + (long)T * BB, r, 0);
    }
    if (T > 0)
      hipLaunchKernelGGL((gemvt_strip_kernel<16>), dim3(BB / 16), dim3(256), 0, s, BB, BB, Wt(T - 1), (long)BB,
                         r + (long)(T - 1) * BB, sol + (long)(T - 1) * BB, 1);
    for (int t = T - 1; t > 0; --t) {
      double* sol_t = sol + (long)t * BB;
      // small launch: the previous block's right-hand side; then x_prev together with the update of everything left of it
      // (16-column strips: 64 workgroups of 128 KiB each; with 64-column strips this launch took 16 us on 16 workgroups)
      hipLaunchKernelGGL((gemvt_strip_kernel<16>), dim3(BB / 16), dim3(256), 0, s, BB, BB, Lblk(t, t - 1), ldl, sol_t,
                         r + (long)(t - 1) * BB, 0);
      const int cols_rest = (t - 1) * BB;
      // wide strips once the update alone has enough workgroups for the chip, narrow ones near the top of the matrix
      if (cols_rest / 128 >= 256)
        hipLaunchKernelGGL((trsv_bwd_step_kernel<128>), dim3(BB / 128 + cols_rest / 128), dim3(256), 0, s, BB / 128, Wt(t - 1),
                           r + (long)(t - 1) * BB, sol + (long)(t - 1) * BB, cols_rest, Lblk(t, 0), ldl, sol_t, r);
      else
        hipLaunchKernelGGL((trsv_bwd_step_kernel<64>), dim3(BB / 64 + cols_rest / 64), dim3(256), 0, s, BB / 64, Wt(t - 1),
                           r + (long)(t - 1) * BB, sol + (long)(t - 1) * BB, cols_rest, Lblk(t, 0), ldl, sol_t, r);
    }
  }
  hipLaunchKernelGGL(vec_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, sol, 1L, x, incx);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_trsv_launch(hipStream_t s, int trans, int n, const double* L, long ldl, const double* dinv, double* x,
                    long incx, double* red, size_t red_elems) {
  if (n <= 0) return 0;
  return trans ? bwd(s, n, L, ldl, dinv, x, incx, red, red_elems) : fwd(s, n, L, ldl, dinv, x, incx);
}
