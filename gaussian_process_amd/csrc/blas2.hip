// Memory-bound helpers: log-determinant, dot, GEMV, symmetrize, diagonal, EP scaling.
// Every reduction uses a fixed association order (per-thread strided partials, LDS tree), so
// results are bitwise reproducible run to run.
#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double block_sum_1024(double v, double* sm) {
  const int t = threadIdx.x;
  sm[t] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  const double r = sm[0];
  __syncthreads();
  return r;
}

// sum_i log L[i][i]  (regression.py:53)
__global__ __launch_bounds__(1024) void logdet_kernel(int n, const double* L, long ldl, double* out) {
  __shared__ double sm[1024];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) v += log(L[(long)i * ldl + i]);
  const double r = block_sum_1024(v, sm);
  if (threadIdx.x == 0) out[0] = r;
}

__global__ __launch_bounds__(1024) void dot_kernel(int n, const double* x, const double* y, double* out) {
  __shared__ double sm[1024];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) v = fma(x[i], y[i], v);
  const double r = block_sum_1024(v, sm);
  if (threadIdx.x == 0) out[0] = r;
}

// y[j] = sum_i A[i][j] x[i]   (A n x m row-major).  Stage 1: row chunks -> partials[chunk][m].
constexpr int GV_ROWS = 256;  // rows per chunk (doubled until the partials fit the scratch)
__global__ __launch_bounds__(256) void gemvT_partial_kernel(int n, int m, const double* A, long lda,
                                                             const double* x, double* partials, int rows) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * rows;
  const int i1 = min(n, i0 + rows);
  if (j >= m) return;
  double acc = 0.0;
  for (int i = i0; i < i1; ++i) acc = fma(A[(long)i * lda + j], x[i], acc);
  partials[(size_t)blockIdx.y * m + j] = acc;
}
__global__ __launch_bounds__(256) void gemvT_final_kernel(int nchunks, int m, const double* partials,
                                                           double alpha, double beta, double* y) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double acc = 0.0;
  for (int c = 0; c < nchunks; ++c) acc += partials[(size_t)c * m + j];
  y[j] = beta != 0.0 ? fma(beta, y[j], alpha * acc) : alpha * acc;
}

// y[i] = sum_j A[i][j] x[j]: one wave per row.
__global__ __launch_bounds__(256) void gemvN_kernel(int n, int m, const double* A, long lda,
                                                    const double* x, double alpha, double beta, double* y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  const double* a = A + (long)row * lda;
  double acc = 0.0;
  for (int j = lane; j < m; j += 64) acc = fma(a[j], x[j], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
  if (lane == 0) y[row] = beta != 0.0 ? fma(beta, y[row], alpha * acc) : alpha * acc;
}

// var[j] = base - sum_i V[i][j]^2  (diagonal of regression.py:41-44 / expectation_propagation.py:118)
__global__ __launch_bounds__(256) void colsumsq_partial_kernel(int n, int m, const double* V, long ldv,
                                                                double* partials, int rows) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * rows;
  const int i1 = min(n, i0 + rows);
  if (j >= m) return;
  double acc = 0.0;
  for (int i = i0; i < i1; ++i) {
    const double v = V[(long)i * ldv + j];
    acc = fma(v, v, acc);
  }
  partials[(size_t)blockIdx.y * m + j] = acc;
}
__global__ __launch_bounds__(256) void colsumsq_final_kernel(int nchunks, int m, const double* partials,
                                                              double base, double* var) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  double acc = 0.0;
  for (int c = 0; c < nchunks; ++c) acc += partials[(size_t)c * m + j];
  var[j] = base - acc;
}

// A[j][i] = A[i][j] for i > j, through a 64x64 LDS tile so both sides are coalesced.
__global__ __launch_bounds__(256) void symmetrize_kernel(int n, double* A, long lda) {
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bj > bi) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    const int i = bi * 64 + r, j = bj * 64 + tx;
    tile[r][tx] = (i < n && j < n) ? A[(long)i * lda + j] : 0.0;
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int i = bj * 64 + r, j = bi * 64 + tx;  // destination (upper side)
    if (i < n && j < n && j > i) A[(long)i * lda + j] = tile[tx][r];
  }
}

__global__ void diag_kernel(int n, const double* A, long lda, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = A[(long)i * lda + i];
}

__global__ void sqrt_kernel(int n, const double* x, double* y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = sqrt(x[i]);
}

// B = I + (s s^T) o K (lower triangle, zeros above inside diagonal blocks), SK = diag(s) K (full).
// expectation_propagation.py:62-63, 91-92.
__global__ __launch_bounds__(256) void ep_scale_kernel(int n, const double* K, long ldk, const double* sv,
                                                        double* B, long ldb, double* SK, long ldsk) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= n) return;
  const double k = K[(long)i * ldk + j];
  const double si = sv[i];
  SK[(long)i * ldsk + j] = si * k;
  if (j <= i) {
    B[(long)i * ldb + j] = (i == j ? 1.0 : 0.0) + si * k * sv[j];
  } else if ((j >> 7) == (i >> 7)) {
    B[(long)i * ldb + j] = 0.0;
  }
}

}  // namespace

// ---- the allocation-free variant's scratch (gpx_gemv_static) ------------------------------------
// Partials of the transposed GEMV live in this file-scope device array instead of the handle's
// hipMalloc'ed scratch.  Every GPU the library is loaded on holds its own copy (2 MiB) at its own
// address until the process ends; host code never uses the array's name as an address -- the per-device
// address comes from hipGetSymbolAddress at gpx_create and is what the kernels receive.
static __device__ __align__(256) double scratch[GPX_STATIC_SCRATCH_ELEMS];
static double* g_scratch_ptr[GPX_MAX_DEVICES] = {nullptr};

int gpx_static_scratch_init(int device) {
  if (device < 0 || device >= GPX_MAX_DEVICES) {
    gpx_set_error("static scratch: device %d outside [0, %d)", device, GPX_MAX_DEVICES);
    return GPX_EINVAL;
  }
  if (g_scratch_ptr[device] != nullptr) return 0;
  double* p = nullptr;   // the caller has made `device` current
  GPX_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
  if (p == nullptr) {
    gpx_set_error("static scratch: hipGetSymbolAddress returned a null address on device %d", device);
    return GPX_EHIP;
  }
  g_scratch_ptr[device] = p;
  return 0;
}

double* gpx_static_scratch(int device) {
  return (device >= 0 && device < GPX_MAX_DEVICES) ? g_scratch_ptr[device] : nullptr;
}

// rows per partial-sum chunk: GV_ROWS, doubled while chunks x m would overflow the scratch
static int chunk_rows(int n, int m, size_t red_elems) {
  int rows = GV_ROWS;
  while ((size_t)((n + rows - 1) / rows) * m > red_elems && rows < (1 << 20)) rows *= 2;
  return rows;
}

// do the partial sums of the transposed n x m product fit a scratch of red_elems doubles?
bool gpx_gemv_partials_fit(int n, int m, size_t red_elems) {
  if (n <= 0 || m <= 0) return true;
  const int rows = chunk_rows(n, m, red_elems);
  return (size_t)((n + rows - 1) / rows) * m <= red_elems;
}

int gpx_logdet_launch(hipStream_t s, int n, const double* L, long ldl, double*, double* out_dev) {
  hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(1024), 0, s, n, L, ldl, out_dev);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_dot_launch(hipStream_t s, int n, const double* x, const double* y, double*, double* out_dev) {
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, s, n, x, y, out_dev);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_gemv_launch(hipStream_t s, int trans, int n, int m, const double* A, long lda,
                    const double* x, double* y, double* red, size_t red_elems, double alpha, double beta) {
  if (n <= 0 || m <= 0) return 0;
  if (trans) {
    const int rows = chunk_rows(n, m, red_elems);
    const int nchunks = (n + rows - 1) / rows;
    if ((size_t)nchunks * m > red_elems) {
      gpx_set_error("gemv: reduction scratch too small (%d x %d)", n, m);
      return GPX_ENOMEM;
    }
    dim3 grid((m + 255) / 256, nchunks);
    hipLaunchKernelGGL(gemvT_partial_kernel, grid, dim3(256), 0, s, n, m, A, lda, x, red, rows);
    hipLaunchKernelGGL(gemvT_final_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nchunks, m, red, alpha, beta, y);
  } else {
    hipLaunchKernelGGL(gemvN_kernel, dim3((n + 3) / 4), dim3(256), 0, s, n, m, A, lda, x, alpha, beta, y);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_colsumsq_sub_launch(hipStream_t s, int n, int m, const double* V, long ldv, double base,
                            double* var, double* red, size_t red_elems) {
  if (m <= 0) return 0;
  const int rows = chunk_rows(n, m, red_elems);
  const int nchunks = n <= 0 ? 0 : (n + rows - 1) / rows;
  if ((size_t)nchunks * m > red_elems) {
    gpx_set_error("colsumsq: reduction scratch too small (%d x %d)", n, m);
    return GPX_ENOMEM;
  }
  dim3 grid((m + 255) / 256, nchunks);
  if (nchunks > 0) hipLaunchKernelGGL(colsumsq_partial_kernel, grid, dim3(256), 0, s, n, m, V, ldv, red, rows);
  hipLaunchKernelGGL(colsumsq_final_kernel, dim3((m + 255) / 256), dim3(256), 0, s, nchunks, m, red,
                     base, var);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_symmetrize_launch(hipStream_t s, int n, double* A, long lda) {
  if (n <= 0) return 0;
  const int nt = (n + 63) / 64;
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nt, nt), dim3(256), 0, s, n, A, lda);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_diag_launch(hipStream_t s, int n, const double* A, long lda, double* out) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(diag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, A, lda, out);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_sqrt_launch(hipStream_t s, int n, const double* x, double* y) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sqrt_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, x, y);
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_ep_scale_launch(hipStream_t s, int n, const double* K, long ldk, const double* sv, double* B,
                        long ldb, double* SK, long ldsk) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(ep_scale_kernel, dim3((n + 255) / 256, n), dim3(256), 0, s, n, K, ldk, sv, B, ldb,
                     SK, ldsk);
  GPX_HIP(hipGetLastError());
  return 0;
}
