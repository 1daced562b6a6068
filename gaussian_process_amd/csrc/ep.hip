// EP site-update sweep on the device (expectation_propagation.py:72-88).
//
// The reference visits the sites strictly in sequence; for site i it reads Sigma[i][i] and mu[i],
// matches moments against the probit likelihood (scalars), changes tau[i], nu[i], applies the rank-1
// update  Sigma -= dtau / (1 + dtau Sigma_ii) * s_i s_i^T  (s_i = column i of Sigma) and recomputes
// mu = Sigma nu.  Only mu[i] of the NEXT site is ever consumed before the per-sweep recompute
// (expectation_propagation.py:90-94) refreshes Sigma and mu from K, so here mu[i] = Sigma[i,:] . nu
// is evaluated on demand while row i is being copied: per site
//   site kernel   (1 workgroup)   row i -> colbuf, mu_i, the scalar moment match, tau[i], nu[i], coef
//   rank-1 kernel (n/8 workgroups) Sigma[r][:r] += coef * colbuf[r] * colbuf[:r]    2 * 4 n^2 bytes
// Sigma is symmetric: only its LOWER triangle is read and kept up to date during a sweep (the strict
// upper triangle goes stale; the per-sweep recompute rewrites the whole matrix).
// HBM-bound: 8 n^2 bytes per site (0.54 GB at n = 8192), nothing crosses to the host.
// phi(z)/Phi(z) is evaluated as sqrt(2/pi) / erfcx(-z/sqrt 2): no cancellation for any z, equal to
// the reference's exp(-z^2/2 - log sqrt(2 pi) - log_ndtr(z)) up to rounding.
#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int SITE_THREADS = 1024;
constexpr int R1_ROWS = 8;

__global__ __launch_bounds__(SITE_THREADS) void ep_site_kernel(int n, int i, const double* Sigma, long ld,
                                                               const double* y, double* tau, double* nu,
                                                               double* colbuf, double* scal) {
  __shared__ double sm[SITE_THREADS];
  const int t = threadIdx.x;
  const double* row = Sigma + (long)i * ld;
  double acc = 0.0;
  for (int c = t; c < n; c += SITE_THREADS) {
    // column i of the symmetric Sigma from its LOWER triangle: the row part is contiguous, the part
    // below the diagonal a strided gather (n - i elements; cheap next to the rank-1 update)
    const double v = c <= i ? row[c] : Sigma[(long)c * ld + i];
    colbuf[c] = v;
    acc = fma(v, nu[c], acc);
  }
  sm[t] = acc;
  __syncthreads();
  for (int s = SITE_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  if (t == 0) {
    const double mu_i = sm[0];
    const double sii = row[i];
    const double yi = y[i];
    const double tau_i = tau[i], nu_i = nu[i];
    const double tau_cav = 1.0 / sii - tau_i;
    const double nu_cav = mu_i / sii - nu_i;
    const double dom = tau_cav * tau_cav + tau_cav;
    const double rdom = sqrt(dom);
    const double z = yi * nu_cav / rdom;
    const double ratio = 0.7978845608028654 / erfcx(-z * 0.7071067811865476);   // phi(z) / Phi(z)
    const double mu_hat = nu_cav / tau_cav + yi * ratio / rdom;
    const double var_hat = 1.0 / tau_cav - ratio / dom * (z + ratio);
    const double dtau = 1.0 / var_hat - tau_cav - tau_i;
    tau[i] = tau_i + dtau;
    nu[i] = mu_hat / var_hat - nu_cav;
    scal[0] = -dtau / (1.0 + dtau * sii);
  }
}

// Sigma[r][c] += coef * colbuf[r] * colbuf[c] for c <= r (lower triangle), R1_ROWS rows per workgroup.
__global__ __launch_bounds__(256) void ep_rank1_kernel(int n, double* Sigma, long ld, const double* colbuf,
                                                       const double* scal) {
  const int t = threadIdx.x;
  const int r0 = blockIdx.x * R1_ROWS;
  const double coef = scal[0];
  double f[R1_ROWS];
#pragma unroll
  for (int q = 0; q < R1_ROWS; ++q) f[q] = r0 + q < n ? coef * colbuf[r0 + q] : 0.0;
  const int nrows = min(R1_ROWS, n - r0);
  const int cend = min(n, r0 + R1_ROWS);   // lower triangle only (plus the few upper entries of the diagonal 8x8 block)
  for (int c = 2 * t; c < cend; c += 512) {
    if (c + 1 < cend) {
      const v2d cv = *reinterpret_cast<const v2d*>(colbuf + c);
      v2d v[R1_ROWS];
#pragma unroll
      for (int q = 0; q < R1_ROWS; ++q)
        if (q < nrows) v[q] = *reinterpret_cast<const v2d*>(Sigma + (long)(r0 + q) * ld + c);
#pragma unroll
      for (int q = 0; q < R1_ROWS; ++q)
        if (q < nrows) {
          v[q].x = fma(f[q], cv.x, v[q].x);
          v[q].y = fma(f[q], cv.y, v[q].y);
          *reinterpret_cast<v2d*>(Sigma + (long)(r0 + q) * ld + c) = v[q];
        }
    } else {
      const double cv = colbuf[c];
      for (int q = 0; q < nrows; ++q) {
        double* p = Sigma + (long)(r0 + q) * ld + c;
        *p = fma(f[q], cv, *p);
      }
    }
  }
}

}  // namespace

int gpx_ep_sweep_launch(hipStream_t s, int n, double* Sigma, long ld, const double* y, double* tau, double* nu,
                        const int* order_host, int count, double* colbuf, double* scal) {
  const int grid = (n + R1_ROWS - 1) / R1_ROWS;
  for (int k = 0; k < count; ++k) {
    const int i = order_host[k];
    if (i < 0 || i >= n) {
      gpx_set_error("ep_sweep: site index %d out of range at position %d", i, k);
      return GPX_EINVAL;
    }
    hipLaunchKernelGGL(ep_site_kernel, dim3(1), dim3(SITE_THREADS), 0, s, n, i, Sigma, ld, y, tau, nu, colbuf, scal);
    hipLaunchKernelGGL(ep_rank1_kernel, dim3(grid), dim3(256), 0, s, n, Sigma, ld, colbuf, scal);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}
