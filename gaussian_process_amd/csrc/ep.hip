// EP site-update sweep on the device (expectation_propagation.py:72-88).
//
// The reference visits the sites strictly in sequence; for site i it reads Sigma[i][i] and mu[i],
// matches moments against the probit likelihood (scalars), changes tau[i], nu[i], applies the rank-1
// update  Sigma += coef * s s^T  (s = column i of Sigma, coef = -dtau / (1 + dtau Sigma_ii)) and
// recomputes mu = Sigma nu: 16 n^2 bytes of HBM traffic per site, 16 n^3 per sweep.
//
// Here the rank-1 updates are DELAYED and applied EPB at a time by the fp64 MFMA GEMM.  Within a block
// of EPB sites Sigma itself is left alone; the columns s_0 .. s_{t-1} already used (Sbuf, n x EPB) and
// their scaled copies coef_j s_j (Tbuf) represent the current matrix,
//   Sigma_t = Sigma_0 + sum_{j<t} coef_j s_j s_j^T,
// so the column the next site needs is  s_t = Sigma_0[:, i] + Sbuf[:, :t] (coef o Sbuf[i, :t])  -- a
// skinny GEMV over n x t doubles (8 MB on average at n = 8192) instead of a pass over the whole matrix.
// mu is carried along exactly: with s the column just formed, dnu the change of nu[i],
//   mu_new = Sigma_new nu_new = mu + s (dnu + coef (mu_i + Sigma_ii dnu)),
// which is row-local.  One kernel per site does all of it: every workgroup repeats the scalar moment
// match (same inputs, same arithmetic, same result), forms its 32 rows of s_t, appends them to
// Sbuf / Tbuf and writes its rows of the state vectors.  The state (tau, nu, mu) ping-pongs between two
// buffers from site to site so that no workgroup reads an entry another one is writing.  After EPB
// sites (and at the end of the sweep)  Sigma(lower) += Tbuf Sbuf^T  runs on the matrix cores.
// Per site: n t doubles read (t < EPB) + O(n) -- the sweep is launch-latency bound, not HBM bound.
//
// Sigma is symmetric: only its LOWER triangle is read and kept up to date (the strict upper triangle
// goes stale; the per-sweep recompute rewrites the whole matrix).
// phi(z)/Phi(z) is evaluated as sqrt(2/pi) / erfcx(-z/sqrt 2): no cancellation for any z, equal to
// the reference's exp(-z^2/2 - log sqrt(2 pi) - log_ndtr(z)) up to rounding.
#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int EPB = GPX_EP_BLOCK;   // sites per delayed block = columns of Sbuf / Tbuf
constexpr int ROWS = 32;            // rows of the state per workgroup
constexpr int THREADS = 256;        // 8 threads per row
static_assert(EPB <= THREADS && EPB % 16 == 0, "one delayed column per thread; 16-wide row chunks");

struct SiteState {   // one parity of the ping-pong state
  double* tau;
  double* nu;
  double* mu;
};

// sum over the 8 consecutive lanes that share a row, fixed order
__device__ inline double reduce8(double v) {
  v += __shfl_xor(v, 4, 8);
  v += __shfl_xor(v, 2, 8);
  v += __shfl_xor(v, 1, 8);
  return v;
}

__global__ __launch_bounds__(THREADS) void ep_site_kernel(int n, int i, int t, const double* Sigma, long ld,
                                                          double* Sbuf, double* Tbuf, double* coef,
                                                          const double* y, SiteState in, SiteState out) {
  __shared__ double w[EPB];        // coef_j * s_j[i], zero for j >= t
  __shared__ double wred[THREADS / 64];
  __shared__ double sc[4];         // coef_t, gamma, tau_new, nu_new
  const int tid = threadIdx.x;
  const int tpad = (t + 15) & ~15;

  // ---- loads that do not depend on anything computed here go first, so their latency overlaps ----
  const int r = blockIdx.x * ROWS + (tid >> 3);
  const int sub = tid & 7;
  double sig_ii = 0.0, mu_i = 0.0, yi = 0.0, tau_i = 0.0, nu_i = 0.0;
  if (tid == 0) {
    sig_ii = Sigma[(long)i * ld + i];
    mu_i = in.mu[i];
    yi = y[i];
    tau_i = in.tau[i];
    nu_i = in.nu[i];
  }
  double base = 0.0, mu_r = 0.0, tau_r = 0.0, nu_r = 0.0;
  if (r < n && sub == 0) {
    base = r <= i ? Sigma[(long)i * ld + r] : Sigma[(long)r * ld + i];   // column i of the symmetric Sigma_0
    mu_r = in.mu[r];
    tau_r = in.tau[r];
    nu_r = in.nu[r];
  }

  // ---- scalars of site i (repeated identically in every workgroup) ----
  double part = 0.0;
  if (tid < EPB) {
    double wj = 0.0;
    if (tid < t) {
      const double sij = Sbuf[(long)i * EPB + tid];
      wj = coef[tid] * sij;
      part = wj * sij;
    }
    w[tid] = wj;
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  if ((tid & 63) == 0) wred[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    double corr = 0.0;
    for (int q = 0; q < THREADS / 64; ++q) corr += wred[q];
    const double sii = sig_ii + corr;
    const double tau_cav = 1.0 / sii - tau_i;
    const double nu_cav = mu_i / sii - nu_i;
    const double dom = tau_cav * tau_cav + tau_cav;
    const double rdom = sqrt(dom);
    const double z = yi * nu_cav / rdom;
    const double ratio = 0.7978845608028654 / erfcx(-z * 0.7071067811865476);   // phi(z) / Phi(z)
    const double mu_hat = nu_cav / tau_cav + yi * ratio / rdom;
    const double var_hat = 1.0 / tau_cav - ratio / dom * (z + ratio);
    const double dtau = 1.0 / var_hat - tau_cav - tau_i;
    const double nu_new = mu_hat / var_hat - nu_cav;
    const double dnu = nu_new - nu_i;
    const double cf = -dtau / (1.0 + dtau * sii);
    sc[0] = cf;
    sc[1] = dnu + cf * (mu_i + sii * dnu);
    sc[2] = tau_i + dtau;
    sc[3] = nu_new;
    if (blockIdx.x == 0) coef[t] = cf;
  }

  // ---- rows of the column s_t (needs only w, available since the barrier above) ----
  double acc = 0.0;
  if (r < n) {
    const double* srow = Sbuf + (long)r * EPB;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = 2 * sub;
    for (; j + 48 < tpad; j += 64) {   // four independent 16-byte loads in flight
      const v2d s0 = *reinterpret_cast<const v2d*>(srow + j);
      const v2d s1 = *reinterpret_cast<const v2d*>(srow + j + 16);
      const v2d s2 = *reinterpret_cast<const v2d*>(srow + j + 32);
      const v2d s3 = *reinterpret_cast<const v2d*>(srow + j + 48);
      a0 = fma(s0.y, w[j + 1], fma(s0.x, w[j], a0));
      a1 = fma(s1.y, w[j + 17], fma(s1.x, w[j + 16], a1));
      a2 = fma(s2.y, w[j + 33], fma(s2.x, w[j + 32], a2));
      a3 = fma(s3.y, w[j + 49], fma(s3.x, w[j + 48], a3));
    }
    for (; j < tpad; j += 16) {
      const v2d sv = *reinterpret_cast<const v2d*>(srow + j);
      a0 = fma(sv.y, w[j + 1], fma(sv.x, w[j], a0));
    }
    acc = (a0 + a1) + (a2 + a3);
  }
  acc = reduce8(acc);
  __syncthreads();   // sc[] from thread 0
  if (r < n && sub == 0) {
    const double col = base + acc;
    Sbuf[(long)r * EPB + t] = col;
    Tbuf[(long)r * EPB + t] = sc[0] * col;
    out.mu[r] = fma(sc[1], col, mu_r);
    out.tau[r] = r == i ? sc[2] : tau_r;
    out.nu[r] = r == i ? sc[3] : nu_r;
  }
}

// out = Sigma x for a symmetric Sigma given by its lower triangle; 32 rows per workgroup.
__global__ __launch_bounds__(THREADS) void ep_symv_lower_kernel(int n, const double* Sigma, long ld, const double* x,
                                                                double* out) {
  __shared__ double part[8][ROWS + 1];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * ROWS;
  {   // below the diagonal: column r of rows c > r, 32 consecutive columns per row read
    const int rl = tid & 31, grp = tid >> 5;
    const int r = r0 + rl;
    double acc = 0.0;
    if (r < n)
      for (int c = r0 + grp; c < n; c += 8)
        if (c > r) acc = fma(Sigma[(long)c * ld + r], x[c], acc);
    part[grp][rl] = acc;
  }
  __syncthreads();
  const int rl = tid >> 3, sub = tid & 7;
  const int r = r0 + rl;
  double acc = 0.0;
  if (r < n) {
    const double* row = Sigma + (long)r * ld;
    for (int c = sub; c <= r; c += 8) acc = fma(row[c], x[c], acc);
  }
  acc = reduce8(acc);
  if (r < n && sub == 0) {
    double below = 0.0;
    for (int g = 0; g < 8; ++g) below += part[g][rl];
    out[r] = acc + below;
  }
}

__global__ void ep_copy2_kernel(int n, const double* a, const double* b, double* oa, double* ob) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    oa[k] = a[k];
    ob[k] = b[k];
  }
}

}  // namespace

size_t gpx_ep_sweep_work_elems_impl(int n) {
  const size_t ne = ((size_t)n + 1) & ~(size_t)1;
  return 2 * (size_t)n * EPB + 6 * ne + EPB;
}

int gpx_ep_sweep_launch(hipStream_t s, int n, double* Sigma, long ld, const double* y, double* tau, double* nu,
                        const int* order_host, int count, double* work) {
  for (int k = 0; k < count; ++k)
    if (order_host[k] < 0 || order_host[k] >= n) {
      gpx_set_error("ep_sweep: site index %d out of range at position %d", order_host[k], k);
      return GPX_EINVAL;
    }
  const size_t ne = ((size_t)n + 1) & ~(size_t)1;
  double* Sbuf = work;
  double* Tbuf = Sbuf + (size_t)n * EPB;
  double* stv = Tbuf + (size_t)n * EPB;
  double* coef = stv + 6 * ne;
  SiteState st[2] = {{stv, stv + ne, stv + 2 * ne}, {stv + 3 * ne, stv + 4 * ne, stv + 5 * ne}};
  const int grid = (n + ROWS - 1) / ROWS;
  const int g1 = (n + 255) / 256;

  hipLaunchKernelGGL(ep_copy2_kernel, dim3(g1), dim3(256), 0, s, n, tau, nu, st[0].tau, st[0].nu);
  hipLaunchKernelGGL(ep_symv_lower_kernel, dim3(grid), dim3(THREADS), 0, s, n, Sigma, ld, nu, st[0].mu);

  GemmDesc g{};
  g.A = Tbuf; g.lda = EPB; g.a_lay = GPX_LAY_KC; g.a_tri = 0;
  g.B = Sbuf; g.ldb = EPB; g.b_lay = GPX_LAY_KC; g.b_tri = 0;
  g.Cin = Sigma; g.ldcin = ld; g.D = Sigma; g.ldd = ld;
  g.M = n; g.N = n; g.K = EPB; g.alpha = 1.0; g.beta = 1.0;
  g.c_lower = 1; g.diag_add = 0.0; g.inplace = GPX_INPLACE_NONE;

  for (int k = 0; k < count; ++k) {
    const int t = k % EPB;
    if (t == 0) GPX_HIP(hipMemsetAsync(Sbuf, 0, 2 * (size_t)n * EPB * sizeof(double), s));   // Sbuf and Tbuf
    hipLaunchKernelGGL(ep_site_kernel, dim3(grid), dim3(THREADS), 0, s, n, order_host[k], t, Sigma, ld, Sbuf, Tbuf,
                       coef, y, st[k & 1], st[(k + 1) & 1]);
    if (t == EPB - 1 || k == count - 1) {
      g.K = (t + 16) & ~15;   // columns beyond t are zero
      const int rc = gpx_gemm_launch(s, g);
      if (rc != 0) return rc;
    }
  }
  const SiteState& fin = st[count & 1];
  hipLaunchKernelGGL(ep_copy2_kernel, dim3(g1), dim3(256), 0, s, n, fin.tau, fin.nu, tau, nu);
  GPX_HIP(hipGetLastError());
  return 0;
}
