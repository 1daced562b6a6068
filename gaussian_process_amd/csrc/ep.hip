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
constexpr int THREADS = 256;        // 8 threads per row: the four waves that own the workgroup's rows
constexpr int SITE_THREADS = THREADS + 64;   // ep_sites_kernel: a fifth wave for the serial moment matches
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

// MS consecutive sites per launch.  Every workgroup repeats, identically, the whole scalar recursion of
// the group: nothing it needs of the columns formed INSIDE the launch lies outside its reach.  With
// rows(q) = Sbuf[i_q, :t0] (the site rows of the columns already there), w_p = coef o rows(p),
//   G[q][p] = rows(q) . w_p                                   (MS x MS dots of length t0)
//   E[q][p] = s_p[i_q] = Sigma_0[i_q, i_p] + G[q][p] + sum_{p' < p} E[q][p'] coef_p' E[p][p']
// give, site after site, Sigma_ii = E[p][p] and mu_i = mu_in[i_p] + sum_{p' < p} gamma_p' E[p][p'],
// hence the moment match and coef_p, gamma_p; then each workgroup forms its 32 rows of all MS columns,
//   s_p[r] = Sigma_0[r, i_p] + Sbuf[r, :t0] . w_p + sum_{p' < p} s_p'[r] coef_p' E[p][p'],
// reading its rows of Sbuf ONCE for the MS columns.  One launch, one pass over Sbuf and one round of
// memory latency serve MS sites; what stays serial per site is the scalar moment match (~1 us).
// (One site per launch: 6.3 us per site, of which ~4 us were launch gap and dependent load latency.)
constexpr int MS = 8;               // sites per launch = threads per row (lane `sub` fetches site sub's base entry)

struct SiteGroup {
  int m;            // sites in this launch, 1 <= m <= MS, pairwise distinct
  int t0;           // columns of Sbuf / Tbuf already filled in this delayed block; the group fills [t0, t0 + m)
  int idx[MS];      // their site indices
};

__device__ __forceinline__ double sym_lower(const double* Sigma, long ld, int r, int c) {
  return r >= c ? Sigma[(long)r * ld + c] : Sigma[(long)c * ld + r];
}

// The serial part -- the m moment matches, ~1.7 us each of dependent divisions, sqrt and erfcx in one lane -- runs
// on a FIFTH wave while the four row waves form the dots of their rows with the columns already there
// (Sbuf[r, :t0] . w_p: the only pass over memory, independent of the moment matches); the two meet at one barrier
// and the row waves finish with the O(m^2) recursion per row.  (Round 2 ran the moment matches on lanes of wave 0
// with the other waves waiting at the barrier: 21 us per launch of 8 sites, 13.6 us of it the serial part.)
__global__ __launch_bounds__(SITE_THREADS) void ep_sites_kernel(int n, SiteGroup grp, const double* Sigma, long ld,
                                                           double* Sbuf, double* Tbuf, double* coef,
                                                           const double* y, SiteState in, SiteState out) {
  __shared__ double rowS[MS][EPB];   // Sbuf[i_q, j], zero for j >= t0
  __shared__ double wv[MS][EPB];     // coef_j * Sbuf[i_p, j]
  __shared__ double S0[MS][MS];      // Sigma_0[i_q, i_p]
  __shared__ double Gm[MS][MS];
  __shared__ double Em[MS][MS];      // E[q][p] (LDS, not a private array: dynamic indexing would go to scratch)
  __shared__ double cE[MS][MS];      // coef_p' * E[p][p'], p' < p
  __shared__ double stv[MS][4];      // mu, tau, nu, y at the sites
  __shared__ double scv[MS][4];      // coef_p, gamma_p, tau_new, nu_new
  const int tid = threadIdx.x;
  const int m = grp.m, t0 = grp.t0;
  const int tpad = (t0 + 15) & ~15;

  // ---- phase 1: everything that depends only on the state before the launch ----
  const bool row_wave = tid < THREADS;           // waves 0-3 own rows; wave 4 runs the moment matches
  const int rl = tid >> 3, sub = tid & 7;
  const int r = row_wave ? blockIdx.x * ROWS + rl : n;
  double base = 0.0, mu_r = 0.0, tau_r = 0.0, nu_r = 0.0;
  if (r < n) {
    if (sub < m) base = sym_lower(Sigma, ld, r, grp.idx[sub]);   // lane `sub`: column i_sub of Sigma_0 at row r
    if (sub == 0) {
      mu_r = in.mu[r];
      tau_r = in.tau[r];
      nu_r = in.nu[r];
    }
  }
  for (int e = tid; e < MS * EPB; e += SITE_THREADS) {   // rows of the unused slots (q >= m) are zero
    const int q = e / EPB, j = e - q * EPB;
    double v = 0.0, c = 0.0;
    if (q < m && j < t0) {
      v = Sbuf[(long)grp.idx[q] * EPB + j];
      c = coef[j];
    }
    rowS[q][j] = v;
    wv[q][j] = c * v;
  }
  if (tid < m * MS) {
    const int q = tid / MS, p2 = tid % MS;
    if (p2 < m) S0[q][p2] = sym_lower(Sigma, ld, grp.idx[q], grp.idx[p2]);
  }
  if (tid < m) {
    const int i = grp.idx[tid];
    stv[tid][0] = in.mu[i];
    stv[tid][1] = in.tau[i];
    stv[tid][2] = in.nu[i];
    stv[tid][3] = y[i];
  }
  __syncthreads();
  if (row_wave) {   // G[q][p]: 4 threads per pair, 64 pairs
    const int pair = tid >> 2, part = tid & 3;
    const int q = pair / MS, p2 = pair % MS;
    double acc = 0.0;
    if (q < m && p2 < m)
      for (int j = part; j < tpad; j += 4) acc = fma(rowS[q][j], wv[p2][j], acc);
    acc += __shfl_xor(acc, 1, 4);
    acc += __shfl_xor(acc, 2, 4);
    if (part == 0) Gm[q][p2] = acc;
  }
  __syncthreads();

  // ---- phase 2: the sites one after the other (scalars; repeated identically in every workgroup) ----
  // Lanes 0 .. MS-1 of the FIFTH wave all run the moment match of site p (same inputs, same result); lane q > p
  // then forms its own entry E[q][p] of the new column, so the m - p - 1 entries cost one step, not m.
  // Reciprocals are shared (1 / Sigma_ii, 1 / sqrt(dom), 1 / tau_cav, 1 / var_hat): 6 divisions per site
  // instead of 11 on the serial path; the results differ from the division-by-division form by rounding.
  double dsum[MS];
#pragma unroll
  for (int p2 = 0; p2 < MS; ++p2) dsum[p2] = 0.0;
  if (row_wave) {
    // ---- phase 3a (row waves, concurrently with phase 2): dots of this workgroup's rows with the columns already there
    if (r < n) {
      const double* srow = Sbuf + (long)r * EPB;
      for (int j = 2 * sub; j < tpad; j += 16) {
        const v2d sv = *reinterpret_cast<const v2d*>(srow + j);
#pragma unroll
        for (int p2 = 0; p2 < MS; ++p2) dsum[p2] = fma(sv.y, wv[p2][j + 1], fma(sv.x, wv[p2][j], dsum[p2]));
      }
    }
#pragma unroll
    for (int p2 = 0; p2 < MS; ++p2) {
      dsum[p2] = reduce8(dsum[p2]);
      // lane 0 of the row collects site p2's base entry from lane p2
      const double b = __shfl(base, (tid & 56) + p2, 64);
      dsum[p2] += b;
    }
  } else if (tid - THREADS < MS) {
    const int q = tid - THREADS;
#pragma unroll 1
    for (int p2 = 0; p2 < m; ++p2) {
      double sii = S0[p2][p2] + Gm[p2][p2];
      double mu_i = stv[p2][0];
      for (int pp = 0; pp < p2; ++pp) {
        sii = fma(cE[p2][pp], Em[p2][pp], sii);
        mu_i = fma(scv[pp][1], Em[p2][pp], mu_i);
      }
      const double tau_i = stv[p2][1], nu_i = stv[p2][2], yi = stv[p2][3];
      const double inv_sii = 1.0 / sii;
      const double tau_cav = inv_sii - tau_i;
      const double nu_cav = mu_i * inv_sii - nu_i;
      const double dom = tau_cav * tau_cav + tau_cav;
      const double inv_rdom = 1.0 / sqrt(dom);
      const double z = yi * nu_cav * inv_rdom;
      const double ratio = 0.7978845608028654 / erfcx(-z * 0.7071067811865476);   // phi(z) / Phi(z)
      const double inv_tc = 1.0 / tau_cav;
      const double mu_hat = nu_cav * inv_tc + yi * ratio * inv_rdom;
      const double var_hat = inv_tc - ratio * (inv_rdom * inv_rdom) * (z + ratio);
      const double inv_var = 1.0 / var_hat;
      const double dtau = inv_var - tau_cav - tau_i;
      const double nu_new = mu_hat * inv_var - nu_cav;
      const double dnu = nu_new - nu_i;
      const double c = -dtau / (1.0 + dtau * sii);
      // entry of the new column at the row of the later site q (lane q), and its scaled copy
      double e = 0.0;
      if (q > p2 && q < m) {
        e = S0[q][p2] + Gm[q][p2];
        for (int pp = 0; pp < p2; ++pp) e = fma(Em[q][pp], cE[p2][pp], e);
      }
      if (q == 0) {
        scv[p2][0] = c;
        scv[p2][1] = dnu + c * (mu_i + sii * dnu);
        scv[p2][2] = tau_i + dtau;
        scv[p2][3] = nu_new;
        if (blockIdx.x == 0) coef[t0 + p2] = c;
      }
      if (q > p2 && q < m) {
        Em[q][p2] = e;
        cE[q][p2] = c * e;
      }
      // (one wave: its LDS writes are visible to its own later reads in program order)
    }
  }
  __syncthreads();

  // ---- phase 3b: this workgroup's rows of the m new columns (needs coef / gamma / E of phase 2) ----
  if (r < n && sub == 0) {
    double sc[MS];
    double mu_new = mu_r, tau_new = tau_r, nu_new = nu_r;
#pragma unroll
    for (int p2 = 0; p2 < MS; ++p2) {
      if (p2 < m) {
        double col = dsum[p2];
#pragma unroll
        for (int pp = 0; pp < MS; ++pp)
          if (pp < p2) col = fma(sc[pp], cE[p2][pp], col);
        sc[p2] = col;
        Sbuf[(long)r * EPB + t0 + p2] = col;
        Tbuf[(long)r * EPB + t0 + p2] = scv[p2][0] * col;
        mu_new = fma(scv[p2][1], col, mu_new);
        if (r == grp.idx[p2]) {
          tau_new = scv[p2][2];
          nu_new = scv[p2][3];
        }
      }
    }
    out.mu[r] = mu_new;
    out.tau[r] = tau_new;
    out.nu[r] = nu_new;
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

  int launches = 0;
  for (int k = 0; k < count;) {
    const int t = k % EPB;
    if (t == 0) GPX_HIP(hipMemsetAsync(Sbuf, 0, 2 * (size_t)n * EPB * sizeof(double), s));   // Sbuf and Tbuf
    // the next group: up to MS sites, inside this delayed block, pairwise distinct
    SiteGroup grp;
    grp.t0 = t;
    grp.m = 0;
    while (grp.m < MS && k + grp.m < count && t + grp.m < EPB) {
      const int i = order_host[k + grp.m];
      bool dup = false;
      for (int q = 0; q < grp.m; ++q) dup = dup || grp.idx[q] == i;
      if (dup) break;
      grp.idx[grp.m++] = i;
    }
    for (int q = grp.m; q < MS; ++q) grp.idx[q] = grp.idx[0];
    hipLaunchKernelGGL(ep_sites_kernel, dim3(grid), dim3(SITE_THREADS), 0, s, n, grp, Sigma, ld, Sbuf, Tbuf, coef, y,
                       st[launches & 1], st[(launches + 1) & 1]);
    ++launches;
    k += grp.m;
    if (k % EPB == 0 || k == count) {
      const int tl = (k - 1) % EPB;     // last column filled
      g.K = (tl + 16) & ~15;            // columns beyond it are zero
      const int rc = gpx_gemm_launch(s, g);
      if (rc != 0) return rc;
    }
  }
  const SiteState& fin = st[launches & 1];
  hipLaunchKernelGGL(ep_copy2_kernel, dim3(g1), dim3(256), 0, s, n, fin.tau, fin.nu, tau, nu);
  GPX_HIP(hipGetLastError());
  return 0;
}
