// C ABI of libgpx (see include/gpx.h): argument checking, the recursive blocked algorithms that
// turn Cholesky / triangular solves / triangular inverse into MFMA GEMMs plus 128x128 leaves,
// and the fused estimator paths.
#include <cmath>
#include <vector>

#include <cstdlib>

#include "gpx_internal.h"

static thread_local char g_err[512] = "";

void gpx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

namespace {

constexpr int NB = GPX_NB;

inline int split_point(int n) {
  // first half: a multiple of NB, about n/2 (power-of-two friendly)
  int h = ((n / 2 + NB - 1) / NB) * NB;
  if (h >= n) h -= NB;
  return h;
}

inline GemmDesc mk(const double* A, long lda, int a_lay, const double* B, long ldb, int b_lay,
                   const double* Cin, long ldcin, double* D, long ldd, int M, int N, int K,
                   double alpha, double beta) {
  GemmDesc g;
  g.A = A; g.lda = lda; g.a_lay = a_lay; g.a_tri = 0;
  g.B = B; g.ldb = ldb; g.b_lay = b_lay; g.b_tri = 0;
  g.Cin = Cin; g.ldcin = ldcin; g.D = D; g.ldd = ldd;
  g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
  g.c_lower = 0; g.diag_add = 0.0; g.inplace = 0;
  return g;
}

#define GPX_TRY(expr)        \
  do {                       \
    int rc_ = (expr);        \
    if (rc_ != 0) return rc_; \
  } while (0)

// X (M x n) <- X L^-T, L lower n x n (starting at block row `off` of the factor).  Right solve
// used on the Cholesky panels: X L^T = B  (the dtrsm inside dpotrf, regression.py:36).
int trsm_rlt(hipStream_t s, int M, int n, const double* L, long ldl, const double* dinv, double* X,
             long ldx) {
  if (M <= 0 || n <= 0) return 0;
  if (n <= NB) {
    // X <- X * dinv^T : in-place safe, each workgroup reads exactly the rows x 128 cols it writes
    GemmDesc g = mk(X, ldx, GPX_LAY_KC, dinv, NB, GPX_LAY_KC, nullptr, 0, X, ldx, M, n, n, 1.0, 0.0);
    g.inplace = GPX_INPLACE_A;
    return gpx_gemm_launch(s, g);
  }
  const int n1 = split_point(n), n2 = n - n1;
  GPX_TRY(trsm_rlt(s, M, n1, L, ldl, dinv, X, ldx));
  // X2 -= X1 * L21^T
  GemmDesc g = mk(X, ldx, GPX_LAY_KC, L + (long)n1 * ldl, ldl, GPX_LAY_KC, X + n1, ldx, X + n1, ldx, M,
                  n2, n1, -1.0, 1.0);
  GPX_TRY(gpx_gemm_launch(s, g));
  return trsm_rlt(s, M, n2, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB,
                  X + n1, ldx);
}

// B (n x m) <- L^-1 B
int trsm_lln(hipStream_t s, int n, int m, const double* L, long ldl, const double* dinv, double* B,
             long ldb) {
  if (n <= 0 || m <= 0) return 0;
  if (n <= NB) {
    GemmDesc g = mk(dinv, NB, GPX_LAY_KC, B, ldb, GPX_LAY_RC, nullptr, 0, B, ldb, n, m, n, 1.0, 0.0);
    g.inplace = GPX_INPLACE_B;
    return gpx_gemm_launch(s, g);
  }
  const int n1 = split_point(n), n2 = n - n1;
  GPX_TRY(trsm_lln(s, n1, m, L, ldl, dinv, B, ldb));
  // B2 -= L21 * B1
  GemmDesc g = mk(L + (long)n1 * ldl, ldl, GPX_LAY_KC, B, ldb, GPX_LAY_RC, B + (long)n1 * ldb, ldb,
                  B + (long)n1 * ldb, ldb, n2, m, n1, -1.0, 1.0);
  GPX_TRY(gpx_gemm_launch(s, g));
  return trsm_lln(s, n2, m, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB,
                  B + (long)n1 * ldb, ldb);
}

// B (n x m) <- L^-T B
int trsm_ltn(hipStream_t s, int n, int m, const double* L, long ldl, const double* dinv, double* B,
             long ldb) {
  if (n <= 0 || m <= 0) return 0;
  if (n <= NB) {
    // B <- dinv^T B : op(A) = dinv^T is stored [k][m] = dinv itself (RC)
    GemmDesc g = mk(dinv, NB, GPX_LAY_RC, B, ldb, GPX_LAY_RC, nullptr, 0, B, ldb, n, m, n, 1.0, 0.0);
    g.inplace = GPX_INPLACE_B;
    return gpx_gemm_launch(s, g);
  }
  const int n1 = split_point(n), n2 = n - n1;
  GPX_TRY(trsm_ltn(s, n2, m, L + (long)n1 * ldl + n1, ldl, dinv + (size_t)(n1 / NB) * NB * NB,
                   B + (long)n1 * ldb, ldb));
  // B1 -= L21^T * B2 : op(A) = L21^T (n1 x n2), stored as L21 [k = n2 rows][n1] -> RC
  GemmDesc g = mk(L + (long)n1 * ldl, ldl, GPX_LAY_RC, B + (long)n1 * ldb, ldb, GPX_LAY_RC, B, ldb, B,
                  ldb, n1, m, n2, -1.0, 1.0);
  GPX_TRY(gpx_gemm_launch(s, g));
  return trsm_ltn(s, n1, m, L, ldl, dinv, B, ldb);
}

// Right-looking Cholesky by 128-column panels for the small diagonal blocks the recursion ends in:
// leaf (factor + inverse of the 128 x 128 diagonal block), one in-place GEMM for the panel below it,
// one lower rank-128 update of the rest -- 3 dependent launches per panel where the recursion needs
// ~8 (its solves and updates split further); the block is latency-bound, not flop-bound.
// (A one-panel look-ahead on a second stream -- next leaf + solve under the rest of the update -- was
// measured and lost: potrf(4096) 2.33 -> 2.74 ms, the two cross-stream event waits per panel cost
// more than the 37 us leaf they hide.)
int potrf_panels(hipStream_t s, int n, double* A, long lda, double* dinv, int* info_dev, int off) {
  for (int j0 = 0; j0 < n; j0 += NB) {
    const int jb = n - j0 < NB ? n - j0 : NB;
    double* Ajj = A + (long)j0 * lda + j0;
    double* dj = dinv + (size_t)(j0 / NB) * NB * NB;
    GPX_TRY(gpx_leaf_potf2_inv(s, jb, Ajj, lda, dj, info_dev, off + j0));
    const int rem = n - j0 - jb;
    if (rem <= 0) break;
    double* A21 = Ajj + (long)jb * lda;
    double* A22 = A21 + jb;
    GPX_TRY(trsm_rlt(s, rem, jb, Ajj, lda, dj, A21, lda));
    GemmDesc g = mk(A21, lda, GPX_LAY_KC, A21, lda, GPX_LAY_KC, A22, lda, A22, lda, rem, rem, jb, -1.0, 1.0);
    g.c_lower = 1;
    GPX_TRY(gpx_gemm_launch(s, g));
  }
  return 0;
}

int potrf_panel_limit() {
  static const int lim = [] {
    const char* e = getenv("GPX_POTRF_PANELS");   // development override
    return e ? atoi(e) : 4096;
  }();
  return lim;
}

// recursive lower Cholesky; `off` = global index of A[0][0] (for info and the dinv block index)
int potrf_rec(hipStream_t s, int n, double* A, long lda, double* dinv, int* info_dev, int off) {
  if (n <= NB) return gpx_leaf_potf2_inv(s, n, A, lda, dinv, info_dev, off);
  if (n <= potrf_panel_limit()) return potrf_panels(s, n, A, lda, dinv, info_dev, off);
  const int n1 = split_point(n), n2 = n - n1;
  GPX_TRY(potrf_rec(s, n1, A, lda, dinv, info_dev, off));
  double* A21 = A + (long)n1 * lda;
  double* A22 = A21 + n1;
  GPX_TRY(trsm_rlt(s, n2, n1, A, lda, dinv, A21, lda));
  // A22 -= A21 A21^T (lower tiles only)
  GemmDesc g = mk(A21, lda, GPX_LAY_KC, A21, lda, GPX_LAY_KC, A22, lda, A22, lda, n2, n2, n1, -1.0, 1.0);
  g.c_lower = 1;
  GPX_TRY(gpx_gemm_launch(s, g));
  return potrf_rec(s, n2, A22, lda, dinv + (size_t)(n1 / NB) * NB * NB, info_dev, off + n1);
}

// L <- L^-1 in place (lower).  Scratch: the block strictly above the diagonal that mirrors L21 -- rows
// [0, n1) x columns [n1, n) of the same square, never referenced by a lower-triangular consumer -- holds
// T^T for T = L21 W11, so the inverse needs no second buffer (every recursion level uses the upper-right
// block of its own square; they are disjoint).
// The same inversion for a block whose order is a power of two times 128, level by level instead of node
// by node: the 128-block inverses go in with one copy launch, then for block size sz = 128, 256, ... ALL
// the 2 sz-blocks of the level form their off-diagonal part W21 = -W22 L21 W11 in two batched launches
// (grid.y = block).  Node by node an n = 4096 block costs 62 + 32 launches of ~10 us fixed cost each for
// 23 Gflop (1.4 ms); level by level 11 launches.
static const int TRTRI_LEVELS_MAX = [] {
  const char* e = getenv("GPX_TRTRI_LEVELS_MAX");   // development override; 0 = node by node everywhere
  return e ? atoi(e) : 4096;
}();

int trtri_levels(hipStream_t s, int n, double* L, long ldl, const double* dinv) {
  GPX_TRY(gpx_copy_dinv_to_diag_batched(s, n / NB, dinv, L, ldl));
  for (int sz = NB; sz < n; sz *= 2) {
    const int batch = n / (2 * sz);
    const long stride = 2L * sz * (ldl + 1);
    double* L21 = L + (long)sz * ldl;
    double* L22 = L21 + sz;
    double* Tt = L + sz;
    GemmDesc g1 = mk(L, ldl, GPX_LAY_RC, L21, ldl, GPX_LAY_KC, nullptr, 0, Tt, ldl, sz, sz, sz, 1.0, 0.0);
    g1.a_tri = 1;
    g1.batch = batch;
    g1.stride_a = g1.stride_b = g1.stride_d = stride;
    GPX_TRY(gpx_gemm_launch(s, g1));
    GemmDesc g2 = mk(L22, ldl, GPX_LAY_KC, Tt, ldl, GPX_LAY_KC, nullptr, 0, L21, ldl, sz, sz, sz, -1.0, 0.0);
    g2.a_tri = 1;
    g2.batch = batch;
    g2.stride_a = g2.stride_b = g2.stride_d = stride;
    GPX_TRY(gpx_gemm_launch(s, g2));
  }
  return 0;
}

int trtri_rec(hipStream_t s, int n, double* L, long ldl, const double* dinv) {
  if (n <= NB) return gpx_copy_dinv_to_diag(s, n, dinv, L, ldl);
  if (n <= TRTRI_LEVELS_MAX && n % NB == 0 && ((n / NB) & (n / NB - 1)) == 0) return trtri_levels(s, n, L, ldl, dinv);
  const int n1 = split_point(n), n2 = n - n1;
  double* L21 = L + (long)n1 * ldl;
  double* L22 = L21 + n1;
  double* Tt = L + n1;   // n1 x n2, leading dimension ldl
  GPX_TRY(trtri_rec(s, n1, L, ldl, dinv));
  GPX_TRY(trtri_rec(s, n2, L22, ldl, dinv + (size_t)(n1 / NB) * NB * NB));
  // T^T = W11^T L21^T: op(A) = W11^T stored as W11 [k][i] (RC, lower: valid i <= k), op(B) = L21^T stored as
  // L21 [j][k] (KC)
  GemmDesc g1 = mk(L, ldl, GPX_LAY_RC, L21, ldl, GPX_LAY_KC, nullptr, 0, Tt, ldl, n1, n2, n1, 1.0, 0.0);
  g1.a_tri = 1;
  GPX_TRY(gpx_gemm_launch(s, g1));
  // L21 = -W22 T: op(A) = W22 (KC, lower: valid k <= i), op(B) = T stored as T^T [j][k] (KC)
  GemmDesc g2 = mk(L22, ldl, GPX_LAY_KC, Tt, ldl, GPX_LAY_KC, nullptr, 0, L21, ldl, n2, n1, n2, -1.0, 0.0);
  g2.a_tri = 1;
  return gpx_gemm_launch(s, g2);
}

// strict upper triangle of W's buffer <- Ky^-1 = W^T W (mirror image), diagonal -> kdiag
int lauum_upper(hipStream_t s, int n, double* W, long ldw, double* kdiag) {
  GemmDesc g = mk(W, ldw, GPX_LAY_RC, W, ldw, GPX_LAY_RC, nullptr, 0, W, ldw, n, n, n, 1.0, 0.0);
  g.a_tri = g.b_tri = g.c_lower = g.c_upper = 1;
  g.diag_out = kdiag;
  return gpx_gemm_launch(s, g);
}

int check_mat(const char* what, const void* p, long ld, long mincols) {
  GPX_REQUIRE(p != nullptr, "%s: null pointer", what);
  GPX_REQUIRE(gpx_aligned16(p), "%s: pointer must be 16-byte aligned", what);
  GPX_REQUIRE((ld & 1) == 0, "%s: leading dimension %ld must be even", what, ld);
  GPX_REQUIRE(ld >= mincols, "%s: leading dimension %ld < %ld", what, ld, mincols);
  return 0;
}

int fill_beta(int kind, int d, const double* beta_host, BetaArg& b, int& nbeta) {
  GPX_REQUIRE(kind == GPX_KIND_RBF || kind == GPX_KIND_ARD, "unknown kernel kind %d", kind);
  GPX_REQUIRE(d >= 1 && d <= GPX_MAXD, "input dimension %d outside [1, %d]", d, GPX_MAXD);
  GPX_REQUIRE(beta_host != nullptr, "beta_host is null");
  nbeta = kind == GPX_KIND_ARD ? d : 1;
  for (int k = 0; k < GPX_MAXD; ++k) b.b[k] = k < d ? (kind == GPX_KIND_ARD ? beta_host[k] : beta_host[0]) : 0.0;
  for (int k = 0; k < d; ++k) GPX_REQUIRE(b.b[k] >= 0.0, "beta[%d] is negative", k);
  return 0;
}

// copy `count` doubles device -> host through the pinned staging buffer and synchronise
int fetch(gpx_handle* h, const double* dev, double* host, size_t count) {
  GPX_REQUIRE(count <= GPX_HOST_PIN, "fetch: too many scalars");
  GPX_HIP(hipMemcpyAsync(h->host_pin, dev, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  GPX_HIP(hipStreamSynchronize(h->stream));
  memcpy(host, h->host_pin, count * sizeof(double));
  return 0;
}

int fetch_info(gpx_handle* h, int* info) {
  GPX_HIP(hipMemcpyAsync(h->host_pin, h->info_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  GPX_HIP(hipStreamSynchronize(h->stream));
  memcpy(info, h->host_pin, sizeof(int));
  return 0;
}

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { enter(dev); }
  // an entry point working for handle h: its device current, its split-K workspace the thread's
  explicit DeviceGuard(const gpx_handle* h) {
    enter(h->device);
    gpx_gemm_set_workspace(h->gemm_ws, h->gemm_ws ? GPX_GEMM_WS_ELEMS : 0);
  }
  void enter(int dev) {
    (void)hipGetDevice(&prev);
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// scratch slots inside h->red (doubles): the tail is for scalars
constexpr size_t RED_SCALARS = 64;
inline double* red_scalars(gpx_handle* h) { return h->red + (GPX_RED_ELEMS - RED_SCALARS); }
constexpr size_t RED_USABLE = GPX_RED_ELEMS - RED_SCALARS;

// B <- L^-1 B / L^-T B: one right-hand side goes to the streaming TRSV (trsv.hip), more to the
// recursive MFMA TRSM.
// Block inverses for the single-right-hand-side solves of one factor (trsv.hip), in the handle's grow-only
// workspace.  Returns the workspace, or nullptr when n is below the threshold or the allocation fails (the
// recursive solve is used then).
double* trsv_prepare(gpx_handle* h, int n, const double* L, long ldl, const double* dinv) {
  static const long min_n = [] {
    const char* e = getenv("GPX_TRSV_BLOCKINV_MIN");   // development override; 0 disables the block-inverse path
    return e ? atol(e) : (long)GPX_TRSV_BLOCKINV_MIN;
  }();
  if (min_n <= 0 || n < min_n) return nullptr;
  const size_t need = gpx_trsv_blockinv_elems(n);
  if (need > h->trsv_ws_elems) {
    if (h->trsv_ws) {
      (void)hipStreamSynchronize(h->stream);   // earlier solves may still read the old buffer
      (void)hipFree(h->trsv_ws);
    }
    h->trsv_ws = nullptr;
    h->trsv_ws_elems = 0;
    if (hipMalloc(&h->trsv_ws, need * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      h->trsv_ws = nullptr;
      return nullptr;
    }
    h->trsv_ws_elems = need;
  }
  if (gpx_trsv_blockinv_prepare(h->stream, n, L, ldl, dinv, h->trsv_ws) != 0) return nullptr;
  return h->trsv_ws;
}

// one right-hand side with the block inverses already prepared for this factor (ws from trsv_prepare, may be null)
int trsv_prepared(gpx_handle* h, double* ws, int trans, int n, const double* L, long ldl, const double* dinv, double* x,
                  long incx) {
  if (ws) return gpx_trsv_blockinv_solve(h->stream, trans, n, L, ldl, dinv, ws, x, incx, h->red, RED_USABLE);
  return gpx_trsv_launch(h->stream, trans, n, L, ldl, dinv, x, incx, h->red, RED_USABLE);
}

int solve_lower(gpx_handle* h, int trans, int n, int m, const double* L, long ldl, const double* dinv,
                double* B, long ldb) {
  if (m == 1) return trsv_prepared(h, trsv_prepare(h, n, L, ldl, dinv), trans, n, L, ldl, dinv, B, ldb);
  return trans ? trsm_ltn(h->stream, n, m, L, ldl, dinv, B, ldb) : trsm_lln(h->stream, n, m, L, ldl, dinv, B, ldb);
}

}  // namespace

extern "C" {

const char* gpx_last_error(void) { return g_err; }
const char* gpx_version(void) { return "gpx 0.1 (gfx950, fp64 MFMA 16x16x4)"; }

size_t gpx_dinv_elems(int n) { return n <= 0 ? 0 : (size_t)((n + NB - 1) / NB) * NB * NB; }

int gpx_create(gpx_handle** out, int device) {
  GPX_REQUIRE(out != nullptr, "gpx_create: null output");
  int ndev = 0;
  GPX_HIP(hipGetDeviceCount(&ndev));
  GPX_REQUIRE(device >= 0 && device < ndev, "gpx_create: device %d not present (%d devices)", device, ndev);
  DeviceGuard dg(device);
  hipDeviceProp_t prop;
  GPX_HIP(hipGetDeviceProperties(&prop, device));
  GPX_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "gpx_create: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
  // this GPU's address of the file-scope scratch of gpx_gemv_static (no allocation: a symbol lookup)
  GPX_TRY(gpx_static_scratch_init(device));
  gpx_handle* h = new gpx_handle();
  h->device = device;
  if (hipMalloc(&h->red, GPX_RED_ELEMS * sizeof(double)) != hipSuccess ||
      hipMalloc(&h->info_dev, 256) != hipSuccess ||
      hipMalloc(&h->gemm_ws, GPX_GEMM_WS_ELEMS * sizeof(double)) != hipSuccess ||
      hipHostMalloc(&h->host_pin, GPX_HOST_PIN * sizeof(double)) != hipSuccess) {
    gpx_set_error("gpx_create: allocation failed");
    gpx_destroy(h);
    return GPX_ENOMEM;
  }
  *out = h;
  return 0;
}

int gpx_destroy(gpx_handle* h) {
  if (!h) return 0;
  DeviceGuard dg(h);
  if (h->red) (void)hipFree(h->red);
  if (h->info_dev) (void)hipFree(h->info_dev);
  if (h->gemm_ws) (void)hipFree(h->gemm_ws);
  if (h->trsv_ws) (void)hipFree(h->trsv_ws);
  if (h->host_pin) (void)hipHostFree(h->host_pin);
  gpx_gemm_set_workspace(nullptr, 0);   // the thread's split-K workspace pointed into this handle
  delete h;
  return 0;
}

int gpx_set_stream(gpx_handle* h, void* stream) {
  GPX_REQUIRE(h != nullptr, "null handle");
  h->stream = reinterpret_cast<hipStream_t>(stream);
  return 0;
}

int gpx_synchronize(gpx_handle* h) {
  GPX_REQUIRE(h != nullptr, "null handle");
  DeviceGuard dg(h);
  GPX_HIP(hipStreamSynchronize(h->stream));
  return 0;
}

int gpx_gram(gpx_handle* h, int kind, const double* X1, int n1, int ldx1, const double* X2, int n2,
             int ldx2, int d, double sigma, const double* beta_host, double diag_add, double* K,
             int ldk, int uplo) {
  GPX_REQUIRE(h != nullptr, "null handle");
  GPX_REQUIRE(n1 >= 0 && n2 >= 0, "gram: negative size");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  if (X2 == nullptr) { X2 = X1; n2 = n1; ldx2 = ldx1; }
  if (n1 == 0 || n2 == 0) return 0;
  GPX_REQUIRE(X1 != nullptr && ldx1 >= d && ldx2 >= d, "gram: bad X");
  GPX_TRY(check_mat("gram: K", K, ldk, n2));
  GPX_REQUIRE(uplo == GPX_FULL || (uplo == GPX_LOWER && n1 == n2), "gram: GPX_LOWER needs a square matrix");
  DeviceGuard dg(h);
  return gpx_gram_launch(h->stream, X1, n1, ldx1, X2, n2, ldx2, d, sigma, b, diag_add, K, ldk, uplo);
}

int gpx_gram_dtheta(gpx_handle* h, int kind, const double* X1, int n1, int ldx1, const double* X2,
                    int n2, int ldx2, int d, double sigma, const double* beta_host, double* K,
                    double* dsigma, double* dbeta, int ld, size_t plane) {
  GPX_REQUIRE(h != nullptr, "null handle");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  if (X2 == nullptr) { X2 = X1; n2 = n1; ldx2 = ldx1; }
  if (n1 <= 0 || n2 <= 0) return 0;
  GPX_REQUIRE(X1 && K && dsigma && dbeta, "gram_dtheta: null pointer");
  GPX_REQUIRE(ld >= n2 && ldx1 >= d && ldx2 >= d, "gram_dtheta: bad leading dimension");
  GPX_REQUIRE(nbeta == 1 || plane >= (size_t)(n1 - 1) * ld + n2, "gram_dtheta: plane stride too small");
  DeviceGuard dg(h);
  return gpx_gram_dtheta_launch(h->stream, X1, n1, ldx1, X2, n2, ldx2, d, nbeta, sigma, b, K, dsigma,
                                dbeta, ld, plane);
}

int gpx_potrf(gpx_handle* h, int n, double* A, int lda, double* dinv) {
  GPX_REQUIRE(h != nullptr, "null handle");
  GPX_REQUIRE(n >= 0, "potrf: negative n");
  if (n == 0) return 0;
  GPX_TRY(check_mat("potrf: A", A, lda, n));
  GPX_REQUIRE(dinv != nullptr && gpx_aligned16(dinv), "potrf: dinv null or misaligned");
  DeviceGuard dg(h);
  GPX_HIP(hipMemsetAsync(h->info_dev, 0, sizeof(int), h->stream));
  GPX_TRY(potrf_rec(h->stream, n, A, lda, dinv, h->info_dev, 0));
  int info = 0;
  GPX_TRY(fetch_info(h, &info));
  return info;
}

int gpx_potrf_async(gpx_handle* h, int n, double* A, int lda, double* dinv, int* info_dev, int info_offset) {
  GPX_REQUIRE(h != nullptr, "null handle");
  GPX_REQUIRE(n >= 0, "potrf_async: negative n");
  if (n == 0) return 0;
  GPX_TRY(check_mat("potrf_async: A", A, lda, n));
  GPX_REQUIRE(dinv != nullptr && gpx_aligned16(dinv) && info_dev != nullptr, "potrf_async: dinv/info null or misaligned");
  DeviceGuard dg(h);
  return potrf_rec(h->stream, n, A, lda, dinv, info_dev, info_offset);
}

int gpx_trsm_right(gpx_handle* h, int m, int n, const double* L, int ldl, const double* dinv, double* X,
                   int ldx) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (m <= 0 || n <= 0) return 0;
  GPX_TRY(check_mat("trsm_right: L", L, ldl, n));
  GPX_TRY(check_mat("trsm_right: X", X, ldx, n));
  GPX_REQUIRE(dinv != nullptr && gpx_aligned16(dinv), "trsm_right: dinv null or misaligned");
  DeviceGuard dg(h);
  return trsm_rlt(h->stream, m, n, L, ldl, dinv, X, ldx);
}

int gpx_diag_inv(gpx_handle* h, int n, const double* L, int ldl, double* dinv) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0) return 0;
  GPX_TRY(check_mat("diag_inv: L", L, ldl, n));
  GPX_REQUIRE(dinv != nullptr, "diag_inv: null dinv");
  DeviceGuard dg(h);
  return gpx_leaf_diag_inv_batched(h->stream, n, L, ldl, dinv);
}

int gpx_trsm(gpx_handle* h, int trans, int n, int nrhs, const double* L, int ldl,
             const double* dinv, double* B, int ldb) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0 || nrhs <= 0) return 0;
  GPX_TRY(check_mat("trsm: L", L, ldl, n));
  if (nrhs == 1) {   // one right-hand side: a strided vector (the TRSV kernels use scalar accesses)
    GPX_REQUIRE(B != nullptr && ldb >= 1, "trsm: bad right-hand side vector");
  } else {
    GPX_TRY(check_mat("trsm: B", B, ldb, nrhs));
  }
  GPX_REQUIRE(dinv != nullptr && gpx_aligned16(dinv), "trsm: dinv null or misaligned");
  DeviceGuard dg(h);
  return solve_lower(h, trans, n, nrhs, L, ldl, dinv, B, ldb);
}

int gpx_logdet(gpx_handle* h, int n, const double* L, int ldl, double* out_host) {
  GPX_REQUIRE(h != nullptr && out_host != nullptr, "logdet: null argument");
  if (n <= 0) { *out_host = 0.0; return 0; }
  DeviceGuard dg(h);
  double* sc = red_scalars(h);
  GPX_TRY(gpx_logdet_launch(h->stream, n, L, ldl, h->red, sc));
  return fetch(h, sc, out_host, 1);
}

int gpx_dot(gpx_handle* h, int n, const double* x, const double* y, double* out_host) {
  GPX_REQUIRE(h != nullptr && out_host != nullptr, "dot: null argument");
  if (n <= 0) { *out_host = 0.0; return 0; }
  DeviceGuard dg(h);
  double* sc = red_scalars(h);
  GPX_TRY(gpx_dot_launch(h->stream, n, x, y, h->red, sc));
  return fetch(h, sc, out_host, 1);
}

int gpx_trtri(gpx_handle* h, int n, double* L, int ldl, const double* dinv) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0) return 0;
  GPX_TRY(check_mat("trtri: L", L, ldl, n));
  GPX_REQUIRE(dinv != nullptr, "trtri: null dinv");
  DeviceGuard dg(h);
  return trtri_rec(h->stream, n, L, ldl, dinv);
}

int gpx_lauum_upper(gpx_handle* h, int n, double* W, int ldw, double* kdiag) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0) return 0;
  GPX_TRY(check_mat("lauum_upper: W", W, ldw, n));
  GPX_REQUIRE(kdiag != nullptr, "lauum_upper: null diagonal vector");
  DeviceGuard dg(h);
  return lauum_upper(h->stream, n, W, ldw, kdiag);
}

int gpx_lauum(gpx_handle* h, int n, const double* W, int ldw, double* Kinv, int ldk) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0) return 0;
  GPX_TRY(check_mat("lauum: W", W, ldw, n));
  GPX_TRY(check_mat("lauum: Kinv", Kinv, ldk, n));
  GPX_REQUIRE(W != Kinv, "lauum: out of place only");
  DeviceGuard dg(h);
  // Kinv = W^T W: op(A) = W^T stored as W [k][i] (RC, lower), op(B) = W stored [k][j] (RC, lower)
  GemmDesc g = mk(W, ldw, GPX_LAY_RC, W, ldw, GPX_LAY_RC, nullptr, 0, Kinv, ldk, n, n, n, 1.0, 0.0);
  g.a_tri = 1;
  g.b_tri = 1;
  g.c_lower = 1;
  return gpx_gemm_launch(h->stream, g);
}

int gpx_syrk_sub(gpx_handle* h, int n, int m, const double* V, int ldv, double* C, int ldc,
                 double diag_add, int uplo) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (m <= 0) return 0;
  GPX_TRY(check_mat("syrk_sub: C", C, ldc, m));
  if (n > 0) GPX_TRY(check_mat("syrk_sub: V", V, ldv, m));
  DeviceGuard dg(h);
  GemmDesc g = mk(V, ldv, GPX_LAY_RC, V, ldv, GPX_LAY_RC, C, ldc, C, ldc, m, m, n < 0 ? 0 : n, -1.0, 1.0);
  g.c_lower = (uplo == GPX_LOWER);
  g.diag_add = diag_add;
  if (n <= 0) { g.A = C; g.B = C; g.lda = g.ldb = ldc; }  // K = 0: operands unused but must be valid
  return gpx_gemm_launch(h->stream, g);
}

int gpx_gemm(gpx_handle* h, int transa, int transb, int M, int N, int K, double alpha,
             const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc) {
  GPX_REQUIRE(h != nullptr, "null handle");
  DeviceGuard dg(h);
  // row-major: A stored M x K (transa=0, k contiguous -> KC) or K x M (transa=1 -> RC);
  //            B stored K x N (transb=0 -> RC) or N x K (transb=1 -> KC).
  GemmDesc g = mk(A, lda, transa ? GPX_LAY_RC : GPX_LAY_KC, B, ldb, transb ? GPX_LAY_KC : GPX_LAY_RC, C,
                  ldc, C, ldc, M, N, K, alpha, beta);
  return gpx_gemm_launch(h->stream, g);
}

int gpx_gemm_colblocks(gpx_handle* h, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* D, int nb, int row_base, const long long* blk_off,
                       const int* blk_row0) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (M <= 0 || N <= 0) return 0;
  GPX_REQUIRE(A && B && D && blk_off && blk_row0, "gemm_colblocks: null argument");
  GPX_REQUIRE(nb >= NB && nb % NB == 0 && row_base >= 0 && row_base % NB == 0,
              "gemm_colblocks: block width %d and row base %d must be multiples of %d", nb, row_base, NB);
  GPX_REQUIRE(gpx_aligned16(D), "gemm_colblocks: D must be 16-byte aligned");
  DeviceGuard dg(h);
  // D_c -= / += A B^T per column block c: A stored M x K, B stored N x K (both k-contiguous)
  GemmDesc g = mk(A, lda, GPX_LAY_KC, B, ldb, GPX_LAY_KC, D, nb, D, nb, M, N, K, alpha, beta);
  if (beta == 0.0) g.beta = 0.0;
  g.cb_off = blk_off;
  g.cb_row0 = blk_row0;
  g.cb_nb = nb;
  g.cb_rowbase = row_base;
  return gpx_gemm_launch(h->stream, g);
}

int gpx_gemv(gpx_handle* h, int trans, int n, int m, const double* A, int lda, const double* x,
             double* y) {
  GPX_REQUIRE(h != nullptr && A && x && y, "gemv: null argument");
  GPX_REQUIRE(lda >= m, "gemv: lda too small");
  DeviceGuard dg(h);
  return gpx_gemv_launch(h->stream, trans, n, m, A, lda, x, y, h->red, RED_USABLE);
}

int gpx_gemv_acc(gpx_handle* h, int trans, int n, int m, double alpha, const double* A, int lda,
                 const double* x, double beta, double* y) {
  GPX_REQUIRE(h != nullptr && A && x && y, "gemv_acc: null argument");
  GPX_REQUIRE(lda >= m, "gemv_acc: lda too small");
  GPX_REQUIRE(x != y, "gemv_acc: x and y must not alias");
  DeviceGuard dg(h);
  return gpx_gemv_launch(h->stream, trans, n, m, A, lda, x, y, h->red, RED_USABLE, alpha, beta);
}

int gpx_gemv_static(gpx_handle* h, int trans, int n, int m, const double* A, int lda, const double* x,
                    double* y) {
  GPX_REQUIRE(h != nullptr && A && x && y, "gemv_static: null argument");
  GPX_REQUIRE(n >= 0 && m >= 0 && lda >= m, "gemv_static: bad shape (n %d, m %d, lda %d)", n, m, lda);
  // the real limit: the partial sums of the transposed product (ceil(n / rows) chunks of m doubles, rows = 256
  // doubled up to 2^20) must fit the array; the plain product uses no scratch at all
  GPX_REQUIRE(!trans || gpx_gemv_partials_fit(n, m, GPX_STATIC_SCRATCH_ELEMS),
              "gemv_static: %d x %d exceeds the static scratch (ceil(n / 2^20) * m <= %zu doubles); use gpx_gemv", n, m,
              GPX_STATIC_SCRATCH_ELEMS);
  double* sc = gpx_static_scratch(h->device);
  GPX_REQUIRE(sc != nullptr, "gemv_static: static scratch of device %d was not initialised", h->device);
  DeviceGuard dg(h);
  // the partial kernel overwrites every slot the final kernel reads, so the previous call's contents
  // of the shared array need no clearing
  return gpx_gemv_launch(h->stream, trans, n, m, A, lda, x, y, sc, GPX_STATIC_SCRATCH_ELEMS);
}

int gpx_colsumsq_sub(gpx_handle* h, int n, int m, const double* V, int ldv, double base, double* var) {
  GPX_REQUIRE(h != nullptr && V && var, "colsumsq_sub: null argument");
  GPX_REQUIRE(n >= 0 && m >= 0 && ldv >= m, "colsumsq_sub: bad shape (n %d, m %d, ldv %d)", n, m, ldv);
  DeviceGuard dg(h);
  return gpx_colsumsq_sub_launch(h->stream, n, m, V, ldv, base, var, h->red, RED_USABLE);
}

int gpx_symmetrize(gpx_handle* h, int n, double* A, int lda) {
  GPX_REQUIRE(h != nullptr && A, "symmetrize: null argument");
  DeviceGuard dg(h);
  return gpx_symmetrize_launch(h->stream, n, A, lda);
}

int gpx_diag(gpx_handle* h, int n, const double* A, int lda, double* diag_out) {
  GPX_REQUIRE(h != nullptr && A && diag_out, "diag: null argument");
  DeviceGuard dg(h);
  return gpx_diag_launch(h->stream, n, A, lda, diag_out);
}

static int finish_grad(int nbeta, double sigma, const BetaArg& b, const double* raw, double* out) {
  // raw[0] = tr(A); raw[1] = sum A o K / sigma; raw[2+k] = sum A o (K/sigma) * beta_k diff_k / 2 (the kernel
  // works with coordinates scaled by sqrt(beta_k / 2))
  out[0] = 0.5 * raw[0];
  out[1] = 0.5 * raw[1];
  for (int k = 0; k < nbeta; ++k) out[2 + k] = -0.5 * sigma * raw[2 + k] / b.b[k];
  return 0;
}

int gpx_grad_trace(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                   const double* beta_host, const double* alpha, const double* Kinv, int ldk,
                   const double* kdiag, double* out_host) {
  GPX_REQUIRE(h != nullptr && out_host != nullptr, "grad_trace: null argument");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && X && alpha && ldx >= d, "grad_trace: bad arguments");
  GPX_TRY(check_mat("grad_trace: Kinv", Kinv, ldk, n));
  for (int k = 0; k < nbeta; ++k) GPX_REQUIRE(b.b[k] > 0.0, "grad_trace: beta[%d] must be positive", k);
  DeviceGuard dg(h);
  double* sc = red_scalars(h);
  GPX_TRY(gpx_grad_trace_launch(h->stream, X, ldx, d, nbeta, sigma, b, alpha, Kinv, ldk, 0, 0, n, n, h->red,
                                RED_USABLE, sc, kdiag));
  double raw[GPX_MAXD + 2];
  GPX_TRY(fetch(h, sc, raw, 2 + nbeta));
  return finish_grad(nbeta, sigma, b, raw, out_host);
}

int gpx_grad_trace_slab(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                        const double* beta_host, const double* alpha, const double* Kinv_slab, int ldk,
                        int row_off, int col_off, int nrows, int ncols, double* out_host) {
  GPX_REQUIRE(h != nullptr && out_host != nullptr, "grad_trace_slab: null argument");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && X && alpha && ldx >= d, "grad_trace_slab: bad arguments");
  GPX_REQUIRE(row_off >= 0 && col_off >= 0 && nrows > 0 && ncols > 0 && row_off + nrows <= n &&
                  col_off + ncols <= n && (col_off & 1) == 0,
              "grad_trace_slab: slab [%d+%d) x [%d+%d) outside the %d x %d matrix (col_off must be even)",
              row_off, nrows, col_off, ncols, n, n);
  GPX_TRY(check_mat("grad_trace_slab: Kinv", Kinv_slab, ldk, ncols));
  for (int k = 0; k < nbeta; ++k) GPX_REQUIRE(b.b[k] > 0.0, "grad_trace_slab: beta[%d] must be positive", k);
  DeviceGuard dg(h);
  double* sc = red_scalars(h);
  GPX_TRY(gpx_grad_trace_launch(h->stream, X, ldx, d, nbeta, sigma, b, alpha, Kinv_slab, ldk, row_off, col_off,
                                nrows, ncols, h->red, RED_USABLE, sc));
  double raw[GPX_MAXD + 2];
  GPX_TRY(fetch(h, sc, raw, 2 + nbeta));
  return finish_grad(nbeta, sigma, b, raw, out_host);
}

int gpx_gpr_lml_grad(gpx_handle* h, int kind, const double* X, int n, int ldx, int d,
                     double sigma, const double* beta_host, double noise, const double* y,
                     double* Ky, int ldk, double* dinv, double* vec,
                     double* alpha_out, double* a_out, double* lml_host, double* grad_host) {
  GPX_REQUIRE(h != nullptr && lml_host != nullptr, "gpr_lml_grad: null argument");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && X && y && vec && dinv && ldx >= d, "gpr_lml_grad: bad arguments");
  GPX_TRY(check_mat("gpr_lml_grad: Ky", Ky, ldk, n));
  GPX_REQUIRE(gpx_aligned16(vec) && gpx_aligned16(dinv), "gpr_lml_grad: scratch must be 16-byte aligned");
  if (grad_host)
    for (int k = 0; k < nbeta; ++k) GPX_REQUIRE(b.b[k] > 0.0, "gpr_lml_grad: beta[%d] must be positive", k);
  DeviceGuard dg(h);
  hipStream_t s = h->stream;
  // Ky = K + noise I (lower), regression.py:47-48,58-59
  GPX_TRY(gpx_gram_launch(s, X, n, ldx, X, n, ldx, d, sigma, b, noise, Ky, ldk, GPX_LOWER));
  // L = chol(Ky), regression.py:49
  GPX_HIP(hipMemsetAsync(h->info_dev, 0, sizeof(int), s));
  GPX_TRY(potrf_rec(s, n, Ky, ldk, dinv, h->info_dev, 0));
  // not positive definite (an L-BFGS-B excursion): return at once, as scipy's cholesky raises at
  // regression.py:49 -- no solves, no inverse, no traces on a garbage factor
  int info = 0;
  GPX_TRY(fetch_info(h, &info));
  if (info != 0) return info;
  // a = L^-1 y (regression.py:50), alpha = L^-T a (= Ky^-1 y, regression.py:61).  vec: n x 2 rhs
  const long ldv = 2;
  double* rhs = vec;            // n x 2 (col 0 used)
  double* a_vec = vec + 2L * n;  // n
  double* al_vec = vec + 3L * n; // n
  double* sc = red_scalars(h);
  if (grad_host) {
    // With the gradient the factor is inverted anyway (regression.py:60): W = L^-1 first, and the two solves
    // become two streaming triangular multiplies, a = W y and alpha = W^T a -- no chain of dependent leaves
    // (n = 4096: 2 x 0.22 ms of solves -> 2 x 15 us; n = 65536: 13.2 -> 6.8 ms).  sum log L_ii is taken from
    // the factor before it goes, so the LML's log-determinant term is the one of the LML-only path bit for bit.
    double* kdiag = vec + 4L * n;
    GPX_TRY(gpx_logdet_launch(s, n, Ky, ldk, h->red, sc + 1));
    GPX_TRY(trtri_rec(s, n, Ky, ldk, dinv));
    GPX_TRY(gpx_trmv_lower_launch(s, 0, n, Ky, ldk, y, a_vec, h->red, RED_USABLE));
    GPX_TRY(gpx_trmv_lower_launch(s, 1, n, Ky, ldk, a_vec, al_vec, h->red, RED_USABLE));
    GPX_TRY(gpx_dot_launch(s, n, a_vec, a_vec, h->red, sc));
    if (a_out) GPX_HIP(hipMemcpyAsync(a_out, a_vec, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    if (alpha_out) GPX_HIP(hipMemcpyAsync(alpha_out, al_vec, n * sizeof(double), hipMemcpyDeviceToDevice, s));
    GPX_TRY(lauum_upper(s, n, Ky, ldk, kdiag));
    GPX_TRY(gpx_grad_trace_launch(s, X, ldx, d, nbeta, sigma, b, al_vec, Ky, ldk, 0, 0, n, n, h->red, RED_USABLE,
                                  sc + 2, kdiag));
    double host[GPX_MAXD + 8];
    GPX_TRY(fetch(h, sc, host, 4 + nbeta));
    *lml_host = -0.5 * host[0] - host[1] - 0.5 * n * log(2.0 * M_PI);
    finish_grad(nbeta, sigma, b, host + 2, grad_host);
    return 0;
  }
  GPX_HIP(hipMemsetAsync(rhs, 0, 2L * n * sizeof(double), s));
  GPX_HIP(hipMemcpy2DAsync(rhs, ldv * sizeof(double), y, sizeof(double), sizeof(double), n,
                           hipMemcpyDeviceToDevice, s));
  double* tws = trsv_prepare(h, n, Ky, ldk, dinv);   // block inverses once, both solves use them
  GPX_TRY(trsv_prepared(h, tws, 0, n, Ky, ldk, dinv, rhs, ldv));
  GPX_HIP(hipMemcpy2DAsync(a_vec, sizeof(double), rhs, ldv * sizeof(double), sizeof(double), n,
                           hipMemcpyDeviceToDevice, s));
  GPX_TRY(trsv_prepared(h, tws, 1, n, Ky, ldk, dinv, rhs, ldv));
  GPX_HIP(hipMemcpy2DAsync(al_vec, sizeof(double), rhs, ldv * sizeof(double), sizeof(double), n,
                           hipMemcpyDeviceToDevice, s));
  GPX_TRY(gpx_dot_launch(s, n, a_vec, a_vec, h->red, sc));        // a . a
  GPX_TRY(gpx_logdet_launch(s, n, Ky, ldk, h->red, sc + 1));       // sum log L_ii
  if (a_out) GPX_HIP(hipMemcpyAsync(a_out, a_vec, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (alpha_out) GPX_HIP(hipMemcpyAsync(alpha_out, al_vec, n * sizeof(double), hipMemcpyDeviceToDevice, s));
  double host[2];
  GPX_TRY(fetch(h, sc, host, 2));
  *lml_host = -0.5 * host[0] - host[1] - 0.5 * n * log(2.0 * M_PI);
  return 0;
}

int gpx_gpr_grad_from_factor(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                             const double* beta_host, double* Ky, int ldk, const double* dinv, double* vec,
                             const double* alpha, double* grad_host) {
  GPX_REQUIRE(h != nullptr && grad_host != nullptr, "gpr_grad_from_factor: null argument");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && X && alpha && vec && dinv && ldx >= d, "gpr_grad_from_factor: bad arguments");
  GPX_TRY(check_mat("gpr_grad_from_factor: Ky", Ky, ldk, n));
  GPX_REQUIRE(gpx_aligned16(vec) && gpx_aligned16(dinv), "gpr_grad_from_factor: scratch must be 16-byte aligned");
  for (int k = 0; k < nbeta; ++k) GPX_REQUIRE(b.b[k] > 0.0, "gpr_grad_from_factor: beta[%d] must be positive", k);
  DeviceGuard dg(h);
  hipStream_t s = h->stream;
  // the second half of gpx_gpr_lml_grad, from the factor an LML-only evaluation left behind
  double* kdiag = vec + 4L * n;
  double* sc = red_scalars(h);
  GPX_TRY(trtri_rec(s, n, Ky, ldk, dinv));
  GPX_TRY(lauum_upper(s, n, Ky, ldk, kdiag));
  GPX_TRY(gpx_grad_trace_launch(s, X, ldx, d, nbeta, sigma, b, alpha, Ky, ldk, 0, 0, n, n, h->red, RED_USABLE, sc,
                                kdiag));
  double raw[GPX_MAXD + 2];
  GPX_TRY(fetch(h, sc, raw, 2 + nbeta));
  return finish_grad(nbeta, sigma, b, raw, grad_host);
}

int gpx_gpr_predict(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                    const double* beta_host, double noise, const double* y, const double* L,
                    int ldl, const double* dinv, const double* Xs, int m, int ldxs, double* V,
                    int ldv, double* vec, double* mean, double* cov, int ldc, double* var,
                    int full_cov) {
  GPX_REQUIRE(h != nullptr, "null handle");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && m > 0 && X && Xs && vec && mean && ldx >= d && ldxs >= d, "gpr_predict: bad arguments");
  GPX_TRY(check_mat("gpr_predict: L", L, ldl, n));
  GPX_TRY(check_mat("gpr_predict: V", V, ldv, m));
  GPX_REQUIRE(gpx_aligned16(vec) && dinv && gpx_aligned16(dinv), "gpr_predict: scratch must be 16-byte aligned");
  if (full_cov) GPX_TRY(check_mat("gpr_predict: cov", cov, ldc, m));
  else GPX_REQUIRE(var != nullptr, "gpr_predict: var is null");
  DeviceGuard dg(h);
  hipStream_t s = h->stream;
  // a = L^-1 y  (regression.py:37)
  double* rhs = vec;
  double* a_vec = vec + 2L * n;
  if (y != nullptr) {   // y == NULL: vec[2n, 3n) still holds a = L^-1 y from gpx_gpr_lml_grad on the same vec
    GPX_HIP(hipMemsetAsync(rhs, 0, 2L * n * sizeof(double), s));
    GPX_HIP(hipMemcpy2DAsync(rhs, 2 * sizeof(double), y, sizeof(double), sizeof(double), n,
                             hipMemcpyDeviceToDevice, s));
    GPX_TRY(solve_lower(h, 0, n, 1, L, ldl, dinv, rhs, 2));
    GPX_HIP(hipMemcpy2DAsync(a_vec, sizeof(double), rhs, 2 * sizeof(double), sizeof(double), n,
                             hipMemcpyDeviceToDevice, s));
  }
  // k* = K(Xtr, Xs) (regression.py:33), V = L^-1 k* (regression.py:38)
  GPX_TRY(gpx_gram_launch(s, X, n, ldx, Xs, m, ldxs, d, sigma, b, 0.0, V, ldv, GPX_FULL));
  GPX_TRY(solve_lower(h, 0, n, m, L, ldl, dinv, V, ldv));
  // mean = a . V (regression.py:39)
  GPX_TRY(gpx_gemv_launch(s, 1, n, m, V, ldv, a_vec, mean, h->red, RED_USABLE));
  if (full_cov) {
    // cov = K(Xs,Xs) - V^T V + noise I (regression.py:41-44)
    GPX_TRY(gpx_gram_launch(s, Xs, m, ldxs, Xs, m, ldxs, d, sigma, b, 0.0, cov, ldc, GPX_LOWER));
    GemmDesc g = mk(V, ldv, GPX_LAY_RC, V, ldv, GPX_LAY_RC, cov, ldc, cov, ldc, m, m, n, -1.0, 1.0);
    g.c_lower = 1;
    g.diag_add = noise;
    GPX_TRY(gpx_gemm_launch(s, g));
    GPX_TRY(gpx_symmetrize_launch(s, m, cov, ldc));
  } else {
    GPX_TRY(gpx_colsumsq_sub_launch(s, n, m, V, ldv, sigma + noise, var, h->red, RED_USABLE));
  }
  return 0;
}

int gpx_gpr_predict_inv(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                        const double* beta_host, double noise, const double* a, const double* W, int ldw,
                        const double* Xs, int m, int ldxs, double* Ks, double* V, int ldv, double* mean,
                        double* cov, int ldc, double* var, int full_cov) {
  GPX_REQUIRE(h != nullptr, "null handle");
  BetaArg b;
  int nbeta;
  GPX_TRY(fill_beta(kind, d, beta_host, b, nbeta));
  GPX_REQUIRE(n > 0 && m > 0 && X && Xs && a && mean && ldx >= d && ldxs >= d, "gpr_predict_inv: bad arguments");
  GPX_TRY(check_mat("gpr_predict_inv: W", W, ldw, n));
  GPX_TRY(check_mat("gpr_predict_inv: Ks", Ks, ldv, m));
  GPX_TRY(check_mat("gpr_predict_inv: V", V, ldv, m));
  GPX_REQUIRE(Ks != V, "gpr_predict_inv: Ks and V must be distinct buffers");
  if (full_cov) GPX_TRY(check_mat("gpr_predict_inv: cov", cov, ldc, m));
  else GPX_REQUIRE(var != nullptr, "gpr_predict_inv: var is null");
  DeviceGuard dg(h);
  hipStream_t s = h->stream;
  // k* = K(Xtr, Xs) (regression.py:33); V = L^-1 k* (regression.py:38) as a triangular multiply by W
  GPX_TRY(gpx_gram_launch(s, X, n, ldx, Xs, m, ldxs, d, sigma, b, 0.0, Ks, ldv, GPX_FULL));
  GemmDesc gw = mk(W, ldw, GPX_LAY_KC, Ks, ldv, GPX_LAY_RC, nullptr, 0, V, ldv, n, m, n, 1.0, 0.0);
  gw.a_tri = 1;
  GPX_TRY(gpx_gemm_launch(s, gw));
  // mean = a . V (regression.py:39)
  GPX_TRY(gpx_gemv_launch(s, 1, n, m, V, ldv, a, mean, h->red, RED_USABLE));
  if (full_cov) {
    // cov = K(Xs,Xs) - V^T V + noise I (regression.py:41-44)
    GPX_TRY(gpx_gram_launch(s, Xs, m, ldxs, Xs, m, ldxs, d, sigma, b, 0.0, cov, ldc, GPX_LOWER));
    GemmDesc g = mk(V, ldv, GPX_LAY_RC, V, ldv, GPX_LAY_RC, cov, ldc, cov, ldc, m, m, n, -1.0, 1.0);
    g.c_lower = 1;
    g.diag_add = noise;
    GPX_TRY(gpx_gemm_launch(s, g));
    GPX_TRY(gpx_symmetrize_launch(s, m, cov, ldc));
  } else {
    GPX_TRY(gpx_colsumsq_sub_launch(s, n, m, V, ldv, sigma + noise, var, h->red, RED_USABLE));
  }
  return 0;
}

int gpx_ep_scale(gpx_handle* h, int n, const double* K, int ldk, const double* sv, double* B,
                 int ldb, double* SK, int ldsk) {
  GPX_REQUIRE(h != nullptr && K && sv && B && SK, "ep_scale: null argument");
  GPX_REQUIRE(n <= 65535, "ep_scale: n too large");
  GPX_REQUIRE(ldk >= n && ldb >= n && ldsk >= n, "ep_scale: bad leading dimension");
  DeviceGuard dg(h);
  return gpx_ep_scale_launch(h->stream, n, K, ldk, sv, B, ldb, SK, ldsk);
}

int gpx_ep_recompute(gpx_handle* h, int n, const double* K, int ldk, const double* tau,
                     const double* nu, double* Bbuf, double* Vbuf, double* Sigma, int ld,
                     double* dinv, double* svec, double* mu) {
  GPX_REQUIRE(h != nullptr && K && tau && nu && svec && mu && dinv, "ep_recompute: null argument");
  GPX_REQUIRE(n > 0 && n <= 65535, "ep_recompute: n out of range");
  GPX_TRY(check_mat("ep_recompute: K", K, ldk, n));
  GPX_TRY(check_mat("ep_recompute: B", Bbuf, ld, n));
  GPX_TRY(check_mat("ep_recompute: V", Vbuf, ld, n));
  GPX_TRY(check_mat("ep_recompute: Sigma", Sigma, ld, n));
  DeviceGuard dg(h);
  hipStream_t s = h->stream;
  GPX_TRY(gpx_sqrt_launch(s, n, tau, svec));                                   // S = sqrt(tau)
  GPX_TRY(gpx_ep_scale_launch(s, n, K, ldk, svec, Bbuf, ld, Vbuf, ld));        // B, S K
  GPX_HIP(hipMemsetAsync(h->info_dev, 0, sizeof(int), s));
  GPX_TRY(potrf_rec(s, n, Bbuf, ld, dinv, h->info_dev, 0));                    // L = chol(B)
  GPX_TRY(trsm_lln(s, n, n, Bbuf, ld, dinv, Vbuf, ld));                        // V = L^-1 S K
  // Sigma = K - V^T V (lower, then mirrored)
  GemmDesc g = mk(Vbuf, ld, GPX_LAY_RC, Vbuf, ld, GPX_LAY_RC, K, ldk, Sigma, ld, n, n, n, -1.0, 1.0);
  g.c_lower = 1;
  GPX_TRY(gpx_gemm_launch(s, g));
  GPX_TRY(gpx_symmetrize_launch(s, n, Sigma, ld));
  GPX_TRY(gpx_gemv_launch(s, 0, n, n, Sigma, ld, nu, mu, h->red, RED_USABLE));  // mu = Sigma nu
  int info = 0;
  GPX_TRY(fetch_info(h, &info));
  return info;
}

int gpx_trmv(gpx_handle* h, int trans, int n, const double* W, int ldw, const double* x, double* y) {
  GPX_REQUIRE(h != nullptr && x != nullptr && y != nullptr, "trmv: null argument");
  if (n <= 0) return 0;
  GPX_TRY(check_mat("trmv: W", W, ldw, n));
  GPX_REQUIRE(x != y, "trmv: y must not alias x");
  DeviceGuard dg(h);
  return gpx_trmv_lower_launch(h->stream, trans, n, W, ldw, x, y, h->red, RED_USABLE);
}

int gpx_trmm(gpx_handle* h, int trans, int n, int m, const double* L, int ldl, const double* B, int ldb,
             double* Out, int ldo) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0 || m <= 0) return 0;
  GPX_TRY(check_mat("trmm: L", L, ldl, n));
  GPX_TRY(check_mat("trmm: B", B, ldb, m));
  GPX_TRY(check_mat("trmm: Out", Out, ldo, m));
  GPX_REQUIRE(Out != B, "trmm: Out must not alias B");
  DeviceGuard dg(h);
  // op(A) = L (stored [i][k], valid k <= i: KC) or L^T (stored [k][i], valid i <= k: RC); op(B) = B stored [k][j]
  GemmDesc g = mk(L, ldl, trans ? GPX_LAY_RC : GPX_LAY_KC, B, ldb, GPX_LAY_RC, nullptr, 0, Out, ldo, n, m, n, 1.0, 0.0);
  g.a_tri = 1;
  return gpx_gemm_launch(h->stream, g);
}

int gpx_trmm_right(gpx_handle* h, int m, int n, const double* W, int ldw, const double* X, int ldx, double* Out,
                   int ldo) {
  GPX_REQUIRE(h != nullptr, "null handle");
  if (n <= 0 || m <= 0) return 0;
  GPX_TRY(check_mat("trmm_right: W", W, ldw, n));
  GPX_TRY(check_mat("trmm_right: X", X, ldx, n));
  GPX_TRY(check_mat("trmm_right: Out", Out, ldo, n));
  GPX_REQUIRE(Out != X, "trmm_right: Out must not alias X");
  DeviceGuard dg(h);
  // Out = X W^T: op(A) = X (stored [i][k]: KC), op(B) = W^T stored as W [j][k] (KC), valid k <= j
  GemmDesc g = mk(X, ldx, GPX_LAY_KC, W, ldw, GPX_LAY_KC, nullptr, 0, Out, ldo, m, n, n, 1.0, 0.0);
  g.b_tri = 1;
  return gpx_gemm_launch(h->stream, g);
}

int gpx_ep_sweep(gpx_handle* h, int n, double* Sigma, int ld, const double* y, double* tau, double* nu,
                 const int* order_host, int count, double* work) {
  GPX_REQUIRE(h != nullptr && y && tau && nu && work, "ep_sweep: null argument");
  GPX_REQUIRE(n > 0 && count >= 0 && (count == 0 || order_host != nullptr), "ep_sweep: bad n / count / order");
  GPX_TRY(check_mat("ep_sweep: Sigma", Sigma, ld, n));
  GPX_REQUIRE(gpx_aligned16(work), "ep_sweep: work must be 16-byte aligned");
  if (count == 0) return 0;
  DeviceGuard dg(h);
  return gpx_ep_sweep_launch(h->stream, n, Sigma, ld, y, tau, nu, order_host, count, work);
}

size_t gpx_ep_sweep_work_elems(int n) { return n > 0 ? gpx_ep_sweep_work_elems_impl(n) : 0; }

}  // extern "C"
