// Internal declarations shared by the libgpx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "gpx.h"

struct gpx_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  // small library-owned scratch (partials of fixed-order reductions, info word)
  double* red = nullptr;      // device, RED_ELEMS doubles
  int* info_dev = nullptr;    // device int
  double* host_pin = nullptr; // pinned host staging for scalars, HOST_PIN doubles
  double* gemm_ws = nullptr;  // device, GPX_GEMM_WS_ELEMS doubles: partial tiles of the GEMM's split-K tail
  // device, grown on demand: block inverses + work vectors of the single-right-hand-side solves at n >= 8192
  // (gpx_trsv_blockinv_elems(n) doubles: 0.5 GB at n = 65536); freed by gpx_destroy
  double* trsv_ws = nullptr;
  size_t trsv_ws_elems = 0;
};

constexpr size_t GPX_RED_ELEMS = 1u << 23;  // 64 MiB of reduction partials
constexpr size_t GPX_HOST_PIN = 4096;
constexpr size_t GPX_STATIC_SCRATCH_ELEMS = 1u << 18;  // 2 MiB file-scope device array (gpx_gemv_static)
constexpr int GPX_MAX_DEVICES = 64;
constexpr int GPX_MAXD = 32;  // max input dimension held in kernel-argument betas

void gpx_set_error(const char* fmt, ...);

#define GPX_HIP(call)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      gpx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return GPX_EHIP;                                                                        \
    }                                                                                         \
  } while (0)

#define GPX_REQUIRE(cond, ...)     \
  do {                             \
    if (!(cond)) {                 \
      gpx_set_error(__VA_ARGS__);  \
      return GPX_EINVAL;           \
    }                              \
  } while (0)

static inline bool gpx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- operand layouts of the MFMA GEMM (gemm_f64.hip) -------------------------------------
// op(A) is M x K.  KC: stored [M][K] (k contiguous).  RC: stored [K][M] (rows of op contiguous).
// op(B) is K x N.  KC: stored [N][K] (k contiguous).  RC: stored [K][N].
enum { GPX_LAY_KC = 0, GPX_LAY_RC = 1 };
enum { GPX_INPLACE_NONE = 0, GPX_INPLACE_A = 1, GPX_INPLACE_B = 2 };

struct GemmDesc {
  const double* A; long lda; int a_lay; int a_tri;  // a_tri: stored matrix is lower (row >= col valid)
  const double* B; long ldb; int b_lay; int b_tri;
  const double* Cin; long ldcin;  // may be null when beta == 0
  double* D; long ldd;
  int M, N, K;
  double alpha, beta;
  int c_lower;       // write only row >= col (D starts on the diagonal)
  int c_upper = 0;   // with c_lower = 1 (tile enumeration) and a SYMMETRIC product: write the mirror image instead --
                     // D[row][col] for col > row only, the diagonal into diag_out[row]: the strict upper triangle of
                     // a buffer whose lower triangle holds the (triangular) operands, which are not disturbed
  double* diag_out = nullptr;
  double diag_add;   // added to D[i][i]
  int inplace;       // GPX_INPLACE_A: D aliases A (N <= 128, row tiles independent); GPX_INPLACE_B: D aliases B
                     // (M <= 128, column tiles independent); one workgroup owns the whole operand panel
  // split-K tail (filled by gpx_gemm_launch, not by callers): the last tiles of a multi-round launch are
  // computed by split_s workgroups each over a slice of K, raw accumulators into ws, then summed in a
  // fixed order by a small kernel (see launch_cfg)
  int split_s = 0;
  unsigned split_tile0 = 0;
  double* ws = nullptr;
  // pacing of the workgroups of a super-tile (filled by gpx_gemm_launch): one zeroed counter per super-tile;
  // nullptr = none (see gemm_f64_kernel)
  unsigned* sync_cnt = nullptr;
  int sync_kt = 128;  // k-tiles between two pacing points (a power of two)
  // batch of independent products of the same shape in ONE launch (grid.y): operand z starts z * stride_x
  // elements after operand 0.  The levels of the blocked triangular inversion: all diagonal blocks of one
  // size are independent, and at the small sizes a launch per block is launch-latency bound.
  int batch = 1;
  long stride_a = 0, stride_b = 0, stride_d = 0;   // (Cin, when given, moves with D)
  // two-level batch (batch_inner > 0): member z = zo * batch_inner + zi starts zi * stride_x + zo * stride2_x after
  // member 0 -- the diagonal sub-blocks (zi) of several separately stored blocks (zo): the block inverses of the
  // single-right-hand-side solves (trsv.hip)
  int batch_inner = 0;
  long stride2_a = 0, stride2_b = 0, stride2_d = 0;
  // Column-block output (the grouped trailing update of the sharded Cholesky, gpx_gemm_colblocks): the N columns
  // are cb_nb-wide blocks, block c of which lives in its own slab of leading dimension cb_nb -- element (row, col)
  // at D[cb_off[c] + (cb_rowbase + row) * cb_nb + col - c * cb_nb] -- and exists only from row cb_row0[c] -
  // cb_rowbase on (tiles above are skipped).  cb_off / cb_row0: DEVICE arrays; Cin == D; cb_nb, cb_rowbase and
  // every cb_row0[c] multiples of 128.
  const long long* cb_off = nullptr;
  const int* cb_row0 = nullptr;
  int cb_nb = 0, cb_rowbase = 0;
};

// Workspace of the split-K tail for the calling thread's current handle (set by the C ABI entry points,
// gpx_api.hip); nullptr = no split-K.
void gpx_gemm_set_workspace(double* ws, size_t elems);
constexpr size_t GPX_GEMM_WS_ELEMS = (size_t)256 * 128 * 128;   // 256 workgroups x one 128 x 128 tile: 32 MiB

int gpx_gemm_launch(hipStream_t s, const GemmDesc& g);

// leaf kernels (leaf.hip: 128x128 diagonal block Cholesky + inverse on the matrix cores)
int gpx_leaf_potf2_inv(hipStream_t s, int nb, double* A, long lda, double* dinv_blk, int* info_dev,
                       int global_off);
int gpx_leaf_diag_inv_batched(hipStream_t s, int n, const double* L, long ldl, double* dinv);
int gpx_copy_dinv_to_diag(hipStream_t s, int nb, const double* dinv_blk, double* L, long ldl);
// the same for nblk consecutive full 128 x 128 diagonal blocks in one launch
int gpx_copy_dinv_to_diag_batched(hipStream_t s, int nblk, const double* dinv, double* L, long ldl);

// gram.hip
struct BetaArg { double b[GPX_MAXD]; };
int gpx_gram_launch(hipStream_t s, const double* X1, int n1, long ldx1, const double* X2, int n2,
                    long ldx2, int d, double sigma, const BetaArg& hb, double diag_add, double* K,
                    long ldk, int uplo);
int gpx_gram_dtheta_launch(hipStream_t s, const double* X1, int n1, long ldx1, const double* X2,
                           int n2, long ldx2, int d, int nbeta, double sigma, const BetaArg& hb,
                           double* K, double* dsigma, double* dbeta, long ld, size_t plane);
int gpx_grad_trace_launch(hipStream_t s, const double* X, long ldx, int d, int nbeta, double sigma,
                          const BetaArg& hb, const double* alpha, const double* Kinv, long ldk, int roff,
                          int coff, int nr, int nc, double* red, size_t red_elems, double* out_dev,
                          const double* kdiag = nullptr);   // kdiag != nullptr: upper storage (see TraceArgs)

// trsv.hip: x <- L^-1 x (trans 0) / L^-T x (trans 1) for one strided vector
int gpx_trsv_launch(hipStream_t s, int trans, int n, const double* L, long ldl, const double* dinv, double* x,
                    long incx, double* red, size_t red_elems);
// The same solves through explicit inverses of the 1024 x 1024 diagonal blocks (64 chain steps of two streaming
// launches at n = 65536 instead of 511 dependent launches).  ws: gpx_trsv_blockinv_elems(n) doubles; prepare() fills
// it from L and dinv (7 launches, n * 1024^2 / 3 flops), solve() may then be called any number of times for the
// same factor.  Worth it from n = GPX_TRSV_BLOCKINV_MIN on.
constexpr int GPX_TRSV_BLOCK = 1024;
constexpr int GPX_TRSV_BLOCKINV_MIN = 8192;
size_t gpx_trsv_blockinv_elems(int n);
int gpx_trsv_blockinv_prepare(hipStream_t s, int n, const double* L, long ldl, const double* dinv, double* ws);
int gpx_trsv_blockinv_solve(hipStream_t s, int trans, int n, const double* L, long ldl, const double* dinv,
                            double* ws, double* x, long incx, double* red, size_t red_elems);
// y = W x (trans = 0) or W^T x (trans = 1) for a lower-triangular W (only j <= i read); y must not alias x
int gpx_trmv_lower_launch(hipStream_t s, int trans, int n, const double* W, long ldw, const double* x, double* y,
                          double* red, size_t red_elems);

// ep.hip: one sweep of site updates in the given order, rank-1 updates delayed GPX_EP_BLOCK at a time;
// work: gpx_ep_sweep_work_elems_impl(n) device doubles
constexpr int GPX_EP_BLOCK = 256;
size_t gpx_ep_sweep_work_elems_impl(int n);
int gpx_ep_sweep_launch(hipStream_t s, int n, double* Sigma, long ld, const double* y, double* tau, double* nu,
                        const int* order_host, int count, double* work);

// blas2.hip
int gpx_static_scratch_init(int device);   // once per device, that device current (called by gpx_create)
double* gpx_static_scratch(int device);    // device address of this GPU's copy of the static scratch
int gpx_logdet_launch(hipStream_t s, int n, const double* L, long ldl, double* red, double* out_dev);
int gpx_dot_launch(hipStream_t s, int n, const double* x, const double* y, double* red, double* out_dev);
int gpx_gemv_launch(hipStream_t s, int trans, int n, int m, const double* A, long lda,
                    const double* x, double* y, double* red, size_t red_elems, double alpha = 1.0,
                    double beta = 0.0);   // y = alpha op(A) x + beta y
bool gpx_gemv_partials_fit(int n, int m, size_t red_elems);   // transposed product: chunks x m <= red_elems?
int gpx_symmetrize_launch(hipStream_t s, int n, double* A, long lda);
int gpx_diag_launch(hipStream_t s, int n, const double* A, long lda, double* out);
int gpx_ep_scale_launch(hipStream_t s, int n, const double* K, long ldk, const double* sv, double* B,
                        long ldb, double* SK, long ldsk);
int gpx_sqrt_launch(hipStream_t s, int n, const double* x, double* y);
int gpx_colsumsq_sub_launch(hipStream_t s, int n, int m, const double* V, long ldv, double base,
                            double* var, double* red, size_t red_elems);
