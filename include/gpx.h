/*
 * gpx.h -- C ABI of libgpx.so: the MI355X (gfx950) hot path of exact Gaussian-process
 * regression (RBF/ARD Gram matrix, blocked fp64 Cholesky on MFMA, triangular solves,
 * log-determinant, K^-1, LML gradient traces, predictive mean/covariance) and the
 * linear algebra inside the EP classifier.
 *
 * The reference (emakryo/gaussian_process, pure Python) has no FFI: its boundary is the
 * Python object protocol of kernels.py / regression.py.  Each entry point below names
 * the reference call site (file:line under gaussian_process/) whose arithmetic it
 * replaces; the Python package gaussian_process_amd is the host-side mirror of that protocol and binds
 * these symbols with ctypes.
 *
 * Conventions
 *   - every matrix is fp64, ROW-major, with an explicit leading dimension (elements);
 *     every pointer is a DEVICE pointer unless the name ends in _host;
 *   - base pointers must be 16-byte aligned and leading dimensions even (the kernels use
 *     16-byte vector accesses); violations return GPX_EINVAL, they never fault;
 *   - "lower" matrices are referenced in their lower triangle only (row >= col);
 *   - the caller owns every buffer; the library allocates nothing in these calls except
 *     inside gpx_create;
 *   - calls are asynchronous on the handle's stream unless they return a host value
 *     (info, scalars), in which case they synchronise that stream;
 *   - return value: 0 ok; >0 LAPACK-style info (order of the leading minor that is not
 *     positive definite); <0 GPX_E* error, text via gpx_last_error().
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef GPX_H
#define GPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_EINVAL (-1)  /* bad argument (shape, alignment, null pointer)      */
#define GPX_EHIP (-2)    /* a HIP runtime call failed                           */
#define GPX_ENOMEM (-3)  /* workspace too small / allocation failed            */

#define GPX_KIND_RBF 0 /* k = sigma * exp(-0.5 * beta    * |x - x'|^2)           kernels.py:9-46  */
#define GPX_KIND_ARD 1 /* k = sigma * exp(-0.5 * sum_k beta_k (x_k - x'_k)^2)    kernels.py:49-88 */

#define GPX_FULL 0  /* write the whole n1 x n2 matrix                                   */
#define GPX_LOWER 1 /* square case only: write row >= col, zero inside diagonal blocks   */

#define GPX_NB 128 /* diagonal-block size of the blocked factorisation / dinv layout      */

typedef struct gpx_handle gpx_handle;

/* ---- life cycle ---------------------------------------------------------------------- */
int gpx_create(gpx_handle** out, int device);
int gpx_destroy(gpx_handle* h);
/* stream: a hipStream_t (torch.cuda.current_stream().cuda_stream); NULL = default stream.
 * A handle's scratch (the reduction partials, the split-K workspace of the GEMM, the pinned staging buffer)
 * serves ONE stream at a time: work given to the handle on two streams must not overlap in time (re-bind only
 * after the new stream waits for the old one, as the Python runtime's bind_stream() does), or use one handle
 * per stream. */
int gpx_set_stream(gpx_handle* h, void* stream);
int gpx_synchronize(gpx_handle* h);
const char* gpx_last_error(void);
const char* gpx_version(void);
/* Number of doubles the caller must provide for `dinv` given n (ceil(n/128) blocks of 128x128). */
size_t gpx_dinv_elems(int n);

/* ---- Gram matrix and derivatives ------------------------------------------------------ */
/* K[i][j] = sigma * exp(-0.5 * sum_k beta_k (X1[i][k]-X2[j][k])^2) (+ diag_add if i==j).
 * Replaces kernels.py:34-37 (RBF, evaluated by exact differences) and kernels.py:77-79 (ARD).
 * beta_host: d values (ARD) or 1 value (RBF), host memory.  X1: n1 x d (ld = ldx1), X2 likewise;
 * X2 == NULL means X2 = X1.  uplo = GPX_LOWER requires X2 == NULL or n1 == n2. */
int gpx_gram(gpx_handle* h, int kind, const double* X1, int n1, int ldx1, const double* X2, int n2,
             int ldx2, int d, double sigma, const double* beta_host, double diag_add, double* K,
             int ldk, int uplo);
/* Dense dK/dtheta planes for the drop-in KernelValue.dtheta (kernels.py:38-39, 80-81):
 * dsigma = K / sigma (n1 x n2) and dbeta[p] = -0.5 * diff_p * K, p < (kind==ARD ? d : 1), plane
 * stride `plane` elements.  Small/medium n only: the fused gradient below never forms them. */
int gpx_gram_dtheta(gpx_handle* h, int kind, const double* X1, int n1, int ldx1, const double* X2,
                    int n2, int ldx2, int d, double sigma, const double* beta_host, double* K,
                    double* dsigma, double* dbeta, int ld, size_t plane);

/* ---- Cholesky and friends ------------------------------------------------------------- */
/* In-place lower Cholesky A = L L^T of the lower triangle of A (n x n), plus the inverses of
 * the 128x128 diagonal blocks of L in dinv (gpx_dinv_elems(n) doubles).  The strict upper
 * triangle inside each diagonal block is zeroed; blocks above the diagonal are not touched.
 * Replaces scipy.linalg.cholesky(lower=True) at regression.py:36,49 and np.linalg.cholesky at
 * expectation_propagation.py:62,91.  Returns LAPACK info (>0: leading minor not PD). */
int gpx_potrf(gpx_handle* h, int n, double* A, int lda, double* dinv);
/* Same factorisation without the host synchronisation: LAPACK info is left in info_dev (a device
 * int the caller zeroes first; only the first failure is recorded).  For callers that pipeline
 * panels (the multi-GPU block-column Cholesky). */
int gpx_potrf_async(gpx_handle* h, int n, double* A, int lda, double* dinv, int* info_dev, int info_offset);
/* X (m x n, row-major) <- X L^-T for the n x n lower factor L: the panel solve
 * A21 L11^-T inside a blocked Cholesky (dtrsm 'R','L','T' inside dpotrf, regression.py:36,49). */
int gpx_trsm_right(gpx_handle* h, int m, int n, const double* L, int ldl, const double* dinv, double* X,
                   int ldx);
/* Recompute dinv for a factor L that did not come from gpx_potrf. */
int gpx_diag_inv(gpx_handle* h, int n, const double* L, int ldl, double* dinv);
/* B <- L^-1 B (trans = 0) or L^-T B (trans = 1); B is n x nrhs row-major (nrhs may be 1, in which case
 * ldb is the stride of the vector and may be any value >= 1).
 * Replaces solve_triangular at regression.py:37-38,50, the second half of cho_solve behind
 * regression.py:61, and np.linalg.solve(L, .) at expectation_propagation.py:63,92,107-108,117,136. */
int gpx_trsm(gpx_handle* h, int trans, int n, int nrhs, const double* L, int ldl,
             const double* dinv, double* B, int ldb);
/* out_host[0] = sum_i log L[i][i]   (regression.py:53, expectation_propagation.py:133). */
int gpx_logdet(gpx_handle* h, int n, const double* L, int ldl, double* out_host);
/* L <- L^-1 in place (lower).  Needs no workspace: the blocks of L's own n x ldl buffer that lie strictly
 * above the 128 x 128 diagonal blocks serve as scratch and hold garbage on exit (no lower-triangular
 * consumer reads them). */
int gpx_trtri(gpx_handle* h, int n, double* L, int ldl, const double* dinv);
/* Kinv(lower) = W^T W for lower-triangular W = L^-1: together with gpx_trtri this is potri and
 * replaces scipy.linalg.inv(Ky) at regression.py:60 (SPD route instead of LU). */
int gpx_lauum(gpx_handle* h, int n, const double* W, int ldw, double* Kinv, int ldk);
/* The same product kept in W's own buffer: Ky^-1[i][j], i < j, goes to W[i * ldw + j] (the strict upper
 * triangle, also inside the diagonal blocks) and its diagonal to kdiag (n doubles); the lower triangle
 * (W itself) is not disturbed.  With gpx_trtri: factor, inverse factor and inverse in ONE n x n buffer
 * (regression.py:60 in half the memory).  gpx_grad_trace reads this storage when given kdiag. */
int gpx_lauum_upper(gpx_handle* h, int n, double* W, int ldw, double* kdiag);
/* C(m x m) <- C - V^T V + diag_add * I, V is n x m.  uplo GPX_FULL or GPX_LOWER.
 * Replaces regression.py:41-44 and `K - V.T @ V` at expectation_propagation.py:64,93. */
int gpx_syrk_sub(gpx_handle* h, int n, int m, const double* V, int ldv, double* C, int ldc,
                 double diag_add, int uplo);
/* General D = alpha * op(A) op(B) + beta * C on the fp64 MFMA kernel (row-major).
 * transa/transb: 0 = stored as op, 1 = stored transposed. */
int gpx_gemm(gpx_handle* h, int transa, int transb, int M, int N, int K, double alpha,
             const double* A, int lda, const double* B, int ldb, double beta, double* C, int ldc);
/* Grouped trailing update of the sharded (2-D block-cyclic) Cholesky -- the arithmetic of the dsyrk / dgemm
 * inside scipy.linalg.cholesky (regression.py:36,49) for SEVERAL tile columns in one launch:
 *   D_c <- alpha * A B_c^T + beta * D_c   for every nb-wide column block c of the N columns,
 * A stored M x K (lda), B stored N x K (ldb), both k-contiguous.  Column block c is its own slab of leading
 * dimension nb: element (row, col) of the product lives at D[blk_off[c] + (row_base + row) * nb + col - c * nb]
 * and exists only for row_base + row >= blk_row0[c] (a slab starts at its diagonal tile; tiles above are
 * skipped).  Bit 0 of blk_row0[c] set: the slab's first nb rows are the column's DIAGONAL tile, and the 128 x 128
 * tiles strictly above that tile's diagonal, which the factorisation never reads, are skipped as well (the launch
 * then does the lower trapezoid's flops; on a P x Q grid a process row that does not own a column's diagonal tile
 * leaves the bit clear).  blk_off (ceil(N / nb) 64-bit element offsets, may be negative) and blk_row0 are DEVICE arrays owned
 * by the caller, who guarantees that every addressed element lies inside D's allocation; nb, row_base and every
 * blk_row0[c] & ~1 must be multiples of 128.  One launch instead of one skinny N = nb product per tile column: the A
 * panel is shared by all column blocks through the L2. */
int gpx_gemm_colblocks(gpx_handle* h, int M, int N, int K, double alpha, const double* A, int lda, const double* B,
                       int ldb, double beta, double* D, int nb, int row_base, const long long* blk_off,
                       const int* blk_row0);
/* y = W x (trans = 0) or y = W^T x (trans = 1) for a lower-triangular n x n matrix W (only W[i][j], j <= i, is
 * read; what lies above the diagonal may be anything); y must not alias x.  With W = L^-1 from gpx_trtri these
 * are a = L^-1 y and alpha = L^-T a of regression.py:50,61 as two streaming passes without a dependent chain:
 * what the fused gradient evaluation uses in place of the two triangular solves. */
int gpx_trmv(gpx_handle* h, int trans, int n, const double* W, int ldw, const double* x, double* y);
/* Out (n x m) = L B (trans = 0) or L^T B (trans = 1) for a lower-triangular n x n matrix L (only its
 * lower triangle is read) and B n x m; Out must not alias B.  Sampling f = L z from the GP prior /
 * posterior with the factor gpx_potrf produced -- the O(n^2) replacement for numpy's SVD-based
 * multivariate_normal at datasets.py:12 -- and, with L = W = a tile's explicit inverse (gpx_trtri),
 * the many-right-hand-side tile solves W B / W^T B of the multi-GPU path (regression.py:38,61). */
int gpx_trmm(gpx_handle* h, int trans, int n, int m, const double* L, int ldl, const double* B, int ldb,
             double* Out, int ldo);
/* Out (m x n) = X W^T for a lower-triangular n x n matrix W (only W[i][j], j <= i, is read), X m x n; Out must not
 * alias X.  With W = L_kk^-1 (gpx_trtri of a diagonal tile) this is the panel solve A21 L11^-T inside
 * scipy.linalg.cholesky (regression.py:36,49) as ONE triangular-aware MFMA product instead of the recursion of
 * gpx_trsm_right over the tile's 128-blocks: the sharded Cholesky's panel step (distributed.py). */
int gpx_trmm_right(gpx_handle* h, int m, int n, const double* W, int ldw, const double* X, int ldx, double* Out,
                   int ldo);
/* y(m) = A^T x for A n x m row-major (trans=1) or y(n) = A x (trans=0); fixed-order reduction.
 * regression.py:39 (mean = a . V), expectation_propagation.py:65,94,116. */
int gpx_gemv(gpx_handle* h, int trans, int n, int m, const double* A, int lda, const double* x,
             double* y);
/* y = alpha * op(A) x + beta * y (same kernels and fixed summation order as gpx_gemv; x and y must not
 * alias): the single-right-hand-side updates r -= L[I, k] x_k / r -= L[I, k]^T x_I of the multi-GPU
 * triangular solves behind regression.py:50,61. */
int gpx_gemv_acc(gpx_handle* h, int trans, int n, int m, double alpha, const double* A, int lda,
                 const double* x, double beta, double* y);
/* var[j] = base - sum_i V[i][j]^2 for V n x m: the diagonal of K** - V^T V + noise I (regression.py:41-44)
 * and of expectation_propagation.py:118 without the m x m matrix; fixed-order reduction.  The multi-GPU
 * path sums the per-rank results (base = 0 on all ranks but one). */
int gpx_colsumsq_sub(gpx_handle* h, int n, int m, const double* V, int ldv, double base, double* var);
/* gpx_gemv for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc): the same kernels, but the partial sums of the transposed product live in a
 * file-scope 2 MiB device array of the library instead of the handle's hipMalloc'ed scratch.  It has no
 * counterpart in the reference.  Calls no hipMalloc / hipFree / allocating routine.  Limit (transposed
 * product, trans = 1): the partial sums -- ceil(n / rows) chunks of m doubles, rows = 256 doubled up to 2^20
 * until they fit -- must fit the 262144 (2^18) doubles of the array, i.e. ceil(n / 2^20) * m <= 2^18; shapes
 * beyond that return GPX_EINVAL ("exceeds the static scratch").  The plain product (trans = 0) uses no scratch
 * and has no limit of its own.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race; calls on one stream are ordered. */
int gpx_gemv_static(gpx_handle* h, int trans, int n, int m, const double* A, int lda, const double* x,
                    double* y);
/* out_host[0] = x . y */
int gpx_dot(gpx_handle* h, int n, const double* x, const double* y, double* out_host);
/* Mirror the lower triangle into the upper one (symmetric full matrix for NumPy consumers). */
int gpx_symmetrize(gpx_handle* h, int n, double* A, int lda);
/* diag_out[i] = A[i][i] */
int gpx_diag(gpx_handle* h, int n, const double* A, int lda, double* diag_out);

/* ---- LML gradient --------------------------------------------------------------------- */
/* out_host[0]      = 0.5 * tr(A)                      d/d(noise)      regression.py:63,70
 * out_host[1]      = 0.5 * sum_ij A_ij K_ij / sigma   d/d(cov sigma)  kernels.py:39,81
 * out_host[2 + p]  = 0.5 * sum_ij A_ij (-0.5 diff_p K_ij), p < (ARD ? d : 1)   kernels.py:38,80
 * with A = alpha alpha^T - Kinv (regression.py:62) and K, diff recomputed from X on the fly: the (d,n,n)
 * tensor of kernels.py:80 never exists.  Kinv storage: kdiag == NULL -- the LOWER triangle of Kinv (what
 * gpx_lauum writes); kdiag != NULL -- the STRICT UPPER triangle of Kinv plus the diagonal in kdiag (what
 * gpx_lauum_upper writes). */
int gpx_grad_trace(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                   const double* beta_host, const double* alpha, const double* Kinv, int ldk,
                   const double* kdiag, double* out_host);
/* The same sums restricted to a slab of the symmetric matrix: Kinv_slab holds rows
 * [row_off, row_off+nrows) x columns [col_off, col_off+ncols) of Ky^-1 (only its entries on or below
 * the global diagonal are read).  The slabs of a partition of the lower triangle add up to
 * gpx_grad_trace's result (the sums are linear): this is how the block-column owners of the
 * multi-GPU path each contribute their share of regression.py:62-75 without forming Ky^-1 whole. */
int gpx_grad_trace_slab(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                        const double* beta_host, const double* alpha, const double* Kinv_slab, int ldk,
                        int row_off, int col_off, int nrows, int ncols, double* out_host);

/* ---- fused estimator paths ------------------------------------------------------------ */
/* One evaluation of (LML, grad) = regression.py:46-76 from one Gram + one factorisation in ONE n x n
 * buffer:
 *   Ky (n x ldk, overwritten): with the gradient, L^-1 in the lower triangle and Ky^-1 in the strict upper
 *   triangle on exit (its diagonal in vec[4n, 5n)); without, L in the lower triangle.
 *   dinv (gpx_dinv_elems(n)), vec (5*n doubles of scratch).
 *   alpha_out (n) = Ky^-1 y, a_out (n) = L^-1 y (either may be NULL).
 *   lml_host[0] = LML; grad_host[0..1+nb] as gpx_grad_trace (NULL = skip the gradient).
 * Returns LAPACK info > 0 right after the factorisation if Ky is not positive definite. */
int gpx_gpr_lml_grad(gpx_handle* h, int kind, const double* X, int n, int ldx, int d,
                     double sigma, const double* beta_host, double noise, const double* y,
                     double* Ky, int ldk, double* dinv, double* vec,
                     double* alpha_out, double* a_out, double* lml_host, double* grad_host);
/* regression.py:57-76 when regression.py:46-55 has just run at the same hyper-parameters (the order of
 * base.py:35-37): continues from what gpx_gpr_lml_grad(grad_host = NULL) left -- L in the lower triangle of Ky,
 * its dinv, alpha = Ky^-1 y -- to the gradient, without a second Gram matrix or factorisation.  On exit
 * Ky, vec[4n, 5n) are as after the fused call (L^-1 lower, Ky^-1 strict upper, its diagonal in vec), so
 * gpx_gpr_predict_inv applies.  vec: the same 5*n doubles; only [4n, 5n) is written. */
int gpx_gpr_grad_from_factor(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                             const double* beta_host, double* Ky, int ldk, const double* dinv, double* vec,
                             const double* alpha, double* grad_host);
/* regression.py:31-44 given the factor L (lower) of Ky and its dinv:
 *   V (n x ldv >= m) scratch, mean (m), cov (m x ldc) = K(Xs,Xs) - V^T V + noise I (full matrix,
 *   or only its lower triangle + mirrored when full_cov=1; full_cov=0 fills only var (m)).
 *   vec: 3*n doubles.  y == NULL: vec[2n, 3n) already holds a = L^-1 y -- what gpx_gpr_lml_grad leaves there when
 *   given the same vec -- and the forward solve of regression.py:37 is not repeated. */
int gpx_gpr_predict(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                    const double* beta_host, double noise, const double* y, const double* L,
                    int ldl, const double* dinv, const double* Xs, int m, int ldxs, double* V,
                    int ldv, double* vec, double* mean, double* cov, int ldc, double* var,
                    int full_cov);

/* The same prediction (regression.py:31-44) when the gradient pass has already turned the factor into
 * its inverse: W = L^-1 (lower, what gpx_gpr_lml_grad leaves in Ky) and a = L^-1 y (its a_out).
 * V = L^-1 k* is then the triangular MULTIPLY W k* -- no second factorisation after an optimiser
 * evaluation, and one GEMM instead of a solve.  Ks: n x ldv scratch for k* (distinct from V). */
int gpx_gpr_predict_inv(gpx_handle* h, int kind, const double* X, int n, int ldx, int d, double sigma,
                        const double* beta_host, double noise, const double* a, const double* W, int ldw,
                        const double* Xs, int m, int ldxs, double* Ks, double* V, int ldv, double* mean,
                        double* cov, int ldc, double* var, int full_cov);

/* ---- EP inner linear algebra (expectation_propagation.py:61-65, 90-94) ------------------ */
/* B = I + (s s^T) o K  (lower) and SK = diag(s) K (full), s = sqrt(tau). */
int gpx_ep_scale(gpx_handle* h, int n, const double* K, int ldk, const double* s, double* B,
                 int ldb, double* SK, int ldsk);
/* One recompute: L = chol(I + S K S) (in Bbuf), V = L^-1 S K (in Vbuf), Sigma = K - V^T V (full,
 * symmetric), mu = Sigma nu.  Returns potrf info. */
int gpx_ep_recompute(gpx_handle* h, int n, const double* K, int ldk, const double* tau,
                     const double* nu, double* Bbuf, double* Vbuf, double* Sigma, int ld,
                     double* dinv, double* svec, double* mu);

/* One sweep of EP site updates on the device, sites visited strictly in the order given
 * (expectation_propagation.py:72-88: cavity, probit moment match, tau[i] / nu[i], the rank-1 update of
 * Sigma, mu = Sigma nu).  order_host: `count` site indices in HOST memory (the caller's
 * np.random.permutation with the frozen sites removed); y, tau, nu: device vectors of n; Sigma: n x ld
 * device matrix, symmetric: only its LOWER triangle is read and updated in place (the strict upper
 * triangle goes stale).  The rank-1 updates are delayed and applied 256 at a time by the MFMA GEMM; the
 * result equals the site-by-site recursion up to rounding.  mu is carried inside the sweep but not
 * returned -- call gpx_ep_recompute after the sweep as the reference does
 * (expectation_propagation.py:90-94).  work: gpx_ep_sweep_work_elems(n) device doubles, 16-byte aligned.
 * Asynchronous on the handle's stream. */
int gpx_ep_sweep(gpx_handle* h, int n, double* Sigma, int ld, const double* y, double* tau, double* nu,
                 const int* order_host, int count, double* work);
/* Number of doubles gpx_ep_sweep needs in `work` for n sites (2 n x 256 delayed columns + 6 n state). */
size_t gpx_ep_sweep_work_elems(int n);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
