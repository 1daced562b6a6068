// Gram-matrix construction and the fused LML-gradient trace (HBM-bound kernels).
//
//   K[i][j] = sigma * exp(-0.5 * sum_k beta_k (x_ik - x'_jk)^2)      kernels.py:37 / 77-79
//
// RBF is the ARD formula with every beta_k equal (the reference evaluates RBF through the
// |x|^2+|y|^2-2xy expansion, kernels.py:34-36; exact differences agree with it to ~2e-15 and
// keep the diagonal exactly sigma).  The (n1,n2,d) tensors of kernels.py:77-80 never exist:
// each output element is produced from LDS/register-resident rows of X and written once.
//
// Layout: a workgroup of 256 threads owns a 64-row x 128-column tile; each wave owns 16 rows
// and every lane two adjacent columns, so one store instruction writes 1 KiB of one row
// (16 B per lane, fully coalesced).  The two X2 rows of a lane live in registers (pre-scaled by
// sqrt(beta_k)); the X1 rows are read from LDS at wave-uniform addresses (broadcast).
#include <cstdlib>
#include <type_traits>

#include "gpx_internal.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int GT_ROWS = 64, GT_COLS = 128, GT_THREADS = 256, GT_SUBS = 4;

// Horner step with the coefficient in an SGPR pair: p*r + c as ONE v_fma_f64.  (Left to itself the
// compiler keeps the non-inline coefficients in VGPRs and pairs every v_fmac with a v_mov_b64,
// which doubles the instruction count of the polynomial.)
__device__ __forceinline__ double horner(double p, double r, double c) {
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
  return o;
}

// exp for the Gram and trace kernels: K = sigma * exp(-h), h = 0.5 * sum_k beta_k (x_k - x'_k)^2 >= 0 handed
// over already halved (the coordinates are pre-scaled by sqrt(beta_k / 2)).  Table-driven: with
// kf = rint(-h * 64 / ln 2) = 64 e + j,  exp(-h) = 2^e * 2^(j/64) * exp(r),  |r| <= ln 2 / 128, so a
// degree-5 Taylor polynomial suffices (truncation r^6 / 720 < 4e-17) and sigma * 2^(j/64) comes from a
// 64-entry table in LDS (one ds_read per element).  15 VALU instructions per element against 24 for the
// round-1 routine (Cody-Waite + degree-13 Horner, then the scaling by sigma): at d = 8 the Gram kernel was
// bound by its fp64 VALU work (40 instructions per element), not by the stores.  Error < 1.5 ulp;
// underflows gradually through v_ldexp_f64.  The coefficients sit in SGPR pairs (see horner()).
constexpr int EXP_TAB = 64;

// h is clamped to 800 first (ONE v_min_f64): exp(-800) = 2^-1154 underflows to an exact 0 through ldexp, kf stays
// within +-2^17 (no int overflow in the conversion -- unscaled features or a large beta on an L-BFGS-B excursion
// reach h ~ 1e9 easily) and h = +inf gives 0 as exp(-inf) does.  v_min_f64 returns the non-NaN operand, so a NaN
// coordinate would give 0 here: the Python layer rejects non-finite training inputs (check_X_y) and restores the
// NaNs of non-finite test points on the host (regression.py / kernels.py: _nan_rows).
__device__ __forceinline__ double exp_neg_tab(double h, const double* tab) {
  h = __builtin_fmin(h, 800.0);
  // (rint + cvt were also tried as a magic-number addition -- h * c + 1.5 * 2^52, the integer read from the low word:
  // no change, 5.35 against 5.17 ms for the trace kernel within box noise; v_rndne_f64 / v_cvt_i32_f64 are not what
  // limits these kernels.  PMC: the Gram kernel issues VALU instructions on 91 % of its SIMD cycles, the trace
  // kernel on 71 % -- both are bound by their fp64 instruction COUNT, 34 resp. 56 per element at d = 8.)
  const double kf = rint(h * -9.23324826168936574e+01);            // -64 / ln 2
  const int ki = (int)kf;
  double r = fma(kf, -1.0830417275428772e-02, -h);                 // r = -h - kf * ln2/64: hi part (32 trailing zero bits: kf * hi is exact)
  r = fma(kf, -7.420820373486988e-09, r);                          // lo part
  const double tj = tab[ki & (EXP_TAB - 1)];                       // sigma * 2^(j/64)
  double p = horner(8.3333333333333332e-03, r, 4.1666666666666664e-02);   // 1/5!, 1/4!
  p = horner(p, r, 1.6666666666666666e-01);                        // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p * tj, ki >> 6);
}

struct GramArgs {
  const double* X1; long ldx1; int n1;
  const double* X2; long ldx2; int n2;
  int d;
  double sigma, diag_add;
  double* K; long ldk;
  int lower;
  int nt;      // interior tiles: non-temporal stores (development switch GPX_GRAM_NT, see gpx_gram_launch)
  BetaArg sb;  // sqrt(beta_k / 2)
  double tab[EXP_TAB];  // sigma * 2^(j/64)
};

__device__ __forceinline__ void store_pair(double* p, double a, double b, bool nt) {
  const v2d v = {a, b};
  if (nt) __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p));
  else *reinterpret_cast<v2d*>(p) = v;
}

// D > 0: compile-time dimension (unrolled, registers); D == 0: runtime d <= GPX_MAXD.
// CP: column pairs per lane.  A workgroup owns GT_SUBS row sub-tiles (64 rows each) of one strip of
// 128 * CP columns; lane l holds columns c0 + 128 q + 2 l, + 1 for q < CP, so every store instruction
// of a wave still writes 1 KiB of one row and a wave writes CP KiB of each row back to back.  CP = 4
// (512-column strips) for the compiled dimensions: the store pattern sustains 6.1 TB/s against 5.6
// for 128-column strips (tools/store_pattern.hip), and a row of X1 read from LDS serves 8 elements.
template <int D, int CP>
__global__ __launch_bounds__(GT_THREADS) void gram_kernel(const GramArgs a) {
  constexpr int DD = D > 0 ? D : GPX_MAXD;
  constexpr int COLS = GT_COLS * CP;
  __shared__ double sx1[GT_SUBS * GT_ROWS * DD];
  __shared__ double stab[EXP_TAB];
  const int d = D > 0 ? D : a.d;
  int bx = blockIdx.x, by = blockIdx.y;
  if (a.lower && CP == 1) {
    // lower triangle: a 1-D grid over the live (row block, strip) pairs only -- row block b has the strips
    // 0 .. 2 b + 1 (a 2-D grid launches as many workgroups again that exit at once: 65 536 of them at
    // n = 65536).  id = b (b + 1) + strip.
    const unsigned id = blockIdx.x;
    unsigned b = (unsigned)((sqrt(4.0 * (double)id + 1.0) - 1.0) * 0.5);
    while ((b + 1) * (b + 2) <= id) ++b;
    while (b * (b + 1) > id) --b;
    by = (int)b;
    bx = (int)(id - b * (b + 1));
  }
  const int rb = by * (GT_SUBS * GT_ROWS), c0 = bx * COLS;
  if (c0 >= a.n2) return;                                  // (ragged last row block of the packed grid)
  if (a.lower && c0 > rb + GT_SUBS * GT_ROWS - 1) return;  // every sub-tile strictly above the diagonal
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  if (t < EXP_TAB) stab[t] = a.tab[t];

  for (int e = t; e < GT_SUBS * GT_ROWS * d; e += GT_THREADS) {
    const int r = e / d, k = e - r * d;
    const int gr = rb + r;
    sx1[e] = gr < a.n1 ? a.X1[(long)gr * a.ldx1 + k] * a.sb.b[k] : 0.0;
  }
  double z[CP][2][DD];
#pragma unroll
  for (int q = 0; q < CP; ++q) {
    const int c = c0 + q * GT_COLS + 2 * lane;
#pragma unroll
    for (int k = 0; k < DD; ++k) {
      if (k < d) {
        z[q][0][k] = c < a.n2 ? a.X2[(long)c * a.ldx2 + k] * a.sb.b[k] : 0.0;
        z[q][1][k] = c + 1 < a.n2 ? a.X2[(long)(c + 1) * a.ldx2 + k] * a.sb.b[k] : 0.0;
      }
    }
  }
  __syncthreads();

  for (int sub = 0; sub < GT_SUBS; ++sub) {
    const int r0 = rb + sub * GT_ROWS;
    if (r0 >= a.n1) break;
    if (a.lower && c0 > r0 + GT_ROWS - 1) continue;  // tile strictly above the diagonal: not touched
    const double* xs = sx1 + sub * GT_ROWS * d;
    // interior tiles (no matrix edge, no diagonal inside): no selects, no bounds tests
    const bool touches_diag = c0 < r0 + GT_ROWS && r0 < c0 + COLS;
    const bool interior = r0 + GT_ROWS <= a.n1 && c0 + COLS <= a.n2 &&
                          !(touches_diag && (a.lower || a.diag_add != 0.0));
    if (interior) {
      double* dst = a.K + (long)(r0 + wid * 16) * a.ldk + c0 + 2 * lane;
      for (int rr = 0; rr < 16; ++rr) {
        const double* xr = xs + (wid * 16 + rr) * d;
        double s[CP][2];
#pragma unroll
        for (int q = 0; q < CP; ++q) s[q][0] = s[q][1] = 0.0;
#pragma unroll
        for (int k = 0; k < DD; ++k) {
          if (k < d) {
            const double x = xr[k];
#pragma unroll
            for (int q = 0; q < CP; ++q) {
              const double d0 = x - z[q][0][k], d1 = x - z[q][1][k];
              s[q][0] = fma(d0, d0, s[q][0]);
              s[q][1] = fma(d1, d1, s[q][1]);
            }
          }
        }
#pragma unroll
        for (int q = 0; q < CP; ++q)
          store_pair(dst + q * GT_COLS, exp_neg_tab(s[q][0], stab), exp_neg_tab(s[q][1], stab), a.nt != 0);
        dst += a.ldk;
      }
      continue;
    }
    for (int rr = 0; rr < 16; ++rr) {
      const int r = wid * 16 + rr;
      const int gr = r0 + r;
      if (gr >= a.n1) break;
      for (int q = 0; q < CP; ++q) {
        const int c = c0 + q * GT_COLS + 2 * lane;
        if (c >= a.n2) break;
        if (a.lower && c0 + q * GT_COLS > r0 + GT_ROWS - 1) break;   // this 128-column piece lies above the diagonal
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < DD; ++k) {
          if (k < d) {
            const double x = xs[r * d + k];
            const double d0 = x - z[q][0][k], d1 = x - z[q][1][k];
            s0 = fma(d0, d0, s0);
            s1 = fma(d1, d1, s1);
          }
        }
        double k0 = exp_neg_tab(s0, stab);
        double k1 = exp_neg_tab(s1, stab);
        if (gr == c) k0 += a.diag_add;
        if (gr == c + 1) k1 += a.diag_add;
        if (a.lower) {
          if (c > gr) k0 = 0.0;
          if (c + 1 > gr) k1 = 0.0;
        }
        double* dst = a.K + (long)gr * a.ldk + c;
        if (c + 1 < a.n2) {
          *reinterpret_cast<v2d*>(dst) = (v2d){k0, k1};
        } else {
          dst[0] = k0;
        }
      }
    }
  }
}

// K, dsigma, dbeta planes (dense KernelValue.dtheta for small problems).
struct DthetaArgs {
  const double* X1; long ldx1; int n1;
  const double* X2; long ldx2; int n2;
  int d, nbeta;
  double sigma;
  double* K; double* dsigma; double* dbeta; long ld; size_t plane;
  BetaArg beta;
};

__global__ void gram_dtheta_kernel(const DthetaArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j >= a.n2 || i >= a.n1) return;
  double s = 0.0, tot = 0.0;
  double dk[GPX_MAXD];
  for (int k = 0; k < a.d; ++k) {
    const double df = a.X1[(long)i * a.ldx1 + k] - a.X2[(long)j * a.ldx2 + k];
    dk[k] = df * df;
    s = fma(a.beta.b[k], dk[k], s);
    tot += dk[k];
  }
  const double e = exp(-0.5 * s);
  const double kv = a.sigma * e;
  const long o = (long)i * a.ld + j;
  a.K[o] = kv;
  a.dsigma[o] = e;  // K / sigma
  if (a.nbeta == 1) {
    a.dbeta[o] = -0.5 * tot * kv;
  } else {
    for (int k = 0; k < a.d; ++k) a.dbeta[k * a.plane + o] = -0.5 * dk[k] * kv;
  }
}

// ---------------------------------------------------------------------------------------
// grad-trace: sums over the lower triangle of A = alpha alpha^T - Kinv against K and diff_k K.
// Partials per workgroup -> fixed-order final reduction (bitwise reproducible).
// ---------------------------------------------------------------------------------------
// The weights cover a slab of the symmetric matrix: local (r, c) is global (roff + r, coff + c),
// nr x nc local elements, Kinv points at the slab.  The full matrix is roff = coff = 0, nr = nc = n.
struct TraceArgs {
  const double* X; long ldx; int nr, nc, roff, coff; int d; int nbeta;
  double sigma;
  const double* alpha; const double* Kinv; long ldk;
  // upper != 0 (full matrix only, roff = coff = 0): Ky^-1 is stored in the STRICT UPPER triangle of the
  // buffer (element (r, c), c > r, at Kinv[r * ldk + c]) with its diagonal in kdiag -- what
  // gpx_lauum_upper leaves beside W = L^-1 in one n x n buffer.  The matrix is symmetric, so the sums are
  // the same; only the tile tests are mirrored.
  int upper; const double* kdiag;
  double* partials;  // [nblocks][nacc]
  BetaArg sb;        // sqrt(beta_k / 2): the squared differences below are beta_k diff_k^2 / 2 (the host doubles the sums)
  double tab[EXP_TAB];  // 2^(j/64): K_ij / sigma = exp_neg_tab(sum_k beta_k diff_k^2 / 2)
};

constexpr int TR_MAXACC = GPX_MAXD + 2;

// One workgroup owns TR_SUBS row sub-tiles (64 rows each) of one 128-column strip: the per-lane X
// columns and alpha stay in registers across them, the accumulators are reduced once per workgroup
// (a 64 x 128 tile per workgroup spent a third of its time in prologue + reduction and left
// 8x as many partials for the final pass).
template <int D>
struct TraceCfg {
  static constexpr int SUBS = D > 0 ? 8 : 4;                 // LDS: SUBS * 64 * d doubles (32 KiB / 64 KiB)
  static constexpr int DD = D > 0 ? D : GPX_MAXD;
};

template <int D>
__global__ __launch_bounds__(GT_THREADS, (D > 0 ? 3 : 1)) void grad_trace_kernel(const TraceArgs a, int band_row0) {
  constexpr int SUBS = TraceCfg<D>::SUBS, DD = TraceCfg<D>::DD;
  __shared__ double sx1[SUBS * GT_ROWS * DD];
  __shared__ double sal[SUBS * GT_ROWS];
  __shared__ double sred[4][TR_MAXACC];
  __shared__ double stab[EXP_TAB];
  const int d = D > 0 ? D : a.d;
  const int nacc = 2 + a.nbeta;
  const int rb = band_row0 + blockIdx.y * (SUBS * GT_ROWS), c0 = blockIdx.x * GT_COLS;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int bid = blockIdx.y * gridDim.x + blockIdx.x;
  const int rows_here = min(SUBS * GT_ROWS, a.nr - rb);
  // every sub-tile on the unused side of the diagonal: nothing to add
  const bool skip = rows_here <= 0 || (a.upper ? c0 + GT_COLS - 1 < rb : a.coff + c0 > a.roff + rb + rows_here - 1);

  // acc[0]: sum A_ii (diag)   acc[1]: sum w A_ij K_ij   acc[2+k]: sum w A_ij K_ij z-diff_k^2
  double acc[2 + DD];
#pragma unroll
  for (int q = 0; q < 2 + DD; ++q) acc[q] = 0.0;

  if (!skip) {
    if (t < EXP_TAB) stab[t] = a.tab[t];
    for (int e = t; e < rows_here * d; e += GT_THREADS) {
      const int r = e / d, k = e - r * d;
      sx1[e] = a.X[(long)(a.roff + rb + r) * a.ldx + k] * a.sb.b[k];
    }
    for (int e = t; e < rows_here; e += GT_THREADS) sal[e] = a.alpha[a.roff + rb + e];
    const int c = c0 + 2 * lane;          // local column
    const int gc = a.coff + c;            // global column
    double z0[DD], z1[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) {
      if (k < d) {
        z0[k] = c < a.nc ? a.X[(long)gc * a.ldx + k] * a.sb.b[k] : 0.0;
        z1[k] = c + 1 < a.nc ? a.X[(long)(gc + 1) * a.ldx + k] * a.sb.b[k] : 0.0;
      }
    }
    const double al0 = c < a.nc ? a.alpha[gc] : 0.0;
    const double al1 = c + 1 < a.nc ? a.alpha[gc + 1] : 0.0;
    __syncthreads();
    for (int sub = 0; sub < SUBS; ++sub) {
      const int r0 = rb + sub * GT_ROWS;
      if (r0 >= a.nr) break;
      if (a.upper ? c0 + GT_COLS - 1 < r0 : a.coff + c0 > a.roff + r0 + GT_ROWS - 1) continue;   // tile on the unused side
      const double* xs = sx1 + sub * GT_ROWS * d;
      const double* as = sal + sub * GT_ROWS;
      // interior: whole tile strictly off the diagonal on the stored side and inside the slab -> weight 2, no predicates
      const bool interior = (a.upper ? c0 >= r0 + GT_ROWS : a.roff + r0 >= a.coff + c0 + GT_COLS) &&
                            r0 + GT_ROWS <= a.nr && c0 + GT_COLS <= a.nc;
      if (interior) {
        const double* src = a.Kinv + (long)(r0 + wid * 16) * a.ldk + c;
        // The accumulators acc[1..] are kept in HALF units (off-diagonal elements weigh 1, diagonal ones 1/2) and
        // doubled once before the reduction -- exact in binary floating point -- so the weight 2 of the symmetric
        // sum costs nothing per element; the products go into the accumulators by fma.  ~62 VALU instructions per
        // element at d = 8 (74 in round 2); the kernel is bound by their issue rate.  Compiled for 3 waves per SIMD (168
        // VGPRs, no spills): n = 65536 5.27 ms against 6.2 ms at 2 waves per SIMD; forcing 4 (128 VGPRs) spills -- 20 ms.
        // (the RBF / ARD choice is made OUTSIDE the row loop: as a branch inside it the compiler carried both sets
        // of accumulators through the loop with ~20 register moves per element)
        v2d qn = *reinterpret_cast<const v2d*>(src);
        if (a.nbeta == 1) {
#pragma unroll 1
          for (int rr = 0; rr < 16; ++rr) {
            const v2d q = qn;
            src += a.ldk;
            if (rr < 15) qn = *reinterpret_cast<const v2d*>(src);   // next row's pair in flight under this row's arithmetic
            const int r = wid * 16 + rr;
            const double ai = as[r];
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int k = 0; k < DD; ++k) {
              if (k < d) {
                const double x = xs[r * d + k];
                const double d0 = x - z0[k], d1 = x - z1[k];
                s0 = fma(d0, d0, s0);
                s1 = fma(d1, d1, s1);
              }
            }
            const double w0 = fma(ai, al0, -q[0]) * exp_neg_tab(s0, stab);
            const double w1 = fma(ai, al1, -q[1]) * exp_neg_tab(s1, stab);
            acc[1] += w0 + w1;
            acc[2] = fma(w1, s1, fma(w0, s0, acc[2]));
          }
        } else {
#pragma unroll 1
          for (int rr = 0; rr < 16; ++rr) {
            const v2d q = qn;
            src += a.ldk;
            if (rr < 15) qn = *reinterpret_cast<const v2d*>(src);   // next row's pair in flight under this row's arithmetic
            const int r = wid * 16 + rr;
            const double ai = as[r];
            double s0 = 0.0, s1 = 0.0;
            double e0[DD], e1[DD];
#pragma unroll
            for (int k = 0; k < DD; ++k) {
              if (k < d) {
                const double x = xs[r * d + k];
                const double d0 = x - z0[k], d1 = x - z1[k];
                e0[k] = d0 * d0;
                e1[k] = d1 * d1;
                s0 += e0[k];
                s1 += e1[k];
              }
            }
            const double w0 = fma(ai, al0, -q[0]) * exp_neg_tab(s0, stab);
            const double w1 = fma(ai, al1, -q[1]) * exp_neg_tab(s1, stab);
            acc[1] += w0 + w1;
#pragma unroll
            for (int k = 0; k < DD; ++k)
              if (k < d) acc[2 + k] = fma(w1, e1[k], fma(w0, e0[k], acc[2 + k]));
          }
        }
        continue;
      }
      for (int rr = 0; rr < 16; ++rr) {
        const int r = wid * 16 + rr;
        const int lr = r0 + r;
        if (lr >= a.nr) break;
        const int gr = a.roff + lr;
        // which of the lane's two columns lie on the stored side of (or on) the diagonal, inside the slab
        const bool v0 = c < a.nc && (a.upper ? gc >= gr : gc <= gr);
        const bool v1 = c + 1 < a.nc && (a.upper ? gc + 1 >= gr : gc + 1 <= gr);
        if (!v0 && !v1) continue;
        const double* src = a.Kinv + (long)lr * a.ldk + c;
        double q0 = 0.0, q1 = 0.0;
        if (a.upper) {          // diagonal entries come from kdiag, the others from the strict upper triangle
          if (v0) q0 = gc == gr ? a.kdiag[gr] : src[0];
          if (v1) q1 = gc + 1 == gr ? a.kdiag[gr] : src[1];
        } else {
          if (v0) q0 = src[0];
          if (v1) q1 = src[1];
        }
        const double ai = as[r];
        double s0 = 0.0, s1 = 0.0;
        double e0[DD], e1[DD];
#pragma unroll
        for (int k = 0; k < DD; ++k) {
          if (k < d) {
            const double x = xs[r * d + k];
            const double d0 = x - z0[k], d1 = x - z1[k];
            e0[k] = d0 * d0;
            e1[k] = d1 * d1;
            s0 += e0[k];
            s1 += e1[k];
          }
        }
        const double A0 = v0 ? ai * al0 - q0 : 0.0;
        const double A1 = v1 ? ai * al1 - q1 : 0.0;
        const double w0 = v0 ? (gc == gr ? 0.5 : 1.0) * A0 * exp_neg_tab(s0, stab) : 0.0;   // half units, see above
        const double w1 = v1 ? (gc + 1 == gr ? 0.5 : 1.0) * A1 * exp_neg_tab(s1, stab) : 0.0;
        if (v0 && gc == gr) acc[0] += A0;
        if (v1 && gc + 1 == gr) acc[0] += A1;
        acc[1] += w0 + w1;
        if (a.nbeta == 1) {
          acc[2] += w0 * s0 + w1 * s1;  // RBF: sum_k beta (x-x')^2 ; divided by beta on the host
        } else {
#pragma unroll
          for (int k = 0; k < DD; ++k)
            if (k < d) acc[2 + k] += w0 * e0[k] + w1 * e1[k];
        }
      }
    }
  }
  // wave reduction (fixed order), then 4 waves through LDS
  for (int q = 0; q < nacc; ++q) {
    double v = q == 0 ? acc[q] : 2.0 * acc[q];   // acc[1..] were kept in half units
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane == 0) sred[wid][q] = v;
  }
  __syncthreads();
  if (t < nacc) a.partials[(size_t)bid * nacc + t] = (sred[0][t] + sred[1][t]) + (sred[2][t] + sred[3][t]);
}

// Final fixed-order reduction of [nblocks][nacc] partials -> out[nacc]: one workgroup per accumulator
// (they run side by side; one workgroup doing all of them in turn took 3.8 ms at n = 65536).
// accumulate != 0 adds to out (the later row bands of a launch split to fit the scratch).
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const double* partials, size_t nblocks,
                                                                int nacc, double* out, int accumulate) {
  __shared__ double sm[1024];
  const int t = threadIdx.x;
  const int q = blockIdx.x;
  double v = 0.0;
  for (size_t b = t; b < nblocks; b += 1024) v += partials[b * nacc + q];
  sm[t] = v;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (t < s) sm[t] += sm[t + s];
    __syncthreads();
  }
  if (t == 0) out[q] = accumulate ? out[q] + sm[0] : sm[0];
}

template <typename F>
void dispatch_d(int d, F&& f) {
  switch (d) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    default: f(std::integral_constant<int, 0>{}); break;
  }
}

}  // namespace

int gpx_gram_launch(hipStream_t s, const double* X1, int n1, long ldx1, const double* X2, int n2,
                    long ldx2, int d, double sigma, const BetaArg& hb, double diag_add, double* K,
                    long ldk, int uplo) {
  if (n1 <= 0 || n2 <= 0) return 0;
  GramArgs a;
  a.X1 = X1; a.ldx1 = ldx1; a.n1 = n1;
  a.X2 = X2; a.ldx2 = ldx2; a.n2 = n2;
  a.d = d; a.sigma = sigma; a.diag_add = diag_add; a.K = K; a.ldk = ldk; a.lower = uplo == GPX_LOWER;
  // The matrix is written once and not read back by this kernel; non-temporal stores (meant to keep the streamed
  // lines from evicting the rows of X that every workgroup re-reads through its XCD's L2) were measured and changed
  // nothing: 3.771 against 3.778 ms lower, 7.06 against 7.02 ms full at n = 65536.  Off by default.
  static const int nt = [] { const char* e = getenv("GPX_GRAM_NT"); return e ? atoi(e) : 0; }();   // development switch
  a.nt = nt;
  for (int k = 0; k < GPX_MAXD; ++k) a.sb.b[k] = k < d ? sqrt(0.5 * hb.b[k]) : 0.0;
  for (int j = 0; j < EXP_TAB; ++j) a.tab[j] = (double)((long double)sigma * exp2l((long double)j / EXP_TAB));
  dispatch_d(d, [&](auto D) {
    constexpr int DV = decltype(D)::value;
    // 512-column strips for full matrices where the per-lane columns fit in registers (compiled d) and the
    // matrix is wide enough to fill them (n = 65536, d = 8: 6.78 against 6.90 ms); 128-column strips
    // otherwise: runtime d <= 32, narrow matrices such as k*, and lower-triangle launches, whose tiles on
    // the diagonal take the predicated path and are 4x larger with wide strips (no gain measured)
    if (DV > 0 && n2 >= 4 * GT_COLS && !a.lower) {
      dim3 grid((n2 + 4 * GT_COLS - 1) / (4 * GT_COLS), (n1 + GT_SUBS * GT_ROWS - 1) / (GT_SUBS * GT_ROWS));
      hipLaunchKernelGGL((gram_kernel<DV, (DV > 0 ? 4 : 1)>), grid, dim3(GT_THREADS), 0, s, a);
    } else {
      dim3 grid((n2 + GT_COLS - 1) / GT_COLS, (n1 + GT_SUBS * GT_ROWS - 1) / (GT_SUBS * GT_ROWS));
      if (a.lower) grid = dim3(grid.y * (grid.y + 1), 1);   // packed: live (row block, strip) pairs, see the kernel
      hipLaunchKernelGGL((gram_kernel<DV, 1>), grid, dim3(GT_THREADS), 0, s, a);
    }
  });
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_gram_dtheta_launch(hipStream_t s, const double* X1, int n1, long ldx1, const double* X2,
                           int n2, long ldx2, int d, int nbeta, double sigma, const BetaArg& hb,
                           double* K, double* dsigma, double* dbeta, long ld, size_t plane) {
  if (n1 <= 0 || n2 <= 0) return 0;
  DthetaArgs a;
  a.X1 = X1; a.ldx1 = ldx1; a.n1 = n1; a.X2 = X2; a.ldx2 = ldx2; a.n2 = n2;
  a.d = d; a.nbeta = nbeta; a.sigma = sigma; a.K = K; a.dsigma = dsigma; a.dbeta = dbeta;
  a.ld = ld; a.plane = plane; a.beta = hb;
  // grid.y carries the row: at most 65535 per launch, so a tall matrix goes out in row chunks
  constexpr int ROWS_PER_LAUNCH = 32768;
  for (int r0 = 0; r0 < n1; r0 += ROWS_PER_LAUNCH) {
    DthetaArgs c = a;
    c.n1 = n1 - r0 < ROWS_PER_LAUNCH ? n1 - r0 : ROWS_PER_LAUNCH;
    c.X1 = X1 + (long)r0 * ldx1;
    c.K = K + (long)r0 * ld;
    c.dsigma = dsigma + (long)r0 * ld;
    c.dbeta = dbeta + (long)r0 * ld;
    dim3 grid((n2 + 255) / 256, c.n1);
    hipLaunchKernelGGL(gram_dtheta_kernel, grid, dim3(256), 0, s, c);
  }
  GPX_HIP(hipGetLastError());
  return 0;
}

int gpx_grad_trace_launch(hipStream_t s, const double* X, long ldx, int d, int nbeta, double sigma,
                          const BetaArg& hb, const double* alpha, const double* Kinv, long ldk, int roff,
                          int coff, int nr, int nc, double* red, size_t red_elems, double* out_dev,
                          const double* kdiag) {
  TraceArgs a;
  a.upper = kdiag != nullptr; a.kdiag = kdiag;
  a.X = X; a.ldx = ldx; a.nr = nr; a.nc = nc; a.roff = roff; a.coff = coff; a.d = d; a.nbeta = nbeta;
  a.sigma = sigma; a.alpha = alpha; a.Kinv = Kinv; a.ldk = ldk; a.partials = red;
  for (int k = 0; k < GPX_MAXD; ++k) a.sb.b[k] = k < d ? sqrt(0.5 * hb.b[k]) : 0.0;
  for (int j = 0; j < EXP_TAB; ++j) a.tab[j] = (double)exp2l((long double)j / EXP_TAB);
  const int nacc = 2 + nbeta;
  int rc = 0;
  dispatch_d(d, [&](auto D) {
    constexpr int DV = decltype(D)::value;
    constexpr int rows_per_block = TraceCfg<DV>::SUBS * GT_ROWS;
    const unsigned gx = (nc + GT_COLS - 1) / GT_COLS;
    const size_t gy_all = (nr + rows_per_block - 1) / rows_per_block;
    // row bands: as many block rows per launch as the partials scratch holds (one band unless n * d is huge)
    size_t gy_band = red_elems / ((size_t)gx * nacc);
    if (gy_band == 0) {
      gpx_set_error("grad_trace: reduction scratch too small for a strip of %d columns", nc);
      rc = GPX_ENOMEM;
      return;
    }
    static const long band_cap = [] {
      const char* e = getenv("GPX_TRACE_BAND_ROWS");   // development/test override: block rows per band
      return e ? atol(e) : 0L;
    }();
    if (band_cap > 0 && (size_t)band_cap < gy_band) gy_band = (size_t)band_cap;
    if (gy_band > gy_all) gy_band = gy_all;
    for (size_t y0 = 0; y0 < gy_all; y0 += gy_band) {
      const unsigned gy = (unsigned)(gy_all - y0 < gy_band ? gy_all - y0 : gy_band);
      hipLaunchKernelGGL((grad_trace_kernel<DV>), dim3(gx, gy), dim3(GT_THREADS), 0, s, a,
                         (int)(y0 * rows_per_block));
      hipLaunchKernelGGL(reduce_partials_kernel, dim3(nacc), dim3(1024), 0, s, red, (size_t)gx * gy, nacc, out_dev,
                         y0 > 0 ? 1 : 0);
    }
  });
  if (rc != 0) return rc;
  GPX_HIP(hipGetLastError());
  return 0;
}
