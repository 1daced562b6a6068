// fp64 GEMM on the CDNA4 matrix cores (v_mfma_f64_16x16x4_f64), the engine under every O(n^3)
// step of the GP path: the SYRK/GEMM trailing updates of the Cholesky (regression.py:36,49),
// the TRSM panels (regression.py:38), V^T V (regression.py:41), trtri/lauum (regression.py:60).
//
//   D = alpha * op(A) op(B) + beta * Cin (+ diag_add on the diagonal),  row-major C/D.
//
// Tiling: 128 x 128 output tile per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave =
// 4 x 4 MFMA tiles, 128 accumulator VGPRs), BK = 16.  Each operand k-tile is staged through LDS
// in the orientation it has in HBM (no transposition at staging time):
//   KC operands (k contiguous in memory): LDS [128 rows][18]  -> fragment read [row][k]
//   RC operands (row index contiguous)  : LDS [16 k][144]     -> fragment read [k][row]
// Both paddings make the ds_read_b64 fragment reads and the ds_write_b128 staging writes
// bank-conflict free (bank = (addr/4) % 64 per 32-lane half, MI355X_MICROARCH.md section LDS).
// Global loads are 16 B per lane, register-staged and issued one k-tile ahead of the MFMAs.
//
// Triangular operands (a_tri / b_tri: the STORED matrix is lower triangular and starts on its
// diagonal) restrict the k range per tile and zero-mask inside the boundary tile, so blocks that
// lie strictly above the diagonal are never read (they may hold anything, including NaN).
#include "gpx_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128, BK = 16, NTHREADS = 256;
constexpr int LDK = BK + 2;    // 18 doubles: KC row stride in LDS
constexpr int LDR = 128 + 16;  // 144 doubles: RC row stride in LDS
constexpr int TILE_ELEMS = 128 * LDK;  // == 16 * LDR == 2304
static_assert(128 * LDK == BK * LDR, "both LDS images have the same size");

// One operand's 128 x 16 k-tile: 4 x 16-byte pieces per thread.
template <int LAY>
struct Stage {
  v2d v[4];

  // P: operand base; ld: leading dim; r0: first op-row (m or n) of the tile; R: extent of that
  // index; k0: first k of the tile; K: k extent (already clipped to the tile's k range end).
  __device__ __forceinline__ void load(const double* __restrict__ P, long ld, int r0, int R, int k0,
                                       int K, int tri, int t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int r, k;
      bool v0, v1;
      const double* src;
      if (LAY == GPX_LAY_KC) {
        const int p = t & 7, rr = (t >> 3) + 32 * q;
        r = r0 + rr;
        k = k0 + 2 * p;
        const bool rin = r < R;
        v0 = rin && (k < K) && (!tri || k <= r);
        v1 = rin && (k + 1 < K) && (!tri || k + 1 <= r);
        src = P + (long)r * ld + k;
      } else {
        const int pr = t & 63, kr = (t >> 6) + 4 * q;
        r = r0 + 2 * pr;
        k = k0 + kr;
        const bool kin = k < K;
        v0 = kin && (r < R) && (!tri || r <= k);
        v1 = kin && (r + 1 < R) && (!tri || r + 1 <= k);
        src = P + (long)k * ld + r;
      }
      v2d x = {0.0, 0.0};
      if (v1) {  // v1 implies v0 (valid sets are prefixes along the contiguous index)
        x = *reinterpret_cast<const v2d*>(src);
      } else if (v0) {
        x[0] = src[0];
      }
      v[q] = x;
    }
  }

  __device__ __forceinline__ void store(double* __restrict__ S, int t) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int off;
      if (LAY == GPX_LAY_KC) {
        const int p = t & 7, rr = (t >> 3) + 32 * q;
        off = rr * LDK + 2 * p;
      } else {
        const int pr = t & 63, kr = (t >> 6) + 4 * q;
        off = kr * LDR + 2 * pr;
      }
      *reinterpret_cast<v2d*>(S + off) = v[q];
    }
  }
};

template <int LAY>
__device__ __forceinline__ double frag(const double* __restrict__ S, int tile, int kk, int i16, int g4) {
  if (LAY == GPX_LAY_KC) return S[(tile * 16 + i16) * LDK + kk * 4 + g4];
  return S[(kk * 4 + g4) * LDR + tile * 16 + i16];
}

template <int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_f64_kernel(const GemmDesc g) {
  __shared__ __attribute__((aligned(16))) double smem[2 * TILE_ELEMS];
  double* sA = smem;
  double* sB = smem + TILE_ELEMS;

  const int bi = blockIdx.y, bj = blockIdx.x;
  if (g.c_lower && bj > bi) return;
  const int i0 = bi * BM, j0 = bj * BN;
  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  const int i16 = lane & 15, g4 = lane >> 4;

  int kb = 0, ke = g.K;
  if (g.a_tri) {
    if (ALAY == GPX_LAY_KC) ke = min(ke, i0 + BM); else kb = max(kb, i0);
  }
  if (g.b_tri) {
    if (BLAY == GPX_LAY_KC) ke = min(ke, j0 + BN); else kb = max(kb, j0);
  }

  v4d acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  if (kb < ke) {
    Stage<ALAY> stA;
    Stage<BLAY> stB;
    stA.load(g.A, g.lda, i0, g.M, kb, ke, g.a_tri, t);
    stB.load(g.B, g.ldb, j0, g.N, kb, ke, g.b_tri, t);
    for (int k0 = kb; k0 < ke; k0 += BK) {
      __syncthreads();  // everyone finished reading the previous tile
      stA.store(sA, t);
      stB.store(sB, t);
      __syncthreads();
      if (k0 + BK < ke) {  // next tile's loads fly under this tile's MFMAs
        stA.load(g.A, g.lda, i0, g.M, k0 + BK, ke, g.a_tri, t);
        stB.load(g.B, g.ldb, j0, g.N, k0 + BK, ke, g.b_tri, t);
      }
#pragma unroll
      for (int kk = 0; kk < BK / 4; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          a[x] = frag<ALAY>(sA, wr * 4 + x, kk, i16, g4);
          b[x] = frag<BLAY>(sB, wc * 4 + x, kk, i16, g4);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 4; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
      }
    }
  }

  // epilogue: C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int x = 0; x < 4; ++x) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i0 + wr * 64 + x * 16 + g4 + 4 * r;
      if (row >= g.M) continue;
#pragma unroll
      for (int y = 0; y < 4; ++y) {
        const int col = j0 + wc * 64 + y * 16 + i16;
        if (col >= g.N) continue;
        if (g.c_lower && col > row) continue;
        double v = alpha * acc[x][y][r];
        if (beta != 0.0) v += beta * g.Cin[(long)row * g.ldcin + col];
        if (row == col) v += g.diag_add;
        g.D[(long)row * g.ldd + col] = v;
      }
    }
  }
}

}  // namespace

int gpx_gemm_launch(hipStream_t s, const GemmDesc& g) {
  if (g.M <= 0 || g.N <= 0) return 0;
  GPX_REQUIRE(g.K >= 0, "gemm: negative K");
  GPX_REQUIRE(g.A && g.B && g.D, "gemm: null operand");
  GPX_REQUIRE(g.beta == 0.0 || g.Cin, "gemm: beta != 0 needs Cin");
  GPX_REQUIRE(gpx_aligned16(g.A) && gpx_aligned16(g.B), "gemm: operands must be 16-byte aligned");
  GPX_REQUIRE((g.lda & 1) == 0 && (g.ldb & 1) == 0, "gemm: leading dimensions must be even");
  GPX_REQUIRE(g.lda >= (g.a_lay == GPX_LAY_KC ? g.K : g.M), "gemm: lda too small");
  GPX_REQUIRE(g.ldb >= (g.b_lay == GPX_LAY_KC ? g.K : g.N), "gemm: ldb too small");
  GPX_REQUIRE(g.ldd >= g.N, "gemm: ldd too small");
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM);
  dim3 block(NTHREADS);
  if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_KC)
    hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_KC>), grid, block, 0, s, g);
  else if (g.a_lay == GPX_LAY_KC && g.b_lay == GPX_LAY_RC)
    hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_KC, GPX_LAY_RC>), grid, block, 0, s, g);
  else if (g.a_lay == GPX_LAY_RC && g.b_lay == GPX_LAY_RC)
    hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_RC>), grid, block, 0, s, g);
  else
    hipLaunchKernelGGL((gemm_f64_kernel<GPX_LAY_RC, GPX_LAY_KC>), grid, block, 0, s, g);
  GPX_HIP(hipGetLastError());
  return 0;
}
