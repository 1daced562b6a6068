// EXPERIMENT, measured and not kept (round 3): one step of the right-looking panel factorisation
// of potrf_panels (gpx_api.hip) as ONE launch -- leaf, panel solve, update of the next panel's columns and the previous
// panel's update of everything behind them, handed over between workgroups inside the launch through agent-scope
// flags (tools/flag_hop.hip: 1.5 us per hand-over) -- instead of three dependent launches per 128 columns.
// It is correct (the library's potrf tests pass with it) and no faster: potrf(512 ... 16384) 0.200 / 0.418 / 0.855 /
// 2.13 / 7.13 / 31.7 ms against 0.21 / 0.42 / 0.83 / 2.08 / 6.94 / 31.2 ms.  The stamps this tool prints say why
// (step 1 of n = 1024, microseconds after the leaf's start): leaf 36.4 (30 stand-alone: 4.4 of it the write-through
// store of the inverse and its acknowledgement), first 32-row solve done at 44.0 (its longest wave is a chain of 64
// dependent MFMAs), first tile of the next panel's columns stored at 52.1, next launch at ~55 -- the hand-overs and
// the single-CU MFMA rate (0.3 TFLOP/s: 3.4 us for a 64 x 64 x 128 tile) cost what the two launch boundaries did.  And
// with 152 KiB of LDS for the leaf role the kernel runs one workgroup per CU, so the trailing tiles (10 us each, load
// latency exposed) take 99 us at step 1 of n = 4096 where the library's GEMM takes 19.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igaussian_process_amd/csrc
//         tools/potrf_step_experiment.hip -o tools/potrf_step_experiment ;  ./tools/potrf_step_experiment [n] [step]
#define GPX_STEP_PROFILE 1
#include "../gaussian_process_amd/csrc/leaf.hip"

constexpr int GPX_STEP_FLAGS = 1024;  // flag 0: the leaf; 1 + b: 32-row block b of the panel; word GPX_STEP_FLAGS: timeout

namespace {

// ---------------------------------------------------------------------------------------------------------------
// One step of the right-looking panel factorisation as ONE launch (potrf_panels in gpx_api.hip; the dpotrf of
// regression.py:36,49 for the diagonal blocks of order <= 4096, where the chain of dependent launches, not the
// flops, is the cost: leaf, panel solve and rank-128 update were 3 launches = 65 us per 128 columns for a 30 us leaf).
// Step k, panel columns [j0, j0 + 128), workgroup roles in block-id order:
//   0            leaf: factor + invert the diagonal block (k, k), as leaf_blocked_kernel; W_kk goes to dinv by
//                agent-scope stores, then flag 0 is raised
//   Ub tiles     trailing 64 x 64 tiles of the columns >= j0 + 256: -= panel k-1 (written by the previous launch).  No
//                dependency inside the launch: they run under the leaf
//   PS blocks    32 rows of the panel below the diagonal block: X <- X W_kk^T in place.  X is staged in LDS first, then
//                the workgroup waits for flag 0, reads its rows of W_kk as MFMA fragments straight from memory
//                (agent-scope loads), writes L by agent-scope stores and raises its own flag
//   Ua tiles     64 x 64 tiles of the NEXT panel's columns [j0 + 128, j0 + 256): -= panel k-1 (no wait), then wait for
//                the PS flags of the tile's rows and columns and -= panel k (agent-scope loads)
// so that after the launch everything the next step reads has absorbed panels <= k, and the dependent chain of a step
// is leaf -> one 32-row solve -> one tile update -> end of launch.  A waiting workgroup only ever waits for workgroups
// with LOWER block ids, which are dispatched first and never wait for a higher one, so the launch cannot deadlock
// however few workgroups are resident; every wait is bounded all the same (a timeout raises *err and the workgroup
// goes on, so the launch always drains).  Hand-over between workgroups on different XCDs: agent-scope relaxed atomic
// accesses for the payload and the flag (write-through stores, L2-bypassing loads), no cache-wide fence -- 1.5 us per
// hop for a 16 KiB payload against 3.6 us with __threadfence() (tools/flag_hop.hip).  The flags are never cleared:
// a launch raises them to its own epoch (a counter of the handle).
#ifdef GPX_STEP_PROFILE   // tools/step_profile.hip: 100 MHz wall-clock stamps of the roles, thread 0
__device__ long long step_prof[64];
#define STEP_TICK(k) do { if (threadIdx.x == 0) step_prof[k] = wall_clock64(); } while (0)
#else
#define STEP_TICK(k)
#endif

constexpr int STEP_RB = 32;     // rows per panel-solve workgroup
constexpr int STEP_T = 64;      // update tile

struct StepArgs {
  double* A; long lda; int n;   // the block potrf_panels factors (order n, lower)
  int j0;                       // first column of panel k
  double* dinv_k;               // inverse of L_kk (128 x 128, row stride 128)
  int* info; int off;           // LAPACK info word, global index of A[0][0]
  unsigned* flags; unsigned epoch; int* err;
  int n_ub, n_ps, n_ua;         // workgroups per role
  int m_ub;                     // 64-blocks per side of the Ub triangle
  int has_prev;                 // panel k-1 exists (k >= 1)
};

__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void wait_flag(const unsigned* f, unsigned epoch, int* err) {
  unsigned spins = 0;
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > (1u << 21)) { *err = 1; break; }
  }
}

// all stores of the workgroup acknowledged, then thread 0 raises the flag
__device__ __forceinline__ void raise_flag(unsigned* f, unsigned epoch) {
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// dst[i][0 .. 128) (row stride LD) <- A[r0 + i][c .. c + 128) for i < ROWS, rows >= n as zeros.  FRESH: the rows were
// written inside this launch (agent-scope loads); otherwise 16-byte loads.
template <int ROWS, bool FRESH>
__device__ __forceinline__ void load_rows(double* dst, const double* A, long lda, int n, int r0, int c) {
  constexpr int PER = ROWS * 64 / THREADS;
  v2d v[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    v[u] = v2d{0.0, 0.0};
    if (r0 + i < n) {
      const double* p = A + (long)(r0 + i) * lda + c + j;
      if (FRESH) { v[u].x = ld_agent(p); v[u].y = ld_agent(p + 1); }
      else v[u] = *reinterpret_cast<const v2d*>(p);
    }
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    *reinterpret_cast<v2d*>(dst + i * LD + j) = v[u];
  }
}

// acc[c] -= Ar(row tile rt) * Br(col tile ct0 + c)^T over K = 128, operands in LDS (row stride LD)
__device__ __forceinline__ void tile_rank128(const double* Ar, const double* Br, int rt, int ct0, v4d acc[2], int lane) {
  const double* pa = Ar + (rt * TS + (lane & 15)) * LD + (lane >> 4);
  const double* pb0 = Br + (ct0 * TS + (lane & 15)) * LD + (lane >> 4);
  const double* pb1 = pb0 + TS * LD;
#pragma unroll 8
  for (int s = 0; s < NB / 4; ++s) {
    const double a = -pa[4 * s];
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb0[4 * s], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, pb1[4 * s], acc[1], 0, 0, 0);
  }
}

// 64 x 64 tile at (r0, c0) of A: -= panel at columns jprev (if >= 0, plain loads), then, if jcur >= 0, wait for the
// flags of the tile's row and column blocks and -= panel at columns jcur (agent-scope loads).  S: 128 x LD doubles.
__device__ void update_tile(double* S, const StepArgs& a, int r0, int c0, int jprev, int jcur, int pb) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int rt = wid >> 1, ct0 = 2 * (wid & 1);
  double* Ar = S;
  double* Br = r0 == c0 ? S : S + STEP_T * LD;
  v4d acc[2];
  (void)pb;
  if (pb >= 0) STEP_TICK(pb);
  const int col0 = c0 + ct0 * TS + (lane & 15);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = r0 + rt * TS + 4 * r + (lane >> 4), gc = col0 + c * TS;
      acc[c][r] = (gr < a.n && gc < a.n) ? a.A[(long)gr * a.lda + gc] : 0.0;
    }
  if (jprev >= 0) {
    load_rows<STEP_T, false>(Ar, a.A, a.lda, a.n, r0, jprev);
    if (r0 != c0) load_rows<STEP_T, false>(Br, a.A, a.lda, a.n, c0, jprev);
    __syncthreads();
    tile_rank128(Ar, Br, rt, ct0, acc, lane);
    __syncthreads();
  }
  if (pb >= 0) STEP_TICK(pb + 1);
  if (jcur >= 0) {
    // PS block b covers rows jcur + 128 + 32 b ..: two blocks per 64 rows
    const int first = jcur + NB;
    if (t < 4) {
      const int base = t < 2 ? r0 : c0;
      const int b = (base - first) / STEP_RB + (t & 1);
      if (b < a.n_ps) wait_flag(a.flags + 1 + b, a.epoch, a.err);
    }
    __syncthreads();
    if (pb >= 0) STEP_TICK(pb + 2);
    load_rows<STEP_T, true>(Ar, a.A, a.lda, a.n, r0, jcur);
    if (r0 != c0) load_rows<STEP_T, true>(Br, a.A, a.lda, a.n, c0, jcur);
    __syncthreads();
    if (pb >= 0) STEP_TICK(pb + 3);
    tile_rank128(Ar, Br, rt, ct0, acc, lane);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = r0 + rt * TS + 4 * r + (lane >> 4), gc = col0 + c * TS;
      if (gr < a.n && gc < a.n) a.A[(long)gr * a.lda + gc] = acc[c][r];
    }
  if (pb >= 0) STEP_TICK(pb + 4);
}

// rows [r0, r0 + 32) of the panel: X <- X W_kk^T in place
__device__ void solve_rows(double* S, const StepArgs& a, int b) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int r0 = a.j0 + NB + b * STEP_RB;
  const int pb = b == 0 ? 8 : (b == a.n_ps - 1 ? 32 : -1);
  (void)pb;
  if (pb >= 0) STEP_TICK(pb);
  load_rows<STEP_RB, false>(S, a.A, a.lda, a.n, r0, a.j0);
  if (pb >= 0) STEP_TICK(pb + 1);
  if (t == 0) wait_flag(a.flags, a.epoch, a.err);
  __syncthreads();
  if (pb >= 0) STEP_TICK(pb + 2);
  // output columns 16 wid .. 16 wid + 15: W rows 16 wid + (lane & 15), k <= 16 wid + 15 (zeros beyond the diagonal)
  double wf[NB / 4];
  const double* pw = a.dinv_k + (wid * TS + (lane & 15)) * NB + (lane >> 4);
#pragma unroll
  for (int s = 0; s < NB / 4; ++s) wf[s] = s < 4 * (wid + 1) ? ld_agent(pw + 4 * s) : 0.0;
  if (pb >= 0) STEP_TICK(pb + 3);
#pragma unroll
  for (int rt = 0; rt < STEP_RB / TS; ++rt) {
    const double* px = S + (rt * TS + (lane & 15)) * LD + (lane >> 4);
    v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NB / 4; ++s)
      if (s < 4 * (wid + 1)) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(px[4 * s], wf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gr = r0 + rt * TS + 4 * r + (lane >> 4);
      if (gr < a.n) st_agent(a.A + (long)gr * a.lda + a.j0 + wid * TS + (lane & 15), acc[r]);
    }
  }
  if (pb >= 0) STEP_TICK(pb + 4);
  raise_flag(a.flags + 1 + b, a.epoch);
  if (pb >= 0) STEP_TICK(pb + 5);
}

__device__ __forceinline__ void store_dinv_agent(const double* S, double* dinv) {
  constexpr int PER = NB * NB / 2 / THREADS;
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int q = threadIdx.x + THREADS * u;
    const int i = q >> 6, j = (q & 63) * 2;
    v2d w = *reinterpret_cast<const v2d*>(S + i * LD + j);
    if (j > i) w.x = 0.0;
    if (j + 1 > i) w.y = 0.0;
    st_agent(dinv + i * NB + j, w.x);
    st_agent(dinv + i * NB + j + 1, w.y);
  }
}

__global__ __launch_bounds__(THREADS) void potrf_step_kernel(const StepArgs a) {
  __shared__ double S[NB * LD];
  __shared__ double Wall[NT * TS * WS];
  __shared__ int fail;
  const int t = threadIdx.x;
  int id = blockIdx.x;
  if (id == 0) {
    double* Ajj = a.A + (long)a.j0 * a.lda + a.j0;
    if (t == 0) fail = 0;
    STEP_TICK(0);
    load_block(S, Ajj, a.lda, NB);
    __syncthreads();
    STEP_TICK(1);
    factor_lower(S, Wall, &fail);
    STEP_TICK(2);
    store_block(S, Ajj, a.lda, NB);
    if (t == 0 && fail != 0 && *a.info == 0) *a.info = a.off + a.j0 + fail;
    __syncthreads();
    invert_lower(S, Wall);
    STEP_TICK(3);
    store_dinv_agent(S, a.dinv_k);
    raise_flag(a.flags, a.epoch);
    STEP_TICK(4);
    return;
  }
  id -= 1;
  if (id < a.n_ub) {
    // lower triangle of m_ub x m_ub tiles starting at column j0 + 256; id = rb (rb + 1) / 2 + cb
    int rb = (int)((sqrtf(8.0f * (float)id + 1.0f) - 1.0f) * 0.5f);
    while ((rb + 1) * (rb + 2) / 2 <= id) ++rb;
    while (rb * (rb + 1) / 2 > id) --rb;
    const int cb = id - rb * (rb + 1) / 2;
    const int base = a.j0 + 2 * NB;
    update_tile(S, a, base + rb * STEP_T, base + cb * STEP_T, a.j0 - NB, -1, id == 0 ? 24 : (id == a.n_ub - 1 ? 48 : -1));
    return;
  }
  id -= a.n_ub;
  if (id < a.n_ps) {
    solve_rows(S, a, id);
    return;
  }
  id -= a.n_ps;
  {
    // tiles of the columns [j0 + 128, j0 + 256): column half h, row blocks rb >= h; id = 2 rb - (rb > 0) + h  (rb = 0: h = 0)
    const int rb = id == 0 ? 0 : (id + 1) / 2, h = id == 0 ? 0 : (id + 1) & 1;
    const int base = a.j0 + NB;
    if (base + h * STEP_T < a.n)
      update_tile(S, a, base + rb * STEP_T, base + h * STEP_T, a.has_prev ? a.j0 - NB : -1, a.j0, id == 0 ? 16 : (id == a.n_ua - 1 ? 40 : -1));
  }
}

}  // namespace

// One step of potrf_panels (see potrf_step_kernel).  Requires a full panel: j0 + 128 <= n, and n - j0 <= 4096.
int gpx_potrf_step_launch(hipStream_t s, int n, int j0, double* A, long lda, double* dinv_k, int* info_dev, int off,
                          unsigned* flags, unsigned epoch, int* err_dev) {
  GPX_REQUIRE(j0 >= 0 && j0 % NB == 0 && j0 + NB <= n && n - j0 <= 4096 + NB && lda % 2 == 0, "potrf step: bad panel");
  StepArgs a;
  a.A = A; a.lda = lda; a.n = n; a.j0 = j0; a.dinv_k = dinv_k; a.info = info_dev; a.off = off;
  a.flags = flags; a.epoch = epoch; a.err = err_dev;
  a.has_prev = j0 >= NB;
  const int rem = n - j0 - NB;                       // rows below the diagonal block
  a.n_ps = (rem + STEP_RB - 1) / STEP_RB;
  const int nrb = (rem + STEP_T - 1) / STEP_T;       // 64-row blocks below
  a.n_ua = nrb == 0 ? 0 : 2 * nrb - 1;               // (rb, h) with rb >= h, h < 2
  const int rem2 = rem - NB;                         // side of the Ub triangle
  a.m_ub = a.has_prev && rem2 > 0 ? (rem2 + STEP_T - 1) / STEP_T : 0;
  a.n_ub = a.m_ub * (a.m_ub + 1) / 2;
  GPX_REQUIRE(a.n_ps <= GPX_STEP_FLAGS - 1, "potrf step: too many row blocks");
  hipLaunchKernelGGL(potrf_step_kernel, dim3(1 + a.n_ub + a.n_ps + a.n_ua), dim3(THREADS), 0, s, a);
  GPX_HIP(hipGetLastError());
  return 0;
}

#include <cmath>
#include <cstdlib>
#include <vector>

void gpx_set_error(const char* fmt, ...) { (void)fmt; }

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int step = argc > 2 ? atoi(argv[2]) : 1;
  const long ld = n;
  std::vector<double> A((size_t)n * n, 0.0);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      s = s * 1664525u + 1013904223u;
      A[(size_t)i * n + j] = i == j ? 4.0 + n / 64.0 : ((double)(s >> 8) / (1 << 24) - 0.5) * 0.1;
    }
  double *dA, *dinv; int* info; unsigned* flags;
  hipMalloc(&dA, (size_t)n * n * 8); hipMalloc(&dinv, (size_t)(n / 128) * 128 * 128 * 8); hipMalloc(&info, 4);
  hipMalloc(&flags, (GPX_STEP_FLAGS + 16) * 4);
  hipMemset(info, 0, 4); hipMemset(flags, 0, (GPX_STEP_FLAGS + 16) * 4);
  unsigned epoch = 0;
  long long prof[64];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), (size_t)n * n * 8, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    for (int j0 = 0; j0 + 128 < n; j0 += 128) {
      gpx_potrf_step_launch(nullptr, n, j0, dA, ld, dinv + (size_t)(j0 / 128) * 128 * 128, info, 0, flags, ++epoch,
                            (int*)(flags + GPX_STEP_FLAGS));
      if (j0 / 128 == step) { hipDeviceSynchronize(); hipMemcpyFromSymbol(prof, HIP_SYMBOL(step_prof), sizeof(prof)); }
    }
    gpx_leaf_potf2_inv(nullptr, 128, dA + (size_t)(n - 128) * ld + n - 128, ld, dinv + (size_t)(n / 128 - 1) * 128 * 128, info, n - 128);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("rep %d: potrf(%d) by fused steps %.3f ms (incl. one sync for the profile copy)\n", rep, n, ms);
  }
  int hinfo, herr;
  hipMemcpy(&hinfo, info, 4, hipMemcpyDeviceToHost); hipMemcpy(&herr, flags + GPX_STEP_FLAGS, 4, hipMemcpyDeviceToHost);
  printf("info %d timeout %d\n", hinfo, herr);
  auto us = [&](int k) { return prof[k] == 0 ? -1.0 : (double)(prof[k] - prof[0]) / 100.0; };
  printf("step %d of n = %d, microseconds after the leaf's start\n", step, n);
  printf("leaf      : loaded %.1f  factored %.1f  stored+inverted %.1f  flag raised %.1f\n", us(1), us(2), us(3), us(4));
  printf("PS first  : start %.1f  X staged %.1f  flag seen %.1f  W loaded %.1f  stores issued %.1f  flag raised %.1f\n", us(8), us(9), us(10), us(11), us(12), us(13));
  printf("PS last   : start %.1f  X staged %.1f  flag seen %.1f  W loaded %.1f  stores issued %.1f  flag raised %.1f\n", us(32), us(33), us(34), us(35), us(36), us(37));
  printf("Ua first  : start %.1f  prev panel done %.1f  flags seen %.1f  rows staged %.1f  stored %.1f\n", us(16), us(17), us(18), us(19), us(20));
  printf("Ua last   : start %.1f  prev panel done %.1f  flags seen %.1f  rows staged %.1f  stored %.1f\n", us(40), us(41), us(42), us(43), us(44));
  printf("Ub first  : start %.1f  done %.1f (stamps 25-27 unused: %.1f)   Ub last: start %.1f done %.1f\n", us(24), us(28), us(25), us(48), us(52));
  return 0;
}
