// Microbenchmark: fp64 MFMA (v_mfma_f64_16x16x4_f64) issue rate and the clock held under load.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#define hipMalloc (void)hipMalloc

#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC, int WPS>
__global__ __launch_bounds__(256, WPS) void mfma_loop(double* out, int iters, unsigned long long* cyc,
                                                 unsigned long long* rt) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int NACC, int WPS>
void run(int waves_per_simd, int iters, int blocks_override = 0) {
  int blocks = blocks_override ? blocks_override : 256 * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
  double* out; unsigned long long *cyc, *rt;
  hipMalloc(&out, sizeof(double) * blocks * 256);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipMalloc(&rt, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<NACC, WPS>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc, rt);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(blocks), r(blocks);
    hipMemcpy(c.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    hipMemcpy(r.data(), rt, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    std::nth_element(c.begin(), c.begin() + blocks / 2, c.end());
    std::nth_element(r.begin(), r.begin() + blocks / 2, r.end());
    double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * blocks * 4;
    double cyc_per_mfma = (double)c[blocks / 2] / ((double)NACC * iters);
    double clk_ghz = (double)c[blocks / 2] / ((double)r[blocks / 2] * 10.0) ;  // memrealtime = 100 MHz
    printf("NACC=%d bounds=%d waves/SIMD=%d: %.2f ms  %.2f TFLOP/s  cycles/MFMA(per wave)=%.1f  clock=%.2f GHz\n", NACC,
           waves_per_simd, ms, flops / ms / 1e9, cyc_per_mfma, clk_ghz);
  }
  hipFree(out); hipFree(cyc); hipFree(rt);
}

int main() {
  run<4, 2>(1, 3000, 1);
  run<4, 2>(1, 300, 1);
  run<16, 1>(1, 40000);
  run<16, 2>(1, 40000);
  run<16, 2>(2, 40000);
  run<8, 2>(2, 40000);
  run<4, 4>(4, 40000);
  run<4, 2>(2, 40000);
  run<2, 8>(8, 40000);
  return 0;
}
