// fp64 VALU issue-rate microbenchmark (development tool): cycles per wave64 v_fma_f64 / v_add_f64
// per SIMD with 1..8 waves per SIMD and 8 independent accumulator chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double x) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fma(a[i], x, 1e-9);
      else if (OP == 1) a[i] = a[i] + x;
      else a[i] = a[i] * x;
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* out; (void)hipMalloc(&out, 256 * 8 * 256 * 8 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int op = 0; op < 3; ++op)
    for (int wps = 1; wps <= 8; wps *= 2) {       // blocks per CU = wps (256 threads = 1 wave per SIMD)
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (op == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999999);
        if (op == 1) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 1e-9);
        if (op == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wps), dim3(256), 0, 0, out, iters, 0.999999);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double instr_per_simd = (double)iters * 8 * wps;
      printf("%s waves/SIMD %d: %.3f ms  %.2f cycles per wave instr at 2.4 GHz  (%.1f T lane-ops/s)\n",
             op == 0 ? "fma" : op == 1 ? "add" : "mul", wps, best, best * 1e-3 * 2.4e9 / instr_per_simd,
             instr_per_simd * 1024 * 64 / (best * 1e-3) / 1e12);
    }
  return 0;
}
