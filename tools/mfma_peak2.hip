// Microbenchmark 2: what keeps a 4x4-tile f64 MFMA loop from 64 cycles/MFMA?
// MODE 0: operands fixed in registers; 1: + LDS fragment reads per k-step; 2: + barrier per 64 MFMAs
// MODE 3: mode 2 with 1 block/CU.  Build: hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak2.hip -o tools/mfma_peak2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(double* out, int iters) {
  __shared__ double lds[4096];
  const int t = threadIdx.x;
  for (int i = t; i < 4096; i += 256) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  v4d acc[4][4];
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) acc[x][y] = (v4d){0, 0, 0, 0};
  double a[4], b[4];
  int junk = t;
  for (int x = 0; x < 4; ++x) { a[x] = 1.0 + t * 1e-3 + x; b[x] = 2.0 - t * 1e-4 + x; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (MODE >= 1) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          a[x] = lds[((kk * 4 + x) * 64 + (t & 63)) & 4095];
          b[x] = lds[(2048 + (kk * 4 + x) * 64 + (t & 63)) & 4095];
        }
      } else {
#pragma unroll
        for (int x = 0; x < 4; ++x) { asm volatile("" : "+v"(a[x])); asm volatile("" : "+v"(b[x])); }
      }
      if (MODE >= 4) {  // NV integer VALU ops per 16 MFMAs
#pragma unroll
        for (int v = 0; v < (MODE - 3) * 16; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(junk) : "v"(t));
      }
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
    }
    if (MODE >= 2) __syncthreads();
  }
  double s = 0;
  for (int x = 0; x < 4; ++x) for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  out[blockIdx.x * 256 + t] = s + junk;
}

template <int MODE>
void run(int blocks_per_cu, int iters) {
  int blocks = 256 * blocks_per_cu;
  double* out; (void)hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = 2048.0 * 64 * iters * blocks * 4;
    if (rep) printf("MODE=%d blocks/CU=%d: %.2f ms  %.2f TFLOP/s\n", MODE, blocks_per_cu, ms, flops / ms / 1e9);
  }
  (void)hipFree(out);
}
int main() {
  run<2>(2, 4000); run<4>(2, 4000); run<5>(2, 4000); run<7>(2, 4000); run<4>(1, 4000); run<7>(1, 4000);
  return 0;
}
