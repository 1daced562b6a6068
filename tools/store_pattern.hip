// Store-pattern microbenchmark for the Gram kernel (development tool): how fast can 256-thread
// workgroups write an n x n fp64 matrix when each owns an R-row x C-column tile and every wave
// store instruction writes 1 KiB of one row?  Compared with a linear fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));

template <int R, int C, bool XFAST>
__global__ __launch_bounds__(256) void tile_store(double* K, long ld, int tiles_x, int tiles_y, double v) {
  int bx, by;
  if (XFAST) { bx = blockIdx.x % tiles_x; by = blockIdx.x / tiles_x; }
  else { by = blockIdx.x % tiles_y; bx = blockIdx.x / tiles_y; }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  constexpr int SEG = C / 128;              // 1 KiB segments per row of the tile
  constexpr int RW = R / 4;                 // rows per wave
  double* base = K + (long)(by * R + wid * RW) * ld + (long)bx * C + 2 * lane;
#pragma unroll 4
  for (int r = 0; r < RW; ++r) {
#pragma unroll
    for (int s = 0; s < SEG; ++s) *reinterpret_cast<v2d*>(base + (long)r * ld + s * 128) = (v2d){v, v + r};
  }
}

__global__ void linear_fill(double* K, size_t n, double v) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 2;
  for (; i < n; i += stride) *reinterpret_cast<v2d*>(K + i) = (v2d){v, v};
}

template <int R, int C, bool XFAST>
void run(const char* name, double* K, int n) {
  const int tx = n / C, ty = n / R;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((tile_store<R, C, XFAST>), dim3(tx * ty), dim3(256), 0, 0, K, (long)n, tx, ty, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-28s %7.3f ms  %6.0f GB/s\n", name, best, 8.0 * n * n / best / 1e6);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 65536;
  double* K;
  if (hipMalloc(&K, (size_t)n * n * 8) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(linear_fill, dim3(256 * 16), dim3(256), 0, 0, K, (size_t)n * n, 1.0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-28s %7.3f ms  %6.0f GB/s\n", "linear fill", ms, 8.0 * n * n / ms / 1e6);
  }
  run<64, 128, true>("64x128 x-fast", K, n);
  run<64, 128, false>("64x128 y-fast", K, n);
  run<16, 512, true>("16x512 x-fast", K, n);
  run<16, 512, false>("16x512 y-fast", K, n);
  run<32, 256, true>("32x256 x-fast", K, n);
  run<8, 1024, true>("8x1024 x-fast", K, n);
  run<4, 2048, true>("4x2048 x-fast", K, n);
  run<64, 256, true>("64x256 x-fast", K, n);
  run<128, 128, true>("128x128 x-fast", K, n);
  (void)hipFree(K);
  return 0;
}
