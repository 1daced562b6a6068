// Which XCD does workgroup `bid` of a 1-D grid land on?  (development tool: checks the bid % 8 assumption of
// the GEMM's super-tile mapping)  hipcc --offload-arch=gfx950 -O2 tools/xcc_map.hip -o tools/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void probe(int* xcc, int spin) {
  // HW_REG_XCC_ID = 20, bits [3:0]: s_getreg_b32 imm = ((size - 1) << 11) | (offset << 6) | id
  const int id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
  double v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0000001 + 1e-9;   // keep the workgroup resident for a while
  if (threadIdx.x == 0) xcc[blockIdx.x] = id + (v < 0 ? 1 : 0);
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 4096;
  int* d;
  if (hipMalloc(&d, nwg * sizeof(int)) != hipSuccess) return 1;
  for (int dyn : {0, 65536}) {   // static-only and with 64 KiB of dynamic LDS (as the GEMM: two workgroups per CU)
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(256), dyn, 0, d, 20000);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::vector<int> h(nwg);
    (void)hipMemcpy(h.data(), d, nwg * sizeof(int), hipMemcpyDeviceToHost);
    int match = 0, hist[8][16] = {};
    for (int b = 0; b < nwg; ++b) {
      match += (h[b] & 15) == (b & 7);
      hist[b & 7][h[b] & 15]++;
    }
    printf("dynamic LDS %d: %d of %d workgroups on XCD bid %% 8\n", dyn, match, nwg);
    for (int r = 0; r < 8; ++r) {
      printf("  bid %% 8 = %d ->", r);
      for (int x = 0; x < 8; ++x) printf(" xcc%d:%d", x, hist[r][x]);
      printf("\n");
    }
    printf("  first 32 ids:");
    for (int b = 0; b < 32 && b < nwg; ++b) printf(" %d", h[b] & 15);
    printf("\n");
  }
  return 0;
}
