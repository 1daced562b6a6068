// Latency of a device-scope hand-over between workgroups on different XCDs (development microbenchmark).
// Workgroup i waits for flag[i-1], reads the 16 KiB payload workgroup i-1 wrote, adds 1, writes its own payload and
// sets flag[i].  Total time / hops = cost of one dependent hand-over.  Variant 0: plain stores + __threadfence();
// variant 1: agent-scope atomic (write-through / L2-bypassing) accesses for payload and flag, no cache-wide fence.
// Every spin is bounded, so the kernel always ends.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int PAY = 2048;   // doubles per payload (16 KiB)

template <int VARIANT>
__global__ __launch_bounds__(256) void hop_kernel(double* pay, unsigned* flag, unsigned epoch, int* err) {
  const int b = blockIdx.x, t = threadIdx.x;
  double v[PAY / 256];
  if (b > 0) {
    if (t == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(flag + b - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1u << 22)) { *err = b; break; }
      }
    }
    __syncthreads();
    if (VARIANT == 0) __threadfence();
    const double* src = pay + (size_t)(b - 1) * PAY;
    for (int i = 0; i < PAY / 256; ++i)
      v[i] = VARIANT == 0 ? src[t + 256 * i] : __hip_atomic_load(src + t + 256 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    for (int i = 0; i < PAY / 256; ++i) v[i] = 0.0;
  }
  double* dst = pay + (size_t)b * PAY;
  for (int i = 0; i < PAY / 256; ++i) {
    if (VARIANT == 0) dst[t + 256 * i] = v[i] + 1.0;
    else __hip_atomic_store(dst + t + 256 * i, v[i] + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (VARIANT == 0) __threadfence();
  else __builtin_amdgcn_s_waitcnt(0);   // all stores of this thread acknowledged
  __syncthreads();
  if (t == 0) __hip_atomic_store(flag + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  const int hops = 256;
  double* pay; unsigned* flag; int* err;
  CHECK(hipMalloc(&pay, sizeof(double) * PAY * hops));
  CHECK(hipMalloc(&flag, sizeof(unsigned) * hops));
  CHECK(hipMalloc(&err, sizeof(int)));
  CHECK(hipMemset(flag, 0, sizeof(unsigned) * hops));
  CHECK(hipMemset(err, 0, sizeof(int)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  unsigned epoch = 0;
  for (int variant = 0; variant < 2; ++variant) {
    for (int rep = 0; rep < 4; ++rep) {
      ++epoch;
      CHECK(hipEventRecord(e0));
      if (variant == 0) hop_kernel<0><<<hops, 256>>>(pay, flag, epoch, err);
      else hop_kernel<1><<<hops, 256>>>(pay, flag, epoch, err);
      CHECK(hipEventRecord(e1));
      CHECK(hipDeviceSynchronize());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<double> h(PAY); int herr;
      CHECK(hipMemcpy(h.data(), pay + (size_t)(hops - 1) * PAY, sizeof(double) * PAY, hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
      int bad = 0; for (double x : h) bad += x != (double)hops;
      printf("variant %d (%s) rep %d: %d hops %.1f us -> %.2f us per hop; wrong payload entries %d, timeout at %d\n", variant,
             variant == 0 ? "plain + __threadfence" : "agent-scope atomics", rep, hops, ms * 1e3, ms * 1e3 / hops, bad, herr);
    }
  }
  return 0;
}
