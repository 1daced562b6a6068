// Cycle breakdown of the 128x128 leaf (development tool): hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -ffp-contract=off -Iinclude -Igaussian_process_amd/csrc tools/leaf_profile.hip -o tools/leaf_profile
#define GPX_LEAF_PROFILE 1
#include "../gaussian_process_amd/csrc/leaf.hip"

#include <cmath>
#include <vector>

void gpx_set_error(const char* fmt, ...) { (void)fmt; }

int main() {
  const int n = 128;
  std::vector<double> G(n * n), A(n * n);
  unsigned s = 12345;
  for (auto& g : G) { s = s * 1664525u + 1013904223u; g = (double)(s >> 8) / (1 << 24) - 0.5; }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double a = i == j ? 1.0 : 0.0;
      for (int k = 0; k < n; ++k) a += G[i * n + k] * G[j * n + k] / n;
      A[i * n + j] = a;
    }
  double *dA, *dinv;
  int* info;
  hipMalloc(&dA, n * n * 8); hipMalloc(&dinv, n * n * 8); hipMalloc(&info, 4);
  hipMemset(info, 0, 4);
  long long prof[64];
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
    gpx_leaf_potf2_inv(nullptr, n, dA, n, dinv, info, 0);
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(prof, HIP_SYMBOL(leaf_prof), sizeof(prof));
  }
  // accuracy against a host Cholesky and of the inverse
  std::vector<double> Lr(A), Ld(n * n), Di(n * n);
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < j; ++k)
      for (int i = j; i < n; ++i) Lr[i * n + j] -= Lr[i * n + k] * Lr[j * n + k];
    const double dj = sqrt(Lr[j * n + j]);
    for (int i = j; i < n; ++i) Lr[i * n + j] /= dj;
  }
  hipMemcpy(Ld.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(Di.data(), dinv, n * n * 8, hipMemcpyDeviceToHost);
  double eL = 0, eI = 0, eU = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      if (j <= i) eL = fmax(eL, fabs(Ld[i * n + j] - Lr[i * n + j]));
      else eU = fmax(eU, fabs(Ld[i * n + j]) + fabs(Di[i * n + j]));
      double a = 0;
      for (int k = j; k <= i; ++k) a += Di[i * n + k] * Ld[k * n + j];
      eI = fmax(eI, fabs(a - (i == j ? 1.0 : 0.0)));
    }
  printf("max |L - Lhost| %.2e   max |dinv L - I| %.2e   upper %.1e\n", eL, eI, eU);
  auto d = [&](int a, int b) { return (double)(prof[b] - prof[a]); };
  printf("cycles (clock64): load %.0f  A(0) %.0f\n", d(0, 1), d(1, 2));
  double sb = 0, sc = 0;
  for (int p = 0; p < 7; ++p) {
    const double b = d(2 + 2 * p, 3 + 2 * p), c = d(3 + 2 * p, 4 + 2 * p);
    printf("  panel %d: B %.0f  C || A(p+1) %.0f\n", p, b, c);
    sb += b; sc += c;
  }
  printf("factor: B %.0f  C||A %.0f ; store A %.0f ; inverse: l0 %.0f l1 %.0f l2 %.0f l3 %.0f ; store dinv %.0f ; total %.0f\n",
         sb, sc, d(16, 29), d(29, 30), d(30, 31), d(31, 32), d(32, 33), d(33, 34), d(0, 34));
  return 0;
}
