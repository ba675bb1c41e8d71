// TEST-ONLY: div_by (multi_agent_solver_amd/csrc/mas_platform.hpp) against the device's own IEEE division, bit for bit, on random
// numerators over the whole exponent range in which the path runs (and far beyond it), the special values, and the divisors the
// kernels use (2 eps, eps^2, 4 eps^2 for eps = 1e-6 / 1e-5, the wheelbase 2.5) plus random ones.
// Built by __graft_entry__.build() into tests/hip/build/div_by_check; run by tests/test_gpu_parity.py (needs a GPU).
// usage: div_by_check [millions of cases per divisor]      prints "div_by_check: <cases> cases, <mismatches> mismatches"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../multi_agent_solver_amd/csrc/mas_platform.hpp"

__device__ inline unsigned long long mix(unsigned long long z) {   // splitmix64
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

__global__ void check(double b, unsigned long long seed, int per_thread, int exp_lo, int exp_hi, unsigned long long* mismatches, double* first_bad) {
  const InvDiv d = make_inv_div(b);
  unsigned long long s = seed + (static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x) * 0x632BE59BD9B4E019ull;
  unsigned long long bad = 0;
  for (int i = 0; i < per_thread; ++i) {
    s = mix(s);
    // random sign and mantissa, exponent uniform in [exp_lo, exp_hi]
    const unsigned long long mant = s & 0xFFFFFFFFFFFFFull, sign = (s >> 63) << 63;
    const unsigned long long e = static_cast<unsigned long long>(exp_lo + static_cast<int>((s >> 52) % static_cast<unsigned long long>(exp_hi - exp_lo + 1)) + 1023);
    const unsigned long long bits = sign | (e << 52) | mant;
    double p; memcpy(&p, &bits, sizeof p);
    const double want = p / b, got = div_by(p, d);
    unsigned long long wb, gb; memcpy(&wb, &want, 8); memcpy(&gb, &got, 8);
    if (wb != gb) { if (bad == 0 && atomicAdd(mismatches, 0ull) == 0ull) { first_bad[0] = p; first_bad[1] = b; first_bad[2] = want; first_bad[3] = got; } ++bad; }
  }
  if (bad) atomicAdd(mismatches, bad);
}

// the edge of div_by's domain: numerators 2^-1022 .. 2^-940.  The residual p - b q is a multiple of 2^(e_p - 104), exactly
// representable only for |p| >= 2^-969; below that (or with a subnormal quotient) the last bit may differ from `/`.  Inside the domain: 0.
__global__ void check_boundary(double b, unsigned long long seed, int per_thread, unsigned long long* mismatches, unsigned long long* outside) {
  const InvDiv d = make_inv_div(b);
  unsigned long long s = seed + 977ull * (blockIdx.x * blockDim.x + threadIdx.x);
  unsigned long long bad = 0, out = 0;
  for (int i = 0; i < per_thread; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const int e = -1022 + static_cast<int>((s >> 40) % 83);                       // -1022 .. -940
    const double m = 1.0 + static_cast<double>((s >> 11) & ((1ull << 40) - 1)) / 1099511627776.0;
    double p = ldexp(m, e);
    if (s & 1ull) p = -p;
    const double want = p / b, got = div_by(p, d);
    unsigned long long wb, gb; memcpy(&wb, &want, 8); memcpy(&gb, &got, 8);
    if (wb != gb) { if (fabs(p) >= 0x1p-969 && fabs(want) >= 0x1p-1022) ++bad; else ++out; }
  }
  if (bad) atomicAdd(mismatches, bad);
  if (out) atomicAdd(outside, out);
}

__global__ void check_special(double b, unsigned long long* mismatches) {
  const InvDiv d = make_inv_div(b);
  const double inf = __longlong_as_double(0x7FF0000000000000ll), nan = __longlong_as_double(0x7FF8000000000000ll);
  const double vals[] = {0.0, -0.0, inf, -inf, nan, 1.0, -1.0, 1e308, -1e308, 1e-300, 4.9406564584124654e-324};
  unsigned long long bad = 0;
  for (double p : vals) {
    const double want = p / b, got = div_by(p, d);
    unsigned long long wb, gb; memcpy(&wb, &want, 8); memcpy(&gb, &got, 8);
    const bool both_nan = (want != want) && (got != got);
    const bool subnormal_quotient = want != 0.0 && (fabs(want) < 0x1p-1022 || fabs(p) < 0x1p-969);      // outside div_by's stated domain (mas_platform.hpp): quotient subnormal or the residual no longer exact
    if (wb != gb && !both_nan && !subnormal_quotient) { ++bad; printf("  special: p=%a b=%a want=%a got=%a\n", p, b, want, got); }
  }
  if (bad) atomicAdd(mismatches, bad);
}

int main(int argc, char** argv) {
  const int millions = argc > 1 ? std::atoi(argv[1]) : 20;
  unsigned long long* dm; unsigned long long* dout; double* fb;
  if (hipMalloc(&dm, 8) != hipSuccess || hipMalloc(&dout, 8) != hipSuccess || hipMalloc(&fb, 32) != hipSuccess) { std::printf("div_by_check: no device\n"); return 2; }
  (void)hipMemset(dm, 0, 8); (void)hipMemset(dout, 0, 8); (void)hipMemset(fb, 0, 32);
  std::vector<double> divisors = {2e-6, 1e-6 * 1e-6, 4 * 1e-6 * 1e-6, 2e-5, 1e-5 * 1e-5, 4 * 1e-5 * 1e-5, 2.5, 3.0, 0.1, 9.81, 1.0, 7.0, 1.9999999999999998, 1.0000000000000002};
  unsigned long long rs = 12345;
  for (int i = 0; i < 18; ++i) { rs = rs * 6364136223846793005ull + 1442695040888963407ull; divisors.push_back(0.5 + 3.0 * (static_cast<double>(rs >> 11) / 9007199254740992.0)); }
  const int threads = 256, blocks = 1024, per_thread = static_cast<int>(static_cast<long long>(millions) * 1000000 / (threads * blocks)) + 1;
  unsigned long long cases = 0;
  for (size_t k = 0; k < divisors.size(); ++k) {
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, divisors[k], 1000ull + k, per_thread, -900, 900, dm, fb);   // |p| in [2^-900, 2^900]
    hipLaunchKernelGGL(check, dim3(blocks), dim3(threads), 0, 0, divisors[k], 5000ull + k, per_thread, -60, 20, dm, fb);     // the range the path lives in
    hipLaunchKernelGGL(check_special, dim3(1), dim3(1), 0, 0, divisors[k], dm);
    hipLaunchKernelGGL(check_boundary, dim3(64), dim3(threads), 0, 0, divisors[k], 9000ull + k, 256, dm, dout);                 // numerators at the edge of the domain
    cases += 2ull * threads * blocks * per_thread + 64ull * threads * 256;
  }
  if (hipDeviceSynchronize() != hipSuccess) { std::printf("div_by_check: device error\n"); return 2; }
  unsigned long long m = 0; double bad[4];
  (void)hipMemcpy(&m, dm, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(bad, fb, 32, hipMemcpyDeviceToHost);
  unsigned long long mo = 0; (void)hipMemcpy(&mo, dout, 8, hipMemcpyDeviceToHost);
  std::printf("div_by_check: %llu cases, %llu mismatches\n", cases, m);
  std::printf("  (outside the stated domain -- |p| < 2^-969 or a subnormal quotient: %llu last-bit differences, as documented)\n", mo);
  if (m) std::printf("  first: p=%a b=%a want=%a got=%a\n", bad[0], bad[1], bad[2], bad[3]);
  return m ? 1 : 0;
}
