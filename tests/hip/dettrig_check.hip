// TEST-ONLY: include/mas_dettrig.h evaluated ON THE DEVICE, (1) against the host evaluation of the very same header -- the bits the CPU
// oracle's "det" mode computes: every GPU parity test rests on the two being the same arithmetic -- and (2) against glibc's sin / cos /
// tan in units in the last place, densely over the argument ranges the workloads reach (headings of a 100-200 step lap, steering
// angles inside the input bounds, pendulum angles) and a wide range on top.  tests/test_dettrig.py makes the second statement for
// the CPU build only; this program makes both for the device build.
// Built by __graft_entry__.build() into tests/hip/build/dettrig_check; run by tests/test_gpu_parity.py (needs a GPU).
// usage: dettrig_check [millions of points per range]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mas_dettrig.h"

__global__ void eval(const double* x, double* sn, double* cs, double* tn, double* t1, double* s2, double* c2, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  sn[i] = mas_dt_sin(x[i]); cs[i] = mas_dt_cos(x[i]); tn[i] = mas_dt_tan(x[i]); t1[i] = mas_dt_tan_onediv(x[i]);
  mas_dt_sincos(x[i], &s2[i], &c2[i]);
}

static double ulps(double got, double want) {
  if (got == want) return 0.0;
  if (std::isnan(got) || std::isnan(want)) return (std::isnan(got) && std::isnan(want)) ? 0.0 : 1e300;
  const double sp = std::nextafter(std::fabs(want), INFINITY) - std::fabs(want);
  return std::fabs(got - want) / sp;
}
static bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (std::isnan(a) && std::isnan(b)); }

int main(int argc, char** argv) {
  const size_t per = static_cast<size_t>(argc > 1 ? std::atoi(argv[1]) : 2) * 1000000;
  struct Range { const char* what; double lo, hi; };
  const Range ranges[] = {{"steering angles", -0.8, 0.8}, {"headings of a lap", -70.0, 70.0}, {"pendulum angles", -25.0, 25.0}, {"wide", -1e5, 1e5}, {"tiny", -1e-8, 1e-8}};
  double *dx, *dout[6];
  if (hipMalloc(&dx, per * 8) != hipSuccess) { std::printf("dettrig_check: no device\n"); return 2; }
  for (auto& p : dout) if (hipMalloc(&p, per * 8) != hipSuccess) return 2;
  std::vector<double> x(per), out[6];
  for (auto& o : out) o.resize(per);
  unsigned long long s = 88172645463325252ull, mismatches = 0, points = 0;
  double worst_sin = 0, worst_cos = 0, worst_tan = 0;
  for (const Range& r : ranges) {
    for (size_t i = 0; i < per; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = r.lo + (r.hi - r.lo) * (static_cast<double>(s >> 11) / 9007199254740992.0); }
    x[0] = 0.0; x[1] = -0.0; x[2] = r.lo; x[3] = r.hi;
    (void)hipMemcpy(dx, x.data(), per * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(eval, dim3(static_cast<unsigned>((per + 255) / 256)), dim3(256), 0, 0, dx, dout[0], dout[1], dout[2], dout[3], dout[4], dout[5], per);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("dettrig_check: device error\n"); return 2; }
    for (int k = 0; k < 6; ++k) (void)hipMemcpy(out[k].data(), dout[k], per * 8, hipMemcpyDeviceToHost);
    double ws = 0, wc = 0, wt = 0; unsigned long long bad = 0;
    for (size_t i = 0; i < per; ++i) {
      double hs2, hc2; mas_dt_sincos(x[i], &hs2, &hc2);
      const bool ok = same_bits(out[0][i], mas_dt_sin(x[i])) && same_bits(out[1][i], mas_dt_cos(x[i])) && same_bits(out[2][i], mas_dt_tan(x[i])) &&
                      same_bits(out[3][i], mas_dt_tan_onediv(x[i])) && same_bits(out[4][i], hs2) && same_bits(out[5][i], hc2) &&
                      same_bits(out[2][i], out[3][i]) && same_bits(out[0][i], out[4][i]) && same_bits(out[1][i], out[5][i]);
      if (!ok) { if (bad == 0) std::printf("  first mismatch in '%s': x = %a\n", r.what, x[i]); ++bad; }
      ws = std::fmax(ws, ulps(out[0][i], std::sin(x[i]))); wc = std::fmax(wc, ulps(out[1][i], std::cos(x[i]))); wt = std::fmax(wt, ulps(out[2][i], std::tan(x[i])));
    }
    std::printf("  %-18s [%g, %g]: %zu points, device != host header: %llu, max ulp vs glibc: sin %.3f cos %.3f tan %.3f\n", r.what, r.lo, r.hi, per, bad, ws, wc, wt);
    mismatches += bad; points += per; worst_sin = std::fmax(worst_sin, ws); worst_cos = std::fmax(worst_cos, wc); worst_tan = std::fmax(worst_tan, wt);
  }
  const bool accurate = worst_sin <= 1.0 && worst_cos <= 1.0 && worst_tan <= 3.0;
  std::printf("dettrig_check: %llu points, %llu mismatches between device and host evaluation, max ulp vs glibc sin %.3f cos %.3f tan %.3f (%s)\n",
              points, mismatches, worst_sin, worst_cos, worst_tan, accurate ? "within 1 / 1 / 3" : "OUTSIDE 1 / 1 / 3");
  return (mismatches == 0 && accurate) ? 0 : 1;
}
