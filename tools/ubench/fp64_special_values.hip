// Does the FP64 pipe of gfx950 take longer on special operands?  One wavefront, dependent chain a = a*c + b with
// different operand classes in ONE lane (the other 63 hold ordinary numbers): normal, denormal, NaN, Inf.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/fp64_special_values.hip -o tools/ubench/build/fp64_special_values
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <limits>

__global__ void bench(double* out, long long* cyc, const double* seed_a, const double* seed_b, const double* seed_c) {
  double a = seed_a[threadIdx.x], b = seed_b[threadIdx.x], c = seed_c[threadIdx.x];
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < 32; ++r) { a = a * c; a = a + b; }
  }
  long long t1 = clock64();
  out[threadIdx.x] = a;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

static void run(const char* name, double a0, double b0, double c0, int lanes_special) {
  double ha[64], hb[64], hc[64];
  for (int i = 0; i < 64; ++i) { ha[i] = 1.25 + i * 1e-3; hb[i] = 1e-9; hc[i] = 0.999999; }
  for (int i = 0; i < lanes_special; ++i) { ha[i] = a0; hb[i] = b0; hc[i] = c0; }
  double *da, *db, *dc, *out; long long* cyc;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dc, 512); hipMalloc(&out, 512); hipMalloc(&cyc, 8);
  hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice); hipMemcpy(dc, hc, 512, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 3; ++rep) bench<<<1, 64>>>(out, cyc, da, db, dc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  double ho[64]; hipMemcpy(ho, out, 512, hipMemcpyDeviceToHost);
  printf("%-44s ticks per op %.2f   (lane0 result %g)\n", name, (double)h / (64 * 32 * 2), ho[0]);
}

int main() {
  const double den = 4.9406564584124654e-324 * 1000, nan = std::numeric_limits<double>::quiet_NaN(), inf = std::numeric_limits<double>::infinity();
  for (int pass = 0; pass < 2; ++pass) {     // second pass: clocks are up
    run("all lanes normal", 1.25, 1e-9, 0.999999, 0);
    run("one lane denormal (a, b denormal, c = 1)", den, den, 1.0, 1);
    run("all lanes denormal", den, den, 1.0, 64);
    run("one lane NaN", nan, 1.0, 1.0, 1);
    run("all lanes NaN", nan, 1.0, 1.0, 64);
    run("one lane Inf", inf, 1.0, 1.0, 1);
    run("one lane huge (1e300, stays finite)", 1e300, 1.0, 1.0, 1);
    run("one lane underflowing to denormal", 1e-300, 0.0, 1e-3, 1);
  }
  return 0;
}
