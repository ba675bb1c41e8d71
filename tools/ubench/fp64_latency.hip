// Micro-benchmarks behind DESIGN.md's latency arguments: cycles per dependent / independent FP64 op for ONE wave
// on a SIMD (the situation of every wave of the forward and backward kernels at 4096 agents).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/fp64_latency.hip -o /tmp/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define REP 256

template <int MODE>
__global__ void bench(double* out, long long* cyc, double seed, int waves_note) {
  double a = seed + threadIdx.x * 1e-3, b = seed * 0.5 + 1.0, c = 1.0000001, d = 0.9999999;
  double e = a + 1, f = a + 2, g = a + 3;
  __shared__ double lds[256];
  lds[threadIdx.x] = a;
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 16; ++r) {
      if (MODE == 0) { a = a + b; }                                     // dependent add
      if (MODE == 1) { a = a * c; }                                     // dependent mul
      if (MODE == 2) { a = a + b; e = e + b; f = f + b; g = g + b; }    // 4 independent add streams
      if (MODE == 3) { a = a / c; }                                     // dependent IEEE division
      if (MODE == 4) { a = sqrt(a + 2.0); }                             // dependent IEEE sqrt (+1 add)
      if (MODE == 5) { lds[threadIdx.x] = a; __syncthreads(); a = lds[(threadIdx.x + 1) & 63] + b; __syncthreads(); }
      if (MODE == 6) { int lo = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) * 4, __double2loint(a));
                       int hi = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) * 4, __double2hiint(a));
                       a = __hiloint2double(hi, lo) + b; }
      if (MODE == 7) { a = a * c + b; }                                 // dependent mul+add (two roundings)
      if (MODE == 8) { a = a + b; e = e + b; }                          // 2 independent streams
      if (MODE == 9) { a = __builtin_amdgcn_rcp(a) + b; }               // v_rcp_f64 + add
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + e + f + g + d;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int ops_per_rep, int grid) {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 64 * grid);
  hipMalloc(&cyc, sizeof(long long) * grid);
  bench<MODE><<<grid, 64>>>(out, cyc, 1.25, grid);
  bench<MODE><<<grid, 64>>>(out, cyc, 1.25, grid);
  hipDeviceSynchronize();
  long long* h = new long long[grid];
  hipMemcpy(h, cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
  // clock64 ticks at the 100 MHz constant clock? report raw and let the reader scale; also time by events below
  printf("%-34s grid %5d  ticks/op %.2f\n", name, grid, s / grid / REP / ops_per_rep);
  delete[] h; hipFree(out); hipFree(cyc);
}

__global__ void wallclock_probe(long long* out) {
  long long c0 = clock64(), w0 = wall_clock64();
  double a = threadIdx.x;
  for (int i = 0; i < 20000; ++i) a = a * 1.0000001 + 1e-9;
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)a; }
}

int main() {
  long long* d; hipMalloc(&d, 32); wallclock_probe<<<1, 64>>>(d); hipDeviceSynchronize();
  long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("clock64 ticks %lld vs wall_clock64 (100 MHz) ticks %lld -> clock64 runs at %.1f MHz\n", h[0], h[1], 100.0 * h[0] / h[1]);
  for (int grid : {1, 1024, 2048, 4096}) {
    run<0>("dependent add", 1, grid);
    run<1>("dependent mul", 1, grid);
    run<7>("dependent mul,add", 2, grid);
    run<8>("2 independent add streams", 2, grid);
    run<2>("4 independent add streams", 4, grid);
    run<3>("dependent div", 1, grid);
    run<4>("dependent sqrt(+add)", 1, grid);
    run<9>("dependent v_rcp_f64(+add)", 1, grid);
    run<5>("LDS write,barrier,read,barrier", 1, grid);
    run<6>("ds_bpermute f64 (+add)", 1, grid);
  }
  return 0;
}
