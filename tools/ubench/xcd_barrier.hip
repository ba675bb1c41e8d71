// Micro-benchmark behind the centralised strategy's rung agreement: how long does an all-to-all
// "arrive and wait" among 16 workgroups of an ordinary launch take, by where the workgroups sit and by the scopes of the
// arriving atomic and of the polling load?
//   placement  spread : workgroups 0..15 (dealt round-robin over the 8 XCDs)
//              one XCD: the 16 workgroups with blockIdx % 8 == 0 of a 128-workgroup launch (the others exit at once);
//                       every participant records its HW XCC_ID so the placement assumption is CHECKED, not assumed
//   scopes     atomic agent / load agent        what mas_kernels_hip.hpp's ladder_agree uses today
//              atomic workgroup / load agent    atomic resolved in the XCD's own L2 (?), load bypasses the CU's L1
//              atomic workgroup / load workgroup
// A variant is only usable if it COMPLETES (every wait ends within the spin bound) -- reported per variant.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_barrier.hip -o tools/ubench/build/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kIters = 2000, kParticipants = 16;

template <int ASCOPE, int LSCOPE>
__global__ void __launch_bounds__(256) barrier_bench(unsigned* words, int stride_blocks, long long* ticks, int* xcc, int* failed) {
  if (blockIdx.x % stride_blocks != 0) return;
  const int me = blockIdx.x / stride_blocks;
  if (me >= kParticipants) return;
  unsigned id = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[me] = static_cast<int>(id & 0xF);
  long long t0 = 0;
  for (int it = 0; it < kIters; ++it) {
    if (it == 100) t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(words + it, 1u, __ATOMIC_RELAXED, ASCOPE);
      int spins = 0;
      while (__hip_atomic_load(words + it, __ATOMIC_RELAXED, LSCOPE) != kParticipants) {
        if (++spins > (1 << 16)) { atomicExch(failed, 1); break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (__hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
  }
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) ticks[me] = t1 - t0;
}

template <int ASCOPE, int LSCOPE>
void run(const char* name, bool one_xcd) {
  unsigned* words; long long* ticks; int *xcc, *failed;
  hipMalloc(&words, sizeof(unsigned) * kIters); hipMalloc(&ticks, sizeof(long long) * kParticipants);
  hipMalloc(&xcc, sizeof(int) * kParticipants); hipMalloc(&failed, sizeof(int));
  hipMemset(words, 0, sizeof(unsigned) * kIters); hipMemset(ticks, 0, sizeof(long long) * kParticipants); hipMemset(failed, 0, sizeof(int));
  const int stride = one_xcd ? 8 : 1;
  hipLaunchKernelGGL((barrier_bench<ASCOPE, LSCOPE>), dim3(kParticipants * stride), dim3(256), 0, 0, words, stride, ticks, xcc, failed);
  hipError_t e = hipDeviceSynchronize();
  std::vector<long long> t(kParticipants); std::vector<int> x(kParticipants); int f = 0;
  hipMemcpy(t.data(), ticks, sizeof(long long) * kParticipants, hipMemcpyDeviceToHost);
  hipMemcpy(x.data(), xcc, sizeof(int) * kParticipants, hipMemcpyDeviceToHost);
  hipMemcpy(&f, failed, sizeof(int), hipMemcpyDeviceToHost);
  long long mx = 0; for (auto v : t) mx = v > mx ? v : mx;
  std::printf("%-44s %s  %7.1f ns per arrive-and-wait   xcc ids:", name, (e == hipSuccess && f == 0) ? "completes" : "FAILS    ", 10.0 * mx / (kIters - 100));
  for (auto v : x) std::printf(" %d", v);
  std::printf("\n");
  hipFree(words); hipFree(ticks); hipFree(xcc); hipFree(failed);
}

int main() {
  constexpr int AG = __HIP_MEMORY_SCOPE_AGENT, WG = __HIP_MEMORY_SCOPE_WORKGROUP;
  run<AG, AG>("spread,  atomic agent     / load agent", false);
  run<AG, AG>("one XCD, atomic agent     / load agent", true);
  run<WG, AG>("one XCD, atomic workgroup / load agent", true);
  run<WG, WG>("one XCD, atomic workgroup / load workgroup", true);
  run<WG, AG>("spread,  atomic workgroup / load agent", false);
  return 0;
}
