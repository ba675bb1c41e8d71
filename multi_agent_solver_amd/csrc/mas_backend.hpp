// mas_backend.hpp -- memory, launch and timing primitives under the batch driver.
//
// Product build (hipcc, gfx950): HIP runtime, one stream per handle, 64-lane agent groups.
// MAS_HOST_EMULATION build (g++, tests/hostcheck only): the same kernel bodies are replayed
// lane by lane in plain loops so the driver and the arithmetic can be checked on a CPU-only
// machine.  The emulation is never part of libmas_hip.so.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mas_platform.hpp"
#if !defined(MAS_HOST_EMULATION)
#include <hip/hip_ext.h>   // hipExtLaunchKernelGGL: events bound to the dispatch itself (kernel time without launch gaps)
#endif

namespace mas {

struct KernelStat {
  int64_t launches = 0;
  double total_ms = 0.0;
  double algorithmic_bytes = 0.0;
};

#if defined(MAS_HOST_EMULATION)

using Stream = void*;
#define MAS_ATOMIC_ADD_INT(ptr, v) (*(ptr) += (v))
#define MAS_ATOMIC_ADD_I64(ptr, v) (*(ptr) += (v))

struct Backend {
  Stream stream = nullptr, cur = nullptr;
  bool profiling = false;
  std::map<std::string, KernelStat> stats;

  static int device_count() { return 1; }
  void init(int, void*) {}
  void destroy() {}
  void* alloc(size_t bytes) { void* p = std::calloc(bytes ? bytes : 1, 1); if (!p) throw std::runtime_error("host alloc failed"); return p; }
  void free(void* p) { std::free(p); }
  void* alloc_pinned(size_t bytes) { return alloc(bytes); }
  void free_pinned(void* p) { std::free(p); }
  void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  void d2h(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  void d2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  void zero(void* d, size_t n) { std::memset(d, 0, n); }
  void fill32(int* d, int v, size_t count) { for (size_t i = 0; i < count; ++i) d[i] = v; }
  void sync() {}
  void resolve_profile() {}
  // events (no-ops here: the emulation runs everything in program order)
  struct Event {};
  Event make_event() { return Event{}; }
  void drop_event(Event) {}
  void record(Event&) {}
  void wait(Event&) {}
  void d2h_async(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
  size_t mark() const { return 0; }
  void scale_bytes(size_t, size_t, double) {}

  template <int BY, class F>
  void launch_1d(const char* name, const F& f, int n, double alg_bytes = 0.0) { launch_1d_range<BY>(name, f, 0, n, alg_bytes); }
  template <int BY, class F>
  void launch_2d(const char* name, const F& f, int n, int ny, double alg_bytes = 0.0) { launch_2d_range<BY>(name, f, 0, n, ny, alg_bytes); }
  template <int BY, class F>
  void launch_1d_range(const char* name, const F& f, int lo, int hi, double alg_bytes = 0.0) {
    auto t0 = std::chrono::steady_clock::now();
    for (int a = lo; a < hi; ++a) f(a);
    account(name, t0, alg_bytes);
  }
  template <int BY, class F>
  void launch_2d_range(const char* name, const F& f, int lo, int hi, int ny, double alg_bytes = 0.0) {
    auto t0 = std::chrono::steady_clock::now();
    for (int y = 0; y < ny; ++y)
      for (int a = lo; a < hi; ++a) f(a, y);
    account(name, t0, alg_bytes);
  }
  // one cooperating group of `n` lanes that needs block-wide reductions: emulated serially by
  // the body itself (see CentralBackwardBody)
  template <class F>
  void launch_group(const char* name, const F& f, int n, double alg_bytes = 0.0, size_t = 0, int = 0) {
    auto t0 = std::chrono::steady_clock::now();
    f.run_serial(n);
    account(name, t0, alg_bytes);
  }

 private:
  void account(const char* name, std::chrono::steady_clock::time_point t0, double alg_bytes) {
    if (!profiling) return;
    KernelStat& s = stats[name];
    s.launches++;
    s.total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    s.algorithmic_bytes += alg_bytes;
  }
};

#else  // ------------------------------------------------------------------ HIP product --

using Stream = hipStream_t;
#define MAS_ATOMIC_ADD_INT(ptr, v) atomicAdd((ptr), (v))
#define MAS_ATOMIC_ADD_I64(ptr, v) atomicAdd(reinterpret_cast<unsigned long long*>(ptr), static_cast<unsigned long long>(v))

#define MAS_HIP_CHECK(expr)                                                                                   \
  do {                                                                                                        \
    hipError_t _e = (expr);                                                                                   \
    if (_e != hipSuccess)                                                                                     \
      throw std::runtime_error(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                               std::to_string(__LINE__) + ")");                                              \
  } while (0)

// lane x = agent inside a 64-wide group (one wavefront); y = time step / alpha trial
template <class F>
__global__ void __launch_bounds__(64) mas_kernel_1d(const F f, int lo, int hi) {
  const int a = lo + blockIdx.x * 64 + threadIdx.x;
  if (a < hi) f(a);
}
template <class F, int BY>
__global__ void __launch_bounds__(64 * BY) mas_kernel_2d(const F f, int lo, int hi, int ny) {
  const int a = lo + blockIdx.x * 64 + threadIdx.x;
  int y = blockIdx.y;                                    // wave-uniform (a wavefront is one row of the block): kept scalar
  if constexpr (BY > 1) y = blockIdx.y * BY + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.y));
  if (a < hi && y < ny) f(a, y);
}
// one workgroup whose lanes cooperate through barriers (centralised strategy); MAXT bounds the
// block so the register allocator knows how many VGPRs a lane may keep
template <class F, int MAXT>
__global__ void __launch_bounds__(MAXT) mas_kernel_group(const F f, int n) {
  f.run_block(n);
}

struct Backend {
  Stream stream = nullptr;       // the handle's main stream
  Stream cur = nullptr;          // stream every launch / async copy goes to (== stream)
  bool own_stream = false;
  int device = 0;
  bool profiling = false;
  std::map<std::string, KernelStat> stats;
  struct Pending { std::string name; hipEvent_t e0, e1; double bytes; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;

  static int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
  }
  void init(int dev, void* external_stream) {
    device = dev;
    MAS_HIP_CHECK(hipSetDevice(dev));
    if (external_stream) { stream = static_cast<hipStream_t>(external_stream); own_stream = false; }
    else { MAS_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)); own_stream = true; }
    cur = stream;
  }
  // completion events of the iteration rounds (the host runs one round ahead of the device)
  using Event = hipEvent_t;
  Event make_event() { Event e; MAS_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return e; }
  void drop_event(Event e) { (void)hipEventDestroy(e); }
  void record(Event& e) { MAS_HIP_CHECK(hipEventRecord(e, cur)); }
  void wait(Event& e) { MAS_HIP_CHECK(hipEventSynchronize(e)); }
  void d2h_async(void* d, const void* s, size_t n) { MAS_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, cur)); }
  // profiling: launches of a round are recorded with per-agent byte counts and scaled once the round's
  // active-agent count is known (the host runs one round ahead of the device)
  size_t mark() const { return pending.size(); }
  void scale_bytes(size_t from, size_t to, double n) { for (size_t i = from; i < to && i < pending.size(); ++i) pending[i].bytes *= n; }
  void destroy() {
    (void)hipSetDevice(device);
    for (auto& p : pending) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : event_pool) (void)hipEventDestroy(e);
    pending.clear(); event_pool.clear();
    if (own_stream && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    MAS_HIP_CHECK(hipSetDevice(device));
    MAS_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 8));
    MAS_HIP_CHECK(hipMemsetAsync(p, 0, bytes ? bytes : 8, stream));
    return p;
  }
  void free(void* p) { if (p) (void)hipFree(p); }
  // pinned, device-mapped and COHERENT (fine-grained): the commit kernel publishes a round's counters into it (Publish, mas_bodies.hpp)
  // and the host reads them after the round's event -- stated explicitly rather than left to HIP_HOST_COHERENT's default
  void* alloc_pinned(size_t bytes) { void* p = nullptr; MAS_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 8, hipHostMallocMapped | hipHostMallocCoherent)); return p; }
  void free_pinned(void* p) { if (p) (void)hipHostFree(p); }
  void h2d(void* d, const void* s, size_t n) { MAS_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream)); MAS_HIP_CHECK(hipStreamSynchronize(stream)); }
  void d2h(void* d, const void* s, size_t n) { MAS_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream)); MAS_HIP_CHECK(hipStreamSynchronize(stream)); }
  void d2d(void* d, const void* s, size_t n) { MAS_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, cur)); }
  void zero(void* d, size_t n) { MAS_HIP_CHECK(hipMemsetAsync(d, 0, n, cur)); }
  void fill32(int* d, int v, size_t count) { MAS_HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d), v, count, cur)); }
  void sync() { MAS_HIP_CHECK(hipStreamSynchronize(stream)); }

  hipEvent_t get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e; MAS_HIP_CHECK(hipEventCreate(&e)); return e;
  }
  // When profiling, every launch carries its own start/stop events (hipExtLaunchKernelGGL), i.e. the
  // time between them is the kernel's execution on the stream -- the quantity rocprofv3 --kernel-trace reports.
  void begin(const char* name, double bytes) {
    ev0 = nullptr; ev1 = nullptr;
    if (!profiling) return;
    Pending p; p.name = name; p.bytes = bytes; p.e0 = get_event(); p.e1 = get_event();
    pending.push_back(p);
    ev0 = p.e0; ev1 = p.e1;
  }
  void end() {}
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  template <class K, class... Args>
  void launch_kernel(K kernel, dim3 grid, dim3 block, Args... args) {
    hipExtLaunchKernelGGL(kernel, grid, block, 0, cur, ev0, ev1, 0, args...);
    MAS_HIP_CHECK(hipGetLastError());
  }
  template <class K, class... Args>
  void launch_kernel_lds(K kernel, dim3 grid, dim3 block, size_t lds_bytes, Args... args) {
    // (above 32 KB: dynamic + the kernel's static LDS may pass the 64-KB default together)
    if (lds_bytes > 32 * 1024) MAS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes)));
    hipExtLaunchKernelGGL(kernel, grid, block, static_cast<uint32_t>(lds_bytes), cur, ev0, ev1, 0, args...);
    MAS_HIP_CHECK(hipGetLastError());
  }
  void resolve_profile() {
    if (pending.empty()) return;
    MAS_HIP_CHECK(hipDeviceSynchronize());
    for (auto& p : pending) {
      float ms = 0.f;
      MAS_HIP_CHECK(hipEventElapsedTime(&ms, p.e0, p.e1));
      KernelStat& s = stats[p.name];
      s.launches++; s.total_ms += ms; s.algorithmic_bytes += p.bytes;
      event_pool.push_back(p.e0); event_pool.push_back(p.e1);
    }
    pending.clear();
  }

  template <int BY, class F>
  void launch_1d(const char* name, const F& f, int n, double alg_bytes = 0.0) { launch_1d_range<BY>(name, f, 0, n, alg_bytes); }
  template <int BY, class F>
  void launch_2d(const char* name, const F& f, int n, int ny, double alg_bytes = 0.0) { launch_2d_range<BY>(name, f, 0, n, ny, alg_bytes); }
  template <int BY, class F>
  void launch_1d_range(const char* name, const F& f, int lo, int hi, double alg_bytes = 0.0) {
    if (hi <= lo) return;
    begin(name, alg_bytes);
    launch_kernel(mas_kernel_1d<F>, dim3((hi - lo + 63) / 64), dim3(64), f, lo, hi);
    end();
  }
  template <int BY, class F>
  void launch_2d_range(const char* name, const F& f, int lo, int hi, int ny, double alg_bytes = 0.0) {
    if (hi <= lo || ny <= 0) return;
    begin(name, alg_bytes);
    launch_kernel(mas_kernel_2d<F, BY>, dim3((hi - lo + 63) / 64, (ny + BY - 1) / BY), dim3(64, BY), f, lo, hi, ny);
    end();
  }
  // force_threads: workgroup size when it is not one lane per item (the body shares its loops out over the threads itself)
  template <class F>
  void launch_group(const char* name, const F& f, int n, double alg_bytes = 0.0, size_t lds_bytes = 0, int force_threads = 0) {
    if (n <= 0) return;
    begin(name, alg_bytes);
    const int threads = force_threads > 0 ? force_threads : ((n + 63) / 64) * 64;
    if (threads <= 256) launch_kernel_lds(mas_kernel_group<F, 256>, dim3(1), dim3(threads), lds_bytes, f, n);
    else if (threads <= 512) launch_kernel_lds(mas_kernel_group<F, 512>, dim3(1), dim3(threads), lds_bytes, f, n);
    else launch_kernel_lds(mas_kernel_group<F, 1024>, dim3(1), dim3(threads), lds_bytes, f, n);
    end();
  }
};

#endif

}  // namespace mas
