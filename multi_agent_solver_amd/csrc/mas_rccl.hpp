// mas_rccl.hpp -- RCCL inside the library, for agents sharded over the GPUs of a node (HIP build only).
//
// The reference fans its agents out over OpenMP threads (strategies/nash.hpp:59-71); here the id-sorted block list
// (multi_agent_problem.hpp:37-50) is cut into contiguous ranges, one per GPU, and the two exchanges the Nash strategies have --
// the publish step (every rank receives every agent's trajectory) and the total cost of ALL agents that the line-search strategy
// accepts a joint step on (nash.hpp:45-49,119,134-143) -- run over RCCL on device buffers, on the handle's stream.
// librccl.so.1 is opened on first use (dlopen), so libmas_hip.so loads and runs single-GPU without it; communicators made by
// these entry points are used by these entry points only (never mixed with another RCCL instance of the process).
#pragma once

#include <dlfcn.h>

#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace mas {

// the slice of rccl.h this file needs (ABI-stable NCCL types)
using ncclComm_t = struct ncclComm*;
struct ncclUniqueId { char internal[128]; };
enum : int { kNcclSuccess = 0, kNcclDouble = 8 /* ncclFloat64 */ };

struct RcclApi {
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  std::string error;

  static RcclApi& get() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { api.load(); });
    return api;
  }
  bool ok() const { return handle != nullptr && error.empty(); }

 private:
  void load() {
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);       // (local: its symbols are reached through dlsym only)
      if (handle) break;
    }
    if (!handle) { error = std::string("librccl.so.1 could not be opened: ") + dlerror(); return; }
    auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p && error.empty()) error = std::string("librccl lacks ") + n; return p; };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
  }
};

// contiguous id range [lo, hi) of `rank`; sizes differ by at most one agent (the same rule as multi_agent_solver_amd/dist.py)
inline void shard_range(int n_total, int n_ranks, int rank, int* lo, int* hi) {
  const int base = n_total / n_ranks, extra = n_total % n_ranks;
  *lo = rank * base + (rank < extra ? rank : extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
}

struct Comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0, device = 0;

  void check(int rc, const char* what) const {
    if (rc != kNcclSuccess) {
      const RcclApi& api = RcclApi::get();
      throw std::runtime_error(std::string(what) + " failed: " + (api.GetErrorString ? api.GetErrorString(rc) : "rccl error"));
    }
  }
  // all-gather of the ranks' shards of `per_agent` doubles per agent into ONE contiguous [n_total][per_agent] device buffer, in
  // place (rank r's shard already lies at its displacement).  Equal shards: one ncclAllGather; uneven shards: one broadcast per
  // rank inside a group -- by displacement, nothing is padded.
  void all_gather_shards(double* all, int n_total, size_t per_agent, hipStream_t stream) const {
    const RcclApi& api = RcclApi::get();
    int lo = 0, hi = 0;
    if (n_total % n_ranks == 0) {
      shard_range(n_total, n_ranks, rank, &lo, &hi);
      const size_t count = static_cast<size_t>(hi - lo) * per_agent;
      check(api.AllGather(all + static_cast<size_t>(lo) * per_agent, all, count, kNcclDouble, comm, stream), "ncclAllGather");
      return;
    }
    check(api.GroupStart(), "ncclGroupStart");
    for (int r = 0; r < n_ranks; ++r) {
      shard_range(n_total, n_ranks, r, &lo, &hi);
      double* p = all + static_cast<size_t>(lo) * per_agent;
      check(api.Broadcast(p, p, static_cast<size_t>(hi - lo) * per_agent, kNcclDouble, r, comm, stream), "ncclBroadcast");
    }
    check(api.GroupEnd(), "ncclGroupEnd");
  }
};

}  // namespace mas
