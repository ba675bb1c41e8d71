// mas_rccl.hpp -- RCCL inside the library, for agents sharded over the GPUs of a node.
//
// The reference fans its agents out over OpenMP threads (strategies/nash.hpp:59-71); here the id-sorted block list
// (multi_agent_problem.hpp:37-50) is cut into contiguous ranges, one per GPU, and the two exchanges the Nash strategies have --
// the publish step (every rank receives every agent's trajectory) and the total cost of ALL agents that the line-search strategy
// accepts a joint step on (nash.hpp:45-49,119,134-143) -- run over RCCL on device buffers, on the handle's stream.
// librccl.so.1 is opened on first use (dlopen), so libmas_hip.so loads and runs single-GPU without it; communicators made by
// these entry points are used by these entry points only (never mixed with another RCCL instance of the process).
//
// The collectives are reached through a table of function pointers (RcclApi).  Besides librccl there is ONE other provider, the
// LOOPBACK transport below: N ranks as host threads of one process, rendezvous through a mutex and a condition variable, copies
// between the ranks' buffers.  It exists so that everything above the table -- displacements of even and uneven shards, the
// grouped broadcasts, communicator creation for several devices, the total cost of the line search over all ranks -- runs with
// more than one rank where no second GPU exists: in the CPU tier (host emulation: the "device" buffers are host memory) and with
// several shards on ONE GPU (which RCCL itself refuses).  Selected by mas_hip_comm_use_loopback (include/mas_hip.h); never the default.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(MAS_HOST_EMULATION)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#endif

namespace mas {

#if defined(MAS_HOST_EMULATION)
using CommStream = void*;
#else
using CommStream = hipStream_t;
#endif

// the slice of rccl.h this file needs (ABI-stable NCCL types)
using ncclComm_t = struct ncclComm*;
struct ncclUniqueId { char internal[128]; };
enum : int { kNcclSuccess = 0, kNcclSystemError = 2, kNcclInvalidArgument = 4, kNcclDouble = 8 /* ncclFloat64 */ };

struct RcclApi {
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*CommAbort)(ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, CommStream) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, CommStream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  void* handle = nullptr;
  bool loopback = false;
  std::string error;

  // the table in use: librccl (loaded on first use) unless the loopback transport was selected
  static RcclApi& get() { return use_loopback_flag() ? loopback_table() : rccl_table(); }
  static void use_loopback(bool on) { use_loopback_flag() = on; }
  bool ok() const { return (loopback || handle != nullptr) && error.empty(); }

 private:
  static bool& use_loopback_flag() { static bool on = false; return on; }
  static RcclApi& rccl_table() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { api.load(); });
    return api;
  }
  static RcclApi& loopback_table();
  void load() {
#if defined(MAS_HOST_EMULATION)
    error = "no RCCL in the host emulation (only the loopback transport)";
#else
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
    CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    Broadcast = reinterpret_cast<decltype(Broadcast)>(sym("ncclBroadcast"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
#endif
  }
};

// contiguous id range [lo, hi) of `rank`; sizes differ by at most one agent (the same rule as multi_agent_solver_amd/dist.py)
inline void shard_range(int n_total, int n_ranks, int rank, int* lo, int* hi) {
  const int base = n_total / n_ranks, extra = n_total % n_ranks;
  *lo = rank * base + (rank < extra ? rank : extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
}

// What one collective of all_gather_shards moves, as data: (root rank or -1 for the all-gather, offset and count in doubles).  Kept
// apart from the calls so that the displacement arithmetic can be checked on its own (tests/hostcheck/shards_check.cpp).
struct ShardPiece { int root; size_t offset, count; };
inline std::vector<ShardPiece> shard_pieces(int n_total, int n_ranks, int rank, size_t per_agent) {
  std::vector<ShardPiece> out;
  int lo = 0, hi = 0;
  if (n_total % n_ranks == 0) {                     // equal shards: ONE in-place all-gather, this rank's shard at its displacement
    shard_range(n_total, n_ranks, rank, &lo, &hi);
    out.push_back({-1, static_cast<size_t>(lo) * per_agent, static_cast<size_t>(hi - lo) * per_agent});
    return out;
  }
  for (int r = 0; r < n_ranks; ++r) {               // uneven shards: one in-place broadcast per root inside a group -- by displacement, nothing is padded
    shard_range(n_total, n_ranks, r, &lo, &hi);
    out.push_back({r, static_cast<size_t>(lo) * per_agent, static_cast<size_t>(hi - lo) * per_agent});
  }
  return out;
}

struct Comm {
  ncclComm_t comm = nullptr;
  int n_ranks = 1, rank = 0, device = 0;
  const RcclApi* api = nullptr;          // the table the communicator was made by (librccl or the loopback transport)
  // abort() may come from ANOTHER thread than the one that drives this rank (a failing shard releases its peers): librccl frees the
  // communicator in ncclCommAbort, so enqueueing and aborting exclude each other; the loopback transport only raises a flag on the group
  mutable std::mutex use;
  std::atomic<bool> aborted{false};

  const RcclApi& table() const { return api ? *api : RcclApi::get(); }
  void abort() {
    aborted.store(true);
    const RcclApi& t = table();
    if (t.loopback) { if (comm) t.CommAbort(comm); return; }          // (flags the group, frees nothing)
    std::lock_guard<std::mutex> lk(use);
    if (comm) { ncclComm_t c = comm; comm = nullptr; check(t.CommAbort ? t.CommAbort(c) : t.CommDestroy(c), "ncclCommAbort"); }
  }
  void check(int rc, const char* what) const {
    if (rc != kNcclSuccess) {
      const RcclApi& t = table();
      throw std::runtime_error(std::string(what) + " failed: " + (t.GetErrorString ? t.GetErrorString(rc) : "rccl error"));
    }
  }
  // all-gather of the ranks' shards of `per_agent` doubles per agent into ONE contiguous [n_total][per_agent] device buffer, in
  // place (rank r's shard already lies at its displacement).
  void all_gather_shards(double* all, int n_total, size_t per_agent, CommStream stream) const {
    const RcclApi& t = table();
    std::unique_lock<std::mutex> lk(use, std::defer_lock);
    if (!t.loopback) lk.lock();                       // (a loopback collective blocks until every rank has arrived: it must stay abortable)
    if (aborted.load() || !comm) throw std::runtime_error("the communicator was aborted (a rank of the group failed)");
    const std::vector<ShardPiece> pieces = shard_pieces(n_total, n_ranks, rank, per_agent);
    if (pieces.size() == 1 && pieces[0].root < 0) {
      check(t.AllGather(all + pieces[0].offset, all, pieces[0].count, kNcclDouble, comm, stream), "ncclAllGather");
      return;
    }
    check(t.GroupStart(), "ncclGroupStart");
    for (const ShardPiece& p : pieces) check(t.Broadcast(all + p.offset, all + p.offset, p.count, kNcclDouble, p.root, comm, stream), "ncclBroadcast");
    check(t.GroupEnd(), "ncclGroupEnd");
  }
};

// ------------------------------------------------------------------------------------------------- loopback transport --
// ncclComm of the loopback transport: ranks are host threads of this process.  A collective is a rendezvous of all ranks of the
// group (every rank first waits for its own stream, so its contribution is complete), then every rank copies what it receives
// from its peers' buffers, then a second rendezvous (nobody overwrites a buffer a peer still reads).  A rank that fails calls
// CommAbort: every waiter of the group wakes up and every later collective on it returns an error at once -- no rank waits forever.
namespace loopback {

struct Group {
  int n = 0;
  std::mutex m;
  std::condition_variable cv;
  int arrived = 0; long generation = 0; bool aborted = false; int joined = 0, alive = 0;
  struct Slot { const void* send = nullptr; void* recv = nullptr; size_t count = 0; };
  std::vector<Slot> slots;
  // all ranks of the group meet; false when the group was aborted
  bool barrier() {
    std::unique_lock<std::mutex> lk(m);
    if (aborted) return false;
    const long gen = generation;
    if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); return true; }
    cv.wait(lk, [&] { return generation != gen || aborted; });
    return !aborted;
  }
  void abort() { std::lock_guard<std::mutex> lk(m); aborted = true; cv.notify_all(); }
};
struct Op { bool gather; const void* send; void* recv; size_t count; int root; CommStream stream; };

}  // namespace loopback

struct ncclComm {
  std::shared_ptr<loopback::Group> group;
  int rank = 0;
  std::vector<loopback::Op> queued;       // collectives issued inside GroupStart / GroupEnd (per thread: a rank is a thread)
};

namespace loopback {

inline std::mutex& registry_mutex() { static std::mutex m; return m; }
inline std::map<unsigned long long, std::shared_ptr<Group>>& registry() { static std::map<unsigned long long, std::shared_ptr<Group>> r; return r; }
inline int& group_depth() { static thread_local int depth = 0; return depth; }
inline std::vector<ncclComm_t>& group_comms() { static thread_local std::vector<ncclComm_t> v; return v; }

inline void copy_bytes(void* dst, const void* src, size_t bytes) {
  if (dst == src || bytes == 0) return;
#if defined(MAS_HOST_EMULATION)
  std::memcpy(dst, src, bytes);
#else
  if (hipMemcpy(dst, src, bytes, hipMemcpyDefault) != hipSuccess) throw std::runtime_error("loopback transport: hipMemcpy failed");
#endif
}
inline void wait_stream(CommStream s) {
#if defined(MAS_HOST_EMULATION)
  (void)s;
#else
  if (hipStreamSynchronize(s) != hipSuccess) throw std::runtime_error("loopback transport: hipStreamSynchronize failed");
#endif
}
inline int run(ncclComm_t c, const Op& op) {
  Group& g = *c->group;
  try {
    wait_stream(op.stream);
    { std::lock_guard<std::mutex> lk(g.m); g.slots[c->rank] = {op.send, op.recv, op.count}; }
    if (!g.barrier()) return kNcclSystemError;
    if (op.gather) {                                     // recv + r * count <- rank r's send
      for (int r = 0; r < g.n; ++r) copy_bytes(static_cast<char*>(op.recv) + static_cast<size_t>(r) * op.count * sizeof(double), g.slots[r].send, op.count * sizeof(double));
    } else {                                             // recv <- the root's send
      copy_bytes(op.recv, g.slots[op.root].send, op.count * sizeof(double));
    }
    if (!g.barrier()) return kNcclSystemError;
    return kNcclSuccess;
  } catch (const std::exception&) {
    g.abort();
    return kNcclSystemError;
  }
}
inline int submit(ncclComm_t c, const Op& op) {
  if (!c || !c->group) return kNcclInvalidArgument;
  if (group_depth() > 0) {
    c->queued.push_back(op);
    auto& v = group_comms();
    bool known = false; for (ncclComm_t k : v) known = known || k == c;
    if (!known) v.push_back(c);
    return kNcclSuccess;
  }
  return run(c, op);
}

inline int GetUniqueId(ncclUniqueId* id) {
  static unsigned long long next = 1;
  std::lock_guard<std::mutex> lk(registry_mutex());
  const unsigned long long key = next++;
  registry()[key] = std::make_shared<Group>();
  std::memset(id->internal, 0, sizeof(id->internal));
  std::memcpy(id->internal, "MASLOOP", 8); std::memcpy(id->internal + 8, &key, sizeof(key));
  return kNcclSuccess;
}
inline int CommInitRank(ncclComm_t* out, int n, ncclUniqueId id, int rank) {
  if (!out || std::memcmp(id.internal, "MASLOOP", 8) != 0 || n < 1 || rank < 0 || rank >= n) return kNcclInvalidArgument;
  unsigned long long key; std::memcpy(&key, id.internal + 8, sizeof(key));
  std::shared_ptr<Group> g;
  {
    std::lock_guard<std::mutex> lk(registry_mutex());
    auto it = registry().find(key);
    if (it == registry().end()) return kNcclInvalidArgument;
    g = it->second;
    std::lock_guard<std::mutex> lg(g->m);
    if (g->n == 0) { g->n = n; g->slots.resize(n); }
    if (g->n != n) return kNcclInvalidArgument;
    if (++g->joined == n) registry().erase(it);          // every rank holds the group now
    ++g->alive;
  }
  auto c = new ncclComm(); c->group = g; c->rank = rank;
  *out = c;
  return kNcclSuccess;
}
inline int CommInitAll(ncclComm_t* out, int n, const int*) {
  if (!out || n < 1) return kNcclInvalidArgument;
  auto g = std::make_shared<Group>(); g->n = n; g->slots.resize(n); g->joined = n; g->alive = n;
  for (int r = 0; r < n; ++r) { out[r] = new ncclComm(); out[r]->group = g; out[r]->rank = r; }
  return kNcclSuccess;
}
// (unlike ncclCommAbort this frees nothing: the rank's own thread may be inside a collective; CommDestroy frees)
inline int CommAbort(ncclComm_t c) { if (!c) return kNcclInvalidArgument; if (c->group) c->group->abort(); return kNcclSuccess; }
inline int CommDestroy(ncclComm_t c) { if (!c) return kNcclInvalidArgument; delete c; return kNcclSuccess; }
inline int AllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t c, CommStream s) {
  if (dtype != kNcclDouble) return kNcclInvalidArgument;
  return submit(c, Op{true, send, recv, count, 0, s});
}
inline int Broadcast(const void* send, void* recv, size_t count, int dtype, int root, ncclComm_t c, CommStream s) {
  if (dtype != kNcclDouble || !c || !c->group || root < 0 || root >= c->group->n) return kNcclInvalidArgument;
  return submit(c, Op{false, send, recv, count, root, s});
}
inline int GroupStart() { ++group_depth(); return kNcclSuccess; }
inline int GroupEnd() {
  if (group_depth() <= 0) return kNcclInvalidArgument;
  if (--group_depth() > 0) return kNcclSuccess;
  int rc = kNcclSuccess;
  std::vector<ncclComm_t> comms; comms.swap(group_comms());
  for (ncclComm_t c : comms) {                           // (one communicator per thread in every use of this file)
    std::vector<Op> ops; ops.swap(c->queued);
    for (const Op& op : ops) { const int r = run(c, op); if (r != kNcclSuccess) rc = r; }
  }
  return rc;
}
inline const char* GetErrorString(int rc) {
  return rc == kNcclSuccess ? "no error" : rc == kNcclInvalidArgument ? "invalid argument (loopback transport)" : "a rank of the group failed or aborted (loopback transport)";
}

}  // namespace loopback

inline RcclApi& RcclApi::loopback_table() {
  static RcclApi api = [] {
    RcclApi a;
    a.GetUniqueId = loopback::GetUniqueId; a.CommInitRank = loopback::CommInitRank; a.CommInitAll = loopback::CommInitAll;
    a.CommDestroy = loopback::CommDestroy; a.CommAbort = loopback::CommAbort; a.AllGather = loopback::AllGather; a.Broadcast = loopback::Broadcast;
    a.GroupStart = loopback::GroupStart; a.GroupEnd = loopback::GroupEnd; a.GetErrorString = loopback::GetErrorString;
    a.loopback = true;
    return a;
  }();
  return api;
}

}  // namespace mas
