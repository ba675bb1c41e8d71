// mas_capi.hip -- extern "C" boundary of libmas_hip.so (declared in include/mas_hip.h).
//
// Every entry point catches C++ exceptions and turns them into integer status codes; the
// message is kept on the handle (mas_hip_last_error).  Without a HIP device create() fails
// with MAS_HIP_ERR_NO_DEVICE -- there is no CPU fallback in the product.
#include "../../include/mas_hip.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <sstream>
#include <algorithm>
#include <vector>
#include <dirent.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mas_solver.hpp"

struct mas_hip_comm { mas::Comm c; };

struct mas_hip_solver {
  std::unique_ptr<mas::SolverBase> impl;
  int N = 0, T = 0;
};

namespace {
thread_local std::string g_create_error;

template <class Fn>
int guarded(mas_hip_solver* h, Fn&& fn) {
  if (!h || !h->impl) return MAS_HIP_ERR_INVALID;
  try {
    fn(*h->impl);
    return MAS_HIP_OK;
  } catch (const std::invalid_argument& e) {
    h->impl->last_error = e.what();
    return MAS_HIP_ERR_INVALID;
  } catch (const std::logic_error& e) {
    h->impl->last_error = e.what();
    return MAS_HIP_ERR_STATE;
  } catch (const std::exception& e) {
    h->impl->last_error = e.what();
    return MAS_HIP_ERR_RUNTIME;
  } catch (...) {
    h->impl->last_error = "unknown exception";
    return MAS_HIP_ERR_RUNTIME;
  }
}

int create_impl(const mas_hip_problem_desc* desc, void* stream, mas_hip_solver** out) {
  if (!desc || !out) { g_create_error = "null argument"; return MAS_HIP_ERR_INVALID; }
  *out = nullptr;
  int nx = 0, nu = 0;
  if (!mas::model_dims(desc->model_id, &nx, &nu)) { g_create_error = "unknown model id"; return MAS_HIP_ERR_INVALID; }
  if (desc->n_agents < 1 || desc->horizon_steps < 1 || !(desc->dt > 0.0)) { g_create_error = "n_agents, horizon_steps and dt must be positive"; return MAS_HIP_ERR_INVALID; }
  if ((desc->input_lower_bounds == nullptr) != (desc->input_upper_bounds == nullptr)) { g_create_error = "input bounds must be given as a pair"; return MAS_HIP_ERR_INVALID; }
  const int ndev = mas::Backend::device_count();
  if (ndev <= 0) { g_create_error = "no HIP device visible: libmas_hip has no CPU fallback"; return MAS_HIP_ERR_NO_DEVICE; }
  if (desc->device < 0 || desc->device >= ndev) { g_create_error = "device ordinal out of range"; return MAS_HIP_ERR_INVALID; }
  mas::ProblemDesc pd;
  pd.model_id = desc->model_id; pd.n_agents = desc->n_agents; pd.horizon_steps = desc->horizon_steps; pd.device = desc->device; pd.dt = desc->dt;
  if (desc->params) { pd.has_params = true; std::memcpy(pd.params, desc->params, sizeof(double) * 16); }
  if (desc->input_lower_bounds) { pd.has_bounds = true; for (int i = 0; i < nu; ++i) { pd.lb[i] = desc->input_lower_bounds[i]; pd.ub[i] = desc->input_upper_bounds[i]; } }
  pd.no_bounds = desc->no_input_bounds != 0;
  try {
    auto h = new mas_hip_solver();
    h->impl = mas::make_solver(pd, stream);
    h->N = desc->n_agents; h->T = desc->horizon_steps;
    *out = h;
    return MAS_HIP_OK;
  } catch (const std::exception& e) {
    g_create_error = e.what();
    return MAS_HIP_ERR_RUNTIME;
  }
}
thread_local std::string g_comm_error;
template <class Fn>
int comm_guarded(Fn&& fn) {
  try {
    mas::RcclApi& api = mas::RcclApi::get();
    if (!api.ok()) { g_comm_error = api.error.empty() ? "librccl not available" : api.error; return MAS_HIP_ERR_RUNTIME; }
    fn(api);
    return MAS_HIP_OK;
  } catch (const std::invalid_argument& e) { g_comm_error = e.what(); return MAS_HIP_ERR_INVALID;
  } catch (const std::exception& e) { g_comm_error = e.what(); return MAS_HIP_ERR_RUNTIME; }
}
// ---- user-defined workloads (mas_hip_register_model) ----
// The reference takes a workload as callbacks (types.hpp:21-50); device code cannot call those, so a workload here is a struct of
// static functions the kernel templates are instantiated with.  A new one does not need a rebuild of this library: its source is
// compiled -- host driver and kernels, the very headers this library was built from -- into a plugin library by the ROCm compiler
// at run time, loaded, and its factory entered in the registry that mas_hip_create consults.
#if !defined(MAS_HOST_EMULATION)
std::mutex g_register_mutex;
std::string directory_of_this_library() {
  Dl_info info;
  if (!dladdr(reinterpret_cast<const void*>(&directory_of_this_library), &info) || !info.dli_fname) return std::string();
  std::string path = info.dli_fname;
  const size_t cut = path.find_last_of('/');
  return cut == std::string::npos ? std::string(".") : path.substr(0, cut);
}
bool file_exists(const std::string& path) { std::ifstream f(path); return f.good(); }
std::string tail_of_file(const std::string& path, size_t bytes) {
  std::ifstream f(path); std::stringstream ss; ss << f.rdbuf();
  const std::string all = ss.str();
  return all.size() > bytes ? all.substr(all.size() - bytes) : all;
}
int register_model_impl(const char* struct_source, const char* struct_name, const char* source_root, int* model_id) {
  if (!struct_source || !struct_name || !model_id) { g_create_error = "null argument"; return MAS_HIP_ERR_INVALID; }
  for (const char* c = struct_name; *c; ++c)
    if (!((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9' && c != struct_name) || *c == '_')) { g_create_error = "struct_name is not an identifier"; return MAS_HIP_ERR_INVALID; }
  if (source_root) for (const char* c = source_root; *c; ++c) if (*c == '\'' || *c == '\n') { g_create_error = "source_root must not contain quotes or line breaks"; return MAS_HIP_ERR_INVALID; }
  // (the registry, dlopen and the cache directory are touched under the mutex; the compiler runs outside it, so several workloads
  // registered from several threads compile side by side)
  std::unique_lock<std::mutex> lock(g_register_mutex);
  // the headers: <root>/multi_agent_solver_amd/csrc and <root>/include; by default the tree this library was built in
  std::string root = source_root ? source_root : "";
  if (root.empty()) { const std::string lib = directory_of_this_library(); root = lib + "/../.."; }
  const std::string csrc = root + "/multi_agent_solver_amd/csrc", inc = root + "/include";
  if (!file_exists(csrc + "/mas_solver.hpp") || !file_exists(inc + "/mas_dettrig.h")) { g_create_error = "kernel headers not found under " + root + " (pass source_root)"; return MAS_HIP_ERR_INVALID; }
  // compiled models are kept: the key is a hash of the struct, its name and every header it is compiled against, so a process that registers
  // the same workload again (every start of an application) loads the plugin instead of waiting a minute for the compiler
  unsigned long long key = 1469598103934665603ull;
  auto mix = [&](const std::string& bytes) { for (unsigned char c : bytes) { key ^= c; key *= 1099511628211ull; } key ^= 0xffu; key *= 1099511628211ull; };
  const char* env_cc = std::getenv("HIPCC");
  const std::string cc = env_cc ? env_cc : (file_exists("/opt/rocm/bin/hipcc") ? "/opt/rocm/bin/hipcc" : "hipcc");
  mix(struct_source); mix(struct_name); mix(cc); mix("--offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off");
  for (const std::string& hdir : {csrc, inc}) {
    std::vector<std::string> names;
    if (DIR* dp = opendir(hdir.c_str())) { while (dirent* e = readdir(dp)) { const std::string n = e->d_name; if (n.size() > 2 && n[0] != '.') names.push_back(n); } closedir(dp); }
    std::sort(names.begin(), names.end());
    for (const std::string& n : names) { if (n.find(".h") == std::string::npos && n.find(".inc") == std::string::npos) continue; mix(n); std::ifstream f(hdir + "/" + n, std::ios::binary); std::stringstream ss; ss << f.rdbuf(); mix(ss.str()); }
  }
  const char* env_cache = std::getenv("MAS_HIP_MODEL_CACHE");
  const std::string cache_dir = env_cache ? env_cache : ("/tmp/mas_hip_models_" + std::to_string(static_cast<unsigned long>(getuid())));
  (void)mkdir(cache_dir.c_str(), 0700);
  char keyhex[32]; std::snprintf(keyhex, sizeof keyhex, "%016llx", key);
  const std::string cached = cache_dir + "/" + struct_name + "_" + keyhex + ".so";
  std::string so = cached;
  if (!file_exists(cached)) {
    char dir_template[] = "/tmp/mas_model_XXXXXX";
    const char* dir = mkdtemp(dir_template);
    if (!dir) { g_create_error = "mkdtemp failed"; return MAS_HIP_ERR_RUNTIME; }
    const std::string d = dir, src = d + "/model.hip", log = d + "/build.log";
    so = d + "/model.so";
    {
      std::ofstream f(src);
      f << "// generated by mas_hip_register_model\n#include \"mas_solver.hpp\"\nnamespace mas {\n" << struct_source << "\n}  // namespace mas\n"
        << "extern \"C\" mas::SolverBase* mas_plugin_make(const mas::ProblemDesc* desc, void* stream) { return new mas::SolverImpl<mas::" << struct_name << ">(*desc, stream); }\n"
        << "extern \"C\" void mas_plugin_dims(int* nx, int* nu) { *nx = mas::" << struct_name << "::NX; *nu = mas::" << struct_name << "::NU; }\n";
    }
    // the options libmas_hip.so itself is built with (__graft_entry__.py): nothing is contracted, IEEE division and square root
    const std::string cmd = "'" + cc + "' --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fPIC -shared -I'" + csrc + "' -I'" + inc + "' '" + src + "' -o '" + so + "' > '" + log + "' 2>&1";
    lock.unlock();
    const int compiler_status = std::system(cmd.c_str());
    lock.lock();
    if (compiler_status != 0 || !file_exists(so)) { g_create_error = "the model did not compile (" + log + "):\n" + tail_of_file(log, 2000); return MAS_HIP_ERR_INVALID; }
    std::remove(src.c_str()); std::remove(log.c_str());
    if (std::rename(so.c_str(), cached.c_str()) == 0) { so = cached; (void)rmdir(d.c_str()); }      // (a rename inside one file system; otherwise the plugin is loaded from where it was built)
  }
  void* lib = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!lib) { g_create_error = std::string("dlopen of the compiled model failed: ") + dlerror(); return MAS_HIP_ERR_RUNTIME; }
  mas::RegisteredModel rm;
  rm.make = reinterpret_cast<mas::SolverBase* (*)(const mas::ProblemDesc*, void*)>(dlsym(lib, "mas_plugin_make"));
  auto dims = reinterpret_cast<void (*)(int*, int*)>(dlsym(lib, "mas_plugin_dims"));
  if (!rm.make || !dims) { dlclose(lib); g_create_error = "the compiled model lacks its factory"; return MAS_HIP_ERR_RUNTIME; }
  dims(&rm.nx, &rm.nu);
  if (rm.nx < 1 || rm.nx > 8 || rm.nu < 1 || rm.nu > 8) { dlclose(lib); g_create_error = "NX and NU must lie in 1 .. 8"; return MAS_HIP_ERR_INVALID; }
  rm.library = lib; rm.name = struct_name;
  const int id = mas::register_model_entry(rm);
  if (id < 0) { dlclose(lib); g_create_error = "model registry full"; return MAS_HIP_ERR_RUNTIME; }
  *model_id = id;
  return MAS_HIP_OK;
}
#else
int register_model_impl(const char*, const char*, const char*, int*) { g_create_error = "the host emulation registers no models"; return MAS_HIP_ERR_RUNTIME; }
#endif

void select_device(int device) {
#if !defined(MAS_HOST_EMULATION)
  MAS_HIP_CHECK(hipSetDevice(device));
#else
  (void)device;
#endif
}
}  // namespace

extern "C" {

int mas_hip_abi_version(void) { return MAS_HIP_ABI_VERSION; }
int mas_hip_device_count(void) { return mas::Backend::device_count(); }
int mas_hip_model_dims(int model_id, int* nx, int* nu) { return mas::model_dims(model_id, nx, nu) ? MAS_HIP_OK : MAS_HIP_ERR_INVALID; }
int mas_hip_model_default_params(int model_id, int horizon_steps, double* params16) {
  if (!params16 || !mas::model_dims(model_id, nullptr, nullptr)) return MAS_HIP_ERR_INVALID;
  mas::model_default_params(model_id, horizon_steps, params16);
  return MAS_HIP_OK;
}
int mas_hip_register_model(const char* struct_source, const char* struct_name, const char* source_root, int* model_id) {
  try { return register_model_impl(struct_source, struct_name, source_root, model_id); }
  catch (const std::exception& e) { g_create_error = e.what(); return MAS_HIP_ERR_RUNTIME; }
}
int mas_hip_create(const mas_hip_problem_desc* desc, mas_hip_solver** out) { return create_impl(desc, nullptr, out); }
int mas_hip_create_on_stream(const mas_hip_problem_desc* desc, void* hip_stream, mas_hip_solver** out) { return create_impl(desc, hip_stream, out); }
int mas_hip_destroy(mas_hip_solver* h) {
  if (!h) return MAS_HIP_ERR_INVALID;
  delete h;
  return MAS_HIP_OK;
}
const char* mas_hip_last_error(const mas_hip_solver* h) { return (h && h->impl) ? h->impl->last_error.c_str() : g_create_error.c_str(); }

int mas_hip_set_param(mas_hip_solver* h, const char* key, double value) {
  if (!key) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { s.set_param(key, value); });
}
int mas_hip_set_option(mas_hip_solver* h, const char* key, int64_t value) {
  if (!key) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { if (!s.set_option(key, value)) throw std::invalid_argument(std::string("unknown option or value: ") + key); });
}
int mas_hip_check_params(const mas_hip_solver* h) {
  if (!h || !h->impl) return MAS_HIP_ERR_INVALID;
  if (!h->impl->params_complete()) { h->impl->last_error = "SolverParams lacks max_iterations, tolerance or max_ms (iLQR::set_params uses params.at())"; return MAS_HIP_ERR_INVALID; }
  return MAS_HIP_OK;
}

int mas_hip_set_initial_states_host(mas_hip_solver* h, const double* x0) {
  if (!x0) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { s.set_initial_states(x0, false); });
}
int mas_hip_set_controls_host(mas_hip_solver* h, const double* U) { return guarded(h, [&](mas::SolverBase& s) { s.set_controls(U, false); }); }
int mas_hip_set_initial_states_device(mas_hip_solver* h, const void* x0) {
  if (!x0) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { s.set_initial_states(static_cast<const double*>(x0), true); });
}
int mas_hip_set_controls_device(mas_hip_solver* h, const void* U) { return guarded(h, [&](mas::SolverBase& s) { s.set_controls(static_cast<const double*>(U), true); }); }
int mas_hip_initialize_problem(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.initialize_problem(); s.be.sync(); }); }
int mas_hip_save_initial(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.save_initial(); s.be.sync(); }); }
int mas_hip_restore_initial(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.restore_initial(); }); }   // stream-ordered: device-to-device copies on the handle's stream, nothing the host could read

int mas_hip_solve(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.solve(); }); }
int mas_hip_run_sequential(mas_hip_solver* h, int max_outer) { return guarded(h, [&](mas::SolverBase& s) { s.run_sequential(max_outer); }); }
int mas_hip_run_trust_region(mas_hip_solver* h, int max_outer) { return guarded(h, [&](mas::SolverBase& s) { s.run_trust_region(max_outer); }); }
int mas_hip_run_line_search(mas_hip_solver* h, int max_outer) { return guarded(h, [&](mas::SolverBase& s) { s.run_line_search(max_outer); }); }
int mas_hip_run_centralized(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.run_centralized(); }); }
int mas_hip_run_centralized_dense(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.run_centralized_dense(); }); }

int mas_hip_get_solution_host(mas_hip_solver* h, double* X, double* U, double* cost, int32_t* iterations, int32_t* exit_status) {
  return guarded(h, [&](mas::SolverBase& s) { s.get_solution(X, U, cost, iterations, exit_status, false); });
}
int mas_hip_get_solution_device(mas_hip_solver* h, void* X, void* U, void* cost) {
  return guarded(h, [&](mas::SolverBase& s) { s.get_solution(static_cast<double*>(X), static_cast<double*>(U), static_cast<double*>(cost), nullptr, nullptr, true); });
}
int mas_hip_total_cost(mas_hip_solver* h, double* total) {
  if (!total) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { *total = s.total_cost(); });
}
int mas_hip_set_cost_exchange(mas_hip_solver* h, mas_hip_cost_exchange_fn fn, void* ctx) {
  return guarded(h, [&](mas::SolverBase& s) { s.cost_exchange = fn; s.cost_exchange_ctx = ctx; });
}
// ---- communicators (mas_rccl.hpp): librccl, or the in-process loopback transport when it was selected ----
const char* mas_hip_comm_last_error(void) { return g_comm_error.c_str(); }
int mas_hip_comm_use_loopback(int enabled) { mas::RcclApi::use_loopback(enabled != 0); return MAS_HIP_OK; }
int mas_hip_comm_unique_id(void* id_out) {
  if (!id_out) return MAS_HIP_ERR_INVALID;
  return comm_guarded([&](mas::RcclApi& api) {
    mas::ncclUniqueId id;
    const int rc = api.GetUniqueId(&id);
    if (rc != mas::kNcclSuccess) throw std::runtime_error(std::string("ncclGetUniqueId failed: ") + api.GetErrorString(rc));
    std::memcpy(id_out, id.internal, sizeof(id.internal));
  });
}
int mas_hip_comm_create(const void* id, int n_ranks, int rank, int device, mas_hip_comm** out) {
  if (!id || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks) return MAS_HIP_ERR_INVALID;
  *out = nullptr;
  return comm_guarded([&](mas::RcclApi& api) {
    select_device(device);
    mas::ncclUniqueId uid;
    std::memcpy(uid.internal, id, sizeof(uid.internal));
    auto c = std::make_unique<mas_hip_comm>();
    c->c.n_ranks = n_ranks; c->c.rank = rank; c->c.device = device; c->c.api = &api;
    c->c.check(api.CommInitRank(&c->c.comm, n_ranks, uid, rank), "ncclCommInitRank");
    *out = c.release();
  });
}
int mas_hip_comm_create_all(int n_devices, const int* devices, mas_hip_comm** out) {
  if (!out || n_devices < 1) return MAS_HIP_ERR_INVALID;
  for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
  return comm_guarded([&](mas::RcclApi& api) {
    std::vector<int> devs(n_devices);
    for (int i = 0; i < n_devices; ++i) devs[i] = devices ? devices[i] : i;
    std::vector<mas::ncclComm_t> comms(n_devices, nullptr);
    mas::Comm probe; probe.api = &api;
    probe.check(api.CommInitAll(comms.data(), n_devices, devs.data()), "ncclCommInitAll");
    for (int i = 0; i < n_devices; ++i) {
      out[i] = new mas_hip_comm();
      out[i]->c.comm = comms[i]; out[i]->c.n_ranks = n_devices; out[i]->c.rank = i; out[i]->c.device = devs[i]; out[i]->c.api = &api;
    }
  });
}
int mas_hip_comm_destroy(mas_hip_comm* c) {
  if (!c) return MAS_HIP_ERR_INVALID;
  int rc = MAS_HIP_OK;
  try { if (c->c.comm) c->c.check(c->c.table().CommDestroy(c->c.comm), "ncclCommDestroy"); }
  catch (const std::exception& e) { g_comm_error = e.what(); rc = MAS_HIP_ERR_RUNTIME; }
  delete c;
  return rc;
}
// a rank that cannot go on (its upload failed, its handle was refused, its solve gave up) releases every peer that waits for it in
// a collective -- ncclCommAbort; may be called from another thread than the one driving the rank; the object stays valid (every
// later collective on it fails at once) until mas_hip_comm_destroy
int mas_hip_comm_abort(mas_hip_comm* c) {
  if (!c) return MAS_HIP_ERR_INVALID;
  try { c->c.abort(); return MAS_HIP_OK; }
  catch (const std::exception& e) { g_comm_error = e.what(); return MAS_HIP_ERR_RUNTIME; }
}
void mas_hip_shard_range(int n_total, int n_ranks, int rank, int* lo, int* hi) {
  int a = 0, b = 0;
  if (n_ranks > 0 && rank >= 0 && rank < n_ranks) mas::shard_range(n_total, n_ranks, rank, &a, &b);
  if (lo) *lo = a;
  if (hi) *hi = b;
}
int mas_hip_all_gather_solution(mas_hip_solver* h, mas_hip_comm* c, int n_total, void* X_all, void* U_all, void* cost_all) {
  if (!c) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) {
    if (!c->c.table().ok()) throw std::runtime_error("librccl not available: " + c->c.table().error);
    s.all_gather_solution(&c->c, n_total, static_cast<double*>(X_all), static_cast<double*>(U_all), static_cast<double*>(cost_all));
  });
}
int mas_hip_set_comm_cost_exchange(mas_hip_solver* h, mas_hip_comm* c, int n_total) {
  return guarded(h, [&](mas::SolverBase& s) { s.set_comm_exchange(c ? &c->c : nullptr, n_total); });
}
int mas_hip_get_trust_radii_host(mas_hip_solver* h, double* radii) {
  if (!radii) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) { s.get_trust_radii(radii); });
}
int mas_hip_debug_backward_host(mas_hip_solver* h, double* derivs, double* k, double* K, int64_t* reg_events) {
  return guarded(h, [&](mas::SolverBase& s) { s.debug_backward(derivs, k, K, reg_events); });
}

int mas_hip_get_stats(mas_hip_solver* h, mas_hip_stats* out) {
  if (!out) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) {
    const mas::HostStats st = s.get_stats();
    out->agent_iterations = st.agent_iterations; out->converged_solves = st.converged_solves; out->solves = st.solves;
    out->reg_events = st.reg_events; out->batch_iterations = st.batch_iterations; out->wall_ms = st.wall_ms;
  });
}
int mas_hip_reset_stats(mas_hip_solver* h) { return guarded(h, [&](mas::SolverBase& s) { s.reset_stats(); s.be.stats.clear(); }); }

int mas_hip_set_profiling(mas_hip_solver* h, int enabled) {
  return guarded(h, [&](mas::SolverBase& s) { s.be.resolve_profile(); s.be.profiling = enabled != 0; });
}
int mas_hip_get_kernel_times(mas_hip_solver* h, mas_hip_kernel_time* out, int capacity, int* count) {
  if (!count) return MAS_HIP_ERR_INVALID;
  return guarded(h, [&](mas::SolverBase& s) {
    s.be.resolve_profile();
    s.fold_kernel_stamps();
    int n = 0;
    for (const auto& kv : s.be.stats) {
      if (out && n < capacity) {
        std::memset(&out[n], 0, sizeof(out[n]));
        std::strncpy(out[n].name, kv.first.c_str(), sizeof(out[n].name) - 1);
        out[n].launches = kv.second.launches; out[n].total_ms = kv.second.total_ms; out[n].algorithmic_bytes = kv.second.algorithmic_bytes;
      }
      ++n;
    }
    *count = n;
  });
}

}  // extern "C"
