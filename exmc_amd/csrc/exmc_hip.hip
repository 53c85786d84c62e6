// exmc_hip.hip — C ABI of libexmc_hip.so (include/exmc_hip.h) over the gfx950 kernels.
// Host control flow mirrors lib/exmc/nuts/sampler.ex (warmup schedule, dual averaging, Welford
// mass matrix are plain Erlang-float code in the reference and plain C++ here); every leapfrog,
// log-density, gradient, tree merge and U-turn test runs in the HIP kernels. There is no CPU
// fallback: without a HIP device every compute entry point returns EXMC_ERR_NO_DEVICE.
#include "../../include/exmc_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/exmc_zig_tables.h"
#include "exmc_kernels.hpp"
#include "exmc_native_tree.hpp"

using namespace exmc;

#ifdef EXMC_PLUGIN_SPLIT
// a plug-in built in parts (exmc_plugin_part.hip): the four heavy kernels are compiled in translation
// units of their own, in parallel; here they are only declared
namespace exmc {
#include "exmc_plugin_kernels.inc"
}
#endif

namespace {

#if EXMC_PROFILE_SECTIONS
hipError_t print_sections(const char* what, double ms) {
  unsigned long long h[16], z[16] = {0};
  hipError_t e = hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h));
  if (e != hipSuccess) return e;
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
  if (e != hipSuccess) return e;
  static const char* nm[9] = {"transition_start", "doubling_start", "leapfrog+model", "leaf",
                              "ascend", "outer_merge", "transition_done", "", "loop"};
  unsigned long long tot = 0;
  for (int i = 0; i < 9; i++) tot += h[i];
  fprintf(stderr, "[exmc prof] %s %.3f ms, passes (sum over waves) %llu, cycles %llu\n", what, ms, h[9], tot);
  for (int i = 0; i < 9; i++)
    if (i != 7 && h[9])
      fprintf(stderr, "[exmc prof]   %-18s %6.1f%%  %8.1f cycles/pass\n", nm[i],
              100.0 * (double)h[i] / (double)tot, (double)h[i] / (double)h[9]);
  if (h[10])
    fprintf(stderr, "[exmc prof]   integrator wave: %llu units, %.1f cycles/unit compute, %.1f barrier wait, %.1f other\n",
            h[10], (double)h[11] / (double)h[10], (double)h[12] / (double)h[10], (double)h[13] / (double)h[10]);
  return hipSuccess;
}
#endif


thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(EXMC_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
  } while (0)

const uint64_t kZigKi[256] = EXMC_ZIG_KI_INIT;
const double kZigWi[256] = EXMC_ZIG_WI_INIT;
const double kZigFi[256] = EXMC_ZIG_FI_INIT;

double f32r(double x) { return (double)(float)x; }

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return EXMC_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    HIP_TRY(hipMalloc(&p, bytes));
    cap = bytes;
    return EXMC_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return (T*)p; }
};

// ---- launching the model-dependent kernels --------------------------------------------------
// libexmc_hip.so and one-unit plug-ins: the kernel is an instantiation of this translation unit,
// launched through its host stub. A plug-in built in parts with -DEXMC_PLUGIN_MODULES (the default
// of exmc_amd/codegen.py build_plugin since round 4): the model-dependent kernels are compiled
// DEVICE-ONLY, one code object per part, embedded in the library as data (exmc_blob_table, written
// by the build next to this unit) and loaded with hipModuleLoadData on first use; a launch looks the
// kernel up by its device-side mangled name and goes through hipModuleLaunchKernel. No host pass is
// spent on the parts and no host stub exists for these kernels -- SURVEY section 8 row f3's "IR ->
// kernel via a run-time compiled code object", with hipcc's device pass in the role of hiprtc.
#ifdef EXMC_PLUGIN_MODULES
extern "C" const unsigned char* const exmc_blob_table[];   // start_0, end_0, start_1, end_1, ..., null
struct ModSet {
  int device = -1;
  std::vector<hipModule_t> mods;
  std::unordered_map<std::string, hipFunction_t> fns;
};
std::mutex g_mod_mu;
std::vector<ModSet*> g_modsets;

// the function `name` of this plug-in's code objects on `device` (the current device of the caller)
hipError_t mod_function(int device, const char* name, hipFunction_t* out) {
  std::lock_guard<std::mutex> lock(g_mod_mu);
  ModSet* ms = nullptr;
  for (ModSet* x : g_modsets)
    if (x->device == device) ms = x;
  if (!ms) {
    ms = new ModSet;
    ms->device = device;
    for (int i = 0; exmc_blob_table[2 * i] != nullptr; i++) {
      hipModule_t mod = nullptr;
      const hipError_t e = hipModuleLoadData(&mod, exmc_blob_table[2 * i]);
      if (e != hipSuccess) {
        for (hipModule_t m_ : ms->mods) (void)hipModuleUnload(m_);
        delete ms;
        return e;
      }
      ms->mods.push_back(mod);
    }
    g_modsets.push_back(ms);
  }
  auto it = ms->fns.find(name);
  if (it != ms->fns.end()) {
    *out = it->second;
    return hipSuccess;
  }
  // (a kernel lives in one of the code objects: the lookups in the others fail, and the runtime
  // remembers the last failure -- it is not the caller's error, whichever way the search ends)
  hipFunction_t found = nullptr;
  for (hipModule_t mod : ms->mods) {
    hipFunction_t f = nullptr;
    if (hipModuleGetFunction(&f, mod, name) == hipSuccess && f) {
      found = f;
      break;
    }
  }
  (void)hipGetLastError();
  if (!found) return hipErrorInvalidDeviceFunction;
  ms->fns.emplace(name, found);
  *out = found;
  return hipSuccess;
}

template <class... A>
hipError_t mod_launch(int device, const char* name, dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                      A... args) {
  hipFunction_t f = nullptr;
  const hipError_t e = mod_function(device, name, &f);
  if (e != hipSuccess) return e;
  void* params[] = {(void*)&args...};
  return hipModuleLaunchKernel(f, grid.x, grid.y, grid.z, block.x, block.y, block.z, (unsigned)lds, stream,
                               params, nullptr);
}
// (the kernel arguments are passed by value exactly as the kernel declares them: every call site
// hands over the parameter structs themselves)
#define EXMC_KLAUNCH(dev, kernel, grid, block, lds, stream, ...)   HIP_TRY(mod_launch((dev), __builtin_get_device_side_mangled_name kernel, (grid), (block), (lds), (stream), __VA_ARGS__))
// dynamic LDS above 64 KB needs no opt-in for a module function on this runtime; a launch that asks
// for more than the device has fails in hipModuleLaunchKernel
#define EXMC_KMAXLDS(kernel, bytes)   do {                                } while (0)
#else
#define EXMC_KLAUNCH(dev, kernel, grid, block, lds, stream, ...)   hipLaunchKernelGGL(kernel, (grid), (block), (lds), (stream), __VA_ARGS__)
#define EXMC_KMAXLDS(kernel, bytes)   HIP_TRY(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)))
#endif

}  // namespace

struct exmc_hip_model {
  int kind = 0, d = 0, device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double last_ms = 0.0;
  EightSchoolsConsts es{};
  SimpleConsts sp{};
  SVConsts sv{};
  LogisticConsts lg{};
  RadonConsts rd{};
#ifdef EXMC_CUSTOM_HEADER
  CustomConsts cu{};
#endif
  DevBuf data;      // model data kept in HBM (logistic X,y; radon u,starts,floor,y)
  DevBuf zig;       // ki[256] u64, wi[256], fi[256]
  DevBuf flat;      // perm[D], rank[D] (int32) when the kernel order is not the reference's flat order
  bool flat_set = false;
  DevBuf tuning;    // inv_mass[D], sqrt_inv_mass[D]
  DevBuf state;     // q[D][C], g[D][C], logp[C], rng[2][C]
  DevBuf stack;
  DevBuf misc;      // [0] eps_out, [1..2] counters (u64), [3] trace scratch word, [8..8+D) init_q
  DevBuf trace;     // staging for host-trace entry points
  DevBuf io;        // staging for host vectors
  DevBuf scores;    // ess_bulk: the rank-normalised copy of the caller's trace
  DevBuf dense;     // opts[:dense_mass]: cov[D][D], chol[D][D] while a dense mass is in force
  // lane layouts (LaneDenseModel): the same two matrices with their columns permuted to kernel
  // dimensions, [D][GD] each, built from the host copies for the lane count of the launch; the
  // warmup kernel's global workspace; rank of each kernel dimension in the flat vector
  DevBuf densep, densews;
  DevBuf migboard;  // chain migration board of the sampling kernel (exmc_nuts.hpp)
  // push-style stream: page-locked host memory [progress word | trace of the run in flight]
  void* pin_host = nullptr;
  size_t pin_bytes = 0;
  std::atomic<bool> stream_in_flight{false};
  int densep_gd = 0;
  std::vector<double> h_dense;
  std::vector<int32_t> h_rank;
  DevBuf esswork;   // ESS: [count | EssTailItem...] of the series that go on to ess_tail_kernel
  bool dense_on = false;
  int state_chains = 0;
  // resident chains (exmc_hip_chains_init / _advance)
  int res_C = 0, res_lanes = 0, res_max_depth = 10;
  double res_eps = 0.0;
  int simds = 1024;   // 4 x the device's compute units (MI355X: 256 CUs)
};

namespace {

ChainState state_view(exmc_hip_model* m, int C) {
  ChainState s;
  double* b = m->state.as<double>();
  s.q = b;
  s.g = b + (size_t)m->d * C;
  s.logp = b + (size_t)2 * m->d * C;
  s.rng = (uint64_t*)(b + (size_t)2 * m->d * C + C);
  return s;
}
size_t state_bytes(int d, int C) { return ((size_t)2 * d * C + (size_t)3 * C) * 8; }

int ensure_state(exmc_hip_model* m, int C) {
  m->res_C = 0;  // whoever re-lays-out the state buffer evicts the resident chains
  int rc = m->state.ensure(state_bytes(m->d, C));
  if (rc) return rc;
  m->state_chains = C;
  return EXMC_OK;
}

int upload_tuning(exmc_hip_model* m, const double* inv_mass) {
  std::vector<double> h(2 * (size_t)m->d);
  for (int i = 0; i < m->d; i++) {
    h[i] = inv_mass[i];
    h[m->d + i] = std::sqrt(inv_mass[i]);  // :math.sqrt(inv_m), sampler.ex:397
  }
  int rc = m->tuning.ensure(h.size() * 8);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(m->tuning.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return EXMC_OK;
}

// ---- model/lanes dispatch ---------------------------------------------------------------
template <class M_, int G_, int LDSL_>
struct Tag {
  using M = M_;
  static constexpr int G = G_;
  static constexpr int LDSL = LDSL_;  // tree-stack levels kept in LDS by nuts_kernel
};

// LDS budget per one-wave workgroup: 4096 chains x G=16 lanes = 1024 workgroups = 4 per CU of
// 160 KB; sv at 2048 chains x 64 lanes = 8 per CU.
// kOneChain: the launches of the shared one-chain warmup (chain init, warmup kernels); a generated
// lane layout of fewer than 64 lanes per chain has a form of its own for them (lanes = 64: the
// chain's model terms over the whole wavefront, exmc_models.hpp CustomSplit)
template <bool kOneChain = false, class F>
int dispatch(exmc_hip_model* m, int lanes, F&& f) {
  switch (m->kind) {
#ifdef EXMC_CUSTOM_HEADER
    case EXMC_MODEL_CUSTOM:   // a generated model (exmc_amd/codegen.py)
#if defined(EXMC_GEN_LANES) && EXMC_GEN_LANES < 64
      if constexpr (kOneChain) {
        if (lanes == 64) return f(Tag<CustomSplit, EXMC_GEN_LANES, EXMC_GEN_LDSL>{}, m->cu);
      }
#endif
#ifdef EXMC_GEN_ONE_LANE
      if (lanes == 1) return f(Tag<Custom<1>, 1, EXMC_GEN_LDS_LEVELS>{}, m->cu);   // one lane per chain
#endif
#ifdef EXMC_GEN_VEC
      if (lanes == 16) return f(Tag<Custom<16>, 16, 6>{}, m->cu);   // plates across lanes
#endif
#ifdef EXMC_GEN_LANES
      if (lanes == EXMC_GEN_LANES)   // several dimensions per lane (codegen_lanes.py)
        return f(Tag<Custom<EXMC_GEN_LANES>, EXMC_GEN_LANES, EXMC_GEN_LDSL>{}, m->cu);
#endif
      break;
#endif
#ifndef EXMC_ONLY_CUSTOM      // plug-in builds carry the generated model only
// development builds for kernel work carry one model in one layout: -DEXMC_DEV_ES16_ONLY,
// -DEXMC_DEV_ONLY=EXMC_DEV_SV64 (or _RADON64, _LOGISTIC16)
#define EXMC_DEV_SV64 1
#define EXMC_DEV_RADON64 2
#define EXMC_DEV_LOGISTIC16 3
#if defined(EXMC_DEV_ES16_ONLY)
    case EXMC_MODEL_EIGHT_SCHOOLS:
      if (lanes == 16) return f(Tag<EightSchools<16>, 16, 6>{}, m->es);
      break;
#elif defined(EXMC_DEV_ONLY)
#if EXMC_DEV_ONLY == EXMC_DEV_SV64
    case EXMC_MODEL_SV:
      if (lanes == 64) return f(Tag<SV<64>, 64, 3>{}, m->sv);
      break;
#elif EXMC_DEV_ONLY == EXMC_DEV_RADON64
    case EXMC_MODEL_RADON:
      if (lanes == 64) return f(Tag<Radon<64>, 64, 3>{}, m->rd);
      break;
#elif EXMC_DEV_ONLY == EXMC_DEV_LOGISTIC16
    case EXMC_MODEL_LOGISTIC:
      if (lanes == 16) return f(Tag<Logistic<16>, 16, 2>{}, m->lg);
      if (lanes == 64) return f(Tag<Logistic<64>, 64, 2>{}, m->lg);
      break;
#endif
#else
    case EXMC_MODEL_EIGHT_SCHOOLS:
      switch (lanes) {
        case 1: return f(Tag<EightSchools<1>, 1, 2>{}, m->es);
        case 2: return f(Tag<EightSchools<2>, 2, 3>{}, m->es);
        case 4: return f(Tag<EightSchools<4>, 4, 4>{}, m->es);
        case 8: return f(Tag<EightSchools<8>, 8, 5>{}, m->es);
        case 16: return f(Tag<EightSchools<16>, 16, 6>{}, m->es);
        default: break;
      }
      break;
    case EXMC_MODEL_SIMPLE:
      if (lanes == 1) return f(Tag<Simple<1>, 1, 6>{}, m->sp);
      break;
    case EXMC_MODEL_SV:
      switch (lanes) {
        case 32: return f(Tag<SV<32>, 32, 2>{}, m->sv);
        case 64: return f(Tag<SV<64>, 64, 3>{}, m->sv);
        default: break;
      }
      break;
    case EXMC_MODEL_LOGISTIC:
      switch (lanes) {
        case 4: return f(Tag<Logistic<4>, 4, 2>{}, m->lg);   // matrix-core path
        case 8: return f(Tag<Logistic<8>, 8, 2>{}, m->lg);
        case 16: return f(Tag<Logistic<16>, 16, 2>{}, m->lg);
        case 64: return f(Tag<Logistic<64>, 64, 2>{}, m->lg);   // one chain per wave: the warmup's layout
        default: break;
      }
      break;
    case EXMC_MODEL_RADON:
      switch (lanes) {
        case 32: return f(Tag<Radon<32>, 32, 2>{}, m->rd);
        case 64: return f(Tag<Radon<64>, 64, 3>{}, m->rd);
        default: break;
      }
      break;
#endif
#endif
    default: break;
  }
  return fail(EXMC_ERR_UNSUPPORTED, "model kind / lanes_per_chain combination not compiled in");
}

// the layouts that carry a dense mass matrix: a whole chain in one lane, or eight_schools' row
// layout (16 lanes, one dimension per lane) through RowDenseModel
bool dense_layout_ok(const exmc_hip_model* m, int lanes) {
  return lanes == 1 || (m->kind == EXMC_MODEL_EIGHT_SCHOOLS && lanes == 16) ||
         (m->kind == EXMC_MODEL_SV && lanes == 64) || (m->kind == EXMC_MODEL_RADON && lanes == 64) ||
         (m->kind == EXMC_MODEL_LOGISTIC && lanes == 16);
}

// covp[s GD + i] = cov[rank(i)][s], cholp[j GD + i] = chol[j][rank(i)], zero columns past D
// (exmc_device.hpp LaneDense), for the GD = lanes x dimensions-per-lane of the launch
int ensure_densep(exmc_hip_model* m, int GD) {
  if (m->densep_gd == GD) return EXMC_OK;
  const int d = m->d;
  if (m->h_dense.size() != 2 * (size_t)d * d) return fail(EXMC_ERR_BADARG, "no dense mass is set");
  std::vector<double> h(2 * (size_t)d * GD, 0.0);
  const double* cov = m->h_dense.data();
  const double* chol = cov + (size_t)d * d;
  for (int s = 0; s < d; s++)
    for (int i = 0; i < d; i++) {
      const int r = m->h_rank.empty() ? i : m->h_rank[i];
      h[(size_t)s * GD + i] = cov[(size_t)r * d + s];
      h[(size_t)d * GD + (size_t)s * GD + i] = chol[(size_t)s * d + r];
    }
  int rc = m->densep.ensure(h.size() * 8);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(m->densep.p, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  m->densep_gd = GD;
  return EXMC_OK;
}

// dispatch for the launches whose mass-dependent operations depend on the dense mode
template <bool kOneChain = false, class F>
int dispatch_mass(exmc_hip_model* m, int lanes, bool dense, F&& f) {
#if !defined(EXMC_ONLY_CUSTOM) && !defined(EXMC_DEV_ONLY)
  if (dense && m->kind == EXMC_MODEL_EIGHT_SCHOOLS && lanes == 16)
    return f(Tag<RowDenseModel<EightSchools<16>>, 16, 6>{}, m->es);
  if (dense && m->kind == EXMC_MODEL_SV && lanes == 64)
    return f(Tag<LaneDenseModel<SV<64>, 64>, 64, 2>{}, m->sv);
  if (dense && m->kind == EXMC_MODEL_RADON && lanes == 64)
    return f(Tag<LaneDenseModel<Radon<64>, 64>, 64, 2>{}, m->rd);
  if (dense && m->kind == EXMC_MODEL_LOGISTIC && lanes == 16)
    return f(Tag<LaneDenseModel<Logistic<16>, 16>, 16, 2>{}, m->lg);
#endif
  return dispatch<kOneChain>(m, lanes, f);
}

int default_lanes(int kind) {
  switch (kind) {
    case EXMC_MODEL_EIGHT_SCHOOLS: return 16;
    case EXMC_MODEL_SIMPLE: return 1;
    case EXMC_MODEL_SV: return 64;
    case EXMC_MODEL_LOGISTIC: return 16;
    case EXMC_MODEL_RADON: return 64;
#if defined(EXMC_GEN_LANES)
    case EXMC_MODEL_CUSTOM: return EXMC_GEN_LANES;
#elif defined(EXMC_GEN_VEC)
    case EXMC_MODEL_CUSTOM: return 16;
#endif
    default: return 1;
  }
}

int resolve_lanes(const exmc_hip_model* m, int lanes) { return lanes > 0 ? lanes : default_lanes(m->kind); }

dim3 grid_for(int n_chains, int lanes, int block) {
  size_t threads = (size_t)n_chains * lanes;
  return dim3((unsigned)((threads + block - 1) / block));
}

constexpr int kBlock = 64;

FlatOrder flat_order(exmc_hip_model* m) {
  FlatOrder f;
  if (m->flat_set) {
    f.perm = m->flat.as<int32_t>();
    f.rank = m->flat.as<int32_t>() + m->d;
  }
  return f;
}

// perm[r] = kernel dimension of flat entry r; identity clears the table
int set_flat_order(exmc_hip_model* m, const int32_t* perm) {
  const int d = m->d;
  std::vector<int32_t> h(2 * (size_t)d, -1);
  bool identity = true;
  for (int r = 0; r < d; r++) {
    if (perm[r] < 0 || perm[r] >= d || h[d + perm[r]] >= 0)
      return fail(EXMC_ERR_BADARG, "flat order is not a permutation of 0..d-1");
    h[r] = perm[r];
    h[d + perm[r]] = r;
    identity = identity && perm[r] == r;
  }
  m->h_rank.assign(h.begin() + d, h.end());
  m->densep_gd = 0;
  if (identity) {
    m->flat_set = false;
    return EXMC_OK;
  }
  int rc = m->flat.ensure(h.size() * 4);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(m->flat.p, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  m->flat_set = true;
  return EXMC_OK;
}

// the kinds whose free-RV names the kind fixes: ids sorted as strings (point_map.ex:37)
int default_flat_order(exmc_hip_model* m) {
  std::vector<std::string> names;
  if (m->kind == EXMC_MODEL_SV) {          // kernel order s_1..s_T, sigma, nu
    for (int t = 1; t <= m->d - 2; t++) names.push_back("s_" + std::to_string(t));
    names.push_back("sigma");
    names.push_back("nu");
  } else if (m->kind == EXMC_MODEL_LOGISTIC) {   // kernel order alpha, beta_1..beta_K
    names.push_back("alpha");
    for (int j = 1; j < m->d; j++) names.push_back("beta_" + std::to_string(j));
  } else {
    return EXMC_OK;   // sorted already (eight_schools, simple, generated models) or caller-defined (radon)
  }
  std::vector<int32_t> perm(m->d);
  for (int i = 0; i < m->d; i++) perm[i] = i;
  std::sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return names[a] < names[b]; });
  return set_flat_order(m, perm.data());
}

const uint64_t* zig_ki(exmc_hip_model* m) { return m->zig.as<uint64_t>(); }
const double* zig_wi(exmc_hip_model* m) { return m->zig.as<double>() + 256; }
const double* zig_fi(exmc_hip_model* m) { return m->zig.as<double>() + 512; }

// one_chain: the start of the shared warmup (a generated lane layout may have its one-chain form)
int launch_init(exmc_hip_model* m, int lanes, int C, int chain_lo, uint64_t seed,
                const double* init_q_host, bool one_chain = false) {
  const double* init_dev = nullptr;
  if (init_q_host) {
    double* dst = m->misc.as<double>() + 8;
    HIP_TRY(hipMemcpyAsync(dst, init_q_host, (size_t)m->d * 8, hipMemcpyHostToDevice, m->stream));
    init_dev = dst;
  }
  InitParams P;
  P.st = state_view(m, C);
  P.n_chains = C;
  P.chain_lo = chain_lo;
  P.base_seed = seed;
  P.init_q = init_dev;
  P.zig_ki = zig_ki(m); P.zig_wi = zig_wi(m); P.zig_fi = zig_fi(m);
  P.nor_r = EXMC_NOR_R;
  P.flat = flat_order(m);
  auto launch = [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    const size_t xlds = aux_lds_bytes<typename T::M>();
    EXMC_KLAUNCH(m->device, (init_chains_kernel<typename T::M, T::G>), grid_for(C, T::G, kBlock),
                 dim3(kBlock), xlds, m->stream, P, mc);
    HIP_TRY(hipGetLastError());
    return (int)EXMC_OK;
  };
  return one_chain ? dispatch<true>(m, lanes, launch) : dispatch(m, lanes, launch);
}

// the kernels a push-style stream may run: each kind in its default layout
template <class M> inline constexpr bool kStreamKernel = false;
#if !defined(EXMC_ONLY_CUSTOM) && !defined(EXMC_DEV_ONLY)
template <> inline constexpr bool kStreamKernel<EightSchools<16>> = true;
template <> inline constexpr bool kStreamKernel<Simple<1>> = true;
template <> inline constexpr bool kStreamKernel<SV<64>> = true;
template <> inline constexpr bool kStreamKernel<Logistic<16>> = true;
template <> inline constexpr bool kStreamKernel<Radon<64>> = true;
#endif
#ifdef EXMC_CUSTOM_HEADER
#ifdef EXMC_GEN_ONE_LANE
template <> inline constexpr bool kStreamKernel<Custom<1>> = true;
#endif
#ifdef EXMC_GEN_VEC
template <> inline constexpr bool kStreamKernel<Custom<16>> = true;
#endif
#ifdef EXMC_GEN_LANES
template <> inline constexpr bool kStreamKernel<Custom<EXMC_GEN_LANES>> = true;
#endif
#endif

int launch_nuts(exmc_hip_model* m, int lanes, int C, int n_draws, int draw_offset, double eps,
                int max_depth, TraceDev tr, bool timed, int* progress = nullptr) {
  if (max_depth < 1 || max_depth > kMaxLevels) return fail(EXMC_ERR_BADARG, "max_tree_depth out of range");
  return dispatch_mass(m, lanes, m->dense_on, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    using M = typename T::M;
    dim3 grid = grid_for(C, T::G, kNutsBlock);
    size_t nthreads = (size_t)grid.x * kNutsBlock;
    constexpr int kSpill = (kMaxLevels > T::LDSL) ? (kMaxLevels - T::LDSL) : 1;
    int rc = m->stack.ensure((size_t)kSpill * nuts_nslot<M>() * nthreads * 8);
    if (rc) return rc;
    NutsParams P;
    P.st = state_view(m, C);
    P.n_chains = C;
    P.n_draws = n_draws;
    P.draw_offset = draw_offset;
    P.eps = eps;
    P.max_depth = max_depth;
    P.inv_mass = m->tuning.as<double>();
    P.sqrt_inv_mass = m->tuning.as<double>() + m->d;
    P.tr = tr;
    P.stack = m->stack.as<double>();
    P.counters = (unsigned long long*)(m->misc.as<double>() + 1);
    P.scratch = m->misc.as<double>() + 3;
    P.zig_ki = zig_ki(m); P.zig_wi = zig_wi(m); P.zig_fi = zig_fi(m);
    P.nor_r = EXMC_NOR_R;
    P.flat = flat_order(m);
    P.progress = progress;
    P.mig = nullptr;
    {
      const char* pe = std::getenv("EXMC_HIP_PRIO");   // 0: leave the arbiter's oldest-first order alone
      P.prio = (pe && pe[0] == '0') ? 0 : 1;
    }
    P.simds = m->simds;
    if (progress) {
      if constexpr (kStreamKernel<M>) {
        if (timed) HIP_TRY(hipEventRecord(m->ev0, m->stream));
        const size_t lds_s = nuts_lds_bytes<M, T::LDSL, M::kNutsZigInLds>();
        EXMC_KLAUNCH(m->device, (nuts_kernel<M, T::G, T::LDSL, false, true>), grid, dim3(kNutsBlock), lds_s,
                     m->stream, P, mc);
        HIP_TRY(hipGetLastError());
        if (timed) HIP_TRY(hipEventRecord(m->ev1, m->stream));
        return (int)EXMC_OK;
      } else {
        return fail(EXMC_ERR_UNSUPPORTED, "a push-style stream runs in the model kind's default layout, diagonal mass");
      }
    }
    if constexpr (M::kMigrate) {
      // worth it when the launch puts two chains on a SIMD (more waves than the 1024 SIMDs) and
      // runs long enough to have a tail; EXMC_HIP_MIGRATE=0 / 1 forces it off / on
      const char* me = std::getenv("EXMC_HIP_MIGRATE");
      const bool on = me ? (me[0] == '1') : ((int)grid.x > (m->simds > 0 ? m->simds : 1024) && n_draws >= 100);
      if (on) {
        const size_t nb = mig_board_ints(grid.x) * sizeof(int);
        rc = m->migboard.ensure(nb);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(m->migboard.p, 0, nb, m->stream));
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)m->migboard.p, C, 1, m->stream));   // board[0] = chains left
        P.mig = m->migboard.as<int>();
      }
    }
    if (m->dense_on) {
      if (T::G != 1 && !M::kRowDense && !M::kLaneDense)
        return fail(EXMC_ERR_UNSUPPORTED, "no dense mass matrix for this model at this lanes_per_chain");
      P.dm.cov = m->dense.as<double>();
      P.dm.chol = m->dense.as<double>() + (size_t)m->d * m->d;
      if constexpr (M::kLaneDense) {
        constexpr int GD = T::G * M::DPL;
        rc = ensure_densep(m, GD);
        if (rc) return rc;
        P.dm.covp = m->densep.as<double>();
        P.dm.cholp = m->densep.as<double>() + (size_t)m->d * GD;
      }
    }
    if constexpr (M::kPipeNutsLevels > 0) {
      // wave pairs (tree + integrator), fewer stack levels in LDS to make room for the mailbox.
      // Opt-in (EXMC_HIP_NUTS_PIPE=1): bit-identical, but the two waves alternate more than they overlap
      // (a doubling starts from the decision of the previous one), DESIGN section 5 --
      // eight_schools 4096 x 1000: 15.3 ms against 14.6 ms for one wave per SIMD.
      const char* pe = std::getenv("EXMC_HIP_NUTS_PIPE");
      if (pe && pe[0] == '1') {
        constexpr int PL = M::kPipeNutsLevels;
        constexpr int kSpillP = (kMaxLevels > PL) ? (kMaxLevels - PL) : 1;
        rc = m->stack.ensure((size_t)kSpillP * nuts_nslot<M>() * nthreads * 8);
        if (rc) return rc;
        P.stack = m->stack.as<double>();
        if (timed) HIP_TRY(hipEventRecord(m->ev0, m->stream));
        const size_t lds_p = nuts_lds_bytes<M, PL>() + pipe_lds_doubles<M::DPL>() * 8;
        EXMC_KLAUNCH(m->device, (nuts_kernel<M, T::G, PL, true>), grid, dim3(2 * kNutsBlock), lds_p,
                     m->stream, P, mc);
        HIP_TRY(hipGetLastError());
        if (timed) HIP_TRY(hipEventRecord(m->ev1, m->stream));
        return (int)EXMC_OK;
      }
    }
    if constexpr (M::kWgWaves > 0 && !M::kLaneDense && !M::kRowDense) {
      // Workgroups of kWgWaves wavefronts around one LDS image of the model's data (exmc_nuts.hpp
      // nuts_kernel_wg). Worth it as soon as some SIMD has to hold two wavefronts anyway (more waves
      // than SIMDs); below that every wavefront has a SIMD to itself in the one-wave form and a
      // workgroup per compute unit would leave units idle. EXMC_HIP_NUTS_WG=0 / 1 forces either form.
      constexpr int W = M::kWgWaves, WL = M::kWgLdsLevels;
      const char* we = std::getenv("EXMC_HIP_NUTS_WG");
      const size_t waves = grid.x;
      const bool on = we ? (we[0] == '1') : ((int)waves > (m->simds > 0 ? m->simds : 1024));
      if (on && !m->dense_on && M::wg_ok(mc)) {
        const dim3 wgrid((unsigned)((waves + W - 1) / W));
        const size_t wthreads = (size_t)wgrid.x * W * kNutsBlock;
        constexpr int kSpillW = (kMaxLevels > WL) ? (kMaxLevels - WL) : 1;
        rc = m->stack.ensure((size_t)kSpillW * nuts_nslot<M>() * wthreads * 8);
        if (rc) return rc;
        P.stack = m->stack.as<double>();
        const size_t lds_w = nuts_wg_lds_bytes<M, WL, W>();
        static_assert(nuts_wg_lds_bytes<M, WL, W>() <= 160 * 1024, "one workgroup per compute unit");
        EXMC_KMAXLDS((nuts_kernel_wg<M, T::G, WL, W>), lds_w);
        if (timed) HIP_TRY(hipEventRecord(m->ev0, m->stream));
        EXMC_KLAUNCH(m->device, (nuts_kernel_wg<M, T::G, WL, W>), wgrid, dim3(W * kNutsBlock), lds_w,
                     m->stream, P, mc);
        HIP_TRY(hipGetLastError());
        if (timed) HIP_TRY(hipEventRecord(m->ev1, m->stream));
        return (int)EXMC_OK;
      }
    }
    if (timed) HIP_TRY(hipEventRecord(m->ev0, m->stream));
    const size_t lds_bytes = nuts_lds_bytes<M, T::LDSL, M::kNutsZigInLds>();
    if (lds_bytes > 64 * 1024)   // the lane layouts' dense mass keeps M^-1 in LDS
      EXMC_KMAXLDS((nuts_kernel<M, T::G, T::LDSL>), lds_bytes);
    EXMC_KLAUNCH(m->device, (nuts_kernel<M, T::G, T::LDSL>), grid, dim3(kNutsBlock), lds_bytes,
                 m->stream, P, mc);
    HIP_TRY(hipGetLastError());
    if (timed) HIP_TRY(hipEventRecord(m->ev1, m->stream));
    if (P.mig && std::getenv("EXMC_HIP_MIGRATE_STATS")) {
      int h[4];
      HIP_TRY(hipMemcpyAsync(h, P.mig, sizeof(h), hipMemcpyDeviceToHost, m->stream));
      HIP_TRY(hipStreamSynchronize(m->stream));
      fprintf(stderr, "[exmc migrate] chains left %d, moved %d, hosts %d\n", h[0], h[2], h[3]);
    }
    return (int)EXMC_OK;
  });
}

int reset_counters(exmc_hip_model* m) {
  HIP_TRY(hipMemsetAsync(m->misc.as<double>() + 1, 0, 16, m->stream));
  return EXMC_OK;
}
int read_counters(exmc_hip_model* m, int64_t* lf, int32_t* div) {
  unsigned long long h[2];
  HIP_TRY(hipMemcpyAsync(h, m->misc.as<double>() + 1, 16, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (lf) *lf = (int64_t)h[0];
  if (div) *div = (int32_t)h[1];
  return EXMC_OK;
}

int finish_timing(exmc_hip_model* m) {
  HIP_TRY(hipEventSynchronize(m->ev1));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  m->last_ms = ms;
#ifdef EXMC_XCC_PROBE
  if (const char* path = std::getenv("EXMC_WAVE_PROBE_OUT")) {
    static std::vector<double> h(4096 * 5);
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_wave_probe), h.size() * 8));
    if (FILE* f = std::fopen(path, "w")) {   // rewritten after every timed launch: the last one stays
      std::fprintf(f, "%.3f\n", ms);
      for (int i = 0; i < 4096; i++)   // workgroup, placement, shader clocks, leapfrogs, wall start, wall end
        std::fprintf(f, "%d %.0f %.0f %.0f %.0f %.0f\n", i, h[i * 5], h[i * 5 + 1], h[i * 5 + 2], h[i * 5 + 3], h[i * 5 + 4]);
      std::fclose(f);
    }
  }
#endif
#if EXMC_PROFILE_SECTIONS
  HIP_TRY(print_sections("nuts", ms));
#endif
  return EXMC_OK;
}

// [C][n][d] host <- [n][d][C] device-layout host copy
void transpose_trace_vec(const double* src, double* dst, int n, int d, int C) {
  for (int s = 0; s < n; s++)
    for (int i = 0; i < d; i++)
      for (int c = 0; c < C; c++) dst[((size_t)c * n + s) * d + i] = src[((size_t)s * d + i) * C + c];
}
template <class T>
void transpose_trace_scalar(const T* src, T* dst, int n, int C) {
  for (int s = 0; s < n; s++)
    for (int c = 0; c < C; c++) dst[(size_t)c * n + s] = src[(size_t)s * C + c];
}

struct TraceLayout {
  size_t off_draws, off_logp, off_depth, off_nsteps, off_div, off_acc, off_energy, total;
};
TraceLayout trace_layout(int S, int d, int C) {
  TraceLayout L;
  size_t o = 0;
  L.off_draws = o; o += (size_t)S * d * C * 8;
  L.off_logp = o; o += (size_t)S * C * 8;
  L.off_acc = o; o += (size_t)S * C * 8;
  L.off_energy = o; o += (size_t)S * C * 8;
  L.off_depth = o; o += (size_t)S * C * 4;
  L.off_nsteps = o; o += (size_t)S * C * 4;
  L.off_div = o; o += (size_t)S * C * 4;
  L.total = (o + 7) & ~(size_t)7;
  return L;
}
TraceDev trace_view(void* base, const TraceLayout& L) {
  char* b = (char*)base;
  TraceDev t;
  t.draws = (double*)(b + L.off_draws);
  t.logp = (double*)(b + L.off_logp);
  t.accept_prob = (double*)(b + L.off_acc);
  t.energy = (double*)(b + L.off_energy);
  t.tree_depth = (int32_t*)(b + L.off_depth);
  t.n_steps = (int32_t*)(b + L.off_nsteps);
  t.divergent = (int32_t*)(b + L.off_div);
  return t;
}

// copy a device-layout staging trace to the caller's host trace ([chain][draw][dim])
int download_trace(exmc_hip_model* m, const TraceLayout& L, int S, int C, exmc_hip_trace out) {
  std::vector<char> h(L.total);
  HIP_TRY(hipMemcpyAsync(h.data(), m->trace.p, L.total, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  TraceDev t = trace_view(h.data(), L);
  if (out.draws) transpose_trace_vec(t.draws, out.draws, S, m->d, C);
  if (out.logp) transpose_trace_scalar(t.logp, out.logp, S, C);
  if (out.accept_prob) transpose_trace_scalar(t.accept_prob, out.accept_prob, S, C);
  if (out.energy) transpose_trace_scalar(t.energy, out.energy, S, C);
  if (out.tree_depth) transpose_trace_scalar(t.tree_depth, out.tree_depth, S, C);
  if (out.n_steps) transpose_trace_scalar(t.n_steps, out.n_steps, S, C);
  if (out.divergent) transpose_trace_scalar(t.divergent, out.divergent, S, C);
  return EXMC_OK;
}

// ---- host-side adaptation: step_size.ex:13-50, mass_matrix.ex:40-97, sampler.ex:764-785 ----
struct DualAvg {
  double log_epsilon, log_epsilon_bar, h_bar, mu;
  int m;
  double gamma, t0, kappa, target;
  // exp/log through the numeric contract (exmc_detmath.h), m^-kappa as exp(-kappa*log(m)):
  // the same arithmetic warmup_kernel performs on the device.
  void init(double epsilon, double target_accept) {
    log_epsilon = exmc_log(epsilon);
    log_epsilon_bar = exmc_log(epsilon);
    h_bar = 0.0;
    mu = exmc_log(10.0 * epsilon);
    m = 0;
    gamma = 0.05; t0 = 10.0; kappa = 0.75;
    target = target_accept;
  }
  void update(double accept_stat) {
    const int mm = m + 1;
    const double eta = 1.0 / (mm + t0);
    const double hb = (1.0 - eta) * h_bar + eta * (target - accept_stat);
    const double le = mu - std::sqrt((double)mm) / gamma * hb;
    const double mk = exmc_exp(-kappa * exmc_log((double)mm));
    const double leb = mk * le + (1.0 - mk) * log_epsilon_bar;
    m = mm; h_bar = hb; log_epsilon = le; log_epsilon_bar = leb;
  }
  double current() const { return exmc_exp(log_epsilon); }
  double finalize() const { return exmc_exp(log_epsilon_bar); }
};

struct WelfordDiag {
  int n = 0, d = 0;
  std::vector<double> mean, m2;
  void init(int dim) { n = 0; d = dim; mean.assign(dim, 0.0); m2.assign(dim, 0.0); }
  void update(const double* q) {
    const int nn = n + 1;
    for (int i = 0; i < d; i++) {
      const double delta = q[i] - mean[i];
      const double nm = mean[i] + delta / ((double)nn * 1.0);
      const double d2 = q[i] - nm;
      m2[i] = m2[i] + delta * d2;
      mean[i] = nm;
    }
    n = nn;
  }
  void finalize(double* inv_mass) const {
    if (n < 3) {
      for (int i = 0; i < d; i++) inv_mass[i] = 1.0;
      return;
    }
    const double alpha = 5.0 / (n + 5.0);
    for (int i = 0; i < d; i++) {
      double var = m2[i] / ((double)(n - 1) * 1.0);
      var = std::fmax(var, 1.0e-6);
      inv_mass[i] = (1.0 - alpha) * var + alpha * 1.0e-3;
    }
  }
};

std::vector<std::pair<int, int>> build_windows(int from, int to, int base) {
  std::vector<std::pair<int, int>> w;
  if (to - from <= 0) return w;
  int cur = from;
  double size = base;
  while (cur < to) {
    const int remaining = to - cur;
    const int actual = ((double)remaining <= size * 1.5) ? remaining : (int)size;
    w.push_back({cur, cur + actual});
    cur += actual;
    size *= 2;
  }
  return w;
}

// One warmup transition of chain 0 on the GPU; returns accept stat, divergence flag, new q.
struct WarmupStep {
  double accept;
  int divergent;
};

int warmup_transition(exmc_hip_model* m, int lanes, double eps, int max_depth, std::vector<double>& qhost,
                      WarmupStep* out) {
  // mini trace: draws [d], accept [1], divergent (int32) in the next 8 bytes
  const int d = m->d;
  int rc = m->trace.ensure((size_t)(d + 2) * 8);
  if (rc) return rc;
  TraceDev tr{};
  tr.draws = m->trace.as<double>();
  tr.accept_prob = m->trace.as<double>() + d;
  tr.divergent = (int32_t*)(m->trace.as<double>() + d + 1);
  rc = launch_nuts(m, lanes, 1, 1, 0, eps, max_depth, tr, false);
  if (rc) return rc;
  std::vector<double> h(d + 2);
  HIP_TRY(hipMemcpyAsync(h.data(), m->trace.p, (size_t)(d + 2) * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  qhost.assign(h.begin(), h.begin() + d);
  out->accept = h[d];
  int32_t dv;
  std::memcpy(&dv, &h[d + 1], 4);
  out->divergent = dv;
  return EXMC_OK;
}

int find_eps(exmc_hip_model* m, int lanes, double* eps) {
  FindEpsParams P;
  P.st = state_view(m, 1);
  P.n_chains = 1;
  P.inv_mass = m->tuning.as<double>();
  P.sqrt_inv_mass = m->tuning.as<double>() + m->d;
  P.log_half = std::log(0.5);
  P.eps_out = m->misc.as<double>();
  P.zig_ki = zig_ki(m); P.zig_wi = zig_wi(m); P.zig_fi = zig_fi(m);
  P.nor_r = EXMC_NOR_R;
  P.flat = flat_order(m);
  int rc = dispatch(m, lanes, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    const size_t lds_bytes = nuts_lds_bytes<typename T::M, 0>();
    EXMC_KLAUNCH(m->device, (find_eps_kernel<typename T::M, T::G>), dim3(1), dim3(kNutsBlock), lds_bytes,
                 m->stream, P, mc);
    HIP_TRY(hipGetLastError());
    return (int)EXMC_OK;
  });
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(eps, m->misc.p, 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  return EXMC_OK;
}

// run_warmup (sampler.ex:537-762) on the single-chain state already initialised in m->state
int run_warmup(exmc_hip_model* m, int lanes, exmc_hip_opts o, exmc_hip_tuning* tun,
               const exmc_hip_tuning* start = nullptr) {
  const int d = m->d, W = o.num_warmup;
  std::vector<double> im(d, 1.0), qh(d);
  if (start) im.assign(start->inv_mass, start->inv_mass + d);
  int rc = upload_tuning(m, im.data());
  if (rc) return rc;
  double eps = 1.0;
  if (start) eps = start->epsilon;   // warm start: no initial search (sampler.ex:180-194)
  else rc = find_eps(m, lanes, &eps);
  if (rc) return rc;
  int divergences = 0;
  auto finish = [&](double e) {
    tun->epsilon = e;
    for (int i = 0; i < d; i++) tun->inv_mass[i] = im[i];
    tun->warmup_divergences = divergences;
    return (int)EXMC_OK;
  };
  if (W == 0) return finish(eps);
  const int init_buffer = (75 < W / 3) ? 75 : W / 3;
  const int adapt_end = W - 50;
  DualAvg da;
  da.init(eps, o.target_accept);
  WarmupStep ws;
  for (int i = 0; i < init_buffer; i++) {
    rc = warmup_transition(m, lanes, da.current(), o.max_tree_depth, qh, &ws);
    if (rc) return rc;
    divergences += ws.divergent;
    da.update(ws.accept);
  }
  eps = da.current();
  if (adapt_end <= init_buffer) return finish(da.finalize());
  for (auto win : build_windows(init_buffer, adapt_end, 25)) {
    WelfordDiag wf;
    wf.init(d);
    da.init(eps, o.target_accept);
    for (int i = win.first; i < win.second; i++) {
      const int cap = (i < 200) ? (o.max_tree_depth < 8 ? o.max_tree_depth : 8) : o.max_tree_depth;
      rc = warmup_transition(m, lanes, da.current(), cap, qh, &ws);
      if (rc) return rc;
      divergences += ws.divergent;
      da.update(ws.accept);
      if (!ws.divergent) wf.update(qh.data());
    }
    wf.finalize(im.data());
    rc = upload_tuning(m, im.data());
    if (rc) return rc;
    rc = find_eps(m, lanes, &eps);
    if (rc) return rc;
  }
  da.init(eps, o.target_accept);
  for (int i = adapt_end; i < W; i++) {
    rc = warmup_transition(m, lanes, da.current(), o.max_tree_depth, qh, &ws);
    if (rc) return rc;
    divergences += ws.divergent;
    da.update(ws.accept);
  }
  return finish(da.finalize());
}

// run_warmup (sampler.ex:537-762) in one launch: warmup_kernel keeps dual averaging, Welford and
// the step-size searches on the device; the host only lays out the window schedule.
int run_warmup_device(exmc_hip_model* m, int lanes, exmc_hip_opts o, exmc_hip_tuning* tun,
                      const exmc_hip_tuning* start = nullptr, double* cov_out = nullptr,
                      double* chol_out = nullptr) {
  const bool dense = cov_out != nullptr;
  if (o.max_tree_depth < 1 || o.max_tree_depth > kMaxLevels)
    return fail(EXMC_ERR_BADARG, "max_tree_depth out of range");
  const int d = m->d, W = o.num_warmup;
  WarmupParams P;
  P.st = state_view(m, 1);
  P.num_warmup = W;
  P.max_depth = o.max_tree_depth;
  P.target_accept = o.target_accept;
  P.log_half = std::log(0.5);
  P.init_buffer = (75 < W / 3) ? 75 : W / 3;
  P.adapt_end = W - 50;
  // dense windows are max(25, 10 d) long at the start (sampler.ex:682)
  auto wins = build_windows(P.init_buffer, P.adapt_end, dense ? std::max(25, 10 * d) : 25);
  if (wins.size() > 32) return fail(EXMC_ERR_BADARG, "num_warmup needs more than 32 windows");
  P.dense = dense ? 1 : 0;
  P.dense_ws = nullptr;
  P.n_windows = (int)wins.size();
  for (int k = 0; k < 32; k++) {
    P.win_start[k] = k < P.n_windows ? wins[k].first : -1;
    P.win_end[k] = k < P.n_windows ? wins[k].second : -1;
  }
  P.zig_ki = zig_ki(m); P.zig_wi = zig_wi(m); P.zig_fi = zig_fi(m);
  P.nor_r = EXMC_NOR_R;
  P.flat = flat_order(m);
  P.eps0 = 0.0;
  P.inv_mass0 = P.sqrt_inv_mass0 = nullptr;
  if (start) {   // warm start: previous inverse mass and step size, no initial search
    int rcs = upload_tuning(m, start->inv_mass);
    if (rcs) return rcs;
    P.eps0 = start->epsilon;
    P.inv_mass0 = m->tuning.as<double>();
    P.sqrt_inv_mass0 = m->tuning.as<double>() + d;
  }
  const size_t n_out = (size_t)(8 + d) + (dense ? 2 * (size_t)d * d : 0);
  int rc = m->io.ensure(n_out * 8);
  if (rc) return rc;
  P.out = m->io.as<double>();
  rc = dispatch_mass<true>(m, lanes, dense, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    using M = typename T::M;
    constexpr int kSpill = (kMaxLevels > T::LDSL) ? (kMaxLevels - T::LDSL) : 1;
    // Replicas: the chain is deterministic and the chip is otherwise idle, so the same warmup runs
    // in `reps` workgroups at once and the first one to finish publishes the result. The time of
    // a one-workgroup kernel depends on the CU it lands on (16.6 to 24.6 ms for the same
    // eight_schools launch over the CUs of one XCD); the race takes the fastest CU of the draw.
    const char* re = std::getenv("EXMC_HIP_WARMUP_REPLICAS");
    int reps = re ? std::atoi(re) : 32;
    reps = reps < 1 ? 1 : (reps > 256 ? 256 : reps);
    if (dense && T::G != 1 && !M::kRowDense && !M::kLaneDense)
      return fail(EXMC_ERR_UNSUPPORTED, "no dense mass matrix for this model at this lanes_per_chain");
    if constexpr (M::kLaneDense) {
      reps = 1;   // one global workspace
      int r3 = m->densews.ensure(LaneDenseWs<T::G, M::DPL, M::D>::doubles() * 8);
      if (r3) return r3;
      P.dense_ws = m->densews.as<double>();
    }
    P.stack_stride = (size_t)kSpill * nuts_nslot<M>() * kNutsBlock;
    int r2 = m->stack.ensure(P.stack_stride * 8 * (size_t)reps);
    if (r2) return r2;
    P.stack = m->stack.as<double>();
    P.race = nullptr;
    if (reps > 1) {
      P.race = (int*)(m->misc.as<double>() + 4);
      HIP_TRY(hipMemsetAsync(P.race, 0, 8, m->stream));
    }
    size_t lds_bytes = nuts_lds_bytes<M, T::LDSL>();
    P.stage_model = 0;
    if (M::kStageDoubles > 0 && lds_bytes + (size_t)M::kStageDoubles * 8 <= 160 * 1024 &&
        (m->kind != EXMC_MODEL_LOGISTIC || m->lg.Npad <= 512)) {
      P.stage_model = 1;
      lds_bytes += (size_t)M::kStageDoubles * 8;
    }
    // two-wave form (tree wave + integrator wave, exmc_nuts.hpp PipeBox) where the model gains
    // from it; EXMC_HIP_WARMUP_PIPE=0 / 1 forces the one-wave / two-wave kernel
    const char* pe = std::getenv("EXMC_HIP_WARMUP_PIPE");
    const bool pipe = !dense && ((pe && pe[0] == '1') || (M::kPipeWarmup && !(pe && pe[0] == '0'))) &&
                      lds_bytes + pipe_lds_doubles<M::DPL>() * 8 <= 160 * 1024;
    if (dense && !M::kLaneDense) lds_bytes += 3 * (size_t)d * d * 8;   // m2, cov, chol behind everything else
    if (lds_bytes > 160 * 1024) return fail(EXMC_ERR_UNSUPPORTED, "dense warmup state does not fit in LDS");
    if constexpr (!M::kLaneDense && !M::kRowDense && M::kHasPipeWarmup) {   // the dense variants have no two-wave form
      if (pipe) {
        lds_bytes += pipe_lds_doubles<M::DPL>() * 8;
        if (lds_bytes > 64 * 1024)
          EXMC_KMAXLDS((warmup_kernel<M, T::G, T::LDSL, true>), lds_bytes);
        HIP_TRY(hipEventRecord(m->ev0, m->stream));
        EXMC_KLAUNCH(m->device, (warmup_kernel<M, T::G, T::LDSL, true>), dim3(reps), dim3(2 * kNutsBlock),
                     lds_bytes, m->stream, P, mc);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(m->ev1, m->stream));
        return (int)EXMC_OK;
      }
    }
    if (lds_bytes > 64 * 1024)
      EXMC_KMAXLDS((warmup_kernel<M, T::G, T::LDSL>), lds_bytes);
    HIP_TRY(hipEventRecord(m->ev0, m->stream));
    EXMC_KLAUNCH(m->device, (warmup_kernel<M, T::G, T::LDSL>), dim3(reps), dim3(kNutsBlock), lds_bytes,
                 m->stream, P, mc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev1, m->stream));
    return (int)EXMC_OK;
  });
  if (rc) return rc;
  std::vector<double> h(n_out);
  HIP_TRY(hipMemcpyAsync(h.data(), m->io.p, h.size() * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  rc = finish_timing(m);
  if (rc) return rc;
#if EXMC_PROFILE_SECTIONS
  HIP_TRY(print_sections("warmup", m->last_ms));
  fprintf(stderr, "[exmc prof]   warmup leapfrogs %.0f\n", h[2]);
#endif
#ifdef EXMC_XCC_PROBE
  fprintf(stderr, "[xcc probe] warmup ran on xcc/cu/se %d, %.3f ms  shader clocks %.0f  wall ticks %.0f\n",
          (int)h[2], m->last_ms, h[3 + d], h[4 + d]);
#endif
  tun->epsilon = h[0];
  tun->warmup_divergences = (int)h[1];
  for (int i = 0; i < d; i++) tun->inv_mass[i] = h[3 + i];
  if (dense) {
    if (!(h[0] > 0.0)) return fail(EXMC_ERR_BADARG, "dense warmup: a window covariance was not positive definite");
    std::memcpy(cov_out, h.data() + 3 + d, (size_t)d * d * 8);
    std::memcpy(chol_out, h.data() + 3 + d + (size_t)d * d, (size_t)d * d * 8);
  }
  return EXMC_OK;
}

// sample_chains(..., vectorized: false): the kinds' default layouts carry the independent-adaptation
// kernel (exmc_nuts.hpp indep_kernel), like the stream form
template <class M> inline constexpr bool kIndepKernel = kStreamKernel<M>;
#if defined(EXMC_DEV_ES16_ONLY)
template <> inline constexpr bool kIndepKernel<EightSchools<16>> = true;
#elif defined(EXMC_DEV_ONLY)
#if EXMC_DEV_ONLY == EXMC_DEV_SV64
template <> inline constexpr bool kIndepKernel<SV<64>> = true;
#elif EXMC_DEV_ONLY == EXMC_DEV_RADON64
template <> inline constexpr bool kIndepKernel<Radon<64>> = true;
#elif EXMC_DEV_ONLY == EXMC_DEV_LOGISTIC16
template <> inline constexpr bool kIndepKernel<Logistic<16>> = true;
#endif
#endif

// one launch: every chain of [chain_lo, chain_hi) adapts and samples on its own (indep_kernel)
int launch_independent(exmc_hip_model* m, int lanes, int C, exmc_hip_opts o, TraceDev tr, double* tune_dev) {
  if (o.max_tree_depth < 1 || o.max_tree_depth > kMaxLevels)
    return fail(EXMC_ERR_BADARG, "max_tree_depth out of range");
  const int W = o.num_warmup;
  IndepParams P;
  P.st = state_view(m, C);
  P.n_chains = C;
  P.num_warmup = W;
  P.num_samples = o.num_samples;
  P.max_depth = o.max_tree_depth;
  P.target_accept = o.target_accept;
  P.log_half = std::log(0.5);
  P.init_buffer = (75 < W / 3) ? 75 : W / 3;
  P.adapt_end = W - 50;
  auto wins = build_windows(P.init_buffer, P.adapt_end, 25);
  if (wins.size() > 32) return fail(EXMC_ERR_BADARG, "num_warmup needs more than 32 windows");
  P.n_windows = (int)wins.size();
  for (int k = 0; k < 32; k++) {
    P.win_start[k] = k < P.n_windows ? wins[k].first : -1;
    P.win_end[k] = k < P.n_windows ? wins[k].second : -1;
  }
  P.tr = tr;
  P.tune_out = tune_dev;
  P.counters = (unsigned long long*)(m->misc.as<double>() + 1);
  P.zig_ki = zig_ki(m); P.zig_wi = zig_wi(m); P.zig_fi = zig_fi(m);
  P.nor_r = EXMC_NOR_R;
  P.flat = flat_order(m);
  return dispatch(m, lanes, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    using M = typename T::M;
    if constexpr (kIndepKernel<M>) {
      dim3 grid = grid_for(C, T::G, kNutsBlock);
      const size_t nthreads = (size_t)grid.x * kNutsBlock;
      constexpr int kSpill = (kMaxLevels > T::LDSL) ? (kMaxLevels - T::LDSL) : 1;
      int rc = m->stack.ensure((size_t)kSpill * nuts_nslot<M>() * nthreads * 8);
      if (rc) return rc;
      P.stack = m->stack.as<double>();
      const size_t lds_bytes = nuts_lds_bytes<M, T::LDSL>();
      if (lds_bytes > 64 * 1024) EXMC_KMAXLDS((indep_kernel<M, T::G, T::LDSL>), lds_bytes);
      HIP_TRY(hipEventRecord(m->ev0, m->stream));
      EXMC_KLAUNCH(m->device, (indep_kernel<M, T::G, T::LDSL>), grid, dim3(kNutsBlock), lds_bytes, m->stream, P, mc);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(m->ev1, m->stream));
      return (int)EXMC_OK;
    } else {
      return fail(EXMC_ERR_UNSUPPORTED, "independent adaptation runs in the model kind's default layout");
    }
  });
}

// Every entry point that touches the handle's buffers, stream or counters goes through here. While
// a push-style stream run is in flight (exmc_hip_stream_start .. _finish) the launch is writing the
// page-locked trace and owns ev0 / ev1 and the counters: everything else is refused until
// exmc_hip_stream_finish has been called (the poller may be a thread of the caller's own).
int check_handle(const exmc_hip_model* m) {
  if (!m) return fail(EXMC_ERR_BADARG, "null model handle");
  return EXMC_OK;
}
int check_model(const exmc_hip_model* m) {
  if (!m) return fail(EXMC_ERR_BADARG, "null model handle");
  if (m->stream_in_flight.load(std::memory_order_acquire))
    return fail(EXMC_ERR_BADARG, "a stream run is in flight on this handle: call exmc_hip_stream_finish first");
  return EXMC_OK;
}

}  // namespace

// =========================================== C ABI ==========================================
extern "C" {

const char* exmc_hip_last_error(void) { return g_last_error.c_str(); }

int exmc_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int exmc_hip_model_create(int kind, int d, const double* data, int n_data, int device,
                          exmc_hip_model** out) {
  if (!out) return fail(EXMC_ERR_BADARG, "out is null");
  *out = nullptr;
  int ndev = exmc_hip_device_count();
  if (ndev <= 0)
    return fail(EXMC_ERR_NO_DEVICE, "no HIP device visible: libexmc_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(EXMC_ERR_BADARG, "device index out of range");
#ifdef EXMC_ONLY_CUSTOM
  if (kind != EXMC_MODEL_CUSTOM)
    return fail(EXMC_ERR_UNSUPPORTED, "this plug-in build carries one generated model (EXMC_MODEL_CUSTOM) only");
#endif
  exmc_hip_model* m = new exmc_hip_model();
  m->kind = kind;
  m->device = device;
  const double log2pi32 = f32r(std::log(f32r(2.0 * M_PI)));
  switch (kind) {
    case EXMC_MODEL_EIGHT_SCHOOLS: {
      if (n_data != 16 || !data) { delete m; return fail(EXMC_ERR_BADARG, "eight_schools needs y[8],sigma[8]"); }
      m->d = 10;
      for (int j = 0; j < 8; j++) {
        m->es.y[j] = data[j];
        m->es.sg[j] = data[8 + j];
        m->es.lsg[j] = std::log(data[8 + j]);
      }
      m->es.c_mu = log2pi32 + 2.0 * std::log(5.0);
      m->es.c_hc = f32r(std::log(2.0 / M_PI)) - std::log(5.0);
      m->es.c1 = log2pi32 + 2.0 * 0.0;
      break;
    }
    case EXMC_MODEL_SIMPLE: {
      if (n_data < 1 || n_data > 64 || !data) { delete m; return fail(EXMC_ERR_BADARG, "simple needs 1..64 observations"); }
      m->d = 2;
      m->sp.n = n_data;
      for (int i = 0; i < n_data; i++) m->sp.y[i] = data[i];
      m->sp.c_mu = log2pi32 + 2.0 * std::log(5.0);
      m->sp.log2pi32 = log2pi32;
      m->sp.tiny32 = f32r(1.0e-30);
      break;
    }
    case EXMC_MODEL_SV: {
      if (n_data != 100 || !data) { delete m; return fail(EXMC_ERR_BADARG, "sv is compiled for T = 100 returns"); }
      m->d = 102;
      static const double lanczos[9] = {0.99999999999980993,  676.5203681218851,     -1259.1392167224028,
                                        771.32342877765313,   -176.61502916214059,   12.507343278686905,
                                        -0.13857109526572012, 9.9843695780195716e-6, 1.5056327351493116e-7};
      for (int i = 0; i < 100; i++) m->sv.r[i] = data[i];
      for (int i = 0; i < 9; i++) m->sv.lanczos[i] = f32r(lanczos[i]);
      m->sv.half_log_2pi32 = f32r(0.5 * std::log(2.0 * M_PI));
      m->sv.log2pi32 = log2pi32;
      m->sv.pi32 = f32r(M_PI);
      m->sv.tiny32 = f32r(1.0e-30);
      m->sv.lam_s = 50.0;
      m->sv.lam_n = f32r(0.1);
      m->sv.log_lam_s32 = f32r(std::log(50.0));
      m->sv.log_lam_n32 = f32r(std::log(f32r(0.1)));
      break;
    }
    case EXMC_MODEL_LOGISTIC: {
      // data = X[N][20] row-major, y[N]
      if (!data || n_data < 21 || n_data % 21 != 0) { delete m; return fail(EXMC_ERR_BADARG, "logistic needs X[N][20], y[N]"); }
      m->d = 21;
      m->lg.N = n_data / 21;
      m->lg.c10 = log2pi32 + 2.0 * std::log(10.0);
      m->lg.lo = f32r(1.0e-7);
      m->lg.hi = 1.0 - f32r(1.0e-7);
      break;
    }
    case EXMC_MODEL_RADON: {
      // data = u[85], county_start[86], floor[N], y[N] (observations sorted by county)
      const int J = 85;
      if (!data || n_data < 2 * J + 1 || (n_data - (2 * J + 1)) % 2 != 0) { delete m; return fail(EXMC_ERR_BADARG, "radon needs u[85], start[86], floor[N], y[N]"); }
      const int N = (n_data - (2 * J + 1)) / 2;
      if ((int)data[J] != 0 || (int)data[2 * J] != N) { delete m; return fail(EXMC_ERR_BADARG, "radon county offsets do not cover the observations"); }
      // the 64-lane layout keeps a lane's observations in registers, 16 slots of 64 (exmc_models.hpp)
      if (N > 1024) { delete m; return fail(EXMC_ERR_UNSUPPORTED, "the radon kind holds at most 1024 observations"); }
      for (int j = 0; j < J; j++)
        if (data[J + j + 1] < data[J + j]) { delete m; return fail(EXMC_ERR_BADARG, "radon county offsets must be non-decreasing"); }
      m->d = J + 5;
      m->rd.log2pi32 = log2pi32;
      m->rd.tiny32 = f32r(1.0e-30);
      m->rd.c_mu10 = log2pi32 + 2.0 * std::log(10.0);
      m->rd.c_n5 = log2pi32 + 2.0 * std::log(5.0);
      m->rd.c1 = log2pi32 + 2.0 * 0.0;
      m->rd.c_hc = f32r(std::log(2.0 / M_PI)) - std::log(2.5);
      break;
    }
#ifdef EXMC_CUSTOM_HEADER
    case EXMC_MODEL_CUSTOM: {
      int kGenData = EXMC_GEN_NDATA;
#ifdef EXMC_GEN_VEC
      kGenData += EXMC_GEN_NVU + 16 * EXMC_GEN_NLR;
#endif
#ifdef EXMC_GEN_LANES
      kGenData += EXMC_GEN_NLT;
#endif
      if (n_data != kGenData || (n_data > 0 && !data)) { delete m; return fail(EXMC_ERR_BADARG, "generated model: data length differs from the one it was generated for"); }
      m->d = EXMC_GEN_D;
      break;
    }
#endif
    default:
      delete m;
      return fail(EXMC_ERR_UNSUPPORTED, "model kind not compiled into libexmc_hip");
  }
  if (d != 0 && d != m->d) { delete m; return fail(EXMC_ERR_BADARG, "d does not match the model kind"); }
  auto bail = [&](int rc) { exmc_hip_model_destroy(m); return rc; };
  if (hipSetDevice(device) != hipSuccess) return bail(fail(EXMC_ERR_HIP, "hipSetDevice failed"));
  if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(fail(EXMC_ERR_HIP, "hipStreamCreate failed"));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
      m->simds = 4 * cus;
  }
  if (hipEventCreate(&m->ev0) != hipSuccess || hipEventCreate(&m->ev1) != hipSuccess)
    return bail(fail(EXMC_ERR_HIP, "hipEventCreate failed"));
  int rc = m->zig.ensure(768 * 8);
  if (rc) return bail(rc);
  rc = m->misc.ensure((8 + EXMC_HIP_MAX_D) * 8);
  if (rc) return bail(rc);
  if (hipMemcpy(m->zig.p, kZigKi, 256 * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->zig.as<double>() + 256, kZigWi, 256 * 8, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(m->zig.as<double>() + 512, kZigFi, 256 * 8, hipMemcpyHostToDevice) != hipSuccess)
    return bail(fail(EXMC_ERR_HIP, "table upload failed"));
  if (kind == EXMC_MODEL_LOGISTIC || kind == EXMC_MODEL_RADON) {
    std::vector<double> blob(data, data + n_data);
    size_t off_xat = 0, off_xa32 = 0, off_yp = 0;
    if (kind == EXMC_MODEL_LOGISTIC) {
      // MFMA operands: Xa = [1 | X] zero-padded to 24 / 32 features and Npad observations
      const int N = m->lg.N, K = 20, Npad = (N + 15) / 16 * 16;
      m->lg.Npad = Npad;
      off_xat = blob.size();
      blob.resize(blob.size() + (size_t)24 * Npad, 0.0);
      off_xa32 = blob.size();
      blob.resize(blob.size() + (size_t)Npad * 32, 0.0);
      off_yp = blob.size();
      blob.resize(blob.size() + (size_t)Npad, 0.0);
      for (int n = 0; n < N; n++) {
        blob[off_xat + n] = 1.0;
        blob[off_xa32 + (size_t)n * 32] = 1.0;
        for (int j = 0; j < K; j++) {
          const double x = data[(size_t)n * K + j];
          blob[off_xat + (size_t)(1 + j) * Npad + n] = x;
          blob[off_xa32 + (size_t)n * 32 + 1 + j] = x;
        }
        blob[off_yp + n] = data[(size_t)N * K + n];
      }
    }
    size_t off_pad = 0;
    constexpr int kRadonPad = Radon<64>::kObsPad;   // the capacity + one slot: every (slot, lane) index is an address
    if (kind == EXMC_MODEL_RADON) {
      // the 64-lane layout's copies of y, floor and county, zero-padded to 17 slots of 64 so that lane l
      // fetches slot s at base + 8 (64 s + l) with no clamp: [y | floor | 8 * county] (RadonConsts::pobs;
      // the county as the byte offset of its intercept in the kernel's alpha strip, an integer)
      const int J = 85, N = (n_data - (2 * J + 1)) / 2;
      off_pad = blob.size();
      blob.resize(blob.size() + (size_t)3 * kRadonPad, 0.0);
      for (int i = 0; i < N; i++) {
        blob[off_pad + i] = data[2 * J + 1 + N + i];
        blob[off_pad + kRadonPad + i] = data[2 * J + 1 + i];
      }
      for (int j = 0; j < J; j++)
        for (int i = (int)data[J + j]; i < (int)data[J + j + 1]; i++) {
          const uint64_t off8 = 8ull * (uint64_t)j;
          std::memcpy(&blob[off_pad + 2 * kRadonPad + i], &off8, 8);
        }
    }
    rc = m->data.ensure(blob.size() * 8);
    if (rc) return bail(rc);
    if (hipMemcpy(m->data.p, blob.data(), blob.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(EXMC_ERR_HIP, "model data upload failed"));
    const double* base = m->data.as<double>();
    if (kind == EXMC_MODEL_LOGISTIC) {
      m->lg.X = base;
      m->lg.y = base + (size_t)m->lg.N * 20;
      m->lg.XaT = base + off_xat;
      m->lg.Xa32 = base + off_xa32;
      m->lg.ypad = base + off_yp;
    } else {
      const int J = 85, N = (n_data - (2 * J + 1)) / 2;
      m->rd.u = base;
      m->rd.cs = base + J;
      m->rd.fl = base + 2 * J + 1;
      m->rd.y = base + 2 * J + 1 + N;
      m->rd.pobs = base + off_pad;
    }
  }
#ifdef EXMC_CUSTOM_HEADER
  if (kind == EXMC_MODEL_CUSTOM) {
    // everything that depends on the data only, evaluated once (same arithmetic as the kernels)
#ifdef EXMC_GEN_VEC
    std::vector<double> folded(EXMC_GEN_NCONST + EXMC_GEN_NVC + 16 * EXMC_GEN_NLC);
    {
      const double* vdata = data + EXMC_GEN_NDATA;
      double* vc = folded.data() + EXMC_GEN_NCONST;
      exmc_gen_vfold(vdata, vc);
      for (int l = 0; l < 16; l++)
        exmc_gen_vfold_lane(vc, vdata + EXMC_GEN_NVU + l * EXMC_GEN_NLR,
                            vc + EXMC_GEN_NVC + l * EXMC_GEN_NLC);
    }
#else
    std::vector<double> folded(EXMC_GEN_NCONST);
#endif
#ifdef EXMC_GEN_ONE_LANE
    exmc_gen_fold(data, folded.data());
#endif
#ifdef EXMC_GEN_LANES
    while (folded.size() % 16) folded.push_back(0.0);   // the lane layout's table on a 128-byte boundary (codegen_lanes.py)
#endif
    const size_t n_folded = folded.size();
#ifdef EXMC_GEN_LANES
    // the lane layout's tables were folded when the model was generated: they travel as they are
    folded.insert(folded.end(), data + EXMC_GEN_LOFF, data + EXMC_GEN_LOFF + EXMC_GEN_NLT);
#endif
    rc = m->data.ensure(folded.size() * 8);
    if (rc) return bail(rc);
    if (hipMemcpy(m->data.p, folded.data(), folded.size() * 8, hipMemcpyHostToDevice) != hipSuccess)
      return bail(fail(EXMC_ERR_HIP, "model data upload failed"));
    m->cu.c = m->data.as<double>();
    m->cu.vc = m->data.as<double>() + EXMC_GEN_NCONST;
    m->cu.lt = m->data.as<double>() + n_folded;
  }
#endif
  rc = default_flat_order(m);
  if (rc) return bail(rc);
  *out = m;
  return EXMC_OK;
}

int exmc_hip_model_set_flat_order(exmc_hip_model* m, const int32_t* perm, int d) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!perm || d != m->d) return fail(EXMC_ERR_BADARG, "flat order needs d entries");
  HIP_TRY(hipSetDevice(m->device));
  m->res_C = 0;   // resident chains were initialised under the previous order
  return set_flat_order(m, perm);
}

void exmc_hip_model_destroy(exmc_hip_model* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->stream) (void)hipStreamSynchronize(m->stream);   // a stream run may still be writing its page-locked trace
  m->zig.release(); m->tuning.release(); m->state.release(); m->stack.release();
  m->misc.release(); m->trace.release(); m->io.release(); m->data.release(); m->flat.release(); m->scores.release(); m->dense.release(); m->esswork.release();
  m->densep.release(); m->densews.release(); m->migboard.release();
  if (m->ev0) (void)hipEventDestroy(m->ev0);
  if (m->ev1) (void)hipEventDestroy(m->ev1);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  if (m->pin_host) (void)hipHostFree(m->pin_host);
  delete m;
}

int exmc_hip_model_dim(const exmc_hip_model* m) { return m ? m->d : -1; }
int exmc_hip_model_default_lanes(const exmc_hip_model* m) { return m ? default_lanes(m->kind) : -1; }

int exmc_hip_model_default_warmup_lanes(const exmc_hip_model* m) {
  if (!m) return -1;
  // logistic: the shared warmup is ONE chain, so its 500 observations are best spread over a whole
  // wavefront (8 per lane instead of 32: 158 -> 62 ms); sampling keeps 16 lanes per chain
  if (m->kind == EXMC_MODEL_LOGISTIC) return 64;
#if defined(EXMC_GEN_LANES) && EXMC_GEN_LANES < 64
  // a generated lane layout: the chain's model terms over the whole wavefront (CustomSplit)
  if (m->kind == EXMC_MODEL_CUSTOM) return 64;
#endif
  return default_lanes(m->kind);
}
int exmc_hip_model_default_dense_lanes(const exmc_hip_model* m) {
  if (!m) return -1;
  switch (m->kind) {
    case EXMC_MODEL_SV: return 64;
    case EXMC_MODEL_RADON: return 64;
    case EXMC_MODEL_LOGISTIC: return 16;
    default: return 1;
  }
}
void* exmc_hip_model_stream(const exmc_hip_model* m) { return m ? (void*)m->stream : nullptr; }
double exmc_hip_last_kernel_ms(const exmc_hip_model* m) { return m ? m->last_ms : 0.0; }

int exmc_hip_logp_grad_host(exmc_hip_model* m, const double* q, int C, int lanes, double* logp,
                            double* grad) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!q || C < 1) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  lanes = resolve_lanes(m, lanes);
  const int d = m->d;
  // io: q [d][C], grad [d][C], logp [C]
  int rc = m->io.ensure(((size_t)2 * d * C + C) * 8);
  if (rc) return rc;
  std::vector<double> h((size_t)d * C);
  for (int c = 0; c < C; c++)
    for (int i = 0; i < d; i++) h[(size_t)i * C + c] = q[(size_t)c * d + i];
  double* dq = m->io.as<double>();
  double* dg = dq + (size_t)d * C;
  double* dl = dg + (size_t)d * C;
  HIP_TRY(hipMemcpyAsync(dq, h.data(), h.size() * 8, hipMemcpyHostToDevice, m->stream));
  rc = dispatch(m, lanes, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    const size_t xlds = aux_lds_bytes<typename T::M>();
    EXMC_KLAUNCH(m->device, (logp_grad_kernel<typename T::M, T::G>), grid_for(C, T::G, kBlock),
                 dim3(kBlock), xlds, m->stream, (const double*)dq, (int)C, (double*)dl, (double*)dg, mc);
    HIP_TRY(hipGetLastError());
    return (int)EXMC_OK;
  });
  if (rc) return rc;
  std::vector<double> hg((size_t)d * C), hl(C);
  HIP_TRY(hipMemcpyAsync(hg.data(), dg, hg.size() * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipMemcpyAsync(hl.data(), dl, (size_t)C * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  for (int c = 0; c < C; c++) {
    if (logp) logp[c] = hl[c];
    if (grad)
      for (int i = 0; i < d; i++) grad[(size_t)c * d + i] = hg[(size_t)i * C + c];
  }
  return EXMC_OK;
}

int exmc_hip_multi_step(exmc_hip_model* m, const double* q, const double* p, const double* g,
                        double eps, const double* inv_mass_host, int n_steps, int n_chains,
                        int lanes, double* all_q, double* all_p, double* all_logp, double* all_g) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!q || !p || !g || !inv_mass_host || n_steps < 0 || n_chains < 1 || !all_q || !all_p ||
      !all_logp || !all_g)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  lanes = resolve_lanes(m, lanes);
  int rc = upload_tuning(m, inv_mass_host);
  if (rc) return rc;
  MultiStepParams P;
  P.q = q; P.p = p; P.g = g;
  P.eps = eps;
  P.inv_mass = m->tuning.as<double>();
  P.n_steps = n_steps;
  P.n_chains = n_chains;
  P.all_q = all_q; P.all_p = all_p; P.all_g = all_g; P.all_logp = all_logp;
  return dispatch(m, lanes, [&](auto tag, const auto& mc) {
    using T = decltype(tag);
    HIP_TRY(hipEventRecord(m->ev0, m->stream));
    const size_t xlds = aux_lds_bytes<typename T::M>();
    EXMC_KLAUNCH(m->device, (multi_step_kernel<typename T::M, T::G>), grid_for(n_chains, T::G, kBlock),
                 dim3(kBlock), xlds, m->stream, P, mc);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->ev1, m->stream));
    return finish_timing(m);
  });
}

int exmc_hip_multi_step_host(exmc_hip_model* m, const double* q, const double* p, const double* g,
                             double eps, const double* inv_mass, int n_steps, int C, int lanes,
                             double* all_q, double* all_p, double* all_logp, double* all_g) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!q || !p || !g || !inv_mass || n_steps < 0 || C < 1)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  const int d = m->d;
  const size_t vec = (size_t)d * C, rows = (size_t)n_steps * d * C;
  int rc = m->io.ensure((3 * vec + 3 * rows + (size_t)n_steps * C) * 8);
  if (rc) return rc;
  double* dq = m->io.as<double>();
  double* dp = dq + vec;
  double* dg = dp + vec;
  double* aq = dg + vec;
  double* ap = aq + rows;
  double* ag = ap + rows;
  double* al = ag + rows;
  std::vector<double> h(3 * vec);
  for (int c = 0; c < C; c++)
    for (int i = 0; i < d; i++) {
      h[(size_t)i * C + c] = q[(size_t)c * d + i];
      h[vec + (size_t)i * C + c] = p[(size_t)c * d + i];
      h[2 * vec + (size_t)i * C + c] = g[(size_t)c * d + i];
    }
  HIP_TRY(hipMemcpyAsync(dq, h.data(), h.size() * 8, hipMemcpyHostToDevice, m->stream));
  rc = exmc_hip_multi_step(m, dq, dp, dg, eps, inv_mass, n_steps, C, lanes, aq, ap, al, ag);
  if (rc) return rc;
  std::vector<double> out(3 * rows + (size_t)n_steps * C);
  HIP_TRY(hipMemcpyAsync(out.data(), aq, out.size() * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  if (all_q) transpose_trace_vec(out.data(), all_q, n_steps, d, C);
  if (all_p) transpose_trace_vec(out.data() + rows, all_p, n_steps, d, C);
  if (all_g) transpose_trace_vec(out.data() + 2 * rows, all_g, n_steps, d, C);
  if (all_logp) transpose_trace_scalar(out.data() + 3 * rows, all_logp, n_steps, C);
  return EXMC_OK;
}

int exmc_hip_transitions_host(exmc_hip_model* m, double* q, double* logp, double* grad,
                              uint64_t* rng, int C, int n_draws, double eps,
                              const double* inv_mass, int max_depth, int lanes,
                              exmc_hip_trace trace) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!q || !logp || !grad || !rng || !inv_mass || C < 1 || n_draws < 0)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  lanes = resolve_lanes(m, lanes);
  const int d = m->d;
  int rc = ensure_state(m, C);
  if (rc) return rc;
  rc = upload_tuning(m, inv_mass);
  if (rc) return rc;
  std::vector<double> hs(state_bytes(d, C) / 8);
  double* hq = hs.data();
  double* hg = hq + (size_t)d * C;
  double* hl = hg + (size_t)d * C;
  uint64_t* hr = (uint64_t*)(hl + C);
  for (int c = 0; c < C; c++) {
    for (int i = 0; i < d; i++) {
      hq[(size_t)i * C + c] = q[(size_t)c * d + i];
      hg[(size_t)i * C + c] = grad[(size_t)c * d + i];
    }
    hl[c] = logp[c];
    hr[c] = rng[2 * (size_t)c];
    hr[(size_t)C + c] = rng[2 * (size_t)c + 1];
  }
  HIP_TRY(hipMemcpyAsync(m->state.p, hs.data(), hs.size() * 8, hipMemcpyHostToDevice, m->stream));
  TraceLayout L = trace_layout(n_draws > 0 ? n_draws : 1, d, C);
  rc = m->trace.ensure(L.total);
  if (rc) return rc;
  rc = reset_counters(m);
  if (rc) return rc;
  rc = launch_nuts(m, lanes, C, n_draws, 0, eps, max_depth, trace_view(m->trace.p, L), true);
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(hs.data(), m->state.p, hs.size() * 8, hipMemcpyDeviceToHost, m->stream));
  HIP_TRY(hipStreamSynchronize(m->stream));
  for (int c = 0; c < C; c++) {
    for (int i = 0; i < d; i++) {
      q[(size_t)c * d + i] = hq[(size_t)i * C + c];
      grad[(size_t)c * d + i] = hg[(size_t)i * C + c];
    }
    logp[c] = hl[c];
    rng[2 * (size_t)c] = hr[c];
    rng[2 * (size_t)c + 1] = hr[(size_t)C + c];
  }
  if (n_draws > 0) return download_trace(m, L, n_draws, C, trace);
  return EXMC_OK;
}

namespace {
int warmup_impl(exmc_hip_model* m, const double* init_q, exmc_hip_opts o, const exmc_hip_tuning* start,
                exmc_hip_tuning* tuning) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!tuning || o.num_warmup < 0) return fail(EXMC_ERR_BADARG, "bad arguments");
  if (start) {
    if (!(start->epsilon > 0.0)) return fail(EXMC_ERR_BADARG, "warm start needs a positive step size");
    for (int i = 0; i < m->d; i++)
      if (!(start->inv_mass[i] > 0.0)) return fail(EXMC_ERR_BADARG, "warm start needs a positive inverse mass");
  }
  HIP_TRY(hipSetDevice(m->device));
  // lanes_per_chain = 0: the library's layout for the one-chain warmup (logistic, generated lane
  // layouts of fewer than 64 lanes: the whole wavefront), not the sampling default
  int lanes = o.lanes_per_chain > 0 ? o.lanes_per_chain : exmc_hip_model_default_warmup_lanes(m);
  int rc = ensure_state(m, 1);
  if (rc) return rc;
  const char* hw = std::getenv("EXMC_HIP_HOST_WARMUP");
  const bool host_driven = hw && hw[0] == '1';
#if defined(EXMC_GEN_LANES) && EXMC_GEN_LANES < 64
  // the host-driven form launches the sampling kernels: a generated layout's one-chain form is not
  // among them, so it runs in the sampling layout (other bits than the default, the same schedule)
  if (host_driven && m->kind == EXMC_MODEL_CUSTOM && lanes == 64) lanes = EXMC_GEN_LANES;
#endif
  rc = launch_init(m, lanes, 1, 0, o.seed, init_q, !host_driven);
  if (rc) return rc;
  if (start && o.num_warmup == 0) {   // sampler.ex:195-196: nothing to tune
    *tuning = *start;
    tuning->warmup_divergences = 0;
    return EXMC_OK;
  }
  // EXMC_HIP_HOST_WARMUP=1 keeps the adaptation scalars on the host (one launch per
  // transition); the default runs the whole schedule in one kernel. Both give the same bits.
  // (The host-driven form runs in sampling layouts only: not in a generated layout's one-chain form.)
  if (host_driven) return run_warmup(m, lanes, o, tuning, start);
  return run_warmup_device(m, lanes, o, tuning, start);
}
}  // namespace

int exmc_hip_warmup(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                    exmc_hip_tuning* tuning) {
  return warmup_impl(m, init_q, o, nullptr, tuning);
}

int exmc_hip_warmup_from(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                         const exmc_hip_tuning* warm_start, exmc_hip_tuning* tuning) {
  if (!warm_start) return fail(EXMC_ERR_BADARG, "warm_start is null");
  o.num_warmup = o.num_warmup < 50 ? o.num_warmup : 50;   // short_warmup, sampler.ex:188
  return warmup_impl(m, init_q, o, warm_start, tuning);
}

int exmc_hip_model_set_dense_mass(exmc_hip_model* m, const double* cov, const double* chol, int d) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!cov || !chol || d != m->d) return fail(EXMC_ERR_BADARG, "dense mass needs cov and chol_cov, d x d each");
  for (int i = 0; i < d; i++)
    if (!(chol[(size_t)i * d + i] > 0.0)) return fail(EXMC_ERR_BADARG, "chol_cov needs a positive diagonal");
  HIP_TRY(hipSetDevice(m->device));
  int rc = m->dense.ensure(2 * (size_t)d * d * 8);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(m->dense.p, cov, (size_t)d * d * 8, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(m->dense.as<double>() + (size_t)d * d, chol, (size_t)d * d * 8, hipMemcpyHostToDevice));
  m->h_dense.assign(cov, cov + (size_t)d * d);
  m->h_dense.insert(m->h_dense.end(), chol, chol + (size_t)d * d);
  m->densep_gd = 0;
  m->dense_on = true;
  m->res_C = 0;
  return EXMC_OK;
}

int exmc_hip_model_clear_dense_mass(exmc_hip_model* m) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  m->dense_on = false;
  return EXMC_OK;
}

int exmc_hip_warmup_dense(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                          exmc_hip_tuning* tuning, double* cov, double* chol) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!tuning || !cov || !chol || o.num_warmup < 0) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  const int lanes = resolve_lanes(m, o.lanes_per_chain);
  if (!dense_layout_ok(m, lanes))
    return fail(EXMC_ERR_UNSUPPORTED, "no dense mass matrix for this model at this lanes_per_chain (1; eight_schools and logistic 16; sv and radon 64)");
  m->dense_on = false;   // Phase I runs on the identity diagonal (sampler.ex:560-575)
  int rc = ensure_state(m, 1);
  if (rc) return rc;
  rc = launch_init(m, lanes, 1, 0, o.seed, init_q);
  if (rc) return rc;
  rc = run_warmup_device(m, lanes, o, tuning, nullptr, cov, chol);
  if (rc) return rc;
  return exmc_hip_model_set_dense_mass(m, cov, chol, m->d);   // in force for the sampling that follows
}

int exmc_hip_chains_init(exmc_hip_model* m, const exmc_hip_tuning* tuning, const double* init_q,
                         int n_chains, int chain_lo, int chain_hi, exmc_hip_opts o) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!tuning || n_chains < 1 || chain_lo < 0 || chain_hi > n_chains || chain_hi <= chain_lo)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  const int lanes = resolve_lanes(m, o.lanes_per_chain);
  const int C = chain_hi - chain_lo;
  m->res_C = 0;
  int rc = ensure_state(m, C);
  if (rc) return rc;
  rc = upload_tuning(m, tuning->inv_mass);
  if (rc) return rc;
  rc = launch_init(m, lanes, C, chain_lo, o.seed, init_q);
  if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(m->stream));
  m->res_C = C;
  m->res_lanes = lanes;
  m->res_eps = tuning->epsilon;
  m->res_max_depth = o.max_tree_depth;
  return EXMC_OK;
}

int exmc_hip_chains_advance(exmc_hip_model* m, int n_draws, int row_offset, exmc_hip_trace tr,
                            int64_t* leapfrogs, int32_t* divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (m->res_C < 1) return fail(EXMC_ERR_BADARG, "no resident chains: call exmc_hip_chains_init");
  if (n_draws < 0 || row_offset < 0) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  int rc = reset_counters(m);
  if (rc) return rc;
  TraceDev t;
  t.draws = tr.draws; t.logp = tr.logp; t.tree_depth = tr.tree_depth; t.n_steps = tr.n_steps;
  t.divergent = tr.divergent; t.accept_prob = tr.accept_prob; t.energy = tr.energy;
  rc = launch_nuts(m, m->res_lanes, m->res_C, n_draws, row_offset, m->res_eps, m->res_max_depth, t,
                   true);
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  return read_counters(m, leapfrogs, divergences);
}

int exmc_hip_sample_chains(exmc_hip_model* m, const exmc_hip_tuning* tuning, const double* init_q,
                           int n_chains, int chain_lo, int chain_hi, exmc_hip_opts o,
                           exmc_hip_trace tr, int64_t* total_leapfrogs,
                           int32_t* total_divergences) {
  if (o.num_samples < 0) return fail(EXMC_ERR_BADARG, "bad arguments");
  int rc = exmc_hip_chains_init(m, tuning, init_q, n_chains, chain_lo, chain_hi, o);
  if (rc) return rc;
  return exmc_hip_chains_advance(m, o.num_samples, 0, tr, total_leapfrogs, total_divergences);
}

int exmc_hip_sample_chains_host(exmc_hip_model* m, const exmc_hip_tuning* tuning,
                                const double* init_q, int n_chains, int chain_lo, int chain_hi,
                                exmc_hip_opts o, exmc_hip_trace tr, int64_t* total_leapfrogs,
                                int32_t* total_divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (chain_hi <= chain_lo || o.num_samples < 0) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  const int C = chain_hi - chain_lo;
  if (o.num_samples == 0) {   // nothing to draw: the chains are initialised, the traces stay empty
    if (total_leapfrogs) *total_leapfrogs = 0;
    if (total_divergences) *total_divergences = 0;
    return exmc_hip_chains_init(m, tuning, init_q, n_chains, chain_lo, chain_hi, o);
  }
  TraceLayout L = trace_layout(o.num_samples, m->d, C);
  int rc = m->trace.ensure(L.total);
  if (rc) return rc;
  TraceDev t = trace_view(m->trace.p, L);
  exmc_hip_trace dv;
  dv.draws = t.draws; dv.logp = t.logp; dv.tree_depth = t.tree_depth; dv.n_steps = t.n_steps;
  dv.divergent = t.divergent; dv.accept_prob = t.accept_prob; dv.energy = t.energy;
  rc = exmc_hip_sample_chains(m, tuning, init_q, n_chains, chain_lo, chain_hi, o, dv,
                              total_leapfrogs, total_divergences);
  if (rc) return rc;
  return download_trace(m, L, o.num_samples, C, tr);
}

int exmc_hip_sample_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                         exmc_hip_trace tr, exmc_hip_tuning* tuning_out, int32_t* divergences) {
  return exmc_hip_sample_warm_host(m, init_q, o, nullptr, tr, tuning_out, divergences);
}

int exmc_hip_sample_warm_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                              const exmc_hip_tuning* warm_start, exmc_hip_trace tr,
                              exmc_hip_tuning* tuning_out, int32_t* divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (o.num_samples < 1) return fail(EXMC_ERR_BADARG, "num_samples must be >= 1");
  // this call runs the DIAGONAL adaptation itself: a dense mass left on the handle by an earlier
  // exmc_hip_warmup_dense / _model_set_dense_mass belongs to that run's tuning, not to this one
  m->dense_on = false;
  exmc_hip_tuning tun;
  // one call, one layout: the warmup runs in the layout the draws are sampled in (sample/3 is one
  // chain from start to end; the shared warmup of sample_chains has a layout of its own)
  const int lanes = resolve_lanes(m, o.lanes_per_chain);
  o.lanes_per_chain = lanes;
  // leaves chain 0's state in m->state
  int rc = warm_start ? exmc_hip_warmup_from(m, init_q, o, warm_start, &tun)
                      : exmc_hip_warmup(m, init_q, o, &tun);
  if (rc) return rc;
  rc = upload_tuning(m, tun.inv_mass);
  if (rc) return rc;
  TraceLayout L = trace_layout(o.num_samples, m->d, 1);
  rc = m->trace.ensure(L.total);
  if (rc) return rc;
  rc = reset_counters(m);
  if (rc) return rc;
  rc = launch_nuts(m, lanes, 1, o.num_samples, 0, tun.epsilon, o.max_tree_depth,
                   trace_view(m->trace.p, L), true);
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  int32_t div = 0;
  rc = read_counters(m, nullptr, &div);
  if (rc) return rc;
  if (divergences) *divergences = div + tun.warmup_divergences;  // stats.divergences, sampler.ex:245
  if (tuning_out) *tuning_out = tun;
  return download_trace(m, L, o.num_samples, 1, tr);
}

int exmc_hip_sample_independent(exmc_hip_model* m, const double* init_q, int n_chains, int chain_lo,
                                int chain_hi, exmc_hip_opts o, exmc_hip_trace tr, double* tuning_host,
                                int64_t* total_leapfrogs, int32_t* total_divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (n_chains < 1 || chain_lo < 0 || chain_hi > n_chains || chain_hi <= chain_lo || o.num_samples < 0 ||
      o.num_warmup < 0)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  m->dense_on = false;   // every chain runs the diagonal adaptation itself
  const int C = chain_hi - chain_lo, d = m->d;
  const int lanes = resolve_lanes(m, o.lanes_per_chain);
  int rc = ensure_state(m, C);
  if (rc) return rc;
  rc = launch_init(m, lanes, C, chain_lo, o.seed, init_q);   // chain i: seed + 7919 i, as sample/3 seeds chain 0
  if (rc) return rc;
  const size_t n_tune = (size_t)C * (3 + d);
  rc = m->io.ensure(n_tune * 8);
  if (rc) return rc;
  rc = reset_counters(m);
  if (rc) return rc;
  TraceDev t;
  t.draws = tr.draws; t.logp = tr.logp; t.tree_depth = tr.tree_depth; t.n_steps = tr.n_steps;
  t.divergent = tr.divergent; t.accept_prob = tr.accept_prob; t.energy = tr.energy;
  rc = launch_independent(m, lanes, C, o, t, m->io.as<double>());
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  m->res_C = 0;   // the chains ran with tunings of their own: nothing chains_advance could continue
  if (tuning_host)
    HIP_TRY(hipMemcpy(tuning_host, m->io.p, n_tune * 8, hipMemcpyDeviceToHost));
  return read_counters(m, total_leapfrogs, total_divergences);
}

int exmc_hip_sample_independent_host(exmc_hip_model* m, const double* init_q, int n_chains, int chain_lo,
                                     int chain_hi, exmc_hip_opts o, exmc_hip_trace tr, double* tuning_host,
                                     int64_t* total_leapfrogs, int32_t* total_divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (chain_hi <= chain_lo || o.num_samples < 1) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  const int C = chain_hi - chain_lo;
  TraceLayout L = trace_layout(o.num_samples, m->d, C);
  int rc = m->trace.ensure(L.total);
  if (rc) return rc;
  TraceDev t = trace_view(m->trace.p, L);
  exmc_hip_trace dv;
  dv.draws = t.draws; dv.logp = t.logp; dv.tree_depth = t.tree_depth; dv.n_steps = t.n_steps;
  dv.divergent = t.divergent; dv.accept_prob = t.accept_prob; dv.energy = t.energy;
  rc = exmc_hip_sample_independent(m, init_q, n_chains, chain_lo, chain_hi, o, dv, tuning_host,
                                   total_leapfrogs, total_divergences);
  if (rc) return rc;
  return download_trace(m, L, o.num_samples, C, tr);
}

int exmc_hip_sample_dense_host(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                               exmc_hip_trace tr, exmc_hip_tuning* tuning_out, double* cov, double* chol,
                               int32_t* divergences) {
  // Sampler.sample/3 with dense_mass: true (sampler.ex:156, 170-176): the dense warmup, then the
  // warmup chain goes on sampling under the dense mass
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (o.num_samples < 1) return fail(EXMC_ERR_BADARG, "num_samples must be >= 1");
  exmc_hip_tuning tun;
  const int lanes = dense_layout_ok(m, resolve_lanes(m, o.lanes_per_chain)) ? resolve_lanes(m, o.lanes_per_chain) : 1;
  o.lanes_per_chain = lanes;
  int rc = exmc_hip_warmup_dense(m, init_q, o, &tun, cov, chol);   // leaves chain 0's state in m->state
  if (rc) return rc;
  rc = upload_tuning(m, tun.inv_mass);
  if (rc) return rc;
  TraceLayout L = trace_layout(o.num_samples, m->d, 1);
  rc = m->trace.ensure(L.total);
  if (rc) return rc;
  rc = reset_counters(m);
  if (rc) return rc;
  rc = launch_nuts(m, lanes, 1, o.num_samples, 0, tun.epsilon, o.max_tree_depth, trace_view(m->trace.p, L), true);
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  int32_t div = 0;
  rc = read_counters(m, nullptr, &div);
  if (rc) return rc;
  if (divergences) *divergences = div + tun.warmup_divergences;
  if (tuning_out) *tuning_out = tun;
  return download_trace(m, L, o.num_samples, 1, tr);
}

int exmc_hip_stream_begin(exmc_hip_model* m, const double* init_q, exmc_hip_opts o,
                          exmc_hip_tuning* tuning_out) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  m->dense_on = false;   // the diagonal adaptation follows (see exmc_hip_sample_warm_host)
  exmc_hip_tuning tun;
  o.lanes_per_chain = resolve_lanes(m, o.lanes_per_chain);   // one chain, one layout (as sample/3)
  int rc = exmc_hip_warmup(m, init_q, o, &tun);  // leaves chain 0's state in m->state
  if (rc) return rc;
  rc = upload_tuning(m, tun.inv_mass);
  if (rc) return rc;
  // the warmup chain stays resident and is advanced on demand (same state the one-launch
  // exmc_hip_sample_host continues from)
  m->res_C = 1;
  m->res_lanes = resolve_lanes(m, o.lanes_per_chain);
  m->res_eps = tun.epsilon;
  m->res_max_depth = o.max_tree_depth;
  if (tuning_out) *tuning_out = tun;
  return EXMC_OK;
}

int exmc_hip_stream_next_host(exmc_hip_model* m, int n_draws, exmc_hip_trace tr,
                              int32_t* divergences) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (m->res_C != 1) return fail(EXMC_ERR_BADARG, "no stream: call exmc_hip_stream_begin");
  if (n_draws < 1) return fail(EXMC_ERR_BADARG, "n_draws must be >= 1");
  HIP_TRY(hipSetDevice(m->device));
  TraceLayout L = trace_layout(n_draws, m->d, 1);
  int rc = m->trace.ensure(L.total);
  if (rc) return rc;
  rc = reset_counters(m);
  if (rc) return rc;
  rc = launch_nuts(m, m->res_lanes, 1, n_draws, 0, m->res_eps, m->res_max_depth,
                   trace_view(m->trace.p, L), true);
  if (rc) return rc;
  rc = finish_timing(m);
  if (rc) return rc;
  int32_t div = 0;
  rc = read_counters(m, nullptr, &div);
  if (rc) return rc;
  if (divergences) *divergences = div;
  return download_trace(m, L, n_draws, 1, tr);
}

namespace {
// the body of exmc_hip_stream_start once the handle is claimed
int stream_start_claimed(exmc_hip_model* m, int n_draws, exmc_hip_trace* view,
                         const volatile int32_t** progress) {
  if (m->res_C != 1) return fail(EXMC_ERR_BADARG, "no stream: call exmc_hip_stream_begin");
  if (n_draws < 1 || !view || !progress) return fail(EXMC_ERR_BADARG, "bad arguments");
  if (m->dense_on) return fail(EXMC_ERR_UNSUPPORTED, "a push-style stream runs under the diagonal mass");
  HIP_TRY(hipSetDevice(m->device));
  const TraceLayout L = trace_layout(n_draws, m->d, 1);
  const size_t need = 64 + L.total;
  if (m->pin_bytes < need) {
    if (m->pin_host) (void)hipHostFree(m->pin_host);
    m->pin_host = nullptr;
    m->pin_bytes = 0;
    HIP_TRY(hipHostMalloc(&m->pin_host, need, hipHostMallocMapped | hipHostMallocPortable));
    m->pin_bytes = need;
  }
  std::memset(m->pin_host, 0, 64);
  void* dev = nullptr;
  HIP_TRY(hipHostGetDevicePointer(&dev, m->pin_host, 0));
  int rc = reset_counters(m);
  if (rc) return rc;
  rc = launch_nuts(m, m->res_lanes, 1, n_draws, 0, m->res_eps, m->res_max_depth,
                   trace_view((char*)dev + 64, L), true, (int*)dev);
  if (rc) return rc;
  // with one chain the device layout [draw][dim][chain] is the host layout [draw][dim]
  const TraceDev h = trace_view((char*)m->pin_host + 64, L);
  view->draws = h.draws;
  view->logp = h.logp;
  view->tree_depth = h.tree_depth;
  view->n_steps = h.n_steps;
  view->divergent = h.divergent;
  view->accept_prob = h.accept_prob;
  view->energy = h.energy;
  *progress = (const volatile int32_t*)m->pin_host;
  return EXMC_OK;
}
}  // namespace

int exmc_hip_stream_start(exmc_hip_model* m, int n_draws, exmc_hip_trace* view,
                          const volatile int32_t** progress) {
  if (check_handle(m)) return EXMC_ERR_BADARG;
  // The claim is the compare-and-swap itself: of two callers racing on one handle (two dirty
  // schedulers of a VM) exactly one goes on, the other is refused before it touches the page-locked
  // trace or launches anything. Every failure below gives the claim back.
  bool idle = false;
  if (!m->stream_in_flight.compare_exchange_strong(idle, true, std::memory_order_acq_rel))
    return fail(EXMC_ERR_BADARG, "a stream run is in flight on this handle: call exmc_hip_stream_finish first");
  const int rc = stream_start_claimed(m, n_draws, view, progress);
  if (rc) {
    (void)hipStreamSynchronize(m->stream);   // whatever was queued before the failure has drained
    m->stream_in_flight.store(false, std::memory_order_release);
  }
  return rc;
}

int exmc_hip_stream_finish(exmc_hip_model* m, int32_t* divergences) {
  if (check_handle(m)) return EXMC_ERR_BADARG;
  if (!m->stream_in_flight.load(std::memory_order_acquire)) return fail(EXMC_ERR_BADARG, "no stream run in flight");
  // the handle stays busy until the launch has drained, whatever the calls below return
  int rc = (hipSetDevice(m->device) == hipSuccess) ? finish_timing(m) : fail(EXMC_ERR_HIP, "hipSetDevice failed");
  int32_t div = 0;
  if (!rc) rc = read_counters(m, nullptr, &div);
  if (rc) (void)hipStreamSynchronize(m->stream);
  m->stream_in_flight.store(false, std::memory_order_release);
  if (rc) return rc;
  if (divergences) *divergences = div;
  return EXMC_OK;
}

int exmc_hip_build_full_tree_host(int device, int C, int d, const double* q0, const double* p0,
                                  const double* g0, const double* logp0, const double* fwd_q,
                                  const double* fwd_p, const double* fwd_logp, const double* fwd_g,
                                  int n_fwd, const double* bwd_q, const double* bwd_p,
                                  const double* bwd_logp, const double* bwd_g, int n_bwd,
                                  const double* inv_mass, const double* jlp0, int max_depth,
                                  const uint64_t* seeds, double* q_out, double* logp_out,
                                  double* g_out, int32_t* n_steps, int32_t* divergent,
                                  double* accept_sum, int32_t* depth) {
  if (C < 1 || d < 1 || n_fwd < 0 || n_bwd < 0 || max_depth < 0 || max_depth > kFtLevels ||
      !q0 || !p0 || !g0 || !logp0 || !inv_mass || !jlp0 || !seeds || !q_out || !logp_out ||
      !g_out || !n_steps || !divergent || !accept_sum || !depth ||
      (n_fwd > 0 && (!fwd_q || !fwd_p || !fwd_logp || !fwd_g)) ||
      (n_bwd > 0 && (!bwd_q || !bwd_p || !bwd_logp || !bwd_g)))
    return fail(EXMC_ERR_BADARG, "bad arguments");   // NifResult badarg (lib.rs)
  int ndev = exmc_hip_device_count();
  if (ndev <= 0)
    return fail(EXMC_ERR_NO_DEVICE, "no HIP device visible: libexmc_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(EXMC_ERR_BADARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  const size_t vec = (size_t)C * d, fw = (size_t)C * n_fwd * d, bw = (size_t)C * n_bwd * d;
  // one device arena: inputs, scratch, outputs (doubles), then int32 outputs
  const size_t n_in = 3 * vec + C + 3 * fw + (size_t)C * n_fwd + 3 * bw + (size_t)C * n_bwd + d + C + C;
  const size_t n_scr = (size_t)C * (kFtLevels + 3) * d;
  const size_t n_out = 2 * vec + 2 * (size_t)C;
  const size_t total = (n_in + n_scr + n_out) * 8 + 3 * (size_t)C * 4;
  DevBuf arena;
  int rc = arena.ensure(total);
  if (rc) return rc;
  double* b = arena.as<double>();
  FullTreeParams P;
  P.n_chains = C; P.d = d; P.n_fwd = n_fwd; P.n_bwd = n_bwd; P.max_depth = max_depth;
  std::vector<std::pair<const void*, size_t>> ups;
  auto put = [&](const double* src, size_t n) {
    double* dst = b;
    if (n) ups.push_back({src, n});
    b += n;
    return (const double*)dst;
  };
  P.q0 = put(q0, vec); P.p0 = put(p0, vec); P.g0 = put(g0, vec); P.logp0 = put(logp0, C);
  P.fwd_q = put(fwd_q, fw); P.fwd_p = put(fwd_p, fw); P.fwd_g = put(fwd_g, fw);
  P.fwd_logp = put(fwd_logp, (size_t)C * n_fwd);
  P.bwd_q = put(bwd_q, bw); P.bwd_p = put(bwd_p, bw); P.bwd_g = put(bwd_g, bw);
  P.bwd_logp = put(bwd_logp, (size_t)C * n_bwd);
  P.inv_mass = put(inv_mass, d); P.jlp0 = put(jlp0, C);
  P.seeds = (const uint64_t*)put((const double*)seeds, C);
  {
    double* dst = arena.as<double>();
    for (auto& u : ups) {
      if (hipMemcpy(dst, u.first, u.second * 8, hipMemcpyHostToDevice) != hipSuccess) {
        arena.release();
        return fail(EXMC_ERR_HIP, "upload failed");
      }
      dst += u.second;
    }
  }
  P.scratch = b; b += n_scr;
  P.out_q = b; b += vec;
  P.out_g = b; b += vec;
  P.out_logp = b; b += C;
  P.out_accept_sum = b; b += C;
  int32_t* ib = (int32_t*)b;
  P.out_n_steps = ib; P.out_divergent = ib + C; P.out_depth = ib + 2 * (size_t)C;
  hipLaunchKernelGGL(full_tree_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, 0, P);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(q_out, P.out_q, vec * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(g_out, P.out_g, vec * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(logp_out, P.out_logp, (size_t)C * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(accept_sum, P.out_accept_sum, (size_t)C * 8, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(n_steps, P.out_n_steps, (size_t)C * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(divergent, P.out_divergent, (size_t)C * 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(depth, P.out_depth, (size_t)C * 4, hipMemcpyDeviceToHost);
  arena.release();
  if (e != hipSuccess) return fail(EXMC_ERR_HIP, std::string("build_full_tree: ") + hipGetErrorString(e));
  return EXMC_OK;
}

namespace {
// Diagnostics.ess of every series of a [S][D][C] device array: one lane per series up to lag 15,
// then a wavefront per unfinished series (ess_tail_kernel needs the series in LDS; longer ones
// stay with their lane to the end)
int launch_ess(exmc_hip_model* m, const double* src, int n_draws, int d, int n_chains, double* ess_dev) {
  const size_t series = (size_t)d * n_chains;
  const dim3 grid((unsigned)((series + kEssBlock - 1) / kEssBlock));
  const size_t lds = (size_t)n_draws * 8;
  if (lds > 160 * 1024) {
    hipLaunchKernelGGL(ess_series_kernel, grid, dim3(kEssBlock), 0, m->stream, src, n_draws, d, n_chains,
                       ess_dev, (int*)nullptr, (EssTailItem*)nullptr);
    HIP_TRY(hipGetLastError());
    return EXMC_OK;
  }
  int rc = m->esswork.ensure(16 + series * sizeof(EssTailItem));
  if (rc) return rc;
  int* count = (int*)m->esswork.p;
  EssTailItem* items = (EssTailItem*)((char*)m->esswork.p + 16);
  HIP_TRY(hipMemsetAsync(count, 0, 16, m->stream));
  hipLaunchKernelGGL(ess_series_kernel, grid, dim3(kEssBlock), 0, m->stream, src, n_draws, d, n_chains,
                     ess_dev, count, items);
  HIP_TRY(hipGetLastError());
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)ess_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds));
  const unsigned tail_blocks = (unsigned)(series < 2048 ? series : 2048);
  hipLaunchKernelGGL(ess_tail_kernel, dim3(tail_blocks), dim3(64), lds, m->stream, src, n_draws, d,
                     n_chains, ess_dev, (const int*)count, (const EssTailItem*)items);
  HIP_TRY(hipGetLastError());
  return EXMC_OK;
}
}  // namespace

int exmc_hip_ess(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                 double* ess_dev) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!draws_dev || !ess_dev || n_draws < 1 || d < 1 || n_chains < 1)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  HIP_TRY(hipEventRecord(m->ev0, m->stream));
  int rc = launch_ess(m, draws_dev, n_draws, d, n_chains, ess_dev);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(m->ev1, m->stream));
  return finish_timing(m);
}

int exmc_hip_ess_bulk(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                      double* ess_dev) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!draws_dev || !ess_dev || n_draws < 1 || d < 1 || n_chains < 1)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  if (n_draws < 4) return exmc_hip_ess(m, draws_dev, n_draws, d, n_chains, ess_dev);   // diagnostics.ex:62
  const size_t lds = (size_t)n_draws * 8;
  if (lds > 160 * 1024) return fail(EXMC_ERR_UNSUPPORTED, "n_draws too large for the LDS rank kernel (max 20480)");
  HIP_TRY(hipSetDevice(m->device));
  const size_t series = (size_t)d * n_chains;
  int rc = m->scores.ensure(series * (size_t)n_draws * 8);   // the normal scores, [S][D][C]
  if (rc) return rc;
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute((const void*)rank_scores_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  HIP_TRY(hipEventRecord(m->ev0, m->stream));
  int P = 2;
  while (P < n_draws) P <<= 1;
  const char* re = std::getenv("EXMC_HIP_RANK_SORT");   // 0: the counting kernel (A/B runs, tests)
  if (P <= kRankSortMaxP && !(re && re[0] == '0')) {
    // ranks by sorting the series in LDS (exmc_kernels.hpp rank_scores_sort_kernel)
    hipLaunchKernelGGL(rank_scores_sort_kernel, dim3((unsigned)series), dim3(256), (size_t)P * 12, m->stream,
                       draws_dev, n_draws, P, d, n_chains, m->scores.as<double>());
  } else {
    hipLaunchKernelGGL(rank_scores_kernel, dim3((unsigned)series), dim3(256), lds, m->stream,
                       draws_dev, n_draws, d, n_chains, m->scores.as<double>());
  }
  HIP_TRY(hipGetLastError());
  rc = launch_ess(m, (const double*)m->scores.as<double>(), n_draws, d, n_chains, ess_dev);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(m->ev1, m->stream));
  return finish_timing(m);
}

int exmc_hip_rhat(exmc_hip_model* m, const double* draws_dev, int n_draws, int d, int n_chains,
                  double* rhat_dev) {
  if (check_model(m)) return EXMC_ERR_BADARG;
  if (!draws_dev || !rhat_dev || n_draws < 4 || d < 1 || n_chains < 1)
    return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(m->device));
  int rc = m->io.ensure((size_t)d * 4 * n_chains * 8);   // half-chain means and variances
  if (rc) return rc;
  HIP_TRY(hipEventRecord(m->ev0, m->stream));
  const unsigned slices = (unsigned)((2 * n_chains + 255) / 256);
  hipLaunchKernelGGL(rhat_kernel, dim3((unsigned)d, slices), dim3(256), 0, m->stream, draws_dev,
                     n_draws, d, n_chains, m->io.as<double>(), rhat_dev, 0);
  hipLaunchKernelGGL(rhat_kernel, dim3((unsigned)d), dim3(64), 0, m->stream, draws_dev, n_draws, d,
                     n_chains, m->io.as<double>(), rhat_dev, 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(m->ev1, m->stream));
  return finish_timing(m);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// The NIF's incremental trajectory interface, batched (include/exmc_hip.h; kernels in
// exmc_native_tree.hpp). Blocking calls on the null stream: the interface is a compatibility
// seam (one call per doubling), not a throughput path.
// ------------------------------------------------------------------------------------------
struct exmc_hip_traj {
  int device = 0, C = 0, d = 0;
  DevBuf state;
  TrajDev T{};
};

namespace {

// carve a TrajDev out of one allocation: 9 vectors [C][d], 3 scalars [C], 4 int32 [C]
size_t traj_bytes(int C, int d) { return ((size_t)9 * C * d + 3 * (size_t)C) * 8 + 4 * (size_t)C * 4; }
TrajDev traj_view(void* base, int C, int d) {
  TrajDev t;
  double* b = (double*)base;
  const size_t v = (size_t)C * d;
  t.qL = b; t.pL = b + v; t.gL = b + 2 * v; t.qR = b + 3 * v; t.pR = b + 4 * v; t.gR = b + 5 * v;
  t.qP = b + 6 * v; t.gP = b + 7 * v; t.rho = b + 8 * v;
  b += 9 * v;
  t.logpP = b; t.lsw = b + C; t.acc = b + 2 * (size_t)C;
  int32_t* ib = (int32_t*)(b + 3 * (size_t)C);
  t.n = ib; t.depth = ib + C; t.div = ib + 2 * (size_t)C; t.turn = ib + 3 * (size_t)C;
  return t;
}

int select_device(int device) {
  const int ndev = exmc_hip_device_count();
  if (ndev <= 0)
    return fail(EXMC_ERR_NO_DEVICE, "no HIP device visible: libexmc_hip has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(EXMC_ERR_BADARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  return EXMC_OK;
}

// uploads of one subtree call: the four state arrays, inv_mass and the per-chain scalars
struct SubtreeUpload {
  DevBuf buf;
  SubtreeParams P{};
  int fill(int C, int d, const double* all_q, const double* all_p, const double* all_logp,
           const double* all_g, int n_states, const double* inv_mass, const double* jlp0,
           const int32_t* depth, const int32_t* go_right, const uint64_t* seeds) {
    if (C < 1 || d < 1 || n_states < 1 || !all_q || !all_p || !all_logp || !all_g || !inv_mass ||
        !jlp0 || !depth || !go_right || !seeds)
      return fail(EXMC_ERR_BADARG, "bad arguments");
    for (int c = 0; c < C; c++)
      if (depth[c] >= kFtLevels || (depth[c] >= 0 && ((size_t)1 << depth[c]) > (size_t)n_states))
        return fail(EXMC_ERR_BADARG, "depth needs more pre-computed states than were passed");
    const size_t st = (size_t)C * n_states * d, sc = (size_t)C * n_states;
    const size_t nd = 3 * st + sc + d + C /*jlp0*/ + C /*seeds*/ + (size_t)C * (kFtLevels + 2) * d;
    int rc = buf.ensure(nd * 8 + 2 * (size_t)C * 4);
    if (rc) return rc;
    double* b = buf.as<double>();
    auto up = [&](const void* src, size_t n_doubles) -> const double* {
      double* dst = b;
      if (hipMemcpy(dst, src, n_doubles * 8, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
      b += n_doubles;
      return dst;
    };
    P.n_chains = C; P.d = d; P.n_states = n_states;
    P.all_q = up(all_q, st); P.all_p = up(all_p, st); P.all_g = up(all_g, st);
    P.all_logp = up(all_logp, sc);
    P.inv_mass = up(inv_mass, d); P.jlp0 = up(jlp0, C);
    P.seeds = (const uint64_t*)up(seeds, C);
    P.scratch = b; b += (size_t)C * (kFtLevels + 2) * d;
    int32_t* ib = (int32_t*)b;
    if (!P.all_q || !P.all_p || !P.all_g || !P.all_logp || !P.inv_mass || !P.jlp0 || !P.seeds ||
        hipMemcpy(ib, depth, (size_t)C * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ib + C, go_right, (size_t)C * 4, hipMemcpyHostToDevice) != hipSuccess)
      return fail(EXMC_ERR_HIP, "upload failed");
    P.depth = ib; P.go_right = ib + C;
    return EXMC_OK;
  }
};

// scratch of exmc_hip_leapfrog_chain_normal_host (see there), one per device, never freed
struct ChainScratch {
  std::mutex mu;
  void* dev = nullptr;
  void* host = nullptr;
};
constexpr size_t kChainScratchBytes = (size_t)8 << 20;
constexpr int kChainScratchDevices = 16;
constexpr size_t kChainZeroCopyBytes = (size_t)1 << 20;   // rows the kernel writes straight into the staging buffer
ChainScratch g_chain_scratch[kChainScratchDevices];

int down(void* dst, const void* src, size_t bytes) {
  if (!dst) return EXMC_OK;
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return EXMC_OK;
}

}  // namespace

extern "C" {

int exmc_hip_traj_create(int device, int C, int d, const double* q, const double* p,
                         const double* grad, const double* logp, exmc_hip_traj** out) {
  if (!out || C < 1 || d < 1 || !q || !p || !grad || !logp) return fail(EXMC_ERR_BADARG, "bad arguments");
  *out = nullptr;
  int rc = select_device(device);
  if (rc) return rc;
  exmc_hip_traj* t = new exmc_hip_traj();
  t->device = device; t->C = C; t->d = d;
  rc = t->state.ensure(traj_bytes(C, d));
  if (rc) { delete t; return rc; }
  t->T = traj_view(t->state.p, C, d);
  const size_t v = (size_t)C * d * 8;
  hipError_t e = hipMemset(t->state.p, 0, traj_bytes(C, d));   // lsw, acc, n, depth, div, turn = 0
  // Trajectory::new (types.rs:136-152): both endpoints and the proposal are the start state, rho = p
  const double* src[9] = {q, p, grad, q, p, grad, q, grad, p};
  double* dst[9] = {t->T.qL, t->T.pL, t->T.gL, t->T.qR, t->T.pR, t->T.gR, t->T.qP, t->T.gP, t->T.rho};
  for (int i = 0; i < 9 && e == hipSuccess; i++) e = hipMemcpy(dst[i], src[i], v, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(t->T.logpP, logp, (size_t)C * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    t->state.release();
    delete t;
    return fail(EXMC_ERR_HIP, std::string("traj_create: ") + hipGetErrorString(e));
  }
  *out = t;
  return EXMC_OK;
}

void exmc_hip_traj_destroy(exmc_hip_traj* t) {
  if (!t) return;
  (void)hipSetDevice(t->device);
  t->state.release();
  delete t;
}

int exmc_hip_traj_get_endpoint_host(exmc_hip_traj* t, const int32_t* go_right, double* q, double* p,
                                    double* grad) {
  if (!t || !go_right || !q || !p || !grad) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(t->device));
  const size_t row = (size_t)t->d * 8;
  for (int c = 0; c < t->C; c++) {
    const size_t o = (size_t)c * t->d;
    const bool r = go_right[c] != 0;
    HIP_TRY(hipMemcpy(q + o, (r ? t->T.qR : t->T.qL) + o, row, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(p + o, (r ? t->T.pR : t->T.pL) + o, row, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(grad + o, (r ? t->T.gR : t->T.gL) + o, row, hipMemcpyDeviceToHost));
  }
  return EXMC_OK;
}

int exmc_hip_traj_build_and_merge_host(exmc_hip_traj* t, const double* all_q, const double* all_p,
                                       const double* all_logp, const double* all_grad,
                                       int n_states, const double* inv_mass, const double* jlp0,
                                       const int32_t* depth, const int32_t* go_right,
                                       const uint64_t* seeds) {
  if (!t) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(t->device));
  SubtreeUpload u;
  int rc = u.fill(t->C, t->d, all_q, all_p, all_logp, all_grad, n_states, inv_mass, jlp0, depth,
                  go_right, seeds);
  if (rc == EXMC_OK) {
    u.P.T = t->T;
    hipLaunchKernelGGL(traj_build_and_merge_kernel, dim3((unsigned)((t->C + 63) / 64)), dim3(64), 0,
                       0, u.P);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = fail(EXMC_ERR_HIP, std::string("build_and_merge: ") + hipGetErrorString(e));
  }
  u.buf.release();
  return rc;
}

int exmc_hip_traj_is_terminated_host(exmc_hip_traj* t, int32_t* out) {
  if (!t || !out) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(t->device));
  std::vector<int32_t> dv(t->C), tn(t->C);
  HIP_TRY(hipMemcpy(dv.data(), t->T.div, (size_t)t->C * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(tn.data(), t->T.turn, (size_t)t->C * 4, hipMemcpyDeviceToHost));
  for (int c = 0; c < t->C; c++) out[c] = (dv[c] || tn[c]) ? 1 : 0;   // types.rs:160-162
  return EXMC_OK;
}

int exmc_hip_traj_get_result_host(exmc_hip_traj* t, double* q, double* logp, double* grad,
                                  int32_t* n_steps, int32_t* divergent, double* accept_sum,
                                  int32_t* depth) {
  if (!t) return fail(EXMC_ERR_BADARG, "bad arguments");
  HIP_TRY(hipSetDevice(t->device));
  const size_t v = (size_t)t->C * t->d * 8, s8 = (size_t)t->C * 8, s4 = (size_t)t->C * 4;
  int rc = down(q, t->T.qP, v);
  if (!rc) rc = down(grad, t->T.gP, v);
  if (!rc) rc = down(logp, t->T.logpP, s8);
  if (!rc) rc = down(accept_sum, t->T.acc, s8);
  if (!rc) rc = down(n_steps, t->T.n, s4);
  if (!rc) rc = down(divergent, t->T.div, s4);
  if (!rc) rc = down(depth, t->T.depth, s4);
  return rc;
}

int exmc_hip_build_subtree_host(int device, int C, int d, const double* all_q, const double* all_p,
                                const double* all_logp, const double* all_grad, int n_states,
                                const double* inv_mass, const double* jlp0, const int32_t* depth,
                                const int32_t* going_right, const uint64_t* seeds, double* q_left,
                                double* p_left, double* grad_left, double* q_right, double* p_right,
                                double* grad_right, double* q_prop, double* logp_prop,
                                double* grad_prop, double* log_sum_weight, int32_t* n_steps,
                                int32_t* divergent, double* accept_sum, int32_t* turning,
                                int32_t* subtree_depth, double* rho) {
  int rc = select_device(device);
  if (rc) return rc;
  SubtreeUpload u;
  rc = u.fill(C, d, all_q, all_p, all_logp, all_grad, n_states, inv_mass, jlp0, depth, going_right,
              seeds);
  DevBuf outb;
  if (rc == EXMC_OK) rc = outb.ensure(traj_bytes(C, d));
  if (rc == EXMC_OK) {
    u.P.out = traj_view(outb.p, C, d);
    hipError_t e = hipMemset(outb.p, 0, traj_bytes(C, d));
    if (e == hipSuccess) {
      hipLaunchKernelGGL(build_subtree_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, 0, u.P);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = fail(EXMC_ERR_HIP, std::string("build_subtree: ") + hipGetErrorString(e));
  }
  if (rc == EXMC_OK) {
    const TrajDev& o = u.P.out;
    const size_t v = (size_t)C * d * 8, s8 = (size_t)C * 8, s4 = (size_t)C * 4;
    const std::pair<void*, const void*> vec[9] = {{q_left, o.qL}, {p_left, o.pL}, {grad_left, o.gL},
                                                  {q_right, o.qR}, {p_right, o.pR}, {grad_right, o.gR},
                                                  {q_prop, o.qP}, {grad_prop, o.gP}, {rho, o.rho}};
    for (int i = 0; i < 9 && !rc; i++) rc = down(vec[i].first, vec[i].second, v);
    if (!rc) rc = down(logp_prop, o.logpP, s8);
    if (!rc) rc = down(log_sum_weight, o.lsw, s8);
    if (!rc) rc = down(accept_sum, o.acc, s8);
    if (!rc) rc = down(n_steps, o.n, s4);
    if (!rc) rc = down(divergent, o.div, s4);
    if (!rc) rc = down(turning, o.turn, s4);
    if (!rc) rc = down(subtree_depth, o.depth, s4);
  }
  u.buf.release();
  outb.release();
  return rc;
}

// B2' (tree.ex:613-653): see leapfrog_chain_normal_kernel. Blocking, on the null stream, like the other
// handle-less seams: the reference's hook uploads, dispatches and downloads per call as well. The hook exists
// to cut the cost of a DISPATCH (tree.ex:613-619), so a call of ordinary size pays for no allocation: a scratch
// per device (device buffer + page-locked staging, kChainScratchBytes each, made on first use and kept), ONE
// packed upload (q, p, inv_mass) and ONE packed download (the three row sets and logp) -- and up to 1 MB of rows no
// copy at all (the kernel works on the staging buffer itself). Calls on a device are
// serialised by its scratch's lock (they share the null stream anyway). Larger batches allocate and free.
int exmc_hip_leapfrog_chain_normal_host(int device, int C, int d, const double* q, const double* p,
                                        const double* inv_mass, int k, double signed_eps, double mu,
                                        double sigma, double* q_chain, double* p_chain,
                                        double* grad_chain, double* logp_chain) {
  if (C < 1 || d < 1 || d > kChainNormalMaxD || k < 0 || !q || !p || !inv_mass)
    return fail(EXMC_ERR_BADARG, "leapfrog_chain_normal: need n_chains >= 1, 1 <= d <= 256, k >= 0 and q, p, inv_mass");
  int rc = select_device(device);
  if (rc) return rc;
  if (k == 0) return EXMC_OK;
  const size_t vec = (size_t)C * d, rows = (size_t)C * k * d, lps = (size_t)C * k;
  const size_t n_in = 2 * vec + (size_t)d, n_out = 3 * rows + lps;
  const bool cached = (n_in + n_out) * 8 <= kChainScratchBytes && device < kChainScratchDevices;
  ChainScratch* sc = cached ? &g_chain_scratch[device] : nullptr;
  std::unique_lock<std::mutex> lock;
  DevBuf buf;
  double* dbase = nullptr;
  double* hbase = nullptr;
  if (sc) {
    lock = std::unique_lock<std::mutex>(sc->mu);
    if (!sc->dev) HIP_TRY(hipMalloc(&sc->dev, kChainScratchBytes));
    if (!sc->host) HIP_TRY(hipHostMalloc(&sc->host, kChainScratchBytes, hipHostMallocDefault));
    dbase = (double*)sc->dev;
    hbase = (double*)sc->host;
  } else {
    rc = buf.ensure((n_in + n_out) * 8);
    if (rc) return rc;
    dbase = (double*)buf.p;
  }
  ChainNormalParams P{};
  P.q = dbase; P.p = dbase + vec; P.inv_mass = dbase + 2 * vec;
  P.d = d; P.k = k; P.n_chains = C;
  P.eps = signed_eps; P.mu = mu; P.sigma = sigma;
  P.tiny32 = f32r(1.0e-30);
  P.log2pi32 = f32r(std::log(f32r(2.0 * M_PI)));
  P.q_chain = dbase + n_in;
  P.p_chain = P.q_chain + rows;
  P.g_chain = P.p_chain + rows;
  P.logp_chain = P.g_chain + rows;
  hipError_t e;
  // a call of ordinary size (the reference's: one chain, K = 32) moves so little that the two copies cost more
  // than the kernel: it reads its inputs from, and writes its rows to, the page-locked staging directly (the
  // buffer is device-accessible; the kernel's end makes the rows visible to the host) -- no copy kernels at all
  const bool zero_copy = hbase && n_out * 8 <= kChainZeroCopyBytes;
  if (zero_copy) {
    std::memcpy(hbase, q, vec * 8);
    std::memcpy(hbase + vec, p, vec * 8);
    std::memcpy(hbase + 2 * vec, inv_mass, (size_t)d * 8);
    P.q = hbase; P.p = hbase + vec; P.inv_mass = hbase + 2 * vec;
    P.q_chain = hbase + n_in;
    P.p_chain = P.q_chain + rows;
    P.g_chain = P.p_chain + rows;
    P.logp_chain = P.g_chain + rows;
    hipLaunchKernelGGL(leapfrog_chain_normal_kernel, dim3((unsigned)C), dim3(64), 0, 0, P);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e != hipSuccess) return fail(EXMC_ERR_HIP, std::string("leapfrog_chain_normal: ") + hipGetErrorString(e));
    const double* o = hbase + n_in;
    if (q_chain) std::memcpy(q_chain, o, rows * 8);
    if (p_chain) std::memcpy(p_chain, o + rows, rows * 8);
    if (grad_chain) std::memcpy(grad_chain, o + 2 * rows, rows * 8);
    if (logp_chain) std::memcpy(logp_chain, o + 3 * rows, lps * 8);
    return EXMC_OK;
  }
  if (hbase) {
    std::memcpy(hbase, q, vec * 8);
    std::memcpy(hbase + vec, p, vec * 8);
    std::memcpy(hbase + 2 * vec, inv_mass, (size_t)d * 8);
    e = hipMemcpy(dbase, hbase, n_in * 8, hipMemcpyHostToDevice);
  } else {
    e = hipMemcpy(dbase, q, vec * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dbase + vec, p, vec * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dbase + 2 * vec, inv_mass, (size_t)d * 8, hipMemcpyHostToDevice);
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(leapfrog_chain_normal_kernel, dim3((unsigned)C), dim3(64), 0, 0, P);
    e = hipGetLastError();
  }
  if (hbase) {
    // the blocking copy waits for the kernel (same stream)
    if (e == hipSuccess) e = hipMemcpy(hbase + n_in, P.q_chain, n_out * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(EXMC_ERR_HIP, std::string("leapfrog_chain_normal: ") + hipGetErrorString(e));
    const double* o = hbase + n_in;
    if (q_chain) std::memcpy(q_chain, o, rows * 8);
    if (p_chain) std::memcpy(p_chain, o + rows, rows * 8);
    if (grad_chain) std::memcpy(grad_chain, o + 2 * rows, rows * 8);
    if (logp_chain) std::memcpy(logp_chain, o + 3 * rows, lps * 8);
    return EXMC_OK;
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) rc = fail(EXMC_ERR_HIP, std::string("leapfrog_chain_normal: ") + hipGetErrorString(e));
  if (!rc) rc = down(q_chain, P.q_chain, rows * 8);
  if (!rc) rc = down(p_chain, P.p_chain, rows * 8);
  if (!rc) rc = down(grad_chain, P.g_chain, rows * 8);
  if (!rc) rc = down(logp_chain, P.logp_chain, lps * 8);
  buf.release();
  return rc;
}

}  // extern "C"
