// capi_rtc.hip -- C ABI: caller-supplied dynamics in the batched AL-iLQR loop (plan LANE), compiled at run time.
//
// ALTROSolver::SetExplicitDynamics (altro_solver.cpp:68-81) takes two host callbacks (typedefs.hpp:31-53); a std::function
// cannot run on the device, so the batched path offers the device-side equivalent: the caller hands the SOURCE of the
// continuous dynamics and of its Jacobian -- hand-written HIP, two function templates -- and the library compiles its own
// lane-per-problem iLQR kernels (kernels/ilqr_lane.hip) around them with hiprtc, once per (source, n, m, element type) and
// process.  The kernels are the very ones the compiled-in models use (MODEL_PENDULUM, MODEL_BICYCLE ...): the pendulum
// supplied as source solves to the same bits as MODEL_PENDULUM (tests/test_gpu_user_model.py).  The library's own sources
// travel inside libaltro_hip.so (.incbin below); hiprtc is resolved with dlopen at first use, so the library keeps its one
// link-time dependency (the HIP runtime).  Whole solves of such a handle run on the launch-sequenced loop (the one-launch
// fused kernel is not compiled at run time: it is the library's longest compile).
#include "capi_internal.h"

#include <cctype>

#include <dlfcn.h>
#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <string>

using namespace altro_hip;
using namespace altro_hip::capi;

// ---- the sources hiprtc needs, embedded at build time (paths relative to the -I of csrc) ------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
#define ALTRO_EMBED(sym, path)                                                                                       \
  __asm__(".pushsection .rodata\n.global " #sym "\n" #sym ":\n.incbin \"" path "\"\n.byte 0\n.popsection\n"); \
  extern "C" const char sym[];
#else
#define ALTRO_EMBED(sym, path) extern "C" const char sym[];
#endif
ALTRO_EMBED(altro_rtc_src_rtc_compat, "rtc_compat.h")
ALTRO_EMBED(altro_rtc_src_fp_contract, "fp_contract.h")
ALTRO_EMBED(altro_rtc_src_models, "models.h")
ALTRO_EMBED(altro_rtc_src_linesearch, "linesearch_sm.h")
ALTRO_EMBED(altro_rtc_src_ilqr_types, "kernels/ilqr_types.h")
ALTRO_EMBED(altro_rtc_src_al_types, "kernels/al_types.h")
ALTRO_EMBED(altro_rtc_src_al_lane, "kernels/al_lane.hip")
ALTRO_EMBED(altro_rtc_src_tvlqr_lane, "kernels/tvlqr_lane.hip")
ALTRO_EMBED(altro_rtc_src_lane_body, "kernels/tvlqr_lane_body.inc")
ALTRO_EMBED(altro_rtc_src_quad_body, "kernels/tvlqr_quad_body.inc")
ALTRO_EMBED(altro_rtc_src_quad2_body, "kernels/tvlqr_quad2_body.inc")
ALTRO_EMBED(altro_rtc_src_ilqr_lane, "kernels/ilqr_lane.hip")
// ... and of the tile plan's row-layout kernels (a caller's model on plan MFMA16)
ALTRO_EMBED(altro_rtc_src_mfma16_layout, "kernels/mfma16_layout.h")
ALTRO_EMBED(altro_rtc_src_ilqr_mfma16, "kernels/ilqr_mfma16.hip")
ALTRO_EMBED(altro_rtc_src_merit2_dpp, "kernels/ilqr_merit2_dpp.hip")
ALTRO_EMBED(altro_rtc_src_tile_model, "kernels/ilqr_tile_model.hip")
// ... and of plan GENERIC's loop kernels (a caller's model on plans GENERIC / MFMA32)
ALTRO_EMBED(altro_rtc_src_generic_arrays, "kernels/generic_arrays.h")
ALTRO_EMBED(altro_rtc_src_ilqr_generic, "kernels/ilqr_generic.hip")
ALTRO_EMBED(altro_rtc_src_ilqr_row32, "kernels/ilqr_row32.hip")

namespace {

struct Hiprtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) CreateProgram = nullptr;
  decltype(&hiprtcCompileProgram) CompileProgram = nullptr;
  decltype(&hiprtcDestroyProgram) DestroyProgram = nullptr;
  decltype(&hiprtcAddNameExpression) AddNameExpression = nullptr;
  decltype(&hiprtcGetLoweredName) GetLoweredName = nullptr;
  decltype(&hiprtcGetCodeSize) GetCodeSize = nullptr;
  decltype(&hiprtcGetCode) GetCode = nullptr;
  decltype(&hiprtcGetProgramLogSize) GetProgramLogSize = nullptr;
  decltype(&hiprtcGetProgramLog) GetProgramLog = nullptr;
  decltype(&hiprtcGetErrorString) GetErrorString = nullptr;
};

int hiprtc_api(const Hiprtc** out) {
  static Hiprtc r;
  static std::string why = "symbols missing";
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = std::getenv("ALTRO_HIP_HIPRTC");
    void* lib = env ? dlopen(env, RTLD_NOW | RTLD_GLOBAL) : nullptr;
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);       // a copy the process already has
    for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"})
      if (!lib) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { const char* e = dlerror(); if (e) why = e; }
      }
    if (!lib) return;
    const char* missing = nullptr;
#define SYM(field, name) \
  do { r.field = (decltype(r.field))dlsym(lib, name); if (!r.field && !missing) missing = name; } while (0)
    SYM(CreateProgram, "hiprtcCreateProgram"); SYM(CompileProgram, "hiprtcCompileProgram"); SYM(DestroyProgram, "hiprtcDestroyProgram");
    SYM(AddNameExpression, "hiprtcAddNameExpression"); SYM(GetLoweredName, "hiprtcGetLoweredName"); SYM(GetCodeSize, "hiprtcGetCodeSize");
    SYM(GetCode, "hiprtcGetCode"); SYM(GetProgramLogSize, "hiprtcGetProgramLogSize"); SYM(GetProgramLog, "hiprtcGetProgramLog");
    SYM(GetErrorString, "hiprtcGetErrorString");
#undef SYM
    if (missing) why = std::string("symbol ") + missing + " not found in libhiprtc";
    else r.lib = lib;
  });
  if (!r.lib) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "hiprtc (libhiprtc.so) could not be loaded: %s", why.c_str());
  *out = &r;
  return 0;
}

// the kernels of the launch-sequenced loop that depend on (model, n, m, T): ilqr_launch_f64.hip's list, by name
const char* const kKernelExpr[RTC_NUM] = {
    "altro_hip::ilqr_rollout_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_accept_kernel<%d, %d, %s>",
    "altro_hip::ilqr_expand_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_merit_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_merit_roll_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_merit_point_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_merit_sum_kernel<altro_hip::MODEL_USER, %d, %d, %s>",
    "altro_hip::ilqr_spec_select_kernel<%d, %d, %s>",
    "altro_hip::ilqr_zero_residuals_kernel<%s>",
    "altro_hip::ilqr_stationarity_kernel<%d, %d, %s>",
    "altro_hip::ilqr_dual_update_kernel<%d, %d, %s>",
    "altro_hip::ilqr_shift_kernel<%d, %d, %s>",
};

// ck: IlqrArgs::cost_kind -- the kernels that read the cost record take it as their last template argument (kernels/ilqr_lane.hip)
std::string kernel_expr(int which, int n, int m, const char* T, int ck) {
  char buf[256];
  if (which == RTC_ZERO_RESIDUALS) std::snprintf(buf, sizeof(buf), kKernelExpr[which], T);
  else std::snprintf(buf, sizeof(buf), kKernelExpr[which], n, m, T);
  std::string e = buf;
  if (ck && (which == RTC_EXPAND || which == RTC_MERIT || which == RTC_MERIT_POINT)) e.insert(e.size() - 1, ", " + std::to_string(ck));
  return e;
}
// does `src` define a function of this name?  (the identifier followed by an opening parenthesis, outside // comments)
bool defines_function(const std::string& src, const char* name) {
  const size_t len = std::strlen(name);
  for (size_t p = src.find(name); p != std::string::npos; p = src.find(name, p + 1)) {
    if (p > 0 && (std::isalnum((unsigned char)src[p - 1]) || src[p - 1] == '_')) continue;
    size_t q = p + len;
    while (q < src.size() && std::isspace((unsigned char)src[q])) ++q;
    if (q >= src.size() || src[q] != '(') continue;
    const size_t line = src.rfind('\n', p);
    const size_t cmt = src.rfind("//", p);
    if (cmt != std::string::npos && (line == std::string::npos || cmt > line)) continue;   // inside a line comment
    return true;
  }
  return false;
}
bool source_has_constraints(const std::string& src) {
  return defines_function(src, "altro_user_constraint") && defines_function(src, "altro_user_constraint_jacobian");
}

}  // namespace

namespace altro_hip {
namespace capi {

struct RtcModule {
  hipModule_t module = nullptr;
  hipFunction_t fn[RTC_NUM] = {};
  int device = -1;
};

// Compile (or fetch) the module for (source, n, m, dtype) on the handle's device; nullptr + last_error on failure.
static int rtc_module_for(altro_hip_batch* h, const std::string& user_src, int ck, RtcModule** out) {
  *out = nullptr;
  static std::mutex mu;
  static std::map<std::string, RtcModule*> cache;
  const char* T = h->dtype == ALTRO_HIP_F64 ? "double" : "float";
  const std::string key = std::to_string(h->device) + "|" + std::to_string(h->n) + "|" + std::to_string(h->m) + "|" + T + "|" + std::to_string(ck) + "|" + user_src;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) { *out = it->second; return 0; }
  const Hiprtc* R;
  int rc = hiprtc_api(&R);
  if (rc) return rc;
  // the unit: the caller's two templates under contract(on) (like every device function the solve paths share), the
  // library's kernels, and explicit instantiations of the ones this shape needs
  std::string src = "#define ALTRO_HIP_USER_MODEL 1\n";
  if (source_has_constraints(user_src)) src += "#define ALTRO_HIP_USER_CONSTRAINTS 1\n";
  src += "#include \"rtc_compat.h\"\n#include \"fp_contract.h\"\nALTRO_FP_REGION_ON\n";
  src += "#line 1 \"user_model\"\n" + user_src + "\nALTRO_FP_REGION_END\n#include \"kernels/ilqr_lane.hip\"\nnamespace altro_hip {\n";
  for (int w = 0; w < RTC_NUM; ++w) {
    std::string e = kernel_expr(w, h->n, h->m, T, ck);
    e.erase(0, std::strlen("altro_hip::"));
    for (size_t p; (p = e.find("altro_hip::")) != std::string::npos;) e.erase(p, std::strlen("altro_hip::"));
    src += "template __global__ void " + e + "(IlqrArgs<" + T + ">);\n";
  }
  src += "}\n";
  const char* hdr_src[] = {altro_rtc_src_rtc_compat, altro_rtc_src_fp_contract, altro_rtc_src_models, altro_rtc_src_linesearch,
                           altro_rtc_src_ilqr_types, altro_rtc_src_al_types, altro_rtc_src_al_lane, altro_rtc_src_tvlqr_lane,
                           altro_rtc_src_lane_body, altro_rtc_src_quad_body, altro_rtc_src_quad2_body, altro_rtc_src_ilqr_lane};
  const char* hdr_name[] = {"rtc_compat.h", "fp_contract.h", "models.h", "linesearch_sm.h", "kernels/ilqr_types.h", "kernels/al_types.h",
                            "kernels/al_lane.hip", "kernels/tvlqr_lane.hip", "kernels/tvlqr_lane_body.inc", "kernels/tvlqr_quad_body.inc",
                            "kernels/tvlqr_quad2_body.inc", "kernels/ilqr_lane.hip"};
  hiprtcProgram prog = nullptr;
  hiprtcResult rr = R->CreateProgram(&prog, src.c_str(), "altro_user_model.hip", 12, hdr_src, hdr_name);
  if (rr != HIPRTC_SUCCESS) return fail(ALTRO_HIP_ERR_HIP, "hiprtcCreateProgram: %s", R->GetErrorString(rr));
  std::vector<std::string> exprs;
  for (int w = 0; w < RTC_NUM; ++w) {
    exprs.push_back(kernel_expr(w, h->n, h->m, T, ck));
    R->AddNameExpression(prog, exprs.back().c_str());
  }
  hipDeviceProp_t prop;
  std::string arch = "--offload-arch=gfx950";
  if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) arch = std::string("--offload-arch=") + prop.gcnArchName;
  const char* opts[] = {arch.c_str(), "-O3", "-std=c++17"};
  rr = R->CompileProgram(prog, 3, opts);
  if (rr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    R->GetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) R->GetProgramLog(prog, &log[0]);
    if (log.size() > 1800) log.resize(1800);
    R->DestroyProgram(&prog);
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "the model source does not compile (hiprtc: %s):\n%s", R->GetErrorString(rr), log.c_str());
  }
  size_t cs = 0;
  R->GetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  R->GetCode(prog, code.data());
  RtcModule* m = new RtcModule();
  m->device = h->device;
  if (const hipError_t le = hipModuleLoadData(&m->module, code.data()); le != hipSuccess) {
    R->DestroyProgram(&prog);
    delete m;
    return fail(ALTRO_HIP_ERR_HIP, "hipModuleLoadData of the compiled model failed: %s", hipGetErrorString(le));
  }
  for (int w = 0; w < RTC_NUM; ++w) {
    const char* lowered = nullptr;
    if (R->GetLoweredName(prog, exprs[w].c_str(), &lowered) != HIPRTC_SUCCESS || !lowered ||
        hipModuleGetFunction(&m->fn[w], m->module, lowered) != hipSuccess) {
      rc = fail(ALTRO_HIP_ERR_HIP, "kernel %s missing from the compiled model", exprs[w].c_str());
      R->DestroyProgram(&prog);
      (void)hipModuleUnload(m->module);
      delete m;
      return rc;
    }
  }
  R->DestroyProgram(&prog);
  cache[key] = m;
  *out = m;
  return 0;
}

// ---- the same for plan MFMA16: the caller's model inside the tile plan's row-layout kernels (kernels/ilqr_tile_model.hip) ------
// One module per (source, n, m, constraint blocks?, dense cost?): the rollout, the dynamics expansion and the two merit kernels
// (line-search round / two-trial pass) of that combination -- four kernels instead of ten, the merit kernels being the library's
// heaviest compiles.
enum RtcTileKernel { RTT_ROLLOUT = 0, RTT_EXPAND_DYN, RTT_MERIT, RTT_MERIT2, RTT_NUM };
struct RtcTileModule {
  hipModule_t module = nullptr;
  hipFunction_t fn[RTT_NUM] = {};
};
static int rtc_tile_module_for(altro_hip_batch* h, int al, int dense, RtcTileModule** out) {
  *out = nullptr;
  static std::mutex mu;
  static std::map<std::string, RtcTileModule*> cache;
  const std::string& user_src = h->rtc_source;
  const std::string key = std::to_string(h->device) + "|" + std::to_string(h->n) + "|" + std::to_string(h->m) + "|" + std::to_string(al) + "|" +
                          std::to_string(dense) + "|" + user_src;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) { *out = it->second; return 0; }
  const Hiprtc* R;
  int rc = hiprtc_api(&R);
  if (rc) return rc;
  const char* B_[2] = {"false", "true"};
  std::string exprs[RTT_NUM];
  exprs[RTT_ROLLOUT] = "altro_hip::wave_rollout_model_kernel<double, altro_hip::MODEL_USER>";
  exprs[RTT_EXPAND_DYN] = "altro_hip::wave_expand_dyn_kernel<double, altro_hip::MODEL_USER>";
  exprs[RTT_MERIT] = std::string("altro_hip::wave_merit_dpp_kernel<double, ") + B_[al] + ", false, " + B_[dense] + ", altro_hip::MODEL_USER>";
  exprs[RTT_MERIT2] = std::string("altro_hip::wave_merit_dpp_kernel<double, ") + B_[al] + ", true, " + B_[dense] + ", altro_hip::MODEL_USER>";
  std::string src = "#define ALTRO_HIP_USER_MODEL 1\n#define ALTRO_HIP_TILE_N " + std::to_string(h->n) + "\n#define ALTRO_HIP_TILE_M " +
                    std::to_string(h->m) + "\n";
  src += "#include \"rtc_compat.h\"\n#include \"fp_contract.h\"\nALTRO_FP_REGION_ON\n";
  src += "#line 1 \"user_model\"\n" + user_src + "\nALTRO_FP_REGION_END\n#include \"kernels/ilqr_mfma16.hip\"\n#include \"kernels/ilqr_merit2_dpp.hip\"\n"
         "namespace altro_hip {\n";
  for (int w = 0; w < RTT_NUM; ++w) {
    std::string e = exprs[w];
    for (size_t p; (p = e.find("altro_hip::")) != std::string::npos;) e.erase(p, std::strlen("altro_hip::"));
    src += "template __global__ void " + e + "(IlqrWaveArgs<double>);\n";
  }
  src += "}\n";
  const char* hdr_src[] = {altro_rtc_src_rtc_compat, altro_rtc_src_fp_contract, altro_rtc_src_models, altro_rtc_src_linesearch,
                           altro_rtc_src_ilqr_types, altro_rtc_src_al_types, altro_rtc_src_al_lane, altro_rtc_src_tvlqr_lane,
                           altro_rtc_src_lane_body, altro_rtc_src_quad_body, altro_rtc_src_quad2_body, altro_rtc_src_mfma16_layout,
                           altro_rtc_src_ilqr_mfma16, altro_rtc_src_merit2_dpp, altro_rtc_src_tile_model};
  const char* hdr_name[] = {"rtc_compat.h", "fp_contract.h", "models.h", "linesearch_sm.h", "kernels/ilqr_types.h", "kernels/al_types.h",
                            "kernels/al_lane.hip", "kernels/tvlqr_lane.hip", "kernels/tvlqr_lane_body.inc", "kernels/tvlqr_quad_body.inc",
                            "kernels/tvlqr_quad2_body.inc", "kernels/mfma16_layout.h", "kernels/ilqr_mfma16.hip",
                            "kernels/ilqr_merit2_dpp.hip", "kernels/ilqr_tile_model.hip"};
  hiprtcProgram prog = nullptr;
  hiprtcResult rr = R->CreateProgram(&prog, src.c_str(), "altro_user_tile_model.hip", 15, hdr_src, hdr_name);
  if (rr != HIPRTC_SUCCESS) return fail(ALTRO_HIP_ERR_HIP, "hiprtcCreateProgram: %s", R->GetErrorString(rr));
  for (int w = 0; w < RTT_NUM; ++w) R->AddNameExpression(prog, exprs[w].c_str());
  hipDeviceProp_t prop;
  std::string arch = "--offload-arch=gfx950";
  if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) arch = std::string("--offload-arch=") + prop.gcnArchName;
  // (-unroll-threshold: tile_model_step finds the Jacobian's structural zeros with __builtin_constant_p, resolved right after the
  //  compiler's early full-unroll pass -- the caller's zero fill must be unrolled by then, or all 192 entries count as nonzeros: the
  //  merit kernels then spill 240-330 registers to scratch; see rtc_gen_module_for)
  const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-mllvm", "-unroll-threshold=5000"};
  rr = R->CompileProgram(prog, 5, opts);
  if (rr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    R->GetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) R->GetProgramLog(prog, &log[0]);
    if (log.size() > 1800) log.resize(1800);
    R->DestroyProgram(&prog);
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "the model source does not compile for the tile plan (hiprtc: %s):\n%s", R->GetErrorString(rr), log.c_str());
  }
  size_t cs = 0;
  R->GetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  R->GetCode(prog, code.data());
  RtcTileModule* m = new RtcTileModule();
  if (const hipError_t le = hipModuleLoadData(&m->module, code.data()); le != hipSuccess) {
    R->DestroyProgram(&prog);
    delete m;
    return fail(ALTRO_HIP_ERR_HIP, "hipModuleLoadData of the compiled tile model failed: %s", hipGetErrorString(le));
  }
  for (int w = 0; w < RTT_NUM; ++w) {
    const char* lowered = nullptr;
    if (R->GetLoweredName(prog, exprs[w].c_str(), &lowered) != HIPRTC_SUCCESS || !lowered ||
        hipModuleGetFunction(&m->fn[w], m->module, lowered) != hipSuccess) {
      rc = fail(ALTRO_HIP_ERR_HIP, "kernel %s missing from the compiled tile model", exprs[w].c_str());
      R->DestroyProgram(&prog);
      (void)hipModuleUnload(m->module);
      delete m;
      return rc;
    }
  }
  R->DestroyProgram(&prog);
  cache[key] = m;
  *out = m;
  return 0;
}

// ---- the same for plans GENERIC / MFMA32: the caller's model inside that plan's loop kernels (kernels/ilqr_generic.hip) ----------------
// Three kernels per (source, n, m): the open-loop rollout, the dynamics expansion and the merit evaluation.
// On plan MFMA32's shapes three more: the merit kernel, its two-trial pass and the dynamics expansion in the row layout
// (kernels/ilqr_row32.hip: every lane evaluates the caller's model; r32_model_step).  They are USED when they compiled without scratch
// memory (row_ok): a Jacobian the compiler cannot keep in registers -- a dense one of many states, a loop it cannot unroll -- makes
// that formulation spill, and a spilling wave of that kernel waits for its reloads behind the prefetch (3.4 x slower
// than without); the wave-per-problem kernels above are the form for such models.
enum RtcGenKernel { RTG_ROLLOUT = 0, RTG_EXPAND_DYN, RTG_MERIT, RTG_NUM, RTG_ROW_MERIT = RTG_NUM, RTG_ROW_MERIT2, RTG_ROW_EXPAND_DYN, RTG_NUM_ROW };
struct RtcGenModule {
  hipModule_t module = nullptr;
  hipFunction_t fn[RTG_NUM_ROW] = {};
  bool row_ok = false;
};
constexpr int RTG_ROW_SCRATCH_MAX = 256;   // bytes of private segment per lane a row-layout kernel may use and still be chosen
static int rtc_gen_module_for(altro_hip_batch* h, RtcGenModule** out) {
  *out = nullptr;
  static std::mutex mu;
  static std::map<std::string, RtcGenModule*> cache;
  const std::string& user_src = h->rtc_source;
  const std::string key = std::to_string(h->device) + "|" + std::to_string(h->n) + "|" + std::to_string(h->m) + "|" + user_src;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(key);
  if (it != cache.end()) { *out = it->second; return 0; }
  const Hiprtc* R;
  int rc = hiprtc_api(&R);
  if (rc) return rc;
  const std::string nm = std::to_string(h->n) + ", " + std::to_string(h->m);
  const bool row_shape = tile32_supported(h->n, h->m);
  const int nk = row_shape ? (int)RTG_NUM_ROW : (int)RTG_NUM;
  std::string exprs[RTG_NUM_ROW];
  exprs[RTG_ROW_MERIT] = "altro_hip::row32_merit_kernel<double, " + nm + ", 1, false, altro_hip::MODEL_USER>";
  exprs[RTG_ROW_MERIT2] = "altro_hip::row32_merit_kernel<double, " + nm + ", 1, true, altro_hip::MODEL_USER>";
  exprs[RTG_ROW_EXPAND_DYN] = "altro_hip::row32_expand_dyn_kernel<double, " + nm + ", altro_hip::MODEL_USER>";
  exprs[RTG_ROLLOUT] = "altro_hip::generic_model_rollout_kernel<double, altro_hip::MODEL_USER, " + nm + ">";
  exprs[RTG_EXPAND_DYN] = "altro_hip::generic_model_expand_dyn_kernel<double, altro_hip::MODEL_USER, " + nm + ">";
  exprs[RTG_MERIT] = "altro_hip::generic_merit_kernel<double, false, altro_hip::MODEL_USER, " + nm + ">";
  std::string src = "#define ALTRO_HIP_USER_MODEL 1\n";
  src += "#include \"rtc_compat.h\"\n#include \"fp_contract.h\"\nALTRO_FP_REGION_ON\n";
  src += "#line 1 \"user_model\"\n" + user_src + "\nALTRO_FP_REGION_END\n#include \"kernels/ilqr_generic.hip\"\n";
  if (row_shape) src += "#include \"kernels/ilqr_row32.hip\"\n";
  src += "namespace altro_hip {\n";
  for (int w = 0; w < nk; ++w) {
    std::string e = exprs[w];
    for (size_t p; (p = e.find("altro_hip::")) != std::string::npos;) e.erase(p, std::strlen("altro_hip::"));
    src += "template __global__ void " + e + "(IlqrGenArgs<double>);\n";
  }
  src += "}\n";
  const char* hdr_src[] = {altro_rtc_src_rtc_compat, altro_rtc_src_fp_contract, altro_rtc_src_models, altro_rtc_src_linesearch,
                           altro_rtc_src_ilqr_types, altro_rtc_src_al_types, altro_rtc_src_al_lane, altro_rtc_src_tvlqr_lane,
                           altro_rtc_src_lane_body, altro_rtc_src_quad_body, altro_rtc_src_quad2_body, altro_rtc_src_generic_arrays,
                           altro_rtc_src_ilqr_generic, altro_rtc_src_ilqr_row32};
  const char* hdr_name[] = {"rtc_compat.h", "fp_contract.h", "models.h", "linesearch_sm.h", "kernels/ilqr_types.h", "kernels/al_types.h",
                            "kernels/al_lane.hip", "kernels/tvlqr_lane.hip", "kernels/tvlqr_lane_body.inc", "kernels/tvlqr_quad_body.inc",
                            "kernels/tvlqr_quad2_body.inc", "kernels/generic_arrays.h", "kernels/ilqr_generic.hip", "kernels/ilqr_row32.hip"};
  hiprtcProgram prog = nullptr;
  hiprtcResult rr = R->CreateProgram(&prog, src.c_str(), "altro_user_generic_model.hip", 14, hdr_src, hdr_name);
  if (rr != HIPRTC_SUCCESS) return fail(ALTRO_HIP_ERR_HIP, "hiprtcCreateProgram: %s", R->GetErrorString(rr));
  for (int w = 0; w < nk; ++w) R->AddNameExpression(prog, exprs[w].c_str());
  hipDeviceProp_t prop;
  std::string arch = "--offload-arch=gfx950";
  if (hipGetDeviceProperties(&prop, h->device) == hipSuccess) arch = std::string("--offload-arch=") + prop.gcnArchName;
  // (-unroll-threshold: the row-layout kernels find a Jacobian's structural zeros with __builtin_constant_p, which the compiler
  //  resolves right after its EARLY full-unroll pass -- a caller's `for (e < n (n + m)) J[e] = 0` must be unrolled by then, and at
  //  the default threshold it is only unrolled later: every entry then counts as a nonzero and the kernel spills 585 registers)
  const char* opts[] = {arch.c_str(), "-O3", "-std=c++17", "-mllvm", "-unroll-threshold=5000"};
  rr = R->CompileProgram(prog, row_shape ? 5 : 3, opts);
  if (rr != HIPRTC_SUCCESS) {
    size_t ls = 0;
    R->GetProgramLogSize(prog, &ls);
    std::string log(ls, '\0');
    if (ls) R->GetProgramLog(prog, &log[0]);
    if (log.size() > 1800) log.resize(1800);
    R->DestroyProgram(&prog);
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "the model source does not compile for plan GENERIC's loop (hiprtc: %s):\n%s", R->GetErrorString(rr), log.c_str());
  }
  size_t cs = 0;
  R->GetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  R->GetCode(prog, code.data());
  RtcGenModule* m = new RtcGenModule();
  if (const hipError_t le = hipModuleLoadData(&m->module, code.data()); le != hipSuccess) {
    R->DestroyProgram(&prog);
    delete m;
    return fail(ALTRO_HIP_ERR_HIP, "hipModuleLoadData of the compiled model failed: %s", hipGetErrorString(le));
  }
  for (int w = 0; w < nk; ++w) {
    const char* lowered = nullptr;
    if (R->GetLoweredName(prog, exprs[w].c_str(), &lowered) != HIPRTC_SUCCESS || !lowered ||
        hipModuleGetFunction(&m->fn[w], m->module, lowered) != hipSuccess) {
      rc = fail(ALTRO_HIP_ERR_HIP, "kernel %s missing from the compiled model", exprs[w].c_str());
      R->DestroyProgram(&prog);
      (void)hipModuleUnload(m->module);
      delete m;
      return rc;
    }
  }
  R->DestroyProgram(&prog);
  if (row_shape) {   // the row-layout kernels are chosen when none of them spills (see RtcGenKernel)
    m->row_ok = true;
    for (int w = RTG_ROW_MERIT; w < RTG_NUM_ROW; ++w) {
      int scratch = 0;
      if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, m->fn[w]) != hipSuccess) { (void)hipGetLastError(); scratch = 1 << 30; }
      if (scratch > RTG_ROW_SCRATCH_MAX) m->row_ok = false;
    }
  }
  cache[key] = m;
  *out = m;
  return 0;
}
// A model kernel of plan GENERIC's loop from the handle's run-time module: the grids of ilqr_launch_generic.hip.  (IK_EXPAND: the
// dynamics expansion only -- the caller has launched the cost's.)
int rtc_gen_launch(altro_hip_batch* h, int which, const IlqrGenArgs<double>& a) {
  if (h->rtc_source.empty()) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model_source has not been called");
  RtcGenModule* mod = nullptr;
  int rc = rtc_gen_module_for(h, &mod);
  if (rc) return rc;
  IlqrGenArgs<double> args = a;
  void* params[] = {&args};
  auto go = [&](int w, unsigned gx, unsigned lds) -> int {
    const hipError_t e = hipModuleLaunchKernel(mod->fn[w], gx, 1, 1, 64, 1, 1, lds, h->stream, params, nullptr);
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "launch of the run-time compiled kernel %d of plan GENERIC's loop failed: %s", w, hipGetErrorString(e));
    return 0;
  };
  const bool row = a.row32m && mod->row_ok;   // (the host set row32m knowing row_ok: row32_model_eligible)
  switch (which) {
    case IK_ROLLOUT: return go(RTG_ROLLOUT, (unsigned)((a.batch + 63) / 64), 0);
    case IK_EXPAND:
      if (row) return go(RTG_ROW_EXPAND_DYN, (unsigned)(((int64_t)a.batch * a.N + 1) / 2), 0);
      return go(RTG_EXPAND_DYN, (unsigned)(((int64_t)a.batch * a.N + 63) / 64), 0);
    case IK_MERIT2:
      if (row) return go(RTG_ROW_MERIT2, (unsigned)a.batch, 0);
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "the two-trial pass has no run-time compiled kernel for this model");
    case IK_MERIT:
      if (row) return go(RTG_ROW_MERIT, (unsigned)((a.batch + 1) / 2), 0);
      return go(RTG_MERIT, (unsigned)a.batch, a.al.enabled ? (unsigned)(GEN_AL_JV * sizeof(double)) : 0u);
    default: return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d has no run-time compiled kernel on this plan", which);
  }
}

// A model kernel of plan MFMA16's loop from the handle's run-time module: the grids of ilqr_launch_mfma16_model.hip.
int rtc_tile_launch(altro_hip_batch* h, int which, const IlqrWaveArgs<double>& a) {
  if (h->rtc_source.empty()) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model_source has not been called");
  RtcTileModule* mod = nullptr;
  int rc = rtc_tile_module_for(h, a.al.enabled ? 1 : 0, a.cost_dense ? 1 : 0, &mod);
  if (rc) return rc;
  IlqrWaveArgs<double> args = a;
  void* params[] = {&args};
  const unsigned gsh = a.al.enabled ? (unsigned)a.al.Gpad_count * 8u : 0u;   // the merit kernels' dynamic LDS (padded constraint Jacobians)
  auto go = [&](int w, unsigned gx, unsigned gy) -> int {
    const hipError_t e = hipModuleLaunchKernel(mod->fn[w], gx, gy, 1, 64, 1, 1, (w == RTT_MERIT || w == RTT_MERIT2) ? gsh : 0u, h->stream, params, nullptr);
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "launch of the run-time compiled tile kernel %d failed: %s", w, hipGetErrorString(e));
    return 0;
  };
  const unsigned pairs = (unsigned)mf_grid((a.batch + 1) / 2);
  switch (which) {
    case IK_ROLLOUT: return go(RTT_ROLLOUT, (unsigned)((a.batch + 3) / 4), 1);
    case IK_EXPAND: return go(RTT_EXPAND_DYN, (unsigned)((int64_t)((a.batch + 3) / 4) * a.N), 1);
    case IK_MERIT: return go(RTT_MERIT, pairs, (unsigned)(((a.spec_trials > 1 ? a.spec_trials : 1) + 1) / 2));
    case IK_MERIT2: return go(RTT_MERIT2, pairs, 1);
    default: return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d has no run-time compiled tile kernel", which);
  }
}

// One kernel of the launch-sequenced loop from the handle's run-time module: the grids of ilqr_launch_kernel (ilqr_launch_f64.hip).
template <typename T>
int rtc_launch(altro_hip_batch* h, int which, const IlqrArgs<T>& a) {
  if (!h->rtc) return fail(ALTRO_HIP_ERR_NOT_SET, "altro_hip_set_model_source has not been called");
  if (h->rtc_ck != a.cost_kind) {   // the handle's cost changed kind since the module was built: the instantiations for this one
    RtcModule* m2 = nullptr;
    int rc = rtc_module_for(h, h->rtc_source, a.cost_kind, &m2);
    if (rc) return rc;
    h->rtc = m2; h->rtc_ck = a.cost_kind;
  }
  RtcModule* mod = (RtcModule*)h->rtc;
  IlqrArgs<T> args = a;
  void* params[] = {&args};
  const unsigned lanes = (unsigned)((a.batch + 63) / 64);
  const int64_t total = (int64_t)a.batch * (a.N + 1);
  const unsigned flat = (unsigned)std::min<int64_t>((total + 255) / 256, 1 << 20);
  const unsigned flat64 = (unsigned)std::min<int64_t>((total + 63) / 64, 1 << 20);
  const unsigned trials = a.spec_trials > 1 ? a.spec_trials : 1;
  auto go = [&](int w, unsigned gx, unsigned gy, unsigned bx) -> int {
    const hipError_t e = hipModuleLaunchKernel(mod->fn[w], gx, gy, 1, bx, 1, 1, 0, h->stream, params, nullptr);
    if (e != hipSuccess) return fail(ALTRO_HIP_ERR_HIP, "launch of %s failed: %s", kKernelExpr[w], hipGetErrorString(e));
    return 0;
  };
  switch (which) {
    case IK_ROLLOUT: return go(RTC_ROLLOUT, lanes, 1, 64);
    case IK_ACCEPT: return go(RTC_ACCEPT, flat, 1, 256);
    case IK_EXPAND: return go(RTC_EXPAND, flat64, 1, 64);
    case IK_MERIT:
      if (a.merit_jk) {
        int rc = go(RTC_MERIT_ROLL, lanes, trials, 64);
        if (!rc) rc = go(RTC_MERIT_POINT, flat64, trials, 64);
        if (!rc) rc = go(RTC_MERIT_SUM, lanes, trials, 64);
        return rc;
      }
      return go(RTC_MERIT, lanes, trials, 64);
    case IK_SPEC_SELECT: return go(RTC_SPEC_SELECT, flat, 1, 256);
    case IK_DUAL: return go(RTC_DUAL, flat, 1, 256);
    case IK_SHIFT: return go(RTC_SHIFT, (unsigned)std::min<int64_t>(((int64_t)a.batch * (h->n + h->m) + 255) / 256, 1 << 20), 1, 256);
    case IK_STATIONARITY: {
      int rc = go(RTC_ZERO_RESIDUALS, (unsigned)((a.batch + 255) / 256), 1, 256);
      if (!rc) rc = go(RTC_STATIONARITY, flat64, 1, 64);
      return rc;
    }
    default: return fail(ALTRO_HIP_ERR_UNSUPPORTED, "operation %d has no run-time compiled kernel", which);
  }
}
template int rtc_launch<double>(altro_hip_batch*, int, const IlqrArgs<double>&);
template int rtc_launch<float>(altro_hip_batch*, int, const IlqrArgs<float>&);

}  // namespace capi
}  // namespace altro_hip

extern "C" {

int altro_hip_set_model_source(altro_hip_batch* h, const char* source, float timestep) {
  int rc = loop_entry(h);
  if (rc) return rc;
  h->expansion_current = false;
  if (!source) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "source == NULL");
  if (!(timestep > 0.0f)) return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "time step must be positive (ErrorCodes::TimestepNotPositive)");
  HIP_TRY(hipSetDevice(h->device));
  if (h->plan == ALTRO_HIP_PLAN_GENERIC) {   // plans GENERIC / MFMA32: the caller's model inside that plan's loop kernels (any n, m <= 32)
    if (h->ragged) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models need uniform dimensions");
    if (h->dtype != ALTRO_HIP_F64) return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models on plans GENERIC / MFMA32 run on fp64 handles");
    if (h->n > 32 || h->m > 32)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "run-time compiled models on plan GENERIC take n, m <= 32 (got %d, %d)", h->n, h->m);
    const std::string gsrc(source);
    if (defines_function(gsrc, "altro_user_constraint") || defines_function(gsrc, "altro_user_constraint_jacobian"))
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "nonlinear constraint blocks from source are a plan LANE feature; this plan takes the model's two "
                                             "functions and linear blocks (altro_hip_add_linear_constraint)");
    h->rtc_source = gsrc;
    h->model = ModelParams{MODEL_USER, timestep, 0, 2.7, 1.5};
    RtcGenModule* gm = nullptr;   // compile now: a source that does not build must fail HERE, with the compiler's log
    if ((rc = rtc_gen_module_for(h, &gm))) { h->rtc_source.clear(); h->model = ModelParams{MODEL_LINEAR, 0.0f, 0, 2.7, 1.5}; return rc; }
    h->model_set = true; h->rtc_has_constraints = false;
    h->rtc_row32_ok = gm->row_ok;
    HIP_TRY(hipMemsetAsync(h->g_arr[G_f], 0, (size_t)h->batch * h->g_bstride[G_f] * h->esz, h->stream));
    HIP_TRY(hipMemsetAsync(h->g_arr[G_A], 0, (size_t)h->batch * h->g_bstride[G_A] * h->esz, h->stream));
    HIP_TRY(hipMemsetAsync(h->g_arr[G_B], 0, (size_t)h->batch * h->g_bstride[G_B] * h->esz, h->stream));
    h->dyn_set = true; h->has_f = 0;
    return 0;
  }
  if (h->plan != ALTRO_HIP_PLAN_LANE && h->plan != ALTRO_HIP_PLAN_MFMA16)
    return fail(ALTRO_HIP_ERR_UNSUPPORTED, "unknown plan %d", h->plan);
  if (h->plan == ALTRO_HIP_PLAN_MFMA16 && h->auto_plan && lane_supported(h->n, h->m)) {
    // nonlinear constraint blocks from source are plan LANE's: an ALTRO_HIP_PLAN_AUTO handle that rides the padded tile only because
    // of its batch size moves there while it is still empty
    const std::string probe(source);
    if ((defines_function(probe, "altro_user_constraint") || defines_function(probe, "altro_user_constraint_jacobian")) &&
        (rc = replan_empty_handle(h, ALTRO_HIP_PLAN_LANE))) return rc;
  }
  if (h->plan == ALTRO_HIP_PLAN_MFMA16) {   // the tile plan's row-layout kernels around the caller's model (kernels/ilqr_tile_model.hip)
    if (h->dtype != ALTRO_HIP_F64)
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "device models on plan MFMA16 run on fp64 records (create the handle with ALTRO_HIP_F64)");
    const std::string tsrc(source);
    if (defines_function(tsrc, "altro_user_constraint") || defines_function(tsrc, "altro_user_constraint_jacobian"))
      return fail(ALTRO_HIP_ERR_UNSUPPORTED, "nonlinear constraint blocks from source are a plan LANE feature; plan MFMA16 takes the model's "
                                             "two functions and linear blocks (altro_hip_add_linear_constraint)");
    h->rtc_source = tsrc;
    h->model = ModelParams{MODEL_USER, timestep, 0, 2.7, 1.5};
    RtcTileModule* tm = nullptr;   // compile now: a source that does not build must fail HERE, with the compiler's log
    if ((rc = rtc_tile_module_for(h, h->al_defs.empty() ? 0 : 1, h->cost_dense ? 1 : 0, &tm))) { h->rtc_source.clear(); return rc; }
    h->model_set = true; h->rtc_has_constraints = false;
    HIP_TRY(hipMemsetAsync(h->m_in, 0, (size_t)h->batch * h->N * MF_DYN * h->esz, h->stream));
    h->dyn_set = true; h->has_f = 0;
    return 0;
  }
  RtcModule* m = nullptr;
  const std::string src(source);
  const bool c_val = defines_function(src, "altro_user_constraint"), c_jac = defines_function(src, "altro_user_constraint_jacobian");
  if (c_val != c_jac)
    return fail(ALTRO_HIP_ERR_BAD_ARGUMENT, "the model source defines %s but not %s: constraint blocks need both",
                c_val ? "altro_user_constraint" : "altro_user_constraint_jacobian", c_val ? "altro_user_constraint_jacobian" : "altro_user_constraint");
  const int ck = h->cost_dense ? 1 : 0;
  if ((rc = rtc_module_for(h, src, ck, &m))) return rc;   // (altro_hip_last_error has the compiler's log)
  h->rtc = m; h->rtc_ck = ck; h->rtc_source = src;
  h->rtc_has_constraints = c_val && c_jac;
  h->model = ModelParams{MODEL_USER, timestep, 0, 2.7, 1.5};
  h->model_set = true;
  return 0;
}

int altro_hip_model_row_layout(const altro_hip_batch* h) { return (h && row32_model_eligible(h)) ? 1 : 0; }

}  // extern "C"
