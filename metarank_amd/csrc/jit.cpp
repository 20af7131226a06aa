// Run-time specialisation of the fused assembly kernel (pre-pass + per-item program -> scorer tile) for ONE model.
//
// The kernels of rank.hip interpret any model program from device memory: every op costs a scalar load of its
// descriptor, a switch, run-time column offsets and loop bounds, and the record loads of op k+1 cannot be issued
// before op k has finished.  A model's program is fixed when the config is loaded (FeatureMapping.fromFeatureSchema,
// FeatureMapping.scala:56-99 builds the feature list once per model), so the same device code (rank_device.hpp) is
// compiled once more with the program as compile-time constants: hiprtc, gfx950, the flags of the ahead-of-time
// build (-ffp-contract=off: parity is bit-exact).  The generic kernels stay the reference implementation of every
// op and the path for everything the specialised kernel does not cover (f64 matrix on demand, requests too large
// for one workgroup); tests require identical bytes from both (MRK_RANK_JIT=0 | 1).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "features.hpp"
#include "jit.hpp"
#include "runtime.hpp"

namespace mrk {

namespace {

// device_types.hpp + rank.hpp + qs_device.hpp + rank_device.hpp without their #include lines (written by the build:
// metarank_amd/_native.py::embed_jit_sources)
const char *const k_device_source =
#include "jit_embed.inc"
    ;

std::string f64_literal(double v) {
  if (v != v) return "__builtin_nan(\"\")";
  if (v == __builtin_inf()) return "__builtin_inf()";
  if (v == -__builtin_inf()) return "(-__builtin_inf())";
  char buf[64];
  snprintf(buf, sizeof buf, "%a", v);  // hexadecimal floating literal: exact
  return buf;
}

std::string col(const ColRef &c) { return "{" + std::to_string(c.tag) + "," + std::to_string(c.val) + "}"; }

uint64_t fnv1a(const std::string &s) {
  uint64_t h = 1469598103934665603ull;
  for (unsigned char c : s) h = (h ^ c) * 1099511628211ull;
  return h;
}

}  // namespace

std::string jit_source(const Program &prog) {
  std::string s;
  s.reserve(strlen(k_device_source) + 8192);
  s += k_device_source;
  s += "\nnamespace mrk {\nnamespace {\n";
  // tables as function-local constexpr arrays: a namespace-scope / static-member constexpr array is emitted as an
  // externally initialised device variable and its loads are NOT folded
  auto table = [&](const char *name, const char *type, size_t n, const std::string &rows) {
    s += std::string("struct ") + name + " {\n  __device__ __forceinline__ constexpr " + type + " operator[](int i) const {\n    constexpr " + type +
         " t[" + std::to_string(n ? n : 1) + "] = {" + rows + "};\n    return t[i];\n  }\n};\n";
  };
  std::string rows;
  for (const Op &o : prog.ops) {
    rows += "\n      {" + std::to_string(o.kind) + "," + std::to_string(o.dst) + "," + std::to_string(o.dim) + "," + std::to_string(o.scope) + "," +
            col(o.c0) + "," + col(o.c1) + "," + col(o.c2) + "," + col(o.c3) + "," + col(o.c4) + "," + col(o.c5) + "," + std::to_string(o.i0) + "," +
            std::to_string(o.i1) + "," + std::to_string(o.i2) + "," + std::to_string(o.i3) + "," + f64_literal(o.d0) + "},";
  }
  table("JitOps", "Op", prog.ops.size(), rows);
  rows.clear();
  for (const PrepEntry &p : prog.prep)
    rows += "\n      {" + std::to_string(p.kind) + "," + col(p.item_col) + "," + std::to_string(p.list_scope) + "," + col(p.list_col) + "," +
            std::to_string(p.top) + ",0},";
  table("JitPrep", "PrepEntry", prog.prep.size(), rows);
  rows.clear();
  for (uint32_t a : prog.aux) rows += std::to_string(a) + "u,";
  table("JitAux", "uint32_t", prog.aux.size(), rows);
  s += "struct JitProg {\n  static constexpr bool is_static = true;\n";
  s += "  static constexpr int32_t n_ops = " + std::to_string(prog.ops.size()) + ", n_prep = " + std::to_string(prog.prep.size()) +
       ", dim = " + std::to_string(prog.dim) + ", n_consts = " + std::to_string(prog.n_consts) + ";\n";
  s += "  JitOps ops;\n  JitPrep prep;\n  JitAux aux;\n};\n}  // namespace\n}  // namespace mrk\n\n";
  for (int f64 = 0; f64 < 2; ++f64) {
    s += std::string("extern \"C\" __global__ void __launch_bounds__(256)\n") + (f64 ? "mrk_jit_rank_cells_f64" : "mrk_jit_rank_cells_f32") +
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, mrk::QsDev q, uint16_t *cells) {\n"
         "  mrk::rank_fused_cells_body<" + (f64 ? "true" : "false") + ">(st, mrk::JitProg{}, b, tab_entries, vals_cap, q, cells);\n}\n";
  }
  return s;
}

std::vector<char> jit_compile(const std::string &source, std::string &log) {
  hiprtcProgram p = nullptr;
  if (hiprtcCreateProgram(&p, source.c_str(), "mrk_rank_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    throw StatusError(MRK_ERR_DEVICE, "hiprtcCreateProgram failed");
#ifdef MRK_PHASE_CLOCKS
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DMRK_PHASE_CLOCKS"};
#else
  const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
#endif
  const hiprtcResult rc = hiprtcCompileProgram(p, (int)(sizeof opts / sizeof opts[0]), opts);
  size_t n = 0;
  if (hiprtcGetProgramLogSize(p, &n) == HIPRTC_SUCCESS && n > 1) {
    log.resize(n);
    (void)hiprtcGetProgramLog(p, &log[0]);
  }
  std::vector<char> code;
  if (rc == HIPRTC_SUCCESS && hiprtcGetCodeSize(p, &n) == HIPRTC_SUCCESS && n) {
    code.resize(n);
    if (hiprtcGetCode(p, code.data()) != HIPRTC_SUCCESS) code.clear();
  }
  (void)hiprtcDestroyProgram(&p);
  if (code.empty()) throw StatusError(MRK_ERR_DEVICE, "hiprtc could not compile the specialised assembly kernel: " + log.substr(0, 2000));
  return code;
}

struct JitKernels {
  hipModule_t mod = nullptr;
  hipFunction_t fn[2] = {nullptr, nullptr};  // [f64]
  uint64_t hash = 0;
};

int jit_mode() {  // 0 off, 1 on (fall back to the generic kernel with a warning if hiprtc fails), 2 required
  const char *e = getenv("MRK_RANK_JIT");
  if (!e) return 1;
  if (!strcmp(e, "require")) return 2;
  return atoi(e) != 0 ? 1 : 0;
}

// ctx->mu must be held (the program's cache slot is not otherwise protected)
void *jit_rank_function(const Program &prog, bool f64) {
  const int mode = jit_mode();
  if (mode == 0) return nullptr;
  if (prog.jit_failed && mode != 2) return nullptr;
  if (!prog.jit) {
    try {
      const std::string src = jit_source(prog);
      std::string log;
      const std::vector<char> code = jit_compile(src, log);
      auto k = std::make_unique<JitKernels>();
      k->hash = fnv1a(src);
      MRK_HIP(hipModuleLoadData(&k->mod, code.data()));
      MRK_HIP(hipModuleGetFunction(&k->fn[1], k->mod, "mrk_jit_rank_cells_f64"));
      MRK_HIP(hipModuleGetFunction(&k->fn[0], k->mod, "mrk_jit_rank_cells_f32"));
      prog.jit = k.release();
    } catch (const std::exception &e) {
      prog.jit_failed = true;
      if (mode == 2) throw;
      fprintf(stderr, "[mrk] specialised assembly kernel for model '%s' unavailable, using the generic kernel: %s\n", prog.model.c_str(), e.what());
      return nullptr;
    }
  }
  return (void *)((JitKernels *)prog.jit)->fn[f64 ? 1 : 0];
}

#ifdef MRK_PHASE_CLOCKS
// measurement builds: read-and-reset the phase clocks of the specialised kernel of `prog`
extern "C" int mrk_debug_phase_clocks(const Program *prog, unsigned long long *out64) {
  if (!prog || !prog->jit) return -1;
  hipDeviceptr_t p = nullptr;
  size_t bytes = 0;
  if (hipModuleGetGlobal(&p, &bytes, ((JitKernels *)prog->jit)->mod, "mrk_phase_clocks") != hipSuccess || bytes < 64 * 8) return -2;
  if (hipMemcpy(out64, (void *)p, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  (void)hipMemset((void *)p, 0, 64 * 8);
  return 0;
}
#endif

void jit_release(Program &prog) {
  if (!prog.jit) return;
  JitKernels *k = (JitKernels *)prog.jit;
  if (k->mod) (void)hipModuleUnload(k->mod);
  delete k;
  prog.jit = nullptr;
}

}  // namespace mrk
