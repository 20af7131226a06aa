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

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "features.hpp"
#include "forest.hpp"
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

std::string jit_source(const Program &prog, bool f64, int kernel, const QsSignature *sig) {
  std::string s;
  s.reserve(strlen(k_device_source) + 8192);
  // experiments: MRK_JIT_DEFINES="A=1 B=2" - macros the device code reads (MRK_PROBE_W, MRK_PRE_GROUP_BUDGET); part of the text, so of the cache key
  for (size_t at = 0; at < switches().jit_defines.size();) {
    size_t end = switches().jit_defines.find(' ', at);
    if (end == std::string::npos) end = switches().jit_defines.size();
    std::string d = switches().jit_defines.substr(at, end - at);
    at = end + 1;
    if (d.empty()) continue;
    const size_t eq = d.find('=');
    s += "#define " + (eq == std::string::npos ? d : d.substr(0, eq) + " " + d.substr(eq + 1)) + "\n";
  }
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
       ", dim = " + std::to_string(prog.dim) + ", n_consts = " + std::to_string(prog.n_consts) + ", item_fixed = " + std::to_string(switches().jit_record_regs ? prog.item_fixed : 0) + ";\n";
  s += "  JitOps ops;\n  JitPrep prep;\n  JitAux aux;\n};\n";
  // the forest's view signature (forest.hpp): the sinks of the kernels that write the scorer's tile hold it as constants;
  // without one (no bit-vector image, f64-matrix kernels, MRK_JIT_SIG=0) they read the column descriptors from memory
  std::string qs = "mrk::QsDyn";
  if (sig && sig->ok && kernel != JIT_MATRIX && kernel != JIT_PREPASS) {
    table("JitSigRows", "QsSig", sig->cols.size(), sig->text);
    s += "struct JitQs {\n  static constexpr bool is_static = true;\n  static constexpr int n_feats = " + std::to_string(sig->cols.size()) +
         ", n_views = " + std::to_string(sig->n_views) + ";\n  static constexpr uint32_t thr_cap = " + std::to_string(sig->thr_cap) +
         "u, thr_total = " + std::to_string(sig->thr_total) + "u, rt_total = " + std::to_string(sig->rt_total) +
         "u;\n  __device__ __forceinline__ constexpr QsSig operator[](int i) const { return JitSigRows{}[i]; }\n};\n";
    qs = "mrk::JitQs";
  }
  s += "}  // namespace\n}  // namespace mrk\n\n";
  // Wavefronts per SIMD the compiler must leave room for (register cap 512 / n).  With the candidate's record and the
  // second trip of every op in registers the kernel would take ~200 VGPRs = 2 wavefronts per SIMD; measured on c2
  // (profiles/r02_c): 2 -> 0.436 ms, 3 -> 0.351, 4 -> 0.286 - the kernel hides its trips to memory with resident
  // wavefronts, so 4 is asked for (128 VGPRs).  MRK_JIT_WAVES=n overrides (experiments).
  std::string attr;
  {
    const int w = switches().jit_waves >= 1 && switches().jit_waves <= 8 ? switches().jit_waves : 4;
    attr = " __attribute__((amdgpu_waves_per_eu(" + std::to_string(w) + ", " + std::to_string(w) + ")))";
  }
  // One kernel per translation unit when `kernel` names one (JIT_ALL: all of them - what tools/jit_inspect.py looks at): a
  // specialised kernel is ~150 KB of straight-line code and takes the compiler 10 - 20 s, so a model's first rank compiles
  // the ONE kernel its batch shape runs, not four.
  const std::string b64 = f64 ? "true" : "false";
  if (kernel == JIT_ALL || kernel == JIT_RANK)
    s += "extern \"C\" __global__ void __launch_bounds__(256)" + attr + "\nmrk_jit_rank_cells"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, mrk::QsDev q, uint16_t *cells, int mode) {\n"
         "  mrk::rank_fused_cells_body<" + b64 + ", false, " + qs + ">(st, mrk::JitProg{}, b, tab_entries, vals_cap, q, cells, mode);\n}\n";
  // the same for a handful of requests (mrk_rank) or few large ones: up to 512 lanes per workgroup, the item lanes in
  // op_split copies that share the program's ops between them (rank_device.hpp op_owner), `slices` workgroups per request
  if (kernel == JIT_ALL || kernel == JIT_SPLIT)
    s += "extern \"C\" __global__ void __launch_bounds__(512)" + attr + "\nmrk_jit_rank_cells_split"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, mrk::QsDev q, uint16_t *cells, int mode) {\n"
         "  mrk::rank_fused_cells_body<" + b64 + ", true, " + qs + ">(st, mrk::JitProg{}, b, tab_entries, vals_cap, q, cells, mode);\n}\n";
  // the same workgroup-per-request kernel writing the row-major f64 matrix (models scored by the tree walk, explain);
  // the matrix does not depend on the scorer's precision
  if ((kernel == JIT_ALL && f64) || kernel == JIT_MATRIX)
    s += "extern \"C\" __global__ void __launch_bounds__(256)" + attr + "\nmrk_jit_rank_matrix"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, int mode) {\n"
         "  mrk::rank_fused_matrix_body<false>(st, mrk::JitProg{}, b, tab_entries, vals_cap, mode);\n}\n";
  // the item-parallel form (requests too large for one workgroup: tables from a previous pre-pass launch, in HBM)
  if (kernel == JIT_ALL || kernel == JIT_ITEMS)
    s += "extern \"C\" __global__ void __launch_bounds__(256)" + attr + "\nmrk_jit_assemble_cells"
         "(mrk::StoreDev st, mrk::BatchDev b, mrk::QsDev q, uint16_t *cells, uint32_t lds_entries) {\n"
         "  mrk::assemble_cells_body<" + b64 + ", " + qs + ">(st, mrk::JitProg{}, b, q, cells, lds_entries);\n}\n";
  // ... and its persistent form with every threshold table resident in LDS (rank_device.hpp assemble_cells_rt_body): only with a signature
  if ((kernel == JIT_ALL || kernel == JIT_ITEMS_RT) && qs != "mrk::QsDyn" && jit_items_rt_applies(sig))
    s += "extern \"C\" __global__ void __launch_bounds__(512)" + attr + "\nmrk_jit_assemble_cells_rt"
         "(mrk::StoreDev st, mrk::BatchDev b, mrk::QsDev q, uint16_t *cells, uint32_t lds_entries) {\n"
         "  mrk::assemble_cells_rt_body<" + b64 + ", " + qs + ">(st, mrk::JitProg{}, b, q, cells, lds_entries);\n}\n";
  // pre-pass + assembly + forest + ordering of a small request in one launch (rank_device.hpp rank_one_body)
  if (kernel == JIT_ALL || kernel == JIT_ONE)
    // (a request's workgroup is 8 wavefronts = 2 per SIMD and a handful of them run at a time: nothing to gain from the
    //  128-register cap of the batch kernels)
    s += "extern \"C\" __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))\nmrk_jit_rank_one"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, mrk::QsDev q, mrk::QsForestDev f, int mode, mrk::OneOut out) {\n"
         "  mrk::rank_one_body<" + b64 + ", " + qs + ">(st, mrk::JitProg{}, b, tab_entries, vals_cap, q, f, mode, out);\n}\n";
  // full batches of small requests: assembly + forest + ordering in the request's workgroup (rank_fused_score_body)
  if (kernel == JIT_ALL || kernel == JIT_FUSED_SCORE)
    s += "extern \"C\" __global__ void __launch_bounds__(256)" + attr + "\nmrk_jit_rank_fused_score"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t tab_entries, int vals_cap, mrk::QsDev q, mrk::QsForestDev f, uint16_t *cells) {\n"
         "  mrk::rank_fused_score_body<" + b64 + ", " + qs + ">(st, mrk::JitProg{}, b, tab_entries, vals_cap, q, f, cells);\n}\n";
  // the pre-pass of requests too large for one workgroup's assembly (config 4: ONE workgroup per request builds the tables the
  // item-parallel kernel reads; on the critical path of the request, so no register cap)
  if ((kernel == JIT_ALL && f64) || kernel == JIT_PREPASS)
    s += "extern \"C\" __global__ void __launch_bounds__(256)\nmrk_jit_prepass"
         "(mrk::StoreDev st, mrk::BatchDev b, uint32_t lds_entries) {\n"
         "  mrk::prepass_body(st, mrk::JitProg{}, b, lds_entries);\n}\n";
  // ... and its persistent form (rank_device.hpp rank_serve_body)
  if (kernel == JIT_ALL || kernel == JIT_SERVE)
    s += "extern \"C\" __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))\nmrk_jit_rank_serve"
         "(mrk::StoreDev st, mrk::QsDev q, mrk::QsForestDev f, mrk::ServeGangDev gang) {\n"
         "  mrk::rank_serve_body<" + b64 + ", " + qs + ">(st, mrk::JitProg{}, q, f, gang);\n}\n";
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

// one specialised kernel = one module, built when a batch first needs it
struct JitSlot {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  bool failed = false;
  std::string key;              // the stem of its cache file name: hash of the translation unit, its size, the compiler (mrk_config_kernel_keys)
  bool skip_disk = false;       // a code object found on disk did not load on this device (another architecture / a stale file): compile instead
  bool not_on_disk = false;     // a no-compile lookup (the stand-in of a signature's kernel) found no file: not probed again on the request path
  // MRK_RANK_JIT=async: the code object is produced by a background thread while requests are served by the generic kernel
  std::thread worker;
  std::atomic<int> state{0};    // 0 idle, 1 compiling, 2 code ready, 3 failed
  std::vector<char> code;
  std::string error;
};
struct JitSlotSet {
  JitSlot slot[JIT_KERNELS][2];   // [kernel][f64] (the matrix kernel lives in [JIT_MATRIX][1])
};
// A program's kernels, per view signature of the forest they bin for (forest.hpp QsSignature; "" = none: the kernel reads
// the column descriptors from memory - the f64-matrix kernel, models without a bit-vector image, and what serves a
// retrained model whose signature changed while its own kernel compiles).  Sets are never dropped while the program lives
// (a module may have launches in flight); signatures change with the feature list or the missing-value kinds, not per
// retraining, so there are few.
struct JitKernels {
  std::map<std::string, std::unique_ptr<JitSlotSet>> by_sig;
};

const char *const JIT_KERNEL_NAME[JIT_KERNELS] = {"mrk_jit_rank_cells", "mrk_jit_rank_cells_split", "mrk_jit_rank_matrix", "mrk_jit_assemble_cells", "mrk_jit_rank_one", "mrk_jit_rank_serve", "mrk_jit_rank_fused_score", "mrk_jit_prepass", "mrk_jit_assemble_cells_rt"};
// a kernel that exists only with a forest signature: no program-only stand-in (the caller falls back to another KERNEL meanwhile)
static inline bool jit_needs_sig(int kernel) { return kernel == JIT_ITEMS_RT; }
// the resident-table kernel: every table in LDS (<= 64 KB of thresholds leaves room for a request's hash tables and a second workgroup)
bool jit_items_rt_applies(const QsSignature *sig) { return sig && sig->ok && sig->rt_total > 0 && (size_t)sig->rt_total * 8 <= 64 * 1024 && switches().jit_sig && switches().items_rt; }
// kernels that neither write the scorer's tile nor depend on the scorer's precision: one per program, kept in slot [kernel][1]
static inline bool jit_program_only(int kernel) { return kernel == JIT_MATRIX || kernel == JIT_PREPASS; }

// 0 off; 1 on: the first rank of a model waits for the compile (a failure falls back to the generic kernel with a warning);
// 2 required: a failure is an error; 3 async: compile in the background, rank with the generic kernel until it is ready;
// 4 auto (the default): a code object found on disk - the user's cache, or the directory shipped next to the library
// (built for the stock Ranklens program by __graft_entry__.build()) - is loaded at once, anything else compiles in the
// background while the generic kernel ranks: no request ever waits for the compiler (Serve.scala:130-150: warm-up happens
// before the port opens, never under a request)
int jit_mode() { return switches().jit_mode; }

namespace {

// code objects are kept on disk between processes: $MRK_JIT_CACHE_DIR, else $XDG_CACHE_HOME/mrk_jit, else
// ~/.cache/mrk_jit; the key covers the whole translation unit and the compiler's version
// the file name (no directory, no extension) of a translation unit's code object: its hash, its size, the compiler
std::string cache_stem(const std::string &src) {
  int major = 0, minor = 0;
  (void)hiprtcVersion(&major, &minor);
  char name[96];
#ifdef MRK_PHASE_CLOCKS
  const char *flavour = "-clk";  // the measurement build compiles the same text with -DMRK_PHASE_CLOCKS
#else
  const char *flavour = "";
#endif
  snprintf(name, sizeof name, "%016llx-%zu-rtc%d.%d-gfx950%s", (unsigned long long)fnv1a(src), src.size(), major, minor, flavour);
  return name;
}

std::string cache_path(const std::string &src) {
  const std::string &d = switches().jit_cache_dir;
  if (d.empty()) return "";
  return d + "/" + cache_stem(src) + ".co";
}

// <directory of libmrk_hip.so>/jit_cache: code objects built ahead of time (read-only; same file names as the user's cache)
std::string shipped_path(const std::string &src) {
  static const std::string dir = [] {
    Dl_info info;
    if (!dladdr((const void *)&jit_mode, &info) || !info.dli_fname) return std::string();
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/jit_cache";
  }();
  if (dir.empty() || !switches().jit_shipped) return "";
  return dir + "/" + cache_stem(src) + ".co";
}

std::vector<char> read_file(const std::string &path) {
  std::vector<char> out;
  if (path.empty()) return out;
  if (FILE *f = fopen(path.c_str(), "rb")) {
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    fclose(f);
  }
  return out;
}

void write_file(const std::string &path, const std::vector<char> &data) {
  if (path.empty()) return;
  const std::string dir = path.substr(0, path.rfind('/'));
  for (size_t i = 1; i <= dir.size(); ++i)
    if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0700);
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  if (FILE *f = fopen(tmp.c_str(), "wb")) {
    const bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
  }
}

}  // namespace

// Called WITHOUT the context's launch lock: the first batch of a shape compiles for seconds, and other batches keep
// launching meanwhile.  The program's own mutex serialises callers that want a kernel of the same program.
// sig: the forest's view signature (nullptr / !ok: none).  no_compile: only what is loaded or on disk (the stand-in while
// a signature's own kernel compiles in the background).
static void *jit_function_locked(const Program &prog, int kernel, bool f64, bool wait, const QsSignature *sig, bool no_compile) {
  const int mode = wait && jit_mode() >= 3 ? 1 : jit_mode();  // wait: a warm-up call - the one place that may wait for the compiler
  if (mode == 0) return nullptr;
  if (!prog.jit) prog.jit = new JitKernels();
  JitKernels *k = (JitKernels *)prog.jit;
  if (jit_program_only(kernel)) f64 = true;
  const bool keyed = sig && sig->ok && !jit_program_only(kernel) && switches().jit_sig;
  if (jit_needs_sig(kernel) && (!keyed || !jit_items_rt_applies(sig))) return nullptr;
  // (MRK_JIT_DEFINES is part of the translation unit: a process that flips it - tests, A/B scripts - gets the kernels of the text it asked for)
  std::unique_ptr<JitSlotSet> &set = k->by_sig[(keyed ? sig->text : std::string()) + (switches().jit_defines.empty() ? std::string() : "\n#" + switches().jit_defines)];
  if (!set) set.reset(new JitSlotSet());
  JitSlot &sl = set->slot[kernel][f64 ? 1 : 0];
  if (sl.fn) return (void *)sl.fn;
  if (no_compile && sl.not_on_disk) return nullptr;   // O(1) after the first miss: no translation unit rebuilt, no file probed per request
  // while this signature's kernel is not there (compiling, failed, its file unloadable): the program's signature-less
  // kernel, if it is loaded or on disk - never the interpreting kernel when a specialised one exists
  auto stand_in = [&]() -> void * { return keyed && !no_compile && !jit_needs_sig(kernel) ? jit_function_locked(prog, kernel, f64, false, nullptr, true) : nullptr; };
  if (sl.failed && mode != 2) return stand_in();
  // the lambdas run on a background thread too: they own a copy of the signature (the model may be freed meanwhile)
  const std::shared_ptr<const QsSignature> sg = keyed ? std::make_shared<const QsSignature>(*sig) : nullptr;
  auto cached = [&prog, f64, kernel, sg]() {   // the code object, if a previous process (or the build) left it on disk
    const std::string src = jit_source(prog, f64, kernel, sg.get());
    std::vector<char> code = read_file(cache_path(src));
    if (code.empty()) code = read_file(shipped_path(src));
    return code;
  };
  const bool skip_disk = sl.skip_disk;
  auto produce = [&prog, f64, kernel, skip_disk, sg]() {  // host only: no device call (safe on any thread)
    const std::string src = jit_source(prog, f64, kernel, sg.get());
    const std::string path = cache_path(src);
    std::vector<char> code;
    if (!skip_disk) code = read_file(path);
    if (code.empty() && !skip_disk) code = read_file(shipped_path(src));
    if (code.empty()) {
      std::string log;
      code = jit_compile(src, log);
      write_file(path, code);
    }
    return code;
  };
  try {
    std::vector<char> code;
    if ((mode == 4 || no_compile) && sl.state.load() == 0 && !sl.skip_disk) code = cached();
    const bool from_disk = !code.empty();
    if (from_disk) {
      // on disk: loaded below, at once
    } else if (no_compile && sl.state.load() == 0) {
      sl.not_on_disk = true;
      return nullptr;
    } else if (mode == 3 || mode == 4 || sl.state.load() != 0) {
      int st = sl.state.load();
      if (st == 0) {  // first sight of this (program, signature, kernel, precision): start the compile, keep ranking with what there is
        // The stand-ins FIRST, all of them: loading a code object while another thread is inside the compiler waits for it
        // (seen on the GPU box: the first mrk_rank of a model took 4.0 s - one compile - when the background thread reached
        // the compiler before the request's thread reached hipModuleLoadData; hiprtc / comgr serialise).  Once the
        // program-only kernels that are on disk are loaded, no request of this program touches the loader while its
        // signature's kernels compile.
        void *fallback = nullptr;
        if (keyed && !no_compile) {
          for (int kn = 0; kn < JIT_KERNELS; ++kn) {
            if (jit_needs_sig(kn)) continue;
            void *f = jit_function_locked(prog, kn, jit_program_only(kn) ? true : f64, false, nullptr, true);
            if (kn == kernel) fallback = f;
          }
        }
        sl.state.store(1);
        JitSlot *slot = &sl;
        sl.worker = std::thread([slot, produce]() {
          try {
            slot->code = produce();
            slot->state.store(2);
          } catch (const std::exception &e) {
            slot->error = e.what();
            slot->state.store(3);
          }
        });
        return fallback;
      }
      if (st == 1) {
        if (mode != 2 && mode != 1) return stand_in();  // still compiling
        sl.worker.join();                               // a synchronous mode took over: wait for it
        st = sl.state.load();
      }
      if (sl.worker.joinable()) sl.worker.join();
      if (st == 3) throw StatusError(MRK_ERR_DEVICE, sl.error);
      code.swap(sl.code);
    } else {
      code = produce();
    }
    if (from_disk && hipModuleLoadData(&sl.mod, code.data()) != hipSuccess) {
      // the default mode never pins a model to the generic kernel because of a file: the next call starts the background compile
      (void)hipGetLastError();
      sl.mod = nullptr;
      sl.skip_disk = true;
      fprintf(stderr, "[mrk] the code object on disk for %s ('%s') does not load on this device: compiling\n", JIT_KERNEL_NAME[kernel], prog.model.c_str());
      return stand_in();
    }
    if (!from_disk) MRK_HIP(hipModuleLoadData(&sl.mod, code.data()));
    MRK_HIP(hipModuleGetFunction(&sl.fn, sl.mod, JIT_KERNEL_NAME[kernel]));
    sl.key = cache_stem(jit_source(prog, f64, kernel, sg.get()));
  } catch (const std::exception &e) {
    sl.failed = true;
    if (mode == 2) throw;
    fprintf(stderr, "[mrk] specialised kernel %s for model '%s' unavailable, using the generic kernel: %s\n", JIT_KERNEL_NAME[kernel], prog.model.c_str(), e.what());
    return nullptr;
  }
  return (void *)sl.fn;
}

static void *jit_function(const Program &prog, int kernel, bool f64, const QsSignature *sig, bool wait = false) {
  if (jit_mode() == 0) return nullptr;
  std::lock_guard<std::mutex> lk(prog.jit_mu);
  return jit_function_locked(prog, kernel, f64, wait, sig, false);
}

// host only: make sure `dir` holds the code object of every kernel in `kernel_mask` (bit k = kernel k) for this program;
// returns how many had to be compiled
int jit_precompile(const Program &prog, bool f64, unsigned kernel_mask, const std::string &dir, const QsSignature *sig) {
  int compiled = 0;
  for (int k = 0; k < JIT_KERNELS; ++k) {
    if (!(kernel_mask & (1u << k))) continue;
    const bool kf64 = jit_program_only(k) ? true : f64;
    if (jit_needs_sig(k) && !jit_items_rt_applies(sig)) continue;
    const std::string src = jit_source(prog, kf64, k, switches().jit_sig ? sig : nullptr);
    const std::string user = cache_path(src);
    const std::string name = user.empty() ? shipped_path(src) : user;
    if (name.empty()) throw StatusError(MRK_ERR_INVALID_ARG, "no cache file name");
    const std::string path = dir + name.substr(name.rfind('/'));
    if (!read_file(path).empty()) continue;
    std::vector<char> code = read_file(user);
    if (code.empty()) {
      std::string log;
      code = jit_compile(src, log);
      ++compiled;
    }
    write_file(path, code);
  }
  return compiled;
}

void *jit_rank_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_RANK, f64, sig); }
// the item-parallel kernel (nullptr under the same conditions)
void *jit_items_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_ITEMS, f64, sig); }
void *jit_items_rt_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_ITEMS_RT, f64, sig); }
// the op-split / sliced form of the fused kernel (small batches, few large requests)
void *jit_split_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_SPLIT, f64, sig); }
// the f64-matrix form of the fused kernel
void *jit_matrix_function(const Program &prog) { return jit_function(prog, JIT_MATRIX, true, nullptr); }
void *jit_prepass_function(const Program &prog) { return jit_function(prog, JIT_PREPASS, true, nullptr); }
// the one-launch kernel of small requests
void *jit_one_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_ONE, f64, sig); }
void *jit_fused_score_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_FUSED_SCORE, f64, sig); }
void *jit_serve_function(const Program &prog, bool f64, const QsSignature *sig) { return jit_function(prog, JIT_SERVE, f64, sig, /*wait=*/true); }  // mrk_serve_start IS the warm-up

// "<kernel name> <key>\n" for every specialised kernel of `prog` that is loaded (measurement provenance: which code ran)
std::string jit_loaded_keys(const Program &prog) {
  std::lock_guard<std::mutex> lk(prog.jit_mu);
  std::string out;
  if (!prog.jit) return out;
  JitKernels *k = (JitKernels *)prog.jit;
  for (auto &set : k->by_sig)
    for (int kn = 0; kn < JIT_KERNELS; ++kn)
      for (JitSlot &sl : set.second->slot[kn])
        if (sl.fn) out += std::string(JIT_KERNEL_NAME[kn]) + " " + sl.key + ((set.first.empty() || set.first[0] == '\n') ? " program" : " program+forest") + "\n";
  return out;
}

// waits for the background compiles of `prog` that are under way (a warm-up / measurement aid; the next launch loads them)
void jit_wait(const Program &prog) {
  std::lock_guard<std::mutex> lk(prog.jit_mu);
  if (!prog.jit) return;
  JitKernels *k = (JitKernels *)prog.jit;
  for (auto &set : k->by_sig)
    for (auto &per_kernel : set.second->slot)
      for (JitSlot &sl : per_kernel)
        if (sl.state.load() == 1 && sl.worker.joinable()) sl.worker.join();
}

#ifdef MRK_PHASE_CLOCKS
// measurement builds: read-and-reset the phase clocks of the specialised kernel of `prog`
extern "C" int mrk_debug_phase_clocks(const Program *prog, unsigned long long *out64) {
  if (!prog || !prog->jit) return -1;
  JitKernels *k = (JitKernels *)prog->jit;
  hipModule_t mod = nullptr;
  const char *want = getenv("MRK_PHASE_KERNEL");   // "items": the item-parallel kernel's clocks (c4 / c4x)
  const bool items = want && !strcmp(want, "items");
  for (auto &set : k->by_sig)   // the kernel that was compiled last (a measurement build ranks one model)
    for (int f = 1; f >= 0; --f) {
      if (!items && set.second->slot[JIT_RANK][f].mod) mod = set.second->slot[JIT_RANK][f].mod;
      if (items && set.second->slot[JIT_ITEMS][f].mod && !(mod && switches().items_rt)) mod = set.second->slot[JIT_ITEMS][f].mod;
      if (items && switches().items_rt && set.second->slot[JIT_ITEMS_RT][f].mod) mod = set.second->slot[JIT_ITEMS_RT][f].mod;
    }
  hipDeviceptr_t p = nullptr;
  size_t bytes = 0;
  if (!mod || hipModuleGetGlobal(&p, &bytes, mod, "mrk_phase_clocks") != hipSuccess || bytes < 64 * 8) return -2;
  if (hipMemcpy(out64, (void *)p, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -3;
  (void)hipMemset((void *)p, 0, 64 * 8);
  return 0;
}
#endif

void jit_release(Program &prog) {
  if (!prog.jit) return;
  JitKernels *k = (JitKernels *)prog.jit;
  for (auto &set : k->by_sig)
    for (auto &per_kernel : set.second->slot)
      for (JitSlot &sl : per_kernel) {
        if (sl.worker.joinable()) sl.worker.join();  // the worker reads `prog`
        if (sl.mod) (void)hipModuleUnload(sl.mod);
      }
  delete k;
  prog.jit = nullptr;
}

}  // namespace mrk
