// Run-time specialised assembly kernel (jit.cpp).
#pragma once
#include <string>
#include <vector>

namespace mrk {

struct Program;
struct QsSignature;  // forest.hpp: the forest's view signature - the kernels that write the scorer's tile are keyed by it too

// the specialised kernels of a program; each is compiled (and cached on disk) by itself when a batch first needs it
enum { JIT_RANK = 0, JIT_SPLIT = 1, JIT_MATRIX = 2, JIT_ITEMS = 3, JIT_ONE = 4, JIT_SERVE = 5, JIT_FUSED_SCORE = 6, JIT_PREPASS = 7, JIT_ITEMS_RT = 8, JIT_KERNELS = 9, JIT_ALL = -1 };
// the translation unit hiprtc compiles for this model's program: the shared device code + the program as constants +
// the kernel `kernel` (JIT_ALL: every kernel - inspection tools)
std::string jit_source(const Program &prog, bool f64, int kernel = JIT_ALL, const QsSignature *sig = nullptr);
// gfx950 code object of `source`; throws StatusError(MRK_ERR_DEVICE) with the compiler log.  No device needed.
std::vector<char> jit_compile(const std::string &source, std::string &log);
// hipFunction_t of the specialised fused kernel for (program, scorer precision), built on first use; nullptr when
// specialisation is switched off (MRK_RANK_JIT=0) or hiprtc failed (warning on stderr; MRK_RANK_JIT=require throws)
void *jit_rank_function(const Program &prog, bool f64, const QsSignature *sig);
// the item-parallel assembly kernel (mrk_jit_assemble_cells), same conditions
void *jit_items_function(const Program &prog, bool f64, const QsSignature *sig);
// its persistent form with every threshold table of the forest resident in LDS (mrk_jit_assemble_cells_rt): exists only for a
// forest whose signature is known and whose tables fit (jit_items_rt_applies); nullptr otherwise and while it compiles
void *jit_items_rt_function(const Program &prog, bool f64, const QsSignature *sig);
bool jit_items_rt_applies(const QsSignature *sig);
// the fused kernel whose workgroups split the program's ops over copies of the item lanes (mrk_jit_rank_cells_split)
void *jit_split_function(const Program &prog, bool f64, const QsSignature *sig);
// the fused kernel writing the row-major f64 matrix (mrk_jit_rank_matrix), same conditions
void *jit_matrix_function(const Program &prog);
// the pre-pass alone (mrk_jit_prepass: requests too large for one workgroup's assembly - config 4), keyed by the program only
void *jit_prepass_function(const Program &prog);
// pre-pass + assembly + forest + ordering of a small request in one launch (mrk_jit_rank_one), same conditions
void *jit_one_function(const Program &prog, bool f64, const QsSignature *sig);
// the persistent workgroup of the serving queue (mrk_jit_rank_serve), same conditions
void *jit_serve_function(const Program &prog, bool f64, const QsSignature *sig);
// full batches of small requests: assembly + forest + ordering in one launch (mrk_jit_rank_fused_score)
void *jit_fused_score_function(const Program &prog, bool f64, const QsSignature *sig);
int jit_precompile(const Program &prog, bool f64, unsigned kernel_mask, const std::string &dir, const QsSignature *sig = nullptr);
void jit_wait(const Program &prog);
std::string jit_loaded_keys(const Program &prog);
void jit_release(Program &prog);

}  // namespace mrk
