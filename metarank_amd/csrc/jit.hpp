// Run-time specialised assembly kernel (jit.cpp).
#pragma once
#include <string>
#include <vector>

namespace mrk {

struct Program;

// the translation unit hiprtc compiles for this model's program: the shared device code + the program as constants
std::string jit_source(const Program &prog, bool f64);
// gfx950 code object of `source`; throws StatusError(MRK_ERR_DEVICE) with the compiler log.  No device needed.
std::vector<char> jit_compile(const std::string &source, std::string &log);
// hipFunction_t of the specialised fused kernel for (program, scorer precision), built on first use; nullptr when
// specialisation is switched off (MRK_RANK_JIT=0) or hiprtc failed (warning on stderr; MRK_RANK_JIT=require throws)
void *jit_rank_function(const Program &prog, bool f64);
// the item-parallel assembly kernel of the same specialised module (mrk_jit_assemble_cells), same conditions
void *jit_items_function(const Program &prog, bool f64);
// the fused kernel whose workgroups split the program's ops over copies of the item lanes (mrk_jit_rank_cells_split)
void *jit_split_function(const Program &prog, bool f64);
// the fused kernel writing the row-major f64 matrix (mrk_jit_rank_matrix), same conditions
void *jit_matrix_function(const Program &prog);
void jit_release(Program &prog);

}  // namespace mrk
