// Plain types shared by host code and device code: the record tags / scopes of the feature store (store.hpp),
// what kernels see of the store, and the structs of the bit-vector forest format (forest.hpp).  This header, rank.hpp,
// qs_device.hpp and rank_device.hpp are also the translation unit of the run-time specialised assembly kernel
// (jit.cpp: hiprtc), so they use nothing but fixed-width integers.
#pragma once
#ifdef __HIPCC_RTC__
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef short int16_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long size_t;
#else
#include <cstddef>
#include <cstdint>
#endif

namespace mrk {

// ---- feature store (layout: store.hpp)
enum ScopeId : int {
  SC_GLOBAL = 0, SC_ITEM = 1, SC_USER = 2, SC_SESSION = 3, SC_RANKING = 4, SC_FIELD = 5, SC_IRF = 6, SC_COUNT = 7
};

// tag byte of a record cell
enum Tag : uint8_t {
  TAG_MISSING = 0,
  TAG_DOUBLE = 1,       // ScalarValue(SDouble): cell = f64
  TAG_BOOL = 2,         // ScalarValue(SBoolean): cell = f64 0/1
  TAG_STRING = 3,       // ScalarValue(SString): cell = {u32 token, u32 linked field slot + 1 (0 = none)}
  TAG_STRING_LIST = 4,  // ScalarValue(SStringList): cell = {u32 offset, u32 length}; offset & LIST_INLINE_BIT: byte offset of the
                        // tokens inside the record itself (its inline heap), else an index into the token pool
  TAG_DOUBLE_LIST = 5,  // ScalarValue(SDoubleList): cell = {u32 offset, u32 length}; offset & LIST_F32_BIT: the values are kept as
                        // f32 in the f32 pool (every one of them is exactly a float: embeddings are f32 values widened at
                        // ingest, model/Scalar.scala:24-32 - half the bytes per candidate, the same doubles), else f64 pool
  TAG_PRESENT = 1,      // counter / bounded list present; periodic: tag = 1 + min(len, 250)
};

constexpr uint32_t LIST_INLINE_BIT = 0x80000000u;
constexpr uint32_t LIST_F32_BIT = 0x80000000u;
constexpr uint32_t LIST_F32_MIN = 16;   // shorter double lists stay f64 (nothing to gain)

struct TableDev {          // what kernels see
  const uint8_t *rows;
  uint32_t stride;
  uint32_t n_slots;
};

struct StoreDev {
  TableDev tab[SC_COUNT];
  const uint32_t *tok_pool;
  const double *f64_pool;
  const uint32_t *slot_pool;
  const float *f32_pool;
};

// ---- id -> slot table of a scope (store.hpp SlotMap), mirrored to the device for the ITEM table: a request batch
// hands over the UTF-8 bytes of its item ids and the slots are looked up by a kernel (resolve.hip) instead of by
// the host.  Open addressing, linear probing, power-of-two capacity; hash 0 = empty entry.
struct IdEntry { uint64_t hash; uint32_t slot; uint32_t off; };  // off: the id (NUL-terminated) in the arena
struct IdTableDev {
  const IdEntry *table;
  const uint8_t *arena;
  uint32_t mask;       // capacity - 1; 0 with table == nullptr: no ids yet
  uint32_t pad;
};

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define MRK_HD __host__ __device__
#else
#define MRK_HD
#endif

// The hash of an id (bytes [s, s + len)), read a byte at a time so that host and device agree on any alignment:
// 8-byte little-endian words, the tail zero-padded (the host's SlotMap::hash computes the same with word loads).
MRK_HD inline uint64_t id_hash_bytes(const uint8_t *s, size_t len) {
  uint64_t h = 0x9e3779b97f4a7c15ull ^ ((uint64_t)len * 0xff51afd7ed558ccdull);
  size_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t w = 0;
    for (int k = 0; k < 8; ++k) w |= (uint64_t)s[i + k] << (8 * k);
    h = (h ^ w) * 0x9fb21c651e98df25ull;
    h ^= h >> 29;
  }
  uint64_t w = 0;
  for (int k = 0; i + k < len; ++k) w |= (uint64_t)s[i + k] << (8 * k);
  h = (h ^ w) * 0x9fb21c651e98df25ull;
  h ^= h >> 32;
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 29;
  return h ? h : 1;
}

// ---- bit-vector forest format (described in forest.hpp)
constexpr int QS_SLOTS = 16;
constexpr int QS_LEAVES = 16;
constexpr int QS_TREE_WORDS = QS_SLOTS * 2;
constexpr int QS_MAX_VIEWS = 255;
constexpr uint16_t QS_RIGHT = 0x7FFF;
constexpr uint32_t QS_STAGE_CHUNK = 128;  // doubles one global_load_lds_dwordx4 of a wavefront moves (64 lanes x 16 B)
enum QsViewKind : uint8_t { QV_NAN_RIGHT = 0, QV_NAN_LEFT = 1, QV_MISS_RIGHT = 2, QV_MISS_LEFT = 3, QV_CAT = 4, QV_NAN_ZERO = 5 };

constexpr uint16_t QS_CAT_BEYOND = 0x7FFD, QS_CAT_INVALID = 0x7FFE, QS_CAT_NAN = 0x7FFF;
struct QsView {        // 4 B, one per column of the binned tile; grouped by feature
  uint16_t feature;
  uint8_t kind;        // QsViewKind
  uint8_t pad;
};
struct QsCatNode {     // 16 B
  uint32_t view_dl;    // view index | default_left << 16 (XGBoost: where NaN goes)
  uint32_t mm;         // m | m << 16
  uint32_t bits_begin; // first word of the bitset in PackedForestQS::cat_bits
  uint32_t bits_words;
};
struct QsFeature {     // 16 B, one per matrix column (one s_load_dwordx4)
  uint32_t thr_off;                // its sorted distinct thresholds in PackedForestQS::thr ...
  uint16_t thr_len;                // ... <= 32766 of them
  uint16_t zero_bin;               // bin(0.0): the cell of a NaN in a QV_NAN_ZERO view
  uint8_t view_begin, view_end;    // its views (tile columns)
  uint16_t pad;
  uint32_t view_kinds;             // QsViewKind of view_begin + i in bits [4 i, 4 i + 4): at most 6 views per column
};
// What the assembly kernels need to know about a column to bin a value WITHOUT reading its descriptor: the part of a
// QsFeature that survives a retraining on the same features (the thresholds themselves, their exact number and bin(0.0)
// do not).  The per-column rows of a forest are its "view signature" (forest.hpp qs_signature); the run-time specialised
// kernels (jit.cpp) are keyed by it and hold it as compile-time constants.
struct QsSig {
  uint32_t thr_off;                // = QS_STAGE_CHUNK x the chunks of the columns before this one (forest.cpp lays the tables out that way)
  uint16_t chunks;                 // ceil(thr_len / QS_STAGE_CHUNK)
  uint8_t view_begin, view_end;
  uint32_t view_kinds;
  // the column's table in the COMPACT layout the resident-table sinks search (forest.hpp PackedForestQS::thr_rt): offset and
  // EXACT length in doubles (0: no thresholds; QS_RT_NONE: not resident - more than 256 thresholds, searched in global memory)
  uint32_t rt_off;
  uint32_t rt_len;
};
constexpr uint32_t QS_RT_NONE = 0xffffffffu;

}  // namespace mrk
