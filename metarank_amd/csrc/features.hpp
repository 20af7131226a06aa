// Feature registry: Metarank's `features:` / `models:` config -> store layout + per-model
// assembly programs (reference: FeatureMapping.fromFeatureSchema, FeatureMapping.scala:56-99, and
// the schema decoders in feature/*.scala), plus the host half of a request (everything that is a
// function of the RankingEvent alone: slots, request-level constants, per-item overrides).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "rank.hpp"
#include "store.hpp"

struct mrk_encoder;

namespace mrk {

enum class FType {
  Number, Boolean, WordCount, Vector, String, InteractionCount, WindowCount, Rate, InteractedWith, Diversity,
  ItemAge, LocalTime, Position, Relevancy, Biencoder, ExternalRanking, ExternalItem
};

struct FeatureDef {
  FType type = FType::Number;
  std::string name;
  ScopeId scope = SC_ITEM;      // SC_FIELD / SC_IRF stand for item.<f> / ranking.<f> scopes of `rate`
  std::string scope_field;
  std::string field;            // source field name
  bool field_is_ranking = false;
  int dim = 1;
  bool index_encode = false;            // string
  std::vector<std::string> values;      // string: possible values; interacted_with: field names
  std::string top, bottom;              // rate
  bool normalize = false;
  double weight = 0;
  int div_top = 20;
  double position = 0;
  int mapper = 0;                       // local_time
  int norm = NORM_NOOP;                 // bi-encoder
  int qdim = 0;                         // bi-encoder embedding size
  std::string ext_field;                // request / item field carrying host-computed values
  int64_t bucket_ms = 0;                // window_count / rate: bucket length
  std::vector<int32_t> periods;         // window_count / rate: PeriodRange.startOffset per column
  int64_t list_count = 100;             // interacted_with: BoundedListConfig.count
  int64_t list_duration_ms = 24LL * 3600 * 1000;
  bool cross = false;                   // field_match / cross-encoder: per-item logits of (query, item text) pairs
  mrk_encoder *encoder = nullptr;       // bi-encoder with `method.model`: bound by mrk_config_bind_encoder (one reference held)
};

// host-side description of what a request has to supply for one op
struct HostOp {
  const FeatureDef *def = nullptr;
  int dst = 0;
  int const_idx = -1;   // offset of this op's constants inside the per-request const block
  int irf_id = -1;      // index of its per-item IRF slot array
  int prep_base = -1;   // first pre-pass entry
};

struct Program {
  std::string model;
  std::vector<std::string> feature_names;
  std::vector<Op> ops;
  std::vector<HostOp> host_ops;
  std::vector<PrepEntry> prep;
  std::vector<uint32_t> aux;
  int dim = 0;
  int n_consts = 0;
  int n_irf = 0;
  // matrix columns normalised across the request after assembly and overrides (ml/onnx/Normalize.scala:13-45).  A
  // cross-encoder column is normalised here only while an encoder is bound to it (the library then produces the raw
  // logits; without one the host supplies finished values): `cross` points at its definition.
  struct NormCol { int col; int mode; const FeatureDef *cross; };
  std::vector<NormCol> norm_cols;
  bool normalises() const {
    for (const NormCol &n : norm_cols)
      if (!n.cross || n.cross->encoder) return true;
    return false;
  }
  int item_fixed = 0;   // bytes of tags + value cells of an ITEM record (Table::heap_off): what the specialised kernel keeps in registers
  DevBuf d_ops, d_prep, d_aux;
  mutable std::mutex jit_mu;        // guards `jit` (the first ranks of a model may come from several threads)
  mutable void *jit = nullptr;      // JitKernels* (jit.cpp): the kernel specialised for this program, built on first use
  mutable bool jit_failed = false;
  ProgramDev device_view() const;
  ~Program();
};

struct Registry {
  std::vector<std::unique_ptr<FeatureDef>> features;
  std::map<std::string, std::unique_ptr<Program>> programs;
  const Program *program(const std::string &model) const;
  ~Registry();
};

// parses the config, declares every state column in `store`, freezes the layout, builds and
// uploads the programs
// (`upload` false: host-side only, for mrk_config_specialize)
std::unique_ptr<Registry> load_config(const char *json, size_t len, Store &store, bool upload = true);
// iteration order of a Scala 2.13 immutable Map with these String keys inserted in this order (interacted_with's columns)
std::vector<std::string> scala_map_key_order(const std::vector<std::string> &keys);

// ---- host half of a batch -------------------------------------------------------------------
struct HostBatch {
  std::vector<ReqDev> reqs;
  std::vector<int32_t> item_slot;
  std::vector<uint32_t> item_req;
  std::vector<double> consts;
  std::vector<int32_t> irf;
  std::vector<Override> overrides;
  std::vector<PrepOut> prep_out;
  uint64_t arena_entries = 0;
  uint64_t max_req_entries = 0;   // largest per-request table total (fused kernel: LDS sizing)
  int max_doubles = 0;            // most numeric diversity values of any (request, feature)
  int max_items = 0;              // largest request
  int total_items = 0;
  bool device_ids = false;        // item_slot / item_req are left to the device (resolve.hip): the batch came with flat ids
};

// ids == nullptr: the ids are reqs[r].item_ids (NUL-terminated strings), looked up on the host, and the pre-pass tables
// are sized from exact token counts read from the host mirror.  ids != nullptr: reqs[r].item_ids is ignored, the ids of
// all requests lie back to back in ids->bytes (request order); their slots are resolved by the device and the tables are
// sized from upper bounds (list lengths x the longest list a column ever held) - no per-item work on the host.
void resolve_requests(const Program &prog, Store &store, const mrk_request *reqs, int n_req, const mrk_item_ids *ids, HostBatch &out);

}  // namespace mrk
