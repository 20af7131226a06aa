// Feature store + rank entry points of the C ABI (include/mrk.h).
#include "runtime.hpp"

namespace mrk {
void free_rank_state(mrk_ctx *ctx) { (void)ctx; }
static int todo(const char *what) {
  set_last_error(std::string(what) + ": not implemented yet");
  return MRK_ERR_UNSUPPORTED;
}
}  // namespace mrk
using namespace mrk;

extern "C" {
int mrk_config_load_json(mrk_ctx *, const char *, size_t) { return todo("mrk_config_load_json"); }
int mrk_model_dim(mrk_ctx *, const char *) { return todo("mrk_model_dim"); }
int mrk_store_put_double(mrk_ctx *, const char *, double) { return todo("mrk_store_put_double"); }
int mrk_store_put_bool(mrk_ctx *, const char *, int) { return todo("mrk_store_put_bool"); }
int mrk_store_put_string(mrk_ctx *, const char *, const char *) { return todo("mrk_store_put_string"); }
int mrk_store_put_string_list(mrk_ctx *, const char *, const char *const *, int) { return todo("mrk_store_put_string_list"); }
int mrk_store_put_double_list(mrk_ctx *, const char *, const double *, int) { return todo("mrk_store_put_double_list"); }
int mrk_store_put_counter(mrk_ctx *, const char *, int64_t) { return todo("mrk_store_put_counter"); }
int mrk_store_put_periodic(mrk_ctx *, const char *, const int64_t *, int) { return todo("mrk_store_put_periodic"); }
int mrk_store_put_bounded_list(mrk_ctx *, const char *, const char *const *, int) { return todo("mrk_store_put_bounded_list"); }
int mrk_store_delete(mrk_ctx *, const char *) { return todo("mrk_store_delete"); }
int mrk_store_flush(mrk_ctx *) { return todo("mrk_store_flush"); }
int mrk_rank(mrk_ctx *, mrk_model *, const char *, const mrk_request *, double *, int32_t *, double *) { return todo("mrk_rank"); }
int mrk_batch_prepare(mrk_ctx *, const char *, const mrk_request *, int, mrk_batch **) { return todo("mrk_batch_prepare"); }
int mrk_batch_total_items(mrk_batch *) { return todo("mrk_batch_total_items"); }
int mrk_batch_run(mrk_batch *, mrk_model *) { return todo("mrk_batch_run"); }
int mrk_batch_device_outputs(mrk_batch *, double **, int32_t **, double **) { return todo("mrk_batch_device_outputs"); }
int mrk_batch_fetch(mrk_batch *, double *, int32_t *, double *) { return todo("mrk_batch_fetch"); }
void mrk_batch_free(mrk_batch *) {}
}
