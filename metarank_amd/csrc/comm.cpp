// Multi-GPU inside the library: one context per GPU - in one process per GPU (mrk_comm_init, a rendezvous by
// unique id) or all in ONE host process (mrk_comm_init_local: the JVM of HipConfig(devices)) -, RCCL over xGMI linked
// directly - the host language never sees a collective.  The reference has no counterpart (it scores a request on one
// JVM thread, ml/Ranker.scala:27-83; scale-out is whole-request replicas, doc/dev/production-recommendations.md): the
// only exchange this path has is the merge of score slices -
//   item-sharded rank of ONE large request (BASELINE config 4): every rank assembles + scores its tile-aligned slice of
//     the candidates, ONE in-place ncclAllGather of chunk x world f64 scores on the batch's stream, then the sort;
//   request-sharded replicas: nothing to exchange to rank; mrk_batch_gather_scores merges the ranks' score vectors for
//     a host that wants them in one place.
// Rendezvous: rank 0 calls mrk_comm_unique_id and hands the 128 bytes to the other ranks by whatever channel the host
// has (bench.py: one TCP message to MASTER_ADDR:MASTER_PORT); every rank then calls mrk_comm_init.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "runtime.hpp"

namespace mrk {

#define MRK_NCCL(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t _r = (expr);                                                                   \
    if (_r != ncclSuccess)                                                                      \
      throw ::mrk::StatusError(MRK_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(_r)); \
  } while (0)

template <typename F>
static int guard(F &&f) {
  try {
    f();
    return MRK_OK;
  } catch (const StatusError &e) {
    set_last_error(e.what());
    return e.status;
  } catch (const std::exception &e) {
    set_last_error(e.what());
    return MRK_ERR_DEVICE;
  }
}

void comm_destroy(mrk_ctx *ctx) {
  if (ctx->comm) {
    (void)hipSetDevice(ctx->device);
    (void)ncclCommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = 0;
    ctx->comm_world = 1;
  }
}

// in-place all-gather of `chunk` f64 per rank inside `buf` (rank r's slice at buf + r * chunk), on `stream`
void comm_allgather_f64_inplace(mrk_ctx *ctx, double *buf, size_t chunk, hipStream_t stream) {
  if (!ctx->comm) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_comm_init has not been called on this context");
  std::lock_guard<std::mutex> cl(ctx->comm_mu);
  MRK_NCCL(ncclAllGather(buf + (size_t)ctx->comm_rank * chunk, buf, chunk, ncclFloat64, (ncclComm_t)ctx->comm, stream));
}

// all-gather of `count` i32 per rank (the per-request status words of an item-sharded run)
void comm_allgather_i32(mrk_ctx *ctx, const int32_t *send, int32_t *recv, size_t count, hipStream_t stream) {
  if (!ctx->comm) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_comm_init has not been called on this context");
  std::lock_guard<std::mutex> cl(ctx->comm_mu);
  MRK_NCCL(ncclAllGather(send, recv, count, ncclInt32, (ncclComm_t)ctx->comm, stream));
}

void comm_allgather_f64(mrk_ctx *ctx, const double *send, double *recv, size_t count, hipStream_t stream) {
  if (!ctx->comm) throw StatusError(MRK_ERR_INVALID_ARG, "mrk_comm_init has not been called on this context");
  std::lock_guard<std::mutex> cl(ctx->comm_mu);
  MRK_NCCL(ncclAllGather(send, recv, count, ncclFloat64, (ncclComm_t)ctx->comm, stream));
}

}  // namespace mrk

using namespace mrk;

extern "C" {

int mrk_comm_unique_id(uint8_t *out) {
  return guard([&] {
    if (!out) throw StatusError(MRK_ERR_INVALID_ARG, "null output");
    static_assert(sizeof(ncclUniqueId) == MRK_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    MRK_NCCL(ncclGetUniqueId(&id));
    memcpy(out, &id, sizeof id);
  });
}

int mrk_comm_init(mrk_ctx *ctx, const uint8_t *id, int rank, int world) {
  return guard([&] {
    if (!ctx || !id || world < 1 || rank < 0 || rank >= world) throw StatusError(MRK_ERR_INVALID_ARG, "bad communicator arguments");
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->comm) throw StatusError(MRK_ERR_INVALID_ARG, "this context already has a communicator");
    MRK_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    MRK_NCCL(ncclCommInitRank(&comm, world, uid, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    ctx->d_comm.reserve(256);
  });
}

// The ranks of a communicator living in ONE process (the JVM with HipConfig(devices = List(0, 1, ...)): mrk_init made the
// contexts): one ncclCommInitRank per context between ncclGroupStart / ncclGroupEnd - no rendezvous channel, no second
// process.  Rank i = ctxs[i].  Afterwards each context is driven by its own host thread exactly like a rank of a
// multi-process job (mrk_batch_run_sharded and friends; every rank must issue the collectives in the same order).
int mrk_comm_init_local(mrk_ctx *const *ctxs, int n) {
  return guard([&] {
    if (!ctxs || n < 1) throw StatusError(MRK_ERR_INVALID_ARG, "bad communicator arguments");
    for (int i = 0; i < n; ++i) {
      if (!ctxs[i]) throw StatusError(MRK_ERR_INVALID_ARG, "null context");
      if (ctxs[i]->comm) throw StatusError(MRK_ERR_INVALID_ARG, "this context already has a communicator");
      for (int j = 0; j < i; ++j)
        if (ctxs[j] == ctxs[i] || ctxs[j]->device == ctxs[i]->device)
          throw StatusError(MRK_ERR_INVALID_ARG, "mrk_comm_init_local: two ranks on device " + std::to_string(ctxs[i]->device) +
                                                     " (RCCL wants one rank per GPU; contexts sharing a GPU rank independently, without a communicator)");
    }
    ncclUniqueId uid;
    MRK_NCCL(ncclGetUniqueId(&uid));
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    MRK_NCCL(ncclGroupStart());
    ncclResult_t rc = ncclSuccess;
    for (int i = 0; i < n && rc == ncclSuccess; ++i) {
      if (hipSetDevice(ctxs[i]->device) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
      rc = ncclCommInitRank(&comms[(size_t)i], n, uid, i);
    }
    const ncclResult_t end = ncclGroupEnd();
    if (rc != ncclSuccess || end != ncclSuccess) {
      for (ncclComm_t c : comms) if (c) (void)ncclCommAbort(c);
      throw StatusError(MRK_ERR_DEVICE, std::string("mrk_comm_init_local: ") + ncclGetErrorString(rc != ncclSuccess ? rc : end));
    }
    // all or nothing: a failure while the contexts take their communicators (hipSetDevice, the scratch allocation) aborts every
    // communicator and clears the contexts already assigned - no half-initialised world, and a retry is not refused
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    try {
      for (int i = 0; i < n; ++i) {
        std::lock_guard<std::mutex> lk(ctxs[i]->mu);
        MRK_HIP(hipSetDevice(ctxs[i]->device));
        ctxs[i]->d_comm.reserve(256);
        ctxs[i]->comm = comms[(size_t)i];
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = n;
      }
    } catch (...) {
      for (int i = 0; i < n; ++i) {
        std::lock_guard<std::mutex> lk(ctxs[i]->mu);
        if (ctxs[i]->comm == comms[(size_t)i]) { ctxs[i]->comm = nullptr; ctxs[i]->comm_rank = 0; ctxs[i]->comm_world = 1; }
      }
      for (ncclComm_t c : comms) if (c) (void)ncclCommAbort(c);
      if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
      throw;
    }
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
  });
}

// Host-only shard arithmetic (no context, no device): what mrk_batch_shard_chunk / mrk_batch_run_shard use.
int64_t mrk_shard_chunk(int64_t total_items, int shard_count) {
  if (total_items < 0 || shard_count < 1) return MRK_ERR_INVALID_ARG;
  const int64_t per = (total_items + shard_count - 1) / shard_count;
  return (per + MRK_SHARD_TILE - 1) / MRK_SHARD_TILE * MRK_SHARD_TILE;
}

int mrk_shard_range(int64_t total_items, int shard_index, int shard_count, int64_t *lo, int64_t *hi) {
  if (total_items < 0 || shard_count < 1 || shard_index < 0 || shard_index >= shard_count || !lo || !hi) return MRK_ERR_INVALID_ARG;
  const int64_t chunk = mrk_shard_chunk(total_items, shard_count);
  *lo = std::min(chunk * shard_index, total_items);
  *hi = std::min(chunk * (shard_index + 1), total_items);
  return MRK_OK;
}

int mrk_comm_rank(mrk_ctx *ctx) { return ctx ? ctx->comm_rank : MRK_ERR_INVALID_ARG; }
int mrk_comm_world(mrk_ctx *ctx) { return ctx ? ctx->comm_world : MRK_ERR_INVALID_ARG; }

// max over the ranks of a host value (bench.py: the slowest rank's time); doubles as a barrier
int mrk_comm_max_f64(mrk_ctx *ctx, double *value) {
  return guard([&] {
    if (!ctx || !value) throw StatusError(MRK_ERR_INVALID_ARG, "null argument");
    if (!ctx->comm) return;  // a world of one
    std::lock_guard<std::mutex> cl(ctx->comm_mu);  // also the owner of d_comm
    MRK_HIP(hipSetDevice(ctx->device));
    MRK_HIP(hipMemcpyAsync(ctx->d_comm.p, value, 8, hipMemcpyHostToDevice, ctx->stream));
    MRK_NCCL(ncclAllReduce(ctx->d_comm.p, ctx->d_comm.p, 1, ncclFloat64, ncclMax, (ncclComm_t)ctx->comm, ctx->stream));
    MRK_HIP(hipMemcpyAsync(value, ctx->d_comm.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    MRK_HIP(hipStreamSynchronize(ctx->stream));
  });
}

int mrk_comm_barrier(mrk_ctx *ctx) {
  double v = 0.0;
  return mrk_comm_max_f64(ctx, &v);
}

}  // extern "C"
