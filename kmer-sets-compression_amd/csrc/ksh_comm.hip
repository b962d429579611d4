// Transport of the multi-GPU build (one process per GPU): all-gather and point-to-point copies of
// DEVICE buffers between the ranks.
//
//   * RCCL, the transport of a real node (xGMI between the GPUs): librccl is opened at run time
//     (a host program that never builds across GPUs does not need it installed) and its
//     collectives are enqueued on the context's stream -- no host bounce, no Python in the path.
//   * caller-supplied functions, for rehearsals where RCCL cannot run (several ranks sharing one
//     GPU: RCCL refuses two ranks on a device): the tests pass gloo through host memory.
//
// The reference has no counterpart (one process, shared memory, lib/core/kmer_set_set.h:109-427);
// what the ranks exchange is what its pool threads share through memory: the sampled sets
// (:138-153), the SPSS weights (:240-264) and the sets of a merged pair (:333-336).
#include "ksh_internal.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstddef>
#include <cstring>
#include <vector>

#include "ksh_comm.h"

namespace {

// The few RCCL entry points used, with the library's own C signatures (rccl.h of ROCm 7.2:
// ncclUniqueId is 128 opaque bytes passed by value; ncclInt8 == 0).
struct UniqueId {
  char internal[128];
};
using comm_t = void*;
using fn_get_unique_id = int (*)(UniqueId*);
using fn_comm_init_rank = int (*)(comm_t*, int, UniqueId, int);
using fn_comm_destroy = int (*)(comm_t);
using fn_comm_abort = int (*)(comm_t);
using fn_all_gather = int (*)(const void*, void*, size_t, int, comm_t, hipStream_t);
using fn_send = int (*)(const void*, size_t, int, int, comm_t, hipStream_t);
using fn_recv = int (*)(void*, size_t, int, int, comm_t, hipStream_t);
using fn_error_string = const char* (*)(int);

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_comm_abort comm_abort = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_send send = nullptr;
  fn_recv recv = nullptr;
  fn_error_string error_string = nullptr;
};

int load_rccl(Rccl* r) {
  static Rccl cached;
  if (!cached.lib) {
    // the copy already in the process (torch brings its own) before the system's
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names)
      if ((lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!lib)
      for (const char* nm : names)
        if ((lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!lib) return ksh::fail(KSH_INTERNAL, "librccl not found: %s", dlerror());
    Rccl c;
    c.lib = lib;
    c.get_unique_id = reinterpret_cast<fn_get_unique_id>(dlsym(lib, "ncclGetUniqueId"));
    c.comm_init_rank = reinterpret_cast<fn_comm_init_rank>(dlsym(lib, "ncclCommInitRank"));
    c.comm_destroy = reinterpret_cast<fn_comm_destroy>(dlsym(lib, "ncclCommDestroy"));
    c.comm_abort = reinterpret_cast<fn_comm_abort>(dlsym(lib, "ncclCommAbort"));
    c.all_gather = reinterpret_cast<fn_all_gather>(dlsym(lib, "ncclAllGather"));
    c.send = reinterpret_cast<fn_send>(dlsym(lib, "ncclSend"));
    c.recv = reinterpret_cast<fn_recv>(dlsym(lib, "ncclRecv"));
    c.error_string = reinterpret_cast<fn_error_string>(dlsym(lib, "ncclGetErrorString"));
    if (!c.get_unique_id || !c.comm_init_rank || !c.comm_destroy || !c.all_gather || !c.send || !c.recv)
      return ksh::fail(KSH_INTERNAL, "librccl lacks an entry point this library needs");
    cached = c;
  }
  *r = cached;
  return KSH_OK;
}

}  // namespace

struct ksh_comm {
  ksh_ctx* ctx = nullptr;
  int rank = 0, world = 1;
  bool custom = false;
  ksh_comm_fns fns{};
  Rccl rccl;
  comm_t nccl = nullptr;
  // The side channel: a second communicator on a stream of its own, for what travels UNDER the
  // context's compute -- the deferred exchanges of the convergence checks and the sets handed to
  // another rank for encoding.  (A communicator of its own, because operations of one communicator
  // must not run concurrently on two streams.)
  comm_t nccl_side = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t side_ready = nullptr, side_done = nullptr;
  bool aborted = false;
};

namespace ksh {

int comm_rank(const ksh_comm* c) { return c->rank; }
int comm_world(const ksh_comm* c) { return c->world; }

#define KSH_RCCL(c, expr)                                                                          \
  do {                                                                                             \
    int rc__ = (expr);                                                                             \
    if (rc__ != 0)                                                                                 \
      return ::ksh::fail(KSH_INTERNAL, "%s failed: %s", #expr,                                     \
                         (c)->rccl.error_string ? (c)->rccl.error_string(rc__) : "RCCL error");    \
  } while (0)

// Every rank contributes `bytes` from d_send; d_recv receives world * bytes, rank-major.  RCCL:
// enqueued on the context's stream.  Custom: the stream is drained first and the call returns
// when the data is there.
int comm_allgather(ksh_comm* c, const void* d_send, void* d_recv, size_t bytes) {
  if (c->world == 1) {
    if (d_send != d_recv && bytes)
      KSH_HIP(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, c->ctx->stream));
    return KSH_OK;
  }
  if (c->custom) {
    KSH_HIP(hipStreamSynchronize(c->ctx->stream));
    if (c->fns.allgather(c->fns.user, d_send, d_recv, bytes) != 0)
      return fail(KSH_INTERNAL, "the caller's all-gather failed");
    return KSH_OK;
  }
  KSH_RCCL(c, c->rccl.all_gather(d_send, d_recv, bytes, /* ncclInt8 */ 0, c->nccl, c->ctx->stream));
  return KSH_OK;
}

int comm_send(ksh_comm* c, const void* d_buf, size_t bytes, int peer) {
  if (bytes == 0) return KSH_OK;
  if (c->custom) {
    KSH_HIP(hipStreamSynchronize(c->ctx->stream));
    if (c->fns.send(c->fns.user, d_buf, bytes, peer) != 0) return fail(KSH_INTERNAL, "the caller's send failed");
    return KSH_OK;
  }
  KSH_RCCL(c, c->rccl.send(d_buf, bytes, 0, peer, c->nccl, c->ctx->stream));
  return KSH_OK;
}

// ---- the side channel.  Every operation starts when what the context's stream holds so far is done
// (its buffers are complete) and then runs on the side stream, in the order of the calls -- which must be
// the same on all ranks, like on the main channel.  Custom transport: the calls block, in place.
static int side_after_main(ksh_comm* c) {
  KSH_HIP(hipEventRecord(c->side_ready, c->ctx->stream));
  KSH_HIP(hipStreamWaitEvent(c->side, c->side_ready, 0));
  return KSH_OK;
}

int comm_side_allgather(ksh_comm* c, const void* d_send, void* d_recv, size_t bytes) {
  if (c->custom || c->world == 1) return comm_allgather(c, d_send, d_recv, bytes);
  KSH_TRY(side_after_main(c));
  KSH_RCCL(c, c->rccl.all_gather(d_send, d_recv, bytes, 0, c->nccl_side, c->side));
  return KSH_OK;
}

int comm_side_send(ksh_comm* c, const void* d_buf, size_t bytes, int peer) {
  if (bytes == 0) return KSH_OK;
  if (c->custom) return comm_send(c, d_buf, bytes, peer);
  KSH_TRY(side_after_main(c));
  KSH_RCCL(c, c->rccl.send(d_buf, bytes, 0, peer, c->nccl_side, c->side));
  return KSH_OK;
}

int comm_side_recv(ksh_comm* c, void* d_buf, size_t bytes, int peer) {
  if (bytes == 0) return KSH_OK;
  if (c->custom) return comm_recv(c, d_buf, bytes, peer);
  KSH_TRY(side_after_main(c));
  KSH_RCCL(c, c->rccl.recv(d_buf, bytes, 0, peer, c->nccl_side, c->side));
  return KSH_OK;
}

// The stream side operations run on (the context's own stream for the custom transport and for one rank).
hipStream_t comm_side_stream(ksh_comm* c) { return (c->custom || c->world == 1) ? c->ctx->stream : c->side; }

// The context's stream waits for everything the side channel holds so far (data it received).
int comm_side_join_main(ksh_comm* c) {
  if (c->custom || c->world == 1) return KSH_OK;
  KSH_HIP(hipEventRecord(c->side_done, c->side));
  KSH_HIP(hipStreamWaitEvent(c->ctx->stream, c->side_done, 0));
  return KSH_OK;
}

// A rank that cannot go on with the protocol (its replica of the control loop failed, a transport call
// failed) tears the transport down instead of leaving its peers in an exchange it will never join: RCCL:
// ncclCommAbort on both communicators (pending operations of this rank end; the caller is expected to take
// the job down, the peers' pending operations do not complete by themselves); custom transport: the
// caller's abort function, if it gave one.  Idempotent.
void comm_abort(ksh_comm* c) {
  if (!c || c->aborted) return;
  c->aborted = true;
  if (c->custom) {
    if (c->fns.abort) c->fns.abort(c->fns.user);
    return;
  }
  if (c->rccl.comm_abort) {
    if (c->nccl_side) (void)c->rccl.comm_abort(c->nccl_side);
    if (c->nccl) (void)c->rccl.comm_abort(c->nccl);
    c->nccl_side = c->nccl = nullptr;
  }
}

// The host waits for everything the side channel holds so far.
int comm_side_sync(ksh_comm* c) {
  KSH_HIP(hipStreamSynchronize(comm_side_stream(c)));
  return KSH_OK;
}

int comm_recv(ksh_comm* c, void* d_buf, size_t bytes, int peer) {
  if (bytes == 0) return KSH_OK;
  if (c->custom) {
    KSH_HIP(hipStreamSynchronize(c->ctx->stream));
    if (c->fns.recv(c->fns.user, d_buf, bytes, peer) != 0) return fail(KSH_INTERNAL, "the caller's recv failed");
    return KSH_OK;
  }
  KSH_RCCL(c, c->rccl.recv(d_buf, bytes, 0, peer, c->nccl, c->ctx->stream));
  return KSH_OK;
}

}  // namespace ksh

using namespace ksh;

extern "C" {

int ksh_comm_unique_id(unsigned char id[KSH_COMM_ID_BYTES]) {
  if (!id) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  Rccl r;
  KSH_TRY(load_rccl(&r));
  UniqueId u;
  const int rc = r.get_unique_id(&u);
  if (rc != 0) return fail(KSH_INTERNAL, "ncclGetUniqueId failed: %s", r.error_string ? r.error_string(rc) : "");
  static_assert(sizeof(UniqueId) == KSH_COMM_ID_BYTES, "id size");
  std::memcpy(id, u.internal, KSH_COMM_ID_BYTES);
  return KSH_OK;
}

int ksh_comm_create_rccl(ksh_ctx* ctx, int32_t rank, int32_t world, const unsigned char id[KSH_COMM_ID_BYTES],
                         ksh_comm** out) {
  if (!ctx || !id || !out) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(KSH_INVALID_ARGUMENT, "bad rank / world");
  *out = nullptr;
  KSH_HIP(hipSetDevice(ctx->device));
  ksh_comm* c = new ksh_comm;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  int rc = load_rccl(&c->rccl);
  if (rc != KSH_OK) {
    delete c;
    return rc;
  }
  UniqueId u;
  std::memcpy(u.internal, id, KSH_COMM_ID_BYTES);
  const int nrc = c->rccl.comm_init_rank(&c->nccl, world, u, rank);
  if (nrc != 0) {
    const char* msg = c->rccl.error_string ? c->rccl.error_string(nrc) : "RCCL error";
    delete c;
    return fail(KSH_INTERNAL, "ncclCommInitRank failed: %s", msg);
  }
  // the side channel's communicator: rank 0 draws its id, an all-gather on the first one carries it
  {
    auto cleanup = [&](int code, const char* what) {
      if (c->nccl_side) (void)c->rccl.comm_destroy(c->nccl_side);
      if (c->side) (void)hipStreamDestroy(c->side);
      if (c->side_ready) (void)hipEventDestroy(c->side_ready);
      if (c->side_done) (void)hipEventDestroy(c->side_done);
      (void)c->rccl.comm_destroy(c->nccl);
      delete c;
      return fail(code, "side communicator: %s", what);
    };
    UniqueId mine;
    std::memset(&mine, 0, sizeof(mine));
    if (rank == 0 && c->rccl.get_unique_id(&mine) != 0) return cleanup(KSH_INTERNAL, "ncclGetUniqueId failed");
    unsigned char *d_one = nullptr, *d_all = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&d_one), sizeof(UniqueId)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&d_all), sizeof(UniqueId) * size_t(world)) != hipSuccess)
      return cleanup(KSH_INTERNAL, "hipMalloc failed");
    UniqueId first;
    bool ok = hipMemcpyAsync(d_one, &mine, sizeof(UniqueId), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
              c->rccl.all_gather(d_one, d_all, sizeof(UniqueId), 0, c->nccl, ctx->stream) == 0 &&
              hipMemcpyAsync(&first, d_all, sizeof(UniqueId), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
              hipStreamSynchronize(ctx->stream) == hipSuccess;
    (void)hipFree(d_one);
    (void)hipFree(d_all);
    if (!ok) return cleanup(KSH_INTERNAL, "the exchange of its id failed");
    if (c->rccl.comm_init_rank(&c->nccl_side, world, first, rank) != 0) return cleanup(KSH_INTERNAL, "ncclCommInitRank failed");
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->side_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->side_done, hipEventDisableTiming) != hipSuccess)
      return cleanup(KSH_INTERNAL, "stream / event creation failed");
  }
  *out = c;
  return KSH_OK;
}

int ksh_comm_create_custom(ksh_ctx* ctx, int32_t rank, int32_t world, const ksh_comm_fns* fns, ksh_comm** out) {
  if (!ctx || !out) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(KSH_INVALID_ARGUMENT, "bad rank / world");
  if (fns && (fns->struct_size < offsetof(ksh_comm_fns, abort) || fns->struct_size > 4096))
    return fail(KSH_INVALID_ARGUMENT, "ksh_comm_fns::struct_size = %zu: set it to sizeof(ksh_comm_fns)", fns->struct_size);
  if (world > 1 && (!fns || !fns->allgather || !fns->send || !fns->recv))
    return fail(KSH_INVALID_ARGUMENT, "a custom transport needs allgather, send and recv");
  ksh_comm* c = new ksh_comm;
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->custom = true;
  if (fns) {  // (only what the caller's struct holds: members past its struct_size stay NULL)
    std::memcpy(&c->fns, fns, std::min(fns->struct_size, sizeof(ksh_comm_fns)));
    c->fns.struct_size = sizeof(ksh_comm_fns);
  }
  *out = c;
  return KSH_OK;
}

/* Collective (every rank calls it): each rank's id goes round through the transport's all-gather, on
 * device buffers; *n_ranks = the distinct ids that arrived here.  What a benchmark line quotes as the
 * number of ranks that took part (bench.py: multi_gpu.ranks_seen). */
int ksh_comm_ranks_seen(ksh_comm* c, int32_t* n_ranks) {
  if (!c || !n_ranks) return fail(KSH_INVALID_ARGUMENT, "NULL argument");
  *n_ranks = 0;
  KSH_HIP(hipSetDevice(c->ctx->device));
  int64_t *d_one = nullptr, *d_all = nullptr;
  KSH_HIP(hipMalloc(reinterpret_cast<void**>(&d_one), 8));
  KSH_HIP(hipMalloc(reinterpret_cast<void**>(&d_all), 8 * size_t(c->world)));
  const int64_t mine = c->rank;
  std::vector<int64_t> all(size_t(c->world), -1);
  int rc = KSH_OK;
  if (hipMemcpyAsync(d_one, &mine, 8, hipMemcpyHostToDevice, c->ctx->stream) != hipSuccess ||
      hipMemsetAsync(d_all, 0xFF, 8 * size_t(c->world), c->ctx->stream) != hipSuccess)
    rc = fail(KSH_INTERNAL, "ksh_comm_ranks_seen: staging failed");
  if (rc == KSH_OK) rc = comm_allgather(c, d_one, d_all, 8);
  if (rc == KSH_OK && (hipMemcpyAsync(all.data(), d_all, 8 * size_t(c->world), hipMemcpyDeviceToHost, c->ctx->stream) != hipSuccess ||
                       hipStreamSynchronize(c->ctx->stream) != hipSuccess))
    rc = fail(KSH_INTERNAL, "ksh_comm_ranks_seen: read-back failed");
  (void)hipFree(d_one);
  (void)hipFree(d_all);
  if (rc != KSH_OK) return rc;
  std::vector<bool> seen(size_t(c->world), false);
  for (int64_t r : all)
    if (r >= 0 && r < c->world && !seen[size_t(r)]) {
      seen[size_t(r)] = true;
      (*n_ranks)++;
    }
  return KSH_OK;
}

int ksh_comm_destroy(ksh_comm* c) {
  if (!c) return KSH_OK;
  if (c->side) {
    (void)hipStreamSynchronize(c->side);
    if (c->nccl_side) (void)c->rccl.comm_destroy(c->nccl_side);
    (void)hipStreamDestroy(c->side);
    (void)hipEventDestroy(c->side_ready);
    (void)hipEventDestroy(c->side_done);
  }
  if (c->nccl) {
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)c->rccl.comm_destroy(c->nccl);
  }
  delete c;
  return KSH_OK;
}

}  // extern "C"
