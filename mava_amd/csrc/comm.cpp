// The exchange step of the data-parallel path through the C ABI alone: a thin RCCL communicator handle.
//
// Replaces the four jax.lax.pmean calls per minibatch of mava/systems/ppo/ff_mappo.py:224-238 (and the identical lines
// of the other three systems) for a host that does not carry torch.distributed: one process per GPU creates a
// communicator from a shared 128-byte unique id, sums its flat [actor grads | critic grads | loss scalars] buffer over
// the ranks in place (RCCL over xGMI: ring / tree chosen by RCCL; the buffer is ~0.3 MB, latency-bound, so ONE flat
// message per network is the design point) and hands 1 / (U * D) to mava_clip_adam_f32 as grad_scale.
//
// librccl.so is loaded at the first call (dlopen), not linked: a host that already carries an RCCL (PyTorch-ROCm ships
// its own) gets that very library instead of a second copy, and a single-GPU host never needs it.
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

// the few RCCL declarations this file needs (rccl.h: ncclUniqueId is 128 opaque bytes; enums as of NCCL 2.x)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t);
typedef int (*BroadcastFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;

struct Api {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  BroadcastFn broadcast = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn error_string = nullptr;
};
Api g_api;

int load_api() {
  if (g_api.lib != nullptr) return MAVA_OK;
  const char* names[] = {getenv("MAVA_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (n == nullptr || n[0] == 0) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib != nullptr) break;
  }
  if (lib == nullptr) {
    mava_set_error("mava_comm: librccl.so could not be loaded (%s); set MAVA_RCCL_LIB", dlerror());
    return MAVA_EARG(20);
  }
  Api a;
  a.lib = lib;
  a.get_unique_id = (GetUniqueIdFn)dlsym(lib, "ncclGetUniqueId");
  a.comm_init_rank = (CommInitRankFn)dlsym(lib, "ncclCommInitRank");
  a.all_reduce = (AllReduceFn)dlsym(lib, "ncclAllReduce");
  a.broadcast = (BroadcastFn)dlsym(lib, "ncclBroadcast");
  a.comm_destroy = (CommDestroyFn)dlsym(lib, "ncclCommDestroy");
  a.error_string = (GetErrorStringFn)dlsym(lib, "ncclGetErrorString");
  if (!a.get_unique_id || !a.comm_init_rank || !a.all_reduce || !a.broadcast || !a.comm_destroy) {
    mava_set_error("mava_comm: the loaded librccl.so lacks an expected symbol");
    return MAVA_EARG(21);
  }
  g_api = a;
  return MAVA_OK;
}

int rccl_fail(const char* what, int rc) {
  mava_set_error("%s: RCCL error %d (%s)", what, rc, g_api.error_string ? g_api.error_string(rc) : "?");
  return -2000 - rc;
}

struct Handle {
  Comm comm;
  int rank, world;
};

}  // namespace

extern "C" int mava_comm_unique_id(uint8_t* id128) {
  MAVA_ARG_CHECK(id128 != nullptr, 0, "mava_comm_unique_id: null output");
  const int rc0 = load_api();
  if (rc0 != MAVA_OK) return rc0;
  UniqueId id;
  const int rc = g_api.get_unique_id(&id);
  if (rc != 0) return rccl_fail("mava_comm_unique_id", rc);
  memcpy(id128, id.internal, 128);
  return MAVA_OK;
}

extern "C" int mava_comm_create(void** h, int rank, int world, const uint8_t* id128) {
  MAVA_ARG_CHECK(h != nullptr && id128 != nullptr, 0, "mava_comm_create: null argument");
  MAVA_ARG_CHECK(world >= 1 && rank >= 0 && rank < world, 1, "mava_comm_create: rank %d of %d", rank, world);
  *h = nullptr;
  const int rc0 = load_api();
  if (rc0 != MAVA_OK) return rc0;
  UniqueId id;
  memcpy(id.internal, id128, 128);
  Comm c = nullptr;
  const int rc = g_api.comm_init_rank(&c, world, id, rank);  // uses the calling thread's current HIP device
  if (rc != 0) return rccl_fail("mava_comm_create", rc);
  Handle* hd = new Handle{c, rank, world};
  *h = hd;
  return MAVA_OK;
}

extern "C" int mava_allreduce_sum_f32(void* h, float* buf, size_t n, hipStream_t s) {
  MAVA_ARG_CHECK(h != nullptr, 0, "mava_allreduce_sum_f32: null communicator");
  MAVA_ARG_CHECK(buf != nullptr || n == 0, 1, "mava_allreduce_sum_f32: null buffer");
  if (n == 0) return MAVA_OK;
  Handle* hd = static_cast<Handle*>(h);
  const int rc = g_api.all_reduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, hd->comm, s);
  if (rc != 0) return rccl_fail("mava_allreduce_sum_f32", rc);
  return MAVA_OK;
}

extern "C" int mava_broadcast_f32(void* h, float* buf, size_t n, int root, hipStream_t s) {
  MAVA_ARG_CHECK(h != nullptr, 0, "mava_broadcast_f32: null communicator");
  Handle* hd = static_cast<Handle*>(h);
  MAVA_ARG_CHECK((buf != nullptr || n == 0) && root >= 0 && root < hd->world, 1, "mava_broadcast_f32: bad arguments");
  if (n == 0) return MAVA_OK;
  const int rc = g_api.broadcast(buf, buf, n, NCCL_FLOAT32, root, hd->comm, s);
  if (rc != 0) return rccl_fail("mava_broadcast_f32", rc);
  return MAVA_OK;
}

extern "C" int mava_comm_destroy(void* h) {
  if (h == nullptr) return MAVA_OK;
  Handle* hd = static_cast<Handle*>(h);
  int rc = 0;
  if (g_api.comm_destroy != nullptr) rc = g_api.comm_destroy(hd->comm);
  delete hd;
  if (rc != 0) return rccl_fail("mava_comm_destroy", rc);
  return MAVA_OK;
}
