// Collectives of the C ABI: a thin RCCL binding so that a level-B integration
// (ctypes from the reference's own classes) can run the
// per-minibatch gradient all-reduce without torch.distributed.
//
// Reference: the reduction this replaces is a serial loop of peer copies and
// adds onto one device (src/tike/communicators/pool.py:300-395 reduce_gpu /
// allreduce, composed in comm.py:96-136).  Here it is one in-place
// ncclAllReduce(sum) over xGMI on the caller's stream.
//
// librccl.so.1 is bound at the first call, not at link time: a process that
// already holds an RCCL (PyTorch-ROCm ships one) keeps exactly that copy, and
// single-GPU users need no RCCL at all.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>

#include "internal.h"
#include "tike_amd.h"

namespace {

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  bool ok = false;
};

template <typename F>
bool bind(void* h, const char* name, F& f) {
  f = reinterpret_cast<F>(dlsym(h, name));
  return f != nullptr;
}

const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return x;
    x.ok = bind(h, "ncclGetUniqueId", x.GetUniqueId) &
           bind(h, "ncclCommInitRank", x.CommInitRank) &
           bind(h, "ncclCommDestroy", x.CommDestroy) & bind(h, "ncclAllReduce", x.AllReduce) &
           bind(h, "ncclBroadcast", x.Broadcast);
    return x;
  }();
  return r;
}

// ncclResult_t -> the ABI's return convention (0 ok; anything else is an error
// the Python layer raises RuntimeError from)
inline int rc(ncclResult_t e) { return e == ncclSuccess ? 0 : TIKE_ERR_COMM + (int)e; }

}  // namespace

extern "C" int tike_comm_unique_id(void* id) {
  if (!id) return TIKE_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return TIKE_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == TIKE_COMM_ID_BYTES, "id size");
  ncclUniqueId u;
  const int e = rc(r.GetUniqueId(&u));
  if (e) return e;
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int tike_comm_create(const void* id, int nranks, int rank, void** comm) {
  if (!id || !comm || nranks < 1 || rank < 0 || rank >= nranks) return TIKE_ERR_ARG;
  const Rccl& r = rccl();
  if (!r.ok) return TIKE_ERR_UNSUPPORTED;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  ncclComm_t c = nullptr;
  const int e = rc(r.CommInitRank(&c, nranks, u, rank));
  if (e) return e;
  *comm = c;
  return 0;
}

extern "C" int tike_comm_destroy(void* comm) {
  if (!comm) return 0;
  const Rccl& r = rccl();
  if (!r.ok) return TIKE_ERR_UNSUPPORTED;
  return rc(r.CommDestroy(static_cast<ncclComm_t>(comm)));
}

extern "C" int tike_comm_allreduce_sum(void* comm, void* buf, long count, int f64, void* stream) {
  if (!comm || count < 0 || (count > 0 && !buf)) return TIKE_ERR_ARG;
  if (count == 0) return 0;
  const Rccl& r = rccl();
  if (!r.ok) return TIKE_ERR_UNSUPPORTED;
  return rc(r.AllReduce(buf, buf, (size_t)count, f64 ? ncclFloat64 : ncclFloat32, ncclSum,
                        static_cast<ncclComm_t>(comm), static_cast<hipStream_t>(stream)));
}

extern "C" int tike_comm_broadcast(void* comm, void* buf, long nbytes, int root, void* stream) {
  if (!comm || nbytes < 0 || (nbytes > 0 && !buf) || root < 0) return TIKE_ERR_ARG;
  if (nbytes == 0) return 0;
  const Rccl& r = rccl();
  if (!r.ok) return TIKE_ERR_UNSUPPORTED;
  return rc(r.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, static_cast<ncclComm_t>(comm),
                        static_cast<hipStream_t>(stream)));
}
