// RCCL collectives behind the C-ABI (dvae_comm_{init,allreduce,allgather,reducescatter,destroy}).
//
// The reference is single-process; data parallelism over the GPUs of one node is new here: one
// process per GPU, gradients of the flat parameter arena sum-all-reduced over xGMI, the latents all-gathered for the
// global B x B estimator, its column gradients reduce-scattered back.  These wrappers enqueue the RCCL collective on the
// caller's HIP stream -- ordered with the kernels around it, no host synchronisation, no torch types -- so a training
// iteration with collectives is still one stream of C-ABI calls (and can be re-issued from a recorded launch plan).
//
// librccl is loaded at run time (dlopen) the first time a communicator is created: libdvae_hip.so itself has no
// link-time dependency on it, and a single-GPU process never touches it.  Search order: $DVAE_RCCL_LIB, the path given
// to dvae_comm_load(), "librccl.so" (loader path), /opt/rocm/lib/librccl.so.
#include <dlfcn.h>
#include <mutex>
#include "common.h"

namespace dvae {

// the handful of RCCL entry points used, with their rccl.h prototypes (ncclResult_t = int, ncclComm_t = void*,
// ncclDataType_t ncclFloat = 7, ncclRedOp_t ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef int (*fn_GetUniqueId)(UniqueId*);
typedef int (*fn_CommInitRank)(void**, int, UniqueId, int);
typedef int (*fn_CommDestroy)(void*);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_AllGather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_Group)(void);
typedef const char* (*fn_GetErrorString)(int);

static struct Rccl {
  void* h = nullptr;
  fn_GetUniqueId GetUniqueId; fn_CommInitRank CommInitRank; fn_CommDestroy CommDestroy; fn_AllReduce AllReduce;
  fn_AllGather AllGather; fn_ReduceScatter ReduceScatter; fn_Broadcast Broadcast; fn_Group GroupStart, GroupEnd;
  fn_GetErrorString GetErrorString;
} g_rccl;
static std::mutex g_rccl_mu;
static const int kFloat = 7, kSum = 0;

static int rccl_load(const char* path) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return 0;
  const char* cands[4] = {getenv("DVAE_RCCL_LIB"), path, "librccl.so", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (int i = 0; i < 4 && !h; ++i)
    if (cands[i] && cands[i][0]) h = dlopen(cands[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) { set_error("dvae_comm: cannot load librccl (%s)", dlerror()); return -3; }
#define DVAE_SYM(field, name)                                                            \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                                 \
  if (!g_rccl.field) { set_error("dvae_comm: %s not found in librccl", name); dlclose(h); return -3; }
  DVAE_SYM(GetUniqueId, "ncclGetUniqueId") DVAE_SYM(CommInitRank, "ncclCommInitRank") DVAE_SYM(CommDestroy, "ncclCommDestroy")
  DVAE_SYM(AllReduce, "ncclAllReduce") DVAE_SYM(AllGather, "ncclAllGather") DVAE_SYM(ReduceScatter, "ncclReduceScatter")
  DVAE_SYM(Broadcast, "ncclBroadcast") DVAE_SYM(GroupStart, "ncclGroupStart") DVAE_SYM(GroupEnd, "ncclGroupEnd")
  DVAE_SYM(GetErrorString, "ncclGetErrorString")
#undef DVAE_SYM
  g_rccl.h = h;
  return 0;
}

static int rccl_check(int rc, const char* what) {
  if (rc == 0) return 0;
  set_error("dvae_comm: %s failed: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return -4;
}

}  // namespace dvae

using namespace dvae;

struct dvae_comm { void* nccl; int world, rank; };

extern "C" {

int dvae_comm_load(const char* librccl_path) { return rccl_load(librccl_path); }

int dvae_comm_unique_id(void* id128) {
  DVAE_CHECK_ARG(id128);
  if (int r = rccl_load(nullptr)) return r;
  return rccl_check(g_rccl.GetUniqueId((UniqueId*)id128), "ncclGetUniqueId");
}

int dvae_comm_init(dvae_comm** comm, const void* id128, int world, int rank) {
  DVAE_CHECK_ARG(comm && id128 && world >= 1 && rank >= 0 && rank < world);
  if (int r = rccl_load(nullptr)) return r;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void* c = nullptr;
  if (int r = rccl_check(g_rccl.CommInitRank(&c, world, id, rank), "ncclCommInitRank")) return r;
  *comm = new dvae_comm{c, world, rank};
  return 0;
}

int dvae_comm_destroy(dvae_comm* comm) {
  if (!comm) return 0;
  int r = rccl_check(g_rccl.CommDestroy(comm->nccl), "ncclCommDestroy");
  delete comm;
  return r;
}

int dvae_comm_world(const dvae_comm* comm) { return comm ? comm->world : 0; }
int dvae_comm_rank(const dvae_comm* comm) { return comm ? comm->rank : -1; }

int dvae_comm_allreduce(dvae_comm* comm, float* buf, long n, void* stream) {
  DVAE_CHECK_ARG(comm && buf && n > 0);
  return rccl_check(g_rccl.AllReduce(buf, buf, (size_t)n, kFloat, kSum, comm->nccl, (hipStream_t)stream), "ncclAllReduce");
}

int dvae_comm_allgather(dvae_comm* comm, const float* send, float* recv, long n_per_rank, void* stream) {
  DVAE_CHECK_ARG(comm && send && recv && n_per_rank > 0);
  return rccl_check(g_rccl.AllGather(send, recv, (size_t)n_per_rank, kFloat, comm->nccl, (hipStream_t)stream), "ncclAllGather");
}

int dvae_comm_reducescatter(dvae_comm* comm, const float* send, float* recv, long n_per_rank, void* stream) {
  DVAE_CHECK_ARG(comm && send && recv && n_per_rank > 0);
  return rccl_check(g_rccl.ReduceScatter(send, recv, (size_t)n_per_rank, kFloat, kSum, comm->nccl, (hipStream_t)stream),
                    "ncclReduceScatter");
}

int dvae_comm_broadcast(dvae_comm* comm, float* buf, long n, int root, void* stream) {
  DVAE_CHECK_ARG(comm && buf && n > 0 && root >= 0 && root < comm->world);
  return rccl_check(g_rccl.Broadcast(buf, buf, (size_t)n, kFloat, root, comm->nccl, (hipStream_t)stream), "ncclBroadcast");
}

int dvae_comm_group_start(void) { return g_rccl.h ? rccl_check(g_rccl.GroupStart(), "ncclGroupStart") : -3; }
int dvae_comm_group_end(void) { return g_rccl.h ? rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd") : -3; }

}  // extern "C"
