// Gradient exchange behind the C ABI (include/riders_hip.h: rd_comm_*): RCCL over xGMI, one process per GPU.
//
// Replaces the reference's torch.nn.DataParallel wrapping (RCNet/rcnet_model.py:259-265, val_zju.py:341: single process, weights broadcast
// every forward, outputs gathered on GPU 0) with what a one-process-per-GPU design needs: a sum over the ranks of slices of the flat fp32
// gradient arena, started while the backward is still running.
//
//   * RCCL is bound at RUN time (dlopen + dlsym): the copy already loaded into the process is preferred (a PyTorch-ROCm process carries its
//     own librccl.so; two RCCL builds in one address space must not be mixed through the global symbol table), then $RIDERS_RCCL_LIB, then
//     /opt/rocm/lib/librccl.so.  libriders_hip.so therefore has no link-time dependency on RCCL and single-GPU users never load it.
//   * The communicator owns ONE side stream and two events.  rd_allreduce_bucket forks the side stream behind everything queued so far on
//     the caller's compute stream (event record + stream wait) and enqueues the collective there; rd_comm_join makes the compute stream wait
//     for everything issued since the last join.  Both are plain stream operations, so they are legal inside a stream capture: the fork
//     pulls the side stream into the capture and the join merges it back -- forward + backward + every bucket's collective become ONE
//     hipGraph (round 4 needed one graph per stage because torch.distributed's collectives cannot be captured with external events).
//   * xGMI is point-to-point (7 links per GPU): mode 1 issues a bucket as reduce-scatter + all-gather in place (each rank owns the r-th
//     piece; same bytes per link as a ring all-reduce, but the gather of bucket k can overlap the scatter of bucket k + 1), mode 0 leaves
//     the algorithm to RCCL's all-reduce.
//
// Error convention: 0 = ok, negative = argument error, positive = hipError_t / 1000 + ncclResult_t; text in rd_last_error_string().
#include "../../include/riders_hip.h"
#include <hip/hip_runtime.h>      // (the host emulator's stand-in defines RD_EMU)
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" void rd_set_last_error(const char* msg);      // rd_api.cpp: the thread-local message behind rd_last_error_string()

namespace {
int cfail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  rd_set_last_error(buf);
  return code;
}
}  // namespace

#ifdef RD_EMU
// ---- host emulator build (tests/emu): the same entry points over host memory, so that everything ABOVE the transport (bucketing, stage
// hooks, the single-graph step, the rendezvous, both bucket modes) runs on the GPU-less build container -- with world = 1 as a loop-back,
// with world > 1 (one PROCESS per rank, as on the GPUs) through a POSIX shared-memory segment named after the rendezvous id: every rank
// publishes its slice, the ranks meet at a sense-reversing barrier, and each sums the published slices in rank order (bit-identical on every
// rank).  Mode 1 really is reduce-scatter (rank r sums piece r only) followed by all-gather (the pieces are copied back), so the two modes of
// rd_allreduce_bucket are different code here too.  Emulator "streams" are synchronous: a collective has finished when the call returns.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <time.h>

namespace {
constexpr int64_t EMU_SLOT_FLOATS = 1 << 22;      // 16 MiB per rank and exchange: larger buckets go through in pieces
struct EmuShared {
  std::atomic<int> arrived, sense, attached;
  int world;
};
}  // namespace
struct rd_comm_s {
  int rank, world; long issued, joined;
  EmuShared* sh; float* slots; size_t bytes; int local_sense; char name[64];
};

namespace {
int emu_barrier(rd_comm_s* c) {
  if (c->world == 1) return 0;
  c->local_sense ^= 1;
  if (c->sh->arrived.fetch_add(1) + 1 == c->world) { c->sh->arrived.store(0); c->sh->sense.store(c->local_sense); return 0; }
  timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
  while (c->sh->sense.load() != c->local_sense) {
    timespec t1; clock_gettime(CLOCK_MONOTONIC, &t1);
    if (t1.tv_sec - t0.tv_sec > 120) return cfail(1001, "emulator transport: a rank did not reach the barrier within 120 s");
    usleep(50);
  }
  return 0;
}
// sum (or copy) over the ranks of n floats at `buf`, through the shared slots, piecewise
int emu_exchange(rd_comm_s* c, float* buf, int64_t n, int kind, int root) {      // kind 0: all-reduce, 1: reduce-scatter + all-gather, 2: broadcast
  for (int64_t o = 0; o < n; o += EMU_SLOT_FLOATS) {
    const int64_t m = n - o < EMU_SLOT_FLOATS ? n - o : EMU_SLOT_FLOATS;
    float* mine = c->slots + (int64_t)c->rank * EMU_SLOT_FLOATS;
    if (kind != 2 || c->rank == root) memcpy(mine, buf + o, (size_t)m * 4);
    if (int rc = emu_barrier(c)) return rc;
    if (kind == 2) {
      memcpy(buf + o, c->slots + (int64_t)root * EMU_SLOT_FLOATS, (size_t)m * 4);
    } else if (kind == 0) {
      for (int64_t i = 0; i < m; i++) { float s = 0.f; for (int r = 0; r < c->world; r++) s += c->slots[(int64_t)r * EMU_SLOT_FLOATS + i]; buf[o + i] = s; }
    } else {
      // reduce-scatter: this rank owns piece `rank` of the body (NCCL's in-place convention), the tail that does not divide is all-reduced
      const int64_t per = m / c->world, body = per * c->world;
      float* out = c->slots + (int64_t)c->world * EMU_SLOT_FLOATS;      // the gathered result, one extra slot
      for (int64_t i = per * c->rank; i < per * (c->rank + 1); i++) { float s = 0.f; for (int r = 0; r < c->world; r++) s += c->slots[(int64_t)r * EMU_SLOT_FLOATS + i]; out[i] = s; }
      if (int rc = emu_barrier(c)) return rc;
      memcpy(buf + o, out, (size_t)body * 4);                          // all-gather
      for (int64_t i = body; i < m; i++) { float s = 0.f; for (int r = 0; r < c->world; r++) s += c->slots[(int64_t)r * EMU_SLOT_FLOATS + i]; buf[o + i] = s; }
    }
    if (int rc = emu_barrier(c)) return rc;      // nobody overwrites a slot that is still being read
  }
  return 0;
}
}  // namespace

extern "C" {
int rd_comm_available(void) { return 0; }
int rd_comm_unique_id(void* id128) {
  if (!id128) return cfail(-1, "comm_unique_id: null pointer");
  memset(id128, 0, 128);
  timespec t; clock_gettime(CLOCK_REALTIME, &t);
  snprintf((char*)id128, 64, "/riders_emu_%d_%lx%lx", (int)getpid(), (unsigned long)t.tv_sec, (unsigned long)t.tv_nsec);
  return 0;
}
int rd_comm_init(int32_t rank, int32_t world, const void* id128, void** comm) {
  if (!id128 || !comm) return cfail(-1, "comm_init: null pointer");
  if (world < 1 || rank < 0 || rank >= world) return cfail(-1, "comm_init: bad rank %d / world %d", rank, world);
  rd_comm_s* c = (rd_comm_s*)calloc(1, sizeof(rd_comm_s)); c->rank = rank; c->world = world;
  if (world > 1) {
    memcpy(c->name, id128, 63);
    if (c->name[0] != '/') { free(c); return cfail(-1, "comm_init: not an id of rd_comm_unique_id"); }
    c->bytes = 4096 + (size_t)(world + 1) * EMU_SLOT_FLOATS * 4;
    int fd = -1;
    if (rank == 0) {
      fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { if (fd >= 0) { close(fd); shm_unlink(c->name); } free(c); return cfail(1002, "emulator transport: shm_open / ftruncate failed"); }
    } else {
      for (int tries = 0; tries < 24000 && fd < 0; tries++) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(5000); }      // rank 0 creates it
      if (fd < 0) { free(c); return cfail(1002, "emulator transport: rank 0's segment did not appear"); }
      for (int tries = 0; tries < 24000; tries++) { off_t sz = lseek(fd, 0, SEEK_END); if (sz >= (off_t)c->bytes) break; usleep(5000); }
    }
    void* m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { if (rank == 0) shm_unlink(c->name); free(c); return cfail(1002, "emulator transport: mmap failed"); }
    c->sh = (EmuShared*)m; c->slots = (float*)((char*)m + 4096);
    if (rank == 0) c->sh->world = world;       // (a fresh segment is zero-filled: arrived = sense = attached = 0)
    c->sh->attached.fetch_add(1);
    for (int tries = 0; c->sh->attached.load() < world; tries++) {      // ncclCommInitRank likewise returns once every rank has entered
      if (tries > 24000) { munmap(m, c->bytes); if (rank == 0) shm_unlink(c->name); free(c); return cfail(1001, "emulator transport: not every rank attached"); }
      usleep(5000);
    }
    if (rank == 0) shm_unlink(c->name);      // every rank holds a mapping: the name can go (nothing is left behind in /dev/shm)
  }
  *comm = c; return 0;
}
int rd_comm_destroy(void* comm) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (c && c->sh) munmap((void*)c->sh, c->bytes);
  free(comm); return 0;
}
int rd_allreduce_bucket(void* comm, float* buf, int64_t n, int32_t mode, void*) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!comm || !buf || n < 0 || (mode != 0 && mode != 1)) return cfail(-1, "allreduce_bucket: bad arguments");
  if (n == 0) return 0;
  if (c->world > 1) { if (int rc = emu_exchange(c, buf, n, mode, 0)) return rc; }
  c->issued++; return 0;
}
int rd_comm_broadcast(void* comm, float* buf, int64_t n, int32_t root, void*) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!comm || !buf || n < 0 || root < 0 || root >= c->world) return cfail(-1, "comm_broadcast: bad arguments");
  if (n == 0) return 0;
  if (c->world > 1) { if (int rc = emu_exchange(c, buf, n, 2, root)) return rc; }
  c->issued++; return 0;
}
int rd_comm_join(void* comm, void*) { if (!comm) return cfail(-1, "comm_join: null communicator"); rd_comm_s* c = (rd_comm_s*)comm; c->joined = c->issued; return 0; }
int64_t rd_comm_pending(void* comm) { rd_comm_s* c = (rd_comm_s*)comm; return c ? c->issued - c->joined : -1; }
}
#else
#include <dlfcn.h>
#include <link.h>

namespace {
// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x): opaque communicator, 128-byte unique id, enums by value
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccessV = 0 };
enum { ncclFloat32V = 7 };      // ncclDataType_t: int8 0, uint8 1, int32 2, uint32 3, int64 4, uint64 5, float16 6, float32 7
enum { ncclSumV = 0 };          // ncclRedOp_t
struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;

// An RCCL build already mapped into this process, whatever its file name or directory (PyTorch-ROCm bundles its own under torch/lib, and a
// bare dlopen("librccl.so", RTLD_NOLOAD) only matches it when the soname agrees): the first loaded object whose path contains "librccl".
int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
  if (info->dlpi_name && strstr(info->dlpi_name, "librccl")) { snprintf((char*)out, 1024, "%s", info->dlpi_name); return 1; }
  return 0;
}

int load_rccl() {
  if (g_rccl.h) return 0;
  void* h = nullptr;
  char loaded[1024] = {0};
  const bool have = dl_iterate_phdr(find_loaded_rccl, loaded) != 0 && loaded[0];
  if (have) {
    // two RCCL builds in one address space must never be mixed: with one already mapped, THAT one is bound or nothing is
    h = dlopen(loaded, RTLD_NOW | RTLD_NOLOAD);
    if (!h) return cfail(-2, "rd_comm: this process already carries %s but it cannot be re-opened (%s); refusing to load a second RCCL", loaded, dlerror());
  } else {
    const char* env = getenv("RIDERS_RCCL_LIB");
    if (env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return cfail(-2, "rd_comm: cannot load librccl.so (%s)", dlerror());
  }
  Rccl r; r.h = h;
#define RD_SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) return cfail(-2, "rd_comm: librccl.so has no %s", name);
  RD_SYM(GetUniqueId, "ncclGetUniqueId") RD_SYM(CommInitRank, "ncclCommInitRank") RD_SYM(CommDestroy, "ncclCommDestroy")
  RD_SYM(AllReduce, "ncclAllReduce") RD_SYM(ReduceScatter, "ncclReduceScatter") RD_SYM(AllGather, "ncclAllGather")
  RD_SYM(Broadcast, "ncclBroadcast") RD_SYM(GroupStart, "ncclGroupStart") RD_SYM(GroupEnd, "ncclGroupEnd") RD_SYM(GetErrorString, "ncclGetErrorString")
#undef RD_SYM
  g_rccl = r;
  return 0;
}
int nfail(int rc, const char* what) { return cfail(1000 + rc, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"); }
int hfail(hipError_t e, const char* what) { return cfail((int)e, "%s: %s", what, hipGetErrorString(e)); }
#define RD_HIP(call, what) { hipError_t e_ = (call); if (e_ != hipSuccess) return hfail(e_, what); }
#define RD_NCCL(call, what) { int rc_ = (call); if (rc_ != ncclSuccessV) return nfail(rc_, what); }
}  // namespace

struct rd_comm_s {
  ncclComm_t comm;
  int rank, world, device;
  hipStream_t side;           // library-owned communication stream
  hipEvent_t fork, join;      // compute -> side, side -> compute
  long issued, joined;
};

extern "C" {

int rd_comm_unique_id(void* id128) {
  if (!id128) return cfail(-1, "comm_unique_id: null pointer");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  RD_NCCL(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(id128, id.internal, 128);
  return 0;
}

int rd_comm_init(int32_t rank, int32_t world, const void* id128, void** comm) {
  if (!id128 || !comm) return cfail(-1, "comm_init: null pointer");
  if (world < 1 || rank < 0 || rank >= world) return cfail(-1, "comm_init: bad rank %d / world %d", rank, world);
  if (int rc = load_rccl()) return rc;
  rd_comm_s* c = (rd_comm_s*)calloc(1, sizeof(rd_comm_s));
  c->rank = rank; c->world = world;
  { hipError_t e_ = hipGetDevice(&c->device); if (e_ != hipSuccess) { free(c); return hfail(e_, "hipGetDevice"); } }
  ncclUniqueId id; memcpy(id.internal, id128, 128);
  { int rc_ = g_rccl.CommInitRank(&c->comm, world, id, rank); if (rc_ != ncclSuccessV) { free(c); return nfail(rc_, "ncclCommInitRank"); } }
  // from here on every error path gives back what was created (communicator, stream, events, the struct)
  hipError_t e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
  const char* what = "hipStreamCreate";
  if (e == hipSuccess) { e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming); what = "hipEventCreate(fork)"; }
  if (e == hipSuccess) { e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming); what = "hipEventCreate(join)"; }
  if (e != hipSuccess) {
    if (c->join) hipEventDestroy(c->join);
    if (c->fork) hipEventDestroy(c->fork);
    if (c->side) hipStreamDestroy(c->side);
    g_rccl.CommDestroy(c->comm);
    free(c);
    return hfail(e, what);
  }
  *comm = c;
  return 0;
}

int rd_comm_available(void) { return load_rccl(); }

int rd_comm_destroy(void* comm) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!c) return 0;
  hipStreamSynchronize(c->side);
  if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  hipEventDestroy(c->fork); hipEventDestroy(c->join);
  hipStreamDestroy(c->side);
  free(c);
  return 0;
}

static int fork_side(rd_comm_s* c, hipStream_t compute) {
  RD_HIP(hipEventRecord(c->fork, compute), "hipEventRecord(fork)");
  RD_HIP(hipStreamWaitEvent(c->side, c->fork, 0), "hipStreamWaitEvent(side)");
  return 0;
}

int rd_allreduce_bucket(void* comm, float* buf, int64_t n, int32_t mode, void* compute_stream) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!c || !buf || n < 0 || (mode != 0 && mode != 1)) return cfail(-1, "allreduce_bucket: bad arguments");
  if (n == 0) return 0;
  if (int rc = fork_side(c, (hipStream_t)compute_stream)) return rc;
  const int64_t per = n / c->world, body = per * c->world;
  if (mode == 1 && per > 0 && c->world > 1) {
    // in place, NCCL's convention: rank r's shard is the r-th piece of the bucket
    float* shard = buf + per * c->rank;
    RD_NCCL(g_rccl.ReduceScatter(buf, shard, (size_t)per, ncclFloat32V, ncclSumV, c->comm, c->side), "ncclReduceScatter");
    RD_NCCL(g_rccl.AllGather(shard, buf, (size_t)per, ncclFloat32V, c->comm, c->side), "ncclAllGather");
    if (body < n) RD_NCCL(g_rccl.AllReduce(buf + body, buf + body, (size_t)(n - body), ncclFloat32V, ncclSumV, c->comm, c->side), "ncclAllReduce(tail)");
  } else {
    RD_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32V, ncclSumV, c->comm, c->side), "ncclAllReduce");
  }
  c->issued++;
  return 0;
}

int rd_comm_broadcast(void* comm, float* buf, int64_t n, int32_t root, void* compute_stream) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!c || !buf || n < 0 || root < 0 || root >= c->world) return cfail(-1, "comm_broadcast: bad arguments");
  if (n == 0) return 0;
  if (int rc = fork_side(c, (hipStream_t)compute_stream)) return rc;
  RD_NCCL(g_rccl.Broadcast(buf, buf, (size_t)n, ncclFloat32V, root, c->comm, c->side), "ncclBroadcast");
  c->issued++;
  return 0;
}

int rd_comm_join(void* comm, void* compute_stream) {
  rd_comm_s* c = (rd_comm_s*)comm;
  if (!c) return cfail(-1, "comm_join: null communicator");
  if (c->issued == c->joined) return 0;      // nothing in flight (an event that was never recorded must not be waited on inside a capture)
  RD_HIP(hipEventRecord(c->join, c->side), "hipEventRecord(join)");
  RD_HIP(hipStreamWaitEvent((hipStream_t)compute_stream, c->join, 0), "hipStreamWaitEvent(compute)");
  c->joined = c->issued;
  return 0;
}

int64_t rd_comm_pending(void* comm) { rd_comm_s* c = (rd_comm_s*)comm; return c ? c->issued - c->joined : -1; }

}  // extern "C"
#endif
