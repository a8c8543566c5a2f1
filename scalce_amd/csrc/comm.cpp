// comm.cpp -- the collectives a sharded run needs, over device buffers.
//
// The reference has no distributed mode; what it carries ACROSS reads -- bin_size of the tie-break
// (reads.cpp:246,420), prev[] and the 80^3 counters of the quality model (qualities.cpp:179-198), the running record size
// of the spill rule (compress.cpp:702-715), the 10 MiB cuts of the coder on the reordered stream (arithmetic.cpp:318-363)
// -- is what ranks have to exchange when the read stream is split over GPUs.  Two transports:
//   rccl  one process per GPU, RCCL over xGMI: ncclAllGather / ncclAllReduce / grouped ncclSend + ncclRecv, enqueued on
//         the caller's stream.  librccl is opened at run time (dlopen): inside a PyTorch process that is the copy torch
//         has already loaded, so there is one RCCL per process.
//   shm   several processes sharing ONE GPU (or none): POSIX shared memory and a process-shared barrier, device buffers
//         staged through the host.  This is the rehearsal transport of the tests, not a product path.
#include <algorithm>
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"

namespace {

// ---- the few RCCL entry points used.  Types, enum values and prototypes are <rccl/rccl.h>'s own (decltype of the declared
// functions); only the ADDRESSES are resolved at run time, so that inside a PyTorch process the one RCCL torch has loaded
// is the one that runs.
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
  bool load() {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { error = std::string("cannot open librccl: ") + dlerror(); return false; }
    auto sym = [&](const char *n) { void *p = dlsym(lib, n); if (!p) error = std::string("librccl lacks ") + n; return p; };
    GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
    CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
    Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    return error.empty();
  }
};
Rccl g_rccl;

// ---- shared-memory transport ----------------------------------------------------------------------------------
struct ShmHeader {
  pthread_barrier_t barrier;
  uint64_t ready;      // set by rank 0 once the barrier is initialised
  uint64_t slot_bytes;
  uint64_t sizes[64][64];  // all_to_all_v: sizes[src][dst]
  uint64_t box_state[64];  // point-to-point mailbox of every receiver: 0 = empty, else bytes + 1 of the message in it
};
// one message per receiver (behind the slots): 64 MiB holds the tie-break chain's message of the largest core table the
// reference's loader admits (5 M cores: 40 MB of counts, reads.cpp:336)
constexpr size_t SHM_BOX_BYTES = 64u << 20;
constexpr long SHM_SPINS = 6000000;          // x 100 us = ten minutes: the ranks in front may be settling 50 M-read shards

}  // namespace

struct scalce_comm {
  int world = 1, rank = 0, device = 0;
  std::string err;
  // rccl
  ncclComm_t nccl = nullptr;
  // shm
  bool shm = false;
  std::string shm_name;
  ShmHeader *hdr = nullptr;
  uint8_t *slots = nullptr;  // world x slot_bytes
  size_t map_bytes = 0;
  uint64_t slot_bytes = 0;
  std::vector<uint8_t> stage;
  uint64_t piece_bytes = 1ull << 30;  // largest single ncclSend / ncclRecv (scalce_comm_set_piece_bytes)
};

#define CM_HIP(c, expr)                                                                                   \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) { (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_); return SCALCE_ERR_HIP; } \
  } while (0)
// the rehearsal transport also runs without any GPU (device < 0: the "device" buffers are host memory) -- CPU tests of the
// collectives' semantics across processes
static hipError_t cm_copy(const scalce_comm *c, void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s);
#define CM_NCCL(c, expr)                                                                                               \
  do {                                                                                                                 \
    ncclResult_t r_ = (expr);                                                                                          \
    if (r_ != ncclSuccess) { (c)->err = std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); return SCALCE_ERR_HIP; } \
  } while (0)

static hipError_t cm_copy(const scalce_comm *c, void *dst, const void *src, size_t n, hipMemcpyKind kind, hipStream_t s) {
  if (c->device < 0) { memcpy(dst, src, n); return hipSuccess; }
  return hipMemcpyAsync(dst, src, n, kind, s);
}
static hipError_t cm_sync(const scalce_comm *c, hipStream_t s) { return c->device < 0 ? hipSuccess : hipStreamSynchronize(s); }

extern "C" const char *scalce_comm_error(const scalce_comm *c) { return c ? c->err.c_str() : "null communicator"; }
extern "C" int scalce_comm_world(const scalce_comm *c) { return c ? c->world : 1; }
extern "C" void scalce_comm_set_piece_bytes(scalce_comm *c, uint64_t bytes) { if (c && bytes) c->piece_bytes = bytes; }
extern "C" int scalce_comm_rank(const scalce_comm *c) { return c ? c->rank : 0; }

extern "C" int scalce_comm_unique_id(uint8_t id[SCALCE_COMM_ID_BYTES]) {
  if (!id || !g_rccl.load()) return SCALCE_ERR_HIP;
  ncclUniqueId u;
  if (g_rccl.GetUniqueId(&u) != ncclSuccess) return SCALCE_ERR_HIP;
  static_assert(sizeof u == SCALCE_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id, &u, sizeof u);
  return SCALCE_OK;
}

extern "C" int scalce_comm_create_rccl(int device, int world, int rank, const uint8_t id[SCALCE_COMM_ID_BYTES], scalce_comm **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return SCALCE_ERR_ARG;
  scalce_comm *c = new scalce_comm();
  c->world = world; c->rank = rank; c->device = device;
  *out = c;
  if (!g_rccl.load()) { c->err = g_rccl.error; return SCALCE_ERR_HIP; }
  CM_HIP(c, hipSetDevice(device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  CM_NCCL(c, g_rccl.CommInitRank(&c->nccl, world, u, rank));
  int n = 0;
  CM_NCCL(c, g_rccl.CommCount(c->nccl, &n));
  if (n != world) { c->err = "RCCL reports a different world size"; return SCALCE_ERR_HIP; }
  return SCALCE_OK;
}

extern "C" int scalce_comm_create_shm(int device, int world, int rank, const char *name, uint64_t slot_bytes, scalce_comm **out) {
  if (!out || !name || world < 1 || world > 64 || rank < 0 || rank >= world) return SCALCE_ERR_ARG;
  scalce_comm *c = new scalce_comm();
  c->world = world; c->rank = rank; c->device = device; c->shm = true;
  c->shm_name = name;
  c->slot_bytes = slot_bytes ? slot_bytes : (64ull << 20);
  *out = c;
  c->map_bytes = sizeof(ShmHeader) + (size_t)world * c->slot_bytes + (size_t)world * SHM_BOX_BYTES;
  int fd = -1;
  if (rank == 0) {
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { c->err = "shm_open / ftruncate failed"; return SCALCE_ERR_HIP; }
  } else {
    for (int tries = 0; tries < 30000 && fd < 0; tries++) {  // wait for rank 0 to create and size it
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < c->map_bytes)) { close(fd); fd = -1; }
      if (fd < 0) usleep(1000);
    }
    if (fd < 0) { c->err = "shared segment of rank 0 did not appear"; return SCALCE_ERR_HIP; }
  }
  void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { c->err = "mmap of the shared segment failed"; return SCALCE_ERR_HIP; }
  c->hdr = static_cast<ShmHeader *>(p);
  c->slots = static_cast<uint8_t *>(p) + sizeof(ShmHeader);
  if (rank == 0) {
    pthread_barrierattr_t a;
    pthread_barrierattr_init(&a);
    pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
    pthread_barrier_init(&c->hdr->barrier, &a, (unsigned)world);
    pthread_barrierattr_destroy(&a);
    c->hdr->slot_bytes = c->slot_bytes;
    for (int r = 0; r < 64; r++) c->hdr->box_state[r] = 0;
    __atomic_store_n(&c->hdr->ready, 1ull, __ATOMIC_RELEASE);
  } else {
    for (int tries = 0; tries < 30000 && !__atomic_load_n(&c->hdr->ready, __ATOMIC_ACQUIRE); tries++) usleep(1000);
    if (!__atomic_load_n(&c->hdr->ready, __ATOMIC_ACQUIRE)) { c->err = "rank 0 never initialised the shared segment"; return SCALCE_ERR_HIP; }
  }
  pthread_barrier_wait(&c->hdr->barrier);
  if (rank == 0) shm_unlink(name);  // everybody has it mapped: the name can go
  return SCALCE_OK;
}

extern "C" void scalce_comm_destroy(scalce_comm *c) {
  if (!c) return;
  if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
  if (c->hdr) munmap(c->hdr, c->map_bytes);
  delete c;
}

extern "C" int scalce_comm_barrier(scalce_comm *c, void *stream) {
  if (!c) return SCALCE_ERR_ARG;
  if (c->world == 1 && !c->nccl) return SCALCE_OK;
  if (c->shm) { pthread_barrier_wait(&c->hdr->barrier); return SCALCE_OK; }
  // RCCL has no barrier: a one-element all-reduce on the stream, then wait for it
  static thread_local uint64_t *d_one = nullptr;
  if (!d_one) CM_HIP(c, hipMalloc(reinterpret_cast<void **>(&d_one), 8));
  CM_NCCL(c, g_rccl.AllReduce(d_one, d_one, 1, ncclUint64, ncclSum, c->nccl, (hipStream_t)stream));
  CM_HIP(c, hipStreamSynchronize((hipStream_t)stream));
  return SCALCE_OK;
}

// recv[r * bytes .. ) = rank r's send
extern "C" int scalce_comm_all_gather(scalce_comm *c, const void *d_send, void *d_recv, uint64_t bytes, void *stream) {
  if (!c || (bytes && (!d_send || !d_recv))) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!bytes) return SCALCE_OK;
  if (c->world == 1 && !c->nccl) {
    if (d_send != d_recv) CM_HIP(c, cm_copy(c, d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s));
    return SCALCE_OK;
  }
  if (!c->shm) {
    CM_NCCL(c, g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, c->nccl, s));
    return SCALCE_OK;
  }
  if (bytes > c->slot_bytes) { c->err = "shm transport: message larger than a slot (rehearsal transport)"; return SCALCE_ERR_CAPACITY; }
  CM_HIP(c, cm_copy(c, c->slots + (size_t)c->rank * c->slot_bytes, d_send, bytes, hipMemcpyDeviceToHost, s));
  CM_HIP(c, cm_sync(c, s));
  pthread_barrier_wait(&c->hdr->barrier);
  for (int r = 0; r < c->world; r++)
    CM_HIP(c, cm_copy(c, static_cast<uint8_t *>(d_recv) + (size_t)r * bytes, c->slots + (size_t)r * c->slot_bytes, bytes, hipMemcpyHostToDevice, s));
  CM_HIP(c, cm_sync(c, s));
  pthread_barrier_wait(&c->hdr->barrier);
  return SCALCE_OK;
}

extern "C" int scalce_comm_all_reduce_sum_u64(scalce_comm *c, uint64_t *d_buf, uint64_t count, void *stream) {
  if (!c || (count && !d_buf)) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!count || (c->world == 1 && !c->nccl)) return SCALCE_OK;
  if (!c->shm) {
    CM_NCCL(c, g_rccl.AllReduce(d_buf, d_buf, count, ncclUint64, ncclSum, c->nccl, s));
    return SCALCE_OK;
  }
  const uint64_t bytes = count * 8;
  if (bytes > c->slot_bytes) { c->err = "shm transport: message larger than a slot (rehearsal transport)"; return SCALCE_ERR_CAPACITY; }
  uint64_t *mine = reinterpret_cast<uint64_t *>(c->slots + (size_t)c->rank * c->slot_bytes);
  CM_HIP(c, cm_copy(c, mine, d_buf, bytes, hipMemcpyDeviceToHost, s));
  CM_HIP(c, cm_sync(c, s));
  pthread_barrier_wait(&c->hdr->barrier);
  c->stage.resize(bytes);
  uint64_t *acc = reinterpret_cast<uint64_t *>(c->stage.data());
  memset(acc, 0, bytes);
  for (int r = 0; r < c->world; r++) {
    const uint64_t *x = reinterpret_cast<const uint64_t *>(c->slots + (size_t)r * c->slot_bytes);
    for (uint64_t i = 0; i < count; i++) acc[i] += x[i];
  }
  pthread_barrier_wait(&c->hdr->barrier);
  CM_HIP(c, cm_copy(c, d_buf, acc, bytes, hipMemcpyHostToDevice, s));
  CM_HIP(c, cm_sync(c, s));
  return SCALCE_OK;
}

// Point to point: one message from this rank to `peer` / from `peer` to this rank (what a chain of ranks hands on: the
// counts of the tie-break, sharded.cpp).  Over RCCL a plain ncclSend / ncclRecv on the stream; over the rehearsal transport a
// mailbox per receiver in the shared segment (the sender waits for it to be empty, the receiver for it to be full).
extern "C" int scalce_comm_send(scalce_comm *c, const void *d_buf, uint64_t bytes, int peer, void *stream) {
  if (!c || !d_buf || !bytes || peer < 0 || peer >= c->world || peer == c->rank) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!c->shm) {
    if (!c->nccl) return SCALCE_ERR_ARG;
    CM_NCCL(c, g_rccl.Send(d_buf, bytes, ncclUint8, peer, c->nccl, s));
    return SCALCE_OK;
  }
  if (bytes > SHM_BOX_BYTES) { c->err = "shm transport: point-to-point message larger than a mailbox"; return SCALCE_ERR_CAPACITY; }
  uint8_t *box = c->slots + (size_t)c->world * c->slot_bytes + (size_t)peer * SHM_BOX_BYTES;
  for (long tries = 0; __atomic_load_n(&c->hdr->box_state[peer], __ATOMIC_ACQUIRE) != 0; tries++) {
    if (tries > SHM_SPINS) { c->err = "shm transport: the receiver never emptied its mailbox"; return SCALCE_ERR_HIP; }
    usleep(100);
  }
  CM_HIP(c, cm_copy(c, box, d_buf, bytes, hipMemcpyDeviceToHost, s));
  CM_HIP(c, cm_sync(c, s));
  __atomic_store_n(&c->hdr->box_state[peer], bytes + 1, __ATOMIC_RELEASE);
  return SCALCE_OK;
}
extern "C" int scalce_comm_recv(scalce_comm *c, void *d_buf, uint64_t bytes, int peer, void *stream) {
  if (!c || !d_buf || !bytes || peer < 0 || peer >= c->world || peer == c->rank) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!c->shm) {
    if (!c->nccl) return SCALCE_ERR_ARG;
    CM_NCCL(c, g_rccl.Recv(d_buf, bytes, ncclUint8, peer, c->nccl, s));
    return SCALCE_OK;
  }
  uint8_t *box = c->slots + (size_t)c->world * c->slot_bytes + (size_t)c->rank * SHM_BOX_BYTES;
  uint64_t st = 0;
  for (long tries = 0; (st = __atomic_load_n(&c->hdr->box_state[c->rank], __ATOMIC_ACQUIRE)) == 0; tries++) {
    if (tries > SHM_SPINS) { c->err = "shm transport: no message arrived"; return SCALCE_ERR_HIP; }
    usleep(100);
  }
  if (st - 1 != bytes) { c->err = "shm transport: a message of another size than expected"; return SCALCE_ERR_ARG; }
  CM_HIP(c, cm_copy(c, d_buf, box, bytes, hipMemcpyHostToDevice, s));
  CM_HIP(c, cm_sync(c, s));
  __atomic_store_n(&c->hdr->box_state[c->rank], 0ull, __ATOMIC_RELEASE);
  return SCALCE_OK;
}

// Rank r sends send_bytes[d] bytes to every rank d (consecutive ranges of d_send, rank order) and receives
// recv_bytes[src] from every rank src into consecutive ranges of d_recv (rank order).
extern "C" int scalce_comm_all_to_all_v(scalce_comm *c, const void *d_send, const uint64_t *send_bytes, void *d_recv,
                                        const uint64_t *recv_bytes, void *stream) {
  if (!c || !send_bytes || !recv_bytes) return SCALCE_ERR_ARG;
  std::vector<uint64_t> so(c->world), ro(c->world);
  uint64_t a = 0, b = 0;
  for (int r = 0; r < c->world; r++) { so[r] = a; ro[r] = b; a += send_bytes[r]; b += recv_bytes[r]; }
  return scalce_comm_all_to_all_vo(c, d_send, so.data(), send_bytes, d_recv, ro.data(), recv_bytes, stream);
}

// The same with explicit offsets: the bytes for rank d start at d_send + send_off[d], those from rank src land at
// d_recv + recv_off[src] (a rank that keeps its own part where it is sends 0 bytes to itself and leaves the hole in place).
extern "C" int scalce_comm_all_to_all_vo(scalce_comm *c, const void *d_send, const uint64_t *send_off, const uint64_t *send_bytes,
                                         void *d_recv, const uint64_t *recv_off, const uint64_t *recv_bytes, void *stream) {
  if (!c || !send_bytes || !recv_bytes || !send_off || !recv_off) return SCALCE_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const uint8_t *src = static_cast<const uint8_t *>(d_send);
  uint8_t *dst = static_cast<uint8_t *>(d_recv);
  if (c->world == 1 && !c->nccl) {
    if (send_bytes[0] != recv_bytes[0]) { c->err = "all_to_all_v: sizes disagree"; return SCALCE_ERR_ARG; }
    if (send_bytes[0]) CM_HIP(c, cm_copy(c, dst + recv_off[0], src + send_off[0], send_bytes[0], hipMemcpyDeviceToDevice, s));
    return SCALCE_OK;
  }
  if (!c->shm) {
    // The usual grouped send / receive pattern, in pieces of at most 1 GiB: RCCL takes a size_t count, but a 5 GB message
    // (the q' bytes of a 50 M-read shard) did not arrive whole -- the coded stream of a world-1 run over RCCL came out 18 %
    // larger than over a plain copy (round 4; sends and receives to one peer match in the order they are issued).
    // Why pieces: RCCL 2.26.6 (ROCm 7.0.2) delivers only the first max(n / 2, 1 GiB) + 1 bytes of ONE ncclSend / ncclRecv pair of
    // n >= 2 GiB between a rank and itself -- the rest of the receive buffer is never written, no error is returned
    // (tools/rccl_big_send.py, profiles/r05_rccl_big_send.log: 1 GiB arrives whole; 2, 3, 4 GiB lose everything behind 1, 1.5,
    // 2 GiB + 1 byte).  A 32-bit size somewhere in the self-copy path is our reading; whether a pair between two GPUs does the
    // same could not be tried on one GPU, so nothing here sends more than 1 GiB at once.
    const uint64_t PIECE = c->piece_bytes;
    bool any = false;
    for (int r = 0; r < c->world; r++) any = any || send_bytes[r] || recv_bytes[r];
    if (!any) return SCALCE_OK;
    CM_NCCL(c, g_rccl.GroupStart());
    for (int r = 0; r < c->world; r++) {
      for (uint64_t a = 0; a < send_bytes[r]; a += PIECE)
        CM_NCCL(c, g_rccl.Send(src + send_off[r] + a, (size_t)std::min<uint64_t>(PIECE, send_bytes[r] - a), ncclUint8, r, c->nccl, s));
      for (uint64_t a = 0; a < recv_bytes[r]; a += PIECE)
        CM_NCCL(c, g_rccl.Recv(dst + recv_off[r] + a, (size_t)std::min<uint64_t>(PIECE, recv_bytes[r] - a), ncclUint8, r, c->nccl, s));
    }
    CM_NCCL(c, g_rccl.GroupEnd());
    return SCALCE_OK;
  }
  // rehearsal transport: every rank packs what it sends into its slot (destination order), everybody picks up its part
  uint64_t total = 0;
  for (int r = 0; r < c->world; r++) total += send_bytes[r];
  if (total > c->slot_bytes) { c->err = "shm transport: message larger than a slot (rehearsal transport)"; return SCALCE_ERR_CAPACITY; }
  {
    uint64_t at = 0;
    for (int r = 0; r < c->world; r++) {
      if (send_bytes[r]) CM_HIP(c, cm_copy(c, c->slots + (size_t)c->rank * c->slot_bytes + at, src + send_off[r], send_bytes[r], hipMemcpyDeviceToHost, s));
      at += send_bytes[r];
    }
  }
  for (int r = 0; r < c->world; r++) c->hdr->sizes[c->rank][r] = send_bytes[r];
  CM_HIP(c, cm_sync(c, s));
  pthread_barrier_wait(&c->hdr->barrier);
  for (int r = 0; r < c->world; r++) {
    if (c->hdr->sizes[r][c->rank] != recv_bytes[r]) { c->err = "all_to_all_v: a sender's size differs from what the receiver expects"; pthread_barrier_wait(&c->hdr->barrier); return SCALCE_ERR_ARG; }
    uint64_t off = 0;
    for (int d = 0; d < c->rank; d++) off += c->hdr->sizes[r][d];
    if (recv_bytes[r]) CM_HIP(c, cm_copy(c, dst + recv_off[r], c->slots + (size_t)r * c->slot_bytes + off, recv_bytes[r], hipMemcpyHostToDevice, s));
  }
  CM_HIP(c, cm_sync(c, s));
  pthread_barrier_wait(&c->hdr->barrier);
  return SCALCE_OK;
}
