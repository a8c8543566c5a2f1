// pipeline.cpp -- shards in flight: the scheduling loop of a caller that compresses many shards on one GPU.
//
// What it replaces in the reference: thread_c / ac_write handing -T blocks of a batch to coder threads while the reader
// goes on with the next records (arithmetic.cpp:349-357, compress.cpp:781-786).  Here the unit is a shard: the arithmetic
// coder is a long kernel that only depends on its own shard's front stages, so the front stages (ingest .. emit) of the
// NEXT shards run on one stream beside the coder launches of the previous ones on others.  What the measurements on MI355X
// fixed:
//   * one front stream and `coder_streams` coder streams: HIP maps streams onto a handful of hardware queues, and a read-back
//     that lands in a queue behind a 0.5 s coder kernel waits for all of it (callers set GPU_MAX_HW_QUEUES=8);
//   * a launch takes `group` shards (scalce_batch_entropy_begin_group); consecutive launches rotate over the coder streams
//     and run side by side;
//   * a shard is retired on an EVENT recorded behind its coder launch and collected over the FRONT stream;
//   * the last launch of a run is picked for its own latency (scalce_batch_entropy_begin_group_last).
// Rounds 1-3 kept this loop in Python (scalce_amd/pipeline.py, now a binding of these entry points).
// Built on the public C ABI only (include/scalce_hip.h) plus HIP streams and events.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/scalce_hip.h"

struct scalce_pipeline {
  std::vector<scalce_batch *> b;
  int G = 1;
  bool external = false;  // shards arrive with their coder prepared / enqueued by the caller (sharded runs)
  hipStream_t front = nullptr;
  std::vector<hipStream_t> coders;
  hipEvent_t ev_front = nullptr;      // coder streams wait for the front stages through it
  std::vector<hipEvent_t> ev;         // per slot: behind its coder launch
  std::vector<char> busy;
  std::vector<int> pending;
  int next = 0;
  int device = 0;  // the device that was current when the pipeline was made
  uint64_t launches = 0;
  std::string err;
};

namespace {
int fail(scalce_pipeline *p, const char *what, hipError_t e) {
  p->err = std::string(what) + ": " + hipGetErrorString(e);
  return SCALCE_ERR_HIP;
}
#define PL_HIP(p, expr)                                  \
  do {                                                   \
    hipError_t e_ = (expr);                              \
    if (e_ != hipSuccess) return fail(p, #expr, e_);     \
  } while (0)

int flush(scalce_pipeline *p, int last, int *launched) {
  if (p->pending.empty()) return SCALCE_OK;
  hipStream_t coder = p->coders[p->launches++ % p->coders.size()];
  std::vector<scalce_batch *> grp;
  for (int sl : p->pending) grp.push_back(p->b[sl]);
  const int rc = scalce_batch_entropy_begin_group_last(grp.data(), (int)grp.size(), p->front, coder, last ? 1 : 0);
  if (rc) { p->err = "scalce_batch_entropy_begin_group_last failed"; return rc; }
  for (int sl : p->pending) {
    PL_HIP(p, hipEventRecord(p->ev[sl], coder));
    p->busy[sl] = 1;
  }
  p->pending.clear();
  if (launched) *launched = 1;
  return SCALCE_OK;
}
}  // namespace

extern "C" int scalce_pipeline_create(scalce_batch **batches, int nslots, int group, int coder_streams, int external_coder,
                                      scalce_pipeline **out) {
  if (!batches || nslots < 1 || group < 1 || coder_streams < 1 || !out) return SCALCE_ERR_ARG;
  if (group > 1 && nslots < group) return SCALCE_ERR_ARG;  // (a caller that lets launches fill up needs 2 * group slots: the front stages of one group run while the previous one is coded)
  scalce_pipeline *p = new scalce_pipeline;
  *out = p;
  p->b.assign(batches, batches + nslots);
  p->G = group;
  p->external = external_coder != 0;
  p->busy.assign(nslots, 0);
  p->ev.assign(nslots, nullptr);
  PL_HIP(p, hipGetDevice(&p->device));
  PL_HIP(p, hipStreamCreateWithFlags(&p->front, hipStreamNonBlocking));
  for (int i = 0; i < coder_streams; i++) {
    hipStream_t s;
    PL_HIP(p, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    p->coders.push_back(s);
  }
  PL_HIP(p, hipEventCreateWithFlags(&p->ev_front, hipEventDisableTiming));
  for (int i = 0; i < nslots; i++) PL_HIP(p, hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming));
  return SCALCE_OK;
}

extern "C" void scalce_pipeline_destroy(scalce_pipeline *p) {
  if (!p) return;
  for (hipEvent_t e : p->ev) if (e) hipEventDestroy(e);
  if (p->ev_front) hipEventDestroy(p->ev_front);
  for (hipStream_t s : p->coders) if (s) hipStreamDestroy(s);
  if (p->front) hipStreamDestroy(p->front);
  delete p;
}

extern "C" const char *scalce_pipeline_error(const scalce_pipeline *p) { return p ? p->err.c_str() : "no pipeline"; }
extern "C" void *scalce_pipeline_front_stream(scalce_pipeline *p) { return p ? p->front : nullptr; }
extern "C" void *scalce_pipeline_coder_stream(scalce_pipeline *p, int i) {
  return p && i >= 0 && i < (int)p->coders.size() ? p->coders[i] : nullptr;
}

// the slot's previous shard: wait for its coder (the event behind the launch), collect it over the front stream
extern "C" int scalce_pipeline_retire(scalce_pipeline *p, int slot, int *had_shard) {
  if (!p || slot < 0 || slot >= (int)p->b.size()) return SCALCE_ERR_ARG;
  if (had_shard) *had_shard = 0;
  if (!p->busy[slot]) return SCALCE_OK;
  PL_HIP(p, hipSetDevice(p->device));
  PL_HIP(p, hipEventSynchronize(p->ev[slot]));
  p->busy[slot] = 0;
  const int rc = scalce_batch_finish(p->b[slot], p->front);  // sizes of the coded streams, device error word
  if (rc) { p->err = "scalce_batch_finish failed"; return rc; }
  if (had_shard) *had_shard = 1;
  return SCALCE_OK;
}

// Next slot in round-robin order.  *slot is free to be overwritten when the call returns -- unless *retired says that a shard
// has just been collected in it: its outputs are the caller's to take before it starts the next front stages there.
extern "C" int scalce_pipeline_acquire(scalce_pipeline *p, int *slot, int *retired) {
  if (!p || !slot) return SCALCE_ERR_ARG;
  const int sl = p->next;
  p->next = (p->next + 1) % (int)p->b.size();
  if (std::find(p->pending.begin(), p->pending.end(), sl) != p->pending.end()) {  // the caller never submitted enough shards to launch
    const int rc = flush(p, 0, nullptr);
    if (rc) return rc;
  }
  *slot = sl;
  return scalce_pipeline_retire(p, sl, retired);
}

// The front stages of `slot` are enqueued on the front stream: launch the coder now or with the next shards.
// flush_now 1: launch what is pending now, and nothing will run beside that launch (the end of a run, or a wave of shards
// that fills every slot: the next front stages wait for these slots anyway) -- it is shaped for its own latency;
// flush_now 2: launch what is pending now, front stages of further shards follow beside it.
extern "C" int scalce_pipeline_submit(scalce_pipeline *p, int slot, int flush_now, int *launched) {
  if (!p || slot < 0 || slot >= (int)p->b.size()) return SCALCE_ERR_ARG;
  if (launched) *launched = 0;
  PL_HIP(p, hipSetDevice(p->device));
  if (p->G == 1) {
    hipStream_t coder = p->coders[0];
    if (!p->external) {  // (an external caller has already enqueued the coder on coder stream 0)
      coder = p->coders[p->launches++ % p->coders.size()];
      PL_HIP(p, hipEventRecord(p->ev_front, p->front));
      PL_HIP(p, hipStreamWaitEvent(coder, p->ev_front, 0));
      const int rc = scalce_batch_entropy_begin(p->b[slot], nullptr, coder);
      if (rc) { p->err = "scalce_batch_entropy_begin failed"; return rc; }
      if (launched) *launched = 1;
    }
    PL_HIP(p, hipEventRecord(p->ev[slot], coder));
    p->busy[slot] = 1;
    return SCALCE_OK;
  }
  p->pending.push_back(slot);
  if ((int)p->pending.size() >= p->G || flush_now) return flush(p, flush_now == 1, launched);
  return SCALCE_OK;
}

// what is pending goes out as the last launch of a run (a caller that then retires the slots one by one)
extern "C" int scalce_pipeline_flush_last(scalce_pipeline *p) {
  if (!p) return SCALCE_ERR_ARG;
  PL_HIP(p, hipSetDevice(p->device));
  return flush(p, 1, nullptr);
}

// what is pending goes out (as the last launch of a run), every slot is waited for and collected
extern "C" int scalce_pipeline_drain(scalce_pipeline *p) {
  if (!p) return SCALCE_ERR_ARG;
  PL_HIP(p, hipSetDevice(p->device));
  int rc = flush(p, 1, nullptr);
  if (rc) return rc;
  for (int sl = 0; sl < (int)p->b.size(); sl++)
    if ((rc = scalce_pipeline_retire(p, sl, nullptr))) return rc;
  return SCALCE_OK;
}
