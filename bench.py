#!/usr/bin/env python3
"""bench.py -- input FASTQ MB/s of the SCALCE hot path on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (ingest -> quality statistics -> tokenize with exact tie-break
-> bucket/reorder -> emit -> arithmetic coder) over one synthetic shard that is already resident in HBM.
N = 1 runs BASELINE.json configs[1]: 50 M x 100 bp single-end, arithmetic-coded qualities.  N > 1 is weak
scaling: every rank holds a shard of the same size (one process per GPU) and the ranks produce ONE archive --
scalce_sharded_compress, the C++ host of scalce_amd/csrc/sharded.cpp over RCCL (comm.cpp): run-wide -B chunks,
tie-break, quality model and 10 MiB block cutting, byte-identical with the one-GPU archive of the same input.
torch.distributed is only used to hand RCCL's unique id to the ranks and for the timing barrier.

Prints ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline       -- SURVEY 8(d): algorithmic bytes of a step (308 B per read) / ms_per_step against the 8 TB/s HBM peak;
                    beside it the dominant kernel's own views (HBM bytes and issue slots per launch, HIP events)
  cpu_baseline   -- the reference's own compress() (oracle/_ref/ref_full: every reference source but main.cpp, built in
                    the dev container, the binary travels) at -T 1 and at its default thread count, on a bounded sample
  parity_checked -- the run proves its own output: the LAST timed shard is decoded on the device back to FASTQ text whose
                    record multiset equals the input's, and the sample of the CPU leg gives byte-identical archives
                    through the `scalce` binary and through the reference
One GPU on its own plans its coder launches for the length of the run (--launch-plan waves, config.launch_plan): a launch
takes ~0.5 s whether it holds five shards or fifteen, so the remainder of the run goes first and every later launch fills
all slots and all CUs; --launch-plan eager is the pipeline of a stream of unknown length (a launch per six shards).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# The pipeline keeps a front stream and up to three coder streams busy at once: HIP maps streams onto GPU_MAX_HW_QUEUES
# hardware queues (default 4), and two streams that share a queue run one after the other -- a front-stage kernel behind a
# 0.65 s coder launch.  Must be set before the runtime comes up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=36)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--inflight", type=int, default=None,
                    help="shards in flight per GPU: the coder of shard j runs on its own stream while the front "
                         "stages of shard j+1 run (1 = strictly one after the other)")
    ap.add_argument("--group", type=int, default=None,
                    help="shards per coder launch (scalce_batch_entropy_begin_group, four blocks per workgroup); "
                         "1 = one launch per shard with the one-block-per-workgroup kernel")
    ap.add_argument("--reads", type=int, default=50_000_000, help="reads per GPU")
    ap.add_argument("--length", type=int, default=100)
    ap.add_argument("--cpu-sample", type=int, default=1_500_000, help="records of the CPU baseline sample (0 = skip)")
    ap.add_argument("--stage-times", action="store_true", help="also print per-stage HIP-event times to stderr")
    ap.add_argument("--no-e2e", action="store_true", help="skip the file -> archive leg (the `scalce` binary on the same shard written to a file)")
    ap.add_argument("--no-verify", action="store_true", help="skip the self-check of the last timed shard (device decode + record digest)")
    ap.add_argument("--no-table-scale", action="store_true", help="skip the tokenizer leg on a core table of a million cores")
    ap.add_argument("--launch-plan", choices=("waves", "eager"), default="waves",
                    help="one GPU on its own: `waves` = coder launches that fill every slot, the remainder of the run first (a launch takes "
                         "~0.5 s whatever it holds); `eager` = a launch as soon as `--group` shards are ready, side by side (what a stream of "
                         "unknown length gets; rounds 1-5a)")
    ap.add_argument("--shared-input", action="store_true",
                    help="every shard in flight reads the SAME text tensor (rounds 1-3; more shards fit).  Default: one distinct text per shard in flight")
    ap.add_argument("--ref-full-shard", action="store_true",
                    help="also run the reference's own compress() -T 1 on the whole bench shard on this box's host (~5 min for 50 M reads) and "
                         "compare its three archive files with the product's, byte for byte (parity.ref_full_shard)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from scalce_amd import host, synth_gpu
    from scalce_amd.pipeline import ShardPipeline

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # torch.distributed is the control plane only (RCCL's unique id, the timing barrier): gloo.  The data path is the
        # library's own RCCL communicator (scalce_amd/csrc/comm.cpp).
        dist.init_process_group("gloo")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU path exists)"
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n, L = args.reads, args.length
    blob = open(os.path.join(ROOT, "tests", "golden", "patterns.bin"), "rb").read()
    ctx = host.Context(local, patterns_bin=blob)
    SEED0 = 20261003
    text = synth_gpu.fastq_on_device(n, L, dev, seed=SEED0 + rank, first_index=rank * n)
    nbytes = text.numel()
    # quality model from the first 100 000 records of the shard (quality_mapping_init's sample)
    head = text[: min(nbytes, 100000 * (2 * L + 20))].cpu().numpy().tobytes()
    from scalce_amd import format as fmt
    off, vals, Ls = fmt.sample_qmap(head)
    assert Ls == L
    # A sharded run goes through scalce_sharded_compress (C++ host, RCCL).  SCALCE_BENCH_FORCE_SHARDED=1 takes that path at
    # world 1 as well (every collective is then a real RCCL call of one rank); SCALCE_COMM=shm rehearses several ranks on
    # ONE GPU through the shared-memory transport.
    sharded = world > 1 or bool(os.environ.get("SCALCE_BENCH_FORCE_SHARDED"))
    comm = None

    def make_comm(k):
        if os.environ.get("SCALCE_COMM") == "shm":
            name = [("/scalce_bench_%d_%d" % (os.getpid(), k)) if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(name, src=0)
            return host.Comm(local, world, rank, shm_name=name[0], slot_bytes=int(os.environ.get("SCALCE_SHM_SLOT", str(8 << 30))))
        uid = [host.Comm.unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        return host.Comm(local, world, rank, unique_id=uid[0])

    if sharded:
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout is the one JSON line's, so
        # the banner goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = make_comm(0)
            comm.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    # Shards per coder launch / in flight / coder streams: a launch of the one-block-per-lane coder takes ~0.5 s whatever it
    # holds, so the pipeline wants launches side by side beside the front stages of the next shards, and as many slots as the
    # card holds.
    # (round 5: a sharded rank runs the SAME pipeline as one GPU on its own -- one block per lane, grouped launches on several
    #  streams: its reordered stream lives in the shared workspace (scalce_batch_set_stream_scratch) and what stays on the rank
    #  never goes through the transport, so a slot costs it what it costs the plain path)
    G = args.group
    auto_group = G is None
    if G is None:
        G = 4
    G = max(1, G)
    D = max(1, args.inflight) if args.inflight is not None else 3 * G + 2
    # Every shard in flight has a text of its OWN (round 4; VERDICT r3: fourteen jobs that read one tensor are not fourteen
    # jobs a card can hold).  What is free now decides how many fit.
    own_text = not args.shared_input
    in_place = not sharded and not os.environ.get("SCALCE_BENCH_NO_INPLACE")
    # `fronts`: host threads, each with a front stream, a context, a set of front-stage buffers, D / fronts of the slots and -- in
    # a sharded run -- a communicator of its own.  One is right for a GPU on its own (two measured the same, round 3).  For a rank
    # among several, TWO hide what a shard waits for between ranks -- the tie-break's chain (a settle per rank in front of it), the
    # rows and q' bytes on the wire -- behind the other pipeline's work (5.6 x one GPU projected at eight ranks
    # against 3.4 x with one).  SCALCE_BENCH_FRONTS=2 asks for it; it is rehearsed over the shared-memory transport
    # (tests, profiles/r05_sharded_w1_w2.log) but stays opt-in: two RCCL communicators driven from two threads of every rank
    # have never run on real peers here, and a run the driver cannot see the end of is worse than a slower one.
    F = max(1, int(os.environ.get("SCALCE_BENCH_FRONTS", "1")))
    if args.inflight is None:
        free_b, _ = torch.cuda.mem_get_info()
        scale = (n * L) / 5e9
        # a slot: its text (10.8 GB) + reordered q' 5 GB (the coder writes its blocks over it: scalce_batch_set_code_in_place),
        # records 1.25, names 0.54, tables; a sharded rank's stream is a buffer handed to the batch and coded into block buffers
        # of its own (3.2 GB, sized from the table)
        per_slot = (7.4e9 if in_place else 10.7e9) * scale + (nbytes if own_text else 0)
        # shared: the front-stage workspace (~20 GB); a sharded rank adds the reordered stream (5 GB) and, with peers, the receive
        # side of the q' exchange (5 GB), the rows that change owner (text in, half a chunk at most) and the second set of row
        # arrays (scalce_batch_rewindow, 9 GB)
        shared = F * (22e9 * scale + (0 if not sharded else (6e9 + (0 if world == 1 else 14e9 + 0.5 * nbytes)) * scale))
        fit = int((free_b + (nbytes if own_text else 0) - shared - 5e9) // per_slot)
        if world > 1:   # every rank takes the same shape
            tfit = torch.tensor([fit], dtype=torch.int64)
            dist.all_reduce(tfit, op=dist.ReduceOp.MIN)
            fit = int(tfit.item())
        want = D if not auto_group else 16   # (more slots than that buy nothing: tools/r5_sweep.sh, 15 and 16 measure the same)
        if fit != D and (fit < D or auto_group):
            D = max(1, min(fit, want))
            if auto_group:
                # A coder launch takes ~0.5 s whatever it holds: what counts is that every group of slots has a stream of its own
                # and that a slot is free when the front stream wants one.  Six shards per launch on two streams: with twelve
                # slots (round 4) 6 / 2, 4 / 3 and 3 / 4 measured the same; with fifteen (round 5, coding in place) 6 / 2 gives
                # 74.0 ms per shard, 5 / 3 80.5, 7 / 2 81.1 (tools/r5_sweep.sh) -- a third launch side by side takes the CUs the
                # front stages need (72-77 ms per front stage beside two launches of five against 48 beside two of six), and at
                # the driver's 20 steps the last launch is two shards on the kernel with the lower latency.
                G = 6 if D >= 12 else (3 if D >= 6 else max(1, D // 2))
            if F > 1:   # every front takes D / F slots and at least two launches' worth of them
                D -= D % F
                if auto_group:
                    G = 3 if D // F >= 6 else max(1, D // F // 2)
            print("bench: %.0f GB of HBM free: %d shards in flight, %d per coder launch" % (free_b / 1e9, D, G), file=sys.stderr)
    if G > 1:
        D = max(D, 2 * G)
    # slot i reads texts[i]: different seeds, the same record shape (sizes are equal: names and lengths are)
    texts = [text]
    if own_text:
        for i in range(1, D):
            texts.append(synth_gpu.fastq_on_device(n, L, dev, seed=SEED0 + rank + 7919 * i, first_index=rank * n))
            assert texts[-1].numel() == nbytes
    else:
        texts = [text] * D
    # -B: the reference's default, 4 GiB of record bytes per spill chunk (main.cpp:68) -- 50 M reads of 100 bp are 3 chunks
    B = int(os.environ.get("SCALCE_BENCH_BUCKET_SET", str(4 << 30)))
    # the shards in flight share ONE set of front-stage buffers (rows, tokens, events, sort scratch: dead once a shard is
    # emitted, and front stages run one at a time on the front stream): 15 GB instead of 35 GB of HBM per shard in flight
    # `fronts` host threads, each with a front stream, a context, a set of front-stage buffers and D / fronts of the batches
    # of its own: the front stages are ~250 launches per shard with a dozen read-backs between them, and many of the
    # launches (late tie-break sweeps, radix passes over a few thousand keys) leave most of the chip idle -- a second
    # shard's front stages fill those holes
    if F > 1:
        assert G > 1 and D % F == 0 and D // F >= 2 * G, "fronts: D / fronts >= 2 * group"
    ctxs = [ctx] + [host.Context(local, patterns_bin=blob) for _ in range(F - 1)]
    comms = [comm] * F
    if sharded:
        comms = [comm]
        for f in range(1, F):   # a communicator per front: their collectives interleave in any order across the two threads
            sys.stdout.flush()
            saved_stdout = os.dup(1)
            os.dup2(2, 1)
            try:
                comms.append(make_comm(f))
            finally:
                sys.stdout.flush()
                os.dup2(saved_stdout, 1)
                os.close(saved_stdout)
    batches = []
    for f in range(F):
        ws = None if os.environ.get("SCALCE_BENCH_OWN_WORKSPACES") else host.Workspace(ctxs[f])
        batches += [host.Batch(ctxs[f], L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)], bucket_set_size=B,
                               workspace=ws) for _ in range(D // F)]
    for b in batches:   # the coded blocks are framed when they are delivered (scalce_batch_qual_window), not by a copy pass
        b.set_frame_on_demand(True)
        if in_place:    # ... and written over the symbols they were coded from (the texts stay where they are: a shard whose
            b.set_code_in_place(True)   # block outgrows its input is run again from its text)
    batch = batches[0]
    state = {}

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    trace = os.environ.get("BENCH_TRACE")
    tr0 = [time.perf_counter()]

    def mark(what):
        if trace and rank == 0:
            print("  [%8.1f ms] %s" % ((time.perf_counter() - tr0[0]) * 1e3, what), file=sys.stderr)

    # the scheduling loop (scalce_amd/pipeline.py): front stages of the next shards on one stream beside the coder of the
    # previous ones on another, `G` shards per coder launch, shards retired on events
    DF = D // F
    # as many coder streams as groups fit the slots (every launch takes ~0.6 s whatever it holds: with a stream per group in
    # rotation no group waits for another's launch to end; 3 / 4 / 12 measured 88.1 ms per shard against 90.0 at 3 / 3 / 12)
    n_coder_streams = int(os.environ.get("SCALCE_BENCH_CODER_STREAMS", str(min(2 if G >= 6 else 6, DF // G)) if (G > 1 and DF >= 2 * G) else "1"))
    # The launch plan of a run of k shards on D slots (one GPU on its own).  A coder launch takes ~0.5 s whether it holds five
    # shards or fifteen, and its CUs are lost to the front stages beside it: launches of six as they fill up leave the chip to
    # a last launch of two for 0.35 s at the driver's 20 steps.  `waves`: every launch but the first takes ALL D slots --
    # nothing can run beside it anyway, so it is spread over every CU (scalce_pipeline_submit: flush_now 1) -- and the
    # remainder k mod D goes FIRST, beside the front stages of the first full wave: 20 steps = 5 + 15.
    # (tools: SCALCE_BENCH_GROUPS-style sweeps in DESIGN.md section 7: 72.2 ms per shard eager, 69.2 as 5 + 15, 67.9 with the
    #  last launch on all CUs; 36 steps: 71.0 eager, 69.3 as 6 + 15 + 15; 16 steps: 84.6 eager, 71.8 as 1 + 15)
    waves = args.launch_plan == "waves" and not sharded and F == 1 and auto_group and G > 1 and D >= 2 * G

    def launch_plan(k):
        r = k % D
        return ([r] if r else []) + [D] * (k // D)

    if waves and not os.environ.get("SCALCE_AC_LANES_USED"):
        # blocks per workgroup of the one-block-per-lane coder: as few as let a full wave take the chip in ONE round (the library
        # picks that by itself for a launch that has the chip to itself; here the first, smaller launch gets the same shape --
        # it has to be through before the front stages of the wave behind it want its slots: 28 against 32 blocks: 67.9 against
        # 69.2 ms per shard at 20 steps)
        blocks_per_shard = -(-(n * L) // (10 << 20))
        n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
        os.environ["SCALCE_AC_LANES_USED"] = str(max(8, min(64, -(-(D * blocks_per_shard) // n_cus))))

    pipes = [ShardPipeline(batches[f * DF:(f + 1) * DF], group=(D if waves else G), sharded=sharded, trace=mark if trace else None,
                           coder_streams=n_coder_streams)
             for f in range(F)]
    pipe = pipes[0]
    front = pipe.front

    def run(k):
        if F == 1:
            return run_on(pipe, k)
        import threading
        errs = []

        def body(f):
            try:
                torch.cuda.set_device(local)
                run_on(pipes[f], len(range(f, k, F)))
            except BaseException as ex:  # noqa: BLE001
                errs.append(ex)
        ts = [threading.Thread(target=body, args=(f,)) for f in range(F)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    slot_of_pipe = {id(p): f * DF for f, p in enumerate(pipes)}
    front_of_pipe = {id(p): f for f, p in enumerate(pipes)}

    def run_on(pipe, k):
        fr = front_of_pipe[id(pipe)]
        comm, ctx = comms[fr], ctxs[fr]
        ends = {}
        if waves:   # shard index behind which a launch goes out -> 1: alone on the chip (a full wave, the end), 2: beside front stages
            at = 0
            for g in launch_plan(k):
                at += g
                ends[at - 1] = 1 if (g == D or at == k) else 2
        for j in range(k):
            slot, b = pipe.acquire()
            mark(f"shard {j}: front (slot {slot})")
            gslot = slot_of_pipe[id(pipe)] + slot
            tx = texts[gslot]
            with torch.cuda.stream(pipe.front):
                if not sharded:
                    b.front(tx.data_ptr(), nbytes, None, 0, pipe.front.cuda_stream)
                elif G == 1:
                    state[gslot] = host.sharded_compress(comm, ctx, b, tx.data_ptr(), nbytes, flags=host.SHARD_CODER_ASYNC,
                                                         stream=pipe.front.cuda_stream, coder_stream=pipe.coder.cuda_stream,
                                                         result=state.get(gslot))
                else:
                    state[gslot] = host.sharded_compress(comm, ctx, b, tx.data_ptr(), nbytes, flags=host.SHARD_PREPARE_ONLY,
                                                         stream=pipe.front.cuda_stream, result=state.get(gslot))
            mark(f"shard {j}: front done")
            pipe.submit(slot, tag=j, flush=ends.get(j, 0) if waves else (j + 1 == k))
        pipe.drain()

    torch.cuda.synchronize()  # the synthetic shard was generated on the default stream
    warm = max(args.warmup, D) if args.warmup > 0 else 0  # every slot allocates its buffers outside the timed region
    run(warm)
    barrier()
    t1 = time.perf_counter()
    run(1)  # one shard alone, nothing in flight beside it: the latency of a single job
    barrier()
    single_ms = (time.perf_counter() - t1) * 1e3
    for b in batches:
        b.kernel_timing(True)
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    free_b, total_b = torch.cuda.mem_get_info()
    hbm_used_gb = round((total_b - free_b) / 1e9, 1)
    ks = [b.kernel_ms() for b in batches]
    k = {key: sum(x[key] for x in ks) for key in ks[0]}
    stats = batch.stats()
    out_bytes = sum(batch.output_ptr(w, 0)[1] for w in (host.OUT_READS, host.OUT_NAMES, host.OUT_QUAL))

    # ---- the run proves its own output (never inside the timed region) ----
    parity = {}
    decode = None
    if sharded and not args.no_verify:
        from scalce_amd import verify
        last_slot = (pipe._next - 1) % DF
        # (1) one rank (SCALCE_BENCH_FORCE_SHARDED=1: every collective a real RCCL call of one rank) builds the whole archive in one
        # batch: the three hashes of the reference's own full-size files must come out
        gold_path = os.path.join(ROOT, "tests", "golden", "full_size_ref.json")
        if rank == 0 and world == 1:
            try:
                gold = json.load(open(gold_path)) if os.path.exists(gold_path) else None
                if gold and (gold["reads"], gold["length"], gold["seed"]) == (n, L, SEED0) and B == 4 << 30:
                    got = verify.archive_hashes(batches[0], L, off, n)
                    same = {k: got[k] == v["sha256"] for k, v in gold["files"].items()}
                    parity["reference_full_size"] = {"ok": all(same.values()), "files": same,
                                                     "what": "sharded path, one rank: SHA-256 of .scalce{n,r,q} of slot 0's shard equal to the reference's own compress() -T 1 files"}
            except Exception as ex:  # noqa: BLE001
                parity["reference_full_size"] = {"ok": False, "error": repr(ex)[:300]}
        # (2) any number of ranks: the LAST timed shard comes back.  Every rank decodes the coder blocks it holds (its range of the
        # run-wide stream) on the device, the symbols go back to the ranks that emitted the records (the block-range exchange run
        # backwards), every rank rebuilds the FASTQ text of its records, and the record digests -- (count, two 64-bit sums of a
        # per-record hash), summed over the ranks -- must equal the digests of the input texts summed over the ranks: records
        # change owner between ranks, the run's multiset does not.  All of it collective: every rank takes part or none.
        for i, b in enumerate(batches):
            if i not in (0, last_slot):
                b.close()
        for i in range(len(texts)):
            if i not in (0, last_slot):
                texts[i] = None
        torch.cuda.empty_cache()
        tv0 = time.perf_counter()
        vtrace = os.environ.get("SCALCE_TRACE")
        if vtrace:
            print("  [verify, rank %d] batches closed %.2f s after the timed loop" % (rank, time.perf_counter() - t0 - dt), file=sys.stderr, flush=True)
        mine = {"want": None, "got": None, "error": None}
        try:
            mine["want"] = verify.record_digest(texts[last_slot])
            if vtrace:
                print("  [verify, rank %d] input digest at %.2f s" % (rank, time.perf_counter() - tv0), file=sys.stderr, flush=True)
        except Exception as ex:  # noqa: BLE001
            mine["error"] = "input digest: " + repr(ex)[:200]
        try:   # (collective calls inside: a rank that failed above still takes part)
            back = verify.sharded_records_text(comm, ctx, batches[last_slot], state[last_slot], L, off, dev)
            mine["got"] = verify.record_digest(back)
            if vtrace:
                print("  [verify, rank %d] output digest at %.2f s" % (rank, time.perf_counter() - tv0), file=sys.stderr, flush=True)
            del back
        except Exception as ex:  # noqa: BLE001
            mine["error"] = (mine["error"] or "") + " decode: " + repr(ex)[:200]
        everyone = [mine]
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
        if rank == 0:
            errs = [f"rank {r}: {e['error']}" for r, e in enumerate(everyone) if e["error"]]
            if errs:
                parity["sharded_run"] = {"ok": False, "error": "; ".join(errs)[:400]}
            else:
                want = verify.digest_sum([e["want"] for e in everyone])
                got = verify.digest_sum([e["got"] for e in everyone])
                parity["sharded_run"] = {"ok": bool(want == got and want[0] == n * world), "records": got[0], "ranks": world,
                                         "records_per_rank_out": [e["got"][0] for e in everyone],
                                         "what": "last timed shard of the run (ONE archive over all ranks): every rank decodes the coder blocks it holds on the "
                                                 "device, the symbols return to the ranks that emitted the records, every rank rebuilds its records' FASTQ "
                                                 "text; (count, two 64-bit sums of per-record hashes) summed over the ranks equal to the input texts'",
                                         "seconds": round(time.perf_counter() - tv0, 2)}
    if rank == 0 and not sharded and not args.no_verify:
        from scalce_amd import verify
        # the checks need room (5 GB of symbols, 10.8 GB of text, the digest's temporaries): everything but slot 0 and the slot
        # of the last timed shard goes first
        last_slot = (pipe._next - 1) % DF
        for i, b in enumerate(batches):
            if i not in (0, last_slot):
                b.close()
        for i in range(len(texts)):
            if i not in (0, last_slot):
                texts[i] = None
        torch.cuda.empty_cache()
        # (1) byte identity with the REFERENCE at full size.  tests/golden/full_size_ref.json holds the SHA-256 of the three
        # files the reference's own compress() wrote for the text of slot 0 (made on a GPU box by tools/full_size_ref_check.py:
        # 286 s of CPU).  Slot 0's batch still holds the archive streams of the last shard it took in the timed loop: with the
        # file headers in front they must hash to the same three values -- bucket assignment, tie-break winners, in-bucket
        # order, chunk merge and every coder byte.
        gold_path = os.path.join(ROOT, "tests", "golden", "full_size_ref.json")
        try:
            gold = json.load(open(gold_path)) if os.path.exists(gold_path) else None
            if gold and (gold["reads"], gold["length"], gold["seed"]) == (n, L, SEED0) and B == 4 << 30 and world == 1:
                tv0 = time.perf_counter()
                got = verify.archive_hashes(batches[0], L, off, n)
                same = {k: got[k] == v["sha256"] for k, v in gold["files"].items()}
                parity["reference_full_size"] = {
                    "ok": all(same.values()), "files": same, "timed_shard": bool(args.steps >= D),
                    "what": "slot 0's shard of the timed loop: SHA-256 of .scalce{n,r,q} (streams in HBM + file headers) equal to what the "
                            "reference's own compress() -T 1 wrote for the same text (tests/golden/full_size_ref.json; 3 spill chunks, factor 2)",
                    "seconds": round(time.perf_counter() - tv0, 2)}
        except Exception as ex:  # noqa: BLE001
            parity["reference_full_size"] = {"ok": False, "error": repr(ex)[:300]}
        # (2) the LAST timed shard decoded on the device back to FASTQ text, record multiset equal to its input's; the two
        # device stages of that are the decode leg of the measurement
        try:
            last = pipe.batches[last_slot]      # the batch that holds the last timed shard (of the first front)
            tv0 = time.perf_counter()
            want = verify.record_digest(texts[last_slot])
            tm = {}
            back = verify.decode_shard(last.ctx, last, L, off, dev, timings=tm)
            got = verify.record_digest(back)
            parity["full_shard"] = {"ok": bool(got == want and want[0] == n), "records": got[0],
                                    "what": "last timed shard: archive streams -> scalce_ac_decode + scalce_fastq_records on the "
                                            "device -> FASTQ text; (count, two 64-bit sums of per-record hashes) equal to the input's",
                                    "seconds": round(time.perf_counter() - tv0, 2)}
            dsec = tm["ac_decode"] + tm["records"]
            nblk_dec = (tm["symbols"] + 10 * 1024 * 1024 - 1) // (10 * 1024 * 1024)
            decode = {"value": round(tm["text_bytes"] / dsec / 1e6, 1), "unit": "MB/s of FASTQ restored", "seconds": round(dsec, 3),
                      "ac_decode_ms": round(tm["ac_decode"] * 1e3, 1), "records_ms": round(tm["records"] * 1e3, 1),
                      "ns_per_symbol_per_block": round(tm["ac_decode"] * 1e9 / min(tm["symbols"], 10 * 1024 * 1024), 1),
                      "blocks": int(nblk_dec),
                      "what": "the last timed shard's archive, coded stream resident in HBM: scalce_ac_decode (arithmetic.cpp:196-268, one "
                              "serial chain per 10 MiB block) + scalce_fastq_records (decompress.cpp:240-366); one job alone on the card"}
            del back
        except Exception as ex:  # noqa: BLE001
            parity["full_shard"] = {"ok": False, "error": repr(ex)[:300]}
    if args.stage_times and rank == 0:
        batch.stage_reset(True)
        batch.compress(text.data_ptr(), nbytes, None, 0, front.cuda_stream)
        batch.finish(front.cuda_stream)
        print("stage ms:", {s: round(v[0], 2) for s, v in batch.stage_ms().items()}, stats, file=sys.stderr)
        batch.stage_reset(False)
    if rank == 0 and world == 1 and args.ref_full_shard:
        try:
            parity["ref_full_shard"] = ref_full_shard(text, nbytes)
        except Exception as ex:  # noqa: BLE001
            parity["ref_full_shard"] = {"ok": False, "error": repr(ex)[:300]}

    cpu = None
    if rank == 0 and args.cpu_sample > 0:   # (rank 0's host cores, rank 0's own shard: any number of ranks)
        try:
            cpu, parity["sample"] = cpu_baseline(text, n, L, args.cpu_sample)
        except Exception as ex:  # noqa: BLE001
            cpu = {"error": repr(ex)[:300]}
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        for b in batches:   # the CLI is a process of its own and needs the card's memory
            b.close()
        del batches[:], pipe, pipes[:]
        texts[1:] = []
        torch.cuda.empty_cache()
        try:
            e2e = end_to_end(text, nbytes)
        except Exception as ex:  # noqa: BLE001 - a side leg must not take the measured line with it
            e2e = {"error": repr(ex)[:300]}

    table_scale = None
    if rank == 0 and world == 1 and not args.no_table_scale and not args.no_e2e:
        try:
            table_scale = table_scale_leg(dev)
        except Exception as ex:  # noqa: BLE001
            table_scale = {"error": repr(ex)[:300]}

    # Counter figures come from rocprofv3 summaries of THIS build kept under profiles/ (tools/profile_round.sh <tag>,
    # tools/pmc_sq.sh <tag>; SCALCE_PROFILE_TAG names the tag): HBM bytes from --pmc FETCH_SIZE / WRITE_SIZE in separate
    # passes (KB; FETCH_SIZE doubled: gfx950 reports half of a streaming read, MI355X_MICROARCH.md "HBM"), instructions
    # from --pmc SQ_INSTS_*.  A figure whose file is missing is null, never a constant.
    bpw_env = int(os.environ.get("SCALCE_AC_BLOCKS_PER_WG", "0") or 0)
    nblk_launch = G * ((n * L + 10 * 1024 * 1024 - 1) // (10 * 1024 * 1024))
    kname = "ac_encode_k" if G == 1 else ("ac_encode_rows_k" if bpw_env in (4, 8) or (bpw_env == 0 and nblk_launch < 900) else "ac_encode_lanes_k")
    tag = os.environ.get("SCALCE_PROFILE_TAG", "r05_final")
    pmc = os.path.join(ROOT, "profiles", f"{tag}_bench50m_pmc_fetch_write.json")
    sqf = os.path.join(ROOT, "profiles", f"{tag}_pmc_sq.json")
    k_traffic, step_traffic, traffic_src, instr_per_symbol, issue_src = None, None, None, None, None
    if rank == 0 and n == 50_000_000 and L == 100 and os.path.exists(pmc) and not sharded:   # (the profile is the plain path's: a sharded run reports null)
        rows = json.load(open(pmc))
        shards = max((r.get("shards") or 0) for r in rows) or None
        traffic_src = os.path.relpath(pmc, ROOT)
        if shards:
            step_traffic = int(sum((2 * r["FETCH_SIZE_KB"] + r["WRITE_SIZE_KB"]) * 1024 for r in rows) / shards)
        for r in rows:
            if kname in r["kernel"] and (k_traffic is None or r["calls"] > 0):
                k_traffic = int((2 * r["FETCH_SIZE_KB"] + r["WRITE_SIZE_KB"]) * 1024 / max(r["calls"], 1))
                break
    if rank == 0 and os.path.exists(sqf):
        for r in json.load(open(sqf)):
            if kname in r["kernel"] and r.get("symbols"):
                instr_per_symbol = (r["SQ_INSTS_VALU"] + r["SQ_INSTS_SALU"] + r["SQ_INSTS_LDS"]) / r["symbols"]
                issue_src = os.path.relpath(sqf, ROOT)
                break
    if rank == 0:
        total_in = nbytes * world
        ms_per_step = dt / args.steps * 1e3
        value = total_in * args.steps / dt / 1e6
        launches = max(k["launches"], 1)
        per_launch_ms = k["total_ms"] / launches
        k_alg = (k["bytes_in"] + k["bytes_out"]) / launches
        k_ach = k_alg / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        sym_per_launch = k["bytes_in"] / launches
        # compulsory traffic of a step = the record read once + the final streams written once
        alg_step = int(nbytes + out_bytes)
        step_ach = alg_step * world / (ms_per_step * 1e-3) / 1e9
        issue = None
        if instr_per_symbol is not None and per_launch_ms > 0:
            ips = instr_per_symbol * sym_per_launch / (per_launch_ms * 1e-3)
            issue = {"achieved": round(ips / 1e9, 2), "peak": round(1024 * 2.4, 1), "unit": "Ginstr/s", "frac": round(ips / (1024 * 2.4e9), 4),
                     "instr_per_symbol": round(instr_per_symbol, 3), "issue_source": issue_src}
        parity_checked = bool(parity) and all(v.get("ok") for v in parity.values())
        line = {
            "metric": "input FASTQ MB/s compressed, 100 bp reads, bit-exact decompress",
            "value": round(value, 2),
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": warm,
            "ms_per_step": round(ms_per_step, 3),
            "value_single_job": round(nbytes / single_ms / 1e3, 2) if single_ms else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "parity_checked": parity_checked,
            "parity": parity or None,
            "config": {"workload": f"{n} x {L} bp single-end synthetic FASTQ per GPU, arithmetic-coded qualities "
                                   "(BASELINE.json configs[1])", "reads_per_gpu": n, "read_length": L,
                       "input_bytes_per_gpu": nbytes, "output_bytes_per_gpu": int(out_bytes),
                       "core_table": "tests/golden/patterns.bin (15600 cores)", "parallelism": f"shard{world}" + ("" if not sharded else f": read ranges per rank, ONE archive; run-wide -B chunks / tie-break / quality model / 10 MiB blocks over {comm.world} rank(s) of " + ("shared memory (rehearsal)" if os.environ.get("SCALCE_COMM") == "shm" else "RCCL (all-gather, all-reduce, send/recv)")),
                       "bucket_set_size": B, "spill_chunks": stats["chunks"],
                       "tie_reads": stats["tie_reads"], "jacobi_iters": stats["jacobi_iters"],
                       "shards_in_flight": D, "shards_per_coder_launch": (round(args.steps / max(len(launch_plan(args.steps)), 1), 2) if waves else G),
                       "launch_plan": ({"kind": "waves", "launches": launch_plan(args.steps),
                                        "what": "every coder launch but the first takes all slots and all CUs; the remainder of the run goes first, beside the front stages of the first full wave"}
                                       if waves else {"kind": "eager", "shards_per_launch": G}),
                       "coder_streams": n_coder_streams, "front_threads": F,
                       "coded_in_place": bool(in_place),
                       "inputs": ("one text tensor per shard in flight: %d distinct synthetic shards of the same shape (seeds %d + 7919 k), "
                                  "each resident in HBM before the timed region; step j reads the text of slot j mod %d" % (D, SEED0, D)) if own_text
                                 else "ONE text tensor read by every shard in flight (--shared-input)",
                       "hbm_used_gb": hbm_used_gb, "ms_single_shard_alone": round(single_ms, 3) if single_ms else None},
            # achieved = algorithmic bytes of a step (the FASTQ record read once + the three archive streams
            # written once: 308 B per read) / ms_per_step, against the HBM peak.  `kernel` = the dominant kernel on its own:
            # algorithmic bytes of a launch (symbols in + coded bytes out) / its HIP-event time, its counter traffic, and the
            # issue-slot view (it is a serial chain per 10 MiB block: bound by instruction issue, not by HBM).
            "roofline": {"bound": "hbm", "achieved": round(step_ach, 2), "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                         "frac": round(step_ach / (HBM_PEAK_GBS * world), 5),
                         "traffic": step_traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_step": alg_step, "bytes_per_read": round(alg_step / n, 1),
                         "kernel": {"name": kname, "launch_ms": round(per_launch_ms, 3),
                                    "shards_per_launch": (round(args.steps / max(len(launch_plan(args.steps)), 1), 2) if waves else G),
                                    "bound": "issue",
                                    "achieved": round(k_ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k_ach / HBM_PEAK_GBS, 6),
                                    "alg_bytes_per_launch": int(k_alg), "traffic": k_traffic,
                                    "issue": issue,
                                    "ns_per_symbol_per_block": round(per_launch_ms * 1e6 / min(max(sym_per_launch, 1), 10 * 1024 * 1024), 2),
                                    "note": "serial coder chain per 10 MiB block: the time of a launch is 10.5 M steps of one wavefront, "
                                            "whatever the number of blocks beside it"}},
            "cpu_baseline": cpu,
            "decode": (dict(decode, cpu_baseline=(cpu or {}).get("decompress")) if decode else None),
            "e2e": e2e,
            "table_scale": table_scale,
            "note": "value = %d shards (independent jobs of the configs[1] size) through the device-resident hot path, %d in flight, each "
                    "with an input text of its own, fill and drain of the pipeline inside the timed region%s; value_single_job = one such job "
                    "alone, input already in HBM; decode = the inverse path on the last shard; e2e = the scalce binary, file in, archive out"
                    % (args.steps, D, (" (coder launches planned for a run of this length: %s shards)" % " + ".join(str(x) for x in launch_plan(args.steps))) if waves else ""),
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()   # (rank 0's CPU leg and checks run while the others wait here)
        dist.destroy_process_group()


def end_to_end(text, nbytes):
    """File -> archive with the `scalce` binary (C++ streaming host: reader threads, pinned chunks, uploads beside the
    ingest of the previous piece) on the SAME shard written to a file in /dev/shm (tmpfs: what is measured is the host
    path, not a disk).  Never `value`: PCIe, file reads and writes and process start-up are all inside."""
    import re
    import shutil
    cli = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    try:
        free = shutil.disk_usage(base).free
    except OSError:
        free = 0
    if not os.path.exists(cli) or free < 2 * nbytes:
        return None
    d = tempfile.mkdtemp(prefix="scalce_e2e_", dir=base)
    try:
        fq = os.path.join(d, "in_1.fq")
        with open(fq, "wb") as f:
            step = 1 << 30
            for a in range(0, nbytes, step):
                f.write(text[a:min(nbytes, a + step)].cpu().numpy().tobytes())
        # (a file another process has JUST written to tmpfs reads at a third of its speed the first time -- 0.6 s of "waiting for
        #  the reader" in the first run over it and 0.01 s in every later one, whatever the number of reader threads: the file is
        #  read once before the timed runs, so that what is timed is the binary and not the page cache settling)
        with open(fq, "rb") as f:
            while f.read(256 << 20):
                pass
        out = {}
        for cont in ("no", "gz"):
            t0 = time.perf_counter()
            r = subprocess.run([cli, "-c", cont, "-o", os.path.join(d, "arc_" + cont), fq, "--patterns-bin",
                                os.path.join(ROOT, "tests", "golden", "patterns.bin")], capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr[-300:]}
            m = re.search(r"Time elapsed: (.*)", r.stderr)
            asz = sum(os.path.getsize(os.path.join(d, f"arc_{cont}_1.scalce{e}")) for e in "nrq")
            out["c_" + cont] = {"value": round(nbytes / dt / 1e6, 1), "unit": "MB/s", "wall_s": round(dt, 3), "archive_bytes": asz,
                                "where": m.group(1) if m else None}
        out["input"] = f"the bench shard as a file in {base} ({nbytes} bytes, read once before the timed runs), process start to exit"
        # gzipped input (FASTQ arrives gzipped; the reference opens every input through its gz reader, compress.cpp:756): the
        # same shard as a multi-member .gz (4 MiB of text per member, level 1 -- what bgzip / pigz -i style writers and this
        # repo's own -c gz containers produce), inflated member by member on the host's threads (csrc/pargz.hpp)
        import zlib
        from concurrent.futures import ThreadPoolExecutor
        gzp = os.path.join(d, "in_1.fq.gz")
        tz0 = time.perf_counter()
        MEMBER = 4 << 20

        def member(a):
            c = zlib.compressobj(1, zlib.DEFLATED, 31)
            return c.compress(text[a:min(nbytes, a + MEMBER)].cpu().numpy().tobytes()) + c.flush()
        os.remove(fq)
        with open(gzp, "wb") as f, ThreadPoolExecutor(16) as pool:
            for blob in pool.map(member, range(0, nbytes, MEMBER)):
                f.write(blob)
        tz1 = time.perf_counter()
        r = subprocess.run([cli, "-c", "no", "-o", os.path.join(d, "arc_gzin"), gzp, "--patterns-bin",
                            os.path.join(ROOT, "tests", "golden", "patterns.bin")], capture_output=True, text=True)
        dt = time.perf_counter() - tz1
        if r.returncode != 0:
            out["gz_in"] = {"error": r.stderr[-300:]}
        else:
            m = re.search(r"Time elapsed: (.*)", r.stderr)
            same = all(open(os.path.join(d, f"arc_gzin_1.scalce{e}"), "rb").read(1 << 20) == open(os.path.join(d, f"arc_no_1.scalce{e}"), "rb").read(1 << 20)
                       and os.path.getsize(os.path.join(d, f"arc_gzin_1.scalce{e}")) == os.path.getsize(os.path.join(d, f"arc_no_1.scalce{e}")) for e in "nrq")
            out["gz_in"] = {"value": round(nbytes / dt / 1e6, 1), "unit": "MB/s of FASTQ text", "wall_s": round(dt, 3),
                            "gz_bytes": os.path.getsize(gzp), "same_archive_as_plain_input": same,
                            "input": f"multi-member gzip, {MEMBER >> 20} MiB of text per member, written in {tz1 - tz0:.1f} s", "where": m.group(1) if m else None}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def table_scale_leg(dev, reads=2_000_000, L=100):
    """The tokenize stage against a core table of realistic size (the headline's table has 15 600 cores; the
    reference's loader admits five million, reads.cpp:336): 1 M cores of 12-32 bases (tests/bigtable.py, 9.9 M automaton
    states), reads with planted cores.  Stage time by HIP events, second run (tokens are checked against the oracle's trie
    walk by tests/test_gpu_scale.py::test_million_core_table, not here)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bigtable
    from scalce_amd import host, synth
    blob, vals = bigtable.build()
    ctx = host.Context(dev.index or 0, patterns_bin=blob)
    bases = bigtable.reads_with_cores(reads, L, vals)
    quals = np.full((reads, L), ord("I"), dtype=np.uint8)
    fq = synth.fastq_bytes_fast(bases, quals)
    t = torch.frombuffer(bytearray(fq), dtype=torch.uint8).to(dev)
    b = host.Batch(ctx, L, reads + 8, len(fq) + 64)
    for _ in range(2):
        b.stage_reset(True)
        b.front(t.data_ptr(), len(fq))
        torch.cuda.synchronize()
    ms = b.stage_ms()["tokenize"][0]
    st = b.stats()
    b.close()
    return {"cores": int(sum(len(v) for _, v in vals)), "core_lengths": "12-32", "automaton_states": int(ctx.n_states), "reads": reads,
            "tokenize_stage_ms": round(ms, 2), "ns_per_read": round(ms * 1e6 / reads, 3), "ms_per_50M_reads": round(ms * 50e6 / reads, 1),
            "tie_reads": st["tie_reads"],
            "what": "scalce_batch_tokenize (both walks + the exact tie-break) on a 1 M-core table; occurrences are found from their starts "
                    "(tokenize_anchor_k: K-mer bitmap, then down the trie), not by walking the automaton"}


def ref_full_shard(text, nbytes):
    """--ref-full-shard: the reference's own compress() -T 1 on the WHOLE bench shard (as a file in /dev/shm) beside the
    `scalce` binary on the same file; SHA-256 of the three archive files.  Minutes of CPU: off by default (the recorded
    result of the same run is what parity.reference_full_size checks the timed shard against)."""
    import hashlib
    import shutil
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_full")
    cli = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
    pbin = os.path.join(ROOT, "tests", "golden", "patterns.bin")
    if not (os.path.exists(ref) and os.path.exists(cli)):
        return {"ok": False, "error": "oracle/_ref/ref_full or the scalce binary is missing"}
    d = tempfile.mkdtemp(prefix="scalce_refshard_", dir="/dev/shm" if os.access("/dev/shm", os.W_OK) else None)
    try:
        fq = os.path.join(d, "in_1.fq")
        with open(fq, "wb") as f:
            for a in range(0, nbytes, 1 << 30):
                f.write(text[a:min(nbytes, a + (1 << 30))].cpu().numpy().tobytes())
        p = subprocess.Popen([ref, "compress", pbin, fq, os.path.join(d, "ref"), "-c", "no", "-T", "1", "-t", os.path.join(d, "tmp")],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        t0 = time.perf_counter()
        r = subprocess.run([cli, "-c", "no", "-o", os.path.join(d, "hip"), fq, "--patterns-bin", pbin], capture_output=True, text=True)
        if r.returncode:
            p.kill()
            return {"ok": False, "error": r.stderr[-300:]}
        while p.poll() is None:
            time.sleep(15)
            print("bench: --ref-full-shard: the reference has been running for %.0f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
        if p.returncode:
            return {"ok": False, "error": "ref_full failed"}
        same = {}
        for e in "nrq":
            hs = []
            for side in ("hip", "ref"):
                h = hashlib.sha256()
                with open(os.path.join(d, f"{side}_1.scalce{e}"), "rb") as f:
                    for blk in iter(lambda: f.read(64 << 20), b""):
                        h.update(blk)
                hs.append(h.hexdigest())
            same["scalce" + e] = hs[0] == hs[1]
        return {"ok": all(same.values()), "files": same, "reference_seconds": round(time.perf_counter() - t0, 1),
                "what": "`scalce -c no` and the reference's own compress() -T 1 on the whole bench shard as a file: SHA-256 of .scalce{n,r,q}"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(text, n, L, sample):
    """Time the CPU side on the first `sample` records of the same shard, on this box's host cores, and check the
    product against it on the same bytes.  Returns (cpu_baseline object, parity object).

    kind "reference": oracle/_ref/ref_full -- the reference's OWN compress() (compress.cpp:721; every reference source
    but main.cpp compiled where it lies, in the dev container; the binary travels with the repo), file in, archive out,
    at -T 1 (deterministic: the parity contract) and at the reference's default thread count (main.cpp:171).  Falls back
    to kind "port" (oracle/orc_cli, the plain-C restatement) where that binary is missing.
    Parity: the `scalce` binary compresses the same sample file; its three archive files must be byte-identical with the
    -T 1 files of the CPU leg."""
    import hashlib

    import numpy as np
    sample = min(sample, n)
    approx = sample * (2 * L + 8 + len(str(sample)))
    head = text[: min(text.numel(), approx + 4096)].cpu().numpy()
    nl = np.flatnonzero(head == 10)
    sample = min(sample, len(nl) // 4)
    end = int(nl[4 * sample - 1]) + 1
    pbin = os.path.join(ROOT, "tests", "golden", "patterns.bin")
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_full")
    exe = os.path.join(ROOT, "oracle", "orc_cli")
    cli = os.path.join(ROOT, "scalce_amd", "bin", "scalce")
    use_ref = os.path.exists(ref) and os.access(ref, os.X_OK)
    if not use_ref and not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    T = max(1, min(4, (os.cpu_count() or 2) - 1))
    base = "/dev/shm" if os.access("/dev/shm", os.W_OK) else None
    with tempfile.TemporaryDirectory(dir=base) as d:
        fq = os.path.join(d, "s_1.fq")
        head[:end].tofile(fq)

        def run_cpu(threads, out):
            if use_ref:
                cmd = [ref, "compress", pbin, fq, os.path.join(d, out), "-c", "no", "-T", str(threads), "-t", os.path.join(d, "tmp_" + out)]
            else:
                cmd = [exe, "compress", pbin, fq, os.path.join(d, out), "-c", "no", "-T", str(threads)]
            t0 = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, text=True)
            return time.perf_counter() - t0, r.returncode

        dt, rc = run_cpu(1, "cpu1")
        if rc != 0 and use_ref:
            use_ref = False
            dt, rc = run_cpu(1, "cpu1")
        if rc != 0:
            raise RuntimeError("CPU leg failed")
        dt4, rc4 = run_cpu(T, "cpuT") if T > 1 else (None, 1)
        # BASELINE.md section 2, line C1 as written: 1 M reads, -T 1 -c gz (BASELINE.json configs[0]; 17.8 MB/s in the survey's
        # container) -- the same-node counterpart of that table
        c1 = None
        if use_ref and sample >= 1_000_000:
            end1 = int(nl[4 * 1_000_000 - 1]) + 1
            fq1 = os.path.join(d, "c1_1.fq")
            head[:end1].tofile(fq1)
            t0 = time.perf_counter()
            r = subprocess.run([ref, "compress", pbin, fq1, os.path.join(d, "c1"), "-c", "gz", "-T", "1", "-t", os.path.join(d, "tmp_c1")],
                               capture_output=True, text=True)
            dtc = time.perf_counter() - t0
            if r.returncode == 0:
                c1 = {"value": round(end1 / dtc / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "reference",
                      "sample": f"BASELINE.json configs[0]: first 1000000 records ({end1} bytes), ref_full compress -c gz -T 1, {dtc:.2f} s wall"}
            os.remove(fq1)
        what = ("oracle/_ref/ref_full compress -c no (the reference's own compress(), file in, archive out)" if use_ref
                else "orc_cli compress -c no (C restatement)")
        dec = None
        if use_ref:   # the reference's own decompress() on the archive it has just written: the baseline of the decode leg
            t0 = time.perf_counter()
            r = subprocess.run([ref, "decompress", pbin, os.path.join(d, "cpu1_1.scalcen"), os.path.join(d, "back"), "-T", "1"], capture_output=True, text=True)
            ddt = time.perf_counter() - t0
            if r.returncode == 0 and os.path.exists(os.path.join(d, "back_1.fastq")):
                dec = {"value": round(os.path.getsize(os.path.join(d, "back_1.fastq")) / ddt / 1e6, 2), "unit": "MB/s of FASTQ restored", "cores": 1,
                       "kind": "reference", "sample": f"oracle/_ref/ref_full decompress -T 1 (the reference's own decompress()) on the archive of the same "
                                                      f"{sample}-record sample, {ddt:.2f} s wall incl. file I/O on tmpfs"}
                os.remove(os.path.join(d, "back_1.fastq"))
        if not use_ref:
            print("bench: oracle/_ref/ref_full is missing or did not run on this box: cpu_baseline and parity.sample use the C restatement "
                  "(oracle/orc_cli, kind \"port\"), NOT the reference's own code", file=sys.stderr)
        cpu = {"value": round(end / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "reference" if use_ref else "port",
               "decompress": dec,
               "sample": f"first {sample} records ({end} bytes) of the same shard, {what} -T 1, {dt:.2f} s wall incl. file I/O on tmpfs",
               "host_cpus": os.cpu_count(), "configs0_gz": c1,
               "threads_default": None if rc4 != 0 else {
                   "value": round(end / dt4 / 1e6, 2), "unit": "MB/s", "cores": T, "kind": "reference" if use_ref else "port",
                   "sample": f"same sample, -T {T} (main.cpp:171 default thread count"
                             + ("; the reference's output at -T > 1 is racy, its time is not" if use_ref else "; coder blocks on threads, record loop on one")
                             + f"), {dt4:.2f} s wall"}}
        # the product on the same bytes
        parity = {"ok": False, "what": f"`scalce -c no` on the same {sample}-record file: .scalce{{n,r,q}} byte-identical with the "
                                       + ("reference's" if use_ref else "C restatement's") + " -T 1 files"}
        if os.path.exists(cli):
            r = subprocess.run([cli, "-c", "no", "-o", os.path.join(d, "hip"), fq, "--patterns-bin", pbin], capture_output=True, text=True)
            if r.returncode != 0:
                parity["error"] = r.stderr[-300:]
            else:
                same = {}
                for e in "nrq":
                    ha = hashlib.sha256(open(os.path.join(d, f"hip_1.scalce{e}"), "rb").read()).hexdigest()
                    hb = hashlib.sha256(open(os.path.join(d, f"cpu1_1.scalce{e}"), "rb").read()).hexdigest()
                    same["scalce" + e] = ha == hb
                parity["files"] = same
                parity["ok"] = all(same.values())
        else:
            parity["error"] = "scalce binary not built"
    return cpu, parity


if __name__ == "__main__":
    main()
