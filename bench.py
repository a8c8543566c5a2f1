#!/usr/bin/env python3
"""bench.py -- input FASTQ MB/s of the SCALCE hot path on MI355X (BASELINE.json metric).

A step = one pass of the whole hot path (ingest -> quality statistics -> tokenize with exact tie-break
-> bucket/reorder -> emit -> arithmetic coder) over one synthetic shard that is already resident in HBM.
N = 1 runs BASELINE.json configs[1]: 50 M x 100 bp single-end, arithmetic-coded qualities.  N > 1 is weak
scaling: every rank holds a shard of the same size (one process per GPU) and the ranks produce ONE archive --
scalce_sharded_compress, the C++ host of scalce_amd/csrc/sharded.cpp over RCCL (comm.cpp): run-wide -B chunks,
tie-break, quality model and 10 MiB block cutting, byte-identical with the one-GPU archive of the same input.
torch.distributed is only used to hand RCCL's unique id to the ranks and for the timing barrier.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (ac_encode_k): algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  cpu_baseline -- the reference's own hot-path objects (oracle/_ref, else the C port), one thread, on a bounded sample
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
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
    text = synth_gpu.fastq_on_device(n, L, dev, seed=20261003 + rank, first_index=rank * n)
    nbytes = text.numel()
    # quality model from the first 100 000 records of the shard (quality_mapping_init's sample)
    head = text[: min(nbytes, 100000 * (2 * L + 20))].cpu().numpy().tobytes()
    from scalce_amd import format as fmt
    off, vals, Ls = fmt.sample_qmap(head)
    assert Ls == L
    # shards per coder launch / shards in flight: 3 / 6 (199 GB of the 288 GB HBM at 50 M reads per shard on one GPU).  A
    # sharded run also holds every in-flight shard's block range of the run-wide quality stream and the all-to-all
    # buffers.  Measured with the 2-rank rehearsal (tools/mem_probe.sh; 20 M and 28 M reads, 4 and 6 in flight), per rank:
    # 8 GB + 0.4 GB per million reads + 0.73 GB per million reads and shard in flight (x L / 100) = 246 GB at 50 M reads
    # and six in flight.  A rank keeps 3 / 6 when that estimate plus a margin fits what is free now, else 2 / 4 (174 GB).
    # A sharded run goes through scalce_sharded_compress (C++ host, RCCL).  SCALCE_BENCH_FORCE_SHARDED=1 takes that path at
    # world 1 as well (every collective is then a real RCCL call of one rank); SCALCE_COMM=shm rehearses several ranks on
    # ONE GPU through the shared-memory transport.
    sharded = world > 1 or bool(os.environ.get("SCALCE_BENCH_FORCE_SHARDED"))
    comm = None
    if sharded:
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout is the one JSON line's, so
        # the banner goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if os.environ.get("SCALCE_COMM") == "shm":
                name = [("/scalce_bench_%d" % os.getpid()) if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(name, src=0)
                comm = host.Comm(local, world, rank, shm_name=name[0], slot_bytes=int(os.environ.get("SCALCE_SHM_SLOT", str(8 << 30))))
            else:
                uid = [host.Comm.unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(uid, src=0)
                comm = host.Comm(local, world, rank, unique_id=uid[0])
            comm.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    # shards per coder launch / shards in flight: 3 / 6 (135 GB of the 288 GB HBM at 50 M reads per shard with the shared
    # front-stage buffers).  A sharded run also holds every in-flight shard's block range of the run-wide quality stream
    # (5 GB each) and, while a shard is worked on, the text it received from its neighbours and the all-to-all buffers
    # (about 25 GB): 190 GB.  Larger groups do not pay yet: from four shards per launch on the front stages, not the coder,
    # bound the pipeline.
    G = args.group
    if G is None:
        G = 3
    G = max(1, G)
    D = max(1, args.inflight) if args.inflight is not None else 2 * G
    if G > 1:
        D = max(D, 2 * G)
    # -B: the reference's default, 4 GiB of record bytes per spill chunk (main.cpp:68) -- 50 M reads of 100 bp are 3 chunks
    B = int(os.environ.get("SCALCE_BENCH_BUCKET_SET", str(4 << 30)))
    # the shards in flight share ONE set of front-stage buffers (rows, tokens, events, sort scratch: dead once a shard is
    # emitted, and front stages run one at a time on the front stream): 15 GB instead of 35 GB of HBM per shard in flight
    shared_ws = None if os.environ.get("SCALCE_BENCH_OWN_WORKSPACES") else host.Workspace(ctx)
    batches = [host.Batch(ctx, L, max_reads=n + 8, max_text=nbytes + 64, qmap=[(off, vals), (off, vals)], bucket_set_size=B,
                          workspace=shared_ws) for _ in range(D)]
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
    pipe = ShardPipeline(batches, group=G, sharded=sharded, trace=mark if trace else None,
                         coder_streams=int(os.environ.get("SCALCE_BENCH_CODER_STREAMS", "1")))
    front = pipe.front

    def run(k):
        for j in range(k):
            slot, b = pipe.acquire()
            mark(f"shard {j}: front (slot {slot})")
            with torch.cuda.stream(pipe.front):
                if not sharded:
                    b.front(text.data_ptr(), nbytes, None, 0, pipe.front.cuda_stream)
                elif G == 1:
                    state[slot] = host.sharded_compress(comm, ctx, b, text.data_ptr(), nbytes, flags=host.SHARD_CODER_ASYNC,
                                                        stream=pipe.front.cuda_stream, coder_stream=pipe.coder.cuda_stream,
                                                        result=state.get(slot))
                else:
                    state[slot] = host.sharded_compress(comm, ctx, b, text.data_ptr(), nbytes, flags=host.SHARD_PREPARE_ONLY,
                                                        stream=pipe.front.cuda_stream, result=state.get(slot))
            mark(f"shard {j}: front done")
            pipe.submit(slot, tag=j, flush=j + 1 == k)
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

    if args.stage_times and rank == 0:
        batch.stage_reset(True)
        batch.compress(text.data_ptr(), nbytes, None, 0, front.cuda_stream)
        batch.finish(front.cuda_stream)
        print("stage ms:", {s: round(v[0], 2) for s, v in batch.stage_ms().items()}, stats, file=sys.stderr)
        batch.stage_reset(False)

    cpu = None
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        try:
            cpu = cpu_baseline(text, n, L, args.cpu_sample)
        except Exception as ex:  # noqa: BLE001
            cpu = {"error": repr(ex)[:300]}
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        for b in batches:   # the CLI is a process of its own and needs the card's memory
            b.close()
        del batches[:], pipe
        try:
            e2e = end_to_end(text, nbytes)
        except Exception as ex:  # noqa: BLE001 - a side leg must not take the measured line with it
            e2e = {"error": repr(ex)[:300]}

    traffic, traffic_src = None, None
    kname = "ac_encode_rows_k" if G > 1 else "ac_encode_k"
    # HBM bytes of the dominant kernel from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, units of KB;
    # FETCH_SIZE doubled: gfx950 reports half of a streaming read, MI355X_MICROARCH.md "HBM").  The counters come from a
    # profile kept under profiles/ -- of this round's build when PROFILE_TAG names one, and the tag is printed with them.
    tag = os.environ.get("SCALCE_PROFILE_TAG", "r02_final")
    pmc = os.path.join(ROOT, "profiles", f"{tag}_bench50m_pmc_fetch_write.json")
    if rank == 0 and n == 50_000_000 and L == 100 and os.path.exists(pmc):
        nblocks = G * ((n * L + 10 * 1024 * 1024 - 1) // (10 * 1024 * 1024))
        want = (kname + ("<false, 8>" if nblocks > 1024 else "<false, 16>")) if G > 1 else kname + "<"
        for row in json.load(open(pmc)):
            if want in row["kernel"]:
                launches = max(row["calls"], 1)
                traffic = int((2 * row["FETCH_SIZE_KB"] + row["WRITE_SIZE_KB"]) * 1024 / launches)
                traffic_src = os.path.relpath(pmc, ROOT)
    if rank == 0:
        total_in = nbytes * world
        ms_per_step = dt / args.steps * 1e3
        value = total_in * args.steps / dt / 1e6
        per_launch_ms = k["total_ms"] / max(k["launches"], 1)
        alg_bytes = (k["bytes_in"] + k["bytes_out"]) / max(k["launches"], 1)
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
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
            "config": {"workload": f"{n} x {L} bp single-end synthetic FASTQ per GPU, arithmetic-coded qualities "
                                   "(BASELINE.json configs[1])", "reads_per_gpu": n, "read_length": L,
                       "input_bytes_per_gpu": nbytes, "output_bytes_per_gpu": int(out_bytes),
                       "core_table": "tests/golden/patterns.bin (15600 cores)", "parallelism": f"shard{world}" + ("" if not sharded else f": read ranges per rank, ONE archive; run-wide -B chunks / tie-break / quality model / 10 MiB blocks over {comm.world} rank(s) of " + ("shared memory (rehearsal)" if os.environ.get("SCALCE_COMM") == "shm" else "RCCL (all-gather, all-reduce, send/recv)")),
                       "bucket_set_size": B, "spill_chunks": stats["chunks"],
                       "tie_reads": stats["tie_reads"], "jacobi_iters": stats["jacobi_iters"],
                       "shards_in_flight": D, "shards_per_coder_launch": G, "hbm_used_gb": hbm_used_gb, "ms_single_shard_alone": round(single_ms, 3) if single_ms else None},
            # The dominant kernel is a serial coder chain per 10 MiB block: its roof is the chip's instruction issue rate, not
            # HBM.  achieved = instructions it issues per second -- 6.3 per symbol (4.8 VALU + 1.3 SALU + 0.2 LDS,
            # rocprofv3 --pmc SQ_INSTS_*, profiles/r01 pmc_sq) x symbols per launch / launch time -- against 1024 SIMDs x
            # 2.4 GHz.  The HBM view of the same launch (algorithmic bytes / time against 8 TB/s) is kept beside it.
            "roofline": {"bound": "issue", "kernel": kname,
                         "achieved": round(6.3 * k["bytes_in"] / max(k["launches"], 1) / (per_launch_ms * 1e-3) / 1e9, 2) if per_launch_ms > 0 else None,
                         "peak": round(1024 * 2.4, 1), "unit": "Ginstr/s",
                         "frac": round(6.3 * k["bytes_in"] / max(k["launches"], 1) / (1024 * 2.4e9 * per_launch_ms * 1e-3), 4) if per_launch_ms > 0 else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "hbm": {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                                 "alg_bytes_per_launch": int(alg_bytes)},
                         "launch_ms": round(per_launch_ms, 3),
                         "ns_per_symbol_per_block": round(per_launch_ms * 1e6 / min(max(k["bytes_in"] / max(k["launches"], 1), 1), 10 * 1024 * 1024), 2),
                         "shards_per_launch": G,
                         "note": "serial coder chain per 10 MiB block: bound by the issue slots of one wavefront per eight blocks (ns per "
                                 "symbol per block is the figure to watch); blocks run concurrently, "
                                 + ("four or eight per chain wave, one launch for %d shards at one workgroup per CU" % G
                                    if G > 1 else "one 2-wave workgroup each")},
            "cpu_baseline": cpu,
            "e2e": e2e,
            "note": "value = device-resident steady state with %d shards (independent jobs of the configs[1] size) in flight; "
                    "value_single_job = one such job alone, input already in HBM; e2e = the scalce binary, file in, archive out" % D,
        }
        print(json.dumps(line))
    if world > 1:
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
        out["input"] = f"the bench shard as a file in {base} ({nbytes} bytes), process start to exit"
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_baseline(text, n, L, sample):
    """Time the CPU side on the first `sample` records of the same shard, on this box's host cores.

    kind "reference": oracle/_ref/ref_driver -t -- the reference's OWN hot-path objects (aho_search, output_read,
    output_quality, aho_trie_bucket, bin_prepare, ac_coder; built in the dev container from /root/reference, the
    binary travels with the repo) driven by a harness that plays main()/thread() at -T 1.  Falls back to kind
    "port" (oracle/orc_cli, the plain-C restatement) where that binary is missing."""
    import numpy as np
    sample = min(sample, n)
    approx = sample * (2 * L + 8 + len(str(sample)))
    head = text[: min(text.numel(), approx + 4096)].cpu().numpy()
    nl = np.flatnonzero(head == 10)
    sample = min(sample, len(nl) // 4)
    end = int(nl[4 * sample - 1]) + 1
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    exe = os.path.join(ROOT, "oracle", "orc_cli")
    use_ref = os.path.exists(ref) and os.access(ref, os.X_OK)
    if not use_ref and not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    with tempfile.TemporaryDirectory() as d:
        fq = os.path.join(d, "s_1.fq")
        head[:end].tofile(fq)
        t0 = time.perf_counter()
        if use_ref:
            r = subprocess.run([ref, fq, d, "-t"], capture_output=True, text=True)
            use_ref = r.returncode == 0
        if not use_ref:
            t0 = time.perf_counter()
            subprocess.run([exe, "compress", os.path.join(ROOT, "tests", "golden", "patterns.bin"), fq,
                            os.path.join(d, "o"), "-c", "no", "-T", "1"], check=True, capture_output=True)
        dt = time.perf_counter() - t0
        # second leg: the reference's default thread count, main.cpp:171.  The reference's own thread() is not
        # linkable here (needs buffio) and is racy at -T > 1; the C port codes the arithmetic-coder blocks on T threads and
        # keeps the record loop on one.
        T = max(1, min(4, (os.cpu_count() or 2) - 1))
        if not os.path.exists(exe):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
        t1 = time.perf_counter()
        r4 = subprocess.run([exe, "compress", os.path.join(ROOT, "tests", "golden", "patterns.bin"), fq,
                             os.path.join(d, "o4"), "-c", "no", "-T", str(T)], capture_output=True)
        dt4 = time.perf_counter() - t1 if r4.returncode == 0 else None
    what = ("oracle/_ref/ref_driver -t (the reference's own objects, one thread)" if use_ref
            else "orc_cli compress -c no -T 1 (C restatement, one thread)")
    return {"value": round(end / dt / 1e6, 2), "unit": "MB/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"first {sample} records ({end} bytes) of the same shard, {what}, {dt:.2f} s wall incl. file I/O",
            "host_cpus": os.cpu_count(),
            "threads_default": None if dt4 is None else {
                "value": round(end / dt4 / 1e6, 2), "unit": "MB/s", "cores": T, "kind": "port",
                "sample": f"same sample, orc_cli compress -c no -T {T} (main.cpp:171 default thread count; coder blocks on "
                          f"{T} threads, record loop on one), {dt4:.2f} s wall"}}


if __name__ == "__main__":
    main()
