#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

  prof_summary.py stats <dir> <out.csv>           kernel-trace --stats run: per-kernel calls / total / avg / min / max (ms)
  prof_summary.py pmc <fetch_dir> <write_dir> <out.json> [shards]   two --pmc runs (FETCH_SIZE, WRITE_SIZE): KB per kernel
"""
import csv, glob, json, os, sys
from collections import defaultdict


def trace_rows(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    return list(csv.DictReader(open(f[0])))


def short(name):
    return name.split("(")[0].replace("void ", "")[:60]


def stats(d, out):
    acc = defaultdict(list)
    for r in trace_rows(d):
        if "scalce::" not in r["Kernel_Name"]:
            continue
        acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    tot = sum(sum(v) for v in acc.values())
    with open(out, "w") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "percent"])
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v), 3), round(sum(v) / len(v), 4), round(min(v), 4), round(max(v), 4), round(100 * sum(v) / tot, 2)])


def counters(d, name):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc, calls = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != name or "scalce::" not in r["Kernel_Name"]:
            continue
        acc[short(r["Kernel_Name"])] += float(r["Counter_Value"])
        calls[short(r["Kernel_Name"])] += 1
    return acc, calls


def pmc(fd, wd, out, shards=None):
    fe, calls = counters(fd, "FETCH_SIZE")
    wr, _ = counters(wd, "WRITE_SIZE")
    rows = [{"kernel": k, "calls": calls[k], "FETCH_SIZE_KB": fe[k], "WRITE_SIZE_KB": wr.get(k, 0.0)} for k in fe]
    if shards is None:  # one ingest_tiles call per shard that went through the front stages
        shards = next((r["calls"] for r in rows if "ingest_tiles" in r["kernel"]), None)
    if shards:  # shards the profiled run pushed through every stage (a coder launch may hold several)
        for r in rows:
            r["shards"] = shards
    rows.sort(key=lambda r: -(r["FETCH_SIZE_KB"] + r["WRITE_SIZE_KB"]))
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else None)
