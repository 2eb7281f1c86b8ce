#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files committed under profiles/.

usage: summarize_pmc.py <round-tag> <trace_dir> <fetch_dir> <write_dir> [git-head]
  git-head  : commit the profiled build was made from (bench.py quotes it beside `roofline.traffic`)
  trace_dir : rocprofv3 --kernel-trace --stats --output-format csv
  fetch_dir : rocprofv3 --pmc FETCH_SIZE   (its own pass)
  write_dir : rocprofv3 --pmc WRITE_SIZE   (its own pass)
HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced
streaming read, so the read side is doubled; WRITE_SIZE is exact for 16-B streaming stores.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counter_avg(d, name):
    f = find(d, "_counter_collection.csv")
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v), max(v)) for k, v in agg.items()}


def main():
    tag, trace, fetch, write = sys.argv[1:5]
    git_head = sys.argv[5] if len(sys.argv) > 5 else os.environ.get("MVDB_GIT_HEAD")
    here = os.path.dirname(os.path.abspath(__file__))
    ks = find(trace, "_kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(here, f"{tag}_kernel_stats.csv"))
    fs, ws = counter_avg(fetch, "FETCH_SIZE"), counter_avg(write, "WRITE_SIZE")
    stats = {}
    if ks:
        for r in csv.DictReader(open(ks)):
            stats[r["Name"]] = r
    out = []
    for kern in sorted(set(fs) | set(ws)):
        nf, favg, fmax = fs.get(kern, (0, 0.0, 0.0))
        nw, wavg, wmax = ws.get(kern, (0, 0.0, 0.0))
        st = stats.get(kern, {})
        out.append({
            "kernel": kern,
            "launches_fetch_pass": nf, "FETCH_SIZE_KiB_avg": favg, "FETCH_SIZE_KiB_max": fmax,
            "launches_write_pass": nw, "WRITE_SIZE_KiB_avg": wavg, "WRITE_SIZE_KiB_max": wmax,
            "hbm_read_bytes_avg_corrected_x2": favg * 1024 * 2,
            "hbm_write_bytes_avg": wavg * 1024,
            "hbm_traffic_bytes_per_launch_avg": favg * 1024 * 2 + wavg * 1024,
            "trace_calls": int(st.get("Calls", 0) or 0),
            "trace_avg_ns": float(st.get("AverageNs", 0) or 0),
            "git_head": git_head,
        })
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_summary.json"), "w"), indent=1)
    for o in out:
        print(o["kernel"][:60], o["hbm_traffic_bytes_per_launch_avg"], o["trace_avg_ns"])


if __name__ == "__main__":
    main()
