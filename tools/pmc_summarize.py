#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into the files bench.py / profiles/README.md cite.

    python tools/pmc_summarize.py <fetch_pass_dir> <write_pass_dir> <out_prefix> [workload_key [steps_in_pass]]

Each pass directory holds the `*_counter_collection.csv` of ONE `--pmc` counter (FETCH_SIZE or WRITE_SIZE: separate passes,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  Counter values are in KB.  Writes <out_prefix>_pmc_fetch_by_shape.csv,
<out_prefix>_pmc_write_by_shape.csv and <out_prefix>_pmc_traffic.json (HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE: the
gfx950 FETCH_SIZE correction of the guide)."""
import collections
import csv
import glob
import json
import os
import re
import sys


def load(pass_dir, counter):
    rows = collections.defaultdict(lambda: [0, 0.0])      # (kernel, grid) -> [launches, sum KB]
    for path in glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != counter:
                    continue
                key = (r["Kernel_Name"], int(r["Grid_Size"]))
                rows[key][0] += 1
                rows[key][1] += float(r["Counter_Value"])
    return rows


def short(name):
    """vs::kernel<template args>(params) -> kernel<args> without blanks (the spelling of the profiles/*.json keys)"""
    m = re.search(r"vs::(\w+)(<[^>]*>)?", name)
    if not m:
        return name
    return m.group(1) + (m.group(2) or "").replace(" ", "")



def _source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from visinger_amd.csrc import build
    return build.source_hash()


def main():
    fetch_dir, write_dir, prefix = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else None      # bench.py --print-workload-key: the launch shapes these bytes belong to
    fetch, write = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    for rows, cname, suffix in ((fetch, "FETCH_SIZE", "fetch"), (write, "WRITE_SIZE", "write")):
        with open(f"{prefix}_pmc_{suffix}_by_shape.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "grid_threads", "launches", f"avg_{cname}_KB"])
            for (k, g), (n, tot) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
                if "vs::" in k:
                    w.writerow([k, g, n, round(tot / n, 1)])
    per = {}
    for rows, field in ((fetch, "fetch"), (write, "write")):
        for (k, g), (n, tot) in rows.items():
            if "vs::" not in k:
                continue
            d = per.setdefault(short(k), {"fetch": [0, 0.0], "write": [0, 0.0]})
            d[field][0] += n
            d[field][1] += tot * 1024.0
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 1 "
                     "--warmup 1 --no-cpu-baseline; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibration "
                     "in profiles/README.md: x0.53 for 16-B-per-lane streams, x0.70-0.76 for the 4-B-per-lane staging loads, so "
                     "the doubled figure is an upper bound)", "kernels": {}}
    for k, d in per.items():
        nf, nw = d["fetch"][0], d["write"][0]
        if not nf or not nw:
            continue
        fr, wr = d["fetch"][1] / nf, d["write"][1] / nw
        out["kernels"][k] = {"launches": nf, "fetch_size_raw_bytes_per_launch": fr, "write_size_bytes_per_launch": wr,
                             "hbm_bytes_per_launch_corrected": 2 * fr + wr}
    if workload:
        out["workload"] = workload
    # the whole pass, EVERY kernel (the PyTorch-ROCm elementwise / reduction / library kernels of a training step included): what a whole-step bench line
    # (BASELINE config 3: no single dominant kernel) cites as its HBM traffic per step
    steps_in_pass = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    tot_f = sum(tot for (_, _), (n, tot) in fetch.items()) * 1024.0
    tot_w = sum(tot for (_, _), (n, tot) in write.items()) * 1024.0
    out["pass_total"] = {"fetch_size_raw_bytes": tot_f, "write_size_bytes": tot_w, "hbm_bytes_corrected": 2 * tot_f + tot_w,
                         "launches_fetch_pass": sum(n for (_, _), (n, tot) in fetch.items()), "steps_in_pass": steps_in_pass,
                         "hbm_bytes_corrected_per_step": (2 * tot_f + tot_w) / steps_in_pass if steps_in_pass else None}
    with open(f"{prefix}_pmc_traffic.json", "w") as f:
        out["vs_source_hash"] = _source_hash()      # the sources the profiled library was built from (a summary names its build)
        json.dump(out, f, indent=1)
    print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in out["kernels"].items()}, indent=1))


if __name__ == "__main__":
    main()
