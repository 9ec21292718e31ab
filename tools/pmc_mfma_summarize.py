#!/usr/bin/env python3
"""Summarise the MFMA-busy PMC pass of rocprofv3 into profiles/<prefix>_pmc_mfma_busy.json (read by bench.py).

    python tools/pmc_mfma_summarize.py <pass_dir> <out_prefix> [workload_key]

<pass_dir> holds *_counter_collection.csv and *_kernel_trace.csv of
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \\
              GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"""
import collections
import csv
import glob
import json
import os
import re
import sys



def _source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from visinger_amd.csrc import build
    return build.source_hash()


def main():
    pass_dir, prefix = sys.argv[1:3]
    workload = sys.argv[3] if len(sys.argv) > 3 else None
    kt = {}
    for path in glob.glob(os.path.join(pass_dir, "*kernel_trace.csv")):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                kt[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, cnt, seen = collections.defaultdict(float), collections.Counter(), set()
    for path in glob.glob(os.path.join(pass_dir, "*counter_collection.csv")):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "vs::" not in k:
                    continue
                m = re.search(r"vs::(\w+)(<[^>]*>)?", k)
                k = m.group(1) + (m.group(2) or "").replace(" ", "")
                rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dur[k] += kt.get(r["Dispatch_Id"], 0)
                    cnt[k] += 1
    out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY "
                     "GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline (its own "
                     "pass); mfma_tflops_executed = (SQ_INSTS_VALU_MFMA_MOPS_BF16 + _F16) x 512 FLOP / kernel time (one v_mfma_f32_32x32x16_bf16 = "
                     "64 MOPS = 32768 FLOP); mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); "
                     "gfx_clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel time", "kernels": {}}
    for k in sorted(dur, key=lambda k: -dur[k]):
        c = rows[k]
        if not c.get("SQ_VALU_MFMA_BUSY_CYCLES") or not dur[k]:
            continue
        t = dur[k] * 1e-9
        gui = c["GRBM_GUI_ACTIVE"] / 8
        d = {"launches": cnt[k], "total_ms": round(dur[k] / 1e6, 3), "gfx_clock_ghz": round(gui / t / 1e9, 3),
             "mfma_pipe_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui), 4),
             "wave_cycles_waiting_on_issue": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)}
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)      # (the split-f16 kernels issue f16 MFMAs)
        if mops:
            d["mfma_tflops_executed"] = round(mops * 512 / t / 1e12, 1)
            d["frac_of_bf16_peak_2500"] = round(d["mfma_tflops_executed"] / 2500, 4)
        out["kernels"][k] = d
        print(k, d)
    if workload:
        out["workload"] = workload
    with open(f"{prefix}_pmc_mfma_busy.json", "w") as f:
        out["vs_source_hash"] = _source_hash()      # the sources the profiled library was built from (a summary names its build)
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
