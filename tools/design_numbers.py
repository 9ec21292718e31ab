#!/usr/bin/env python3
"""Fills the @NAME@ placeholders of DESIGN.md (sections 4.1 and 8) and README.md (Status) from the committed summaries of one profile tag (profiles/<tag>_*, <tag>_config{2,3,5}_*), so
that the numbers in the text are the numbers in the files.  Usage: python tools/design_numbers.py r05_b [--write]   (without --write: prints the values)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def J(name):
    return json.load(open(os.path.join(P, name)))


def stats(name):
    with open(os.path.join(P, name), newline="") as f:
        return list(csv.DictReader(f))


def values(tag):
    d = J(f"{tag}_bench_details.json")
    h, r = d["headline"], d["headline"]["roofline"]
    oc = d["other_configs"]
    c2, c3, c5 = oc["2"], oc["3"], oc["5"]
    dom = r["kernel"]
    key = dom.replace(" ", "")
    mf = J(f"{tag}_pmc_mfma_busy.json")["kernels"]
    k7 = next(v for k, v in mf.items() if k.startswith("conv_ktap_kernel<7,1,2,0,4,1,8"))
    tr = J(f"{tag}_pmc_traffic.json")["kernels"].get(key, {})
    ks = {row["Name"]: row for row in stats(f"{tag}_bench_kernel_stats.csv")}
    rp = ks[f"void vs::{dom}(vs::ConvParams)"]
    inst = r["all_instances"]

    def grp(*prefix):
        v = [x for k, x in inst.items() if k.startswith(prefix)]
        return sum(x["ms_per_step"] for x in v), min(x["tflops"] for x in v), max(x["tflops"] for x in v)

    def one(name):
        x = inst[name]
        return f"{x['ms_per_step']:.1f} / {x['tflops']:.0f}"

    rb = grp("resblock_f16_kernel")
    trc = grp("conv_split_tr_kernel", "conv_ktap_kernel<2, 1, 2, 4,")      # the transposed convs (k = 2 * stride on conv_ktap since round 5)
    k3 = stats(f"{tag}_config3_bench_kernel_stats.csv")
    tot3 = sum(float(x["TotalDurationNs"]) for x in k3)
    vs3 = sum(float(x["TotalDurationNs"]) for x in k3 if "vs::" in x["Name"] or x["Name"].startswith(("bias_grad_kernel", "l1_mean_", "wn_step_", "weight_norm_multi_")))
    calls3 = sum(int(x["Calls"]) for x in k3)
    steps3 = int(J(f"{tag}_config3_bench_line_profiled.json").get("steps", 5)) + int(J(f"{tag}_config3_bench_line_profiled.json").get("warmup", 2)) + 1
    # launches and device time of ONE steady-state step by the torch profiler (tools/train_launch_count.py); rocprofv3's totals include the first step, which
    # creates and packs every handle one by one
    census = open(os.path.join(P, f"{tag}_config3_launch_census.txt")).read()
    m = re.search(r"kernel launches: (\d+), device time ([0-9.]+) ms", census)
    launches3, dev3 = int(m.group(1)), float(m.group(2))
    cb = h["cpu_baseline"]
    v = {
        "HEAD_MS": f"{h['ms_per_step']:.1f}", "HEAD_MSPS": f"{h['value'] / 1e6:.1f}", "F32_MS": f"{h['fp32_mfma_engine']['ms_per_step']:.1f}",
        "S6_MS": f"{h['split_bf16x6_engine']['ms_per_step']:.1f}", "WERR": f"{h['waveform_max_abs_err']:.1e}", "LDERR": f"{h['flow_logdet_rel_err']:.1e}",
        "CPU_S": f"{cb['seconds']:.1f}", "CPU_K": f"{cb['value'] / 1e3:.1f}",
        "DOM_TF": f"{r['achieved']:.0f}", "DOM_MS": f"{r['avg_launch_ms']:.3f}", "DOM_RP": f"{float(rp['AverageNs']) / 1e6:.3f}", "DOM_FRAC": f"{r['frac']:.3f}",
        "DOM_BUSY": f"{mf[key]['mfma_pipe_util']:.2f}", "DOM_GHZ": f"{mf[key]['gfx_clock_ghz']:.2f}", "DOM_EXE": f"{mf[key]['mfma_tflops_executed']:.0f}",
        "K7_BUSY": f"{k7['mfma_pipe_util']:.2f}", "K7_GHZ": f"{k7['gfx_clock_ghz']:.2f}",
        "DOM_TRAF": f"{(r.get('traffic') or tr.get('hbm_bytes_per_launch_corrected', 0)) / 1e9:.2f}", "DOM_ALGB": f"{r['algorithmic_bytes_per_launch'] / 1e9:.2f}",
        "STEP_TF": f"{r['step']['achieved']:.0f}", "STEP_FRAC": f"{r['step']['frac']:.2f}",
        "STEP_GB": f"{(J(f'{tag}_pmc_traffic.json').get('pass_total', {}) or {}).get('hbm_bytes_corrected_per_step', 0) / 1e9:.0f}",
        "INST_K11": one("conv_ktap_kernel<11, 1, 2, 0, 4, 1, 8, 1>"), "INST_K7": one("conv_ktap_kernel<7, 1, 2, 0, 4, 1, 8, 1>"),
        "INST_K9": one("conv_ktap_kernel<9, 2, 2, 0, 4, 1, 8, 1>"), "INST_K3": one("conv_ktap_kernel<3, 1, 2, 0, 4, 1, 8, 1>"),
        "INST_RB": f"{rb[0]:.1f} / {rb[1]:.0f}–{rb[2]:.0f}", "RB_MS": f"{rb[0]:.1f}", "INST_TR": f"{trc[0]:.1f} / {trc[1]:.0f}–{trc[2]:.0f}", "TR_MS": f"{trc[0]:.1f}",
        "INST_ATT": one("relattn_bf16_kernel<3, 32, 3>"), "INST_GATE": one("conv_ktap_kernel<5, 0, 2, 0, 2, 2, 2, 2>"),
        "C2_MS": f"{c2['ms_per_step']:.1f}", "C2_MSPS": f"{c2['value'] / 1e6:.1f}", "C3_MS": f"{c3['ms_per_step']:.1f}",
        "C3_LAUNCH": f"{launches3:,d}".replace(",", " "), "C3_RP_LAUNCH": f"{calls3 / steps3:,.0f}".replace(",", " "), "C3_ATEN": f"{1 - vs3 / tot3:.2f}", "C3_DEV": f"{dev3:.1f}",
        "C5_MS": f"{c5['ms_per_step']:.1f}", "C5_MSPS": f"{c5['value'] / 1e6:.1f}", "C5ERR": f"{c5['oracle_check']['layer_rms_rel_err']:.1e}",
    }
    return v


def main():
    tag = sys.argv[1]
    v = values(tag)
    if "--write" not in sys.argv:
        for k, x in v.items():
            print(f"{k:10s} {x}")
    for doc in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, doc)
        text = open(path).read()
        names = sorted(set(re.findall(r"@([A-Z0-9_]+)@", text)))
        missing = [n for n in names if n not in v]
        if "--write" not in sys.argv:
            print(f"placeholders in {doc}:", names, "without a value:", missing)
            continue
        assert not missing, missing
        open(path, "w").write(re.sub(r"@([A-Z0-9_]+)@", lambda m: v[m.group(1)], text).replace("profiles/r05_b", f"profiles/{tag}"))
        print(f"{doc}: filled {len(names)} placeholders from profiles/{tag}_*")


if __name__ == "__main__":
    main()
