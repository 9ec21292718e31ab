#!/usr/bin/env python3
"""Prints DESIGN.md section 4.1 ("Measured") from the committed summaries of one profile tag (profiles/<tag>_*, <tag>_config5_*, <tag>_config3_*), so that the
numbers in the text are the numbers in the files.  Usage: python tools/design_numbers.py r04_f  [--write]   (--write replaces the section in DESIGN.md)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def J(name):
    return json.load(open(os.path.join(P, name)))


def main():
    tag = sys.argv[1]
    d = J(f"{tag}_bench_details.json")
    h, r = d["headline"], d["headline"]["roofline"]
    oc = d["other_configs"]
    oc = list(oc.values()) if isinstance(oc, dict) else oc
    c2, c3, c5 = (next(c for c in oc if c["config"].get("baseline_config") == n) for n in (2, 3, 5))
    l3 = json.loads(open(os.path.join(P, f"{tag}_config3_bench_line.json")).read())
    l5 = json.loads(open(os.path.join(P, f"{tag}_config5_bench_line.json")).read())
    d5 = J(f"{tag}_config5_bench_details.json")["headline"]["roofline"]
    mf = J(f"{tag}_pmc_mfma_busy.json")["kernels"]["conv_split_kernel<1,8,4,1,3>"]
    tr = J(f"{tag}_pmc_traffic.json")["kernels"]["conv_split_kernel<1,8,4,1,3>"]
    m5 = J(f"{tag}_config5_pmc_mfma_busy.json")["kernels"]["conv_split_kernel_bf16io<1,8,4,1,1,3>"]
    t5 = J(f"{tag}_config5_pmc_traffic.json")["kernels"]["conv_split_kernel_bf16io<1,8,4,1,1,3>"]
    with open(os.path.join(P, f"{tag}_bench_kernel_stats.csv"), newline="") as f:
        ks = {row["Name"]: row for row in csv.DictReader(f)}
    dom = ks["void vs::conv_split_kernel<1, 8, 4, 1, 3>(vs::ConvParams)"]
    with open(os.path.join(P, f"{tag}_config3_bench_kernel_stats.csv"), newline="") as f:
        k3 = list(csv.DictReader(f))
    steps3 = 8      # (rocprofv3 leg: --steps 5 --warmup 2 + the model's first step)
    inst = r["all_instances"]
    rb = [v for k, v in inst.items() if k.startswith("resblock_f16_kernel")]
    trc = [v for k, v in inst.items() if k.startswith("conv_split_tr_kernel")]
    gate = inst.get("conv_split_kernel<2, 2, 2, 2, 3>", {"ms_per_step": 0, "tflops": 0})
    att = inst.get("relattn_bf16_kernel<3, 32, 6>", {"ms_per_step": 0, "tflops": 0})
    c42 = inst.get("conv_split_kernel<1, 4, 2, 2, 3>", {"ms_per_step": 0, "tflops": 0})
    fp32 = h["fp32_mfma_engine"]
    cb = h["cpu_baseline"]
    step = r["step"]
    ms = h["ms_per_step"]
    alg_b = r["algorithmic_bytes_per_launch"]
    a5 = d5["all_instances"]["relattn_dma_kernel<8>"]
    out = f"""### 4.1 Measured (end of round 4, one MI355X; `profiles/{tag}_*`; boxes of the pool differ by 2–4 %)

Driver-visible stdout of `python bench.py` (`profiles/{tag}_bench_stdout.txt`, {os.path.getsize(os.path.join(P, tag + '_bench_stdout.txt')) / 1000:.1f} KB: one compact line per BASELINE config 2 / 3 / 5, then
the headline line, each < 4 KB; everything bulky in `bench_details.json` = `profiles/{tag}_bench_details.json`).  Headline — B=32 utterances ×
T_mel=1024, hop 256, fp32 tensors, the whole synthesis graph (text encoder + pitch predictor + frame prior + flow inverse + HiFi-GAN), 30 timed
steps after 10 warm-up: **{ms:.1f} ms/step, {h['value'] / 1e6:.1f} M audio samples/s, {h['value'] / 22050:,.0f}× real time**.replace(",", " ") (round 3: 77.5 ms; `profiles/r04_c_*`, the middle
of this round on another box: 77.7 ms; the changes of §4.4 are worth 2.7 ms of the difference in same-box A/B runs, the box the rest).  Same process,
same weights, same steps: exact-fp32 MFMA / F(2,3) engine {fp32['ms_per_step']:.1f} ms ({fp32['value'] / 1e6:.1f} M samples/s = {22.02 / fp32['ms_per_step'] * 1e3:.0f} TFLOP/s = {22.02 / fp32['ms_per_step'] * 1e3 / 157.3:.2f} of the fp32-MFMA
peak: the strictly-same-precision number), split-bf16 ×6 engine {h['split_bf16x6_engine']['ms_per_step']:.1f} ms.  Item 0 of the timed batch is within **{h['waveform_max_abs_err']:.1e}** of the fp32
CPU oracle's synthesis of that item (`waveform_max_abs_err`, bar 1e-4); flow log-det of the affine coupling {h['flow_logdet_rel_err']:.1e} relative (bar 1e-4),
`mean_only` log-det exactly 0.  `cpu_baseline`: the C/OpenMP fp32 port, {cb['cores']} threads, ONE item of the 32 ({cb['samples']:,} samples in {cb['seconds']:.1f} s):
{cb['value'] / 1e3:.0f} k samples/s; with the convolutions on stock PyTorch CPU kernels {h['cpu_baseline_torch']['value'] / 1e3:.0f} k.

`roofline` of the headline line: dominant instance `conv_split_kernel<1, 8, 4, 1, 3>` ({100 * r['share_of_step']:.0f} % of the step, {r['launches_per_step']:.0f} launches per step) — `achieved`
**{r['achieved']:.0f} TFLOP/s of ALGORITHMIC work** ({r['algorithmic_gflop_per_launch']:.1f} GFLOP per launch ÷ {r['avg_launch_ms']:.3f} ms by HIP events; rocprofv3 `profiles/{tag}_bench_kernel_stats.csv`:
{float(dom['AverageNs']) / 1e6:.3f} ms over {dom['Calls']} launches), `peak` 833 = 2500 / 3, **`frac` {r['frac']:.3f}**; the pipe executes {mf['mfma_tflops_executed']:,.0f} TFLOP/s (`SQ_INSTS_VALU_MFMA_MOPS_F16` × 512 / time),
{100 * mf['mfma_pipe_util']:.0f} % busy at {mf['gfx_clock_ghz']:.2f} GHz; HBM traffic by the counters {tr['hbm_bytes_per_launch_corrected'] / 1e9:.3f} GB per launch against {alg_b / 1e9:.3f} GB algorithmic ({tr['hbm_bytes_per_launch_corrected'] / alg_b:.2f}×: tile halos).  Whole step:
{step['algorithmic_tflop_per_step']:.1f} TFLOP ÷ {ms:.1f} ms = {step['achieved']:.0f} TFLOP/s = {step['frac']:.2f} of 833; {step['algorithmic_gb_per_step']:.1f} GB ÷ {ms:.1f} ms = {step['hbm_gbps_algorithmic'] / 1e3:.2f} TB/s = {step['hbm_frac_of_8tbps']:.2f} of 8 TB/s.  Per instance (ms per step / TFLOP/s):
`resblock_f16_kernel` {sum(v['ms_per_step'] for v in rb):.1f} / {min(v['tflops'] for v in rb):.0f}–{max(v['tflops'] for v in rb):.0f} ({len(rb)} instances), `conv_split_kernel<1, 4, 2, 2, 3>` {c42['ms_per_step']:.1f} / {c42['tflops']:.0f} (the 192-row transformer convs: short
launches), transposed convs {sum(v['ms_per_step'] for v in trc):.1f}, `relattn_bf16_kernel<3, 32, 6>` {att['ms_per_step']:.1f} / {att['tflops']:.0f}, the WaveNet gates {gate['ms_per_step']:.1f} / {gate['tflops']:.0f}.

Other configurations (same run, 10 timed steps after 3 warm-up; own rocprofv3 kernel stats and the three PMC passes for configs 3 and 5:
`profiles/{tag}_config{{3,5}}_*`, each PMC summary tagged with the workload it was recorded on — a line only cites bytes of its own launch shapes):
* config 2 (flow inverse + HiFi-GAN, B=8, T_mel=512): **{c2['ms_per_step']:.1f} ms, {c2['value'] / 1e6:.1f} M samples/s** (`frac` {c2['roofline']['frac']:.2f}); its first two items within 1e-7 of the CPU
  port, which runs them at {c2['cpu_baseline']['value'] / 1e3:.0f} k samples/s on 16 threads.
* config 3 (full GAN training step, B=16, T_mel=512, the reference's dropout 0.1): **{c3['ms_per_step']:.1f} ms/step** inside the default run, {l3['ms_per_step']:.1f} run alone
  (round 3: 110.4–111.1; 4.58 TFLOP of conv / attention work per step = {c3['roofline']['achieved']:.0f} TFLOP/s = {c3['roofline']['frac']:.2f} of 833; {sum(int(x['Calls']) for x in k3) / steps3:,.0f} launches).  Round 4: one
  pack launch pair for a conv and its grad-input handle (−2.7 ms in an A/B), weight-gradient loads under the MFMAs, bias gradients in one fast launch,
  the T = 1 weight gradients on the library GEMM, the epilogue waits and scalar trims of §4.4 (−4.8 ms together), the grouped-conv gradient kernels
  unrolled (−1 ms), torch's fused AdamW (−2.3 ms; same-box A/Bs).  Still bound by ~5 600 launches of small kernels (26 ms in 446 launches of
  `conv_split_kernel<1, 1, 1, 4, 3>` whose 32 × 32 wave tiles give a step three MFMAs against ≈ 775 cycles of instruction stream; 7 ms of weight
  packs and weight-norm kernels; ~20 ms of PyTorch elementwise / reduction kernels): VERDICT r3's ≤ 85 ms is NOT met.
* config 5 (T_mel 4096, hidden 512, bf16, bf16-resident generator tensors, B=8): **{c5['ms_per_step']:.1f} ms, {c5['value'] / 1e6:.1f} M samples/s** ({l5['ms_per_step']:.1f} run alone; 64.1–64.7 in
  round 3): the attention core on `relattn_dma_kernel<8>` (§4.3: 10.4 → {a5['ms_per_step']:.1f} ms per step), the staging / prefetch waits and scalar trims of §4.4.
  `conv_split_kernel_bf16io<1, 8, 4, 1, 1, 3>` {d5['achieved']:.0f} TFLOP/s = {d5['achieved'] / 2500:.2f} of the bf16 peak, counters: {t5['hbm_bytes_per_launch_corrected'] / 1e9:.3f} GB per launch against {d5['algorithmic_bytes_per_launch'] / 1e9:.3f}
  algorithmic = {d5['hbm_frac_of_8tbps']:.2f} of 8 TB/s while it runs, pipe {100 * m5['mfma_pipe_util']:.0f} % busy at {m5['gfx_clock_ghz']:.2f} GHz (not power-limited: a (chunk, tap) step of the one-plane
  arithmetic has 8 MFMAs per wave and the per-step instruction stream sets the pace, §4.4); whole step {d5['step']['achieved']:.0f} TFLOP/s = {d5['step']['frac_of_bf16_peak']:.2f} of 2500,
  {d5['step']['hbm_gbps_algorithmic'] / 1e3:.2f} TB/s = {d5['step']['hbm_frac_of_8tbps']:.2f} of 8 TB/s.  VERDICT r3's ≤ 58 ms: {'met' if min(c5['ms_per_step'], l5['ms_per_step']) <= 58.0 else 'missed by %.1f ms' % (min(c5['ms_per_step'], l5['ms_per_step']) - 58.0)}.

`bench.py` PCIe note: inputs are resident in HBM before the timed region (tokens, alignment, noise: 25 MB per step); a step's output is
33.5 MB of waveform.

"""
    out = out.replace("× real time**.replace(\",\", \" \")", "× real time**")
    if "--write" in sys.argv:
        p = os.path.join(ROOT, "DESIGN.md")
        s = open(p).read()
        i0, i1 = s.index("### 4.1 Measured"), s.index("### 4.2 The dominant conv")
        open(p, "w").write(s[:i0] + out + s[i1:])
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
