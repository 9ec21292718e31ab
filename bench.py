#!/usr/bin/env python3
"""bench.py -- VISinger synthesis throughput on MI355X (the BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the synthesis hot path (VISinger.forward(infer=True) body: text encoder -> pitch predictor
-> frame-prior transformer -> reparameterised sample -> flow inverse -> HiFi-GAN generator) over one batch of
B=32 synthetic utterances of T_mel=1024 frames, hop 256 (22.05 kHz): 8 388 608 audio samples per step per GPU.
Inputs (tokens, mel2ph, noise) are resident in HBM before the timed region; weights are random-init of the
reference architecture (no checkpoints offline).  Multi-GPU = one process per GPU, utterances sharded by the
reference's strided rule (batch[rank::world], tasks/base.py:130-133), no data-path collective -> weak scaling.

`--gpus N` without a torchrun environment starts its own N ranks (fresh child processes, one per GPU, started before this
process touches the GPU -- the reference spawns its ranks itself as well, utils/commons/trainer.py:117-138); under
torch.distributed.run it takes RANK / LOCAL_RANK / WORLD_SIZE from the environment and refuses a WORLD_SIZE != N.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is for the dominant kernel (the implicit-GEMM conv
instance with the largest share of the step: a split-bf16 MFMA instance by default), from HIP events recorded around
each of its launches inside the timed steps: `achieved` = ALGORITHMIC FLOPs (the conv's own 2*MAC) / time, `frac` =
achieved / the stated peak; what the matrix pipe executes on top of that (6 bf16 cross products per fp32 product) is under
`frac_executed` / `mfma_executed`.  `cpu_baseline` is the CPU oracle ("port") timed on a bounded sample.

Other BASELINE.json configurations: --config 2 (flow inverse + HiFi-GAN decode, B=8 T_mel=512 fp32), --config 3 (full GAN
training step, B=16), --config 4 (= the default workload, meant for --gpus 8), --config 5 (T_mel=4096 hidden=512 bf16).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HOP = 256
SR = 22050
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16, dense bf16 (no sparsity)
MATH = {"split3": 3, "split6": 6, "f32": 0, "bf16": 1}
DTYPE = {"split3": "f32 (operands as 2 f16 planes under a power-of-two scale per staged tile = 22 significant bits, 3 cross products on the "
                   "f16 MFMA, fp32 accumulate; error vs fp64 below the fp32 MFMA's: tests/test_conv_split_gpu.py)",
         "split6": "f32 (operands split exactly into 3 bf16 planes, 6 cross products on the bf16 MFMA, fp32 accumulate)",
         "f32": "f32", "bf16": "bf16 operands, f32 accumulate (f32 tensors in HBM)"}


def synthetic_batch(B, T, Tph, ph_dict, seed, device, ragged=False, hidden=192):
    """SURVEY.md 8d synthetic inputs: mel2ph = repeat_interleave(arange(1, Tph+1), T/Tph); tokens uniform."""
    g = torch.Generator().manual_seed(seed)
    text = torch.randint(4, ph_dict, (B, Tph), generator=g)
    pitch = torch.randint(1, 117, (B, Tph), generator=g)
    dur = torch.randint(4, 131, (B, Tph), generator=g)
    mel2ph = torch.repeat_interleave(torch.arange(1, Tph + 1), T // Tph)[None].repeat(B, 1)
    if ragged:
        lens = torch.randint(T // 2, T + 1, (B,), generator=g)
        mel2ph = mel2ph * (torch.arange(T)[None] < lens[:, None])
    noise = torch.randn(B, hidden, T, generator=g)
    spk = torch.zeros(B, dtype=torch.long)
    return [t.to(device) for t in (text, pitch, dur, mel2ph, spk, noise)]


def build_model(seed=1234, hop=256, hidden=192):
    from visinger_amd.models.visinger import REFERENCE_HPARAMS, VISinger, hop256_hparams
    torch.manual_seed(seed)
    hp = hop256_hparams() if hop == 256 else dict(REFERENCE_HPARAMS)      # 300: the reference's own 24 kHz configuration
    if hidden != 192:      # BASELINE config 5 width: hidden 512 (2 heads -> 256 channels per head), FFN 2048
        hp.update(hidden_size=hidden, ffn_filter_channels=4 * hidden)
    model = VISinger(64, 117, 131, hp)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():    # non-trivial flow (post convs are zero-initialised in the reference)
        for f in range(4):
            post = model.flow.flows[2 * f].post
            post.weight.copy_(0.05 * torch.randn(post.weight.shape, generator=g))
            post.bias.copy_(0.05 * torch.randn(post.bias.shape, generator=g))
    return model.eval(), hp


def usable_cores():
    """host cores this process may actually run on: affinity mask, capped by the cgroup CPU quota when there is one
    (os.cpu_count() reports the machine's cores even inside a quota-limited container)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]) + 0.5)))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f2:
                        n = min(n, max(1, int(q / int(f2.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def library_source_hash():
    """first 16 hex digits of the hash of the sources libvisinger_hip.so was built from (vs_source_hash; every committed profile summary carries it)"""
    from visinger_amd import _lib as L
    import ctypes
    fn = L.lib().vs_source_hash
    fn.restype = ctypes.c_char_p
    return fn().decode()[:16]


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(model, hp, batch, backend="c", wav_dev=None, f0_dev=None, T_cap=1024, runs=3):
    """The CPU oracle (fp32 'port' of the reference arithmetic, oracle/) timed on a bounded sample of the SAME workload and the
    SAME graph as `value` (same weights, pitch predictor on): ITEM 0 of the timed batch (B=1 of the 32, T_mel as timed, capped at
    T_cap).  backend "c": the C/OpenMP restatement; "torch": the same composition with the convolutions (98 % of the CPU time) on stock
    PyTorch CPU kernels with torch.set_num_threads(all cores) -- the reference's own CPU execution engine.
    With wav_dev (the device's waveforms of the timed batch) the oracle's item is also the checker of the timed run:
    `waveform_max_abs_err` = max |device - fp32 oracle| over item 0 (frames whose voicing logit the oracle puts within 1e-3 of the
    threshold take the device's decision: `voicing_hinted_frames`)."""
    from oracle import visinger_oracle as orc
    orc.build()
    orc.CONV_BACKEND = backend
    cores = usable_cores()
    orc.set_threads(cores)
    if backend == "torch":
        cores = min(cores, 32)      # oneDNN on a B=1 conv does not scale past a few dozen threads (and thrashes at 256)
        torch.set_num_threads(cores)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    text, pitch, dur, mel2ph, spk, noise = [t[:1].cpu().numpy() for t in batch]
    T = min(int(mel2ph.shape[1]), T_cap)
    assert T == mel2ph.shape[1], "the baseline runs whole items of the timed batch"
    hint = None if f0_dev is None else (f0_dev[:1, :, 1] <= 0).cpu().numpy()
    times = []
    for _ in range(max(1, runs)):           # the median of >= 3 runs
        t0 = time.perf_counter()
        out = orc.visinger_infer(sd, hp, text, pitch, dur, mel2ph, spk, noise, dtype=np.float32, return_all=True, voiced_hint=hint,
                                 hint_tol=1e-3 if hint is not None else 0.0)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    orc.CONV_BACKEND = "c"
    n = out["wav_out"].size
    what = "C+numpy port (OpenMP" if backend == "c" else "numpy composition with torch-CPU oneDNN convolutions (torch threads"
    res = {"value": n / dt, "unit": "audio samples/s", "cores": cores, "kind": "port", "items": 1, "of_items": int(batch[0].shape[0]),
           "seconds": dt, "samples": int(n), "runs": len(times), "run_seconds": [round(t, 3) for t in times], "cpu": cpu_model_name(),
           "sample": f"oracle/ fp32 {what}, {cores} threads), median of {len(times)} runs of item 0 of the timed batch: B=1 T_mel={T} hop={HOP}: {n} samples in {dt:.2f}s; the "
                     f"graph `value` times (text-encoder + pitch-predictor + frame-prior + flow-inverse + generator, use_pitch_embed="
                     f"{bool(hp.get('use_pitch_embed'))})"}
    if wav_dev is not None:
        res["waveform_max_abs_err"] = float(np.abs(wav_dev[0].double().cpu().numpy() - out["wav_out"][0]).max())
        res["waveform_tolerance"] = 1e-4
        if out["f0_pred"] is not None:
            res["voicing_hinted_frames"] = int((np.abs(out["f0_pred"][:, :, 1]) <= 1e-3).sum())
    return res


def oracle_check_config5(model, hp, batch, f0_dev):
    """BASELINE configs[4]: an oracle error figure in the config-5 line.  A whole item at T_mel = 4096 / hidden 512 costs the CPU port minutes,
    so the check is bounded to ONE encoder layer with the weights the timed run used: layer 0 of the pitch predictor's RelativeEncoder (2 heads of 256 channels,
    FFN 2048, k = 9; reference modules/rel_transformer.py:148-179, 290-345) on B = 2 x T = 4096 seeded frames in the plain-bf16 arithmetic -- the dispatch of
    the timed run (relattn_dma_kernel<8>, the conv_ktap bf16 instances with masked inputs) -- against oracle.rel_encoder (fp32) of item 0: rms relative error,
    stated bf16 bound 3e-2.  (tests/test_production_dispatch_gpu.py holds two layers and the whole model to the oracle.)"""
    from oracle import visinger_oracle as orc
    from visinger_amd import _lib as L
    from visinger_amd.modules.hipconv import set_conv_math
    from visinger_amd.modules.rel_transformer import RelativeEncoder
    from visinger_amd.ops import PROFILER
    orc.build()
    orc.set_threads(usable_cores())
    src = model.pitch_predictor.pitch_predictor
    H, F, nh, ks = src.hidden_channels, src.filter_channels, src.n_heads, src.kernel_size
    keep = ("attn_layers.0.", "norm_layers_1.0.", "ffn_layers.0.", "norm_layers_2.0.")
    sd1 = {k: v.detach().clone() for k, v in src.state_dict().items() if k.startswith(keep)}
    enc = RelativeEncoder(H, F, nh, 1, kernel_size=ks, p_dropout=0.0)
    enc.load_state_dict(sd1, strict=True)
    enc = enc.to(f0_dev.device).eval()
    set_conv_math(enc, L.MATH_BF16)
    g = torch.Generator().manual_seed(4096)
    T = int(batch[3].shape[1])
    x = torch.randn(2, H, T, generator=g)
    mask = torch.ones(2, 1, T)
    was = PROFILER.enabled
    PROFILER.enabled = False
    with torch.no_grad():
        y = enc(x.to(f0_dev.device), mask.to(f0_dev.device))
    torch.cuda.synchronize()
    PROFILER.enabled = was
    kernel = L.lib().vs_last_kernel_name().decode()
    t0 = time.perf_counter()
    ref = orc.rel_encoder({k: v.cpu().numpy() for k, v in sd1.items()}, x[:1].numpy(), mask[:1].numpy(), None, n_heads=nh, n_layers=1, kernel_size=ks, dtype=np.float32)
    sec = time.perf_counter() - t0
    d = y[:1].double().cpu().numpy() - ref
    rms = float(np.sqrt((np.asarray(ref, np.float64) ** 2).mean()))
    return {"layer_rms_rel_err": float(np.sqrt((d ** 2).mean())) / rms, "layer_max_abs_err": float(np.abs(d).max()), "tolerance_rms_rel": 3e-2, "seconds": sec,
            "what": f"pitch-predictor encoder layer 0 (hidden {H}, T {T}, plain bf16: the timed run's attention / FFN dispatch, last launch {kernel}) vs oracle.rel_encoder (fp32), item 0"}


def cpu_baseline_config2(model, hp, batch, items=8):
    """BASELINE.md 4 asks for the config-2 shape next to the GPU number: flow inverse + HiFi-GAN decode at T_mel=512, all 8 utterances once
    (about 20 s of CPU), on the fp32 C/OpenMP port."""
    from oracle import visinger_oracle as orc
    orc.build()
    cores = usable_cores()
    orc.set_threads(cores)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    z_p, fmask, spk = [t[:items].cpu().numpy() for t in batch]
    g = sd["spk_id_proj.weight"][spk][:, :, None]
    t0 = time.perf_counter()
    z_q = orc.flow_block(orc._sub(sd, "flow"), z_p, fmask, g, reverse=True, channels=z_p.shape[1], hidden_channels=z_p.shape[1], kernel_size=5,
                         dilation_rate=1, n_layers=4, dtype=np.float32) * fmask
    wav = orc.generator(orc._sub(sd, "decoder"), z_q, g, resblock=hp["dec_blocks"], resblock_kernel_sizes=hp["dec_kernel_size"],
                        resblock_dilation_sizes=hp["dec_dilation_sizes"], upsample_rates=hp["upsample_rates"],
                        upsample_kernel_sizes=hp["upsample_kernel_sizes"], dtype=np.float32)[:, 0]
    dt = time.perf_counter() - t0
    return {"value": wav.size / dt, "unit": "audio samples/s", "cores": cores, "kind": "port", "items": items, "of_items": int(batch[0].shape[0]),
            "seconds": dt, "samples": int(wav.size), "runs": 1, "cpu": cpu_model_name(),
            "sample": f"oracle/ fp32 C+numpy port (OpenMP, {cores} threads), {items} of the 8 items of BASELINE config 2 (flow inverse + "
                      f"HiFi-GAN decode, T_mel={z_p.shape[2]}): {wav.size} samples in {dt:.2f}s"}, wav


def flow_logdet_check(model, dev, B=2, T=256, seed=1234):
    """The metric's second half: flow log-det relative error of the HIP coupling layer against the fp64 oracle (part of
    the cpu_baseline leg: the only place bench.py may call oracle/).  The model's own flow is mean_only=True
    (models/visinger.py:66 -> flow.py:26-27), whose log-det must be EXACTLY 0; the affine form (mean_only=False,
    flow.py:76-80) is checked on a coupling layer of the same size (192 ch, 4 WaveNet layers, gin 256)."""
    from oracle import visinger_oracle as orc
    from visinger_amd.modules.visinger.flow import ResidualCouplingLayer
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 192, T, generator=g)
    lens = torch.tensor([T, T - T // 3])[:B]
    mask = (torch.arange(T)[None] < lens[:, None]).float().unsqueeze(1)
    spk = torch.randn(B, 256, 1, generator=g)
    res = {}
    with torch.no_grad():
        lay0 = model.flow.flows[0]
        _, ld0 = lay0(x.to(dev), mask.to(dev), g=spk.to(dev))
        res["mean_only_true_logdet_is_exact_zero"] = bool((ld0 == 0).all())
        torch.manual_seed(seed)
        lay = ResidualCouplingLayer(192, 192, 5, 1, 4, gin_channels=256, mean_only=False)
        lay.post.weight.copy_(0.05 * torch.randn(lay.post.weight.shape, generator=g))
        lay.post.bias.copy_(0.05 * torch.randn(lay.post.bias.shape, generator=g))
        sd = {k: v.detach().numpy().copy() for k, v in lay.state_dict().items()}
        lay = lay.to(dev).eval()
        y, ld = lay(x.to(dev), mask.to(dev), g=spk.to(dev))
    y_ref, ld_ref = orc.coupling_layer(sd, x.numpy(), mask.numpy(), spk.numpy(), channels=192, hidden_channels=192,
                                       kernel_size=5, dilation_rate=1, n_layers=4, mean_only=False, dtype=np.float64)
    res["rel_err"] = float(np.max(np.abs(ld.cpu().numpy() - ld_ref) / np.abs(ld_ref)))
    res["x_max_abs_err"] = float(np.max(np.abs(y.cpu().numpy() - y_ref)))
    res["tolerance"] = 1e-4
    res["sample"] = f"ResidualCouplingLayer(192,192,5,1,4,gin=256,mean_only=False) forward, B={B} T={T} ragged mask, vs fp64 oracle"
    return res


PROFILE_TAGS = ("r06_c", "r06_b", "r06_a", "r05_c", "r05_b", "r05_a", "r04_f", "r04_e", "r04_d", "r04_c", "r04_b", "r04_a", "r03_f", "r03_e", "r03_d", "r03_c", "r03_b", "r03_a", "r02_e", "r02_d", "r02_c", "r02_b", "r02_a", "r01_f", "r01_e",
                "r01_c")      # newest first: profiles/<tag>[_<suffix>]_pmc_*.json
HEADLINE_WORKLOAD = "B32_T1024_h192_hop256_f32"      # what a profiles/*_pmc_*.json without a "workload" field was recorded on (rounds 1-3)


def workload_key(config, B, T, hidden, hop, storage):
    """Names the launch shapes of a run: PMC figures are per launch of a kernel INSTANCE, and an instance's bytes depend on the tensor
    shapes it was launched on, so a recorded figure is only attached to a line of the same workload."""
    return ("c2_" if config == 2 else "c3_" if config == 3 else "") + f"B{B}_T{T}_h{hidden}_hop{hop}_{storage}"


def _norm(kernel):
    return kernel.replace(" ", "")


def _pmc_files(kind, workload):
    import glob
    for tag in PROFILE_TAGS:
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"{tag}*_pmc_{kind}.json"))):
            try:
                with open(path) as f:
                    d = json.load(f)
            except (OSError, ValueError):
                continue
            if d.get("workload", HEADLINE_WORKLOAD) == workload:
                yield os.path.relpath(path, ROOT), d


def pmc_traffic(kernel, workload=HEADLINE_WORKLOAD):
    """HBM bytes per launch of `kernel` from the committed PMC passes of the SAME workload (FETCH_SIZE / WRITE_SIZE cannot be collected
    from inside the timed run: they need rocprofv3 and one pass per counter); None when no pass of this workload was recorded."""
    for path, d in _pmc_files("traffic", workload):
        ks = {_norm(k): v for k, v in d.get("kernels", {}).items()}
        if _norm(kernel) in ks:
            return {"bytes_per_launch": ks[_norm(kernel)]["hbm_bytes_per_launch_corrected"],
                    "source": f"recorded: {path} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"}
    return None


def pmc_step_traffic(workload):
    """HBM bytes per STEP over every kernel of the recorded PMC passes of `workload` (whole-step lines: BASELINE config 3)"""
    for path, d in _pmc_files("traffic", workload):
        t = d.get("pass_total") or {}
        if t.get("hbm_bytes_corrected_per_step"):
            return {"bytes_per_step": t["hbm_bytes_corrected_per_step"],
                    "source": f"recorded: {path} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE over every kernel of the pass / its {t['steps_in_pass']} steps)"}
    return None


def pmc_mfma_executed(kernel, workload=HEADLINE_WORKLOAD):
    """FLOP/s the matrix pipe really executed in `kernel` (SQ_INSTS_VALU_MFMA_MOPS_* x 512 / kernel time, recorded PMC pass of its
    own: tools/pmc_mfma_summarize.py) on the same workload; None when not recorded."""
    for path, d in _pmc_files("mfma_busy", workload):
        k = {_norm(n): v for n, v in d.get("kernels", {}).items()}.get(_norm(kernel))
        if not k or "mfma_tflops_executed" not in k:
            continue
        out = {"tflops": k["mfma_tflops_executed"], "pipe_busy": k["mfma_pipe_util"],
               "source": f"recorded: {path} (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_*, its own pass)"}
        if "gfx_clock_ghz" in k:
            out["gfx_clock_ghz"] = k["gfx_clock_ghz"]
        return out
    return None


CONFIGS = {     # BASELINE.json `configs` (1-based; config 1 is the CPU plumbing case: tests/, not a bench line)
    2: dict(batch=8, frames=512, what="flow inverse + HiFi-GAN decode"),
    3: dict(batch=16, frames=512, what="full GAN training step"),
    4: dict(batch=32, frames=1024, what="batched inference, 32 utterances per GPU (B=256 over 8 GPUs)"),
    5: dict(batch=8, frames=4096, hidden=512, math="bf16", what="long-form stress"),
}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """Start n ranks of this script (one per GPU) as fresh child processes and wait for them.  Nothing in THIS process has
    touched the GPU (never re-exec or fork a process that initialised HIP); rank 0's JSON line passes through on stdout."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, VS_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # Poll ALL children: a rank that dies leaves its siblings in dist.barrier() forever.  On the first non-zero exit the others are
    # terminated (then killed) and the launcher exits non-zero; only these fresh children are ever signalled, by their exact PIDs.
    rc, live = 0, list(procs)
    while live and rc == 0:
        for pr in list(live):
            r = pr.poll()
            if r is not None:
                live.remove(pr)
                if r != 0:
                    rc = abs(r) or 1
                    print(f"[bench] rank process {pr.pid} exited with {r}: stopping the other {len(live)} rank(s)", file=sys.stderr, flush=True)
                    break
        if live and rc == 0:
            time.sleep(0.2)
    for pr in live:
        pr.terminate()
    deadline = time.time() + 10.0
    for pr in live:
        try:
            pr.wait(timeout=max(0.1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            pr.kill()
            pr.wait()
    return rc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=0, choices=(0, 2, 3, 4, 5),
                    help="a BASELINE.json configuration (sets batch / frames / hidden / math); 0 = the headline workload")
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default 32)")
    ap.add_argument("--frames", type=int, default=None, help="T_mel (default 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the timed batches rotate over (visinger_amd.synth.StreamRotation; 1 = every batch on one stream, the rounds 1-5 measurement)")
    ap.add_argument("--ragged", action="store_true", help="SURVEY 8d ragged variant: item lengths ~ U{T/2..T}, tails masked (mel2ph = 0)")
    ap.add_argument("--hidden", type=int, default=None, help="hidden_size (512 = the BASELINE config-5 width)")
    ap.add_argument("--hop", type=int, default=256, choices=(256, 300),
                    help="256: the BASELINE.json benchmark variant (default); 300: the reference's own generator configuration")
    ap.add_argument("--math", default=None, choices=tuple(MATH),
                    help="arithmetic of the conv engine (include/visinger_hip.h vs_conv_math): split3 = fp32-class split-f16 (default, the "
                         "headline), split6 = fp32-class split-bf16 (the default of rounds 1-2), f32 = fp32 MFMA / Winograd F(2,3), bf16 = bf16 "
                         "operands (BASELINE config 5)")
    ap.add_argument("--dropout", type=float, default=0.1, help="config 3: p_dropout of the transformers (reference config: 0.1)")
    ap.add_argument("--storage", default=None, choices=("f32", "bf16"),
                    help="element type of the generator's activations in HBM (bf16 only with --math bf16; default: bf16 for --config 5)")
    ap.add_argument("--no-other-configs", action="store_true", help="default run: skip the reduced lines of BASELINE configs 2 / 3 / 5")
    ap.add_argument("--quick", action="store_true", help="skip the torch-CPU baseline and the split-bf16 x6 engine legs (the fp32 MFMA engine stays)")
    ap.add_argument("--print-workload-key", action="store_true", help="print the key PMC summaries of this command line are filed under, and exit")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous only (no GPU work): every rank joins the process group, barrier, max-over-ranks, one JSON line")
    args = ap.parse_args()
    preset = CONFIGS.get(args.config, {})
    args.batch = args.batch if args.batch is not None else preset.get("batch", 32)
    args.frames = args.frames if args.frames is not None else preset.get("frames", 1024)
    args.hidden = args.hidden if args.hidden is not None else preset.get("hidden", 192)
    if args.math is None:
        args.math = "bf16" if preset.get("math") == "bf16" else "split3"
    if args.storage is None:
        args.storage = "bf16" if (args.config == 5 and args.math == "bf16") else "f32"
    if args.storage == "bf16" and args.math != "bf16":
        raise SystemExit("--storage bf16 needs --math bf16")
    return args


def init_ranks(args):
    """-> (rank, local_rank, world, dist-or-None, backend); exits non-zero when the world is not --gpus."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different world size", file=sys.stderr)
        raise SystemExit(2)
    backend = os.environ.get("VS_BENCH_BACKEND", "nccl")        # "gloo": rehearsal of the N > 1 path without N GPUs
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist, backend


def dry_run(args):
    """The N-rank launch path without GPU work: rendezvous, barrier, max-over-ranks of a per-rank value, census of the ranks."""
    rank, local_rank, world, dist, backend = init_ranks(args)
    from visinger_amd.dp import max_over_ranks
    if os.environ.get("VS_BENCH_DIE_RANK") == str(rank):      # test hook: this rank dies before the first barrier
        os._exit(7)
    seen, pids = [rank], [os.getpid()]
    dev = torch.device("cuda", local_rank) if (backend == "nccl" and world > 1) else None
    if dist is not None:
        dist.barrier()
        t = torch.zeros(2, world, dtype=torch.int64, device=dev)
        t[0, rank], t[1, rank] = rank + 1, os.getpid()
        dist.all_reduce(t)
        seen, pids = [int(v) - 1 for v in t[0].tolist()], [int(v) for v in t[1].tolist()]
    slow = max_over_ranks(float(rank + 1), device=dev)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": seen, "max_over_ranks": slow, "backend": backend if world > 1 else None,
                          "processes": len(set(pids)), "launcher_pid_is_a_rank": os.getppid() in pids}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def percentile_stats(ms):
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {"median_ms": float(np.median(a)), "min_ms": float(a[0]), "max_ms": float(a[-1])}


def roofline_from_profile(prof, dt, steps, workload=None):
    """`roofline` object of a bench line from the per-launch HIP-event records of the timed steps (ops.PROFILER.summary()): the dominant
    kernel instance (largest share of the step), its algorithmic FLOP/s against the roof of its arithmetic, and the whole step."""
    allp = prof
    prof = {k: v for k, v in prof.items() if v["ms"] > 0}        # (sites recorded without events -- PROFILER.note() -- carry no time)
    name, d = max(prof.items(), key=lambda kv: kv[1]["ms"])
    traffic = pmc_traffic(name, workload) if workload else None
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12            # ALGORITHMIC: the convs' own 2*MAC / measured kernel time
    step_flops = sum(v["flops"] for v in allp.values()) / steps
    step_bytes = sum(v["bytes"] for v in allp.values()) / steps
    step_tflops = step_flops / (dt / steps) / 1e12
    kern_ms = sum(v["ms"] for v in prof.values())
    if name.startswith(("conv_split_kernel", "conv_ktap_kernel", "respair_split_kernel", "resblock_f16_kernel", "resblock_bf16_kernel", "relattn_bf16_kernel", "relattn_dma_kernel")):
        targs = [a.strip() for a in name[name.index("<") + 1:].rstrip(">").split(",")]
        ints = [int(a) for a in targs if a.isdigit()]
        if name.startswith("conv_split_kernel"):
            terms = ints[4]                                  # cross products per fp32 product (5th template argument)
        elif name.startswith("conv_ktap_kernel"):
            terms = 3 if ints[2] == 2 else 1                 # conv_ktap_kernel<taps, input transform, planes, tensors, tile...>: two f16 planes = three cross products
        elif name.startswith("resblock_f16_kernel"):
            terms = 3
        elif name.startswith(("relattn_dma_kernel", "resblock_bf16_kernel")):
            terms = 1                                        # plain bf16 operands (BASELINE configs[4])
        else:
            terms = ints[-1] if ints[-1] in (1, 3, 6) else ints[-2]
        peak = BF16_MFMA_PEAK_TFLOPS / terms
        peak_name = (f"dense bf16 / f16 MFMA peak {BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {terms} cross products per fp32 product = the roof of "
                     f"this arithmetic for fp32-class results" if terms > 1 else "dense bf16 MFMA peak")
        roof = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_name": peak_name, "frac_vs_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                "executed_tflops": achieved * terms, "frac_executed": achieved * terms / BF16_MFMA_PEAK_TFLOPS,
                "frac_executed_of_measured_mfma_ceiling": achieved * terms / 1570.0,
                "note": f"achieved = ALGORITHMIC FLOPs (the convs' own 2*MAC, SURVEY 8d) / HIP-event time of the kernel's launches; the "
                        f"matrix pipe executes {terms} bf16 MFMA FLOPs per algorithmic FLOP (executed_tflops, frac_executed vs the 2500 "
                        "dense peak; mfma_executed = the same from rocprofv3's MFMA counters).  Under this load the chip clocks at "
                        "1.6-1.85 GHz: a bare loop of this MFMA sustains 1.57 PFLOP/s at 1.67 GHz on this box "
                        "(tools/ubench/mfma_bf16_rate.hip) = the denominator of frac_executed_of_measured_mfma_ceiling"}
    else:
        roof = {"bound": "mfma", "kernel": name, "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "peak_name": "fp32 MFMA peak (v_mfma_f32_32x32x2_f32)",
                "note": "achieved = ALGORITHMIC (direct-form) FLOPs / time; the F(2,3) minimal-filtering instances execute "
                        "4/6 (k=3, 9), 10/14 (k=7), 15/22 (k=11) of them on the matrix pipe, so achieved can exceed the MFMA "
                        "peak: mfma_executed is what the pipe really did"}
    step_peak = roof["peak"]
    roof.update({
        "traffic": (traffic or {}).get("bytes_per_launch"),
        "traffic_source": (traffic or {}).get("source"),
        "traffic_over_algorithmic": (traffic["bytes_per_launch"] / (d["bytes"] / d["launches"])) if traffic else None,
        "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
        "hbm_gbps_algorithmic": d["bytes"] / (d["ms"] * 1e-3) / 1e9, "hbm_frac_of_8tbps": d["bytes"] / (d["ms"] * 1e-3) / 8e12,
        "mfma_executed": pmc_mfma_executed(name, workload) if workload else None,
        "launches_per_step": d["launches"] / steps,
        "avg_launch_ms": d["ms"] / d["launches"],
        "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
        "share_of_step": d["ms"] / (dt * 1e3),
        "step": {"algorithmic_tflop_per_step": step_flops / 1e12, "achieved": step_tflops, "peak": step_peak, "unit": "TFLOP/s",
                 "frac": step_tflops / step_peak, "frac_vs_fp32_mfma_peak": step_tflops / FP32_MFMA_PEAK_TFLOPS,
                 "frac_of_bf16_peak": step_tflops / BF16_MFMA_PEAK_TFLOPS,
                 "algorithmic_gb_per_step": step_bytes / 1e9, "hbm_gbps_algorithmic": step_bytes / (dt / steps) / 1e9,
                 "hbm_frac_of_8tbps": step_bytes / (dt / steps) / 8e12,
                 "note": "whole step: the conv + attention launches' algorithmic FLOPs (2*MAC) and algorithmic HBM bytes (every launch's "
                         "input once, residual / accumulate inputs once, output once) / wall time of the step"},
        "all_instances": {k: {"ms_per_step": v["ms"] / steps, "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                              "hbm_gbps_algorithmic": v["bytes"] / (v["ms"] * 1e-3) / 1e9,
                              "launches_per_step": v["launches"] / steps}
                          for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
        "timed_kernels_share_of_step": kern_ms / (dt * 1e3)})
    return roof


SHORT_DTYPE = {"split3": "f32 (split-f16 x3 on the f16 MFMA, f32 accumulate)", "split6": "f32 (split-bf16 x6 on the bf16 MFMA, f32 accumulate)",
               "f32": "f32", "bf16": "bf16 operands, f32 accumulate"}
MAX_LINE_BYTES = 4096        # the driver keeps the last ~8 KB of stdout: the final line must fit with room to spare


def _r(v, sig=6):
    """floats to `sig` significant digits (the line is a report, not a checkpoint)"""
    if isinstance(v, float):
        return float(f"{v:.{sig}g}") if np.isfinite(v) else None
    if isinstance(v, dict):
        return {k: _r(x, sig) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, sig) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None} if d else None


def compact_line(full, math=None, details=None):
    """The ONE line the driver parses: the contract keys + roofline + cpu_baseline + the checks, without prose and per-instance tables
    (those go to `details`, a JSON file).  Always < MAX_LINE_BYTES."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline") if k in full}
    out["dtype"] = SHORT_DTYPE.get(math, full.get("dtype", ""))[:96] if math else full.get("dtype", "")[:96]
    if "bf16-resident" in full.get("dtype", ""):
        out["dtype"] = "bf16 operands + bf16-resident generator activations, f32 accumulate"
    out["data"] = full.get("data", "synthetic")
    cfg = full.get("config", {})
    out["config"] = _pick(cfg, ("workload", "baseline_config", "per_gpu_batch", "global_batch", "t_mel", "hop", "hidden", "parallelism",
                                "p_dropout", "realtime_factor", "streams", "vs_source_hash"))
    if out["config"] and len(out["config"].get("workload", "")) > 200:
        out["config"]["workload"] = out["config"]["workload"][:200]
    r = full.get("roofline")
    if r:
        rc = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                       "traffic_over_algorithmic", "avg_launch_ms", "launches_per_step", "algorithmic_gflop_per_launch", "share_of_step",
                       "executed_tflops", "frac_executed", "frac_vs_fp32_mfma_peak", "hbm_frac_of_8tbps"))
        if "traffic" not in rc:
            rc["traffic"] = None
        if r.get("traffic_source"):
            rc["traffic_source"] = r["traffic_source"].split(" (")[0]
        if r.get("mfma_executed"):
            rc["mfma_executed"] = _pick(r["mfma_executed"], ("tflops", "pipe_busy", "gfx_clock_ghz"))
        if r.get("step"):
            rc["step"] = _pick(r["step"], ("achieved", "peak", "frac", "algorithmic_tflop_per_step", "algorithmic_gb_per_step",
                                           "hbm_frac_of_8tbps", "launches_counted"))
        out["roofline"] = rc
    c = full.get("cpu_baseline")
    if c:
        out["cpu_baseline"] = _pick(c, ("value", "unit", "cores", "kind", "items", "of_items", "runs", "seconds", "samples", "waveform_max_abs_err"))
        out["cpu_baseline"]["sample"] = c.get("sample", "").split(":")[0][:160]
    for k in ("waveform_max_abs_err", "flow_logdet_rel_err", "generated_samples_per_s"):
        if k in full:
            out[k] = full[k]
    if "single_stream" in full:
        out["single_stream"] = _pick(full["single_stream"], ("ms_per_step", "value"))
        out.setdefault("roofline", {})["measured_on"] = "single-stream pass (same steps; kernel durations are not attributable under two-stream overlap)"
    if "flow_logdet" in full:
        out["flow_logdet_mean_only_exact_zero"] = full["flow_logdet"].get("mean_only_true_logdet_is_exact_zero")
    for k in ("fp32_mfma_engine", "split_bf16x6_engine"):
        if k in full:
            out[k] = _pick(full[k], ("value", "ms_per_step", "max_abs_waveform_diff_vs_value_run"))
    if "oracle_check" in full:
        out["oracle_check"] = _pick(full["oracle_check"], ("layer_rms_rel_err", "layer_max_abs_err", "tolerance_rms_rel", "seconds"))
    if "cpu_baseline_torch" in full:
        out["cpu_baseline_torch"] = _pick(full["cpu_baseline_torch"], ("value", "cores", "items", "runs", "seconds"))
    if "losses_last_step" in full:
        out["losses_finite"] = bool(all(np.isfinite(v) for v in full["losses_last_step"].values()))
    if details:
        out["details"] = details
    out = _r(out)
    line = json.dumps(out, separators=(",", ":"))
    if len(line) >= MAX_LINE_BYTES:          # never print a line the driver would truncate: drop the optional keys, largest first
        for k in ("cpu_baseline_torch", "split_bf16x6_engine", "details"):
            out.pop(k, None)
        out.get("roofline", {}).pop("traffic_source", None)
        line = json.dumps(out, separators=(",", ":"))
    assert len(line) < MAX_LINE_BYTES, len(line)
    return line


def write_details(obj, path=None):
    """everything the compact lines leave out (per-instance tables, notes, per-step statistics, the other engines) as one JSON file"""
    path = path or os.environ.get("VS_BENCH_DETAILS") or os.path.join(ROOT, "bench_details.json")
    try:
        with open(path, "w") as f:
            json.dump(obj, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)
        return None



class InferenceWorkload:
    """One inference configuration of BASELINE.json (headline / 2 / 4 / 5): model, resident inputs, step()."""

    def __init__(self, config, B, T, hidden, math, storage, hop, ragged, dev, rank=0, world=1):
        from visinger_amd import _lib as L
        from visinger_amd.dp import shard_batch
        self.config, self.B, self.T, self.hidden, self.math, self.storage, self.hop = config, B, T, hidden, math, storage, hop
        L.set_option("VS_CONV_MATH", MATH[math])      # arithmetic of every conv handle created from here on (vs_conv_create)
        self.model, self.hp = build_model(hop=hop, hidden=hidden)
        self.model = self.model.to(dev)
        if storage == "bf16":
            from visinger_amd.modules.hipconv import set_activation_storage
            set_activation_storage(self.model, torch.bfloat16)
        # global batch of B*world utterances, strided shard per rank (tasks/base.py:130-133)
        gb = synthetic_batch(B * world, T, T // 8, 64, 1234, "cpu", ragged=ragged, hidden=hidden)
        self.batch = [t.to(dev) for t in shard_batch(gb, rank, world)]
        text, pitch, dur, mel2ph, spk, noise = self.batch
        model = self.model
        if config == 2:      # flow inverse + generator only (BASELINE configs[1]); inputs: a prior sample z_p, the mask, the speaker
            with torch.no_grad():
                fmask = (mel2ph > 0).float().unsqueeze(1)
                g = model.speaker_embedding(None, spk).transpose(1, 2).contiguous()
                z_p = (noise * fmask).contiguous()
            self.c2_inputs = (z_p, fmask, spk)

            def step():
                with torch.no_grad():
                    z_q = model.flow(z_p, fmask, g=g, reverse=True) * fmask
                    return {"wav_out": model.decoder(z_q, g=g).squeeze(1)}
        else:
            def step():
                with torch.no_grad():
                    return model(text, pitch, dur, mel2ph, spk_id=spk, infer=True, noise=noise)
        self.step = step

    def workload_key(self):
        return workload_key(self.config, self.B, self.T, self.hidden, self.hop, self.storage)

    def dtype_name(self):
        if self.storage == "bf16":
            return "bf16 operands, f32 accumulate (bf16-resident activations between the generator's convs, f32 tensors elsewhere)"
        return DTYPE[self.math]

    def describe(self, world=1):
        what = ("VISinger synthesis (text-enc + pitch-pred + frame-prior + flow-inverse + HiFi-GAN), " if self.config != 2
                else "VISinger flow inverse + HiFi-GAN decode (BASELINE config 2), ")
        return {"workload": what + f"B={self.B}/GPU T_mel={self.T} hop={self.hop} hidden={self.hidden} " +
                            ("fp32 tensors" if self.storage == "f32" else "bf16-resident generator activations") + ", random-init weights",
                "baseline_config": self.config or "headline (north_star: B=32, T_mel=1024, hop 256)",
                "per_gpu_batch": self.B, "global_batch": self.B * world, "t_mel": self.T, "hop": self.hop, "hidden": self.hidden,
                "parallelism": f"dp{world} (utterance shard, no collective)", "vs_source_hash": library_source_hash()}


def timed_run(step, steps, warmup, profile, barrier, nstreams=1):
    """warm-up, then `steps` timed steps between two barriers.  nstreams > 1: the steps -- independent batches -- go through visinger_amd.synth.StreamRotation,
    the product's own batch pipeline (consecutive batches on alternating HIP streams: batch i + 1's latency-bound transformers under batch i's generator);
    the warm-up runs through the same rotation, so every stream's allocator pool is warm."""
    from visinger_amd.ops import PROFILER
    from visinger_amd.synth import StreamRotation
    rot = StreamRotation(nstreams, timing=True) if nstreams > 1 else None
    for i in range(warmup):
        out = rot.run(step)[0] if rot else step()
    if rot:
        rot.join()
    import gc
    nogc = not os.environ.get("VS_BENCH_GC") and gc.isenabled()      # (the cyclic collector out of the timed synthesis steps, as `timeit` does; a synthesis
    if nogc:                                                          #  loop allocates little: 0-0.2 ms a step either way)
        gc.collect()
        gc.disable()
    try:
        barrier()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        if profile:
            PROFILER.start()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            if rot:
                out, marks[i + 1] = rot.run(step)
            else:
                out = step()
                marks[i + 1].record()
        if rot:
            rot.join()
        barrier()
        dt = time.perf_counter() - t0
    finally:
        if nogc:
            gc.enable()
    if profile:
        PROFILER.stop()
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return out, dt, per_step


STREAMS_NOTE = ("value / ms_per_step: consecutive batches through visinger_amd.synth.StreamRotation ({n} HIP streams: batch i + 1's transformers under batch "
                "i's generator); roofline and single_stream: the same steps on ONE stream, where a kernel's HIP-event duration is its own")


def timed_pipeline(step, steps, warmup, barrier, nstreams):
    """(out, dt, per_step, single): the timed region `value` is quoted on, plus -- when it ran on more than one stream -- a second, single-stream pass of the
    same steps with the per-kernel HIP events on (under two-stream overlap a kernel's event duration includes the other stream's work: not a roofline input)."""
    if nstreams <= 1:
        out, dt, per_step = timed_run(step, steps, warmup, True, barrier)
        return out, dt, per_step, None
    out, dt, per_step = timed_run(step, steps, warmup, False, barrier, nstreams)
    _, dt1, per1 = timed_run(step, steps, min(warmup, 3), True, barrier)
    return out, dt, per_step, {"ms_per_step": dt1 / steps * 1e3, "ms_per_step_stats": percentile_stats(per1), "dt": dt1}


def other_config_line(config, dev, steps, warmup, barrier, with_cpu=True, nstreams=2):
    """A reduced bench line for BASELINE config 2 / 3 / 5 inside the default run: same timing discipline as the
    headline (warm-up, barrier + synchronize on both sides, HIP events per launch for the roofline), fewer steps."""
    from visinger_amd.ops import PROFILER
    preset = CONFIGS[config]
    if config == 3:
        return train_line(preset["batch"], preset["frames"], 0.1, "split3", steps, warmup, 0, 1, None, dev, barrier)
    math = "bf16" if preset.get("math") == "bf16" else "split3"
    wl = InferenceWorkload(config, preset["batch"], preset["frames"], preset.get("hidden", 192), math, "bf16" if config == 5 else "f32", 256, False, dev)
    out, dt, per_step, single = timed_pipeline(wl.step, steps, warmup, barrier, nstreams)
    wav = out["wav_out"]
    assert wav.shape == (wl.B, wl.T * 256) and bool(torch.isfinite(wav).all())
    samples = wl.B * wl.T * 256 * steps
    line = {"metric": "audio samples/sec (22.05 kHz)", "value": samples / dt, "unit": "audio samples/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "ms_per_step": dt / steps * 1e3, "ms_per_step_stats": percentile_stats(per_step), "dtype": wl.dtype_name(), "data": "synthetic",
            "config": dict(wl.describe(), realtime_factor=samples / dt / SR, what=preset["what"], streams=nstreams),
            "roofline": roofline_from_profile(PROFILER.summary(), single["dt"] if single else dt, steps, wl.workload_key())}
    if single:
        line["single_stream"] = {"ms_per_step": single["ms_per_step"], "value": samples / single["dt"]}
        line["streams_note"] = STREAMS_NOTE.format(n=nstreams)
    if config == 5 and with_cpu and out.get("f0_pred") is not None:
        line["oracle_check"] = oracle_check_config5(wl.model, wl.hp, wl.batch, out["f0_pred"])
    if config == 2 and with_cpu:
        line["cpu_baseline"], wav_cpu = cpu_baseline_config2(wl.model, wl.hp, wl.c2_inputs)
        line["cpu_baseline"]["waveform_max_abs_err"] = float(np.abs(wav[:wav_cpu.shape[0]].double().cpu().numpy() - wav_cpu).max())
        line["cpu_baseline"]["waveform_tolerance"] = 1e-4
    return line


def main():
    args = parse_args()
    if args.print_workload_key:
        print(workload_key(args.config, args.batch, args.frames, args.hidden, args.hop, args.storage))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))        # this process never touches the GPU
    if args.dry_run:
        return dry_run(args)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    rank, local_rank, world, dist, backend = init_ranks(args)
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from visinger_amd.ops import PROFILER
    from visinger_amd.dp import max_over_ranks
    global HOP, SR
    HOP, SR = args.hop, (22050 if args.hop == 256 else 24000)
    B, T = args.batch, args.frames

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.config == 3:
        line = train_line(B, T, args.dropout, args.math, args.steps, args.warmup, rank, world, dist, dev, barrier)
        if rank == 0:
            det = write_details({"line": line})
            print(compact_line(line, args.math, det), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # BASELINE configs 3 / 2 / 5 of the default run FIRST (one compact line each, printed before the headline line -- the driver parses the last line of
    # stdout -- 10 timed steps each, 20 for the training step, after 6 warm-up steps).  They used to follow the headline's runs in the same process; the training
    # step, which is as much host- as device-bound, then read 87-90 ms where its own process reads 81-84 (seven default runs against nine of its own, round 5):
    # it goes first, before the CPU baselines have started their thread pools and the headline's ~10^5 launches have gone through the runtime.
    pre = {}
    if world == 1 and rank == 0 and not args.no_cpu_baseline and args.config in (0, 4) and not args.no_other_configs:
        for c in (3, 2, 5):
            oc = other_config_line(c, dev, 20 if c == 3 else 10, 6, barrier, nstreams=args.streams)
            oc["n_gpus"] = 1
            pre[c] = oc
            torch.cuda.empty_cache()

    wl = InferenceWorkload(args.config, B, T, args.hidden, args.math, args.storage, args.hop, args.ragged, dev, rank, world)
    model, hp = wl.model, wl.hp
    out_dev, dt, per_step, single = timed_pipeline(wl.step, args.steps, args.warmup, barrier, args.streams)
    wav = out_dev["wav_out"]
    assert wav.shape == (B, T * HOP) and bool(torch.isfinite(wav).all())
    dt = max_over_ranks(dt, device=dev if backend == "nccl" else None)
    samples = B * world * T * HOP * args.steps

    if rank == 0:
        out = {
            "metric": "audio samples/sec (22.05 kHz) + flow log-det rel-err, B=32 T_mel=1024",
            "value": samples / dt,
            "unit": "audio samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_stats": percentile_stats(per_step),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype_name(),
            "data": "synthetic",
            "config": dict(wl.describe(world), realtime_factor=samples / dt / SR, streams=args.streams),
            "roofline": roofline_from_profile(PROFILER.summary(), single["dt"] if single else dt, args.steps, wl.workload_key()),
        }
        if single:
            out["single_stream"] = {"ms_per_step": single["ms_per_step"], "value": B * T * HOP * args.steps / single["dt"], "ms_per_step_stats": single["ms_per_step_stats"]}
            out["streams_note"] = STREAMS_NOTE.format(n=args.streams)
        headline = world == 1 and not args.no_cpu_baseline and args.config in (0, 4)
        if headline:
            # the CPU oracle on item 0 of the timed batch, the same graph: the reported baseline AND the checker of the timed run
            out["cpu_baseline"] = cpu_baseline(model, hp, wl.batch, wav_dev=wav, f0_dev=out_dev.get("f0_pred"))
            out["waveform_max_abs_err"] = out["cpu_baseline"]["waveform_max_abs_err"]
            if not args.quick:
                out["cpu_baseline_torch"] = cpu_baseline(model, hp, wl.batch, backend="torch")
            out["flow_logdet"] = flow_logdet_check(model, dev)
            out["flow_logdet_rel_err"] = out["flow_logdet"]["rel_err"]
        elif world == 1 and args.config == 2 and not args.no_cpu_baseline:
            out["cpu_baseline"], wav_cpu = cpu_baseline_config2(model, hp, wl.c2_inputs)
            out["waveform_max_abs_err"] = float(np.abs(wav[:wav_cpu.shape[0]].double().cpu().numpy() - wav_cpu).max())
        if world == 1 and args.math in ("split3", "split6") and not args.no_cpu_baseline:
            # the same workload on the other fp32-class engines, same process, same weights, same --steps / --warmup: the exact-fp32 MFMA /
            # F(2,3) kernels (the strictly-same-precision number to hold the headline arithmetic against: VERDICT r3 weak #2) and, outside
            # --quick, the split-bf16 x6 engine that was the default of rounds 1-2 (error against fp64: DESIGN.md 4, tests/test_conv_split_gpu.py)
            from visinger_amd.modules.hipconv import set_conv_math
            for other, key, note in (("f32", "fp32_mfma_engine", "bench.py --math f32: v_mfma_f32_32x32x2_f32 + Winograd F(2,3) kernels, no bf16 / f16 anywhere"),
                                     ("split6", "split_bf16x6_engine", "bench.py --math split6: three exact bf16 planes, six cross products (the default of rounds 1-2)")):
                if other == args.math or (args.quick and other != "f32"):
                    continue
                set_conv_math(model, MATH[other])
                o2, dt2, per2 = timed_run(wl.step, args.steps, args.warmup, False, barrier)
                out[key] = {"value": B * T * HOP * args.steps / dt2, "unit": "audio samples/s", "ms_per_step": dt2 / args.steps * 1e3,
                            "ms_per_step_stats": percentile_stats(per2), "steps": args.steps, "warmup": args.warmup,
                            "max_abs_waveform_diff_vs_value_run": float((o2["wav_out"] - wav).abs().max()), "note": note}
            set_conv_math(model, MATH[args.math])
        details = {"headline": out, "other_configs": {}}
        lines = []
        for c in (2, 3, 5):
            if c in pre:
                details["other_configs"][str(c)] = pre[c]
                lines.append(compact_line(pre[c], CONFIGS[c].get("math", "split3")))
        det = write_details(details)
        for ln in lines:
            print(ln, flush=True)
        print(compact_line(out, args.math, det), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_line(B, T, dropout, math, steps, warmup, rank, world, dist, dev, barrier):
    """BASELINE config 3: one full GAN training step (generator pass + discriminator pass, both optimizers; under N > 1 ranks the
    stock DistributedDataParallel gradient all-reduce over RCCL) on synthetic B=16, T_mel=512, segment 32 frames, hop 256.
    -> the bench line (dict).  `roofline.step`: algorithmic FLOPs of one step -- every conv / attention launch of the forward, the
    grad-input launches and the weight-gradient launches, counted by ops.PROFILER in counting mode on one extra un-timed step (no
    events: 6 000 launches per step) -- over the measured step time."""
    from visinger_amd import _lib as L
    from visinger_amd.dp import max_over_ranks
    from visinger_amd.models.visinger import hop256_hparams
    from visinger_amd.ops import PROFILER
    from visinger_amd.train import VISingerTrainer, synthetic_train_batch
    L.set_option("VS_CONV_MATH", MATH[math])
    hp = hop256_hparams(p_dropout=dropout)        # the reference trains with p_dropout 0.1 (config/models/visinger.yaml:9)
    torch.manual_seed(1234)
    tr = VISingerTrainer(64, 117, 131, hp).to(dev).configure().train()
    runner = tr
    if dist is not None:
        runner = torch.nn.parallel.DistributedDataParallel(tr, device_ids=[dev.index], find_unused_parameters=True)
    batch = synthetic_train_batch(B, T, T // 8, tr.hop, 64, hp["num_linear_bins"], 1234 + rank, dev)
    for _ in range(warmup):
        tr.training_step(batch, runner=runner)
    # (the cyclic collector: VISingerTrainer.training_step keeps it out of a step and runs it between steps every `gc_every` steps -- the PRODUCT's policy since
    #  round 6 (the benchmark used to switch it off around the timed loop, which a user's loop did not); VS_BENCH_GC=1: the collector left alone)
    if os.environ.get("VS_BENCH_GC"):
        tr.gc_every = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        logs = tr.training_step(batch, runner=runner)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, device=dev if dist is not None else None)
    PROFILER.start(count_only=True)
    tr.training_step(batch, runner=runner)
    torch.cuda.synchronize()
    PROFILER.stop()
    counts = PROFILER.counts()
    step_flops = sum(v["flops"] for v in counts.values())
    step_tflops = step_flops / (dt / steps) / 1e12
    assert all(np.isfinite(v) for v in logs.values()), logs
    peak = BF16_MFMA_PEAK_TFLOPS / MATH[math] if math in ("split3", "split6") else (FP32_MFMA_PEAK_TFLOPS if math == "f32" else BF16_MFMA_PEAK_TFLOPS)
    return {
        "metric": "GAN training steps/sec (BASELINE config 3: posterior + flow fwd + MRF + MPD/MSD, both optimizer passes)",
        "value": steps * world / dt, "unit": "global steps/s (x n_gpus batches of B)", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE[math], "data": "synthetic",
        "config": {"workload": f"VISinger GAN training step, B={B}/GPU T_mel={T} segment={tr.segment_size} hop={tr.hop}, reference-size "
                               "generator + MPD/MSD, AdamW x2, random-init weights", "baseline_config": 3, "p_dropout": dropout, "per_gpu_batch": B,
                   "global_batch": B * world, "t_mel": T, "parallelism": f"dp{world} (DDP gradient all-reduce over RCCL)",
                   "vs_source_hash": library_source_hash()},
        "roofline": {"bound": "mfma", "unit": "TFLOP/s", "achieved": step_tflops, "peak": peak, "frac": step_tflops / peak,
                     "frac_vs_fp32_mfma_peak": step_tflops / FP32_MFMA_PEAK_TFLOPS,
                     "traffic": (pmc_step_traffic(workload_key(3, B, T, 192, 256, "f32")) or {}).get("bytes_per_step"),
                     "traffic_source": (pmc_step_traffic(workload_key(3, B, T, 192, 256, "f32")) or {}).get("source"),
                     "step": {"algorithmic_tflop_per_step": step_flops / 1e12,
                              "by_kind_tflop": {k: v["flops"] / 1e12 for k, v in sorted(counts.items(), key=lambda kv: -kv[1]["flops"])[:12]},
                              "launches_counted": sum(v["launches"] for v in counts.values())},
                     "note": "whole step (no single dominant kernel: 6 000 launches): algorithmic FLOPs (2*MAC) of every conv / attention "
                             "launch of the forward, the grad-input launches and the weight-gradient launches (library GEMMs of the 1x1 / "
                             "wide discriminator weight gradients included at their 2*MAC), counted on one extra step, / measured step time; "
                             f"peak = the roof of the arithmetic ({peak:.0f} TFLOP/s: dense f16 / bf16 MFMA peak over the cross products per fp32 product of --math {math})"},
        "generated_samples_per_s": B * world * tr.segment_size * tr.hop * steps / dt,
        "losses_last_step": logs}


if __name__ == "__main__":
    main()
