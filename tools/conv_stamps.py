#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of one conv launch (uses the vs_debug_set_stamp_buffer hook)."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp

C, k, d, T, B = int(os.environ.get("C", 128)), int(os.environ.get("K", 3)), int(os.environ.get("D", 1)), int(os.environ.get("T", 65536)), int(os.environ.get("B", 32))
if os.environ.get("MATH"):
    L.set_option("VS_CONV_MATH", int(os.environ["MATH"]))
CO = int(os.environ.get("CO", C))                       # output channels (default: C -> C)
ACT = {"none": L.IN_NONE, "lrelu": L.IN_LRELU}[os.environ.get("ACT", "lrelu")]
op = ConvOp(L.CONV1D, C, CO, k, d, (k * d - d) // 2)
op.set_weights(torch.randn(CO, C, k, device="cuda") * 0.05, None, torch.randn(CO, device="cuda"))
x = torch.randn(B, C, T, device="cuda"); y = torch.empty(B, CO, T, device="cuda"); res = torch.randn_like(y)
if os.environ.get("DT") == "bf16":      # bf16-resident tensors (MATH=1): conv_split_kernel_bf16io
    x, y, res = x.bfloat16(), y.bfloat16(), res.bfloat16()
use_res = os.environ.get("RES", "1") == "1"
for _ in range(2):
    op.forward(x, y=y, res=res if use_res else None, in_act=ACT)
nblk = 65536
buf = torch.zeros(nblk * 64, dtype=torch.int64, device="cuda")
lib = L.lib()
lib.vs_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.vs_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
op.forward(x, y=y, res=res if use_res else None, in_act=ACT)
torch.cuda.synchronize()
lib.vs_debug_set_stamp_buffer(None)
full = buf.cpu().numpy().reshape(-1, 64)
full = full[full[:, 0] != 0]
s = full[:, :8]
t0 = s[:, 0].min()
st = (s[:, :4] - t0) / 100.0     # us
print("workgroups", len(s), "launch span %.1f us" % st[:, 3].max())
pro, main, epi = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
issued = (s[:, 6] - s[:, 2]) / 100.0 if (s[:, 6] != 0).all() else epi * 0
print("kernel:", op.kernel_instance())
for name, v in (("prologue", pro), ("main", main), ("epilogue", epi), ("epi-issue", issued), ("total", st[:, 3] - st[:, 0])):
    print(f"{name:9s} mean {v.mean():8.2f}  p10 {np.percentile(v,10):8.2f}  p50 {np.percentile(v,50):8.2f}  p90 {np.percentile(v,90):8.2f}  max {v.max():8.2f} us")
if (full[:, 8] != 0).all():
    # (-DVS_EPI_STAMPS builds, tools/build_variant.py: wave 0's time stamps inside the vector epilogue; every stamp also waits for the LDS queue)
    names = {8: "first residual loads issued", 16: "pass0 LDS written (+r1 issued)", 17: "pass0 stored", 18: "pass1 LDS written", 19: "pass1 stored",
             20: "pass2 LDS written", 21: "pass2 stored", 22: "pass3 LDS written", 23: "pass3 stored", 6: "loop end", 3: "all stores acknowledged"}
    prev = full[:, 2]
    for slot in (8, 16, 17, 18, 19, 20, 21, 22, 23, 6, 3):
        if (full[:, slot] != 0).all():
            dlt = (full[:, slot] - prev) / 100.0
            print(f"   epilogue +{names[slot]:32s} mean {dlt.mean():6.2f}  p50 {np.percentile(dlt,50):6.2f}  p90 {np.percentile(dlt,90):6.2f} us")
            prev = full[:, slot]
cyc = (s[:, 5] - s[:, 4]).astype(np.float64)
print("main-loop shader cycles: mean %.0f ; implied clock %.3f GHz" % (cyc.mean(), (cyc / (main * 1e3)).mean()))
hw = s[:, 7] & 0xffffffff
xcc = s[:, 7] >> 32
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x3   # HW_ID layout guess: cu_id[11:8], sh_id[12], se_id[15:13]
print("xcc ids seen:", np.unique(xcc))
# first 16 workgroups: start time, placement
order = np.argsort(s[:, 0])
for i in order[:12]:
    print(f"wg {i:5d} start {st[i,0]:8.2f} pro {pro[i]:6.2f} main {main[i]:7.2f} epi {epi[i]:6.2f}  xcc {xcc[i]} hw {hw[i]:08x}")
# concurrency: how many workgroups are in their epilogue at each time
ts = np.linspace(0, st[:, 3].max(), 400)
in_epi = [(np.logical_and(st[:, 2] <= t, st[:, 3] > t)).sum() for t in ts]
in_main = [(np.logical_and(st[:, 1] <= t, st[:, 2] > t)).sum() for t in ts]
print("in-epilogue count: mean %.1f max %d ; in-main mean %.1f" % (np.mean(in_epi), np.max(in_epi), np.mean(in_main)))
ts2 = np.linspace(0, st[:, 3].max(), 4000)
ie = np.array([(np.logical_and(st[:, 2] <= t, st[:, 3] > t)).sum() for t in ts2])
for lo, hi in ((0.0, 0.1), (0.3, 0.4), (0.6, 0.7), (0.85, 0.95)):
    seg = ie[int(lo * len(ie)):int(hi * len(ie))]
    print("  in-epilogue over [%.0f%%, %.0f%%] of the launch: p5 %d p25 %d p50 %d p75 %d p95 %d max %d" % (
        lo * 100, hi * 100, *np.percentile(seg, [5, 25, 50, 75, 95]).astype(int), seg.max()))
# starts per round
starts = np.sort(st[:, 0])
print("start-time quantiles:", np.percentile(starts, [0, 5, 6.3, 12.5, 25, 50, 75, 100]).round(1))

if os.environ.get("STEPS") == "1":        # (-DVS_SPLIT_PERTURB builds: per-step ticks of the main loop in slots 8..)
    steps = full[:, 8:]
    nst = int((steps[0] != 0).sum())
    d = np.diff(steps[:, :nst].astype(np.float64), axis=1)
    print("per-step shader cycles (median over workgroups):", np.median(d, axis=0).round(0).astype(int).tolist())
    print("last step -> main end:", np.median(full[:, 5] - steps[:, nst - 1]))

# which workgroups share a CU in the first dispatch round?
first = np.where(st[:, 0] < 5.0)[0]
cu_of = {}
for i in first:
    key = (int(xcc[i]), int((hw[i] >> 13) & 7), int((hw[i] >> 12) & 1), int((hw[i] >> 8) & 0xf))
    cu_of.setdefault(key, []).append(int(i))
print("first round: %d workgroups on %d CUs" % (len(first), len(cu_of)))
diffs = {}
for k, v in cu_of.items():
    if len(v) == 2:
        diffs[abs(v[0] - v[1])] = diffs.get(abs(v[0] - v[1]), 0) + 1
print("id distance of CU-mates:", sorted(diffs.items(), key=lambda kv: -kv[1])[:6])
print("examples:", list(cu_of.items())[:4])
