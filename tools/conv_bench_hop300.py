import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from visinger_amd import _lib as L
import conv_bench as cb
tot = 0
for C, T in ((256, 5120), (128, 25600), (64, 76800), (32, 153600), (16, 307200)):
    for k in (3, 7, 11):
        for d in (1, 3, 5):
            ms = cb.bench(f"hop300 C={C} k={k} d={d}", L.CONV1D, C, C, k, d, T, in_act=L.IN_LRELU, res=True)
            tot += ms * (4 if d == 1 else 1)
    print(f"  stage C={C} cumulative {tot:.1f} ms", flush=True)
cb.bench("ups0 512->256 k11 u5", L.CONV_TRANSPOSE1D, 512, 256, 11, 5, 1024, in_act=L.IN_LRELU)
cb.bench("ups1 256->128 k11 u5", L.CONV_TRANSPOSE1D, 256, 128, 11, 5, 5120, in_act=L.IN_LRELU)
cb.bench("ups2 128->64 k7 u3", L.CONV_TRANSPOSE1D, 128, 64, 7, 3, 25600, in_act=L.IN_LRELU)
cb.bench("ups3 64->32 k4 u2", L.CONV_TRANSPOSE1D, 64, 32, 4, 2, 76800, in_act=L.IN_LRELU)
cb.bench("ups4 32->16 k4 u2", L.CONV_TRANSPOSE1D, 32, 16, 4, 2, 153600, in_act=L.IN_LRELU)
cb.bench("conv_post 16->1 k7", L.CONV1D, 16, 1, 7, 1, 307200, in_act=L.IN_LRELU, out_act=L.OUT_TANH)
