import os, sys, torch
sys.path.insert(0, os.getcwd())
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
def bench(C,k,d,T,B=32):
    pad=(k*d-d)//2
    op=ConvOp(L.CONV1D,C,C,k,d,pad)
    op.set_weights(torch.randn(C,C,k,device="cuda")*0.05,None,torch.randn(C,device="cuda"))
    x=torch.randn(B,C,T,device="cuda"); y=torch.empty_like(x); res=torch.randn_like(x)
    for _ in range(2): op.forward(x,y=y,res=res,in_act=L.IN_LRELU)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): op.forward(x,y=y,res=res,in_act=L.IN_LRELU)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/5*1e3
for dbg in sys.argv[1:]:
    L.set_option("VS_WINO_DBG", int(dbg))
    print("dbg",dbg, "C128 k11: %.0f us  k3: %.0f us  k7: %.0f   C64 k11: %.0f  k3: %.0f" % (bench(128,11,1,65536), bench(128,3,1,65536), bench(128,7,1,65536), bench(64,11,1,131072), bench(64,3,1,131072)), flush=True)
