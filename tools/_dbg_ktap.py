import sys, torch
sys.path.insert(0, '/root/repo')
from visinger_amd import _lib as L
from visinger_amd.ops import ConvOp
L.set_option("VS_CONV_MATH", 3)
def go(cin, cout, k, B, T, pad):
    g = torch.Generator(device="cuda").manual_seed(1)
    op = ConvOp(L.CONV1D, cin, cout, k, 1, pad)
    w = torch.randn(cout, cin, k, device="cuda", generator=g) * (cin*k) ** -0.5
    op.set_weights(w, None, torch.zeros(cout, device="cuda"))
    x = torch.randn(B, cin, T, device="cuda", generator=g)
    ref = torch.nn.functional.conv1d(x.double(), w.double(), padding=pad)
    outs = {}
    for no in (1, 0):
        L.set_option("VS_NO_KTAP", no)
        y = op.forward(x)
        outs[no] = y.clone()
        e = (y.double() - ref).abs()
        print(cin, cout, k, B, T, pad, op.kernel_instance(), tuple(y.shape), "max err vs fp64", float(e.max()))
    d = (outs[0] - outs[1]).abs(); bad = (d > 0).nonzero()
    print("   old vs new max", float(d.max()), "nbad", len(bad), "cols/128 uniq", sorted(set((bad[:,2]//128).tolist()))[:20], "rows/32", sorted(set((bad[:,1]//32).tolist()))[:10], "col%128 min/max", (int((bad[:,2]%128).min()), int((bad[:,2]%128).max())) if len(bad) else None)
go(1536, 1024, 2, 1, 1937, 0)
go(1536, 1024, 2, 1, 1936, 0)
go(256, 128, 2, 1, 1937, 0)
go(256, 128, 2, 4, 512, 0)
go(256, 128, 2, 4, 512, 1)
go(256, 128, 3, 4, 512, 0)
