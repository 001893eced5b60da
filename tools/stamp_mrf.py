"""per-phase timeline of the fused MRF block (csrc/mrf.hip) from its in-kernel cycle stamps:
python tools/stamp_mrf.py [k]   (k = 3 / 7 / 11: one branch; default: the three-branch block)"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib

B, C, T = 32, 16, 80000
dev = "cuda"
ks = [int(a) for a in sys.argv[1:] if a.isdigit()] or [3, 7, 11]
x = torch.randn(B, C, T, device=dev)
xs = ops.act_split(x, 0.1)
pk = packing.pack_conv_weight_f16x3
branches = [(k, [(pk(torch.randn(C, C, k, device=dev) * 0.7 / np.sqrt(C * k)), torch.randn(C, device=dev) * 0.1,
                  pk(torch.randn(C, C, k, device=dev) * 0.7 / np.sqrt(C * k)), torch.randn(C, device=dev) * 0.1) for _ in range(3)]) for k in ks]
out = torch.empty(B, C, T, device=dev)
n = _lib.lib().sat_mrf_debug_stamps(None)
buf = torch.zeros(n, dtype=torch.int64, device=dev)
for _ in range(2):
    ops.resblock_mrf(xs, B, C, T, branches, out=out, out_div=3.0)
torch.cuda.synchronize()
_lib.lib().sat_mrf_debug_stamps(buf.data_ptr())
ops.resblock_mrf(xs, B, C, T, branches, out=out, out_div=3.0)
torch.cuda.synchronize()
_lib.lib().sat_mrf_debug_stamps(None)
st = buf.cpu().numpy().reshape(4, 80, 8)
nph = 6 * len(ks)
for tile in (1, 2):
    s = st[tile]
    print(f"tile visit {tile}: whole tile {int(s[4 * nph].max() - s[0].min())} cycles")
    print(" phase   read-ops  work(min/med/max over waves)  commit  barrier-wait(min/max)   phase total")
    for ph in range(nph):
        b = 1 + 4 * ph
        start = s[b - 1]                            # after the previous barrier
        rd = s[b] - start
        work = s[b + 1] - s[b]
        com = s[b + 2] - s[b + 1]
        bar = s[b + 3] - s[b + 2]
        tot = s[b + 3] - start
        print(f"  {ph:2d}    {int(np.median(rd)):6d}    {int(work.min()):6d} {int(np.median(work)):6d} {int(work.max()):6d}      {int(np.median(com)):6d}   {int(bar.min()):6d} {int(bar.max()):6d}        {int(np.median(tot)):6d}")
