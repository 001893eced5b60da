"""per-step timeline of the wave-specialised ResBlock step at C = 32 (csrc/pair32s.hip: pairw_kernel) from its in-kernel cycle
stamps: python tools/stamp_pair32.py [k]"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import ops, packing, _lib

B, C, T, dev = 32, 32, 40000, "cuda"
k = int(sys.argv[1]) if len(sys.argv) > 1 else 11
d = 5
x = torch.randn(B, C, T, device=dev)
pk = packing.pack_conv_weight_f16x3
w1, w2 = pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k)), pk(torch.randn(C, C, k, device=dev) * 0.6 / np.sqrt(C * k))
b1, b2 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
xs = ops.act_split(x, 0.1)
ys = ops.split_like(B, C, T, dev)
run = lambda: ops.resblock_pair(x, w1, b1, w2, b2, k, d, x_split=xs, y_split=ys, y_split_slope=0.1, planes_residual=True, no_y=True)
n = _lib.lib().sat_pair32_debug_stamps(None)
buf = torch.zeros(n, dtype=torch.int64, device=dev)
for _ in range(2):
    run()
torch.cuda.synchronize()
_lib.lib().sat_pair32_debug_stamps(buf.data_ptr())
run()
torch.cuda.synchronize()
_lib.lib().sat_pair32_debug_stamps(None)
st = buf.cpu().numpy().reshape(8, 6, 8)
t0 = st[1, 1].min()
print("cycles since the barrier of step 1; rows = steps, per wave: [barrier release | end of subtile 0 1 2 3 | -> next barrier arrival]")
for step in range(1, 6):
    for role, ws in (("conv1", range(0, 4)), ("conv2", range(4, 8))):
        for w in ws:
            s = st[step, :, w]
            nxt = st[step + 1, 0, w]
            print(f"  step {step} {role} wave {w}: released {int(s[1] - t0):7d}  subtiles +{int(s[2] - s[1]):5d} +{int(s[3] - s[2]):5d} +{int(s[4] - s[3]):5d} +{int(s[5] - s[4]) if s[5] else 0:5d}   arrives {int(nxt - t0):7d}")
