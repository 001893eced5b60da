"""time of the generator's output stage alone (32 x 16 x 64000): python tools/scratch/time_convpost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from satools_amd import ops
torch.manual_seed(0)
x = torch.randn(32, 16, 64000, device="cuda")
w = torch.randn(16, 7, device="cuda") * 0.1
b = torch.randn(1, device="cuda")
from satools_amd import _lib
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for quad in (0, 1, 0, 1):
  _lib.check(_lib.lib().sat_conv_set_option(b"convpost_quad", quad), "set_option")
  ts = []
  for _ in range(5):
    y = ops.convpost(x, w, b)
  for r in range(5):
    e0.record()
    for _ in range(20):
        y = ops.convpost(x, w, b)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
  print(f"quad={quad} convpost us per launch:", " ".join(f"{t:.1f}" for t in ts), " checksum %.9e" % float(y.double().sum()), " crc", hex(int(y.view(torch.int32).to(torch.int64).sum()) & 0xffffffffffff))
