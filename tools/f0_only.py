"""YAAPT alone on 32 x 5 s utterances: time per batch (events on the launch stream) — run under rocprofv3
--kernel-trace --stats for the per-kernel split"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satools_amd import f0 as f0_hip
from satools_amd import synthetic

OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
wav = synthetic.harm_batch(list(range(B))).to("cuda")
for _ in range(3):
    f0_hip.yaapt(wav, OPTS)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f0_hip.yaapt(wav, OPTS, defer_status=True)
e1.record()
torch.cuda.synchronize()
# the prefilter alone: events around a run with everything but the first launch... not separable through the C ABI;
# use rocprofv3 --kernel-trace --stats for the split
print(f"get_f0, batch {B} x 5 s: {e0.elapsed_time(e1) / 20:.3f} ms per batch")
