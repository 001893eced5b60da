"""Cycle stamps of the LDS-DMA ring conv in both arithmetic modes (conv_ring16.hip, diagnostic instantiation): per block the
prologue, the K loop, the share of the loop spent at the step heads (s_waitcnt + s_barrier), the epilogue, the clock.
diag bits: 8 = no DMA issue in the loop, 16 = no fragment reads in the loop (results are wrong).  WR=3 in the environment: 128 x 320 tiles at C = 256."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd  # noqa: E402,F401
from satools_amd import ops, packing, _lib  # noqa: E402

B, dev = 32, "cuda"
l = _lib.lib()
_lib.check(l.sat_conv_set_option(b"convring_wr", int(os.environ.get("WR", "0"))), "opt")
for C, T in ((256, 1250), (128, 5000)):
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    xs8 = ops.planes_f8_sidecar(xs)
    ys = ops.split_like(B, C, T, dev)
    for k, bits in ((11, 1), (11, 1 + 8), (11, 1 + 16), (11, 1 + 8 + 16), (3, 1)):
        for mode in (1, 3):
            _lib.check(l.sat_conv_set_option(b"convring", bits), "opt")
            wf = torch.randn(C, C, k, device=dev) * (k * C) ** -0.5
            w = packing.pack_conv_weight_f16f8r(wf) if mode == 3 else packing.pack_conv_weight_f16x3(wf)
            b = torch.randn(C, device=dev)
            kw = dict(x_split8=xs8) if mode == 3 else {}
            f = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=1, pad_left=(k - 1) // 2, mode=mode, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True, **kw)
            for _ in range(10):
                f()
            nblk = 4096
            buf = torch.zeros(nblk * 2 * 8, dtype=torch.int64, device=dev)
            l.sat_convring_debug_stamps(buf.data_ptr())
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            l.sat_convring_debug_stamps(None)
            r = buf.view(nblk, 2, 8).cpu().double()
            r = r[r[:, 0, 4] > 0]
            nstep = (2 * ((k + 1) // 2)) if mode == 3 else k
            for h, name in ((0, "early"), (1, "late ")):
                q = r[:, h]
                clk = (q[:, 4] / q[:, 5]).median().item() * 0.1
                ns = q[0, 7].item()
                print(f"C {C} k {k:2d} {'f8r ' if mode == 3 else 'f16x3'} diag {bits - 1:2d} {name}: loop {q[:, 1].median():8.0f} ({q[:, 1].median() / (C // 32 * nstep * ns):6.0f} / step x {nstep * C // 32}, {ns:.0f} tiles)"
                      f"  waits {q[:, 2].median():8.0f} ({100 * (q[:, 2] / q[:, 1]).median():4.1f} %)  first-operand wait {q[:, 0].median():6.0f}  epilogue {q[:, 3].median():6.0f}  kernel {q[:, 4].median():8.0f} cyc"
                      f" = {q[:, 5].median() / 100:6.1f} us at {clk:4.2f} GHz", flush=True)
_lib.check(l.sat_conv_set_option(b"convring", 1), "opt")
_lib.check(l.sat_conv_set_option(b"convring_wr", 0), "opt")
