"""Cycle stamps of the LDS-DMA ring conv (conv_ring16.hip, diagnostic instantiation): per block the prologue, the K loop, the
share of the loop spent at the step heads (s_waitcnt + s_barrier), the epilogue, and the clock the kernel held."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import satools_amd  # noqa: E402,F401
from satools_amd import ops, packing, _lib  # noqa: E402

B, dev = 32, "cuda"
l = _lib.lib()
for C, T in ((256, 1250), (128, 5000)):
    x = torch.randn(B, C, T, device=dev)
    xs = ops.act_split(x, 0.1)
    ys = ops.split_like(B, C, T, dev)
    for k, bits in ((3, 1), (11, 1), (11, 1 + 8), (11, 1 + 16), (11, 1 + 8 + 16)):
        _lib.check(l.sat_conv_set_option(b"convring", bits), "opt")
        w = packing.pack_conv_weight_f16x3(torch.randn(C, C, k, device=dev) * (k * C) ** -0.5)
        b = torch.randn(C, device=dev)
        f = lambda: ops.conv1d(x, w, C, k, bias=b, dilation=1, pad_left=(k - 1) // 2, mode=1, x_split=xs, y_split=ys, y_split_slope=0.1, no_y=True)
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
        for h, name in ((0, "early"), (1, "late ")):
            q = r[:, h]
            clk = (q[:, 4] / q[:, 5]).median().item() * 0.1
            ns = q[0, 7].item()
            print(f"C {C} k {k:2d} diag {bits - 1:2d} {name}: blocks {len(q)}  first-operand wait {q[:, 0].median():7.0f}  loop {q[:, 1].median():8.0f} ({q[:, 1].median() / (C // 32 * k * ns):6.0f} / step, {ns:.0f} tiles)"
                  f"  waits {q[:, 2].median():8.0f} ({100 * (q[:, 2] / q[:, 1]).median():4.1f} %)  epilogue {q[:, 3].median():6.0f}  kernel {q[:, 4].median():8.0f} cyc"
                  f" = {q[:, 5].median() / 100:6.1f} us at {clk:4.2f} GHz; start spread {(r[:, 0, 6].max() - r[:, 0, 6].min()) / 100:5.1f} us", flush=True)
_lib.check(l.sat_conv_set_option(b"convring", 0), "opt")
