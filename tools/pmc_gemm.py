"""one long 1x1 conv on split planes (4096 -> 4096, 128 utterances x 249 frames: ~1.5 ms) through both kernels, for
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE: MFMA utilisation and the clock the chip holds under each"""
import sys
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import ops, packing, _lib

B, T, cin, cout = 128, 249, 4096, 4096
x = torch.randn(B, cin, T, device="cuda")
w = torch.randn(cout, cin, 1, device="cuda") / cin ** 0.5
wp = packing.pack_conv_weight_f16x3(w)
xs = ops.act_split(x, 1.0)
for opt in (0, 1):
    _lib.check(_lib.lib().sat_conv_set_option(b"k1_gemm", opt), "set_option")
    for _ in range(6):
        ops.conv1d(x, wp, cout, 1, mode=1, x_split=xs)
    torch.cuda.synchronize()
