import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import satools_amd

def t(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3

print("torch threads", torch.get_num_threads(), "cpus", os.cpu_count())
audio = torch.randn(32, 80000)
pin = torch.empty(audio.numel(), dtype=torch.float32, pin_memory=True).view(audio.shape)
print("pin.copy_(audio)            %.2f ms" % t(lambda: pin.copy_(audio)))
pn = pin.numpy(); an = audio.numpy()
print("np.copyto(pin, audio)       %.2f ms" % t(lambda: np.copyto(pn, an)))
def h2d_pin():
    x = pin.to("cuda", non_blocking=True); torch.cuda.synchronize()
print("pinned H2D only             %.2f ms" % t(h2d_pin))
def h2d_pageable():
    x = audio.to("cuda"); torch.cuda.synchronize()
print("pageable H2D                %.2f ms" % t(h2d_pageable))
print("torch.zeros([32, 80000])    %.2f ms" % t(lambda: torch.zeros([32, 80000])))
out = torch.zeros([32, 80000]); a = torch.randn(1, 80000)
def fill():
    for i in range(32): out[i, :80000] = a.squeeze()
print("32 row assignments          %.2f ms" % t(fill))
for nt in (1, 8):
    torch.set_num_threads(nt)
    print(f"threads={nt}: pin.copy_ %.2f ms, zeros %.2f ms, 32 rows %.2f ms" % (t(lambda: pin.copy_(audio)), t(lambda: torch.zeros([32, 80000])), t(fill)))
