import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import satools_amd
from satools_amd import ops, synthetic, f0 as f0_hip
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); model.to("cuda"); model.eval()
B, n = 32, 80000
f32 = synthetic.harm_batch(list(range(B)), n).clone()
tg = synthetic.targets(model.spk, list(range(B)))
pins = [torch.empty(B, n, dtype=torch.float32, pin_memory=True) for _ in range(3)]
lens = [n] * B
d32 = f32.cuda()
side = torch.cuda.Stream()
def loop(mode, steps=10):
    acc = {}
    def tick(name, t0):
        t1 = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t1 - t0); return t1
    with torch.no_grad():
        for it in range(steps + 3):
            if it == 3:
                torch.cuda.synchronize(); acc.clear(); t00 = time.perf_counter()
            t = time.perf_counter()
            if mode == "h2d":
                pin = pins[it % 3]
                pin.copy_(f32); t = tick("host copy", t)
                x = pin.to("cuda", non_blocking=True); t = tick("H2D enqueue", t)
            elif mode == "h2d_np":
                pin = pins[it % 3]
                np.copyto(pin.numpy(), f32.numpy()); t = tick("host copy (numpy)", t)
                x = pin.to("cuda", non_blocking=True); t = tick("H2D enqueue", t)
            else:
                x = d32
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                f0, st = f0_hip.yaapt_ragged(x, lens, model.f0_yaapt_opts, defer_status=True)
                f0 = f0.unsqueeze(0)
            x.record_stream(side); t = tick("yaapt launches", t)
            bn = model.get_bn(x); t = tick("get_bn launches", t)
            cur.wait_stream(side)
            y = model._forward(f0, bn, model.get_spk_id(x, tg)); t = tick("generator launches", t)
            st.check(); t = tick("status wait", t)
        th = time.perf_counter() - t00
        torch.cuda.synchronize()
    print(f"{mode:10s} {(time.perf_counter() - t00) / steps * 1e3:7.2f} ms per batch | " + "  ".join(f"{k} {v / steps * 1e3:.2f}" for k, v in acc.items()), flush=True)
for m in ("resident", "h2d", "h2d_np", "resident", "h2d"):
    loop(m)
torch.set_num_threads(1)
print("torch threads 1")
for m in ("h2d", "resident"):
    loop(m)
