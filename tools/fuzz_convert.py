"""random batch sizes and lengths through convert(): the default arithmetic (split-f16, 8-bit cross terms where the ring kernel serves the
batch) against the exact-f32 kernels of the same model — a cross-check of the dispatch (which kernel serves which shape) over shapes the
tests do not name.  python tools/fuzz_convert.py [n] [seed] [tag]"""
import os, random, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import torch
import satools_amd
from satools_amd import synthetic
from test_hip_f8r import gen_precision

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 0
tag = sys.argv[3] if len(sys.argv) > 3 else "hifigan_bn_tdnnf_600h_vq_48_v1"
rng = random.Random(seed)
model = satools_amd.load_model("synthetic:" + tag); model.to("cuda"); model.eval()
ext = model.bn_extractor
worst, bad, flips, frames, raw_flips = 0.0, [], 0, 0, 0
for i in range(n):
    B = rng.choice([1, 1, 2, 3, 5, 8, 13, 24, 32, 40])
    nsamp = rng.choice([rng.randint(4000, 12000), rng.randint(12000, 40000), rng.randint(40000, 90000), 320 * rng.randint(20, 250), 16000 * rng.randint(1, 5)])
    cap = (16 if "wav2vec2" in tag else 48) * 80000
    if B * nsamp > cap:
        B = max(1, cap // nsamp)
    wav = synthetic.harm_batch([rng.randint(0, 10 ** 6) for _ in range(B)], nsamp).to("cuda")
    tg = synthetic.targets(model.spk, [rng.randint(0, 10 ** 6) for _ in range(B)])
    try:
        with torch.no_grad():
            f0 = model.get_f0(wav)
            idx_raw = ext.extract_bn(wav.clone(), want_aux=True)[1][1]          # the arithmetic as configured, no second decision
            idx = ext.vq_indices(wav)[0]                                       # what the extractor delivers (near-tie guard, round 6)
            model.set_f0(f0.clone())
            y = model.convert(wav, target=tg)
            keys = [k for k in ("precision", "w2v2_precision") if hasattr(ext, k)]
            keep = {k: getattr(ext, k) for k in keys}
            try:
                for k in keys:
                    setattr(ext, k, "f32")
                idx32 = ext.extract_bn(wav.clone(), want_aux=True)[1][1]
                with gen_precision(model.hifigan, "f32"):
                    model.set_f0(f0.clone())
                    y32 = model.convert(wav, target=tg)
            finally:
                for k, v in keep.items():
                    setattr(ext, k, v)
        same = bool(torch.equal(idx, idx32))
        d = (y - y32).double()
        rms = float(d.pow(2).mean().sqrt())
        frames += idx.numel()
        raw_flips += int((idx_raw != idx32).sum())
        if not same:
            flips += int((idx != idx32).sum())
        elif rms > 2e-5 or not torch.isfinite(y).all():
            bad.append((B, nsamp, rms))
            print("FAIL", B, nsamp, rms, flush=True)
        if same:
            worst = max(worst, rms)
    except Exception as e:      # noqa: BLE001
        bad.append((B, nsamp, repr(e)[:200]))
        print("FAIL", B, nsamp, repr(e)[:200], flush=True)
    if i % 10 == 9:
        print(f"{i + 1} shapes, {len(bad)} failures, worst rms {worst:.2e}, {flips} delivered / {raw_flips} raw frames with another VQ index of {frames}; last B {B} n {nsamp}", flush=True)
print(f"fuzz_convert: {n} shapes (seed {seed}), {len(bad)} failures, worst rms vs the exact-f32 kernels {worst:.2e} (equal VQ indices), "
      f"{flips} of {frames} delivered VQ indices differ from the exact kernels' (the raw split-f16 arithmetic: {raw_flips})")
sys.exit(1 if bad else 0)
