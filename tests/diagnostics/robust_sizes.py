import sys, torch, traceback
sys.path.insert(0, ".")
import satools_amd
from satools_amd import synthetic
model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); model.to("cuda"); model.eval()
for B, n in ((1, 1600), (1, 3200), (2, 6400), (1, 200000), (64, 16000), (3, 80001), (1, 639), (33, 48000)):
    try:
        wav = synthetic.harm_batch(list(range(B)), n).to("cuda")
        y = model.convert(wav, target=synthetic.targets(model.spk, list(range(B))) if B > 1 else model.spk[0])
        torch.cuda.synchronize()
        print(f"B={B} n={n}: out {tuple(y.shape)} finite={bool(torch.isfinite(y).all())} rms={float(y.pow(2).mean().sqrt()):.4f}")
    except Exception as e:
        print(f"B={B} n={n}: {type(e).__name__}: {str(e)[:160]}")
