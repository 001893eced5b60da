"""where does the wav2vec2 tag leave the oracle on long utterances?  python tests/diagnostics/w2v2_long_diff.py [seconds ...]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import satools_amd
from satools_amd import synthetic
from oracle import tdnnf as otd, wav2vec2 as ow
from test_hip_robust import _long_batch

TAG = "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"
model = satools_amd.load_model("synthetic:" + TAG); model.to("cuda"); model.eval()
state, _ = synthetic.checkpoint(TAG)
sd = state["base_model_state_dict"]
om = ow.Wav2Vec2Restated(24)
pfx = "bn_extractor.preprocessor."
om.load_state_dict({k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}); om.eval()
asr = {k[len("bn_extractor."):]: v for k, v in sd.items() if k.startswith("bn_extractor.")}
for sec in [int(a) for a in sys.argv[1:]] or [20, 35]:
    wav = _long_batch([3, 4], sec * 16000)
    feats = {}
    aux = {}
    ref_bn = otd.extract_bn_w2v2(asr, wav, aux=aux, hook=lambda k, v: feats.__setitem__(k, v.clone()), model=om)
    ext = model.bn_extractor
    bn, (z, idx, dist) = ext.extract_bn(wav.clone().to("cuda"), want_aux=True)
    y = ext.w2v2_features(wav.to("cuda")).cpu().permute(0, 2, 1)        # [B, T, 1024]
    r = feats["w2v2"]
    d = (y - r).abs()
    print(f"{sec} s: frames {r.shape[1]}; last layer max abs diff {d.max():.3e} (values up to {r.abs().max():.1f}), rel RMS {float((y - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()):.3e}")
    per_t = d.amax(dim=(0, 2))
    worst = torch.topk(per_t, 5)
    print("   worst frames", worst.indices.tolist(), [f"{v:.2e}" for v in worst.values.tolist()])
    keys = [k for k in aux if "idx" in k or "indices" in k]
    print("   aux keys", list(aux)[:12])
    ridx = None
    for k in aux:
        if "idx" in k:
            ridx = aux[k]
    if ridx is not None:
        agree = (idx.cpu().long().flatten() == ridx.long().flatten())
        print(f"   VQ indices agree: {int(agree.sum())}/{agree.numel()}; first mismatches {torch.nonzero(~agree).flatten()[:8].tolist()}")
    print(f"   bn max abs diff {(bn.cpu() - ref_bn).abs().max():.3e}")
