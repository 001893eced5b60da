"""stage-by-stage comparison of the HIP YAAPT with the CPU oracle on named utterances:
python tests/diagnostics/yaapt_stage_diff.py rand107 rand117 ..."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import f0 as f0_hip, synthetic
from oracle import yaapt as oy

OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
torch.set_num_threads(1)
for name in sys.argv[1:]:
    kind, seed = name[:4], int(name[4:])
    wav = synthetic.harm_batch([seed], 80000) if kind == "harm" else synthetic.rand_batch(seed, 1, 80000)
    aux = {}
    ref = oy.yaapt_one(wav[0], OPTS, aux=aux).numpy()
    got, g = f0_hip.yaapt(wav.to("cuda"), OPTS, return_aux=True)
    got = got.cpu().numpy()[0]
    g = {k: v.cpu().numpy()[0] for k, v in g.items()}
    nf = ref.shape[0]
    print(f"== {name}: final differs on {int((got != ref).sum())}/{nf} frames; first {np.flatnonzero(got != ref)[:6].tolist()}")
    L = aux["filt"].numel()
    for i, key in enumerate(("filt", "filt2")):
        d = np.abs(g["filt"][i][:L] - aux[key].numpy())
        print(f"  {key}: max abs diff {d.max():.3e}  ({int((d > 0).sum())} samples differ)")
    e = aux["energy"].numpy()
    print(f"  energy: max abs diff {np.abs(g['energy'] - e).max():.3e}; vuv differs on {int((g['vuv'].astype(bool) != aux['vuv'].numpy()).sum())} frames; voiced {int(aux['vuv'].sum())}")
    cp, cm = aux["cand_pitch"].numpy(), aux["cand_merit"].numpy()
    print(f"  cand_pitch differs: {int((g['cand'][:4] != cp).sum())}, cand_merit max diff {np.abs(g['cand'][4:8] - cm).max():.3e}; frames with a candidate {int((cp[0] > 0).sum())}")
    sp = aux["spec_pitch"].numpy()
    print(f"  spec_pitch: max abs diff {np.abs(g['spec_pitch'] - sp).max():.3e} ({int((g['spec_pitch'] != sp).sum())} frames)  oracle pitch_std {float(aux['pitch_std']):.6f}  gpu scal {g['scal'].tolist()}")
    for key, idx in (("tp1", 0), ("tp2", 1)):
        t = aux[key].numpy()
        print(f"  {key}[0] differs on {int((g['tp'][idx] != t[0][:nf]).sum())} frames; merit max diff {np.abs(g['tm'][idx] - aux['tm' + key[2]].numpy()[0][:nf]).max():.3e}")
    if (got != ref).any():
        i = int(np.flatnonzero(got != ref)[0])
        print(f"  first differing frame {i}: gpu {got[i]} oracle {ref[i]}; spec_pitch gpu {g['spec_pitch'][i]} oracle {sp[i]}; energy {e[i]} vuv {bool(aux['vuv'][i])}")
        print("  oracle final[:12]", ref[:12].tolist())
        print("  gpu    final[:12]", got[:12].tolist())
