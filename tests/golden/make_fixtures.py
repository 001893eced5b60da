#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (SA-toolkit) in the
build container.  Run from the repo root:   python tests/golden/make_fixtures.py [--only NAME ...]

What it does
  1. copies /root/reference to a scratch dir (the reference wants to write conf.pt files) and puts
     the torchaudio stand-in tests/golden/refstub/ on the path (torchaudio is not installed here and
     cannot be fetched; the stand-in restates lowpass/highpass_biquad and hosts the wav2vec2
     factory from oracle/wav2vec2.py — those two pieces are third-party arithmetic and stay
     "parity unpinned");
  2. builds the reference model objects through the reference's own `build(args)` factories, loads
     the seeded synthetic state dict of satools_amd.synthetic into them (strict: this also pins
     the state-dict key names/shapes), calibrates BatchNorm statistics and the VQ codebook once and
     stores them as tests/golden/conditioning_*.npz;
  3. calls the reference's public API (fbank, extract_bn, get_f0, get_spk_id, UttCMVN,
     quantize_f0/awgn_f0, CoreHifiGan with hooks, convert) on seeded synthetic inputs and stores
     inputs-by-seed + expected outputs as small .npz/.json files.

Nothing of the reference's source is stored: fixtures are data only.  The reference never travels
to the GPU box; tests read only these fixtures.
"""
import argparse
import importlib.util
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
SCRATCH = os.environ.get("SAT_FIXTURE_SCRATCH", "/tmp/sat_fixture_ref")
REFSRC = "/root/reference"

F0_OPTS = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}


def setup_reference():
    ref = os.path.join(SCRATCH, "ref")
    if not os.path.exists(ref):
        os.makedirs(SCRATCH, exist_ok=True)
        shutil.copytree(REFSRC, ref)
        for d, _, fs in os.walk(ref):
            os.chmod(d, 0o755)
            for f in fs:
                os.chmod(os.path.join(d, f), 0o644)
    os.environ["SA_JIT_TWEAK"] = "true"
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(GOLD, "refstub"), os.path.join(ref, "satools"), ROOT]
    return ref


def exec_config(path):
    spec = importlib.util.spec_from_file_location("config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference_model(ref, asr_name, f0_transformation=""):
    """reference Net for tag hifigan_<asr_name>, with synthetic weights (un-conditioned)"""
    import satools
    import torch
    from satools_amd import infer_helper, synthetic

    conf = dict(infer_helper.asrbn_conf_from_name(synthetic.ASR_DIR.format(name=asr_name)))
    conf["install_path"] = ref
    d = f"{ref}/egs/asr/librispeech/exp/chain/{asr_name}"
    os.makedirs(d, exist_ok=True)
    torch.save(conf, f"{d}/conf.pt")
    cfg = exec_config(f"{ref}/egs/vc/libritts/local/tuning/hifigan.py")
    args = satools.utils.SimpleNamespace(asrbn_model=synthetic.ASR_DIR.format(name=asr_name),
                                         f0_transformation=f0_transformation)
    net = cfg.build(args)(utt2spk=synthetic.utt2spk())
    return net


def calibrate(net, asr_name, calib_wavs, vq_layer):
    """one forward with the BatchNorm modules in train mode / momentum 1 (so that running stats :=
    batch stats), then a codebook of 48 frames sampled from the VQ layer's own input"""
    import torch
    bx = net.bn_extractor
    bns = [(n, m) for n, m in bx.named_modules() if isinstance(m, torch.nn.BatchNorm1d)]
    for _, m in bns:
        m.train()
        m.momentum = 1.0
    zs = []
    layer = bx.tdnnfs[vq_layer]
    h = layer.tdnn.linearB.register_forward_hook(lambda m, i, o: zs.append(o.detach()))
    with torch.no_grad():
        bx.extract_bn(calib_wavs.clone())
    h.remove()
    for _, m in bns:
        m.eval()
        m.momentum = 0.1
    z = zs[0].reshape(-1, zs[0].shape[-1])
    sel = torch.randperm(z.shape[0], generator=torch.Generator().manual_seed(0))[:48]
    cb = z[sel].clone()
    layer.bottleneck_func.quant._embedding.weight.data.copy_(cb)
    cond = {}
    sd = bx.state_dict()
    for k, v in sd.items():
        used = k.startswith("tdnn1.") or (k.startswith("tdnnfs.") and int(k.split(".")[1]) <= vq_layer)
        if used and (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("_embedding.weight")):
            cond[k] = v.numpy().copy()
    np.savez(os.path.join(GOLD, f"conditioning_{asr_name}.npz"), **cond)
    return cond


def sub(t, step):
    return t[..., ::step].contiguous().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    want = lambda name: a.only is None or name in a.only
    ref = setup_reference()
    import torch
    torch.set_num_threads(8)
    import satools  # noqa: F401  (the reference)
    from satools_amd import synthetic

    asr_name = "bn_tdnnf_600h_vq_48"
    tag = "hifigan_" + asr_name + "_v1"
    net = build_reference_model(ref, asr_name)
    state, mine = synthetic.checkpoint(tag, conditioning=None)
    res = net.load_state_dict(state["base_model_state_dict"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(net.state_dict()) == list(mine.state_dict())
    net.eval()

    if want("keys"):
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in net.state_dict().items()],
                  open(os.path.join(GOLD, "state_dict_keys_fbank.json"), "w"))
    cond_path = os.path.join(GOLD, f"conditioning_{asr_name}.npz")
    if want("conditioning") or not os.path.exists(cond_path):
        calib = torch.cat([synthetic.harm_batch(range(100, 108), 32000),
                           synthetic.rand_batch(5, 2, 32000) * 2 - 1], 0)
        calibrate(net, asr_name, calib, vq_layer=20)
    # reload through the product's own synthetic path so fixtures == what tests will build
    state, _ = synthetic.checkpoint(tag)
    net.load_state_dict(state["base_model_state_dict"], strict=True)
    net.eval()
    spk = net.spk

    if want("spk"):
        x1 = torch.zeros(1, 400)
        tg = [spk[3], spk[200], spk[0]]
        json.dump({"spk": spk, "str_target": spk[5], "str_onehot_argmax": int(net.get_spk_id(x1, spk[5]).argmax()),
                   "list_target": tg, "list_onehot_argmax": net.get_spk_id(x1, tg).argmax(1).tolist(),
                   "shape_str": list(net.get_spk_id(x1, spk[5]).shape), "dtype": str(net.get_spk_id(x1, spk[5]).dtype)},
                  open(os.path.join(GOLD, "fx_spk.json"), "w"))

    if want("fbank"):
        out = {}
        for name, wav in [("harm0_8000", synthetic.harm_batch([0], 8000)),
                          ("harm01_16384", synthetic.harm_batch([0, 1], 16384)),
                          ("rand0_8000", synthetic.rand_batch(0, 1, 8000)),
                          ("harm0_80000", synthetic.harm_batch([0], 80000))]:
            out[name] = satools.kaldifeat.fbank(wav * 32768, num_mel_bins=80, snip_edges=False).numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_fbank.npz"), **out)

    if want("tdnnf"):
        out = {}
        for name, wav in [("harm0_8000", synthetic.harm_batch([0], 8000)), ("rand0_8000", synthetic.rand_batch(0, 1, 8000))]:
            acts = {}
            hooks = []
            bx = net.bn_extractor
            mods = [("tdnn1", bx.tdnn1)] + [(f"tdnnfs.{i}", bx.tdnnfs[i]) for i in range(0, 20, 2)]
            # extract_bn calls `t.forward(x)` directly (no module hooks fire on the layer itself): hook the
            # inner BatchNorm1d ([N, C, T], pre-ReLU) and finish the layer here
            for n, m in mods:
                hooks.append(m.bn.register_forward_hook(
                    lambda mod, i, o, n=n: acts.__setitem__(n, torch.relu(o.detach()).permute(0, 2, 1))))
            hooks.append(bx.tdnnfs[20].tdnn.linearB.register_forward_hook(lambda m, i, o: acts.__setitem__("z", o.detach())))
            hooks.append(bx.tdnnfs[20].bottleneck_func.quant.register_forward_hook(
                lambda m, i, o: acts.update(idx=o[5].detach(), dist=o[4].detach())))
            with torch.no_grad():
                bn = net.get_bn(wav)
            for h in hooks:
                h.remove()
            for n, _ in mods:
                out[f"{name}/{n}"] = acts[n][..., ::16].numpy()      # [1, T', 64]: every 16th channel
            out[f"{name}/z"] = acts["z"].numpy()
            out[f"{name}/idx"] = acts["idx"].reshape(-1).numpy()
            out[f"{name}/dist"] = acts["dist"].numpy()
            out[f"{name}/bn"] = bn.numpy()
        wav = synthetic.harm_batch([0, 1], 80000)
        acts = {}
        h = net.bn_extractor.tdnnfs[20].bottleneck_func.quant.register_forward_hook(
            lambda m, i, o: acts.update(idx=o[5].detach(), dist=o[4].detach()))
        with torch.no_grad():
            bn = net.get_bn(wav)
        h.remove()
        srt = acts["dist"].sort(1)[0]
        out["harm01_80000/idx"] = acts["idx"].reshape(2, -1).numpy()
        out["harm01_80000/margin"] = (srt[:, 1] - srt[:, 0]).reshape(2, -1).numpy()
        out["harm01_80000/bn_sub"] = bn[:, ::8, :].numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_tdnnf.npz"), **out)

    if want("asr"):
        # the ASR half: Net.forward up to the chain / xent outputs (tdnnf_vq.py:259-284)
        out = {}
        bx = net.bn_extractor
        for name, wav in [("harm0_16000", synthetic.harm_batch([0], 16000)), ("harm01_32000", synthetic.harm_batch([0, 1], 32000))]:
            acts = {}
            hk = [bx.tdnnfs_after[0].register_forward_hook(lambda m, i, o: acts.__setitem__("after0", o.detach())),
                  bx.tdnnfs_after[6].register_forward_hook(lambda m, i, o: acts.__setitem__("after6", o.detach())),
                  bx.tdnnfs[20].register_forward_hook(lambda m, i, o: acts.__setitem__("vq_layer", o.detach()))]
            with torch.no_grad():
                chain, xent = bx(wav.clone())
            for h in hk:
                h.remove()
            out[f"{name}/chain_sub"] = chain[..., ::8].numpy()
            out[f"{name}/xent_sub"] = xent[..., ::8].numpy()
            out[f"{name}/xent_lse"] = torch.logsumexp(xent, dim=2).numpy()
            out[f"{name}/after0_sub"] = acts["after0"][..., ::16].numpy()
            out[f"{name}/after6_sub"] = acts["after6"][..., ::16].numpy()
            out[f"{name}/vq_layer_sub"] = acts["vq_layer"][..., ::16].numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_asr.npz"), **out)
        print({k: v.shape for k, v in out.items()})

    if want("f0"):
        out = {}
        for n in (8000, 16384, 80000):
            for s in range(4):
                out[f"harm{s}_{n}"] = net.get_f0(synthetic.harm_batch([s], n)).numpy()
            out[f"rand0_{n}"] = net.get_f0(synthetic.rand_batch(0, 1, n)).numpy()
        out["harm01_80000_batch"] = net.get_f0(synthetic.harm_batch([0, 1], 80000)).numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_f0.npz"), **out)

    if want("f0norm"):
        f0a = net.get_f0(synthetic.harm_batch([0], 16384))      # [1, 52]
        f0b = net.get_f0(synthetic.harm_batch([0, 1], 16384))   # [2, 52]
        out = {"in_1xT": f0a.numpy().copy(), "in_2xT": f0b.numpy().copy()}
        norm = satools.cmvn.UttCMVN(var_norm=True, keep_zeros=True)
        out["out_1xT"] = norm(f0a.clone()).numpy()
        out["out_2xT"] = norm(f0b.clone()).numpy()
        out["out_1x2xT"] = norm(f0b.clone().unsqueeze(0)).numpy()
        z = f0b.clone()
        z[1] = 0
        out["in_zero_row"] = z.numpy().copy()
        out["out_zero_row"] = norm(z.clone()).numpy()
        nf = norm(f0b.clone()).unsqueeze(0).permute(1, 0, 2)    # [2, 1, T]
        out["quant16"] = satools.hifigan.nn.quantize_f0(nf.clone(), num_bins="quant_16_awgn_2").numpy()
        torch.manual_seed(1234)
        q = satools.hifigan.nn.quantize_f0(nf.clone(), num_bins="quant_16_awgn_2")
        out["quant16_awgn2_seed1234"] = satools.hifigan.nn.awgn_f0(q, target_noise_db="quant_16_awgn_2").numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_f0norm.npz"), **out)

    if want("gen"):
        wav = synthetic.harm_batch([0], 8000)
        with torch.no_grad():
            f0, bn, spk_id = net.extract_features(wav, spk[3])
        acts = {}
        hooks = []
        g = net.hifigan
        mods = [("conv_pre", g.conv_pre)] + [(f"ups.{i}", g.ups[i]) for i in range(5)] + \
               [(f"resblocks.{i}", g.resblocks[i]) for i in range(15)]
        for n, m in mods:
            hooks.append(m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach().clone())))
        f0_raw = f0.clone()
        with torch.no_grad():
            y = net._forward(f0, bn, spk_id)
        for h in hooks:
            h.remove()
        out = {"f0_raw": f0_raw.numpy(), "f0_after": f0.numpy(), "bn": bn.numpy(), "spk_argmax": spk_id.argmax(1).numpy(),
               "y": y.numpy(), "conv_pre": acts["conv_pre"].numpy(), "ups.0": acts["ups.0"].numpy()}
        for i in range(3):
            out[f"resblocks.{i}"] = acts[f"resblocks.{i}"].numpy()
        for st, step in zip(range(1, 5), (4, 16, 32, 64)):
            out[f"ups.{st}_sub{step}"] = sub(acts[f"ups.{st}"], step)
            for j in range(3):
                out[f"resblocks.{3 * st + j}_sub{step}"] = sub(acts[f"resblocks.{3 * st + j}"], step)
        np.savez_compressed(os.path.join(GOLD, "fx_gen.npz"), **out)

    if want("e2e"):
        out = {}
        shapes = {}
        with torch.no_grad():
            w = synthetic.harm_batch([0], 80000)
            out["harm0_80000_str"] = net.convert(w, target=spk[3]).numpy()
            w = synthetic.harm_batch([0, 1], 80000)
            out["harm01_80000_list"] = net.convert(w, target=[spk[3], spk[10]]).numpy()
            w = synthetic.rand_batch(0, 1, 16000)
            wc = w.clone()
            out["rand0_16000_str"] = net.convert(w, target=spk[7]).numpy()
            assert torch.equal(w, wc), "convert must not mutate its input"
            for n in (8000, 16000, 8192, 16640, 16384):
                w = synthetic.harm_batch([2], n)
                shapes[str(n)] = [list(net.get_bn(w).shape), list(net.get_f0(w).shape),
                                  list(net.convert(w, target=spk[1]).shape)]
        # option f0-transformation=quant_16_awgn_2 (a second Net built with that arg)
        net2 = build_reference_model(ref, asr_name, f0_transformation="quant_16_awgn_2")
        net2.load_state_dict(state["base_model_state_dict"], strict=True)
        net2.eval()
        with torch.no_grad():
            w = synthetic.harm_batch([0, 1], 16000)
            torch.manual_seed(1234)
            out["harm01_16000_quant16_awgn2_seed1234"] = net2.convert(w, target=[spk[3], spk[10]]).numpy()
        np.savez_compressed(os.path.join(GOLD, "fx_e2e.npz"), **out)
        json.dump(shapes, open(os.path.join(GOLD, "fx_shapes.json"), "w"))
    if want("meanrev"):
        # option f0-transformation=mean-reverv_<alpha>:<n> (hifigan/nn.py:65-90, dispatched at hifigan.py:79-80)
        out = {}
        norm = satools.cmvn.UttCMVN(var_norm=True, keep_zeros=True)
        f0a = norm(net.get_f0(synthetic.harm_batch([0], 16384)).clone()).unsqueeze(0).permute(1, 0, 2)    # [1, 1, 52]
        f0c = norm(net.get_f0(synthetic.harm_batch([2], 80000)).clone()).unsqueeze(0).permute(1, 0, 2)    # [1, 1, 250]
        out["in_T52"], out["in_T250"] = f0a.numpy().copy(), f0c.numpy().copy()
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):          # the reference prints the shape
            for spec in ("mean-reverv_0.5:32", "mean-reverv_0.3:7", "mean-reverv_1:4"):
                out[f"T52/{spec}"] = satools.hifigan.nn.mean_reverv_f0(f0a.clone(), alpha=spec).numpy()
                out[f"T250/{spec}"] = satools.hifigan.nn.mean_reverv_f0(f0c.clone(), alpha=spec).numpy()
            net3 = build_reference_model(ref, asr_name, f0_transformation="mean-reverv_0.5:32")
            net3.load_state_dict(state["base_model_state_dict"], strict=True)
            net3.eval()
            with torch.no_grad():
                out["harm0_16000_meanrev_0.5_32"] = net3.convert(synthetic.harm_batch([0], 16000), target=spk[3]).numpy()
                try:
                    net3.convert(synthetic.harm_batch([0, 1], 16000), target=[spk[3], spk[10]])
                    batch_error = ""
                except Exception as e:      # batches fail inside conv1d (the squeezed [B, T] input is read as B channels)
                    batch_error = type(e).__name__
        np.savez_compressed(os.path.join(GOLD, "fx_meanrev.npz"), **out)
        json.dump({"batch_of_2_raises": batch_error}, open(os.path.join(GOLD, "fx_meanrev.json"), "w"))
        print("mean-reverv: batch of 2 raises", batch_error)

    if want("w2v2"):
        # wav2vec2 tag: the reference's own tdnnf_wav2vec2_vq.Net / hifigan Net with the torchaudio stand-in's
        # wav2vec2 factory (= oracle/wav2vec2.py).  Pins the plumbing around the third-party model
        # (extract_features(x)[0][-1], replicate pad, pad_input(3), TDNNF tail, VQ), not torchaudio's arithmetic.
        import torchaudio
        from oracle import wav2vec2 as ow
        torchaudio.models.wav2vec2.model._factory = ow.build_wav2vec2
        name2 = "bn_tdnnf_wav2vec2_vq_48"
        tag2 = "hifigan_" + name2 + "_v1"
        net2 = build_reference_model(ref, name2)
        st2, mine2 = synthetic.checkpoint(tag2, conditioning=None)
        res = net2.load_state_dict(st2["base_model_state_dict"], strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        assert list(net2.state_dict()) == list(mine2.state_dict())
        net2.eval()
        json.dump([[k, list(v.shape), str(v.dtype)] for k, v in net2.state_dict().items()],
                  open(os.path.join(GOLD, "state_dict_keys_w2v2.json"), "w"))
        calib = torch.cat([synthetic.harm_batch(range(100, 104), 32000), synthetic.rand_batch(5, 1, 32000) * 2 - 1], 0)
        calibrate(net2, name2, calib, vq_layer=2)
        st2, _ = synthetic.checkpoint(tag2)
        net2.load_state_dict(st2["base_model_state_dict"], strict=True)
        net2.eval()
        out = {}
        acts = {}
        h = net2.bn_extractor.tdnnfs[2].bottleneck_func.quant.register_forward_hook(
            lambda m, i, o: acts.update(idx=o[5].detach(), dist=o[4].detach()))
        with torch.no_grad():
            w = synthetic.harm_batch([0, 1], 16000)
            bn = net2.get_bn(w)
            out["harm01_16000/bn"] = bn.numpy()
            out["harm01_16000/idx"] = acts["idx"].reshape(2, -1).numpy()
            srt = acts["dist"].sort(1)[0]
            out["harm01_16000/margin"] = (srt[:, 1] - srt[:, 0]).reshape(2, -1).numpy()
            feats = net2.bn_extractor.preprocessor.extract_features(w)[0][-1]
            out["harm01_16000/w2v2_last_sub"] = feats[:, :, ::16].numpy()
            out["harm0_16000/convert"] = net2.convert(synthetic.harm_batch([0], 16000), target=net2.spk[3]).numpy()
            # the ASR half of the wav2vec2-tag net (Net.forward, tdnnf_wav2vec2_vq.py:316-345)
            bx2 = net2.bn_extractor
            acts2 = {}
            hk = [bx2.tdnnfs_after[0].register_forward_hook(lambda m, i, o: acts2.__setitem__("after0", o.detach())),
                  bx2.tdnnfs[2].register_forward_hook(lambda m, i, o: acts2.__setitem__("vq_layer", o.detach()))]
            chain, xent = bx2(w.clone())
            for h_ in hk:
                h_.remove()
            out["harm01_16000/chain_sub"] = chain[..., ::8].numpy()
            out["harm01_16000/xent_sub"] = xent[..., ::8].numpy()
            out["harm01_16000/xent_lse"] = torch.logsumexp(xent, dim=2).numpy()
            out["harm01_16000/after0_sub"] = acts2["after0"][..., ::16].numpy()
            out["harm01_16000/vq_layer_sub"] = acts2["vq_layer"][..., ::16].numpy()
            # configs[3]: the wav2vec2 tag with option f0-transformation=quant_16_awgn_2 (a second Net built with that arg)
            net4 = build_reference_model(ref, name2, f0_transformation="quant_16_awgn_2")
            net4.load_state_dict(st2["base_model_state_dict"], strict=True)
            net4.eval()
            torch.manual_seed(1234)
            out["harm01_16000/convert_quant16_awgn2_seed1234"] = net4.convert(w.clone(), target=[net4.spk[3], net4.spk[10]]).numpy()
            del net4
            shapes2 = {"forward_2x32000": list(bx2(torch.arange(2 * 32000, dtype=torch.float32).reshape(2, 32000) / 64000.0)[0].shape)}
            for n in (16000, 32000, 80000):
                ww = synthetic.harm_batch([2], n)
                shapes2[str(n)] = [list(net2.get_bn(ww).shape), list(net2.get_f0(ww).shape)]
        h.remove()
        np.savez_compressed(os.path.join(GOLD, "fx_w2v2.npz"), **out)
        json.dump(shapes2, open(os.path.join(GOLD, "fx_shapes_w2v2.json"), "w"))
    if want("w2v2_long"):
        # the wav2vec2 tag on ONE 20 s utterance (999 frames: attention over more than one 256-key block, the long path of
        # the GPU tests) through the reference's own Net: VQ indices, the margin of every decision, a channel subsample of the
        # features, and every 4th sample of convert()'s waveform (F0 by the reference's own get_f0)
        import torchaudio
        from oracle import wav2vec2 as ow
        torchaudio.models.wav2vec2.model._factory = ow.build_wav2vec2
        name2 = "bn_tdnnf_wav2vec2_vq_48"
        tag2 = "hifigan_" + name2 + "_v1"
        net2 = build_reference_model(ref, name2)
        st2, _ = synthetic.checkpoint(tag2)
        net2.load_state_dict(st2["base_model_state_dict"], strict=True)
        net2.eval()
        n = 20 * 16000
        seeds = [3 + 7 * j for j in range(4)]            # = tests/test_hip_robust.py: _long_batch([3], n)
        w = torch.cat([synthetic.harm_batch([sd], 80000)[0] for sd in seeds])[:n].unsqueeze(0)
        acts = {}
        h = net2.bn_extractor.tdnnfs[2].bottleneck_func.quant.register_forward_hook(
            lambda m, i, o: acts.update(idx=o[5].detach(), dist=o[4].detach()))
        out = {"seeds": np.array(seeds), "n": np.array(n)}
        with torch.no_grad():
            bn = net2.get_bn(w)
            out["idx"] = acts["idx"].reshape(1, -1).numpy()
            srt = acts["dist"].sort(1)[0]
            out["d1"] = srt[:, 0].reshape(1, -1).numpy()
            out["d2"] = srt[:, 1].reshape(1, -1).numpy()
            out["bn_sub"] = bn[:, :, ::16].numpy()
            out["w2v2_last_sub"] = net2.bn_extractor.preprocessor.extract_features(w)[0][-1][:, :, ::16].numpy()
            out["f0"] = net2.get_f0(w).numpy()
            out["convert_every4"] = net2.convert(w.clone(), target=net2.spk[5]).numpy()[..., ::4]
        h.remove()
        np.savez_compressed(os.path.join(GOLD, "fx_w2v2_long.npz"), **out)
        print("w2v2_long:", {k: v.shape for k, v in out.items()})
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    main()
