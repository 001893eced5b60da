import sys, torch
sys.path.insert(0, ".")
import satools_amd
from satools_amd import synthetic, ops
from satools_amd.wav2vec2 import CONV_LAYERS
from oracle import wav2vec2 as ow
import torch.nn.functional as F
tag = "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"
state, _ = synthetic.checkpoint(tag)
sd = {k[len("bn_extractor.preprocessor."):]: v for k, v in state["base_model_state_dict"].items() if k.startswith("bn_extractor.preprocessor.")}
m = ow.Wav2Vec2Restated(24); m.load_state_dict(sd); m.eval()
model = satools_amd.load_model("synthetic:" + tag); model.to("cuda"); model.eval()
ext = model.bn_extractor
wav = synthetic.harm_batch([0, 1], 16000)
def cmp(name, a, b):
    print(f"{name:28s} shape {tuple(a.shape)} max|ref| {b.abs().max().item():.3e} maxerr {(a.cpu()-b).abs().max().item():.3e}")
with torch.no_grad():
    # CPU stages
    x = wav.unsqueeze(1); fe_out = []
    for blk in m.feature_extractor.conv_layers:
        x = blk(x); fe_out.append(x)
    proj = m.encoder.feature_projection(x.transpose(1, 2))
    tr = m.encoder.transformer
    xpos = proj + tr.pos_conv_embed(proj)
    xln = tr.layer_norm(xpos)
    l0 = tr.layers[0](xln)
    # GPU stages
    W = ext._prepare_w2v2(torch.device("cuda"))
    lens, t = [], 16000
    for _, k, s in CONV_LAYERS:
        t = (t - k) // s + 1; lens.append(t)
    g = ops.w2v2_conv0(wav.cuda(), W["fe"][0]["w"], W["fe"][0]["b"])
    for i in range(7):
        e = W["fe"][i]; last = i == 6
        gl = ops.layernorm_ch(g, e["g"], e["beta"], gelu=True, split_phases=False)
        cmp(f"fe[{i}] ln+gelu", gl, fe_out[i])
        if not last:
            gs = ops.layernorm_ch(g, e["g"], e["beta"], gelu=True, split_phases=True)
            nxt = W["fe"][i + 1]
            g = ops.conv1d(gs, nxt["w"], 512, nxt["k"], bias=nxt["b"], pad_left=0, pad_right=0, t_out=lens[i + 1])
    x = ops.layernorm_ch(gl, W["fp"]["g"], W["fp"]["beta"])
    x = ops.conv1d(x, W["fp"]["w"], 1024, 1, bias=W["fp"]["b"])
    cmp("projection", x, proj.transpose(1, 2))
    xp = ops.conv1d(x, W["pos"]["w"], 1024, 128, bias=W["pos"]["b"], pad_left=64, pad_right=63, groups=16, gelu=True, post_res=x)
    cmp("pos conv add", xp, xpos.transpose(1, 2))
    xl = ops.layernorm_ch(xp, W["ln"]["g"], W["ln"]["beta"])
    cmp("encoder LN", xl, xln.transpose(1, 2))
    ext._fe_len = lens
    y = ext.w2v2_features(wav.cuda())
    outs = m.extract_features(wav)[0]
    cmp("last layer", y, outs[-1].transpose(1, 2))
