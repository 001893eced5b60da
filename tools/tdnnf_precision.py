import sys, torch, numpy as np, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import synthetic, asrbn
fx = np.load("tests/golden/fx_tdnnf.npz")
wav = synthetic.harm_batch([0, 1], 80000)
big = synthetic.harm_batch(list(range(32)), 80000).cuda()
res = {}
for prec in ("f32", "f16x3"):
    asrbn._TdnnfBase.precision = prec
    m = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); m.to("cuda"); m.eval()
    bn, (z, idx, dist) = m.bn_extractor.extract_bn(wav.clone().cuda(), want_aux=True)
    agree = (idx.cpu().long() == torch.from_numpy(fx["harm01_80000/idx"]).long()).float().mean().item()
    res[prec] = z.cpu()
    for _ in range(2): m.get_bn(big)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): m.get_bn(big)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(prec, "VQ agreement vs reference", agree, "get_bn(32 utt) ms", round(dt * 1e3, 2))
print("z max abs diff f16x3 vs f32:", (res["f32"] - res["f16x3"]).abs().max().item(), "z scale", res["f32"].abs().max().item())
