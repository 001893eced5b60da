import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import satools_amd
from satools_amd import synthetic
m = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1"); m.to("cuda"); m.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).cuda(); tg = synthetic.targets(m.spk, seeds)
for _ in range(3): m.convert(wav, target=tg)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); y = m.convert(wav, target=tg); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
# breakdown
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5): m.convert(wav, target=tg)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
