"""what grouping the steps of several batches by PHASE could give: the front ends (YAAPT + bottleneck extractor) of four batches together on
four streams, then the four generators — against four convert() calls in flight (each its own front end, then its generator)"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import satools_amd
from satools_amd import synthetic

model = satools_amd.load_model("synthetic:hifigan_bn_tdnnf_600h_vq_48_v1")
model.to("cuda")
model.eval()
seeds = list(range(32))
wav = synthetic.harm_batch(seeds).to("cuda")
tg = synthetic.targets(model.spk, seeds)
streams = [torch.cuda.Stream() for _ in range(4)]
NG = int(sys.argv[1]) if len(sys.argv) > 1 else 4


def front(i):
    # the first half of Net.convert(defer_status=True): YAAPT on its side stream, the extractor, nothing awaited
    with torch.cuda.stream(streams[i % 4]):
        model._defer_f0_status, model._f0_status, model._bn_fix = True, None, None
        try:
            f0, bn, spk = model.extract_features(wav, tg)
        finally:
            model._defer_f0_status = False
        fix, model._bn_fix = model._bn_fix, None
        st, model._f0_status = model._f0_status, None
    return f0, bn, spk, fix, st


def gen(i, ctx):
    with torch.cuda.stream(streams[i % 4]):
        f0, bn, spk, fix, st = ctx
        model._keep_ctx = fix is not None
        try:
            y = model._forward(f0, bn, spk)
        finally:
            model._keep_ctx = False
        return model._finish(y, st, fix, True)


def timed(f, n):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def fronts_only():
    return [front(i) for i in range(NG)]


ctxs = fronts_only()
torch.cuda.synchronize()


def gens_only():
    with torch.no_grad():
        return [model._forward(ctxs[i][0].clone(), ctxs[i][1], ctxs[i][2]) if False else gen_plain(i) for i in range(NG)]


def gen_plain(i):
    with torch.cuda.stream(streams[i % 4]):
        return model._forward(ctxs[i][0], ctxs[i][1], ctxs[i][2])


def phased():
    # front ends of the group together, then (every stream waits for every front end) the generators together
    cs = [front(i) for i in range(NG)]
    evs = []
    for i in range(NG):
        e = torch.cuda.Event()
        e.record(streams[i % 4])
        evs.append(e)
    for i in range(NG):
        for e in evs:
            streams[i % 4].wait_event(e)
    ys = [gen(i, cs[i]) for i in range(NG)]
    for _, st in ys:
        st.start()
    evs = []
    for i in range(NG):
        e = torch.cuda.Event()
        e.record(streams[i % 4])
        evs.append(e)
    for i in range(NG):
        for e in evs:
            streams[i % 4].wait_event(e)
    for _, st in ys:
        st.check()
    return ys


def inflight():
    ys = []
    for i in range(NG):
        with torch.cuda.stream(streams[i % 4]):
            ys.append(model.convert(wav, target=tg, defer_status=True))
    for _, st in ys:
        st.check()
    return ys


print(f"group of {NG} batches of 32 x 5 s:")
print(f"  front ends together (YAAPT + extractor + speaker rows): {timed(fronts_only, 6) / NG:6.2f} ms per batch")
print(f"  generators together:                                    {timed(gens_only, 6) / NG:6.2f} ms per batch")
print(f"  phased (front ends, barrier, generators, barrier):      {timed(phased, 6) / NG:6.2f} ms per batch")
print(f"  convert() calls in flight (deferred status):            {timed(inflight, 6) / NG:6.2f} ms per batch")
