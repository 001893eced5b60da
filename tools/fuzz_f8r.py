"""random shapes through the checks of tests/test_hip_f8r.py (SAT_CONV_F16F8R vs its decomposition in float64, the three ResBlock epilogues,
planes / sidecar / hi-only outputs) and through the upsampler-ring and x-vector chain checks: python tools/fuzz_f8r.py [n] [seed]"""
import io, os, random, sys, contextlib, traceback
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests")]
import torch
import test_hip_f8r as tf

n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
bad = []
for i in range(n):
    C = 32 * rng.randint(3, 16)
    k = rng.choice([3, 3, 4, 5, 6, 7, 7, 9, 11, 11])          # (>= 3 taps: what the ring kernel serves)
    dil = rng.choice([d for d in (1, 2, 3, 5, 6) if (k - 1) * d <= 64])
    T = rng.choice([rng.randint(1, 40), rng.randint(41, 700), rng.randint(701, 2600), 160 * rng.randint(1, 6) + rng.choice([-1, 0, 1]), 320 * rng.randint(1, 4) + rng.choice([-1, 0, 1])])
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            tf.test_ring_conv_f16f8r_matches_its_decomposition(C, T, k, dil)
    except Exception as e:     # noqa: BLE001
        bad.append((C, T, k, dil, repr(e)[:300]))
        print("FAIL", C, T, k, dil, repr(e)[:300], flush=True)
    if i % 20 == 19:
        print(f"{i + 1} shapes, {len(bad)} failures; last: C {C} T {T} k {k} dil {dil}: {out.getvalue().strip()[:120]}", flush=True)
print(f"fuzz_f8r: {n} shapes (seed {seed}), {len(bad)} failures")
for b in bad:
    print("  ", b)
sys.exit(1 if bad else 0)
