#!/usr/bin/env python3
"""Benchmark of the anonymize / model.convert() hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" = one `model.convert` over one batch of 32 synthetic 5 s @ 16 kHz utterances per rank
(BASELINE.json configs[1], tag hifigan_bn_tdnnf_600h_vq_48_v1), inputs resident in HBM, followed
for N > 1 by the RCCL all-gather of the anonymized waveforms (weak scaling: per-GPU work fixed).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The path keeps several HIP streams busy per process (jobs in flight x (launch stream + YAAPT side
# stream)); the HIP runtime multiplexes streams onto 4 hardware queues by default, which serialises
# independent streams (measured 15.4 -> 14.3 ms/step at 2 jobs).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

TAG = "hifigan_bn_tdnnf_600h_vq_48_v1"
BATCH = 32
N_SAMPLES = 80000
UTT_SECONDS = 5.0
# algorithmic work per 5 s utterance (SURVEY §8d / DESIGN.md): generator 40.43 GMAC
GEN_FLOP_PER_UTT = 80.86e9
GEN_BYTES_PER_UTT = 731.4e6      # per-layer streaming model of the generator
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0    # dense f16/bf16 MFMA (MI355X_MICROARCH.md); split-f16 issues 3 MFMA products per product
PEAK_HBM_TBS = 8.0


def analytic_f0(seeds, frames=250):
    """F0 track of the `harm` utterances at the YAAPT frame centres (0 where the envelope is off);
    used while F0 is handed over through set_f0 (the anonymize pipeline's hand-off)"""
    import math
    import torch
    t = (torch.arange(frames, dtype=torch.float64) * 320) / 16000.0
    rows = []
    for s in seeds:
        f0 = (100 + 5 * (s % 16)) + 60 * torch.sin(2 * math.pi * 0.7 * t)
        env = (torch.sin(2 * math.pi * 1.5 * t) > -0.3).to(torch.float64)
        rows.append((f0 * env).to(torch.float32))
    return torch.stack(rows)


def cpu_baseline(state, spk, seeds):
    """the CPU oracle (a port of the reference's PyTorch path: YAAPT loop over the batch, fbank,
    TDNNF-VQ, generator) timed on the host cores over a bounded sample of the same workload"""
    import torch
    from oracle import convert as oconv
    from oracle import yaapt as oyaapt
    from satools_amd import synthetic
    wav = synthetic.harm_batch(seeds)
    tg = synthetic.targets(spk, seeds)
    cores = torch.get_num_threads()
    opts = {"frame_length": 35.0, "frame_space": 20.0, "nccf_thresh1": 0.25, "tda_frame_length": 25.0}
    with torch.no_grad():
        t0 = time.perf_counter()
        f0 = oyaapt.yaapt(wav, opts)
        t1 = time.perf_counter()
        oconv.convert_fbank(state["base_model_state_dict"], spk, wav, tg, f0)
        dt = time.perf_counter() - t0
    return {"value": round(len(seeds) * UTT_SECONDS / dt, 3), "unit": "x real-time (audio s / wall s)", "cores": cores,
            "kind": "port", "sample": f"{len(seeds)} utterances x 5 s in one convert() batch, torch CPU f32, {cores} "
                                      f"threads ({dt:.1f} s, of which YAAPT {t1 - t0:.1f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jobs", type=int, default=4,
                    help="convert() calls in flight per GPU, each on its own HIP stream (the reference's "
                         "jobs_per_compute_device, satools/satools/bin/anonymize:85-93)")
    ap.add_argument("--tag", default=TAG, help="model tag (default: the headline config; the wav2vec2 tag is BASELINE configs[2])")
    ap.add_argument("--f0-transformation", default="", help="e.g. quant_16_awgn_2 (BASELINE configs[3])")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        print(f"bench.py --gpus {a.gpus} must be launched with torch.distributed.run --nproc-per-node {a.gpus}",
              file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # SAT_BENCH_FORCE_PG=1: take the RCCL path (process group, all-gather per step) with one rank too, to check it on a
    # 1-GPU box (launch under torch.distributed.run --nproc-per-node 1)
    use_pg = world > 1 or os.environ.get("SAT_BENCH_FORCE_PG") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import satools_amd
    from satools_amd import synthetic

    tag = a.tag
    model = satools_amd.load_model("synthetic:" + tag, option_args={"f0_transformation": a.f0_transformation} if a.f0_transformation else None)
    model.to(dev)
    model.eval()
    seeds = [rank * BATCH + i for i in range(BATCH)]
    wav = synthetic.harm_batch(seeds).to(dev)
    f0 = analytic_f0(seeds).to(dev)
    targets = synthetic.targets(model.spk, seeds)
    gathered = ([torch.empty(world * BATCH, 1, N_SAMPLES + 1, dtype=torch.float32, device=dev)
                 for _ in range(max(1, a.jobs))] if use_pg else None)   # indexed modulo the job count

    # every job stream owns its workspaces and must have run (and allocated) once before the timed region starts: when
    # --warmup is smaller than the number of job streams, the missing runs are done as untimed set-up steps first
    # (reported as config.setup_steps)
    jobs = max(1, a.jobs)
    setup_steps = max(0, jobs - a.warmup)
    streams = [torch.cuda.Stream(device=dev) for _ in range(jobs)]
    step_no = [0]

    def step():
        # the whole path is on the timed region: fbank -> TDNNF-VQ, YAAPT F0, one-hot, generator.
        # Steps are independent batches; like the reference's jobs_per_compute_device they are kept
        # `jobs` deep in flight, each on its own stream, so one batch's latency-bound front end
        # overlaps the previous batch's generator.
        s = streams[step_no[0] % jobs]
        step_no[0] += 1
        with torch.cuda.stream(s):
            y = model.convert(wav, target=targets)
            if use_pg:
                dist.all_gather_into_tensor(gathered[step_no[0] % jobs], y.contiguous())
        return y

    for _ in range(setup_steps + a.warmup):
        step()
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    dt = time.perf_counter() - t0
    if use_pg:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # dominant kernel family: the fused conv1d MFMA kernel inside the generator.  Timed live with
    # events on the launch stream (= torch's current stream) around the generator forward.
    gen_ms = None
    if rank == 0:
        with torch.no_grad():
            f0n = f0.clone()
            bn = model.get_bn(wav)
            spk = model.get_spk_id(wav, targets)
            from satools_amd import ops
            ops.f0_norm_transform_(f0n)
            x = ops.assemble_input(bn, f0n, spk.to(dev, torch.float32).contiguous(), spk.shape[1])
            model.hifigan(x)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, min(a.steps, 10))
            ev0.record()
            for _ in range(reps):
                model.hifigan(x)
            ev1.record()
            torch.cuda.synchronize()
            gen_ms = ev0.elapsed_time(ev1) / reps

    # the single dominant kernel (20 % of GPU time in profiles/r01j_*): the 11-tap split-f16 conv of the
    # generator's first resblock stage, timed alone with events on its launch stream
    dom = None
    if rank == 0 and model.hifigan.precision == "f16x3":
        with torch.no_grad():
            from satools_amd import ops, packing
            C, T, k, d = 256, 1250, 11, 5
            g = torch.Generator(device="cpu").manual_seed(0)
            xk = torch.randn(BATCH, C, T, generator=g).to(dev)
            wk = packing.pack_conv_weight_f16x3((torch.randn(C, C, k, generator=g) * 0.02).to(dev))
            bk = torch.zeros(C, device=dev)
            xs, rs, ys = ops.act_split(xk, 0.1), ops.act_split(xk * 0.5, 0.1), ops.split_like(BATCH, C, T, dev)
            run = lambda: ops.conv1d(xk, wk, C, k, bias=bk, dilation=d, pad_left=(k * d - d) // 2, mode=1, x_split=xs,
                                     y_split=ys, y_split_slope=0.1, res_split=rs, res_split_slope=0.1, no_y=True, out=xk)
            for _ in range(3):
                run()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(20):
                run()
            ev1.record()
            torch.cuda.synchronize()
            us = ev0.elapsed_time(ev1) / 20 * 1e3
            flop = 2.0 * BATCH * C * C * k * T
            dom = {"name": "conv1d_f16x3_planes_kernel<2,2,11,5> (C=256, T=1250, 11 taps, dilation 5, batch 32; 18 launches per forward)",
                   "flop_per_launch": flop, "avg_us_per_launch": round(us, 1), "achieved": round(flop / us / 1e6, 1),
                   "frac": round(flop / us / 1e6 / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)}

    if rank == 0:
        total_audio = world * a.steps * BATCH * UTT_SECONDS
        achieved = GEN_FLOP_PER_UTT * BATCH / (gen_ms * 1e-3) / 1e12
        split = model.hifigan.precision == "f16x3"
        peak = PEAK_F16_MFMA_TFLOPS / 3.0 if split else PEAK_F32_MFMA_TFLOPS
        hbm = GEN_BYTES_PER_UTT * BATCH / (gen_ms * 1e-3) / 1e12
        traffic, traffic_note = None, None
        tpath = os.path.join(ROOT, "profiles", "r01o_generator_traffic.json")
        if os.path.exists(tpath) and tag == TAG:
            tj = json.load(open(tpath))["per_forward"]
            traffic = tj["traffic_GB_raw"] * 1e9
            traffic_note = (f"HBM bytes per generator forward (batch 32) from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                            f"passes (profiles/r01o_generator_traffic.json): fetch {tj['fetch_GB_raw']} GB as counted "
                            f"({tj['fetch_GB_doubled']} GB with the gfx950 wide-load x2 correction as upper bound) + write "
                            f"{tj['write_GB']} GB; per-layer streaming model {tj['algorithmic_GB_per_layer_model']} GB")
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": ("split-f16 conv family (conv1d_f16x3_planes_kernel, resblock_pair16/32_kernel)" if split else "conv1d_mfma_kernel")
                              + f": all launches of one generator forward (+ upsampler split pass and output stage), {gen_ms:.3f} ms per batch of {BATCH}",
                    "dominant_kernel": dom,
                    "practical_ceiling": {"value": 592.0, "unit": "TFLOP/s", "frac": round(achieved / 592.0, 4),
                                          "note": "bare loop of the same three f16 MFMAs per product on random operands: the chip holds "
                                                  "~1.78 GHz under that load (tools/mfma_rate.hip)"} if split else None,
                    "arithmetic": ("f32 operands split hi+lo f16, 3 f16 MFMA products per product, f32 accumulate; peak = "
                                   "dense f16 MFMA peak / 3" if split else "exact f32 MFMA"),
                    "algorithmic_flop_per_launch_group": GEN_FLOP_PER_UTT * BATCH,
                    "frac_of_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                    "hbm_model": {"achieved_TBps": round(hbm, 3), "peak_TBps": PEAK_HBM_TBS, "frac": round(hbm / PEAK_HBM_TBS, 4),
                                  "bytes_per_launch_group": GEN_BYTES_PER_UTT * BATCH}}
        out = {
            "metric": "anonymized audio seconds per wall-clock second (real-time factor), 5 s @ 16 kHz utterances",
            "value": round(total_audio / dt, 2),
            "unit": "x real-time (audio s / wall s)",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (generator matrix products as split-f16 x3 with f32 accumulate)" if model.hifigan.precision == "f16x3" else "f32",
            "data": "synthetic",
            "config": {"workload": f"{tag}{'+f0-transformation=' + a.f0_transformation if a.f0_transformation else ''} model.convert, batch=32 x 5 s @ 16 kHz synthetic `harm` utterances per GPU",
                       "batch_per_gpu": BATCH, "utt_seconds": UTT_SECONDS, "weights": "seeded random (conditioned)",
                       "f0": "YAAPT computed on-path on the GPU inside convert()",
                       "jobs_per_gpu": jobs, "setup_steps": setup_steps,
                       "parallelism": f"dp{world}" + (" + RCCL all_gather of waveforms per step" if world > 1 else "")},
            "roofline": roofline,
        }
        if not a.no_cpu_baseline and tag == TAG and world == 1:     # the CPU leg runs at N = 1 only
            state, _ = synthetic.checkpoint(TAG)
            sample = list(range(4))
            out["cpu_baseline"] = cpu_baseline(state, model.spk, sample)
        print(json.dumps(out), flush=True)
    if use_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
