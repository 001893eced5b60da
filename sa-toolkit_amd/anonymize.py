"""`anonymize --config C --directory D`: anonymize a kaldi wav.scp data dir on the MI355X path
(reference CLI: satools/satools/bin/anonymize:22-110; config helpers satools/satools/script_utils.py:244-290,
:441-495).  Same config file ([cmd] device / ngpu / jobs_per_compute_device / pipeline, [<pipeline>] model /
f0_modification / target_selection_algorithm / target_constant_spkid / results_dir / batch_size /
data_loader_nj / new_datadir_suffix, `${:name}` variables from [var] or the environment), same shards
(`split_dict(wavscp, len(ngpu) * jobs)` in GPU-major order), same output tree.

One process per GPU (the reference: one per GPU *and* job); the jobs of a GPU are HIP streams inside it
(satools_amd.pipeline).  Run as `python -m satools_amd.anonymize ...` or `bin/anonymize ...`."""
import argparse
import configparser
import logging
import multiprocessing
import os
import random
import re
import sys
import threading
import time
from dataclasses import dataclass, field
from typing import List


@dataclass
class Pipeline:
    model: str = "hifigan_bn_tdnnf_wav2vec2_vq_48_v1"
    f0_modification: str = "quant_16_awgn_2"
    target_selection_algorithm: str = "random_per_utt"
    target_constant_spkid: str = "?"
    results_dir: str = "wav"          # output of anonymized wavs: ./data/XXXX_anon/wav
    batch_size: int = 8
    data_loader_nj: int = 5
    new_datadir_suffix: str = "_anon"
    f0_mode: str = "per_utterance"    # satools_amd.pipeline.process_data; "batch" = YAAPT on the padded batch
    device: str = "cuda"


@dataclass
class Cmd:
    device: str = "cuda"
    ngpu: List[str] = field(default_factory=lambda: ["0"])
    jobs_per_compute_device: int = 1
    pipeline: str = "pipeline"


_RE_VAR = re.compile(r"[$][{][:]([a-zA-Z0-9_-]+)[}]")


def vartoml(cfg):
    """`${:name}` in any value -> the environment variable `name`, else the entry `name` of section [var]"""
    var = dict(cfg["var"]) if cfg.has_section("var") else {}

    def repl(m):
        name = m.group(1)
        if name in os.environ:
            return os.environ[name]
        if name not in var:
            raise KeyError(f"config variable ${{:{name}}} is neither in the environment nor in [var]")
        return var[name]

    out = {}
    for sec in cfg.sections():
        out[sec] = {k: _RE_VAR.sub(repl, v) for k, v in cfg.items(sec, raw=True)}
    return out


def visible_gpu_count():
    """number of GPUs this process tree may use, found WITHOUT the HIP runtime: the parent of the per-GPU workers never touches the GPU
    (it only spawns them; a process that has initialised HIP must not be replaced or forked around on this platform).  An explicit
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES list wins; else the amdgpu nodes the kernel driver exposes
    (KFD topology: nodes with a non-zero simd_count are GPUs; fallback /sys/class/drm/card*/device/vendor == 0x1002 render nodes)."""
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(name)
        if v is not None and v.strip() != "":
            return max(1, len([p for p in v.split(",") if p.strip() != ""]))
    import glob
    return max(1, _count_kfd_gpus(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"), "/dev/dri",
                                  glob.glob("/sys/class/drm/renderD*/device/vendor")))


def _count_kfd_gpus(prop_files, dri_dir, vendor_files):
    """GPU nodes of the KFD topology whose render node this process can OPEN: sysfs lists every GPU of the machine even inside a
    container / cgroup that was given only some /dev/dri/renderD* nodes, and HIP can only use those (what torch.cuda.device_count()
    would have said; round-5 advisor item).  A node without a `drm_render_minor` entry counts when no render directory exists at all."""
    n = 0
    have_dri = os.path.isdir(dri_dir)
    for prop in prop_files:
        try:
            kv = dict(line.split(None, 1) for line in open(prop) if " " in line)
            if int(kv.get("simd_count", "0")) <= 0:
                continue
            minor = int(kv.get("drm_render_minor", "-1"))
        except (OSError, ValueError):
            continue
        if minor >= 0 and have_dri:
            node = os.path.join(dri_dir, f"renderD{minor}")
            if not (os.path.exists(node) and os.access(node, os.R_OK | os.W_OK)):
                continue
        n += 1
    if n == 0:
        for ven in vendor_files:
            try:
                node = os.path.join(dri_dir, os.path.basename(os.path.dirname(os.path.dirname(ven))))
                n += open(ven).read().strip().lower() == "0x1002" and (not have_dri or os.access(node, os.R_OK | os.W_OK))
            except OSError:
                pass
    return n
    return max(1, n)


def parse_ngpu(value):
    """'0' / '0,2' / '[0, 2]' -> those ids; 'all' -> every visible GPU; an integer N written as 'N gpus' is not
    supported by the reference either (its bare integers go through safe_gpu: the first N free GPUs)"""
    v = str(value).strip().strip("[]")
    if v.lower() in ("all", "all-force"):
        return [str(i) for i in range(visible_gpu_count())]
    return [p.strip().strip("'\"") for p in v.split(",") if p.strip()]


def load_into(obj, section):
    for key, value in section.items():
        if not hasattr(obj, key):
            continue
        cur = getattr(obj, key)
        if key == "ngpu":
            setattr(obj, key, parse_ngpu(value))
        elif isinstance(cur, bool):
            setattr(obj, key, value.lower() in ("yes", "true", "t", "1"))
        elif isinstance(cur, int):
            setattr(obj, key, int(value))
        else:
            setattr(obj, key, value.strip().strip('"'))
    return obj


def _gpu_worker(gpu_id, directory, shards, settings, progress, rng_state, scp_part, hw_queues):
    os.environ["HIP_VISIBLE_DEVICES"] = str(gpu_id)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(hw_queues))      # one hardware queue per stream in use
    from . import pipeline
    pipeline.process_data(directory, settings.target_selection_algorithm, shards, settings, progress,
                          rng_state=rng_state, scp_out=scp_part, f0_mode=settings.f0_mode)


def _progress_bar(progress, total, stop):
    try:
        from tqdm import tqdm
    except ImportError:
        return
    with tqdm(total=total) as pbar:
        while progress.value < total and not stop.is_set():
            pbar.n = progress.value
            pbar.refresh()
            time.sleep(0.5)
        pbar.n = min(total, progress.value)
        pbar.refresh()


def main(argv=None):
    parser = argparse.ArgumentParser(description="anonymize a kaldi wav.scp formatted dataset (config file + directory)")
    parser.add_argument("--config", required=False)
    parser.add_argument("--directory", default="data/default", required=True)
    parser.add_argument("--pipeline", default=None, required=False)
    args = parser.parse_args(argv)

    cfg_cmd, cfg_pipeline = Cmd(), Pipeline()
    if args.config:
        parse = configparser.ConfigParser()
        if not parse.read(args.config):
            parser.error(f"cannot read config {args.config}")
        sections = vartoml(parse)
        load_into(cfg_cmd, sections.get("cmd", {}))
        load_into(cfg_pipeline, sections[args.pipeline if args.pipeline else cfg_cmd.pipeline])
    cfg_pipeline.device = cfg_cmd.device

    from .pipeline import read_wav_scp, split_dict
    wavscp = read_wav_scp(os.path.join(args.directory, "wav.scp"))
    jobs = max(1, int(cfg_cmd.jobs_per_compute_device))
    shards = split_dict(wavscp, len(cfg_cmd.ngpu) * jobs)           # GPU-major, like the reference's launch loop
    out_dir = str(args.directory) + cfg_pipeline.new_datadir_suffix
    os.makedirs(out_dir, exist_ok=True)

    ctx = multiprocessing.get_context("spawn")                        # the parent never touches the GPU
    progress = ctx.Value("i", 0)
    rng_state = random.getstate()                                     # the reference's forked jobs all inherit this
    procs, parts = [], []
    for gi, gpu_id in enumerate(cfg_cmd.ngpu):
        part = os.path.join(out_dir, f".wav.scp.part{gi}")
        parts.append(part)
        p = ctx.Process(target=_gpu_worker, args=(gpu_id, args.directory, shards[gi * jobs:(gi + 1) * jobs], cfg_pipeline,
                                                  progress, rng_state, part, 4 * jobs))
        p.start()
        procs.append(p)
    stop = threading.Event()
    bar = threading.Thread(target=_progress_bar, args=(progress, len(wavscp), stop), daemon=True)
    bar.start()
    failed = False
    for p in procs:
        p.join()
        if p.exitcode != 0:
            print(f"Process {p.pid} exited with code {p.exitcode}. Terminating.")
            failed = True
            for q in procs:
                if q.is_alive():
                    q.terminate()
            break
    stop.set()
    bar.join(timeout=2)
    if failed:
        sys.exit(1)
    with open(os.path.join(out_dir, "wav.scp"), "wt", encoding="utf-8") as writer:
        for part in parts:
            with open(part) as f:
                writer.write(f.read())
            os.remove(part)
    logging.info("Done")


if __name__ == "__main__":
    main()
