#!/usr/bin/env python3
"""Golden fixtures for the `anonymize` data plane: runs the REFERENCE's own
`satools.bin.pipeline.process_data`, `collate_fn`, `script_utils.split_dict / read_wav_scp` in the build
container on a tiny synthetic kaldi data dir with a stand-in model object (the model is not what is being
pinned here: shards, batch composition and order, the `random` call sequence of the six target-selection
algorithms, cropping, the output tree are).  torchaudio (third party, absent) is the stand-in of
tests/golden/refstub plus load/save shims defined here.   python tests/golden/make_pipeline_fixtures.py"""
import json
import multiprocessing
import os
import random
import shutil
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_fixtures as mf   # noqa: E402

SCRATCH = "/tmp/sat_pipeline_fixture"


sys.path.insert(0, os.path.dirname(HERE))
from pipeline_toy import StandInModel, make_dataset, read_wav, write_wav   # noqa: E402


def main():
    mf.setup_reference()
    import torchaudio   # the stand-in package of tests/golden/refstub

    def ta_load(path, frame_offset=0, num_frames=-1):
        pcm, sr = read_wav(path)
        return torch.from_numpy(pcm.astype(np.float32) / 32768.0).unsqueeze(0), sr

    def ta_save(path, wav, freq, encoding=None, bits_per_sample=None):
        assert encoding == "PCM_S" and bits_per_sample == 16
        write_wav(path, wav.squeeze(0).numpy().astype(np.float64), freq)

    torchaudio.load, torchaudio.save = ta_load, ta_save
    sys.modules.setdefault("tqdm", types.SimpleNamespace(tqdm=lambda *a, **k: None))
    import satools.script_utils as su
    import satools.bin.pipeline as rp
    rp.load_model = lambda *a, **k: StandInModel()

    out = {}
    shutil.rmtree(SCRATCH, ignore_errors=True)
    data = os.path.join(SCRATCH, "data", "toy")
    make_dataset(data)
    wavscp = su.read_wav_scp(os.path.join(data, "wav.scp"))
    out["read_wav_scp_keys"] = list(wavscp.keys())
    out["split_dict_12_into_3"] = [list(d.keys()) for d in su.split_dict(wavscp, 3)]
    out["split_dict_12_into_5"] = [list(d.keys()) for d in su.split_dict(wavscp, 5)]
    ten = {f"k{i}": str(i) for i in range(10)}
    out["split_dict_10_into_3"] = [list(d.keys()) for d in su.split_dict(ten, 3)]

    # collate_fn on unequal lengths
    items = [{"utid": f"u{i}", "audio": torch.arange(n, dtype=torch.float32).unsqueeze(0) / 100, "f0": torch.ones(1, n // 3) * i, "freq": 16000}
             for i, n in enumerate((7, 12, 9))]
    a, f0, lens, utids, freqs = rp.collate_fn(items)
    out["collate"] = {"audio": a.tolist(), "f0": f0.tolist(), "lengths": lens.tolist(), "utids": utids, "freqs": freqs}

    settings = types.SimpleNamespace(model="stand-in", f0_modification="quant_16_awgn_2", target_constant_spkid="tgt007",
                                     results_dir="wav", batch_size=5, data_loader_nj=0, new_datadir_suffix="_anon", device="cpu")
    runs = {}
    for algo in ("constant", "none", "bad_for_evaluation", "random_per_utt", "random_per_spk_uniq", "random_per_spk"):
        shutil.rmtree(data + "_anon", ignore_errors=True)
        StandInModel.calls = []
        random.seed(0)
        progress = multiprocessing.Value("i", 0)
        rp.process_data(data, algo, wavscp, settings, progress)
        scp = open(os.path.join(data + "_anon", "wav.scp")).read().replace(SCRATCH, "$ROOT")
        lens_out = {u: int(len(read_wav(os.path.join(data + "_anon", "wav", u + ".wav"))[0])) for u in wavscp}
        runs[algo] = {"calls": StandInModel.calls, "wav_scp": scp, "out_lengths": lens_out,
                      "copied_files": sorted(f for f in os.listdir(data + "_anon") if os.path.isfile(os.path.join(data + "_anon", f))),
                      "progress": progress.value}
    out["runs"] = runs
    pcm, _ = read_wav(os.path.join(data + "_anon", "wav", "utt03.wav"))
    out["utt03_first_pcm"] = pcm[:16].tolist()
    src, _ = read_wav(os.path.join(data, "clear", "utt03.wav"))
    out["utt03_src_first_pcm"] = src[:16].tolist()
    json.dump(out, open(os.path.join(HERE, "fx_pipeline.json"), "w"), indent=0)
    print("wrote fx_pipeline.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
