"""`anonymize` data plane against fixtures recorded from the reference's own process_data /
collate_fn / split_dict run on the same toy data dir with the same stand-in model
(tests/golden/make_pipeline_fixtures.py): shards, batch composition and order, the `random` call sequence
of the six target-selection algorithms, F0 hand-over shapes, cropping, PCM16 output, output tree."""
import json
import multiprocessing
import os
import random
import types

import pytest
import torch

import satools_amd   # noqa: F401
from satools_amd import pipeline as pl
from pipeline_toy import StandInModel, make_dataset, read_wav

FX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fx_pipeline.json")))


def test_read_wav_scp_and_split_dict(tmp_path):
    data = tmp_path / "data" / "toy"
    make_dataset(str(data))
    scp = pl.read_wav_scp(data / "wav.scp")
    assert list(scp.keys()) == FX["read_wav_scp_keys"]
    assert [list(d.keys()) for d in pl.split_dict(scp, 3)] == FX["split_dict_12_into_3"]
    assert [list(d.keys()) for d in pl.split_dict(scp, 5)] == FX["split_dict_12_into_5"]
    assert [list(d.keys()) for d in pl.split_dict({f"k{i}": str(i) for i in range(10)}, 3)] == FX["split_dict_10_into_3"]


def test_collate_fn_matches_reference():
    items = [{"utid": f"u{i}", "audio": torch.arange(n, dtype=torch.float32).unsqueeze(0) / 100, "f0": torch.ones(1, n // 3) * i, "freq": 16000}
             for i, n in enumerate((7, 12, 9))]
    a, f0, lens, utids, freqs = pl.collate_fn(items)
    c = FX["collate"]
    assert a.tolist() == c["audio"] and f0.tolist() == c["f0"] and lens.tolist() == c["lengths"]
    assert utids == c["utids"] and freqs == c["freqs"]


@pytest.mark.parametrize("algo", list(FX["runs"].keys()))
def test_process_data_matches_reference_run(tmp_path, algo):
    data = tmp_path / "data" / "toy"
    make_dataset(str(data))
    settings = types.SimpleNamespace(model="stand-in", f0_modification="quant_16_awgn_2", target_constant_spkid="tgt007",
                                     results_dir="wav", batch_size=5, data_loader_nj=2, new_datadir_suffix="_anon", device="cpu")
    StandInModel.calls = []
    random.seed(0)
    progress = multiprocessing.Value("i", 0)
    n = pl.process_data(str(data), algo, pl.read_wav_scp(data / "wav.scp"), settings, progress, model=StandInModel())
    ref = FX["runs"][algo]
    assert n == 12 and progress.value == 12
    assert StandInModel.calls == ref["calls"]                        # batch shapes, targets, f0 hand-over, in order
    out = str(data) + "_anon"
    assert open(os.path.join(out, "wav.scp")).read().replace(str(tmp_path), "$ROOT") == ref["wav_scp"]
    assert {u: len(read_wav(os.path.join(out, "wav", u + ".wav"))[0]) for u in ref["out_lengths"]} == ref["out_lengths"]
    assert sorted(f for f in os.listdir(out) if os.path.isfile(os.path.join(out, f))) == ref["copied_files"]
    pcm, sr = read_wav(os.path.join(out, "wav", "utt03.wav"))
    assert sr == 16000 and pcm[:16].tolist() == FX["utt03_first_pcm"]


def test_jobs_of_one_device_keep_their_shards_and_rng(tmp_path):
    """two jobs served by one process: each shard keeps its own batches and replays the launcher's random stream
    (the reference forks one process per job from the same parent state)"""
    data = tmp_path / "data" / "toy"
    make_dataset(str(data))
    scp = pl.read_wav_scp(data / "wav.scp")
    shards = pl.split_dict(scp, 2)
    settings = types.SimpleNamespace(model="stand-in", f0_modification="", target_constant_spkid="?", results_dir="wav",
                                     batch_size=4, data_loader_nj=1, new_datadir_suffix="_anon", device="cpu")
    random.seed(3)
    state = random.getstate()
    StandInModel.calls = []
    pl.process_data(str(data), "random_per_utt", shards, settings, model=StandInModel(), rng_state=state)
    both = StandInModel.calls
    lines = open(str(data) + "_anon/wav.scp").read().split("\n")
    assert [l.split()[0] for l in lines if l] == list(scp.keys())
    singles = []
    for sh in shards:
        StandInModel.calls = []
        pl.process_data(str(data), "random_per_utt", sh, settings, model=StandInModel(), rng_state=state)
        singles.append(StandInModel.calls)
    # interleaved round-robin: job 0 batch 0, job 1 batch 0, job 0 batch 1, ...
    assert both[0::2] == singles[0] and both[1::2] == singles[1]


def test_pipe_entries_and_unknown_algorithm(tmp_path):
    data = tmp_path / "d"
    make_dataset(str(data))
    path = str(data / "clear" / "utt00.wav")
    a, sr = pl.load_wav_from_scp(path)
    b, _ = pl.load_wav_from_scp(f"cat {path} |")
    assert sr == 16000 and torch.equal(a, b) and a.shape == (1, 3000)
    with pytest.raises(ValueError):
        pl.TargetSelector("nearest", ["a"], {})


def test_non_riff_audio_gives_a_clear_error(tmp_path):
    """wav.scp entries that point straight at a flac file (the reference reads them through torchaudio.load): decoded by
    `soundfile` when installed, else an IOError that names the `flac -c -d -s f |` pipe entry"""
    from satools_amd import pipeline as P
    f = tmp_path / "utt.flac"
    f.write_bytes(b"fLaC" + b"\0" * 64)
    try:
        import soundfile  # noqa: F401
        have = True
    except ImportError:
        have = False
    if not have:
        import pytest
        with pytest.raises(IOError, match="flac -c -d -s"):
            P.load_wav_from_scp(str(f))


def _riff(fmt_body, data, extra=b""):
    import struct
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + extra + b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def test_fast_pcm16_reader_and_writer_agree_with_the_general_ones(tmp_path):
    """the batch job keeps 16-bit mono files as int16 up to the device and writes its RIFF files itself: the same samples as
    `load_wav_from_scp` (scipy) x 32768, the same BYTES as scipy.io.wavfile.write; anything but plain 16-bit mono PCM is left to the
    general reader"""
    import struct
    import numpy as np
    from scipy.io import wavfile
    rng = np.random.RandomState(0)
    pcm = rng.randint(-32768, 32768, size=4001).astype(np.int16)
    fmt16 = struct.pack("<HHIIHH", 1, 1, 16000, 32000, 2, 16)
    plain = tmp_path / "plain.wav"
    plain.write_bytes(_riff(fmt16, pcm.tobytes()))
    got, sr = pl.read_pcm16_mono(str(plain))
    ref, sr_ref = pl.load_wav_from_scp(str(plain))
    assert sr == sr_ref == 16000 and got.dtype == np.int16 and np.array_equal(got, pcm)
    assert torch.equal(torch.from_numpy(got.astype(np.float32) * np.float32(1 / 32768)).unsqueeze(0), ref)
    # a LIST chunk in front of the samples, a fmt chunk with an extension field, an odd-sized chunk with its pad byte
    listed = tmp_path / "listed.wav"
    listed.write_bytes(_riff(fmt16 + b"\0\0", pcm.tobytes(), extra=b"LIST" + struct.pack("<I", 5) + b"abcde\0"))
    got2, _ = pl.read_pcm16_mono(str(listed))
    assert np.array_equal(got2, pcm) and torch.equal(pl.load_wav_from_scp(str(listed))[0], ref)
    # not the plain case -> None (the general reader decodes or rejects them)
    stereo = _riff(struct.pack("<HHIIHH", 1, 2, 16000, 64000, 4, 16), pcm[:4000].tobytes())
    eight = _riff(struct.pack("<HHIIHH", 1, 1, 16000, 16000, 1, 8), bytes(100))
    f32 = _riff(struct.pack("<HHIIHH", 3, 1, 16000, 64000, 4, 32), np.zeros(10, np.float32).tobytes())
    ext = _riff(struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 32000, 2, 16) + struct.pack("<HHI", 22, 16, 4) + b"\x01\0" + bytes(14), pcm.tobytes())
    cut = _riff(fmt16, pcm.tobytes())[:-100]
    for name, blob in (("stereo", stereo), ("eight", eight), ("f32", f32), ("ext", ext), ("cut", cut), ("short", b"RIFF1234WAVE"), ("flac", b"fLaC" + bytes(60))):
        f = tmp_path / (name + ".wav")
        f.write_bytes(blob)
        assert pl.read_pcm16_mono(str(f)) is None, name
    assert pl.load_wav_from_scp(str(tmp_path / "stereo.wav"))[0].shape == (2, 2000)
    # collate: s / 32768 of the int16 batch is collate_fn's audio
    items = [{"utid": f"u{i}", "pcm": pcm[:n], "audio": None, "f0": None, "freq": 16000} for i, n in enumerate((7, 4001, 90))]
    a16, f0, lens, utids, freqs = pl.collate_pcm16(items)
    for i in items:
        i["audio"] = torch.from_numpy(i["pcm"].astype(np.float32) / 32768.0).unsqueeze(0)
    a, _, lens_ref, utids_ref, freqs_ref = pl.collate_fn(items)
    assert f0 is None and a16.dtype == np.int16 and torch.equal(torch.from_numpy(a16.astype(np.float32) * np.float32(1 / 32768)), a)
    assert torch.equal(lens, lens_ref) and utids == utids_ref and freqs == freqs_ref
    # the writer: scipy's bytes, one and two channels; save_pcm16 of f32 = of its int16 conversion; ties round to even, clipping
    for shape in ((4001,), (1, 4001), (2, 300)):
        x = rng.randint(-32768, 32768, size=shape).astype(np.int16)
        mine, theirs = tmp_path / "mine.wav", tmp_path / "theirs.wav"
        pl.write_riff_pcm16(mine, x, 22050)
        wavfile.write(str(theirs), 22050, np.ascontiguousarray(x.T))
        assert mine.read_bytes() == theirs.read_bytes(), shape
    x = torch.tensor([[0.5 / 32768, 1.5 / 32768, 2.5 / 32768, -0.5 / 32768, -1.5 / 32768, 1.0, -1.0, 1.5, -1.5, 0.99999, 32766.5 / 32768, 0.1]])
    assert pl.pcm16_of(x).tolist() == [[0, 2, 2, 0, -2, 32767, -32768, 32767, -32768, 32767, 32766, 3277]]
    pl.save_pcm16(tmp_path / "a.wav", x, 16000)
    pl.save_pcm16(tmp_path / "b.wav", torch.from_numpy(pl.pcm16_of(x)), 16000)
    assert (tmp_path / "a.wav").read_bytes() == (tmp_path / "b.wav").read_bytes()
    assert read_wav(tmp_path / "a.wav")[0].tolist() == pl.pcm16_of(x)[0].tolist()
