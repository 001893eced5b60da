"""`anonymize` data plane on the GPU: the batch job's outputs are model.convert's"""
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = "hifigan_bn_tdnnf_600h_vq_48_v1"


def _dataset(root, lengths):
    from pipeline_toy import write_wav
    from satools_amd import synthetic
    os.makedirs(os.path.join(root, "clear"), exist_ok=True)
    scp, u2s = [], []
    for i, n in enumerate(lengths):
        x = synthetic.harm_batch([i], n)[0].numpy().astype(np.float64)
        path = os.path.join(root, "clear", f"utt{i:02d}.wav")
        write_wav(path, x)
        scp.append(f"utt{i:02d} {path}\n")
        u2s.append(f"utt{i:02d} src{i % 2}\n")
    open(os.path.join(root, "wav.scp"), "w").writelines(scp)
    open(os.path.join(root, "utt2spk"), "w").writelines(u2s)


def test_process_data_outputs_are_convert_outputs(tmp_path):
    import satools_amd
    from satools_amd import pipeline as pl
    from pipeline_toy import read_wav
    data = str(tmp_path / "data" / "toy")
    _dataset(data, [16000, 16000, 16000, 12800, 12800])
    model = satools_amd.load_model("synthetic:" + TAG)
    model.to("cuda")
    model.eval()
    target = model.spk[5]
    settings = types.SimpleNamespace(model="-", f0_modification="", target_constant_spkid=target, results_dir="wav",
                                     batch_size=3, data_loader_nj=2, new_datadir_suffix="_anon", device="cuda")
    scp = pl.read_wav_scp(os.path.join(data, "wav.scp"))
    n = pl.process_data(data, "constant", scp, settings, model=model)
    assert n == 5
    # batch 0: three utterances of one length -> exactly convert() of that batch
    wavs = torch.cat([pl.load_wav_from_scp(scp[f"utt{i:02d}"])[0] for i in range(3)]).to("cuda")
    ref = model.convert(wavs, target=[target] * 3).cpu()
    for i in range(3):
        got, sr = read_wav(os.path.join(data + "_anon", "wav", f"utt{i:02d}.wav"))
        exp = np.clip(np.rint(ref[i, 0, :16000].numpy().astype(np.float64) * 32768), -32768, 32767).astype(np.int16)
        assert sr == 16000 and got.shape == (16000,) and np.array_equal(got, exp)
    # batch 1: two utterances of 12800 samples, cropped to their length
    for i in (3, 4):
        got, _ = read_wav(os.path.join(data + "_anon", "wav", f"utt{i:02d}.wav"))
        assert got.shape == (12800,) and np.abs(got).max() > 0


def test_anonymize_cli_two_jobs(tmp_path):
    data = str(tmp_path / "data" / "toy")
    _dataset(data, [16000] * 6)
    cfg = tmp_path / "anon.cfg"
    cfg.write_text(f"""[var]
tag = {TAG}
[cmd]
device = cuda
ngpu = 0
jobs_per_compute_device = 2
pipeline = pipe
[pipe]
model = synthetic:${{:tag}}
f0_modification = quant_16_awgn_2
target_selection_algorithm = random_per_spk
batch_size = 2
data_loader_nj = 2
""")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bin", "anonymize"), "--config", str(cfg), "--directory", data],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = data + "_anon"
    lines = [l.split() for l in open(os.path.join(out, "wav.scp")).read().splitlines()]
    assert [l[0] for l in lines] == [f"utt{i:02d}" for i in range(6)]
    from pipeline_toy import read_wav
    for u, path in lines:
        pcm, sr = read_wav(path)
        assert sr == 16000 and pcm.shape == (16000,) and np.abs(pcm).max() > 0
    assert os.path.exists(os.path.join(out, "utt2spk")) and not [f for f in os.listdir(out) if f.startswith(".wav.scp.part")]


def test_convert_padded_is_the_batch_jobs_set_f0_then_convert(tmp_path):
    """ragged batch: convert_padded(x, lengths, targets) == convert() after set_f0 of the per-utterance tracks
    (each from a one-utterance get_f0 at its own length, zero-padded like the reference's collate)"""
    import satools_amd
    from satools_amd import synthetic
    model = satools_amd.load_model("synthetic:" + TAG)
    model.to("cuda")
    model.eval()
    lens = [16000, 12800, 9600, 16000]
    x = torch.zeros(len(lens), max(lens))
    for i, n in enumerate(lens):
        x[i, :n] = synthetic.harm_batch([i], n)[0]
    x = x.to("cuda")
    tg = synthetic.targets(model.spk, list(range(len(lens))))
    tracks = [model.get_f0(x[i:i + 1, :n].contiguous())[0] for i, n in enumerate(lens)]
    f0 = torch.zeros(len(lens), max(t.shape[0] for t in tracks), device="cuda")
    for i, t in enumerate(tracks):
        f0[i, :t.shape[0]] = t
    model.set_f0(f0)
    ref = model.convert(x, target=tg)
    got = model.convert_padded(x, lens, tg)
    assert torch.equal(got, ref)


def test_pcm16_conversions_on_the_device_are_the_host_ones():
    """sat_pcm16_to_f32 / sat_pcm16_from_f32 (the batch job keeps int16 on both sides of PCIe) against the numpy formulas of
    pipeline.pcm16_of / torchaudio.load's normalisation: every int16 value, ties, clipping, odd lengths, unaligned views"""
    from satools_amd import ops, pipeline as pl
    from satools_amd._lib import SatError
    every = torch.arange(-32768, 32768, dtype=torch.int32).to(torch.int16)
    f = ops.pcm16_to_f32(every.cuda())
    assert torch.equal(f.cpu(), every.to(torch.float32) / 32768.0)
    assert torch.equal(ops.pcm16_from_f32(f).cpu(), every)
    g = torch.Generator().manual_seed(0)
    for shape in ((1, 1), (3, 80001), (32, 1, 8001), (5, 1023), (2, 4097)):
        x = (torch.rand(shape, generator=g) * 2.4 - 1.2)
        x.view(-1)[::7] = (torch.randint(-40000, 40000, x.view(-1)[::7].shape, generator=g).float() + 0.5) / 32768.0        # ties and overflows
        exp = pl.pcm16_of(x)
        got = ops.pcm16_from_f32(x.cuda())
        assert got.dtype == torch.int16 and got.shape == x.shape and np.array_equal(got.cpu().numpy(), exp), shape
        back = ops.pcm16_to_f32(got)
        assert torch.equal(back.cpu(), torch.from_numpy(exp.astype(np.float32) / 32768.0))
    # views that start off the vector alignment take the scalar form
    base = torch.rand(4099, generator=g).cuda() - 0.5
    for off in (1, 2, 3):
        v = base[off:]
        assert np.array_equal(ops.pcm16_from_f32(v).cpu().numpy(), pl.pcm16_of(v.cpu()))
        p16 = ops.pcm16_from_f32(base)[off:]
        assert torch.equal(ops.pcm16_to_f32(p16).cpu(), p16.cpu().float() / 32768.0)
    with pytest.raises(SatError):
        ops.pcm16_to_f32(torch.zeros(4, dtype=torch.float32, device="cuda"))
    with pytest.raises(SatError):
        ops.pcm16_from_f32(torch.zeros(4, device="cuda"), out=torch.zeros(5, dtype=torch.int16, device="cuda"))


def test_int16_data_plane_writes_the_files_of_the_f32_one(tmp_path, monkeypatch):
    """SATOOLS_AMD_PIPELINE_PCM16 = 3 (default: int16 from the file to the device and back, conversions by sat_pcm16_*) against 0 (f32 on
    both sides of PCIe, scipy's parser and numpy's rounding on the host): the same bytes in every output file, ragged lengths, two jobs,
    a stereo-free mix of plain files and one file only the general reader takes (a LIST chunk is fine for both; an 8-bit file is not
    int16: its batch falls back to the f32 collate)"""
    import struct
    import satools_amd
    from satools_amd import pipeline as pl
    data = str(tmp_path / "data" / "toy")
    _dataset(data, [16000, 12800, 9600, 16000, 14400, 11200, 16000])
    # utt06 becomes an 8-bit PCM file (unsigned, 128 = zero): the general reader's case
    from pipeline_toy import read_wav
    pcm, _ = read_wav(os.path.join(data, "clear", "utt06.wav"))
    u8 = (np.clip(pcm // 256, -128, 127) + 128).astype(np.uint8).tobytes()
    fmt = struct.pack("<HHIIHH", 1, 1, 16000, 16000, 1, 8)
    blob = b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(u8)) + u8
    open(os.path.join(data, "clear", "utt06.wav"), "wb").write(b"RIFF" + struct.pack("<I", 4 + len(blob)) + b"WAVE" + blob)
    model = satools_amd.load_model("synthetic:" + TAG)
    model.to("cuda")
    model.eval()
    scp = pl.read_wav_scp(os.path.join(data, "wav.scp"))
    out = {}
    for mode in ("3", "0"):
        monkeypatch.setenv("SATOOLS_AMD_PIPELINE_PCM16", mode)
        settings = types.SimpleNamespace(model="-", f0_modification="", target_constant_spkid=model.spk[2], results_dir="wav", batch_size=3,
                                         data_loader_nj=2, new_datadir_suffix="_anon" + mode, device="cuda")
        assert pl.process_data(data, "constant", pl.split_dict(scp, 2), settings, model=model) == 7
        out[mode] = {u: open(os.path.join(data + "_anon" + mode, "wav", u + ".wav"), "rb").read() for u in scp}
    assert out["3"] == out["0"]
    assert [len(v) for v in out["3"].values()] == [44 + 2 * n for n in (16000, 12800, 9600, 16000, 14400, 11200, 16000)]
    assert torch.get_num_threads() >= 1


def test_an_untrackable_batch_fails_the_job_with_deferred_status(tmp_path):
    """the batch job does not wait for YAAPT's status word in its launching thread (the writer checks it): a batch with an utterance
    that has no voiced frame still fails the job with the error convert() raises — the reference's job dies inside spec_track"""
    import satools_amd
    from satools_amd import pipeline as pl
    from pipeline_toy import write_wav
    data = str(tmp_path / "data" / "toy")
    _dataset(data, [16000] * 6)
    write_wav(os.path.join(data, "clear", "utt04.wav"), np.zeros(16000))
    model = satools_amd.load_model("synthetic:" + TAG)
    model.to("cuda")
    model.eval()
    settings = types.SimpleNamespace(model="-", f0_modification="", target_constant_spkid=model.spk[0], results_dir="wav", batch_size=2,
                                     data_loader_nj=2, new_datadir_suffix="_anon", device="cuda")
    scp = pl.read_wav_scp(os.path.join(data, "wav.scp"))
    with pytest.raises(RuntimeError, match="no voiced frame"):
        pl.process_data(data, "constant", scp, settings, model=model)
    x = torch.cat([pl.load_wav_from_scp(scp[u])[0] for u in ("utt04", "utt05")]).to("cuda")
    with pytest.raises(RuntimeError, match="no voiced frame"):
        model.convert(x, target=[model.spk[0]] * 2)
    # the rows of the page-locked status block all come back (checked, or dropped with their launch)
    from satools_amd import f0
    import gc
    gc.collect()
    assert len(f0._pinned_ints.free) == f0._PinnedInts.ROWS
    y, st = model.convert_padded(x[1:], [16000], [model.spk[0]], defer_status=True)
    st.check()
    st.check()
    assert torch.equal(y, model.convert_padded(x[1:], [16000], [model.spk[0]]))
