"""Data plane of the `anonymize` batch job: wav.scp in, anonymized PCM16 wavs + wav.scp out
(reference: satools/satools/bin/anonymize:22-110, satools/satools/bin/pipeline.py:19-187,
satools/satools/script_utils.py:495-525, satools/satools/utils/kaldi.py:85-128).

Same observable behaviour as the reference's job — the same shards (`split_dict`), the same batches in the
same order (the F0 normalisation and `pad_input` are batch-coupled, so batch composition is part of the
result), the same target choices for the same `random` state, the same output tree — on a data plane laid
out for one MI355X per process:

  * F0 is computed on the GPU inside `convert()`; the reference runs YAAPT in up to 18 DataLoader worker
    PROCESSES per job and hands the track over through `set_f0`.  Here the loader threads only read audio.
  * `jobs_per_compute_device` shards of one GPU are served by ONE process, each shard on its own HIP stream
    (the reference forks one process per shard and lets the driver time-slice them).
  * crop + PCM16 encoding + file writing happen on a small thread pool while the next batch runs (the
    reference forks a process per batch).

torchaudio is third-party to the reference and not present: wav decoding / PCM_S16 encoding are restated
with scipy / numpy (`load_wav_from_scp`, `save_pcm16`) — parity unpinned for the float -> int16 rounding."""
import glob
import io
import logging
import os
import random
import shutil
import struct
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch

from .infer_helper import load_model


# ---- script_utils.py:495-525 -------------------------------------------------------------------
def read_wav_scp(wav_scp):
    """kaldi table file -> dict: first column -> rest of the line joined by single spaces"""
    utt2wav = {}
    with open(wav_scp) as ipf:
        for line in ipf:
            lns = line.strip().split()
            utt2wav[lns[0]] = " ".join(lns[1:])
    return utt2wav


def split_dict(a, n):
    """n contiguous shards in key order, the first len(a) % n one longer"""
    keys = list(a.keys())
    k, m = divmod(len(keys), n)
    return [{key: a[key] for key in keys[i * k + min(i, m):(i + 1) * k + min(i + 1, m)]} for i in range(n)]


# ---- utils/kaldi.py:85-128 (torchaudio.load restated) ---------------------------------------------
def _decode_wav(fileobj):
    """RIFF wav through scipy; anything else (the reference reads flac through torchaudio.load: LibriSpeech and
    LibriTTS ship as flac) through `soundfile` when it is installed, else a clear error naming the shell-pipe entry
    that always works"""
    from scipy.io import wavfile
    head = b""
    if isinstance(fileobj, (str, os.PathLike)):
        with open(fileobj, "rb") as f:
            head = f.read(4)
    else:
        head = fileobj.read(4)
        fileobj.seek(0)
    if head != b"RIFF":
        try:
            import soundfile
        except ImportError:
            kind = "flac" if head == b"fLaC" else repr(head)
            raise IOError(f"{fileobj if isinstance(fileobj, (str, os.PathLike)) else 'piped audio'}: not a RIFF wav ({kind}) and "
                          "the `soundfile` package is not installed; write the wav.scp entry as a decoding pipe, e.g. "
                          "`flac -c -d -s /path/utt.flac |`")
        data, sr = soundfile.read(fileobj, dtype="float32", always_2d=True)
        return torch.from_numpy(np.ascontiguousarray(data.T)), int(sr)
    sr, data = wavfile.read(fileobj)
    if data.dtype == np.int16:
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if x.ndim == 1:
        x = x[None, :]
    else:
        x = x.T
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)


def load_wav_from_scp(wav, frame_offset=0, num_frames=-1):
    """a wav.scp entry (a path, or a shell command ending in `|` that writes a wav to stdout) -> ([channels, n] f32
    in [-1, 1], sample rate)"""
    wav = wav.strip()
    if wav.endswith("|"):
        with open(os.devnull, "w") as devnull:
            try:
                proc = subprocess.Popen(wav[:-1], stdout=subprocess.PIPE, shell=True, stderr=devnull)
                sample, sr = _decode_wav(io.BytesIO(proc.communicate()[0]))
            except Exception as e:
                raise IOError("Error processing wav file: {}\n{}".format(wav, e))
    else:
        sample, sr = _decode_wav(wav)
    if frame_offset or num_frames >= 0:
        end = None if num_frames < 0 else frame_offset + num_frames
        sample = sample[:, frame_offset:end]
    return sample, sr


def read_pcm16_mono(path):
    """the plain case of `load_wav_from_scp` without the float conversion: a RIFF file holding one channel of 16-bit PCM ->
    (int16 samples [n] viewing the file's bytes, sample rate); None for anything else (other sample formats, several channels,
    WAVE_FORMAT_EXTENSIBLE, a data chunk that runs past the end of the file: `_decode_wav` handles or rejects those).  The batch
    job keeps such utterances as int16 up to the device (`sat_pcm16_to_f32`: s / 32768, the value torchaudio.load gives)."""
    with open(path, "rb") as f:
        buf = f.read()
    n = len(buf)
    if n < 44 or buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        return None
    pos, fmt = 12, None
    while pos + 8 <= n:
        cid, size = buf[pos:pos + 4], int.from_bytes(buf[pos + 4:pos + 8], "little")
        body = pos + 8
        if cid == b"fmt ":
            if size < 16 or body + 16 > n:
                return None
            tag, ch, sr, _rate, align, bits = struct.unpack_from("<HHIIHH", buf, body)
            fmt = (tag, ch, sr, align, bits)
        elif cid == b"data":
            if fmt is None or fmt[0] != 1 or fmt[1] != 1 or fmt[3] != 2 or fmt[4] != 16 or size % 2 or body + size > n:
                return None
            return np.frombuffer(buf, dtype="<i2", count=size // 2, offset=body), int(fmt[2])
        pos = body + size + (size & 1)
    return None


def pcm16_of(wav):
    """f32 samples -> the int16 ones of torchaudio.save(..., encoding='PCM_S', bits_per_sample=16): round-half-even of x * 32768, clipped.
    UNPINNED against torchaudio (third party, not installable here): a sox / ffmpeg backend that truncates or scales by 32767
    differs by at most 1 LSB (3e-5 of full scale, below the path's 1e-4 RMS bar).  `sat_pcm16_from_f32` computes the same bits on
    the device (x * 32768 is exact in f32 as in f64)."""
    x = wav.detach().cpu().numpy() if isinstance(wav, torch.Tensor) else np.asarray(wav)
    if x.dtype == np.int16:
        return x
    return np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768, 32767).astype(np.int16)


def write_riff_pcm16(path, pcm, freq):
    """int16 samples [n] or [channels, n] -> a RIFF file: the 44-byte header scipy.io.wavfile.write (and sox) produce for PCM — `fmt `
    chunk of 16 bytes, format tag 1, no `fact` chunk — then the interleaved samples"""
    pcm = np.asarray(pcm)
    ch = 1 if pcm.ndim == 1 else int(pcm.shape[0])
    data = np.ascontiguousarray(pcm.reshape(-1) if ch == 1 else pcm.T, dtype="<i2")
    nbytes = data.size * 2
    if nbytes + 36 > 0xFFFFFFFF:
        raise ValueError("write_riff_pcm16: more than 4 GB of samples do not fit a RIFF file")
    freq = int(freq)
    header = (b"RIFF" + struct.pack("<I", 36 + nbytes) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, freq, freq * ch * 2, ch * 2, 16)
              + b"data" + struct.pack("<I", nbytes))
    with open(str(path), "wb") as f:
        f.write(header)
        f.write(memoryview(data).cast("B"))


def save_pcm16(path, wav, freq):
    """torchaudio.save(path, wav, freq, encoding='PCM_S', bits_per_sample=16) restated: [channels, n] f32 (or int16 samples that are
    already converted) -> RIFF (`pcm16_of`, `write_riff_pcm16`)"""
    write_riff_pcm16(path, pcm16_of(wav), freq)


# ---- pipeline.py:19-66 ---------------------------------------------------------------------------
def copy_data_dir(dataset_path, output_path):
    """utt2spk, wav.scp ... but not the directories inside (they may hold clear or anonymized wavs)"""
    os.makedirs(output_path, exist_ok=True)
    for p in glob.glob(str(Path(dataset_path) / "*"), recursive=False):
        if os.path.isfile(p):
            shutil.copy(p, output_path)


def collate_fn(item_list):
    """zero-pad audio (and f0 when present) to the batch maximum -> (audio [B, n_max], f0 [B, T_max] or None,
    lengths [B] int64, utids, freqs)"""
    batch_size = len(item_list)
    audios = [i["audio"] for i in item_list]
    lengths = torch.tensor([a.shape[-1] for a in audios])
    out = torch.zeros([batch_size, int(torch.max(lengths).item())])
    for i, a in enumerate(audios):
        out[i, :a.shape[-1]] = a.squeeze()
    f0 = None
    if all(i.get("f0") is not None for i in item_list):
        f0s = [i["f0"] for i in item_list]
        f0 = torch.zeros([batch_size, max(f.shape[-1] for f in f0s)])
        for i, f in enumerate(f0s):
            f0[i, :f.shape[-1]] = f.squeeze()
    return out, f0, lengths, [i["utid"] for i in item_list], [i["freq"] for i in item_list]


def collate_pcm16(item_list):
    """`collate_fn` for a batch whose utterances are all still int16 (`read_pcm16_mono`): zero-padded [B, n_max] int16 as a numpy
    array instead of the f32 tensor — s / 32768 of it IS collate_fn's audio — with the same lengths, ids and rates"""
    lengths = torch.tensor([int(i["pcm"].shape[0]) for i in item_list])
    out = np.zeros((len(item_list), int(torch.max(lengths).item())), dtype=np.int16)
    for k, i in enumerate(item_list):
        out[k, :i["pcm"].shape[0]] = i["pcm"]
    return out, None, lengths, [i["utid"] for i in item_list], [i["freq"] for i in item_list]


class TargetSelector:
    """the six target-selection algorithms of pipeline.py:109-143, with the reference's call sequence on a
    `random.Random` stream.  The reference's jobs are forked children: each starts from the PARENT's global
    `random` state, so every shard replays the same stream — pass `random.getstate()` of the launcher."""

    ALGORITHMS = ("constant", "none", "bad_for_evaluation", "random_per_utt", "random_per_spk_uniq", "random_per_spk")

    def __init__(self, algorithm, possible_targets, source_utt2spk, constant_spkid="?", rng_state=None):
        if algorithm not in self.ALGORITHMS:
            raise ValueError(f"{algorithm} not implemented")
        self.algorithm = algorithm
        self.possible_targets = None if possible_targets is None else list(possible_targets)
        self.source_utt2spk = source_utt2spk
        self.constant_spkid = constant_spkid
        self.out_spk2target = {}
        self.rng = random.Random()
        self.rng.setstate(rng_state if rng_state is not None else random.getstate())

    def __call__(self, utids):
        a, rng = self.algorithm, self.rng
        if a == "constant":
            return [self.constant_spkid] * len(utids)
        if a == "none":
            return []
        out = []
        for ut in utids:
            if a == "random_per_utt":
                out.append(rng.choice(self.possible_targets))
                continue
            spk = self.source_utt2spk[ut]
            if a == "bad_for_evaluation":
                if spk not in self.out_spk2target:
                    self.out_spk2target[spk] = rng.sample(self.possible_targets, 2)
                out.append(rng.choice(self.out_spk2target[spk]))
            elif a == "random_per_spk_uniq":
                if spk not in self.out_spk2target:
                    self.out_spk2target[spk] = rng.choice(self.possible_targets)
                    self.possible_targets.remove(self.out_spk2target[spk])     # one target per source speaker
                out.append(self.out_spk2target[spk])
            else:   # random_per_spk
                if spk not in self.out_spk2target:
                    self.out_spk2target[spk] = rng.choice(self.possible_targets)
                out.append(self.out_spk2target[spk])
        return out


class _Shard:
    """one reference 'job': a shard of wav.scp, its batches in order, its target selector, its stream"""

    def __init__(self, wavscp, batch_size, selector, stream):
        self.items = list(wavscp.items())
        self.batch_size = batch_size
        self.selector = selector
        self.stream = stream
        self.pos = 0
        self.ring = [{"buf": {}, "busy": None} for _ in range(3)]
        self.ring_pos = 0

    def next_keys(self):
        if self.pos >= len(self.items):
            return None
        chunk = self.items[self.pos:self.pos + self.batch_size]
        self.pos += self.batch_size
        return chunk


def process_data(dataset_path, target_selection_algorithm, wavscps, settings, progress=None, model=None, rng_state=None,
                 scp_out=None, f0_mode="per_utterance"):
    """anonymize the shard(s) `wavscps` (a dict, or a list of dicts = the jobs of this compute device) of the
    kaldi data dir `dataset_path` into `<dataset_path><new_datadir_suffix>/` (pipeline.py:68-187).
    `settings`: an object with the reference's Pipeline fields (+ `device`).  Returns the number of utterances.

    f0_mode "per_utterance" (default, the reference's semantics: its Dataset runs get_f0 on every utterance at
    its own length, the collate zero-pads the tracks, `set_f0` hands them over; utterances of equal length in a
    batch share one YAAPT launch) or "batch" (YAAPT once on the zero-padded batch inside convert(): the same
    result when the utterances of a batch have one length, faster otherwise).
    scp_out: where the `utt path` lines go (default `<out>/wav.scp`; the CLI gives every process its own part
    file and merges them — the reference lets its jobs overwrite each other's wav.scp)."""
    with _TorchThreads(1 if torch.device(settings.device).type == "cuda" else None):
        return _process_data(dataset_path, target_selection_algorithm, wavscps, settings, progress, model, rng_state, scp_out, f0_mode)


def _process_data(dataset_path, target_selection_algorithm, wavscps, settings, progress, model, rng_state, scp_out, f0_mode):
    if isinstance(wavscps, dict):
        wavscps = [wavscps]
    dataset_path = Path(str(dataset_path))
    output_path = Path(str(dataset_path) + settings.new_datadir_suffix)
    device = torch.device(settings.device)
    copy_data_dir(dataset_path, output_path)
    results_dir = output_path / settings.results_dir
    os.makedirs(results_dir, exist_ok=True)

    if model is None:
        option_args = {}
        if settings.f0_modification != "":
            option_args["f0_transformation"] = settings.f0_modification
        model = load_model(settings.model, option_args=option_args)
        model.to(device)
        model.eval()
    possible_targets = model.spk.copy() if hasattr(model, "spk") else None
    if possible_targets is None:
        logging.info("Model without explicit target")
    source_utt2spk = read_wav_scp(dataset_path / "utt2spk")
    if rng_state is None:
        rng_state = random.getstate()

    use_streams = device.type == "cuda"
    # (process_data:) intra-op threads of torch's CPU kernels: ONE for the duration of the job, as in the reference (`torch.set_num_threads(1)` at the
    # import of satools/hifigan/yaapt.py:27).  Measured on the 256-core host of an MI355X box (tools/scratch/pcm_in_probe5.py): a single
    # parallel CPU op per batch — `pinned.copy_(batch)`, 10 MB — leaves torch's OpenMP team spinning on every core after the copy and the
    # HIP runtime's own threads starved: the batch took 31 ms instead of 10.7 (the wait for the F0 status word 21-27 ms instead of 8.7),
    # with a numpy copy or one torch thread 10.7-10.8.  The host-side work of this job is small copies: nothing here wants a thread team.
    shards = [_Shard(w, settings.batch_size,
                     TargetSelector(target_selection_algorithm, possible_targets, source_utt2spk,
                                    getattr(settings, "target_constant_spkid", "?"), rng_state),
                     torch.cuda.Stream(device=device) if use_streams else None) for w in wavscps]

    nj = max(1, min(int(settings.data_loader_nj), 18))
    # the reference's DataLoader workers (pipeline.py:175): `nj` threads decode files (file IO and the numpy
    # conversions release the GIL), one thread per job assembles its batches, two batches ahead of the GPU
    file_pool = ThreadPoolExecutor(max_workers=nj)
    readers = ThreadPoolExecutor(max_workers=max(1, len(wavscps)))
    writers = ThreadPoolExecutor(max_workers=4)
    depth = 2

    # (SATOOLS_AMD_PIPELINE_PCM16=0: f32 on both sides of PCIe, conversions on the host threads — the A/B switch of tools/bench_pipeline.py)
    defer_f0_status = use_streams and os.environ.get("SATOOLS_AMD_PIPELINE_DEFER_STATUS", "1") != "0"
    _pcm_bits = int(os.environ.get("SATOOLS_AMD_PIPELINE_PCM16", "3"))
    keep_pcm16, out_pcm16 = bool(_pcm_bits & 1), bool(_pcm_bits & 2)

    def read_one(item):
        utid, entry = item
        entry = str(entry).strip()
        if keep_pcm16 and not entry.endswith("|"):
            fast = read_pcm16_mono(entry)
            if fast is not None:
                return {"utid": utid, "pcm": fast[0], "audio": None, "f0": None, "freq": fast[1]}
        audio, freq = load_wav_from_scp(entry)
        return {"utid": utid, "pcm": None, "audio": audio, "f0": None, "freq": freq}

    def read_batch(chunk):
        items = list(file_pool.map(read_one, chunk))
        if all(i["pcm"] is not None for i in items):
            return collate_pcm16(items)             # the usual case: 16-bit mono files stay int16 up to the device
        for i in items:
            if i["audio"] is None:
                i["audio"] = torch.from_numpy(i["pcm"].astype(np.float32) * np.float32(1.0 / 32768.0)).unsqueeze(0)
        return collate_fn(items)

    def staging(slot, kind, like_numel, dtype):
        # page-locked staging buffers are expensive to create (~100 ms for 10 MB): a ring of three slots per job, a buffer per
        # (direction, sample type) in each, reused once the writer that reads the slot's output has finished
        buf = slot["buf"].get((kind, dtype))
        if buf is None or buf.numel() < like_numel:
            buf = slot["buf"][(kind, dtype)] = torch.empty(like_numel, dtype=dtype, pin_memory=True)
        return buf[:like_numel]

    def write_batch(wav_conv, done_event, utid, freq, original_len, f0_status=None, refresh=None):
        if done_event is not None:
            done_event.synchronize()
        if f0_status is not None:
            f0_status.check()                       # what convert() raises for a batch YAAPT cannot track (deferred: see below)
            # ... and the batch's other deferred work: utterances whose VQ decision was a near-tie are decided again on the exact kernels,
            # and where their indices changed their rows were generated again (anonymizer.ConvertStatus) — copied to the host again
            if getattr(f0_status, "rows", None) and refresh is not None:
                refresh(f0_status.rows)
        arr = wav_conv.detach().numpy()             # f32, or int16 already converted on the device (sat_pcm16_from_f32)
        for i in range(arr.shape[0]):
            wav = arr[i]
            if wav.ndim == 1:
                wav = wav[None, :]
            write_riff_pcm16(results_dir / f"{utid[i]}.wav", pcm16_of(wav[:, :int(original_len[i])]), freq)

    pending_writes, n_done = [], 0
    import collections
    recent_status = collections.deque(maxlen=8)
    scp_lines = [[] for _ in shards]
    # round-robin over the jobs of this device: one batch of each in flight, each on its own stream; the next
    # batches are being read meanwhile
    prefetch = [[] for _ in shards]

    def top_up(si):
        while len(prefetch[si]) < depth and (c := shards[si].next_keys()):
            prefetch[si].append(readers.submit(read_batch, c))

    for si in range(len(shards)):
        top_up(si)
    timing = os.environ.get("SATOOLS_AMD_PIPELINE_TIMING") == "1"     # diagnostic: where the launching thread waits
    import time as _time
    t_wait_read = t_wait_slot = t_launch = t_convert = 0.0
    with torch.no_grad():
        while any(prefetch):
            for si, sh in enumerate(shards):
                if not prefetch[si]:
                    continue
                _t0 = _time.perf_counter()
                audio, _f0, original_len, utid, freq = prefetch[si].pop(0).result()
                t_wait_read += _time.perf_counter() - _t0
                top_up(si)
                _t1 = _time.perf_counter()
                targets = sh.selector(utid)
                kw = {"target": targets} if len(targets) != 0 else {}
                ctx = torch.cuda.stream(sh.stream) if sh.stream is not None else _null()
                with ctx:
                    is_pcm = isinstance(audio, np.ndarray)          # (collate_pcm16: int16 [B, n_max])
                    slot = None
                    if use_streams:
                        slot = sh.ring[sh.ring_pos % len(sh.ring)]
                        sh.ring_pos += 1
                        if slot["busy"] is not None:
                            _t2 = _time.perf_counter()
                            slot["busy"].result()
                            t_wait_slot += _time.perf_counter() - _t2
                        src = audio if is_pcm else audio.numpy()
                        pin = staging(slot, "in", src.size, torch.int16 if is_pcm else torch.float32).view(src.shape)
                        np.copyto(pin.numpy(), src)      # (NOT pin.copy_(): see the note on intra-op threads above)
                        x = pin.to(device, non_blocking=True)
                        if is_pcm:
                            x = _ops().pcm16_to_f32(x)
                    else:
                        x = (torch.from_numpy(audio.astype(np.float32) * np.float32(1.0 / 32768.0)) if is_pcm else audio).to(device)
                    fused = f0_mode == "per_utterance" and hasattr(model, "convert_padded") and len(targets) != 0
                    if f0_mode == "per_utterance" and not fused:
                        if hasattr(model, "get_f0_ragged"):
                            f0 = model.get_f0_ragged(x, original_len)
                        else:
                            tracks = [None] * len(utid)
                            by_len = {}
                            for i, n in enumerate(original_len.tolist()):
                                by_len.setdefault(n, []).append(i)
                            for n, idx in by_len.items():
                                f0g = model.get_f0(x[idx, :n])
                                for j, i in enumerate(idx):
                                    tracks[i] = f0g[j]
                            f0 = torch.zeros(len(utid), max(t.shape[-1] for t in tracks), dtype=torch.float32, device=tracks[0].device)
                            for i, t in enumerate(tracks):
                                f0[i, :t.shape[-1]] = t.reshape(-1)
                        model.set_f0(f0.to(device))
                    elif f0_mode not in ("per_utterance", "batch"):
                        raise ValueError(f"unknown f0_mode {f0_mode!r}")
                    _t3 = _time.perf_counter()
                    f0_status = None
                    if fused and defer_f0_status:
                        # the launching thread does not wait for YAAPT's status word of every batch (a round trip to the GPU that kept
                        # two kernels in flight where bench.py's loop keeps four): the writer checks it before it writes the files, the
                        # ring of staging slots is what bounds how far the launches run ahead
                        wav_conv, f0_status = model.convert_padded(x, original_len.tolist(), targets, defer_status=True)
                    else:
                        wav_conv = model.convert_padded(x, original_len.tolist(), targets) if fused else model.convert(x, **kw)
                    t_convert += _time.perf_counter() - _t3
                    refresh = None
                    if use_streams:
                        # the samples the files hold are made on the device: half the bytes to copy back, nothing to round on the host
                        dev_f32 = wav_conv
                        to_pcm = out_pcm16 and wav_conv.is_cuda and wav_conv.dtype == torch.float32
                        if to_pcm:
                            wav_conv = _ops().pcm16_from_f32(wav_conv)
                        host = staging(slot, "out", wav_conv.numel(), wav_conv.dtype).view(wav_conv.shape)
                        host.copy_(wav_conv, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(sh.stream)
                        if f0_status is not None:
                            def refresh(rows, dev_f32=dev_f32, host=host, stream=sh.stream, to_pcm=to_pcm):
                                # rows of the device result rewritten behind the copy above (rare: a few utterances per thousand)
                                with torch.cuda.stream(stream):
                                    part = dev_f32[rows].contiguous()
                                    if to_pcm:
                                        part = _ops().pcm16_from_f32(part)
                                    host[rows] = part.cpu()
                    else:
                        if f0_status is not None:
                            f0_status.check()       # (no streams: the copy below is synchronous anyway; the deferred work first, it may rewrite rows)
                        host, ev = wav_conv.cpu(), None
                t_launch += _time.perf_counter() - _t1
                if f0_status is not None and hasattr(f0_status, "start"):
                    # batches launched before this one: where a batch's VQ launch has completed (a query, no wait), its near-tie utterances go
                    # to the exact kernels now, so that the writer's check() finds them decided (anonymizer.ConvertStatus.start)
                    for st in recent_status:
                        st.start()
                    recent_status.append(f0_status)
                fut = writers.submit(write_batch, host, ev, utid, freq[0], original_len, f0_status, refresh)
                if slot is not None:
                    slot["busy"] = fut
                pending_writes.append(fut)
                for u in utid:
                    scp_lines[si].append(f"{u} {results_dir / f'{u}.wav'}\n")
                n_done += len(utid)
                if progress is not None:
                    with progress.get_lock():
                        progress.value += len(utid)
    for w in pending_writes:
        w.result()
    if timing:
        print(f"[pipeline timing] launching thread: waited {t_wait_read:.3f} s for readers, {t_wait_slot:.3f} s for a free "
              f"staging slot (writers / GPU), {t_launch - t_wait_slot - t_convert:.3f} s in staging, {t_convert:.3f} s inside convert (launches), {n_done} utterances", flush=True)
    readers.shutdown()
    file_pool.shutdown()
    writers.shutdown()
    # like the reference, each job (re)writes wav.scp of the output dir with ITS utterances; with several jobs
    # per process the lines are concatenated in job order
    with open(scp_out if scp_out is not None else output_path / "wav.scp", "wt", encoding="utf-8") as writer:
        for lines in scp_lines:
            writer.writelines(lines)
    return n_done


def _ops():
    from . import ops          # (loads libsatools_hip.so: only a job on a GPU gets here)
    return ops


class _TorchThreads:
    """torch.set_num_threads(n) for the duration of a block (n = None: leave it)"""

    def __init__(self, n):
        self.n, self.prev = n, None

    def __enter__(self):
        if self.n is not None and torch.get_num_threads() != self.n:
            self.prev = torch.get_num_threads()
            torch.set_num_threads(self.n)
        return self

    def __exit__(self, *a):
        if self.prev is not None:
            torch.set_num_threads(self.prev)
        return False


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False
