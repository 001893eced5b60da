"""toy kaldi data dir + stand-in model shared by tests/golden/make_pipeline_fixtures.py (which drives the
REFERENCE's process_data with them) and tests/test_pipeline.py (which drives ours)"""
import os
import wave

import numpy as np
import torch


def write_wav(path, x, sr=16000):
    pcm = np.clip(np.rint(x * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def read_wav(path):
    with wave.open(str(path), "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy(), w.getframerate()


def make_dataset(root):
    """12 utterances of 4 speakers, lengths 3000..8500 samples"""
    os.makedirs(os.path.join(root, "clear"), exist_ok=True)
    rng = np.random.RandomState(0)
    lines, u2s = [], []
    for i in range(12):
        utt, spk = f"utt{i:02d}", f"src{i % 4}"
        n = 3000 + 500 * i
        x = 0.3 * np.sin(2 * np.pi * (110 + 10 * i) * np.arange(n) / 16000) + 0.01 * rng.randn(n)
        path = os.path.join(root, "clear", utt + ".wav")
        write_wav(path, x)
        lines.append(f"{utt} {path}\n")
        u2s.append(f"{utt} {spk}\n")
    open(os.path.join(root, "wav.scp"), "w").writelines(lines)
    open(os.path.join(root, "utt2spk"), "w").writelines(u2s)


class StandInModel:
    """what process_data needs from a model: spk, get_f0, set_f0, convert, to, eval"""
    calls = []

    def __init__(self):
        self.spk = [f"tgt{i:03d}" for i in range(20)]
        self._f0 = None

    def to(self, d):
        return self

    def eval(self):
        return None

    def get_f0(self, audio):
        return torch.full((audio.shape[0], audio.shape[-1] // 320), 100.0)

    def set_f0(self, f0):
        self._f0 = f0

    def convert(self, audio, target=None):
        StandInModel.calls.append({"shape": list(audio.shape), "target": target, "f0_shape": list(self._f0.shape)})
        y = torch.cat([audio * 0.5, torch.zeros(audio.shape[0], 1)], dim=1)       # [B, n + 1]
        return y if audio.shape[0] == 1 else y.unsqueeze(1)
