"""The anonymizer `Net` behind the reference's model-config interface
(egs/vc/libritts/local/tuning/hifigan.py:19-131): convert / extract_features / _forward /
get_bn / get_f0 / set_f0 / get_spk_id / f0_transformation, attributes spk, utt2spk,
bn_extractor, hifigan, f0_yaapt_opts.  Every tensor op of the path runs in libsatools_hip.so."""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, f0_transforms, ops
from .hifigan import CoreHifiGan


class SimpleNamespace:
    """attribute bag whose missing attributes read as None
    (satools/satools/utils/general.py:9-32: `args.f0_transformation` may be absent)"""

    def __init__(self, /, **kwargs):
        self.__dict__.update(kwargs)

    def __getattr__(self, key):
        return self.__dict__.get(key)

    def __getitem__(self, key):
        return self.__dict__.get(key)

    def __repr__(self):
        return "SimpleNamespace({})".format(", ".join(f"{k}={v!r}" for k, v in self.__dict__.items()))


class ConvertStatus:
    """what `convert(..., defer_status=True)` hands back beside y: `check()` raises what the plain call would have raised (YAAPT's
    status word) and finishes the call's deferred work — utterances whose VQ decision was a near-tie of the split-f16 arithmetic are
    decided again on the exact-f32 kernels and their rows of y replaced (asrbn._TdnnfBase.vq_tie_sigmas).  The caller uses y after
    `check()`; when rows were replaced, `check()` returns after they have been written."""

    def __init__(self, f0_status, fix, patch, stream, deferred):
        import threading
        self.f0_status, self.fix, self.patch, self.stream, self.deferred = f0_status, fix, patch, stream, deferred
        self.rows = None
        self._lock = threading.Lock()          # start() from a launching thread, check() from the thread that consumes y

    def start(self):
        """non-blocking: begin the second decision of the batch's near-tie utterances if its VQ launch has completed (asrbn.TieFix.start);
        a loop that keeps several batches in flight calls it a step or two before `check()`, which then finds the work done"""
        if self.fix is not None and self.rows is None and self._lock.acquire(blocking=False):
            try:
                if self.fix is not None and self.rows is None:
                    with torch.cuda.stream(self.stream):
                        return self.fix.start()
            finally:
                self._lock.release()
        return True

    def check(self):
        if self.f0_status is not None:
            self.f0_status.check()
        with self._lock:
            self._check_locked()

    def _check_locked(self):
        if self.rows is None:
            self.rows = []
            if self.fix is not None:
                with torch.cuda.stream(self.stream):
                    rows = self.fix.finish()
                    if rows:
                        self.patch(rows)
                        self.rows = rows
                if self.rows and self.deferred:      # (the plain call returns y with the rows' launches enqueued in order on its stream)
                    self.stream.synchronize()
            self.fix = self.patch = None


def build(args):
    """factory with the reference's signature: build(args) -> Net class (hifigan.py:18-131)"""
    from . import infer_helper

    class Net(nn.Module):
        def __init__(self, utt2spk):
            super().__init__()
            self.bn_extractor_model = args.asrbn_model
            self._build_args = dict(args.__dict__)          # what `build(args)` was called with (frozen export)
            self.bn_extractor = infer_helper.load_model(self.bn_extractor_model, from_file=__file__, load_weight=False)
            self.bn_extractor.eval()
            self.f0_yaapt_opts = {
                "frame_length": 35.0,
                "frame_space": 20.0,
                "nccf_thresh1": 0.25,
                "tda_frame_length": 25.0,
            }
            self.utt2spk = utt2spk
            self.spk = sorted(set([v for v in utt2spk.values()]))
            self.f0: Optional[torch.Tensor] = None
            self._defer_f0_status, self._f0_status, self._f0_stream = False, None, None
            self._bn_fix, self._keep_ctx, self._fwd_ctx = None, False, None
            self.hifigan = CoreHifiGan(
                imput_dim=256 + 1 + len(self.spk),
                upsample_rates=[5, 4, 4, 2, 2],
                upsample_kernel_sizes=[11, 8, 8, 4, 4],
            )

        # ---- nn.Module surface quirks kept from the reference --------------------------------
        def remove_weight_norm(self):
            self.hifigan.remove_weight_norm()

        def train(self, mode=True):
            # hifigan.py:54-56: keeps the extractor in eval and returns None (so .eval() too)
            super().train(mode)
            self.bn_extractor.eval()

        def _apply(self, fn, *a, **k):
            # `.to("cuda")` of a model that infer_helper.load_model built from a real checkpoint: the precision guard runs ONCE, on
            # the first move to the device (SATOOLS_AMD_CHECK_PRECISION=0 skips it) — the reduced-precision defaults ("f16f8r" generator,
            # split-f16 extractor) were measured on seeded-random weights; a trained checkpoint gets its own calibration run, the
            # report is logged and kept (`precision_report`), and a part that fails it falls back (check_precision)
            out = super()._apply(fn, *a, **k)
            if self.__dict__.get("_precision_check_pending") and self._device().type == "cuda":
                self.__dict__["_precision_check_pending"] = False
                import logging
                import os
                if os.environ.get("SATOOLS_AMD_CHECK_PRECISION", "1") not in ("0", "false", "no"):
                    was_training = self.training
                    rep = self.check_precision()
                    self.__dict__["precision_report"] = rep
                    logging.getLogger("satools_amd").info("precision guard on the loaded checkpoint: %s", rep)
                    if was_training:
                        super().train(True)
            return out

        def _device(self):
            return next(self.hifigan.parameters()).device

        def _to_device(self, t):
            dev = self._device()
            if dev.type != "cuda":
                raise _lib.SatError("the model is on the CPU: call .to('cuda') first "
                                    "(the MI355X path has no CPU fallback)")
            return t.to(dev)

        # ---- feature extractors (decorators are pass-through under SA_JIT_TWEAK=true,
        #      utils/feature_extractor_decorator.py:60-71; parse_wavinfo_wav clones) -------------
        def get_bn(self, wavinfo, defer_ties=False):
            """defer_ties=True (inside convert()): -> (bn, fix) — `fix` (asrbn.TieFix or None) decides the batch's near-tie utterances again
            on the exact kernels (`start()` non-blocking, `finish()` -> the rows of bn it rewrote), once the generator is enqueued"""
            wav = self._to_device(getattr(wavinfo, "wav", wavinfo).detach())
            # parse_wavinfo_wav clones because the extractor scales its argument in place (wav_scp_dataset.py:48-53);
            # the private entry leaves the input untouched instead (no clone, no scaling pass)
            private = getattr(self.bn_extractor, "_extract_bn_private", None)
            if private is not None:
                if defer_ties:
                    bn, fix = private(wav, defer_ties=True)
                    return bn.permute(0, 2, 1), fix
                return private(wav).permute(0, 2, 1)
            bn = self.bn_extractor.extract_bn(wav.clone()).permute(0, 2, 1)
            return (bn, None) if defer_ties else bn

        def set_f0(self, f0):
            self.f0 = f0

        def get_f0(self, wavinfo):
            # the reference computes on the CPU and returns on the input's device (yaapt.py:798-799,940);
            # here the whole batch is tracked on the GPU
            from . import f0 as f0_hip
            wav = getattr(wavinfo, "wav", wavinfo).detach()
            out = f0_hip.yaapt(self._to_device(wav), self.f0_yaapt_opts)
            return out.to(wav.device)

        def get_f0_ragged(self, wav, lengths):
            """F0 of zero-padded utterances at their own lengths: [B, n_max] + lengths -> [B, T_max] zero-padded,
            on the device — the reference's data loader semantics (bin/pipeline.py:35-41, :52-62: `get_f0` per
            utterance, tracks zero-padded by the collate) in one launch sequence (sat_yaapt_ragged_f32)."""
            from . import f0 as f0_hip
            xd = self._to_device(wav.detach())
            return f0_hip.yaapt_ragged(xd, lengths, self.f0_yaapt_opts)

        def convert_padded(self, x, lengths, target, defer_status=False):
            """convert() of a zero-padded batch whose F0 tracks are taken per utterance at its own length — the
            result of the reference's batch job (`set_f0` of the data loader's zero-padded per-utterance tracks,
            then `convert`; bin/pipeline.py:35-62, :107-149) — with the ragged YAAPT launch on the F0 side stream
            next to the bottleneck extractor and its status checked after the generator is enqueued.
            defer_status=True: returns (y, status) without waiting for YAAPT's status word; the caller runs `status.check()` before it
            uses y (it raises what this call would have raised: the batch job does so in the thread that writes the files, so that the
            launching thread never waits for the GPU)."""
            from . import f0 as f0_hip
            xd = self._to_device(x.detach())
            cur = torch.cuda.current_stream(xd.device)
            if self._f0_stream is None:
                self._f0_stream = {}
            side = self._f0_stream.get(cur.cuda_stream)
            if side is None:
                side = self._f0_stream[cur.cuda_stream] = torch.cuda.Stream(device=xd.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                f0, st = f0_hip.yaapt_ragged(xd, lengths, self.f0_yaapt_opts, defer_status=True)
                f0 = f0.unsqueeze(0)
            xd.record_stream(side)
            bn, fix = self.get_bn(x, defer_ties=True)
            cur.wait_stream(side)
            f0.record_stream(cur)
            self._keep_ctx = fix is not None
            try:
                y = self._forward(f0, bn, self.get_spk_id(x, target))
            finally:
                self._keep_ctx = False
            return self._finish(y, st, fix, defer_status)

        def _finish(self, y, f0_status, fix, defer_status):
            """y [B, 1, n'] of _forward -> what convert() returns; the deferred work (YAAPT's status word, the near-tie utterances of the
            VQ) done here or handed to the caller as a ConvertStatus"""
            ctx, self._fwd_ctx = self._fwd_ctx, None
            out = y.squeeze(0)

            def patch(rows):
                bn, f0_d, spk, arith = ctx
                xs = ops.assemble_input(bn[rows].contiguous(), f0_d[rows].reshape(len(rows), -1).contiguous(), spk[rows].contiguous(), spk.shape[1])
                # in the arithmetic the batch ran: a few rows are too small a batch for the ring kernel's default dispatch and would run
                # "f16x3" where the batch ran "f16f8r" — a row whose indices did not change then keeps its bits
                f8 = (arith or "").startswith("f16f8r") and not self.hifigan.force_f8
                if f8:
                    self.hifigan.set_force_f8(1)
                try:
                    ys, _ = self.hifigan(xs)
                finally:
                    if f8:
                        self.hifigan.set_force_f8(0)
                y[rows] = ys.to(torch.float32)

            st = ConvertStatus(f0_status, fix, patch if fix is not None else None, torch.cuda.current_stream(y.device), bool(defer_status))
            if defer_status:
                return out, st
            st.check()
            return out

        def check_precision(self, wav=None, tol=2e-5, fallback=True):
            """Load-time guard for the split-f16 arithmetic (f32 operands carried as hi + lo f16): run a
            calibration utterance through the generator and the bottleneck extractor twice — as configured, and on the
            exact-f32 MFMA kernels — and compare.  A checkpoint whose activations leave the f16 range (|x| >= 65504
            saturates hi) or sit far below it (the split is only exact to an absolute 2^-24) shows up as a difference the
            reference's plain-f32 path would not have; with `fallback` the offending part then STAYS on the exact-f32
            kernels (16 x slower matrix work, same results as the reference) instead of degrading silently.
            Returns {"generator": relative RMS difference, "bn_extractor": relative RMS difference of the bottleneck
            projections (before the VQ decision), "bn_index_agreement": fraction of VQ indices equal, "fallback": [...]}.
            `wav` [B, n] on any device (default: one synthetic 1 s voiced utterance).  Not for frozen models."""
            import warnings
            from . import synthetic
            if self.hifigan.__dict__.get("_frozen"):
                raise _lib.SatError("check_precision needs the f32 parameters: run it before export_frozen")
            dev = self._device()
            wav = synthetic.harm_batch([0], 16000) if wav is None else wav
            wav = self._to_device(wav.detach().to(torch.float32))
            out, fell = {}, []
            gen, ext = self.hifigan, self.bn_extractor

            def relrms(a, b):
                a, b = a.double(), b.double()
                d = torch.sqrt(torch.mean((a - b) ** 2))
                return float("inf") if not bool(torch.isfinite(d)) else float(d / torch.sqrt(torch.mean(b ** 2)).clamp_min(1e-30))

            with torch.no_grad():
                # ---- bottleneck extractor: compare the projections the VQ decides on, and the decisions
                cfg = {k: getattr(ext, k) for k in ("precision", "w2v2_precision") if hasattr(ext, k)}
                bn32 = None
                if any(v != "f32" for v in cfg.values()):
                    _, (z, idx, _) = ext.extract_bn(wav.clone(), want_aux=True)
                    keep_f32 = False
                    try:                                   # whatever happens in between (an OOM on the 16x slower exact kernels ...),
                        for k in cfg:                      # the configuration is restored unless the fall-back was DECIDED
                            setattr(ext, k, "f32")
                        _, (z32, idx32, d32) = ext.extract_bn(wav.clone(), want_aux=True)
                        bn32 = self.get_bn(wav)
                        out["bn_extractor"] = relrms(z, z32)
                        out["bn_index_agreement"] = float((idx == idx32).float().mean())
                        # index work is exact work: say how many decisions differ, and whether each is a near-tie of the exact kernels
                        from .asrbn import vq_flip_stats
                        st = vq_flip_stats(z, idx, z32, idx32, d32)
                        out["bn_index_flips"] = {k: st[k] for k in ("frames", "flips", "flips_per_million", "flips_outside_error_bound")}
                        if not (out["bn_extractor"] <= 5 * tol and out["bn_index_agreement"] >= 0.98) and fallback:
                            fell.append("bn_extractor")
                            keep_f32 = True
                    finally:
                        if not keep_f32:
                            for k, v in cfg.items():
                                setattr(ext, k, v)
                # ---- generator, teacher-forced on the exact-f32 extractor's features (computed above while it was configured so)
                if gen.precision != "f32":
                    f0 = self.get_f0(wav).unsqueeze(0)
                    bn = bn32 if bn32 is not None else self.get_bn(wav)
                    f0n = f0.clone()
                    ops.f0_norm_transform_(f0n)
                    spk = F.one_hot(torch.zeros(wav.shape[0], dtype=torch.long), num_classes=len(self.spk))
                    x = ops.assemble_input(bn, f0n.reshape(wav.shape[0], -1), spk.to(dev, torch.float32).contiguous(), spk.shape[1])
                    keep = gen.precision
                    final = keep
                    try:
                        gen.precision = "f32"
                        y32 = gen(x)[0]
                        if keep == "f16f8r":
                            # 8-bit cross terms on the thick stages (csrc/conv_ring16.hip): a calibration batch is too small for the ring
                            # kernel's default dispatch, so THIS handle is told to run them at every batch size (option force_f8; the
                            # process-wide dispatch options are not touched); a checkpoint whose activations or weights leave the
                            # range the 8-bit operands carry falls back to "f16x3" FIRST.  The planes the forward writes are probed
                            # on the way (sat_hifigan_set_range_probe: values past the largest e5m2 / f16, per stage)
                            gen.precision = "f16f8r"
                            gen.set_force_f8(1)
                            try:
                                y8 = gen(x)[0]
                                out["generator_arithmetic_f16f8r_ran"] = gen.last_arithmetic
                                out["generator_range"] = gen.range_probe(x)
                            finally:
                                gen.set_force_f8(0)
                            out["generator_f16f8r"] = relrms(y8, y32)
                            out["generator_f8_weights"] = gen.f8_weight_stats()
                            if (out["generator_f16f8r"] > 10 * tol or sum(out["generator_range"]["past_e5m2_max"]) > 0) and fallback:
                                fell.append("generator: f16f8r -> f16x3")
                                final = "f16x3"
                            gen.precision = "f16x3"
                        else:
                            gen.precision = keep
                        y = gen(x)[0]
                        out["generator"] = relrms(y, y32)
                        if out["generator"] > 50 * tol and fallback:          # waveform RMS ~0.1: 1e-3 relative = the path's 1e-4 bar
                            fell.append("generator")
                            final = "f32"
                    finally:
                        gen.precision = final
                        gen.invalidate()
            out["fallback"] = fell
            if fell:
                warnings.warn(f"satools_amd: {', '.join(fell)} left the range the split-f16 kernels represent on the calibration "
                              f"utterance ({out}); running on the exact-f32 kernels instead (slower, reference-exact)")
            return out

        def get_spk_id(self, wavinfo, target=None):
            if not target:
                target = [self.utt2spk[wavinfo.name]]
            return F.one_hot(torch.tensor([self.spk.index(t) for t in ([target] if isinstance(target, str) else target)]),
                             num_classes=len(self.spk))

        def extract_features(self, x, target):
            if self.f0 is not None:
                f0, self.f0 = self.f0, None
            elif self._defer_f0_status:
                # inside convert(): YAAPT's kernels are latency-bound and occupy few CUs, so they run
                # on a side stream next to the bottleneck extractor; the status word is checked after
                # the generator has been enqueued (no stall of the launch stream)
                from . import f0 as f0_hip
                xd = self._to_device(x.detach())
                cur = torch.cuda.current_stream(xd.device)
                if self._f0_stream is None:
                    self._f0_stream = {}
                side = self._f0_stream.get(cur.cuda_stream)
                if side is None:    # one F0 side stream per launch stream
                    side = self._f0_stream[cur.cuda_stream] = torch.cuda.Stream(device=xd.device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    f0, self._f0_status = f0_hip.yaapt(xd, self.f0_yaapt_opts, defer_status=True)
                    f0 = f0.unsqueeze(0)
                xd.record_stream(side)
                bn, self._bn_fix = self.get_bn(x, defer_ties=True)
                cur.wait_stream(side)
                f0.record_stream(cur)
                spk_id = self.get_spk_id(x, target)
                return (f0, bn, spk_id)
            else:
                f0 = self.get_f0(x).unsqueeze(0)
            if self._defer_f0_status:          # inside convert() (F0 from set_f0): near-tie utterances are decided again behind the generator
                bn, self._bn_fix = self.get_bn(x, defer_ties=True)
            else:
                bn = self.get_bn(x)
            spk_id = self.get_spk_id(x, target)
            return (f0, bn, spk_id)

        def convert(self, x, target, defer_status=False):
            """hifigan.py:58-71.  defer_status=True (not in the reference): returns (y, status) without waiting for the GPU — `status`
            (ConvertStatus) carries the call's deferred work: `status.check()` raises what this call would have raised (YAAPT's status
            word) and finishes the second decision of near-tie utterances of the VQ, whose rows of y it may rewrite (`status.rows`);
            `status.start()` begins that decision without blocking.  A caller that keeps several batches in flight checks a batch's
            status before it uses y instead of making round trips to the GPU inside every call"""
            self._defer_f0_status, self._f0_status, self._bn_fix = True, None, None
            try:
                (f0, bn, spk_id) = self.extract_features(x, target)
            finally:
                self._defer_f0_status = False
            fix, self._bn_fix = self._bn_fix, None
            self._keep_ctx = fix is not None
            try:
                y = self._forward(f0, bn, spk_id)
            finally:
                self._keep_ctx = False
            st, self._f0_status = self._f0_status, None
            return self._finish(y, st, fix, defer_status)

        def f0_transformation(self, f0):
            """host-level entry kept for API parity (hifigan.py:73-81); [B,1,T] device tensor"""
            spec = args.f0_transformation
            quant = f0_transforms.parse_quant_bins(spec) if spec and "quant" in spec else 0
            noise = None
            if spec and "awgn" in spec:
                noise = f0_transforms.draw_awgn(f0.shape, f0_transforms.parse_awgn_db(spec)).to(f0.dtype)
                noise = noise.pin_memory().to(f0.device, non_blocking=True) if f0.is_cuda else noise
            if quant or noise is not None:
                f0 = f0.clone()
                self._apply_transform_(f0, quant, noise)
            if spec and "mean-reverv" in spec:
                # hifigan/nn.py:64-90: the reference's moving average squeezes [B, 1, T'] to 2-D before conv1d, which
                # then reads B as the channel count of ONE sequence: batches of 1 only (RuntimeError otherwise)
                alpha, n = f0_transforms.parse_mean_reverv(spec)
                if f0.shape[0] != 1:
                    raise RuntimeError(f"Given groups=1, weight of size [1, 1, {n}], expected input[1, {f0.shape[0]}, "
                                       f"{f0.shape[-1] + 2 * (n // 2)}] to have 1 channels, but got {f0.shape[0]} channels instead "
                                       "(mean-reverv handles one utterance per call, as in the reference)")
                from ._lib import check, lib, ptr, stream
                src = f0.contiguous()
                out = torch.empty_like(src)
                check(lib().sat_f0_mean_reversion_f32(ptr(src), ptr(out), src.shape[-1], float(alpha), int(n), stream()),
                      "sat_f0_mean_reversion_f32")
                f0 = out
            return f0

        @staticmethod
        def _apply_transform_(f0, quant, noise):
            n = f0.numel()
            from ._lib import check, lib, ptr, stream
            ones = torch.tensor([0.0, 1.0], dtype=torch.float32, device=f0.device)  # identity normalisation
            check(lib().sat_f0_apply_f32(ptr(f0), n, ptr(ones), int(quant), ptr(noise), stream()), "sat_f0_apply_f32")

        def _forward(self, f0, bn, spk_id):
            dev = self._device()
            if dev.type != "cuda":
                raise _lib.SatError("the model is on the CPU: call .to('cuda') first (no CPU fallback)")
            # f0 = self.f0_norm(f0): batch-coupled, IN PLACE on the caller's tensor (cmvn.py:143-155)
            f0_in = f0
            f0_d = f0_in.to(device=dev, dtype=torch.float32)
            if not f0_d.is_contiguous():
                f0_d = f0_d.contiguous()
            shared = f0_in.is_cuda and f0_d.data_ptr() == f0_in.data_ptr()
            ops.f0_norm_transform_(f0_d)
            if not shared:
                f0_in.copy_(f0_d)  # the reference leaves the normalised values in the caller's tensor
            if f0_d.dim() == 2:
                f0_d = f0_d.unsqueeze(0)
            f0_d = f0_d.permute(1, 0, 2).contiguous()  # [B, 1, T]
            f0_d = self.f0_transformation(f0_d)
            bn = bn.to(device=dev, dtype=torch.float32).contiguous()
            B, c_bn, T = bn.shape
            if f0_d.shape[0] != B:
                raise AssertionError("f0 and bn batch sizes differ")
            if spk_id.is_cuda:
                spk = spk_id.to(device=dev, dtype=torch.float32).contiguous()
            else:
                # pinned + non_blocking: a pageable host-to-device copy would block the host until every
                # kernel already queued on this stream has finished and serialise successive convert() calls
                spk = spk_id.to(torch.float32).contiguous().pin_memory().to(dev, non_blocking=True)
            assert B == spk.shape[0], \
                "len(target) != len(input_wav), check if the waveform batch size == target=len(['6081','4214'])"
            x = ops.assemble_input(bn, f0_d.reshape(B, -1), spk, spk.shape[1])
            y, _ = self.hifigan(x)
            if self._keep_ctx:            # convert(): the rows of near-tie utterances are assembled and generated again (_finish)
                self._fwd_ctx = (bn, f0_d, spk, self.hifigan.last_arithmetic)
            return y.to(torch.float32)

        def forward(self, egs_with_feat):
            return self._forward(egs_with_feat["get_f0"], egs_with_feat["get_bn"], egs_with_feat["get_spk_id"])

    return Net
