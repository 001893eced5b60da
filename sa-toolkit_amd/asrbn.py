"""ASR-bottleneck extractors behind the reference's `Net.extract_bn` interface.

fbank tag:    egs/asr/librispeech/local/chain/tuning/tdnnf_vq.py:20-286
wav2vec2 tag: egs/asr/librispeech/local/chain/tuning/tdnnf_wav2vec2_vq.py:21-345

A TDNNF layer (satools/satools/chain/nn.py:267-292) is `unfold` over `context_len` frames
followed by two matmuls; with activations kept channel-major [B, C, T] that is a *valid*
conv over frames (kernel = context_len, stride = subsampling_factor) into the bottleneck and a
1x1 conv out of it, whose epilogue carries the bypass `+0.66*input[:, l:r:sub]`, the eval-mode
BatchNorm1d(affine=False) and the ReLU (chain/nn.py:338-347).  Both run on the fused MFMA conv
kernel; the VQ bottleneck (chain/nn.py:402-476) has its own kernel.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib, ops, packing
from .params import TDNNFBatchNormParams, _InnerNat, tdnnf_stack


def get_padding(kernel_sizes, subsampling_factors):
    """ChainE2EModel.get_padding (satools/satools/chain/model.py:466-473)"""
    pad, g = 0, 1
    for k, s in zip(kernel_sizes, subsampling_factors):
        pad += (k - 1) * g
        g *= s
    return int(pad)


# ---- fbank tables (kaldifeature.py:144-146 povey window; :386-457 get_mel_banks) ----------
def povey_window(n=400):
    return torch.hann_window(n, periodic=False, dtype=torch.float32).pow(0.85)


def mel_banks(num_bins=80, n_fft=512, sample_freq=16000.0, low_freq=20.0, high_freq=0.0):
    """triangular mel filters [num_bins, n_fft/2 + 1] (last column zero), f32, computed with the
    same operation order as the reference so the table is bit-identical to its `bins`."""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    bin_width = sample_freq / n_fft
    mel = lambda f: 1127.0 * math.log(1.0 + f / 700.0)
    mel_lo, mel_hi = mel(low_freq), mel(high_freq)
    delta = (mel_hi - mel_lo) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left = mel_lo + b * delta
    center = mel_lo + (b + 1.0) * delta
    right = mel_lo + (b + 2.0) * delta
    m = (1127.0 * (1.0 + (bin_width * torch.arange(n_fft / 2)) / 700.0).log()).unsqueeze(0)
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    bins = torch.max(torch.zeros(1), torch.min(up, down))
    return torch.nn.functional.pad(bins, (0, 1)).to(torch.float32).contiguous()


def vq_flip_stats(z, idx, z_exact, idx_exact, dist_exact):
    """VQ decisions of one arithmetic against the exact-f32 kernels' on the same frames (chain/nn.py:424-459: argmin over the
    codebook's squared distances).  z / z_exact [B, D, T] projections before the decision, idx / idx_exact [B, T], dist_exact
    [B, T, n_codes].  A frame whose index differs ("flip") is INSIDE the bound when the exact kernels' own two best distances are
    closer than what that frame's measured feature difference dz can move them, |d_k(z + dz) - d_k(z)| <= 2 |dz| sqrt(d_k) + |dz|^2
    for both candidates — a near-tie, not an error of the arithmetic.  Returns counts (python ints / floats)."""
    idx, idx_exact = idx.reshape(-1).long(), idx_exact.reshape(-1).long()
    flips = torch.nonzero(idx != idx_exact).flatten()
    out = {"frames": int(idx.numel()), "flips": int(flips.numel()), "flips_per_million": 1e6 * flips.numel() / max(1, idx.numel()),
           "flips_outside_error_bound": 0, "flip_frames": flips.tolist()[:64]}
    if flips.numel():
        d2 = torch.sort(dist_exact.reshape(-1, dist_exact.shape[-1]).double()[flips], dim=1)[0].clamp_min(0)
        D = z.shape[1]
        dz = (z.permute(0, 2, 1).reshape(-1, D)[flips].double() - z_exact.permute(0, 2, 1).reshape(-1, D)[flips].double()).norm(dim=1)
        bound = 2 * dz * (d2[:, 0].sqrt() + d2[:, 1].sqrt()) + 2 * dz ** 2
        gap = d2[:, 1] - d2[:, 0]
        out["flips_outside_error_bound"] = int((gap > bound).sum())
        out["largest_gap_over_bound"] = float((gap / bound.clamp_min(1e-300)).max())
    return out


def tie_windows(Tf, S, W, spans):
    """Input windows for the second decision of near-tie utterances: output frame t of a stack of 'valid' windows reads the input frames
    S t .. S t + W - 1, so the output frames lo .. hi of an utterance need the inputs S lo .. S hi + W - 1.  -> (L, starts): ONE window
    length for all rows (the longest need, rounded so that Tf - L is a multiple of S) and per row the first input frame, a multiple of
    S, moved left where the window would pass the end of the Tf input frames; the run on inputs starts[i] .. starts[i] + L - 1 yields
    the output frames starts[i] / S .. (starts[i] + L - W) / S, which contain lo .. hi.  A need of three quarters of the utterance or more
    returns the whole utterance (L = Tf, starts 0)."""
    need = max(S * (hi - lo) + W for lo, hi in spans)
    need += (Tf - need) % S
    if need * 4 >= Tf * 3:
        return Tf, [0] * len(spans)
    return need, [min(S * lo, Tf - need) for lo, _ in spans]


class TieStatus:
    """Deferred near-tie report of one VQ launch (`sat_vq_argmin_gather_tie_f32`): the per-utterance counts travel to a page-locked
    row behind the launch, `rows()` waits for them (the VQ runs before the generator: by the time a caller has enqueued the rest of
    `convert()` they have usually landed) and names the utterances to decide again on the exact-f32 kernels."""

    def __init__(self, counts_dev, idx_dev):
        from .f0 import _pinned_ints
        self._pool = _pinned_ints
        self.B = counts_dev.shape[1]
        self.host, self.row = _pinned_ints.take(counts_dev.numel())
        self.host.copy_(counts_dev.reshape(-1), non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.counts = None
        self.idx = idx_dev             # [B, T'] indices of the arithmetic as configured (device)

    def rows(self):
        if self.counts is None:
            self.event.synchronize()
            self.counts = self.host.clone().reshape(3, self.B)      # counts | first | last near-tie frame
            self.host = None
            self._pool.give(self.row)
            self.row = None
        return torch.nonzero(self.counts[0]).flatten().tolist()

    def spans(self, rows):
        """(first, last) near-tie frame of each of `rows` (after rows())"""
        return [(int(self.counts[1, r]), int(self.counts[2, r])) for r in rows]

    def __del__(self):
        try:
            if self.row is not None:
                self.event.synchronize()
                self._pool.give(self.row)
        except Exception:
            pass


class TieFix:
    """The second decision of a batch's near-tie utterances, in two steps so that a caller with several batches in flight never waits:
    `start()` (non-blocking: returns False while the batch's VQ launch has not completed) sends the flagged utterances through the
    exact-f32 kernels on a side stream that waits for that launch only — not for what the caller enqueued behind it (the generator);
    `finish()` (blocking, starts what `start()` has not) waits for the side stream, and where an utterance's indices CHANGED rewrites
    its rows of `bn` (and of `idx`) on the stream current at that time.  -> the changed rows.  A flagged utterance costs its
    extractor again (2 % of 5 s utterances at the default window), a changed one (a few per thousand) also what the caller derives
    from its rows (`convert()` generates them again)."""

    def __init__(self, ext, status, bn, feats, wav, idx=None):
        self.ext, self.status, self.bn, self.feats, self.wav, self.idx = ext, status, bn, feats, wav, idx
        ext.__dict__.setdefault("tie_stats", {"utterances": 0, "rerun": 0, "changed": 0})
        self.stage, self.hit = (2, []) if status is None else (0, None)
        self.rows = self.zq = self.idx_x = self.t0 = self.flags = self.flag_row = self.event = self.side = None

    def start(self, block=False):
        if self.stage:
            return True
        if not block and not self.status.event.query():
            return False
        ext = self.ext
        rows = self.status.rows()
        if not rows:
            self.stage, self.hit = 2, []
            self._drop()
            return True
        dev = self.feats.device
        cur = torch.cuda.current_stream(dev)
        sides = ext.__dict__.setdefault("_tie_streams", {})
        side = sides.get(cur.cuda_stream)
        if side is None:
            # (normal priority: a HIGH-priority stream, once created, left every later launch of the process slower — generator alone
            # 7.7 -> 10.7 -> 12.0 ms over the configs of one bench.py run, DESIGN toolchain note 22)
            side = sides[cur.cuda_stream] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("SATOOLS_AMD_VQ_TIE_STREAM_PRIORITY", "0")))
        side.wait_event(self.status.event)              # feats, wav and the indices precede the VQ launch's event on the batch's stream
        from .f0 import _pinned_ints
        self.flags, self.flag_row = _pinned_ints.take(len(rows))
        # (the exact run takes the extractor's arithmetic lock while it launches: a launching thread and a writer thread that
        # finishes another batch's decision do not meet half way; nothing in it asks for the guard again)
        with torch.cuda.stream(side):
            self.zq, self.idx_x, self.t0 = ext._exact_rows(rows, self.feats, self.wav, spans=self.status.spans(rows))
            Lq = self.idx_x.shape[1]
            if all(t == 0 for t in self.t0) and Lq == self.status.idx.shape[1]:
                changed = (self.idx_x != self.status.idx[rows]).any(dim=1)
            else:
                changed = torch.stack([(self.idx_x[i] != self.status.idx[r, t:t + Lq]).any() for i, (r, t) in enumerate(zip(rows, self.t0))])
            self.flags.copy_(changed.to(torch.int32), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(side)
        self.rows, self.side, self.stage = rows, side, 1
        return True

    def _drop(self):
        if self.flag_row is not None:
            from .f0 import _pinned_ints
            _pinned_ints.give(self.flag_row)
            self.flag_row = None
        self.feats = self.wav = self.status = None

    def finish(self):
        if self.stage == 0:
            self.start(block=True)
        if self.stage == 1:
            ext, rows = self.ext, self.rows
            self.event.synchronize()
            flags = [1] * len(rows) if ext.vq_tie_force_patch else self.flags.tolist()
            st = ext.__dict__.setdefault("tie_stats", {"utterances": 0, "rerun": 0, "changed": 0})
            st["rerun"] += len(rows)
            sel = [i for i, c in enumerate(flags) if c]
            self.hit = [rows[i] for i in sel]
            st["changed"] = st.get("changed", 0) + len(self.hit)
            if self.hit:
                cur = torch.cuda.current_stream(self.bn.device)
                cur.wait_stream(self.side)
                self.zq.record_stream(cur)
                self.idx_x.record_stream(cur)
                Lq = self.idx_x.shape[1]
                for i in sel:                 # (the frames the exact run computed: the whole utterance, or the window around its near-ties)
                    r, t = rows[i], self.t0[i]
                    self.bn[r, t:t + Lq] = self.zq[i].t()
                    if self.idx is not None:
                        self.idx[r, t:t + Lq] = self.idx_x[i]
            self.stage = 2
            self.zq = self.idx_x = None
            self._drop()
        return self.hit

    def __del__(self):
        try:
            if self.stage == 1 and self.event is not None:
                self.event.synchronize()
            self._drop()
        except Exception:
            pass


class _TieCtx:
    """what the VQ layer of a guarded run is handed (pair distances, window, per-utterance counts) and hands back (its indices)"""
    __slots__ = ("pair", "scale", "counts", "idx")

    def __init__(self, pair, scale, counts):
        self.pair, self.scale, self.counts, self.idx = pair, scale, counts, None


class _LayerCache:
    """device-side, kernel-ready form of one TDNNFBatchNorm layer"""
    __slots__ = ("wB", "bB", "wA", "bA", "scale", "shift", "codebook", "modeA", "modeB")


class _TdnnfBase(nn.Module):
    """shared machinery of the two ASR-BN nets: TDNNF stack forward on the HIP kernels"""

    #: arithmetic of the TDNNF matrix products: "f16x3" (split-f16 on the f16 matrix cores, ~2^-21 relative per
    #: product — the size of the f32 re-association differences between any two f32 implementations, which is
    #: what decides a VQ arg-min near a tie either way; VQ indices identical to the reference's on every fixture
    #: frame with both settings, tests/test_hip_parity.py) or "f32" (exact f32 MFMA, 0.8 ms per batch slower)
    precision = os.environ.get("SATOOLS_AMD_TDNNF_PRECISION", "f16x3")

    #: near-tie guard of the VQ decision (chain/nn.py:424-459 is INDEX work: the bar is exact).  In split-f16 arithmetic a frame whose
    #: two best codes lie closer than this many standard deviations of the arithmetic's own (calibrated) feature error is counted
    #: on the device, and its utterance is decided again on the exact-f32 kernels — so the default arithmetic returns the exact
    #: kernels' indices (536-utterance sweep per tag: tests/test_hip_robust.py).  0 switches the guard off; frozen models (no f32
    #: parameters to fall back on) run without it.
    vq_tie_sigmas = float(os.environ.get("SATOOLS_AMD_VQ_TIE_SIGMAS", "4"))

    def _init_cache(self):
        self._cache = None
        self._cache_key = None

    def _stack_layers(self):
        return [self.tdnn1] + [m for m in self.tdnnfs if isinstance(m, TDNNFBatchNormParams)]

    def _param_key(self):
        # the module tree is fixed after construction: walk it once, then only look at the tensors (this runs on
        # every call).  `.to()` and `load_state_dict` keep the Parameter objects and show up in data_ptr / _version;
        # `.to()` REPLACES buffer tensors, so buffers are looked up through their owning module every time
        flat = self.__dict__.get("_flat_params")
        if flat is None:
            flat = self.__dict__["_flat_params"] = (list(self.parameters()),
                                                    [(m, n) for m in self.modules() for n in m._buffers])
        ps, bufs = flat
        return tuple((p.data_ptr(), p._version) for p in ps) + tuple(
            (t.data_ptr(), t._version) for t in (m._buffers[n] for m, n in bufs) if t is not None)

    def _layer_cache(self, lay, device, split):
        pack = packing.pack_conv_weight_f16x3 if split else packing.pack_conv_weight
        c = _LayerCache()
        wB = lay.tdnn.linearB.inner_nat.weight.detach().to(device=device, dtype=torch.float32)
        bott, kin = wB.shape
        ctx, feat = lay.context_len, lay.feat_dim
        assert kin == ctx * feat
        # unfold window = ctx consecutive frames of `feat` values: column j*feat + c
        sub = lay.subsampling_factor
        # split-f16 kernels are stride 1: a subsampling layer without context (ctx 1, sub > 1) is a 1x1 conv on
        # every sub-th frame (sub = 2), or on the half-frame-shifted windows of sub = 1.5, run on a gathered copy
        c.modeB = 1 if (split and ctx in (1, 2, 3) and (sub == 1 or ctx == 1)) else 0
        c.wB = (pack if c.modeB else packing.pack_conv_weight)(wB.reshape(bott, ctx, feat).permute(0, 2, 1).contiguous())
        c.bB = lay.tdnn.linearB.inner_nat.bias.detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
        wA = lay.tdnn.linearA.weight.detach().to(device=device, dtype=torch.float32)
        c.modeA = 1 if split else 0
        c.wA = pack(wA.unsqueeze(-1).contiguous())
        c.bA = lay.tdnn.linearA.bias.detach().to(device=device, dtype=torch.float32).contiguous()
        mean = lay.bn.running_mean.detach().to(device=device, dtype=torch.float32)
        var = lay.bn.running_var.detach().to(device=device, dtype=torch.float32)
        # eval BatchNorm1d(affine=False): y = x*invstd + (-mean*invstd), eps = 1e-5
        c.scale = (1.0 / torch.sqrt(var + 1e-5)).contiguous()
        c.shift = (-mean * c.scale).contiguous()
        c.codebook = None
        if hasattr(lay, "bottleneck_func"):
            c.codebook = lay.bottleneck_func.quant._embedding.weight.detach().to(
                device=device, dtype=torch.float32).contiguous()
        return c

    def _prepare(self, device):
        if self.__dict__.get("_frozen"):           # caches installed by frozen.load_frozen
            return
        key = (self.precision,) + self._param_key()
        if self._cache_key == key:
            return
        # one cache per precision is kept (the near-tie guard runs a few utterances of most batches on the exact kernels: the
        # switch must not fold and pack the weights again)
        store = self.__dict__.setdefault("_cache_store", {})
        if self._cache_key is not None:
            store[self._cache_key[0]] = (self._cache_key, self._cache, self._cache_full)
        hit = store.get(self.precision)
        if hit is not None and hit[0] == key:
            self._cache_key, self._cache, self._cache_full = hit
            return
        split = self.precision == "f16x3"
        _lib.cache_rebuild_begin(device, hit is not None)
        self._cache = [self._layer_cache(lay, device, split) for lay in self._stack_layers()]
        self._cache_full = None
        self._cache_key = key
        _lib.cache_rebuild_end(device)

    #: inside the bottleneck stack the layers hand on split planes ONLY (no f32 store of a layer's output, the bypass rebuilt from the input
    #: planes; sat_tdnnf_layer_f32 with x = y = NULL): SATOOLS_AMD_TDNNF_PLANES_ONLY=0 keeps the f32 tensors
    tdnnf_planes_only = os.environ.get("SATOOLS_AMD_TDNNF_PLANES_ONLY", "1") != "0"

    @staticmethod
    def _takes_planes(lay, c, channels, need_z=False):
        """does `_tdnnf_layer` read this layer's input from split planes alone (never the f32 tensor)?  A plain layer does (linearB and,
        inside sat_tdnnf_layer_f32, the bypass); a subsampling layer only when it has no bypass (its strided bypass reads the f32 input)"""
        sub = lay.subsampling_factor
        if c.modeB != 1 or c.modeA != 1 or channels % 16 or sub == 1.5:
            return False
        if int(sub) == 1:
            # (the layer with the quantiser inside / the bottleneck read-out multiplies the planes and has no bypass to take from f32)
            return need_z or (c.codebook is None and lay.bottleneck_dim % 16 == 0 and lay.out_dim % 16 == 0)
        return (not need_z and not lay.use_bypass and channels % 64 == 0 and lay.bottleneck_dim % 16 == 0 and c.codebook is None)

    def _plain_on_planes(self, lay, c, channels):
        """a layer `_tdnnf_layer` serves by sat_tdnnf_layer_f32 with planes in, planes out and the bottleneck as planes"""
        return (int(lay.subsampling_factor) == 1 and lay.subsampling_factor != 1.5 and c.codebook is None and c.modeA == 1 and c.modeB == 1 and
                channels % 16 == 0 and lay.bottleneck_dim % 16 == 0 and lay.out_dim % 16 == 0)

    def _tdnnf_layer(self, lay, c, x, xs=None, return_bottleneck=False, want_aux=False, row_frames=None, tie=None, skip_f32=False, x_valid=True):
        """x [B, feat, T] -> [B, out, T'] (or the bottleneck [B, bott, T']).  In split-f16 mode the layers
        hand their activations on as split planes as well (`xs`, csrc/conv1d_mfma.hip): linearB then
        stages its 1024 x 3 input with 16-byte copies and linearA reads the bottleneck from planes; the
        f32 tensor stays for the bypass connection.  Returns (y, planes of y or None).
        skip_f32: the f32 output is not stored (y is then an unwritten shape carrier: the successor takes the planes); x_valid = False:
        `x` is such a carrier and the bypass is rebuilt from `xs` — both only where _plain_on_planes holds (_run_stack decides)."""
        ctx, sub = lay.context_len, int(lay.subsampling_factor)
        B = x.shape[0]
        if lay.subsampling_factor == 1.5:
            # chain/nn.py:267-304: windows of the flattened input every 1.5 frames + the add_padd bypass
            assert ctx == 1 and not return_bottleneck and c.codebook is None
            win, byp = ops.tdnnf_unfold15(x)
            if row_frames is not None:
                # ragged rows: the reference's bypass of this layer takes int(T / 1.5) frames and zero-fills the rest of
                # the windows (chain/nn.py:294-304) — T being the row's OWN frame count, not the padded batch's
                for i, n in enumerate(row_frames):
                    byp[i, :, int(n / 1.5):] = 0
            z = ops.conv1d(win, c.wB, lay.bottleneck_dim, 1, bias=c.bB, pad_left=0, pad_right=0, mode=c.modeB)
            kw = dict(res=byp, res_scale=lay.bypass_scale) if lay.use_bypass else {}
            ys = ops.split_like(B, lay.out_dim, z.shape[2], x.device) if (c.modeA == 1 and lay.out_dim % 16 == 0) else None
            y = ops.conv1d(z, c.wA, lay.out_dim, 1, bias=c.bA, ch_scale=c.scale, ch_shift=c.shift, relu=True, mode=c.modeA,
                           y_split=ys, y_split_slope=1.0, **kw)
            return y, ys
        planes_in = c.modeB == 1 and sub == 1 and xs is not None and x.shape[1] % 16 == 0
        need_z = c.codebook is not None or return_bottleneck
        if sub == 1 and not need_z and c.modeA == c.modeB and c.modeB in (0, 1):
            # the plain layer (no subsampling, no quantiser inside): ONE call across the C-ABI (sat_tdnnf_layer_f32) — linearB, linearA with
            # bypass + BatchNorm + ReLU; on split planes the bottleneck exists only as planes
            t_q = x.shape[2] - (ctx - 1)
            planes = c.modeB == 1 and lay.bottleneck_dim % 16 == 0
            zs = ops.split_like(B, lay.bottleneck_dim, t_q, x.device) if planes else None
            ys = ops.split_like(B, lay.out_dim, t_q, x.device) if (c.modeA == 1 and lay.out_dim % 16 == 0) else None
            y = ops.tdnnf_layer(x, c.wB, c.bB, c.wA, c.bA, lay.bottleneck_dim, lay.out_dim, ctx, bn_scale=c.scale, bn_shift=c.shift,
                                bypass_scale=float(lay.bypass_scale) if lay.use_bypass else 0.0, mode=c.modeB,
                                x_split=xs if planes_in else None, y_split=ys, z_split=zs, no_y=skip_f32, bypass_from_planes=not x_valid)
            return y, ys
        assert x_valid or (c.modeB == 1 and xs is not None), "an unwritten f32 input reached a layer that reads it"

        zs = None
        if planes_in and not need_z and lay.bottleneck_dim % 16 == 0:
            t_q = x.shape[2] - (ctx - 1)
            zs = ops.split_like(B, lay.bottleneck_dim, t_q, x.device)
            z = ops.conv1d(x, c.wB, lay.bottleneck_dim, ctx, bias=c.bB, pad_left=0, pad_right=0, mode=1,
                           x_split=xs, y_split=zs, y_split_slope=1.0, no_y=True)     # z: shape carrier only
        elif c.modeB == 1 and sub > 1 and xs is not None and x.shape[1] % 64 == 0 and lay.bottleneck_dim % 16 == 0 and not need_z:
            # context-free subsampling layer on split planes: the 1x1 product on EVERY frame (the GEMM kernels read the planes the
            # previous layer wrote: 8 x fewer bytes than the strided f32 copy of x the stride-1 kernels needed, round 6), every
            # sub-th frame of the 128-row result kept and handed on as planes
            z_all = ops.conv1d(x, c.wB, lay.bottleneck_dim, 1, bias=c.bB, pad_left=0, pad_right=0, mode=1, x_split=xs)
            z = z_all[:, :, ::sub].contiguous()
            zs = ops.act_split(z, 1.0)
        elif c.modeB == 1 and sub > 1:
            z = ops.conv1d(x[:, :, ::sub].contiguous(), c.wB, lay.bottleneck_dim, 1, bias=c.bB, pad_left=0, pad_right=0, mode=1)
        elif c.modeB == 1 and sub == 1 and not need_z and lay.bottleneck_dim % 16 == 0 and c.modeA == 1:
            # f32 input (the first layer: features), bottleneck still handed on as planes: linearA runs on the GEMM kernels
            t_q = x.shape[2] - (ctx - 1)
            zs = ops.split_like(B, lay.bottleneck_dim, t_q, x.device)
            z = ops.conv1d(x, c.wB, lay.bottleneck_dim, ctx, bias=c.bB, pad_left=0, pad_right=0, mode=1, y_split=zs, y_split_slope=1.0, no_y=True)
        else:
            z = ops.conv1d(x, c.wB, lay.bottleneck_dim, ctx, bias=c.bB, stride=sub, pad_left=0, pad_right=0, mode=c.modeB,
                           x_split=xs if planes_in else None)
        aux = None
        if c.codebook is not None:
            zq, idx, dist = ops.vq(z, c.codebook, want_dist=want_aux, tie=None if tie is None else (tie.pair, tie.scale, tie.counts))
            if tie is not None:
                tie.idx = idx
            aux = (z, idx, dist)
            z = zq
        if return_bottleneck:
            return (z, aux) if want_aux else z
        kw = {}
        if lay.use_bypass:
            lidx = ctx // 2 if ctx > 1 else 0
            if ctx == 2:
                lidx = 1
            kw = dict(res=x, res_scale=lay.bypass_scale, res_toff=lidx, res_tstride=sub)
        ys = None
        if c.modeA == 1 and lay.out_dim % 16 == 0:
            ys = ops.split_like(B, lay.out_dim, z.shape[2], x.device)
        y = ops.conv1d(z, c.wA, lay.out_dim, 1, bias=c.bA, ch_scale=c.scale, ch_shift=c.shift, relu=True, mode=c.modeA,
                       x_split=zs, y_split=ys, y_split_slope=1.0, **kw)
        return y, ys

    def vq_indices(self, wav):
        """VQ indices [N, T'] of `wav` [N, n] (device tensor; left untouched) as the extractor DELIVERS them: the configured
        arithmetic with the near-tie guard, flagged utterances decided again on the exact kernels.  -> (idx, rows decided again)"""
        with self._lock():
            feats = self._features_of(wav.clone())
            (zq, (_, idx, _)), status = self._run_stack_guarded(feats, want_aux=True)
            rows = self.resolve_ties(status, zq.permute(0, 2, 1), feats, wav, idx=idx)
        return idx, rows

    def vq_flip_report(self, wav):
        """`vq_flip_stats` of this extractor as configured against its exact-f32 kernels on `wav` [N, n] (device tensor; left
        untouched): how many VQ indices the split-f16 arithmetic decides differently, and whether every one of them is a near-tie."""
        keys = [k for k in ("precision", "w2v2_precision") if hasattr(self, k)]
        cfg = {k: getattr(self, k) for k in keys}
        _, (z, idx, _) = self.extract_bn(wav.clone(), want_aux=True)
        try:
            for k in keys:
                setattr(self, k, "f32")
            _, (z32, idx32, d32) = self.extract_bn(wav.clone(), want_aux=True)
        finally:
            for k, v in cfg.items():
                setattr(self, k, v)
        return vq_flip_stats(z, idx, z32, idx32, d32)

    def _run_stack(self, x, want_aux=False, tie=None):
        """x [B, C, T] (already padded) through tdnn1, tdnnfs[:-2], and the bottleneck of tdnnfs[-2]"""
        self._prepare(x.device)
        layers = self._stack_layers()
        xs, valid = None, True
        for i, (lay, c) in enumerate(zip(layers[:-1], self._cache[:-1])):
            # a layer whose successor reads its input from planes alone writes no f32 output; its successor then rebuilds the bypass from planes
            nxt, cn = layers[i + 1], self._cache[i + 1]
            skip = (self.tdnnf_planes_only and self._plain_on_planes(lay, c, x.shape[1]) and (xs is not None or not lay.use_bypass) and
                    self._takes_planes(nxt, cn, lay.out_dim, need_z=i + 2 == len(layers)))
            x, xs = self._tdnnf_layer(lay, c, x, xs, skip_f32=skip, x_valid=valid)
            valid = not skip
        return self._tdnnf_layer(layers[-1], self._cache[-1], x, xs, return_bottleneck=True, want_aux=want_aux, tie=tie, x_valid=valid)

    # ---- exact VQ indices in the default arithmetic: near-ties counted on the device, their utterances decided again ----------
    def _precision_keys(self):
        return [k for k in ("precision", "w2v2_precision") if hasattr(self, k)]

    def _lock(self):
        """the precision attributes select the kernels at launch time: a thread that decides flagged utterances again on the exact
        kernels (the batch job checks deferred statuses in its writer thread) must not meet the launching thread half way"""
        lk = self.__dict__.get("_arith_lock")
        if lk is None:
            import threading
            lk = self.__dict__.setdefault("_arith_lock", threading.RLock())
        return lk

    class _exact:
        """context: the extractor on its exact-f32 kernels (both packings stay cached)"""

        def __init__(self, net):
            self.net = net

        def __enter__(self):
            self.net._lock().acquire()
            self.keep = {k: getattr(self.net, k) for k in self.net._precision_keys()}
            for k in self.keep:
                setattr(self.net, k, "f32")

        def __exit__(self, *a):
            for k, v in self.keep.items():
                setattr(self.net, k, v)
            self.net._lock().release()

    def _tie_guard(self, device):
        """-> (pair distances of the codebook [n, n], tie_scale) for sat_vq_argmin_gather_tie_f32, or None when the guard is off.
        tie_scale = 2 K sigma_rel / sqrt(D): sigma_rel = |z - z_exact| / |z_exact| of this extractor's split-f16 arithmetic against its
        exact-f32 twin on two synthetic 2 s utterances, measured once per (weights, precision) — the error of d[a] - d[a'] is
        2 dz . (e_a' - e_a), i.e. ~ N(0, (2 sigma_c |e_a - e_a'|)^2) with sigma_c = sigma_rel |z_t| / sqrt(D) per component."""
        if not self.vq_tie_sigmas or self.__dict__.get("_frozen") or self.__dict__.get("_tie_busy"):
            return None
        if all(getattr(self, k) == "f32" for k in self._precision_keys()):
            return None
        key = tuple(getattr(self, k) for k in self._precision_keys()) + self._param_key() + (str(device),)
        g = self.__dict__.get("_tie")
        if g is None or g[0] != key:
            from . import synthetic
            self.__dict__["_tie_busy"] = True
            try:
                with torch.no_grad():
                    wav = synthetic.harm_batch([9001, 9002], 32000).to(device)
                    _, (z, _, _) = self.extract_bn(wav.clone(), want_aux=True)
                    with self._exact(self):
                        _, (z32, _, _) = self.extract_bn(wav.clone(), want_aux=True)
                    sigma_rel = float((z.double() - z32.double()).norm() / z32.double().norm().clamp_min(1e-30))
                    cb = self._cache[-1].codebook
                    pair = torch.cdist(cb.double(), cb.double()).to(torch.float32).contiguous()
            finally:
                self.__dict__["_tie_busy"] = False
            g = self.__dict__["_tie"] = (key, pair, sigma_rel, z.shape[1])
        _, pair, sigma_rel, D = g
        return pair, 2.0 * self.vq_tie_sigmas * sigma_rel / math.sqrt(D)

    def _run_stack_guarded(self, feats, want_aux=False):
        """`_run_stack` with the near-tie count: -> (result, TieStatus or None)"""
        guard = self._tie_guard(feats.device)
        if guard is None:
            return self._run_stack(feats, want_aux=want_aux), None
        B = feats.shape[0]
        init = self.__dict__.setdefault("_tie_init", {})
        tmpl = init.get((B, str(feats.device)))
        if tmpl is None:            # counts 0 | first near-tie frame INT32_MAX | last -1
            tmpl = init[(B, str(feats.device))] = torch.tensor([[0] * B, [2 ** 31 - 1] * B, [-1] * B], dtype=torch.int32).to(feats.device)
        ctx = _TieCtx(guard[0], guard[1], tmpl.clone())
        out = self._run_stack(feats, want_aux=want_aux, tie=ctx)
        return out, TieStatus(ctx.counts, ctx.idx)

    def _stack_receptive_field(self):
        """(S, W): output frame t of the TDNNF stack (tdnn1 .. the VQ layer's bottleneck) reads the padded input frames S t .. S t + W - 1
        ('valid' windows of context_len frames every subsampling_factor frames, chain/nn.py:267-278)"""
        S, W = 1, 1
        for lay in reversed(self._stack_layers()):
            sub, ctx = int(lay.subsampling_factor), int(lay.context_len)
            W = (W - 1) * sub + ctx
            S *= sub
        return S, W

    def _exact_rows(self, rows, feats, wav, spans=None):
        """Utterances `rows` on the exact-f32 kernels, from the batch's padded features `feats` (f32 front end: the fbank tag; the
        wav2vec2 tag overrides this and recomputes its encoder for those rows).  -> (zq [n, D, L'], idx [n, L'], t0 [n]): row i holds
        the output frames t0[i] .. t0[i] + L' - 1 of utterance rows[i].
        `spans` = (first, last) near-tie frame per row: only a WINDOW of the utterance is computed — every layer is a 'valid' window over
        frames, so the frames first .. last depend on the input frames S first .. S last + W - 1 alone, and the exact kernels give a
        frame the same bits whatever the tile it falls in: the window's frames ARE the full run's (asserted by
        tests/test_hip_guards.py).  Rows share one window length (the longest, moved left where it would pass the end); a window of
        three quarters of the utterance or more is the full run.  A flagged utterance then costs launches of 2 - 4 blocks instead of 16."""
        n, Tf = len(rows), feats.shape[2]
        S, W = self._stack_receptive_field()
        L, starts = tie_windows(Tf, S, W, spans) if spans is not None else (Tf, [0] * n)
        with self._exact(self):
            if L == Tf:
                sub = feats[rows].contiguous()
            else:
                sub = torch.stack([feats[r, :, a:a + L] for r, a in zip(rows, starts)]).contiguous()
            zq, (_, idx_x, _) = self._run_stack(sub, want_aux=True)
        return zq, idx_x, [a // S for a in starts]

    #: tests: treat every flagged utterance as changed (the rows are rewritten and the caller generates them again)
    vq_tie_force_patch = False

    def resolve_ties(self, status, bn, feats, wav, idx=None):
        """Decide the flagged utterances of `status` again on the exact kernels; those whose indices CHANGE get their rows of `bn`
        ([B, T', D] view of the stack's output, as extract_bn returns it) rewritten, and their VQ indices written into `idx` [B, T'] when
        given.  -> the changed rows (empty list: nothing for the caller to redo).  Blocking form of `TieFix`."""
        return TieFix(self, status, bn, feats, wav, idx).finish()

    # ---- the ASR half: Net.forward up to the chain / xent outputs --------------------------
    def _prepare_full(self, device):
        self._prepare(device)
        if self._cache_full is None:
            split = self.precision == "f16x3"
            f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
            after = [m for m in self.tdnnfs_after if isinstance(m, TDNNFBatchNormParams)]
            full = {"after": [(m, self._layer_cache(m, device, split)) for m in after],
                    "prefinal": [(m, self._layer_cache(m, device, split)) for m in (self.prefinal_chain, self.prefinal_xent)],
                    "out": [((packing.pack_conv_weight_f16x3 if split else packing.pack_conv_weight)(f32(o.weight).unsqueeze(-1)),
                             f32(o.bias).reshape(-1), o.weight.shape[0]) for o in (self.chain_output, self.xent_output)]}
            self._cache_full = full
            _lib.cache_rebuild_end(device)
        return self._cache_full

    def _asr_outputs(self, x):
        """x [B, C, T] (features, padded) -> (chain_out, log_softmax(xent_out)) as [B, T', output_dim] views"""
        full = self._prepare_full(x.device)
        layers = self._stack_layers()
        xs = None
        for lay, c in zip(layers, self._cache):          # the VQ layer runs through (quantised bottleneck -> linearA, BN, ReLU)
            x, xs = self._tdnnf_layer(lay, c, x, xs)
        x = ops.pad_replicate(x, self.padding_after, self.padding_after, interleave_right=True)
        xs = None
        for lay, c in full["after"]:
            x, xs = self._tdnnf_layer(lay, c, x, xs)
        outs = []
        for (lay, c), (w, b, odim) in zip(full["prefinal"], full["out"]):
            h, hs = self._tdnnf_layer(lay, c, x, xs)
            outs.append(ops.conv1d(h, w, odim, 1, bias=b, mode=c.modeA, x_split=hs))
        ops.log_softmax_channels_(outs[1])
        return outs[0].permute(0, 2, 1), outs[1].permute(0, 2, 1)


    # ---- batched ASR forward of utterances of different lengths --------------------------------------------------
    @staticmethod
    def _layers_out_len(layers, T):
        """frames after a run of TDNNF layers ('valid' windows of context_len frames every subsampling_factor frames;
        1.5: windows of the flattened input every int(1.5 D) values, chain/nn.py:267-304)"""
        for lay in layers:
            if lay.subsampling_factor == 1.5:
                T = ((T - 1) * lay.feat_dim) // int(1.5 * lay.feat_dim) + 1
            else:
                T = (T - lay.context_len) // int(lay.subsampling_factor) + 1
        return T

    def forward_ragged(self, wavs):
        """`forward()` for a LIST of utterances of different lengths in two batched passes (the reference decodes one
        utterance per call, chain/decoder.py:24-39; a batch of equal-length rows is all its `forward` takes):
        wavs = [n_i] or [1, n_i] waveforms in [-1, 1] on the HIP device -> [(chain_out [T'_i, output_dim],
        log_softmax(xent_out) [T'_i, output_dim])], each what `forward(wav_i[None])` returns.
        Every layer is a 'valid' window over frames, so frame t of utterance i only ever reads utterance i's own
        (individually replicate-padded) frames; rows are zero-filled behind their own length and cut at T'_i.  The two
        `pad_input` steps (before the stack, before `tdnnfs_after`) are applied per utterance.  Inputs are not modified."""
        if not hasattr(self, "_fbank_tables"):
            # wav2vec2 front end: its positional conv and attention look at every frame of a row — one utterance per call
            out = []
            for w in wavs:
                c, x = self.forward(w.reshape(1, -1).clone())
                out.append((c[0], x[0]))
            return out
        dev = wavs[0].device
        feats = [self.features(w.reshape(1, -1).to(torch.float32).contiguous(), scale=32768.0) for w in wavs]
        B = len(feats)
        L0 = [f.shape[2] for f in feats]
        x = torch.zeros(B, feats[0].shape[1], max(L0), dtype=torch.float32, device=dev)
        for i, f in enumerate(feats):
            x[i, :, :L0[i]] = f[0]
        full = self._prepare_full(dev)
        layers = self._stack_layers()
        xs = None
        for lay, c in zip(layers, self._cache):
            x, xs = self._tdnnf_layer(lay, c, x, xs)
        T1 = [self._layers_out_len(layers, n) for n in L0]
        pa = self.padding_after
        x2 = torch.zeros(B, x.shape[1], max(T1) + 2 * pa, dtype=torch.float32, device=dev)
        for i in range(B):
            x2[i, :, :T1[i] + 2 * pa] = ops.pad_replicate(x[i:i + 1, :, :T1[i]].contiguous(), pa, pa, interleave_right=True)[0]
        x, xs = x2, None
        cur = [n + 2 * pa for n in T1]
        for lay, c in full["after"]:
            x, xs = self._tdnnf_layer(lay, c, x, xs, row_frames=cur)
            cur = [self._layers_out_len([lay], n) for n in cur]
        T2 = cur
        outs = []
        for (lay, c), (w, b, odim) in zip(full["prefinal"], full["out"]):
            h, hs = self._tdnnf_layer(lay, c, x, xs)
            outs.append(ops.conv1d(h, w, odim, 1, bias=b, mode=c.modeA, x_split=hs))
        ops.log_softmax_channels_(outs[1])
        return [(outs[0][i, :, :T2[i]].t(), outs[1][i, :, :T2[i]].t()) for i in range(B)]


class _AsrHead(nn.Module):
    """the ASR output half of the reference nets (tdnnfs_after, prefinal_*, *_output): out of
    the anonymization path but part of the checkpoint, so the tensors are held."""

    @staticmethod
    def attach(net, hidden, bottleneck, prefinal, ks, subs, output_dim):
        seq = []
        from .params import _Identity
        for k, s in zip(ks, subs):
            seq += [TDNNFBatchNormParams(hidden, hidden, bottleneck, k, s), _Identity()]
        net.tdnnfs_after = nn.Sequential(*seq)
        net.prefinal_chain = TDNNFBatchNormParams(hidden, hidden, prefinal, 1, 1)
        net.prefinal_xent = TDNNFBatchNormParams(hidden, hidden, prefinal, 1, 1)
        net.chain_output = _InnerNat(hidden, output_dim)
        net.xent_output = _InnerNat(hidden, output_dim)


class TdnnfVqNet(_TdnnfBase):
    """fbank front end + 12 TDNNF layers + VQ bottleneck (tdnnf_vq.py:20-168)."""

    def __init__(self, output_dim, hidden_dim=1024, bottleneck_dim=128, prefinal_bottleneck_dim=256,
                 kernel_size_list=([3, 3, 3, 1, 3, 3, 3, 3, 3, 3, 3, 3], [1, 3, 3, 3]),
                 subsampling_factor_list=([1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1], [1.5, 1, 1, 1]),
                 p_dropout=0.1, codebook_size=48):
        super().__init__()
        self.input_dim = 80
        self.output_dim = output_dim
        self.padding = get_padding(kernel_size_list[0], subsampling_factor_list[0]) // 2
        self.padding_after = get_padding(kernel_size_list[1], subsampling_factor_list[1]) // 2
        self.tdnn1, self.tdnnfs = tdnnf_stack(self.input_dim, hidden_dim, bottleneck_dim, prefinal_bottleneck_dim,
                                              kernel_size_list[0], subsampling_factor_list[0], codebook_size)
        _AsrHead.attach(self, hidden_dim, bottleneck_dim, prefinal_bottleneck_dim, kernel_size_list[1],
                        subsampling_factor_list[1], output_dim)
        self._tables = None
        self._init_cache()

    def _fbank_tables(self, device):
        if self._tables is None or self._tables[0].device != device:
            mel = mel_banks(self.input_dim)
            nz = mel > 0
            lo = torch.where(nz.any(1), nz.float().argmax(1), torch.zeros(mel.shape[0], dtype=torch.long))
            hi = mel.shape[1] - torch.flip(nz, [1]).float().argmax(1)
            hi = torch.where(nz.any(1), hi, torch.zeros_like(hi))
            self._tables = (povey_window().to(device), mel.to(device), lo.to(torch.int32).to(device),
                            hi.to(torch.int32).to(device))
        return self._tables

    def features(self, x, scale=1.0):
        """fbank + UttCMVN + pad_input -> [N, 80, T + 2*padding]; `scale` multiplies the samples inside the framing
        kernel (1.0: x already scaled by 32768)"""
        win, mel, lo, hi = self._fbank_tables(x.device)
        return ops.fbank_cmvn_pad(x, win, mel, lo, hi, scale=scale, pad=self.padding, cmvn=True)

    def _features_of(self, x):
        return self.features(x.to(torch.float32).contiguous(), scale=32768.0)

    def _extract_bn_private(self, x, defer_ties=False):
        """extract_bn for a caller that owns no view of `x` afterwards (Net.get_bn, whose reference version clones
        its input only because extract_bn scales in place): the 32768 scaling happens inside the framing kernel,
        the input is left untouched and neither the clone nor the scaling pass is launched.  Same values: the
        kernel multiplies each sample by the scale before anything else."""
        if not x.is_cuda or x.dim() != 2:
            raise _lib.SatError("extract_bn expects a 2-dimensional tensor [N, samples] on the HIP device")
        with self._lock():
            return self._bn_guarded(self.features(x.to(torch.float32).contiguous(), scale=32768.0), x, defer_ties)

    def _bn_guarded(self, feats, wav, defer_ties=False):
        """stack + VQ with the near-tie guard: -> bn [N, T', D], or (bn, fix) with `fix()` -> rows decided again (the caller runs it
        once the rest of its launches are enqueued, and repeats what it derived from those rows of bn)"""
        out, status = self._run_stack_guarded(feats)
        bn = out.permute(0, 2, 1)
        st = self.__dict__.setdefault("tie_stats", {"utterances": 0, "rerun": 0, "changed": 0})
        st["utterances"] += bn.shape[0] if status is not None else 0
        if defer_ties:
            return bn, (TieFix(self, status, bn, feats, wav) if status is not None else None)
        self.resolve_ties(status, bn, feats, wav)
        return bn

    def extract_bn(self, x: torch.Tensor, want_aux=False) -> torch.Tensor:
        """inputs [N, n] -> [N, T, 256]   (tdnnf_vq.py:236-257; like the reference this scales
        its argument in place by 32768 — callers go through get_bn, which clones)"""
        if not x.is_cuda:
            raise _lib.SatError("extract_bn runs on the HIP device only (no CPU fallback); move the input to 'cuda'")
        if x.dim() != 2:
            raise _lib.SatError("extract_bn expects a 2-dimensional tensor [N, samples]")
        x *= 32768
        feats = self.features(x.to(torch.float32))
        if want_aux:                       # diagnostics: the arithmetic as configured, no second decision
            out = self._run_stack(feats, want_aux=True)
            return out[0].permute(0, 2, 1), out[1]
        with self._lock():
            return self._bn_guarded(feats, x)

    def forward(self, x):
        """waveforms [N, n] in [-1, 1] -> (chain_out, log_softmax(xent_out)), each [N, T', output_dim]
        (tdnnf_vq.py:259-284, eval mode).  Like the reference this scales its argument in
        place by 32768."""
        if not x.is_cuda:
            raise _lib.SatError("forward runs on the HIP device only (no CPU fallback); move the input to 'cuda'")
        if x.dim() != 2:
            raise _lib.SatError("forward expects a 2-dimensional tensor [N, samples]")
        x *= 32768
        return self._asr_outputs(self.features(x.to(torch.float32)))
