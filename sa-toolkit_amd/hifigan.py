"""HiFi-GAN generator behind the reference's `CoreHifiGan` interface
(reference: satools/satools/hifigan/archi.py:21-115).  Parameters live in the reference's
state-dict layout (params.CoreHifiGanParams); the forward runs entirely in the HIP library
(sat_hifigan_forward_f32).  Weight-norm is folded once (exact), weights are
re-laid out for the MFMA conv kernel, and both are cached until parameters change."""
import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _lib, packing
from ._lib import check, lib, ptr, stream
from .params import CoreHifiGanParams


class CoreHifiGan(CoreHifiGanParams):
    #: matrix-product arithmetic of the generator convs: "f16x3" (split-f16 on the f16 matrix cores,
    #: ~2^-21 relative per product), "f16f8r" (round 5, the default: "f16x3" with the ResBlock convs of the THICK stages — C >= 128, the
    #: LDS-DMA ring kernel — computing hi*hi in f16 and both cross terms on the block-scaled e4m3 MFMA: 2 MFMA units per
    #: product instead of 3, ~2^-15 per product on those layers; batches too small for the ring kernel run "f16x3"),
    #: "f16f8" (round 1: every conv that way on the register-staged tile, slower than "f16x3") or "f32" (exact f32 MFMA).
    #: The output stage is always f32.
    precision = os.environ.get("SATOOLS_AMD_GEN_PRECISION", "f16f8r")
    #: "f16f8r": bit i = stage i may run its ResBlock convs with e4m3 cross terms (default: every stage the ring kernel serves)
    f8_stages = int(os.environ.get("SATOOLS_AMD_GEN_F8_STAGES", "255"))
    #: hand activations between layers as split planes (csrc/hifigan.hip); 0 = f32 tensors
    split_acts = int(os.environ.get("SATOOLS_AMD_GEN_SPLIT_ACTS", "1"))
    #: number of leading generator stages whose three resblock branches run on separate HIP streams: -2 % for a
    #: single convert() in flight, a loss once two are (bench --jobs 2), so off by default
    branch_streams = int(os.environ.get("SATOOLS_AMD_GEN_BRANCH_STREAMS", "0"))

    #: bit mask of the ResBlock steps of the C = 64 stage that run as ONE launch (csrc/pair64.hip): 1 = 3 taps, 2 = 7 taps,
    #: 4 = 11 taps (measured slower fused)
    fuse_pair64 = int(os.environ.get("SATOOLS_AMD_GEN_FUSE_PAIR64", "3"))

    #: run the whole MRF block of a stage (all branches and steps + the mean) as ONE launch where csrc/mrf.hip supports it
    #: (C = 16); same bits as the launch-by-launch path
    fuse_mrf = int(os.environ.get("SATOOLS_AMD_GEN_FUSE_MRF", "1"))

    #: thick stages (C > 64): the i-th conv of all three MRF branches as ONE sat_conv1d_multi_f32 call (one launch of the
    #: LDS-DMA ring kernel, csrc/conv_ring16.hip, where it serves them; else the single launches) — same bits either way
    multi_branch = int(os.environ.get("SATOOLS_AMD_GEN_MULTI_BRANCH", "1"))

    #: the two thin upsamplers (64 -> 32 and 32 -> 16 channels) on the streaming kernel of csrc/ups2.hip
    ups2 = int(os.environ.get("SATOOLS_AMD_GEN_UPS2", "1"))

    #: the stride-4 upsamplers (256 -> 128 and 128 -> 64 channels) on the LDS-DMA ring of csrc/conv_ring16.hip: their polyphase rows
    #: are PACKED grouped by phase (sat_conv1d_desc.up_grouped) and the all-zero tap slots of a phase are skipped (2 of 3 slots carry
    #: weights at k = 8).  Needs the split-f16 generator on the split-plane pipeline (precision "f16x3", split_acts)
    ups_ring = int(os.environ.get("SATOOLS_AMD_GEN_UPS_RING", "1"))

    #: the f32 mean of an MRF block that only the next upsampler consumes (as planes) is not stored (sat_conv1d_desc.accum_no_store)
    skip_dead_sum = int(os.environ.get("SATOOLS_AMD_GEN_SKIP_DEAD_SUM", "1"))

    #: per-stream workspaces kept (3.3 GB each at 32 x 5 s: 26 GB at the default); beyond it the least recently used one is
    #: dropped.  One per convert() job in flight on the GPU (the reference's jobs_per_compute_device, bench.py --jobs) is
    #: what is needed; raise SATOOLS_AMD_GEN_MAX_WORKSPACES for more concurrent streams
    max_workspaces = int(os.environ.get("SATOOLS_AMD_GEN_MAX_WORKSPACES", "8"))

    #: "f16f8r" at EVERY batch size (the calibration batch of Net.check_precision is too small for the ring kernel's default dispatch)
    force_f8 = 0

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._handle = None
        self._packed = None
        self._packed_key = None
        self._ws = None

    # -- device-side weight cache ---------------------------------------------------------
    def _param_key(self):
        # the module tree is fixed after construction: walk it once, then only look at the tensors (this runs on
        # every forward; `.to()` and `load_state_dict` keep the Parameter objects and show up in data_ptr / _version)
        ps = self.__dict__.get("_flat_params")
        if ps is None:
            ps = self.__dict__["_flat_params"] = list(self.parameters())
        return (self.precision, self.split_acts, self.branch_streams, self.fuse_pair64, self.fuse_mrf, self.ups2, self.multi_branch, self.ups_ring, self.f8_stages, self.skip_dead_sum) + tuple((p.data_ptr(), p._version) for p in ps)

    def invalidate(self):
        self._packed_key = None

    def remove_weight_norm(self):
        # the reference's Net API (hifigan.py:51-52): weight_g / weight_v are replaced by a new `weight` Parameter, so
        # the walked parameter list of _param_key is stale and the packed weights must be rebuilt from the new tensors
        super().remove_weight_norm()
        self.__dict__.pop("_flat_params", None)
        self.invalidate()

    def _conv_modules(self):
        mods = [self.conv_pre] + list(self.ups)
        for rb in self.resblocks:
            for i in range(3):
                mods += [rb.convs1[i], rb.convs2[i]]
        return mods + [self.conv_post]

    def _prepare(self, device):
        if self.__dict__.get("_frozen"):           # packed weights installed by frozen.load_frozen: no parameters to fold
            return
        key = self._param_key()
        if self._packed_key == key and self._handle is not None:
            return
        _lib.cache_rebuild_begin(device, self._packed is not None)
        packed, modes, packed8 = [], [], {}
        mods = self._conv_modules()
        n_ups = len(self.ups)
        for i, m in enumerate(mods):
            w = m.folded_weight().to(device=device, dtype=torch.float32)
            b = m.bias.detach().to(device=device, dtype=torch.float32).contiguous()
            if self.precision not in ("f16x3", "f16f8", "f16f8r", "f32"):
                raise _lib.SatError(f"unknown generator precision {self.precision!r}")
            split = self.precision != "f32"
            mode = _lib.CONV_F16X3 if split else _lib.CONV_F32
            pack = packing.pack_conv_weight_f16x3 if split else packing.pack_conv_weight
            if self.precision == "f16f8" and i >= 1:      # conv_pre stages the f32 input itself: split-f16
                mode, pack = _lib.CONV_F16F8, packing.pack_conv_weight_f16f8
            if i == len(mods) - 1:
                wp = w.reshape(w.shape[1], w.shape[2]).contiguous()  # conv_post: [C][7], f32 streaming kernel
                mode = _lib.CONV_F32
            elif 1 <= i <= n_ups:
                u, k = self.upsample_rates[i - 1], self.upsample_kernel_sizes[i - 1]
                grouped = self._ups_grouped() and packing.upsample_grouped_supported(w.shape[0], w.shape[1], k, u, (k - u) // 2)
                wc, _, _ = packing.convtranspose_as_phase_conv(w, u, (k - u) // 2, grouped=grouped)
                wp = pack(wc, up=u)
            else:
                wp = pack(w)
                # ResBlock convs of the thick stages (C >= 128: what csrc/conv_ring16.hip serves): a second packing with e4m3 cross terms
                stage = (i - 1 - n_ups) // (6 * len(self.resblock_kernel_sizes))
                if (i > n_ups and self.precision == "f16f8r" and bool(self.split_acts) and (self.f8_stages >> stage) & 1 and w.shape[0] >= 128
                        and w.shape[1] % 32 == 0 and w.shape[2] >= 3):
                    packed8[i] = packing.pack_conv_weight_f16f8r(w)
            packed.append((wp, b))
            modes.append(mode)
        self._install_packed(packed, modes, packed8=packed8)
        self._packed_key = key
        _lib.cache_rebuild_end(device)

    def _ups_grouped(self):
        """whether the stride-4 upsamplers' rows are packed grouped by phase (what the C handle is told as option ups_ring)"""
        return bool(self.ups_ring) and self.precision in ("f16x3", "f16f8r") and bool(self.split_acts)

    def _install_packed(self, packed, modes, ups_grouped=None, packed8=None):
        """hand the kernel-ready weights [(packed weight, bias)] of every conv to the C handle (also the entry point of
        frozen.load_frozen, which brings them from a file instead of folding and packing parameters)"""
        l = lib()
        if self._handle is None:
            h = C.c_void_p()
            dil = [d for ds in self.resblock_dilation_sizes for d in ds]
            check(l.sat_hifigan_create(C.byref(h), self.imput_dim, self.upsample_initial_channel,
                                       len(self.upsample_rates), _lib.int_array(self.upsample_rates),
                                       _lib.int_array(self.upsample_kernel_sizes), len(self.resblock_kernel_sizes),
                                       _lib.int_array(self.resblock_kernel_sizes), _lib.int_array(dil)),
                  "sat_hifigan_create")
            self._handle = h
        n = l.sat_hifigan_num_convs(self._handle)
        if len(packed) != n or len(modes) != n:
            raise _lib.SatError(f"generator: {len(packed)} packed convolutions / {len(modes)} modes for an architecture of {n}")
        for i, ((wp, b), mode) in enumerate(zip(packed, modes)):
            check(l.sat_hifigan_set_conv(self._handle, i, ptr(wp), ptr(b), mode), "sat_hifigan_set_conv")
            if mode == _lib.CONV_F16X3:      # the packed weights' power-of-two layer scale (packing.pack_conv_weight_f16x3)
                if not hasattr(wp, "w_descale"):
                    raise _lib.SatError(f"generator conv {i}: split-f16 packed weights without .w_descale (lost by .to() / .clone(): "
                                        "use packing.move_packed, or set it to 1.0 for weights packed with scale=False)")
                check(l.sat_hifigan_set_conv_descale(self._handle, i, float(wp.w_descale)), "sat_hifigan_set_conv_descale")
        check(l.sat_hifigan_set_option(self._handle, b"split_acts", int(self.split_acts)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"branch_streams", int(self.branch_streams)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"fuse_pair64", int(self.fuse_pair64)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"fuse_mrf", int(self.fuse_mrf)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"multi_branch", int(self.multi_branch)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"ups2", int(self.ups2)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"force_f8", int(self.force_f8)), "sat_hifigan_set_option")
        check(l.sat_hifigan_set_option(self._handle, b"skip_dead_sum", int(self.skip_dead_sum)), "sat_hifigan_set_option")
        # (a frozen model brings the row order its weights were packed in)
        self._packed_ups_grouped = self._ups_grouped() if ups_grouped is None else bool(ups_grouped)
        check(l.sat_hifigan_set_option(self._handle, b"ups_ring", int(self._packed_ups_grouped)), "sat_hifigan_set_option")
        # second packings (SAT_CONV_F16F8R) of the thick stages' ResBlock convs: {conv id: packed tensor}
        packed8 = dict(packed8 or {})
        stages = 0
        n_ups, nk = len(self.upsample_rates), len(self.resblock_kernel_sizes)
        for i, w8 in packed8.items():
            check(l.sat_hifigan_set_conv_f8r(self._handle, int(i), ptr(w8)), "sat_hifigan_set_conv_f8r")
            stages |= 1 << ((int(i) - 1 - n_ups) // (6 * nk))
        check(l.sat_hifigan_set_option(self._handle, b"f8_stages", stages & int(self.f8_stages)), "sat_hifigan_set_option")
        self._packed8 = packed8
        self._packed = packed  # keeps the device buffers alive
        self._packed_modes = list(modes)

    def _workspace(self, B, T, device):
        # one workspace per launch stream: concurrent convert() calls on different streams (the
        # reference's `jobs_per_compute_device`, bin/anonymize:85-93) must not share scratch buffers
        need = lib().sat_hifigan_workspace_bytes(self._handle, B, T)
        key = torch.cuda.current_stream(device).cuda_stream
        if self._ws is None:
            self._ws = OrderedDict()
        ws = self._ws.get(key)
        if ws is not None:
            self._ws.move_to_end(key)
        if ws is None or ws.numel() * 4 < need or ws.device != device:
            self._ws.pop(key, None)
            while len(self._ws) >= self.max_workspaces:
                self._ws.popitem(last=False)   # least recently used: a stream that no longer exists (keys are raw handles)
            ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
            self._ws[key] = ws
        return ws, need

    # -- which arithmetic ran / what range it saw ------------------------------------------------
    def set_force_f8(self, value):
        """option force_f8 of the C handle (no re-packing: the second packings are installed whenever `precision` is "f16f8r")"""
        self.force_f8 = int(bool(value))
        if self._handle is not None:
            check(lib().sat_hifigan_set_option(self._handle, b"force_f8", self.force_f8), "sat_hifigan_set_option")

    @property
    def last_arithmetic(self):
        """arithmetic of the ResBlock convs in this generator's LAST forward: "f32", "f16x3", "f16f8", or "f16f8r(stages i,j)" when
        stages i, j ran with 8-bit cross terms (a batch too small for the ring kernel runs "f16x3" although `precision` says
        "f16f8r": the same utterance alone and inside a batch of 32 differ by the mode, ~1e-6 on the waveform)"""
        if self._handle is None:
            return None
        if self.precision in ("f32", "f16f8", "f16x3"):
            return self.precision
        v = C.c_int(0)
        check(lib().sat_hifigan_get_option(self._handle, b"last_f8_stages", C.byref(v)), "sat_hifigan_get_option")
        if not v.value:
            return "f16x3"
        return "f16f8r(stages " + ",".join(str(i + 1) for i in range(len(self.upsample_rates)) if (v.value >> i) & 1) + ")"     # (stage 1 = C 256)

    def range_probe(self, x):
        """one forward of x with the planes it writes probed (sat_hifigan_set_range_probe): per stage the number of hi halves past
        57 344 (the largest e5m2, where the 8-bit sidecar of "f16f8r" saturates; f16 ends at 65 504) or not finite, and the largest
        |hi| seen.  -> {"past_e5m2_max": [per stage], "max_abs": [per stage]}.  Diagnostic of Net.check_precision."""
        n = len(self.upsample_rates)
        buf = torch.zeros(2 * n, dtype=torch.int64, device=x.device)
        self._prepare(x.device)
        check(lib().sat_hifigan_set_range_probe(self._handle, ptr(buf)), "sat_hifigan_set_range_probe")
        try:
            self.forward_resnet(x)
        finally:
            check(lib().sat_hifigan_set_range_probe(self._handle, None), "sat_hifigan_set_range_probe")
        w = buf.cpu()
        mx = (w[1::2] & 0xFFFFFFFF).to(torch.int32).view(torch.float32)
        return {"past_e5m2_max": [int(v) for v in w[0::2]], "max_abs": [float(v) for v in mx]}

    def f8_weight_stats(self):
        """what the e4m3 cross-term operands of the installed "f16f8r" packings lose: {"values", "clipped" (|v| > 448: none by
        construction of the layer scale), "flushed" (non-zero values below e4m3's smallest subnormal), "subnormal" (kept with fewer
        than 4 significant bits: rows whose gain lies far below the layer's largest)} summed over the convs (packing.pack_conv_weight_f16f8r)"""
        tot = {"values": 0, "clipped": 0, "flushed": 0, "subnormal": 0}
        for w8 in (getattr(self, "_packed8", None) or {}).values():
            for k, v in getattr(w8, "f8_stats", {}).items():
                tot[k] += v
        return tot

    # -- reference interface ----------------------------------------------------------------
    def forward_resnet(self, x):
        if not x.is_cuda:
            raise _lib.SatError("CoreHifiGan runs on the HIP device only (no CPU fallback)")
        x = x.to(torch.float32).contiguous()
        B, c, T = x.shape
        if c != self.imput_dim:
            raise _lib.SatError(f"generator expects {self.imput_dim} input channels, got {c}")
        self._prepare(x.device)
        ws, need = self._workspace(B, T, x.device)
        up = 1
        for u in self.upsample_rates:
            up *= u
        y = torch.empty(B, 1, T * up + 1, dtype=torch.float32, device=x.device)
        check(lib().sat_hifigan_forward_f32(self._handle, ptr(x), ptr(y), ptr(ws), need, B, T, stream()),
              "sat_hifigan_forward_f32")
        return y

    def forward(self, x):
        # reference returns (signal, torch.empty((1))) when iSTFTNetout is False (archi.py:93-107)
        return self.forward_resnet(x), torch.empty((1))

    def __del__(self):
        try:
            if self._handle is not None:
                lib().sat_hifigan_destroy(self._handle)
        except Exception:
            pass
