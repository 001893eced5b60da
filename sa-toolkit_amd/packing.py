"""Host-side weight re-layout for the fused conv1d kernel (done once at load time).

packed[g][cin_pad][ksize][co_pad], co fastest: a half-wave's A-fragment read is one 128-B row.
"""
import ctypes as C

import torch

from . import _lib


def packed_dims(c_in, c_out, up=1, groups=1):
    cin_pad, co_pad = C.c_int(), C.c_int()
    _lib.check(_lib.lib().sat_conv1d_packed_dims(c_in, c_out, up, groups, C.byref(cin_pad), C.byref(co_pad)),
               "sat_conv1d_packed_dims")
    return cin_pad.value, co_pad.value


def phase_dims(k, u, pad):
    ks, pl = C.c_int(), C.c_int()
    _lib.check(_lib.lib().sat_convtranspose_phase_dims(k, u, pad, C.byref(ks), C.byref(pl)),
               "sat_convtranspose_phase_dims")
    return ks.value, pl.value


def pack_conv_weight(w: torch.Tensor, groups: int = 1, up: int = 1) -> torch.Tensor:
    """w [rows = C_out*up, C_in/groups, K] (torch Conv1d layout) -> packed, same device as w."""
    rows, cin_g, k = w.shape
    assert rows % (groups * up) == 0
    c_out = rows // up
    cin_pad, co_pad = packed_dims(cin_g * groups, c_out, up, groups)
    rows_g = rows // groups
    out = torch.zeros(groups, cin_pad, k, co_pad, dtype=torch.float32, device=w.device)
    wg = w.to(torch.float32).reshape(groups, rows_g, cin_g, k)
    out[:, :cin_g, :, :rows_g] = wg.permute(0, 2, 3, 1)
    return out.contiguous()


def upsample_grouped_supported(c_in, c_out, k, stride, padding):
    """the shapes whose polyphase rows may be ordered (16-channel group, phase, channel): sat_conv1d_desc.up_grouped"""
    return bool(_lib.lib().sat_upsample_grouped_supported(int(c_in), int(c_out), int(k), int(stride), int(padding)))


def convtranspose_zero_taps(k, stride, padding):
    """bit (slot * 4 + phase) set: that tap slot of the polyphase conv is all zero for that phase (sat_conv1d_desc.up_zero_taps)"""
    return int(_lib.lib().sat_convtranspose_zero_taps(int(k), int(stride), int(padding)))


def convtranspose_as_phase_conv(w: torch.Tensor, stride: int, padding: int, grouped: bool = False):
    """ConvTranspose1d weight [C_in, C_out, K] -> equivalent conv weight [C_out*u, C_in, K'] whose
    row co*u + r produces output phase r (t = q*u + r), plus (K', pad_left).
    Output t reads input s = q + delta through tap j = r + padding - u*delta.
    `grouped`: row (co // 16 * u + r) * 16 + co % 16 instead (sat_conv1d_desc.up_grouped; C_out % 16 == 0)."""
    c_in, c_out, k = w.shape
    u = stride
    kp, pad_left = phase_dims(k, u, padding)
    wc = torch.zeros(c_out * u, c_in, kp, dtype=w.dtype, device=w.device)
    for r in range(u):
        for jp in range(kp):
            delta = jp - pad_left
            j = r + padding - u * delta
            if 0 <= j < k:
                wc[r::u, :, jp] = w[:, :, j].t()
    if grouped:
        assert c_out % 16 == 0
        wc = wc.reshape(c_out // 16, 16, u, c_in, kp).permute(0, 2, 1, 3, 4).reshape(c_out * u, c_in, kp).contiguous()
    return wc, kp, pad_left


def grouped_zero_taps(wc: torch.Tensor, u: int) -> int:
    """the all-zero (tap slot, phase) pairs of a polyphase weight whose rows are grouped by phase (convtranspose_as_phase_conv(...,
    grouped=True)), read from the WEIGHTS: bit (slot * 4 + phase).  pack_conv_weight_f16x3 attaches it to the packed tensor of such a
    weight (`.up_zero_taps`) when told so, and ops.conv1d refuses a descriptor mask that claims more zeros than the weights have
    (sat_conv1d_desc.up_zero_taps is a promise about the weights: a wrong mask drops products silently)."""
    rows, c_in, kp = wc.shape
    v = wc.reshape(rows // (16 * u), u, 16, c_in, kp)
    mask = 0
    for slot in range(kp):
        for r in range(u):
            if not bool(v[:, r, :, :, slot].any()):
                mask |= 1 << (slot * 4 + r)
    return mask


class SplitRangeError(ValueError):
    """weights that the split-f16 representation cannot carry (non-finite values)"""


#: the largest |w| of a layer is moved into [2^(F16X3_TARGET_EXP - 1), 2^F16X3_TARGET_EXP) before the split: hi = f16(w') then
#: stays far from the f16 limit (65504 = 2^16), and lo = f16(w' - hi) ~ 2^-11 w' is a NORMAL f16 (>= 2^-14) for every
#: weight down to 2^-13 of the largest — without the scale a weight below 0.125 has a subnormal lo and the split only
#: carries it to an ABSOLUTE 2^-25 (a layer of 1e-4-sized weights: 3e-4 relative)
F16X3_TARGET_EXP = 10


def f16x3_scale_exponent(w: torch.Tensor) -> int:
    """power-of-two exponent e of the layer scale: the packed weights are w * 2^e, the kernels multiply the accumulator
    by 2^-e (exact).  0 for an all-zero tensor; raises SplitRangeError for non-finite weights."""
    if not bool(torch.isfinite(w).all()):
        raise SplitRangeError("pack_conv_weight_f16x3: non-finite weights")
    m = float(w.detach().abs().max())
    if m == 0.0:
        return 0
    import math
    e = F16X3_TARGET_EXP - (math.frexp(m)[1])        # m = f * 2^k, f in [0.5, 1): m * 2^e in [2^(TARGET-1), 2^TARGET)
    return max(-100, min(100, e))                     # (2^-e and w * 2^e must stay normal f32)


def pack_conv_weight_f16x3(w: torch.Tensor, groups: int = 1, up: int = 1, scale: bool = True) -> torch.Tensor:
    """split-f16 packing for SAT_CONV_F16X3: w [rows, C_in/groups, K] f32 ->
    [g][cin_pad/16][K][2 (hi|lo)][2 (channel half)][co_pad][8] f16, hi = f16(w'), lo = f16(w' - hi), w' = w * 2^e
    (w' - hi is exact in f32, so hi + lo carries 22 significand bits of w).  One (chunk, tap, part, half)
    segment holds 8 channels of every row: the kernels copy 32*MT-row pieces of it straight into LDS.
    The per-layer power-of-two scale (f16x3_scale_exponent; `scale=False`: e = 0) travels with the tensor as the
    attribute `.w_descale` = 2^-e, which ops.conv1d / ops.resblock_pair / the generator hand to the kernels
    (sat_conv1d_desc.w_descale): results are those of the unscaled weights to the last bit wherever the unscaled split
    was exact to 22 bits, and stay 22-bit accurate for layers of any magnitude."""
    e = f16x3_scale_exponent(w) if scale else 0
    p = pack_conv_weight(w, groups=groups, up=up)                    # [g][cin_pad][K][co_pad] f32
    if e:
        p = p * float(2.0 ** e)
    g, cin_pad, k, co_pad = p.shape
    p = p.reshape(g, cin_pad // 16, 2, 8, k, co_pad).permute(0, 1, 4, 2, 5, 3).contiguous()   # [g][nch][K][half][co][8]
    hi = p.to(torch.float16)
    lo = (p - hi.to(torch.float32)).to(torch.float16)
    out = torch.stack([hi, lo], dim=3).contiguous()                  # [g][nch][K][part][half][co][8]
    out.w_descale = float(2.0 ** -e)
    if up == 4 and groups == 1 and w.shape[0] % 64 == 0 and w.shape[2] <= 8:
        # the zero (tap slot, phase) pairs this weight WOULD have with its rows grouped by phase (the caller knows whether they are:
        # ops.conv1d(up_grouped=True, up_zero_taps=...) compares its mask with this one)
        out.up_zero_taps = grouped_zero_taps(w, up)
    return out


def move_packed(w: torch.Tensor, *args, **kwargs) -> torch.Tensor:
    """`w.to(*args, **kwargs)` for a tensor packed by pack_conv_weight_f16x3, keeping its `.w_descale` (a plain .to() / .clone() /
    .contiguous() returns a tensor without the attribute, which ops.* refuse in split-f16 mode)"""
    out = w.to(*args, **kwargs)
    for attr in ("w_descale", "up_zero_taps"):
        if hasattr(w, attr):
            setattr(out, attr, getattr(w, attr))
    return out


F8_W_HI_EXP, F8_W_LO_EXP, F8_X_LO_EXP = 6, 16, 10      # power-of-two scales of the e4m3 operands (csrc/conv1d_mfma.hip)


def pack_conv_weight_f16f8(w: torch.Tensor, groups: int = 1, up: int = 1) -> torch.Tensor:
    """packing for SAT_CONV_F16F8: w [rows, C_in/groups, K] f32 -> uint8
    [g][cin_pad/16][K][4][co_pad][16 B] with the segments  hi f16 ch 0-7 | hi f16 ch 8-15 |
    e4m3(lo * 2^16), 16 channels | e4m3(hi * 2^6), 16 channels   (hi = f16(w), lo = w - hi).
    The fixed scales need |w| < 7 (448 / 2^6); larger weights are refused so the caller can fall back
    to SAT_CONV_F16X3."""
    if float(w.abs().max()) >= 7.0:
        raise ValueError("pack_conv_weight_f16f8: |w| >= 7 does not fit the fixed e4m3 scale")
    p = pack_conv_weight(w, groups=groups, up=up).cpu()             # [g][cin_pad][K][co_pad] f32
    g, cin_pad, k, co_pad = p.shape
    nch = cin_pad // 16
    hi = p.to(torch.float16)
    lo = p - hi.to(torch.float32)
    hi16 = hi.reshape(g, nch, 2, 8, k, co_pad).permute(0, 1, 4, 2, 5, 3).contiguous()       # [g][nch][K][half][co][8] f16
    seg01 = hi16.view(torch.uint8).reshape(g, nch, k, 2, co_pad, 16)

    def e4m3(t, exp):                                                # [g][cin_pad][K][co] -> [g][nch][K][co][16] bytes
        q = (t * float(2 ** exp)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        return q.reshape(g, nch, 16, k, co_pad).permute(0, 1, 3, 4, 2).contiguous()

    seg2 = e4m3(lo, F8_W_LO_EXP).unsqueeze(3)
    seg3 = e4m3(hi.to(torch.float32), F8_W_HI_EXP).unsqueeze(3)
    return torch.cat([seg01, seg2, seg3], dim=3).contiguous().to(w.device)


F8R_W_HI_EXP, F8R_W_LO_EXP = -2, 9      # power-of-two scales of the e4m3 weight operands of SAT_CONV_F16F8R (csrc/conv_common.h F8R_E_*)


def pack_conv_weight_f16f8r(w: torch.Tensor) -> torch.Tensor:
    """packing for SAT_CONV_F16F8R (the LDS-DMA ring kernel with 8-bit cross terms, csrc/conv_ring16.hip):
    w [C_out, C_in, K] f32 (C_in % 32 == 0) -> uint8 [2 * ceil(C_in/32 * K / 2) steps][8 planes][co_pad][16 B].
    hi = f16(w'), lo = f16(w' - hi) of w' = w * 2^e, the SAT_CONV_F16X3 layer scale (`.w_descale` = 2^-e travels with the
    tensor).  The K dimension is the LINEAR sequence of (32-channel pair pp, tap t), L = pp * K + t; consecutive elements (2 q, 2 q + 1)
    form a pair — the last tap of a channel pair goes with the first tap of the next one, so an odd K costs no padding (only an odd
    C_in/32 * K ends with one zero element).  Per pair two steps:
      E step, plane (element j, chunk c, half hf) = 4 j + 2 c + hf: hi f16 of channels 32 pp_j + 16 c + 8 hf .. + 7 at tap t_j
      O step, plane (element j, term, chunk c)    = 4 j + 2 term + c: 16 channels of chunk c as e4m3 — term 0: lo * 2^9,
              term 1: hi * 2^-2 (the products W_lo . x_hi and W_hi . x_lo of the cross terms)"""
    rows, cin, k = w.shape
    if cin % 32 != 0:
        raise ValueError("pack_conv_weight_f16f8r: C_in must be a multiple of 32")
    e = f16x3_scale_exponent(w)
    p = pack_conv_weight(w)[0]                                       # [cin_pad][K][co_pad] f32
    if e:
        p = p * float(2.0 ** e)
    cin_pad, _, co_pad = p.shape
    npair = cin_pad // 32
    nlin = npair * k
    nq = (nlin + 1) // 2
    hi = p.to(torch.float16)
    lo = (p - hi.to(torch.float32)).to(torch.float16)

    def linear(t):                                                   # [cin_pad][K][co] -> [2 nq][32][co], element L = pp * K + t (zero padded)
        v = t.reshape(npair, 32, k, co_pad).permute(0, 2, 1, 3).reshape(nlin, 32, co_pad)
        if 2 * nq != nlin:
            v = torch.cat([v, torch.zeros(1, 32, co_pad, dtype=v.dtype, device=v.device)], dim=0)
        return v

    # E images: [q][j][c][hf][8][co] -> [q][j][c][hf][co][8]
    e_img = linear(hi).reshape(nq, 2, 2, 2, 8, co_pad).permute(0, 1, 2, 3, 5, 4).contiguous()
    e_img = e_img.view(torch.uint8).reshape(nq, 8, co_pad, 16)

    stats = {"values": 0, "clipped": 0, "flushed": 0, "subnormal": 0}

    def e4m3(t, exp):                                                # -> [q][j][c][co][16] bytes
        v = linear(t).to(torch.float32) * float(2.0 ** exp)
        q8 = v.clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
        # what the 8-bit operand loses (Net.check_precision reports it): clipped at 448, flushed to zero, kept as a subnormal (< 2^-6)
        a = v.abs()
        stats["values"] += int((a > 0).sum())
        stats["clipped"] += int((a > 448.0).sum())
        stats["flushed"] += int(((a > 0) & (q8.to(torch.float32) == 0)).sum())
        stats["subnormal"] += int(((a > 0) & (a < 2.0 ** -6)).sum())
        q = q8.view(torch.uint8)
        return q.reshape(nq, 2, 2, 16, co_pad).permute(0, 1, 2, 4, 3).contiguous()

    o_img = torch.stack([e4m3(lo, F8R_W_LO_EXP), e4m3(hi, F8R_W_HI_EXP)], dim=2)       # [q][j][term][c][co][16]
    o_img = o_img.reshape(nq, 8, co_pad, 16)
    out = torch.stack([e_img, o_img], dim=1).reshape(2 * nq, 8, co_pad, 16).contiguous()
    out.w_descale = float(2.0 ** -e)
    out.f8_stats = stats
    return out
