"""Thin host wrappers: torch tensors (device memory, current stream) -> C ABI calls of libcmk_hip.so.

PyTorch is plumbing here (allocation, streams); the arithmetic happens in the HIP kernels.  Activations are
NHWC float32; `View` is (NHWC tensor, channel offset, channels) so producers write straight into slices of
an OSA concat buffer.  Every function raises if the tensors are not on a GPU — there is no CPU fallback.
"""
import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, FcosLevel, check


def _stream() -> ctypes.c_void_p:
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise _lib.CmkError("{}: tensor is on {}; the CenterMask2 HIP path needs a GPU (no CPU fallback)".format(what, t.device))
    if t.dtype != torch.float32:
        raise _lib.CmkError("{}: expected float32, got {}".format(what, t.dtype))
    _need_current_device(t, what)


def _need_current_device(t: torch.Tensor, what: str) -> None:
    """The library launches on the stream it is handed — torch's CURRENT stream, i.e. the current device's.  A tensor of another
    device would be read through a stream (and split-K / GroupNorm workspaces) of the wrong GPU: refuse instead of faulting."""
    cur = torch.cuda.current_device()
    if t.device.index is not None and t.device.index != cur:
        raise _lib.CmkError("{}: tensor lives on cuda:{} but the current device is cuda:{}; wrap the call in "
                            "`with torch.cuda.device({})` (one process per GPU is the supported layout)".format(what, t.device.index, cur, t.device.index))


class View:
    """Channel slice [co, co+c) of a contiguous NHWC tensor (N,H,W,CS)."""
    __slots__ = ("t", "co", "c")

    def __init__(self, t: torch.Tensor, co: int = 0, c: Optional[int] = None):
        assert t.dim() == 4 and t.is_contiguous(), "View needs a contiguous (N,H,W,C) tensor"
        self.t = t
        self.co = co
        self.c = t.shape[3] - co if c is None else c
        assert 0 <= co and co + self.c <= t.shape[3]

    @property
    def cs(self) -> int:
        return self.t.shape[3]

    @property
    def nhw(self) -> Tuple[int, int, int]:
        return self.t.shape[0], self.t.shape[1], self.t.shape[2]

    def nchw(self) -> torch.Tensor:
        """Logical (N,C,H,W) view with channels_last strides (what the plugin API hands to callers)."""
        return self.t[..., self.co:self.co + self.c].permute(0, 3, 1, 2)


def as_view(x) -> "View":
    """Accept a View, an NHWC-contiguous 4-D tensor wrapped earlier, or a logical NCHW tensor (made channels_last)."""
    if isinstance(x, View):
        return x
    assert x.dim() == 4
    nhwc = x.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    return View(nhwc)


# ---------------------------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,k,k) -> [tap][Cin/16][cout_pad][16] (zero padded), the layout conv_igemm.hip streams."""
    lib = _lib.load()
    cout, cin, k, _ = w.shape
    cin_pad = (cin + 15) // 16 * 16
    cout_pad = lib.cmk_conv_cout_pad(cout)
    wp = torch.zeros((k * k, cin_pad, cout_pad), dtype=torch.float32)
    wp[:, :cin, :cout] = w.detach().float().cpu().permute(2, 3, 1, 0).reshape(k * k, cin, cout)
    wp = wp.reshape(k * k, cin_pad // 16, 16, cout_pad).permute(0, 1, 3, 2).contiguous()
    assert wp.numel() == lib.cmk_conv_packed_floats(cout, cin, k)
    return wp


_WINO_G = torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64)


def _wino_u(w: torch.Tensor) -> torch.Tensor:
    """(Cout,Cin,3,3) -> U = G g G^T as [chunk][ntile][16 freq in step order][64 co][16 ci] (float64 transform, rounded once)."""
    cout, cin = w.shape[0], w.shape[1]
    u = torch.einsum("ik,ockl,jl->ocij", _WINO_G, w.detach().double().cpu(), _WINO_G).reshape(cout, cin, 16)
    cin_pad, nt = (cin + 15) // 16 * 16, (cout + 63) // 64
    up = torch.zeros((cin_pad, nt * 64, 16), dtype=torch.float64)
    up[:cin, :cout] = u.permute(1, 0, 2)
    order = [f for g in range(4) for f in (2 * g, 2 * g + 1, 8 + 2 * g, 9 + 2 * g)]      # the kernels' streaming order
    up = up[:, :, order]
    return up.reshape(cin_pad // 16, 16, nt, 64, 16).permute(0, 2, 4, 3, 1).contiguous().float()


def pack_wino_weight(w: torch.Tensor) -> torch.Tensor:
    """Packed U for the Winograd kernel (cmk_conv_desc.w_wino):
    [chunk][ntile][step 4][fh 2][ng 2][fl 2][piece 2][lane = 32*hh + li][4] with freq = step*4 + fh*2 + fl (step order),
    co = ng*32 + li, ci = 8*hh + 4*piece + j — every operand load of a wave is one contiguous KiB."""
    lib = _lib.load()
    up = _wino_u(w)
    nc, nt = up.shape[0], up.shape[1]
    r = up.reshape(nc, nt, 4, 2, 2, 2, 32, 2, 2, 4)               # [chunk][nt][g][fh][fl][ng][li][hh][piece][j]
    r = r.permute(0, 1, 2, 3, 5, 4, 8, 7, 6, 9).contiguous().reshape(nc, nt, 16, 64, 16)
    assert r.numel() == lib.cmk_wino_packed_floats(w.shape[0], w.shape[1])
    return r


_WINO6_G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                        dtype=torch.float64)


def pack_wino6_weight(w: torch.Tensor) -> torch.Tensor:
    """Packed U = G g G^T of the Winograd F(4x4,3x3) kernel (cmk_conv_desc.w_wino6; float64 transform, rounded once):
    [Cin/8][ceil(Cout/32)][wave 4][slot 9][lane = 32*hh + li][4] — slot k < 6: frequency (wave, k); k >= 6: (4 + wave//2, 3*(wave%2) + k - 6);
    co = tile*32 + li, ci = chunk*8 + 4*hh + j — every operand load of a wave is one contiguous KiB."""
    lib = _lib.load()
    cout, cin = w.shape[0], w.shape[1]
    u = torch.einsum("ik,ockl,jl->ocij", _WINO6_G, w.detach().double().cpu(), _WINO6_G)              # (O, C, 6, 6)
    cin_pad, nt = (cin + 15) // 16 * 16, (cout + 31) // 32       # the same channel padding as the direct layout (PackedConv.cin_pad)
    up = torch.zeros((nt * 32, cin_pad, 6, 6), dtype=torch.float64)
    up[:cout, :cin] = u
    fa = torch.tensor([[wv if k < 6 else 4 + wv // 2 for k in range(9)] for wv in range(4)])
    fb = torch.tensor([[k if k < 6 else 3 * (wv % 2) + k - 6 for k in range(9)] for wv in range(4)])
    sel = up[:, :, fa, fb]                                                                               # (O, C, 4 waves, 9 slots)
    r = sel.reshape(nt, 32, cin_pad // 8, 2, 4, 4, 9)                                                    # [tile][li][chunk][hh][j][wave][slot]
    r = r.permute(2, 0, 5, 6, 3, 1, 4).contiguous().float()                                             # [chunk][tile][wave][slot][hh][li][j]
    assert r.numel() == lib.cmk_wino6_packed_floats(cout, cin_pad)
    return r.reshape(cin_pad // 8, nt, 4, 9, 64, 4)


def pack_split_weight(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin[,1,1]) fp32 -> the bf16-split packing of cmk_conv_desc.w_split (opt-in tune_wm 10): every weight as three bf16 values
    hi + mid + lo (round to nearest even, exact to 2^-24), laid out [Cin/16][cout_pad/32][piece][lane = 32*hh + li][8]:
    input channel 16*chunk + 8*hh + e of output channel 32*tile + li; cout_pad = Cout rounded up to 128, zero filled."""
    lib = _lib.load()
    if w.dim() == 4 and w.shape[2] == 3:          # 3x3 conv in the gather form: K walks the taps (kh, kw) outermost, then the padded input channels
        cout, cin = w.shape[0], w.shape[1]
        cin_pad1 = (cin + 15) // 16 * 16
        wk = torch.zeros((cout, 9, cin_pad1), dtype=torch.float32)
        wk[:, :, :cin] = w.detach().float().cpu().permute(0, 2, 3, 1).reshape(cout, 9, cin)
        w2 = wk.reshape(cout, 9 * cin_pad1)
        taps = 9
    else:
        w2 = w.detach().float().cpu().reshape(w.shape[0], w.shape[1])
        taps = 1
    cout, cin = w2.shape
    cin_pad, cout_pad = (cin + 15) // 16 * 16, (cout + 127) // 128 * 128
    full = torch.zeros((cout_pad, cin_pad), dtype=torch.float32)
    full[:cout, :cin] = w2
    pieces, rest = [], full
    for _ in range(3):
        p = rest.to(torch.bfloat16)
        pieces.append(p)
        rest = rest - p.float()
    st = torch.stack(pieces, 0)                                                    # (3, cout_pad, cin_pad)
    r = st.reshape(3, cout_pad // 32, 32, cin_pad // 16, 2, 8)                      # [piece][tile][li][chunk][hh][e]
    r = r.permute(3, 1, 0, 4, 2, 5).contiguous()                                   # [chunk][tile][piece][hh][li][e]
    assert r.numel() == taps * lib.cmk_split_packed_halves(cout, cin_pad // taps)
    return r.reshape(cin_pad // 16, cout_pad // 32, 3, 64, 8)


def pack_splith_weight(w: torch.Tensor):
    """(Cout, Cin, k, k) fp32, k = 1 | 3 -> (the fp16 two-piece packing of cmk_conv_desc.w_splith, 1 / S_w) for the opt-in fp16-split forms (tune_wm 11, 12):
    w' = w * S_w with S_w the power of two that puts max |w'| in [2^14, 2^15); pieces h = fp16(w'), m = fp16(w' - h) (the residual is exact in
    fp32); laid out [tap][Cin/16][cout_pad/32][piece][lane = 32*hh + li][8]: input channel 16*chunk + 8*hh + e of output channel 32*tile + li;
    cout_pad = Cout rounded up to 128, zero filled."""
    import math
    lib = _lib.load()
    cout, cin = w.shape[0], w.shape[1]
    cin_pad, cout_pad = (cin + 15) // 16 * 16, (cout + 127) // 128 * 128
    wf = w.detach().float().cpu()
    amax = float(wf.abs().max())
    s_w = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 and math.isfinite(amax) else 1.0
    taps = w.shape[2] * w.shape[3] if w.dim() == 4 else 1
    full = torch.zeros((cout_pad, taps, cin_pad), dtype=torch.float32)
    full[:cout, :, :cin] = wf.reshape(cout, cin, taps).permute(0, 2, 1) * s_w
    h = full.to(torch.float16)
    m = (full - h.float()).to(torch.float16)
    st = torch.stack([h, m], 0)                                                     # (2, cout_pad, taps, cin_pad)
    r = st.reshape(2, cout_pad // 32, 32, taps, cin_pad // 16, 2, 8)                # [piece][tile][li][tap][chunk][hh][e]
    r = r.permute(3, 4, 1, 0, 5, 2, 6).contiguous()                                 # [tap][chunk][tile][piece][hh][li][e]
    assert r.numel() == taps * lib.cmk_splith_packed_halves(cout, cin_pad)
    return r.reshape(taps, cin_pad // 16, cout_pad // 32, 2, 64, 8), 1.0 / s_w


class PackedConv:
    """Device-resident packed weights + per-channel epilogue (scale, shift) of one conv / linear layer."""

    def __init__(self, weight: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], device,
                 stride: int = 1):
        if weight.dim() == 2:
            weight = weight[:, :, None, None]
        self.cout, self.cin, self.k, _ = weight.shape
        self.cin_pad = (self.cin + 15) // 16 * 16
        self.stride = stride
        self.w = pack_conv_weight(weight).to(device)
        self.w_wino = pack_wino_weight(weight).to(device) if (self.k == 3 and stride == 1 and self.cin >= 16) else None
        # F(4x4,3x3) weights: 4x the 3x3 filter bank; packed for every conv that can use them (PACK_WINO6 = False skips it)
        self.w_wino6 = pack_wino6_weight(weight).to(device) if (PACK_WINO6 and self.k == 3 and stride == 1 and self.cin >= 32) else None
        # opt-in (ALLOW_SPLIT_BF16): the bf16-split packing for the pointwise GEMM's fp32-accurate split form (cmk.h tune_wm 10)
        self.w_split = pack_split_weight(weight).to(device) if (ALLOW_SPLIT_BF16 and self.cin_pad % 32 == 0 and (
            (self.k == 1 and self.cout > 224) or (self.k == 3 and (self.cout > 224 or 96 < self.cout <= 128)))) else None
        # opt-in (ALLOW_SPLIT_F16): the fp16 two-piece packing for the direct 3x3 form (cmk.h tune_wm 11)
        self.w_splith, self.w_splith_scale = None, 0.0
        if ALLOW_SPLIT_F16 and self.cin >= 32 and (self.k == 3 or (self.k == 1 and self.cout > 224 and self.cin_pad % 32 == 0)):
            wh_, self.w_splith_scale = pack_splith_weight(weight)
            self.w_splith = wh_.to(device)
        self.scale = (torch.ones(self.cout) if scale is None else scale.detach().float().cpu()).contiguous().to(device)
        self.shift = (torch.zeros(self.cout) if shift is None else shift.detach().float().cpu()).contiguous().to(device)


def fold_frozen_bn(weight, bias, running_mean, running_var, eps=1e-5):
    """FrozenBN as a per-channel epilogue: y = conv * scale + shift (d2 FrozenBatchNorm2d, eps 1e-5)."""
    scale = weight.double() / torch.sqrt(running_var.double() + eps)
    shift = bias.double() - running_mean.double() * scale
    return scale.float(), shift.float()


def _fill_desc(d: ConvDesc, x: View, pc: PackedConv, y: View, relu, relu_upto, res, res_upsample, in_relu, in_affine=None) -> None:
    _need_gpu(x.t, "conv2d")
    n, h, w = x.nhw
    d.x, d.x_cs, d.x_co = x.t.data_ptr(), x.cs, x.co
    d.w = pc.w.data_ptr()
    d.w_wino = pc.w_wino.data_ptr() if getattr(pc, "w_wino", None) is not None else None
    d.w_wino6 = pc.w_wino6.data_ptr() if getattr(pc, "w_wino6", None) is not None else None
    d.w_split = pc.w_split.data_ptr() if getattr(pc, "w_split", None) is not None else None
    d.w_splith = pc.w_splith.data_ptr() if getattr(pc, "w_splith", None) is not None else None
    d.w_splith_scale = float(getattr(pc, "w_splith_scale", 0.0))
    d.scale, d.shift = pc.scale.data_ptr(), pc.shift.data_ptr()
    if res is not None:
        d.res, d.res_cs, d.res_co = res.t.data_ptr(), res.cs, res.co
        d.res_mode = 2 if res_upsample else 1
        d.Hr, d.Wr = res.t.shape[1], res.t.shape[2]
    else:
        d.res, d.res_mode = None, 0
    d.y, d.y_cs, d.y_co = y.t.data_ptr(), y.cs, y.co
    d.N, d.H, d.W, d.Cin, d.Cout = n, h, w, pc.cin_pad, pc.cout
    assert x.c == pc.cin_pad, "conv2d: input view has {} channels, packed weights expect {}".format(x.c, pc.cin_pad)
    assert y.c == pc.cout
    ho = h if pc.stride == 1 else (h - 1) // 2 + 1
    wo = w if pc.stride == 1 else (w - 1) // 2 + 1
    assert tuple(y.t.shape[:3]) == (n, ho, wo), (tuple(y.t.shape), (n, ho, wo))
    d.ksize, d.stride = pc.k, pc.stride
    d.relu_upto = (pc.cout if relu else 0) if relu_upto is None else relu_upto
    d.in_relu = int(in_relu)
    if in_affine is not None:
        d.in_scale, d.in_shift = in_affine[0].data_ptr(), in_affine[1].data_ptr()
    else:
        d.in_scale = d.in_shift = None


# ---- tile-variant autotuning (host side; the library itself stays stateless) -----------------------------------
FUSE_POOL = os.environ.get("CMK_FUSE_POOL", "1") != "0"      # eSE: average-pool partial sums from the aggregation conv's epilogue (A/B switch)
PAIR_TOWERS = os.environ.get("CMK_PAIR_TOWERS", "1") != "0"  # FCOS head: conv k of the cls and the bbox tower in one launch (A/B switch)
ALLOW_SPLIT_BF16 = os.environ.get("CMK_ALLOW_SPLIT_BF16", "0") == "1"   # OPT-IN: pack the bf16-split weights and let the tuner / tables use the
                          # pointwise GEMM's split form (fp32-accurate products from bf16 pieces, cmk.h tune_wm 10).  Off: nothing in the package uses it.
ALLOW_SPLIT_F16 = os.environ.get("CMK_ALLOW_SPLIT_F16", "0") == "1"   # OPT-IN: pack the fp16 two-piece weights and let the tuner / tables use the direct
                          # 3x3 form on fp16-split products (cmk.h tune_wm 11, conv_sp3.hip: 22-bit operands, three products, fp32 accumulation —
                          # the error of an fp32 accumulation).  Off: nothing in the package uses it.
PACK_WINO6 = True         # pack the F(4x4,3x3) weights too (4x the filter bank per 3x3 stride-1 conv)
ALLOW_WINOGRAD = True     # let the tuner pick the Winograd F(2x2,3x3) kernel where it is faster (fp32, differs by rounding only)
TUNE_LOG = []             # (key, {candidate: ms}) per tuned problem
FORCE_VARIANT = None      # (wm, sc, wn[, splitk]) for every conv launched through the wrappers below (tests, A/B tools); None = table/tuner/default
TUNE_ONLY = None          # callable(key) -> [(wm, sc, wn, splitk), ...]: the tuner's candidates for that problem instead of the whole menu (targeted
                          # re-tuning: tools/tune_sp3.py); TUNE_REPS timed launches per candidate, best of TUNE_ROUNDS interleaved rounds
TUNE_REPS, TUNE_ROUNDS = 2, 1
AUTOTUNE = False          # when True, the first call of every distinct conv problem times the variant menu (needs an idle, non-capturing stream)
_TUNED = {}               # problem key -> (wm, sc, wn)


def set_autotune(flag: bool) -> None:
    global AUTOTUNE
    AUTOTUNE = bool(flag)


def tuned_variants() -> dict:
    return dict(_TUNED)


def _key_to_str(key) -> str:
    k, stride, cin, cout, xcs, ycs, res, shapes = key
    return "k{}s{}_cin{}_cout{}_xcs{}_ycs{}_res{}_".format(k, stride, cin, cout, xcs, ycs, res) + "+".join("{}x{}x{}".format(*s) for s in shapes)


def _str_to_key(s: str):
    head, shapes = s.rsplit("_", 1)
    f = head.split("_")
    k, stride = f[0][1:].split("s")
    vals = [int(f[1][3:]), int(f[2][4:]), int(f[3][3:]), int(f[4][3:]), int(f[5][3:])]
    return (int(k), int(stride), vals[0], vals[1], vals[2], vals[3], vals[4], tuple(tuple(int(v) for v in t.split("x")) for t in shapes.split("+")))


def save_tuned(path: str) -> None:
    """Write the measured variant table (problem -> [wm, sc, wn]) as JSON; shipped tables live in centermask2_amd/tuned/."""
    import json
    with open(path, "w") as f:
        json.dump({_key_to_str(k): list(v) for k, v in sorted(_TUNED.items(), key=lambda kv: _key_to_str(kv[0]))}, f, indent=0)


def _variant_on_menu(tv) -> bool:
    """(wm, sc, wn[, splitk]) names a kernel this library has (older tables may carry variants that were removed since)."""
    wm, sc, wn = tv[:3]
    sk = tv[3] if len(tv) > 3 else 1
    if wm == 6 and sc == 64:          # F(4x4,3x3), shared-V form (conv_wino6s.hip)
        return wn in (1, 2) and sk == 1
    if wm == 10:                      # pointwise GEMM from bf16-split products: only where the caller opted in
        return ALLOW_SPLIT_BF16 and sc == 32 and wn == 4 and sk == 1
    if wm == 12:                      # pointwise GEMM from fp16-split products (two pieces, three products)
        return ALLOW_SPLIT_F16 and sc == 32 and wn == 4 and sk == 1
    if wm == 11:                      # direct 3x3 conv from bf16-split products (conv_sp3.hip): sc = pieces, wn = geometry
        return ALLOW_SPLIT_F16 and sc in (2, 21) and 0 <= wn <= 3 and sk == 1
    return (wm in (1, 2, 5, 6, 7, 8, 9) and sc in (16, 32) and 1 <= wn <= 7 and sk in (1, 2, 4, 8)) or tuple(tv[:3]) == (0, 0, 0)


def load_tuned(path: str) -> int:
    """Read a measured variant table; entries naming a variant that is not on the menu are dropped (that problem falls back to the
    library default / the start-up tuner), so a stale table cannot break a run."""
    import json
    with open(path) as f:
        table = json.load(f)
    n = 0
    for k, v in table.items():
        if _variant_on_menu(v):
            _TUNED[_str_to_key(k)] = tuple(v)
            n += 1
    return n


def _out_pixels(d) -> int:
    ho = d.H if d.stride == 1 else (d.H - 1) // 2 + 1
    wo = d.W if d.stride == 1 else (d.W - 1) // 2 + 1
    return d.N * ho * wo


def _set_variant(descs, n, tv):
    """Write a (wm, sc, wn[, splitk]) choice into the descriptors; returns the split-K workspace (keep it alive until the launch)."""
    wm, sc, wn = tv[:3]
    sk = tv[3] if len(tv) > 3 else 1
    for i in range(n):
        descs[i].tune_wm, descs[i].tune_sc, descs[i].tune_wn = wm, sc, wn
        descs[i].splitk, descs[i].splitk_ws = 0, None
    ws = None
    if sk > 1:
        d = descs[0]
        ws = torch.empty((sk * _out_pixels(d) * _lib.load().cmk_conv_cout_pad(d.Cout),), dtype=torch.float32,
                         device=torch.device("cuda", torch.cuda.current_device()))       # == the inputs' device (_need_current_device)
        d.splitk, d.splitk_ws = sk, ws.data_ptr()
    return ws


def _tune(descs, n, key) -> None:
    """Time every available (wm, sc, wn, splitk) variant on the real buffers and remember the fastest.  The direct variants
    (incl. the gather form and split-K, whose K order is unchanged up to the final sum) agree to rounding; variants the
    library rejects for this shape are skipped."""
    lib = _lib.load()
    st = _stream()
    d0 = descs[0]

    def run():
        return lib.cmk_conv2d_nhwc_multi(descs, n, st) if n > 1 else lib.cmk_conv2d_nhwc(ctypes.byref(descs[0]), st)

    best, best_ms = (0, 0, 0), float("inf")
    small = n == 1 and d0.res_mode != 2 and _out_pixels(d0) <= 32768      # few M tiles: split-K / gather forms can pay
    sks = (1, 2, 4, 8) if small else (1,)
    cands = [(wm, sc, wn, sk) for wn in range(1, 8) for wm in (1, 2) for sc in (16, 32) for sk in sks]
    if small and d0.ksize == 3:
        cands += [(7, 32, wn, sk) for wn in (1, 2, 4) for sk in sks]      # gather form
    if d0.ksize == 3:
        cands += [(9, 32, mt, sk) for mt in (4, 2) for sk in sks]   # gather form of a 3x3 conv on the pointwise GEMM kernel (conv_pw.hip GA)
        if ALLOW_SPLIT_BF16:
            cands += [(10, 32, 4, 1)]                               # ... and its opt-in bf16-split form
        if ALLOW_SPLIT_F16:
            cands += [(12, 32, 4, 1)]                               # ... and the fp16 two-piece form of the gather GEMM
            cands += [(11, 2, g, 1) for g in range(4)]              # opt-in: direct 3x3 on two fp16 pieces per operand (conv_sp3.hip), four tile geometries
            cands += [(11, 21, g, 1) for g in range(4)]             # ... one cout tile per wave: less cout padding, more and smaller workgroups
    if d0.ksize == 1:
        cands += [(8, 32, mt, sk) for mt in (4, 2) for sk in sks]   # pointwise GEMM kernel (conv_pw.hip), 256- or 128-pixel workgroups; same K order, same bits
        if ALLOW_SPLIT_BF16:
            cands += [(10, 32, 4, 1)]                               # ... its opt-in bf16-split form (the library refuses it where it does not apply)
        if ALLOW_SPLIT_F16:
            cands += [(12, 32, 4, 1)]                               # ... its opt-in fp16-split form (two pieces, three products)
    if ALLOW_WINOGRAD:
        cands += [(5, 16, 2, 1)]                  # fused Winograd F(2x2,3x3)
        cands += [(6, 16, 1, 1), (6, 16, 2, 1)]   # fused Winograd F(4x4,3x3): map tiles / pairs of RoI maps (the library rejects what does not apply)
        cands += [(6, 64, 1, 1), (6, 64, 2, 1)]   # ... its shared-V form: 64 couts per workgroup from one frequency image (conv_wino6s.hip)
        if small and d0.ksize == 3:
            cands += [(6, 16, 1, sk) for sk in (2, 4)]      # ... with the chunk loop split over 2 / 4 workgroups (launches of about one round)
    if TUNE_ONLY is not None:
        cands = [tuple(tv) + (1,) * (4 - len(tv)) for tv in TUNE_ONLY(key)]
    times = {}
    for _ in range(TUNE_ROUNDS):
        for tv in cands:
            if times.get(tv) == float("inf"):
                continue
            ws = _set_variant(descs, n, tv)
            if run() != 0:
                times[tv] = float("inf")          # not on the menu for this shape
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _r in range(TUNE_REPS):
                run()
            e1.record()
            e1.synchronize()
            times[tv] = min(times.get(tv, float("inf")), e0.elapsed_time(e1) / TUNE_REPS)
            del ws
    for tv in cands:                              # first of equals wins, as before
        if times.get(tv, float("inf")) < best_ms:
            best, best_ms = (tv[:3] if tv[3] == 1 else tv), times[tv]
    _TUNED[key] = best
    TUNE_LOG.append((key, {tv: times[tv] for tv in cands if tv in times}))


def _default_is_wino6(descs, n, pc) -> bool:
    """The library's untuned choice between the two Winograd forms (conv_igemm.hip run()): F(4x4,3x3) when its 12x40 tiles are at
    least 55 % full over the launch and there are at least 256 workgroups."""
    if getattr(pc, "w_wino6", None) is None or pc.k != 3 or pc.stride != 1 or pc.cin_pad < 32:
        return False
    cd = lambda a, b: -(-a // b)
    px = sum(descs[i].N * descs[i].H * descs[i].W for i in range(n))
    tiles = sum(descs[i].N * cd(descs[i].H, 12) * cd(descs[i].W, 40) for i in range(n))
    return px >= 0.55 * tiles * 480 and tiles * cd(pc.cout, 32) >= 256


def _default_variant(d):
    """No table entry and no tuner: library defaults, plus split-K for skinny 1x1 GEMMs (maskiou_fc1: 400 x 12544 x 1024)."""
    if d.ksize == 1 and d.res_mode != 2 and _out_pixels(d) <= 1024 and d.Cin >= 4096:
        chunks = d.Cin // 16
        for sk in (4, 2):
            if chunks % (2 * sk) == 0:
                return (0, 0, 0, sk)
    return (0, 0, 0)


def _apply_tuning(descs, n, key):
    if FORCE_VARIANT is not None:
        return _set_variant(descs, n, FORCE_VARIANT)
    tv = _TUNED.get(key)
    if tv is None and AUTOTUNE and not torch.cuda.is_current_stream_capturing():
        _tune(descs, n, key)
        tv = _TUNED[key]
    if tv is None:
        tv = _default_variant(descs[0]) if n == 1 else (0, 0, 0)
    return _set_variant(descs, n, tv)


def _problem_key(descs, n):
    d = descs[0]
    # res slot: 0/1/2 = residual mode, +4 when the fused input affine is on (restricts the variants that may be chosen)
    return (d.ksize, d.stride, d.Cin, d.Cout, d.x_cs, d.y_cs, d.res_mode + (4 if d.in_scale else 0),
            tuple((descs[i].N, descs[i].H, descs[i].W) for i in range(n)))


def conv2d(x: View, pc: PackedConv, y: View, relu: bool = False, relu_upto: Optional[int] = None,
           res: Optional[View] = None, res_upsample: bool = False, in_relu: bool = False, pool: Optional[list] = None) -> None:
    """pool: a list that receives (partial sums, rows per record) when the kernel this conv runs on also leaves the average-pool partial
    sums of its output behind (cmk_conv_desc.pool_ws: the pointwise GEMM kernel, for the eSE gate); left empty otherwise."""
    lib = _lib.load()
    descs = (ConvDesc * 1)()
    _fill_desc(descs[0], x, pc, y, relu, relu_upto, res, res_upsample, in_relu)
    ws = _apply_tuning(descs, 1, _problem_key(descs, 1))      # split-K workspace (if any) stays referenced across the launch
    if pool is not None and FUSE_POOL:
        rows = lib.cmk_conv_pool_rows(ctypes.byref(descs[0]))
        if rows > 0:
            d = descs[0]
            pws = torch.empty((2 * (-(-(d.N * d.H * d.W) // rows)), pc.cout), dtype=torch.float32, device=y.t.device)
            d.pool_ws = pws.data_ptr()
            pool.append((pws, rows))
    check(lib.cmk_conv2d_nhwc(ctypes.byref(descs[0]), _stream()), "cmk_conv2d_nhwc")
    del ws


def conv2d_multi(xs: Sequence[View], pcs: Sequence[PackedConv], ys: Sequence[View], relu: bool = False,
                 relu_upto: Optional[int] = None, in_affine=None) -> None:
    """One launch over several inputs that share the packed weights (pcs[i].w is the same tensor; scale/shift may differ).
    in_affine[i] = (scale, shift) of shape (N, Cin): the producer's GroupNorm+ReLU applied while staging input i."""
    lib = _lib.load()
    n = len(xs)
    descs = (ConvDesc * n)()
    for i in range(n):
        assert pcs[i].w.data_ptr() == pcs[0].w.data_ptr()
        _fill_desc(descs[i], xs[i], pcs[i], ys[i], relu, relu_upto, None, False, False, in_affine[i] if in_affine is not None else None)
    ws = _apply_tuning(descs, n, _problem_key(descs, n))
    check(lib.cmk_conv2d_nhwc_multi(descs, n, _stream()), "cmk_conv2d_nhwc_multi")
    del ws


def conv_gn_multi(xs: Sequence[View], pcs: Sequence[PackedConv], gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32,
                  eps: float = 1e-5, in_affine=None):
    """Tower conv (no activation) over several levels + the statistics of the GroupNorm that follows (fcos.py:182-186).
    Returns (raw conv outputs, [(scale, shift)] per level) — the affine is applied by the NEXT conv while staging.
    When the Winograd kernel runs the conv, its epilogue produces the statistics (no pass over the output)."""
    lib = _lib.load()
    n = len(xs)
    pc = pcs[0]
    ys = [View(torch.empty((x.t.shape[0], x.t.shape[1], x.t.shape[2], pc.cout), dtype=torch.float32, device=x.t.device)) for x in xs]
    descs = (ConvDesc * n)()
    for i in range(n):
        assert pcs[i].w.data_ptr() == pc.w.data_ptr()
        _fill_desc(descs[i], xs[i], pcs[i], ys[i], False, None, None, False, False, in_affine[i] if in_affine is not None else None)
    key = _problem_key(descs, n)
    ws = _apply_tuning(descs, n, key)

    def launch():
        if PROFILE is None:
            return check(lib.cmk_conv2d_nhwc_multi(descs, n, _stream()), "cmk_conv2d_nhwc_multi")
        taps = pc.k * pc.k
        flops = sum(2.0 * y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * pc.cin * pc.cout * taps for y in ys)
        nbytes = sum(4.0 * y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * (pc.cin + pc.cout) for y in ys) + 4.0 * pc.cin * pc.cout * taps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.cmk_conv2d_nhwc_multi(descs, n, _stream()), "cmk_conv2d_nhwc_multi")
        e1.record()
        PROFILE.append((_kernel_name(taps, 1, _TUNED.get(key), in_affine is not None), flops, nbytes, e0, e1, None,
                        executed_flops(taps, 1, _TUNED.get(key), [tuple(y.t.shape[:3]) for y in ys], pc.cin_pad, pc.cout)))

    d0 = descs[0]
    cpg = pc.cout // groups if groups > 0 and pc.cout % groups == 0 else 0
    untuned = (d0.tune_wm, d0.tune_sc, d0.tune_wn) == (0, 0, 0)
    wino = d0.tune_wm in (5, 6, 11) or (untuned and pc.w_wino is not None and pc.cin_pad >= 32 and pc.stride == 1)
    gn_form = 110 + d0.tune_wn if d0.tune_wm == 11 else 6 if (d0.tune_wm == 6 or (untuned and _default_is_wino6(descs, n, pc))) else 5
    fused = wino and 0 < cpg <= 32 and (cpg & (cpg - 1)) == 0 and all(x.t.shape[0] == xs[0].t.shape[0] for x in xs)
    if not fused:
        launch()
        del ws
        return ys, groupnorm_affine_multi([y.t for y in ys], gamma, beta, groups, eps)
    nimg, dev = xs[0].t.shape[0], xs[0].t.device
    recs_l = [lib.cmk_conv_gn_records(y.t.shape[1], y.t.shape[2], gn_form) for y in ys]
    gws = torch.empty((nimg * sum(recs_l), groups, 2), dtype=torch.float64, device=dev)
    for i in range(n):
        descs[i].gn_ws, descs[i].gn_groups = gws.data_ptr(), groups
    launch()
    out = [(torch.empty((nimg, pc.cout), dtype=torch.float32, device=dev), torch.empty((nimg, pc.cout), dtype=torch.float32, device=dev)) for _ in ys]
    hs, wss, recs = (ctypes.c_int * n)(), (ctypes.c_int * n)(), (ctypes.c_int * n)()
    ps, pb = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    for i, y in enumerate(ys):
        hs[i], wss[i], recs[i] = y.t.shape[1], y.t.shape[2], recs_l[i]
        ps[i], pb[i] = out[i][0].data_ptr(), out[i][1].data_ptr()
    check(lib.cmk_groupnorm_affine_tiles(gws.data_ptr(), hs, wss, recs, n, gamma.data_ptr(), beta.data_ptr(), nimg, pc.cout, groups, eps, ps, pb,
                                         _stream()), "cmk_groupnorm_affine_tiles")
    del ws
    return ys, out


def conv_gn_multi_pair(xs_a: Sequence[View], pc_a: PackedConv, gn_a, xs_b: Sequence[View], pc_b: PackedConv, gn_b, groups: int = 32, eps: float = 1e-5,
                       in_affine_a=None, in_affine_b=None):
    """Two tower convs with DIFFERENT weights (the cls and the bbox tower of the FCOS head, fcos.py:227-231) over the same level shapes in
    ONE launch of up to 10 problems — half the launch ramps and tails of two launches — each followed by its own GroupNorm statistics
    (gn_x = (gamma, beta)).  Only the F(4x4) map kernels take per-problem weights: returns None when the measured / default variant of this
    problem is another kernel or the fused statistics do not apply (the caller then runs the two towers separately).
    Returns ((ys_a, affine_a), (ys_b, affine_b)) like two conv_gn_multi calls."""
    lib = _lib.load()
    na, nb = len(xs_a), len(xs_b)
    n = na + nb
    if not PAIR_TOWERS:
        return None
    if na != nb or n > 10 or pc_a.cout != pc_b.cout or pc_a.cin_pad != pc_b.cin_pad or pc_a.w_wino6 is None or pc_b.w_wino6 is None:
        return None
    if (in_affine_a is None) != (in_affine_b is None) or any(tuple(a.t.shape) != tuple(b.t.shape) for a, b in zip(xs_a, xs_b)):
        return None
    cpg = pc_a.cout // groups if groups > 0 and pc_a.cout % groups == 0 else 0
    if not (0 < cpg <= 32 and (cpg & (cpg - 1)) == 0) or any(x.t.shape[0] != xs_a[0].t.shape[0] for x in xs_a):
        return None
    xs, pcs = list(xs_a) + list(xs_b), [pc_a] * na + [pc_b] * nb
    affs = (list(in_affine_a) + list(in_affine_b)) if in_affine_a is not None else None
    ys = [View(torch.empty((x.t.shape[0], x.t.shape[1], x.t.shape[2], pc_a.cout), dtype=torch.float32, device=x.t.device)) for x in xs]
    descs = (ConvDesc * n)()
    for i in range(n):
        _fill_desc(descs[i], xs[i], pcs[i], ys[i], False, None, None, False, False, affs[i] if affs is not None else None)
    # the variant measured for ONE tower's launch (the table is keyed by the 5 level shapes) decides; the 10-problem launch has no entry of its own
    half = (ConvDesc * na)()
    for i in range(na):
        _fill_desc(half[i], xs[i], pcs[i], ys[i], False, None, None, False, False, affs[i] if affs is not None else None)
    key = _problem_key(half, na)
    tv = FORCE_VARIANT if FORCE_VARIANT is not None else _TUNED.get(key)
    if tv is None:
        tv = (6, 16, 1) if _default_is_wino6(half, na, pc_a) else None
    sp3 = tv is not None and tv[0] == 11 and pc_a.w_splith is not None and pc_b.w_splith is not None      # opt-in direct fp16-split form
    if tv is None or not (sp3 or (tv[0] == 6 and tv[2] == 1)):
        return None
    _set_variant(descs, n, tv)
    nimg, dev = xs[0].t.shape[0], xs[0].t.device
    recs_l = [lib.cmk_conv_gn_records(y.t.shape[1], y.t.shape[2], 110 + tv[2] if sp3 else 6) for y in ys]
    gws = torch.empty((nimg * sum(recs_l), groups, 2), dtype=torch.float64, device=dev)
    for i in range(n):
        descs[i].gn_ws, descs[i].gn_groups = gws.data_ptr(), groups
    taps = 9
    if PROFILE is None:
        check(lib.cmk_conv2d_nhwc_multi(descs, n, _stream()), "cmk_conv2d_nhwc_multi (tower pair)")
    else:
        flops = sum(2.0 * y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * pc_a.cin * pc_a.cout * taps for y in ys)
        nbytes = sum(4.0 * y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * (pc_a.cin + pc_a.cout) for y in ys) + 2 * 4.0 * pc_a.cin * pc_a.cout * taps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.cmk_conv2d_nhwc_multi(descs, n, _stream()), "cmk_conv2d_nhwc_multi (tower pair)")
        e1.record()
        PROFILE.append((_kernel_name(taps, 1, tv, affs is not None), flops, nbytes, e0, e1, None,
                        executed_flops(taps, 1, tv, [tuple(y.t.shape[:3]) for y in ys], pc_a.cin_pad, pc_a.cout)))
    out = []
    rec_off = 0
    for t, (gamma, beta) in enumerate((gn_a, gn_b)):
        sl = slice(t * na, (t + 1) * na)
        res = [(torch.empty((nimg, pc_a.cout), dtype=torch.float32, device=dev), torch.empty((nimg, pc_a.cout), dtype=torch.float32, device=dev)) for _ in range(na)]
        hs, wss, recs = (ctypes.c_int * na)(), (ctypes.c_int * na)(), (ctypes.c_int * na)()
        ps, pb = (ctypes.c_void_p * na)(), (ctypes.c_void_p * na)()
        for i, y in enumerate(ys[sl]):
            hs[i], wss[i], recs[i] = y.t.shape[1], y.t.shape[2], recs_l[t * na + i]
            ps[i], pb[i] = res[i][0].data_ptr(), res[i][1].data_ptr()
        # the records of the second tower's problems follow those of the first (spatial tiles are numbered problem after problem)
        check(lib.cmk_groupnorm_affine_tiles(gws.data_ptr() + rec_off * groups * 2 * 8, hs, wss, recs, na, gamma.data_ptr(), beta.data_ptr(), nimg, pc_a.cout,
                                             groups, eps, ps, pb, _stream()), "cmk_groupnorm_affine_tiles")
        rec_off += nimg * sum(recs_l[sl])
        out.append((ys[sl], res))
    return out[0], out[1]


def conv_out_multi(xs: Sequence[View], pcs: Sequence[PackedConv], **kw) -> List[View]:  # kw: relu, relu_upto, in_affine
    ys = [View(torch.empty((x.t.shape[0], x.t.shape[1], x.t.shape[2], pcs[0].cout), dtype=torch.float32, device=x.t.device)) for x in xs]
    conv2d_multi(xs, pcs, ys, **kw)
    return ys


def conv_out(x: View, pc: PackedConv, **kw) -> View:
    n, h, w = x.nhw
    ho = h if pc.stride == 1 else (h - 1) // 2 + 1
    wo = w if pc.stride == 1 else (w - 1) // 2 + 1
    y = View(torch.empty((n, ho, wo, pc.cout), dtype=torch.float32, device=x.t.device))
    conv2d(x, pc, y, **kw)
    return y


def linear(x2d: torch.Tensor, pc: PackedConv, relu: bool = False) -> torch.Tensor:
    """y = x @ W^T + b as a 1x1 conv over the rows (maskiou_head.py:89-91, 116-119)."""
    r, k = x2d.shape
    y = torch.empty((1, 1, r, pc.cout), dtype=torch.float32, device=x2d.device)
    conv2d(View(x2d.reshape(1, 1, r, k)), pc, View(y), relu=relu)
    return y.reshape(r, pc.cout)


# ---------------------------------------------------------------------------------------------------------------
# backbone pieces
# ---------------------------------------------------------------------------------------------------------------
def stem_conv(x_nchw: torch.Tensor, w27: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor) -> View:
    lib = _lib.load()
    _need_gpu(x_nchw, "stem_conv")
    x_nchw = x_nchw.contiguous()
    n, c, h, w = x_nchw.shape
    assert c == 3
    cout = w27.shape[1]
    y = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, cout), dtype=torch.float32, device=x_nchw.device)
    check(lib.cmk_stem_conv_nchw3(x_nchw.data_ptr(), w27.data_ptr(), scale.data_ptr(), shift.data_ptr(), y.data_ptr(),
                                  n, h, w, cout, _stream()), "cmk_stem_conv_nchw3")
    return View(y)


def pack_dw_weight(weight: torch.Tensor) -> torch.Tensor:
    """(C,1,3,3) depth-wise weight -> tap-major [9][C] (the layout cmk_dwconv3x3_nhwc reads)."""
    c = weight.shape[0]
    assert tuple(weight.shape) == (c, 1, 3, 3), "depth-wise 3x3 weight must be (C,1,3,3)"
    return weight.detach().float().cpu().reshape(c, 9).t().contiguous()


def dwconv3x3(x: View, w9c: torch.Tensor, y: Optional[View] = None, stride: int = 1) -> View:
    """Depth-wise 3x3, pad 1, no bias / activation (vovnet.py:110-119)."""
    lib = _lib.load()
    _need_gpu(x.t, "dwconv3x3")
    n, h, w = x.nhw
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    if y is None:
        y = View(torch.empty((n, ho, wo, x.c), dtype=torch.float32, device=x.t.device))
    assert y.nhw == (n, ho, wo) and y.c == x.c and tuple(w9c.shape) == (9, x.c)
    check(lib.cmk_dwconv3x3_nhwc(x.t.data_ptr(), x.cs, x.co, w9c.data_ptr(), y.t.data_ptr(), y.cs, y.co, n, h, w, x.c, stride, _stream()),
          "cmk_dwconv3x3_nhwc")
    return y


def maxpool3x3s2_ceil(x: View, y: Optional[View] = None, gate: Optional[torch.Tensor] = None) -> View:
    lib = _lib.load()
    _need_gpu(x.t, "maxpool3x3s2_ceil")
    n, h, w = x.nhw
    ho = -(-(h - 3) // 2) + 1
    wo = -(-(w - 3) // 2) + 1
    if (ho - 1) * 2 >= h:
        ho -= 1
    if (wo - 1) * 2 >= w:
        wo -= 1
    if y is None:
        y = View(torch.empty((n, ho, wo, x.c), dtype=torch.float32, device=x.t.device))
    assert tuple(y.t.shape[:3]) == (n, ho, wo) and y.c == x.c
    check(lib.cmk_maxpool3x3s2_ceil_nhwc(x.t.data_ptr(), x.cs, x.co, y.t.data_ptr(), y.cs, y.co, n, h, w, x.c,
                                         gate.data_ptr() if gate is not None else None, _stream()),
          "cmk_maxpool3x3s2_ceil_nhwc")
    return y


def maxpool1x1s2(x: View) -> View:
    """MaxPool2d(kernel_size=1, stride=2): every second pixel (d2 LastLevelMaxPool; vovnet.py:504-524)."""
    lib = _lib.load()
    _need_gpu(x.t, "maxpool1x1s2")
    n, h, w = x.nhw
    y = View(torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, x.c), dtype=torch.float32, device=x.t.device))
    check(lib.cmk_maxpool1x1s2_nhwc(x.t.data_ptr(), x.cs, x.co, y.t.data_ptr(), y.cs, y.co, n, h, w, x.c, _stream()), "cmk_maxpool1x1s2_nhwc")
    return y


def _ese_chunks(hw: int, c: int) -> int:
    """Pixel chunks per image for the eSE average pool: ~32K elements per workgroup so even the 25x40 stage fills the chip."""
    return max(1, min(256, (hw * c) // 32768))


def ese_gate(x: View, fc_w: torch.Tensor, fc_b: torch.Tensor) -> torch.Tensor:
    """gate (N,C) = hsigmoid(fc(mean_HW(x)))   (vovnet.py:255-259)."""
    lib = _lib.load()
    _need_gpu(x.t, "ese_gate")
    n, h, w = x.nhw
    hw, c = h * w, x.c
    chunks = _ese_chunks(hw, c)
    ws = torch.empty((n, chunks, c), dtype=torch.float32, device=x.t.device)
    gate = torch.empty((n, c), dtype=torch.float32, device=x.t.device)
    check(lib.cmk_ese_gate(x.t.data_ptr(), x.cs, x.co, fc_w.data_ptr(), fc_b.data_ptr(), gate.data_ptr(), ws.data_ptr(), chunks,
                           n, hw, c, _stream()), "cmk_ese_gate")
    return gate


def ese_gate_pooled(pooled, fc_w: torch.Tensor, fc_b: torch.Tensor, n: int, hw: int) -> torch.Tensor:
    """The same gate from the partial sums the producing conv left behind (conv2d(..., pool=[...])): no pass over the map."""
    lib = _lib.load()
    pws, rows = pooled
    c = pws.shape[1]
    gate = torch.empty((n, c), dtype=torch.float32, device=pws.device)
    check(lib.cmk_ese_gate_pooled(pws.data_ptr(), rows, fc_w.data_ptr(), fc_b.data_ptr(), gate.data_ptr(), n, hw, c, _stream()), "cmk_ese_gate_pooled")
    return gate


def ese(x: View, fc_w: torch.Tensor, fc_b: torch.Tensor, y: View, identity: Optional[View] = None, gate: Optional[torch.Tensor] = None) -> None:
    """y = x * hsigmoid(fc(mean_HW(x))) (+ identity)   (vovnet.py:255-260, :329-330).  gate: already computed (ese_gate_pooled)."""
    lib = _lib.load()
    _need_gpu(x.t, "ese")
    n, h, w = x.nhw
    hw, c = h * w, x.c
    if gate is None:
        chunks = _ese_chunks(hw, c)
        ws = torch.empty((n, chunks, c), dtype=torch.float32, device=x.t.device)
        gate = torch.empty((n, c), dtype=torch.float32, device=x.t.device)
        check(lib.cmk_ese_gate(x.t.data_ptr(), x.cs, x.co, fc_w.data_ptr(), fc_b.data_ptr(), gate.data_ptr(), ws.data_ptr(), chunks,
                               n, hw, c, _stream()), "cmk_ese_gate")
    idp, idcs, idco = (identity.t.data_ptr(), identity.cs, identity.co) if identity is not None else (None, 0, 0)
    check(lib.cmk_ese_scale(x.t.data_ptr(), x.cs, x.co, gate.data_ptr(), idp, idcs, idco, y.t.data_ptr(), y.cs, y.co,
                            n, hw, c, _stream()), "cmk_ese_scale")


def groupnorm_affine(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5):
    """GroupNorm statistics of a dense NHWC tensor as per-(image, channel) (scale, shift); the consumer conv applies
    relu(x*scale + shift) while staging (cmk_conv_desc.in_scale/in_shift), so the normalised tensor is never written."""
    lib = _lib.load()
    _need_gpu(x, "groupnorm_affine")
    n, h, w, c = x.shape
    hw = h * w
    chunks = max(1, min(128, hw // 128))
    ws = torch.empty((n, groups, chunks, 2), dtype=torch.float64, device=x.device)
    sc = torch.empty((n, c), dtype=torch.float32, device=x.device)
    sh = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib.cmk_groupnorm_affine(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), chunks, n, hw, c, groups, eps,
                                   sc.data_ptr(), sh.data_ptr(), _stream()), "cmk_groupnorm_affine")
    return sc, sh


def groupnorm_affine_multi(xs: Sequence[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5):
    """groupnorm_affine for several dense NHWC tensors with the same (N, C) — the FPN levels — in two launches."""
    lib = _lib.load()
    nl = len(xs)
    n, c = xs[0].shape[0], xs[0].shape[3]
    dev = xs[0].device
    chunks = 64
    ws = torch.empty((nl, n, groups, chunks, 2), dtype=torch.float64, device=dev)
    out = [(torch.empty((n, c), dtype=torch.float32, device=dev), torch.empty((n, c), dtype=torch.float32, device=dev)) for _ in xs]
    px, ps, pb = (ctypes.c_void_p * nl)(), (ctypes.c_void_p * nl)(), (ctypes.c_void_p * nl)()
    hws = (ctypes.c_int * nl)()
    for i, x in enumerate(xs):
        _need_gpu(x, "groupnorm_affine_multi")
        assert x.is_contiguous() and x.shape[0] == n and x.shape[3] == c
        px[i], ps[i], pb[i] = x.data_ptr(), out[i][0].data_ptr(), out[i][1].data_ptr()
        hws[i] = x.shape[1] * x.shape[2]
    check(lib.cmk_groupnorm_affine_multi(px, hws, nl, gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), chunks, n, c, groups, eps, ps, pb,
                                         _stream()), "cmk_groupnorm_affine_multi")
    return out


def groupnorm_relu_(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5, relu: bool = True) -> None:
    """GroupNorm (+ ReLU) in place on a dense NHWC tensor (fcos.py:182-186; relu=False: d2's get_norm("GN") behind a conv without activation)."""
    lib = _lib.load()
    _need_gpu(x, "groupnorm_relu_")
    n, h, w, c = x.shape
    hw = h * w
    chunks = max(1, min(128, hw // 128))
    ws = torch.empty((n, groups, chunks, 2), dtype=torch.float64, device=x.device)
    fn = lib.cmk_groupnorm_relu_nhwc if relu else lib.cmk_groupnorm_nhwc
    check(fn(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), chunks, n, hw, c, groups, eps, _stream()), "cmk_groupnorm[_relu]_nhwc")


def upsample2x_add_(y: View, coarse: View) -> None:
    """y += nearest-2x-upsampling(coarse), dense NHWC tensors (d2 FPN top-down sum behind a norm)."""
    lib = _lib.load()
    _need_gpu(y.t, "upsample2x_add_")
    assert y.co == 0 and coarse.co == 0 and y.cs == y.c and coarse.cs == coarse.c and y.c == coarse.c, "upsample2x_add_ needs dense NHWC tensors"
    n, h, w = y.nhw
    check(lib.cmk_upsample2x_add_nhwc(y.t.data_ptr(), coarse.t.data_ptr(), n, h, w, coarse.t.shape[1], coarse.t.shape[2], y.c, _stream()), "cmk_upsample2x_add_nhwc")


# ---------------------------------------------------------------------------------------------------------------
# FCOS post-head: candidate selection, sort + NMS + top-k (all counts stay on the device)
# ---------------------------------------------------------------------------------------------------------------
def fcos_select(logits: Sequence[torch.Tensor], regctr: Sequence[torch.Tensor], strides: Sequence[int], thresh: float, cap: int,
                thresh_with_ctr: bool = False):
    """logits[l]: (N,H,W,C) dense NHWC; regctr[l]: (N,H,W,5).  Returns dict of candidate buffers (N,cap,...) + counts (N)."""
    lib = _lib.load()
    n, c = logits[0].shape[0], logits[0].shape[3]
    dev = logits[0].device
    _need_gpu(logits[0], "fcos_select")
    nl = len(logits)
    lv = (FcosLevel * nl)()
    for i, (lg, rc, s) in enumerate(zip(logits, regctr, strides)):
        assert lg.is_contiguous() and rc.is_contiguous() and rc.shape[3] == 5 and lg.shape[:3] == rc.shape[:3]
        lv[i].logits, lv[i].regctr = lg.data_ptr(), rc.data_ptr()
        lv[i].H, lv[i].W, lv[i].stride = lg.shape[1], lg.shape[2], int(s)
    wslen = lib.cmk_fcos_select_ws_len(lv, nl, n, c)
    assert wslen > 0
    ws = torch.empty((wslen,), dtype=torch.int32, device=dev)
    out = dict(box=torch.empty((n, cap, 4), dtype=torch.float32, device=dev),
               score=torch.empty((n, cap), dtype=torch.float32, device=dev),
               cls=torch.empty((n, cap), dtype=torch.int32, device=dev),
               loc=torch.empty((n, cap, 2), dtype=torch.float32, device=dev),
               counts=torch.empty((n,), dtype=torch.int32, device=dev), cap=cap)
    check(lib.cmk_fcos_select(lv, nl, n, c, float(thresh), int(bool(thresh_with_ctr)), out["box"].data_ptr(), out["score"].data_ptr(), out["cls"].data_ptr(),
                              out["loc"].data_ptr(), out["counts"].data_ptr(), ws.data_ptr(), wslen, cap, _stream()), "cmk_fcos_select")
    return out


def nms_topk(cand: dict, iou_thr: float, topk: int):
    lib = _lib.load()
    n, cap = cand["score"].shape
    dev = cand["score"].device
    _need_gpu(cand["score"], "nms_topk")
    out = dict(box=torch.empty((n, topk, 4), dtype=torch.float32, device=dev),
               score=torch.empty((n, topk), dtype=torch.float32, device=dev),
               cls=torch.empty((n, topk), dtype=torch.int64, device=dev),
               loc=torch.empty((n, topk, 2), dtype=torch.float32, device=dev),
               idx=torch.empty((n, topk), dtype=torch.int32, device=dev),
               counts=torch.empty((n,), dtype=torch.int32, device=dev))
    ws = torch.empty((n, 4, cap), dtype=torch.int32, device=dev)
    check(lib.cmk_nms_topk(cand["box"].data_ptr(), cand["score"].data_ptr(), cand["cls"].data_ptr(), cand["loc"].data_ptr(),
                           cand["counts"].data_ptr(), n, cap, float(iou_thr), topk, out["box"].data_ptr(), out["score"].data_ptr(),
                           out["cls"].data_ptr(), out["loc"].data_ptr(), out["idx"].data_ptr(), out["counts"].data_ptr(),
                           ws.data_ptr(), _stream()), "cmk_nms_topk")
    return out


# ---------------------------------------------------------------------------------------------------------------
# ROI heads
# ---------------------------------------------------------------------------------------------------------------
def roi_align_ratio(feats: Sequence[View], scales: Sequence[float], boxes: torch.Tensor, counts: torch.Tensor,
                    img_area: torch.Tensor, out_size: int, sampling_ratio: int, y: torch.Tensor, min_level: int,
                    aligned: bool = True, assign_by_area: bool = False, canonical_box_size: float = 224.0, canonical_level: int = 4):
    """feats: dense NHWC levels; boxes (N,topk,4); y: (N*topk,out,out,y_cs) receives channels [0,C).  Returns levels (N*topk) int32.
    aligned False = ROIAlign v1; assign_by_area = FPN Eqn.(1) instead of CenterMask's ratio rule (pooler.py:121-152)."""
    lib = _lib.load()
    _need_gpu(feats[0].t, "roi_align")
    nl = len(feats)
    n, topk = boxes.shape[0], boxes.shape[1]
    c = feats[0].c
    ptrs = (ctypes.c_void_p * nl)()
    hs, ws_ = (ctypes.c_int * nl)(), (ctypes.c_int * nl)()
    sc = (ctypes.c_float * nl)()
    for i, f in enumerate(feats):
        assert f.co == 0 and f.cs == c, "roi_align needs dense NHWC features"
        ptrs[i] = f.t.data_ptr()
        hs[i], ws_[i] = f.t.shape[1], f.t.shape[2]
        sc[i] = float(scales[i])
    levels = torch.empty((n * topk,), dtype=torch.int32, device=boxes.device)
    check(lib.cmk_roi_align_pool(ptrs, hs, ws_, sc, nl, min_level, c, boxes.data_ptr(), counts.data_ptr(), img_area.data_ptr(),
                                 n, topk, out_size, sampling_ratio, int(bool(aligned)), int(bool(assign_by_area)), float(canonical_box_size),
                                 int(canonical_level), y.data_ptr(), y.shape[3], levels.data_ptr(), _stream()),
          "cmk_roi_align_pool")
    return levels


def spatial_attention_(x: torch.Tensor, w: torch.Tensor, counts: torch.Tensor, topk: int) -> None:
    lib = _lib.load()
    _need_gpu(x, "spatial_attention")
    r, s, _, c = x.shape
    check(lib.cmk_spatial_attention(x.data_ptr(), w.data_ptr(), counts.data_ptr(), topk, r, s, c, _stream()), "cmk_spatial_attention")


def mask_predict(dec: torch.Tensor, pw: torch.Tensor, pb: torch.Tensor, cls: torch.Tensor, counts: torch.Tensor, topk: int,
                 want_logits: bool = False):
    """dec: (R,S,S,4*C) relu(deconv); returns masks (R,1,2S,2S) [, selected-class logits (R,2S,2S)]."""
    lib = _lib.load()
    _need_gpu(dec, "mask_predict")
    r, s = dec.shape[0], dec.shape[1]
    c = dec.shape[3] // 4
    masks = torch.empty((r, 1, 2 * s, 2 * s), dtype=torch.float32, device=dec.device)
    logits = torch.empty((r, 2 * s, 2 * s), dtype=torch.float32, device=dec.device) if want_logits else None
    check(lib.cmk_mask_predict(dec.data_ptr(), pw.data_ptr(), pb.data_ptr(), cls.data_ptr(), counts.data_ptr(), topk, r, s, c,
                               masks.data_ptr(), logits.data_ptr() if want_logits else None, _stream()), "cmk_mask_predict")
    return (masks, logits) if want_logits else masks


def mask_pool_concat_(masks: torch.Tensor, y: torch.Tensor, y_co: int) -> None:
    lib = _lib.load()
    r, s = y.shape[0], y.shape[1]
    check(lib.cmk_mask_pool_concat(masks.data_ptr(), y.data_ptr(), y.shape[3], y_co, r, s, _stream()), "cmk_mask_pool_concat")


def mask_iou_score(iou: torch.Tensor, scores: torch.Tensor, cls: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    r = iou.shape[0]
    out = torch.empty((r,), dtype=torch.float32, device=iou.device)
    check(lib.cmk_mask_iou_score(iou.data_ptr(), iou.shape[1], scores.data_ptr(), cls.data_ptr(), out.data_ptr(), r, _stream()),
          "cmk_mask_iou_score")
    return out


# ---------------------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): events on the launch stream around every conv
# ---------------------------------------------------------------------------------------------------------------
PROFILE = None       # when a list, conv2d appends (kernel_key, flops, algorithmic_bytes, start_event, end_event, shape, executed_flops)


def kernel_source_hash() -> str:
    """Hash of the conv kernel sources the loaded library was built from (the tree travels with the .so): PMC summaries under
    profiles/ carry it, so a summary of an older kernel is detected instead of quoted."""
    import hashlib
    import os
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for f in ("conv_args.hpp", "wino6_common.hpp", "conv_igemm.hip", "conv_wino6.hip", "conv_wino6s.hip", "conv_pw.hip", "conv_sp3.hip"):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def executed_flops(taps: int, stride: int, tv, shapes, cin_pad: int, cout: int) -> float:
    """FLOPs the matrix pipe EXECUTES for a conv launch, tile padding included (what a roofline fraction must be priced on; the
    direct-convolution count is the algorithmic figure).  shapes: [(N, Ho, Wo)] per problem of the launch.
    Winograd forms: 16 (F(2x2)) / 36 (F(4x4)) frequency GEMMs per tile of 2x2 / 4x4 outputs, 32 tiles x 32 couts per MFMA block."""
    cd = lambda a, b: -(-a // b)
    wm, sc, wn = (tuple(tv[:3]) if tv else (0, 0, 0))
    if wm == 5:      # workgroup = 8x16 outputs x 64 couts: 256 MFMAs of 4096 FLOP per 16-channel chunk
        return float(sum(n * cd(h, 8) * cd(w, 16) for n, h, w in shapes) * cd(cout, 64) * (cin_pad // 16) * 256 * 4096)
    if wm == 6:      # workgroup = 12x40 outputs (wn 1) or two whole RoI maps (wn 2) x 32 couts: 144 MFMAs per 8-channel chunk
        # (the shared-V form, sc 64, runs two such cout tiles per workgroup; a wave group without a tile issues no MFMAs: the same count)
        wgs = sum(cd(n, 2) for n, h, w in shapes) if wn == 2 else sum(n * cd(h, 12) * cd(w, 40) for n, h, w in shapes)
        return float(wgs * cd(cout, 32) * (cin_pad // 8) * 144 * 4096)
    if wm == 11:     # direct 3x3 on split products (conv_sp3.hip), priced in fp32-equivalent FLOPs (x 3 products on the 16-bit pipe): tiles of geometry wn
        th, tw, ct = ((8, 32, 128), (4, 32, 256), (16, 16, 128), (8, 16, 256))[wn]
        if sc == 21 or (wn == 0 and cout <= 64):
            ct //= 2
        return float(sum(n * cd(h, th) * cd(w, tw) for n, h, w in shapes) * th * tw) * (cd(cout, ct) * ct) * cin_pad * 9 * 2.0
    cout_pad = _lib.load().cmk_conv_cout_pad(cout)
    if wm in (8, 9, 10, 12):  # workgroup = 64*wn pixels x 128 couts; 9 = the gather form (K = 9 taps x Cin); 10 / 12 = the split forms, priced in fp32-equivalent FLOPs
        return float(sum(cd(n * h * w, 64 * wn) for n, h, w in shapes)) * (64 * wn) * (cd(cout, 128) * 128) * cin_pad * taps * 2.0
    if wm not in (1, 2):                      # cost-model / gather / split-K variants: geometry of the smallest tile
        wm, sc = 1, (32 if taps == 1 else 16)
    if taps == 9 and tuple(tv[:1]) != (7,):
        thh, tww = (32 // sc) * 4 * wm, sc
        tiles = sum(n * cd(h, thh) * cd(w, tww) for n, h, w in shapes)
    else:
        tiles = sum(cd(n * h * w, 128 * wm) for n, h, w in shapes)
    return float(tiles) * (128 * wm) * cout_pad * taps * cin_pad * 2.0

_conv2d_plain = conv2d


def _kernel_name(taps, stride, tv, aff=False, pool=False, upres=False, cout=None) -> str:
    """The template instantiation rocprofv3 will report (minus the `void cmk::` prefix and the argument list)."""
    if not tv or tuple(tv[:3]) == (0, 0, 0):
        return "conv_igemm_kernel<{}, {}, cost-model variant>".format(taps, stride)
    if tv[0] == 5:
        return "conv_wino4r_kernel<{}>".format("true" if aff else "false")
    if tv[0] == 6:
        return "conv_wino6{}_kernel<{}, {}>".format("s" if tv[1] == 64 else "", "true" if aff else "false", 1 if tv[2] == 2 else 0)
    wm, sc, wn = tv[:3]
    sk = "true" if (len(tv) > 3 and tv[3] > 1) else "false"
    if wm == 8:
        return "conv_pw_kernel<{}, {}, false, {}, {}, 0>".format(wn, "true" if pool else "false", "true" if upres else "false", sk)
    if wm in (10, 12):
        return "conv_pw_kernel<4, {}, {}, {}, false, {}>".format("true" if pool else "false", "true" if taps == 9 else "false", "true" if upres else "false", 1 if wm == 10 else 2)
    if wm == 11:
        return "conv_sp3_kernel<{}, {}, 2, {}>".format(wn, 1 if (sc == 21 or (wn == 0 and cout is not None and cout <= 64)) else 2, "true" if aff else "false")
    if wm == 9:
        return "conv_pw_kernel<{}, false, true, false, {}, 0>".format(wn, sk)
    if wm == 7:
        return "conv_igemm_kernel<1, 1, 1, {}, 32, true>".format(wn)
    return "conv_igemm_kernel<{}, {}, {}, {}, {}, false>".format(taps, stride, wm, wn, 32 if taps == 1 else sc)


def conv2d(x, pc, y, **kw):  # noqa: F811
    if PROFILE is None:
        return _conv2d_plain(x, pc, y, **kw)
    lib = _lib.load()
    n, h, w = x.nhw
    ho, wo = y.t.shape[1], y.t.shape[2]
    taps = pc.k * pc.k
    flops = 2.0 * n * ho * wo * pc.cin * pc.cout * taps
    nbytes = 4.0 * (n * h * w * pc.cin + n * ho * wo * pc.cout + pc.cin * pc.cout * taps)
    descs = (ConvDesc * 1)()
    _fill_desc(descs[0], x, pc, y, kw.get("relu", False), kw.get("relu_upto"), kw.get("res"), kw.get("res_upsample", False), kw.get("in_relu", False))
    tv = _TUNED.get(_problem_key(descs, 1))
    key = _kernel_name(taps, pc.stride, tv, pool=kw.get("pool") is not None and FUSE_POOL, upres=bool(kw.get("res_upsample")), cout=pc.cout)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _conv2d_plain(x, pc, y, **kw)
    e1.record()
    PROFILE.append((key, flops, nbytes, e0, e1, (n, h, w, pc.cin, pc.cout, pc.k, pc.stride),
                    executed_flops(taps, pc.stride, tv, [(n, ho, wo)], pc.cin_pad, pc.cout)))


_conv2d_multi_plain = conv2d_multi


def conv2d_multi(xs, pcs, ys, **kw):  # noqa: F811
    if PROFILE is None:
        return _conv2d_multi_plain(xs, pcs, ys, **kw)
    pc = pcs[0]
    taps = pc.k * pc.k
    flops = sum(2.0 * y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * pc.cin * pc.cout * taps for y in ys)
    nbytes = sum(4.0 * (x.t.shape[0] * x.t.shape[1] * x.t.shape[2] * pc.cin + y.t.shape[0] * y.t.shape[1] * y.t.shape[2] * pc.cout) for x, y in zip(xs, ys))
    nbytes += 4.0 * pc.cin * pc.cout * taps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _conv2d_multi_plain(xs, pcs, ys, **kw)
    e1.record()
    descs = (ConvDesc * len(xs))()
    for i in range(len(xs)):
        _fill_desc(descs[i], xs[i], pcs[i], ys[i], kw.get("relu", False), kw.get("relu_upto"), None, False, False,
                   kw["in_affine"][i] if kw.get("in_affine") is not None else None)
    tv = _TUNED.get(_problem_key(descs, len(xs)))
    PROFILE.append((_kernel_name(taps, 1, tv, kw.get("in_affine") is not None), flops, nbytes, e0, e1, None,
                    executed_flops(taps, 1, tv, [tuple(y.t.shape[:3]) for y in ys], pc.cin_pad, pc.cout)))


# ---------------------------------------------------------------------------------------------------------------
# before / after the model
# ---------------------------------------------------------------------------------------------------------------
def preprocess_images(images: Sequence[torch.Tensor], mean, std, size_divisibility: int = 32, fixed_size: Optional[int] = None):
    """CHW uint8/float32 images on the GPU -> (N,3,H,W) float32 normalised, zero-padded right/bottom; returns (batch, sizes).
    fixed_size=1344 reproduces deploy_utils.single_preprocessing; otherwise the batch max rounded up to the divisibility."""
    lib = _lib.load()
    sizes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    if fixed_size:
        H = W = int(fixed_size)
    else:
        d = max(1, size_divisibility)
        H = (max(s[0] for s in sizes) + d - 1) // d * d
        W = (max(s[1] for s in sizes) + d - 1) // d * d
    dev = images[0].device
    out = torch.empty((len(images), 3, H, W), dtype=torch.float32, device=dev)
    m3, s3 = (ctypes.c_float * 3)(*[float(v) for v in mean]), (ctypes.c_float * 3)(*[float(v) for v in std])
    for i, im in enumerate(images):
        if not im.is_cuda or im.dtype not in (torch.uint8, torch.float32) or im.dim() != 3 or im.shape[0] != 3:
            raise _lib.CmkError("preprocess_images: need (3,h,w) uint8/float32 CUDA tensors")
        im = im.contiguous()
        check(lib.cmk_preprocess_chw(im.data_ptr(), int(im.dtype == torch.uint8), out[i].data_ptr(), sizes[i][0], sizes[i][1], H, W,
                                     m3, s3, _stream()), "cmk_preprocess_chw")
    return out, sizes


def paste_masks(masks: torch.Tensor, boxes: torch.Tensor, height: int, width: int, threshold: float = 0.5) -> torch.Tensor:
    """(R,S,S) float masks + (R,4) boxes -> (R,height,width) bool bitmasks."""
    lib = _lib.load()
    r, s = masks.shape[0], masks.shape[-1]
    out = torch.empty((r, height, width), dtype=torch.uint8, device=masks.device)
    if r:
        _need_gpu(masks, "paste_masks")
        check(lib.cmk_paste_masks(masks.contiguous().data_ptr(), boxes.contiguous().float().data_ptr(), r, s, height, width, float(threshold),
                                  out.data_ptr(), _stream()), "cmk_paste_masks")
    return out.bool()
