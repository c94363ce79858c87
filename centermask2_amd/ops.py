"""Thin host wrappers: torch tensors (device memory, current stream) -> C ABI calls of libcmk_hip.so.

PyTorch is plumbing here (allocation, streams); the arithmetic happens in the HIP kernels.  Activations are
NHWC float32; `View` is (NHWC tensor, channel offset, channels) so producers write straight into slices of
an OSA concat buffer.  Every function raises if the tensors are not on a GPU — there is no CPU fallback.
"""
import ctypes
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
        self.scale = (torch.ones(self.cout) if scale is None else scale.detach().float().cpu()).contiguous().to(device)
        self.shift = (torch.zeros(self.cout) if shift is None else shift.detach().float().cpu()).contiguous().to(device)


def fold_frozen_bn(weight, bias, running_mean, running_var, eps=1e-5):
    """FrozenBN as a per-channel epilogue: y = conv * scale + shift (d2 FrozenBatchNorm2d, eps 1e-5)."""
    scale = weight.double() / torch.sqrt(running_var.double() + eps)
    shift = bias.double() - running_mean.double() * scale
    return scale.float(), shift.float()


def conv2d(x: View, pc: PackedConv, y: View, relu: bool = False, relu_upto: Optional[int] = None,
           res: Optional[View] = None, res_upsample: bool = False, in_relu: bool = False) -> None:
    lib = _lib.load()
    _need_gpu(x.t, "conv2d")
    n, h, w = x.nhw
    d = ConvDesc()
    d.x, d.x_cs, d.x_co = x.t.data_ptr(), x.cs, x.co
    d.w = pc.w.data_ptr()
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
    check(lib.cmk_conv2d_nhwc(ctypes.byref(d), _stream()), "cmk_conv2d_nhwc")


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


def maxpool3x3s2_ceil(x: View, y: Optional[View] = None) -> View:
    lib = _lib.load()
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
    check(lib.cmk_maxpool3x3s2_ceil_nhwc(x.t.data_ptr(), x.cs, x.co, y.t.data_ptr(), y.cs, y.co, n, h, w, x.c, _stream()),
          "cmk_maxpool3x3s2_ceil_nhwc")
    return y


def ese(x: View, fc_w: torch.Tensor, fc_b: torch.Tensor, y: View, identity: Optional[View] = None) -> None:
    """y = x * hsigmoid(fc(mean_HW(x))) (+ identity)   (vovnet.py:255-260, :329-330)."""
    lib = _lib.load()
    n, h, w = x.nhw
    hw, c = h * w, x.c
    chunks = max(1, min(64, hw // 256))
    ws = torch.empty((n, chunks, c), dtype=torch.float32, device=x.t.device)
    gate = torch.empty((n, c), dtype=torch.float32, device=x.t.device)
    check(lib.cmk_ese_gate(x.t.data_ptr(), x.cs, x.co, fc_w.data_ptr(), fc_b.data_ptr(), gate.data_ptr(), ws.data_ptr(), chunks,
                           n, hw, c, _stream()), "cmk_ese_gate")
    idp, idcs, idco = (identity.t.data_ptr(), identity.cs, identity.co) if identity is not None else (None, 0, 0)
    check(lib.cmk_ese_scale(x.t.data_ptr(), x.cs, x.co, gate.data_ptr(), idp, idcs, idco, y.t.data_ptr(), y.cs, y.co,
                            n, hw, c, _stream()), "cmk_ese_scale")


def groupnorm_relu_(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32, eps: float = 1e-5) -> None:
    """In place on a dense NHWC tensor (fcos.py:182-186)."""
    lib = _lib.load()
    _need_gpu(x, "groupnorm_relu_")
    n, h, w, c = x.shape
    hw = h * w
    chunks = max(1, min(128, hw // 128))
    ws = torch.empty((n, groups, chunks, 2), dtype=torch.float64, device=x.device)
    check(lib.cmk_groupnorm_relu_nhwc(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), chunks, n, hw, c, groups,
                                      eps, _stream()), "cmk_groupnorm_relu_nhwc")
