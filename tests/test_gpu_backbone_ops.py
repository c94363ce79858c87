"""-m gpu: HIP kernels of the backbone/towers through the C ABI vs the oracle's torch fp32 ops (same seeded inputs).
Tolerance: fp32 logits/features within 1e-3 ABSOLUTE (north_star) and, for single ops, also within 2e-4 * max(1, max|ref|)."""
import pytest
import torch
import torch.nn.functional as F

from centermask2_amd import ops
from centermask2_amd.ops import View

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _close(got, ref, rel=2e-4):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs().max().item()
    bound = min(1e-3, rel * max(1.0, ref.abs().max().item()))      # never looser than north_star's absolute 1e-3
    assert err <= bound, "max abs err {} > {}".format(err, bound)


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride
    (2, 20, 36, 128, 128, 3, 1),     # OSA2 layer shape class (WM2 WN4)
    (1, 17, 23, 160, 160, 3, 1),     # WN5, ragged tile edges
    (1, 25, 40, 224, 224, 3, 1),     # WN7 (stage5 map size)
    (2, 13, 20, 192, 192, 3, 1),     # WN6
    (1, 16, 16, 64, 64, 3, 1),       # WN2
    (1, 9, 11, 256, 80, 3, 1),       # cls_logits: Cout 80 -> WN3 with masked columns
    (1, 9, 11, 256, 5, 3, 1),        # bbox_pred+ctrness fused: WN1
    (2, 14, 14, 256, 256, 3, 1),     # mask head (two N tiles)
    (1, 21, 35, 64, 128, 3, 2),      # stem_3 class, stride 2, odd sizes
    (2, 14, 14, 272, 256, 3, 2),     # maskiou conv4 class
    (2, 13, 19, 768, 256, 1, 1),     # OSA concat 1x1
    (1, 25, 40, 2144, 1024, 1, 1),   # largest concat
    (1, 1, 37, 12544, 1024, 1, 1),   # maskiou_fc1 as 1x1 over 37 rows
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_igemm_matches_torch(dev, case):
    n, h, w, cin, cout, k, stride = case
    x = _rand((n, cin, h, w), 1)
    wt = _rand((cout, cin, k, k), 2, (2.0 / (cin * k * k)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(3)) + 0.5
    shift = _rand((cout,), 4, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, stride=stride, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    pc = ops.PackedConv(wt, scale, shift, dev, stride=stride)
    xv = ops.as_view(x.to(dev))
    y = ops.conv_out(xv, pc, relu=True)
    torch.cuda.synchronize()
    _close(y.nchw(), ref)


def test_conv_channel_views_residual_and_partial_relu(dev):
    """Reads a channel slice of a wider buffer, writes into a slice of a concat buffer, adds an upsampled residual,
    ReLU only on the first 4 channels (bbox_pred/ctrness epilogue)."""
    n, h, w, cin, cout = 2, 10, 12, 32, 48
    big = _rand((n, h, w, 80), 5).to(dev)
    wt = _rand((cout, cin, 3, 3), 6, 0.1)
    res = _rand((n, 5, 6, cout), 7).to(dev)
    out = torch.full((n, h, w, 96), -7.0, device=dev)
    pc = ops.PackedConv(wt, None, _rand((cout,), 8, 0.1), dev)
    ops.conv2d(View(big, 16, cin), pc, View(out, 32, cout), relu_upto=4, res=View(res), res_upsample=True, in_relu=True)
    torch.cuda.synchronize()
    xin = F.relu(big[..., 16:48].permute(0, 3, 1, 2).cpu())
    ref = F.conv2d(xin, wt, pc.shift.cpu(), padding=1) + F.interpolate(res.permute(0, 3, 1, 2).cpu(), scale_factor=2.0, mode="nearest")
    ref[:, :4] = F.relu(ref[:, :4])
    _close(out[..., 32:80].permute(0, 3, 1, 2), ref)
    assert float(out[..., :32].max()) == -7.0 and float(out[..., 80:].min()) == -7.0   # neighbours untouched


def test_conv_rejects_bad_arguments(dev, cmk_lib):
    from centermask2_amd._lib import CmkError
    x = torch.zeros((1, 4, 4, 24), device=dev)
    pc = ops.PackedConv(torch.zeros((8, 16, 3, 3)), None, None, dev)
    with pytest.raises((CmkError, AssertionError)):
        ops.conv2d(View(x, 0, 24), pc, View(torch.zeros((1, 4, 4, 8), device=dev)))


@pytest.mark.parametrize("case", [(2, 37, 53, 64, 1), (1, 40, 64, 112, 2), (3, 9, 7, 80, 1), (1, 1, 1, 4, 2), (2, 30, 31, 96, 2)])
def test_depthwise_conv3x3(dev, case):
    n, h, w, c, stride = case
    g = torch.Generator().manual_seed(5)
    buf = torch.randn((n, h, w, c + 16), generator=g).to(dev)            # read a channel slice of a wider buffer
    wt = torch.randn((c, 1, 3, 3), generator=g)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    ybuf = torch.full((n, ho, wo, c + 8), 7.0, device=dev)
    y = ops.dwconv3x3(View(buf, 8, c), ops.pack_dw_weight(wt).to(dev), View(ybuf, 4, c), stride=stride)
    torch.cuda.synchronize()
    ref = F.conv2d(buf[..., 8:8 + c].permute(0, 3, 1, 2).cpu(), wt, None, stride=stride, padding=1, groups=c).permute(0, 2, 3, 1)
    _close(ybuf[..., 4:4 + c], ref, 1e-5)
    assert float(ybuf[..., :4].min()) == 7.0 and float(ybuf[..., 4 + c:].max()) == 7.0   # neighbours untouched
    assert y.c == c


def test_stem_conv(dev):
    x = _rand((2, 3, 37, 50), 11, 40.0)
    wt = _rand((64, 3, 3, 3), 12, 0.2)
    scale = torch.rand(64, generator=torch.Generator().manual_seed(13)) * 0.05 + 0.02
    shift = _rand((64,), 14, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, stride=2, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    w27 = wt.permute(2, 3, 1, 0).reshape(27, 64).contiguous().to(dev)
    y = ops.stem_conv(x.to(dev), w27, scale.to(dev), shift.to(dev))
    torch.cuda.synchronize()
    _close(y.nchw(), ref)


@pytest.mark.parametrize("hw", [(200, 320), (25, 40), (9, 13), (3, 3), (4, 6)])
def test_maxpool_ceil(dev, hw):
    h, w = hw
    x = _rand((2, 32, h, w), 21)
    ref = F.max_pool2d(x, 3, 2, ceil_mode=True)
    y = ops.maxpool3x3s2_ceil(ops.as_view(x.to(dev)))
    torch.cuda.synchronize()
    assert tuple(y.nchw().shape) == tuple(ref.shape)
    assert torch.equal(y.nchw().cpu(), ref)          # max is exact


@pytest.mark.parametrize("c,h,w", [(256, 40, 64), (512, 20, 32), (768, 10, 16), (1024, 5, 8), (1024, 3, 3)])
def test_ese_with_identity(dev, c, h, w):
    x = _rand((2, c, h, w), 31)
    idn = _rand((2, c, h, w), 32)
    fw = _rand((c, c, 1, 1), 33, c ** -0.5)
    fb = _rand((c,), 34)
    g = F.relu6(F.conv2d(F.adaptive_avg_pool2d(x, 1), fw, fb) + 3.0) / 6.0
    ref = x * g + idn
    xv, iv = ops.as_view(x.to(dev)), ops.as_view(idn.to(dev))
    out = View(torch.empty_like(xv.t))
    ops.ese(xv, fw.reshape(c, c).contiguous().to(dev), fb.to(dev), out, identity=iv)
    torch.cuda.synchronize()
    _close(out.nchw(), ref, 1e-5)


@pytest.mark.parametrize("h,w", [(100, 160), (13, 20), (7, 10), (1, 1)])
def test_groupnorm_relu(dev, h, w):
    x = _rand((2, 256, h, w), 41, 3.0) + 1.5
    gamma = torch.rand(256, generator=torch.Generator().manual_seed(42)) + 0.5
    beta = _rand((256,), 43, 0.1)
    ref = F.relu(F.group_norm(x, 32, gamma, beta, eps=1e-5))
    t = x.permute(0, 2, 3, 1).contiguous().to(dev)
    ops.groupnorm_relu_(t, gamma.to(dev), beta.to(dev))
    torch.cuda.synchronize()
    _close(t.permute(0, 3, 1, 2), ref, 2e-5)


def test_conv_multi_level_launch(dev):
    """FCOS-style: one launch over 5 inputs of different sizes sharing the weights; per-problem scale/shift."""
    shapes = [(25, 40), (13, 20), (7, 10), (4, 5), (2, 3)]
    wt = _rand((5, 256, 3, 3), 51, 0.03)
    base = ops.PackedConv(wt, None, None, dev)
    xs, pcs, refs = [], [], []
    for l, (h, w) in enumerate(shapes):
        x = _rand((2, 256, h, w), 60 + l)
        sc = torch.tensor([1.0 + l] * 4 + [1.0])
        sh = _rand((5,), 70 + l, 0.1)
        pc = ops.PackedConv.__new__(ops.PackedConv)
        pc.__dict__.update(base.__dict__)
        pc.scale, pc.shift = sc.to(dev), sh.to(dev)
        ref = F.conv2d(x, wt, None, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        ref[:, :4] = F.relu(ref[:, :4])
        xs.append(ops.as_view(x.to(dev))); pcs.append(pc); refs.append(ref)
    ys = ops.conv_out_multi(xs, pcs, relu_upto=4)
    torch.cuda.synchronize()
    for y, ref in zip(ys, refs):
        _close(y.nchw(), ref)


def _run_variant(dev, x, wt, scale, shift, tv, stride=1, relu=True):
    """One conv through the C ABI with the tile variant written into the descriptor (cmk_conv_desc.tune_*), as the tuner does."""
    import ctypes
    from centermask2_amd import _lib
    pc = ops.PackedConv(wt, scale, shift, dev, stride=stride)
    xv = ops.as_view(x.to(dev))
    n, _, h, w = x.shape
    ho, wo = (h, w) if stride == 1 else ((h - 1) // 2 + 1, (w - 1) // 2 + 1)
    y = View(torch.full((n, ho, wo, wt.shape[0]), -5.0, device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], xv, pc, y, relu, None, None, False, False)
    ws = ops._set_variant(d, 1, tv)
    rc = _lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream())
    torch.cuda.synchronize()
    del ws
    return rc, y


DIRECT_VARIANT_SHAPES = [(2, 37, 45, 64, 128), (1, 20, 70, 32, 96), (1, 9, 33, 48, 160), (1, 17, 23, 64, 192), (1, 11, 19, 32, 224)]


@pytest.mark.parametrize("wm,sc", [(1, 16), (1, 32), (2, 16), (2, 32)])
def test_conv_direct_tile_variants(dev, wm, sc):
    """Every (WM, sub-tile, WN) instantiation of the direct 3x3 stride-1 kernel that the library admits for a shape, driven through
    the real knob (the descriptor's tune_* fields).  The start-up tuner may pick any of them, so each must be right by itself."""
    ran = 0
    for (n, h, w, cin, cout) in DIRECT_VARIANT_SHAPES:
        x = _rand((n, cin, h, w), 31)
        wt = _rand((cout, cin, 3, 3), 32, (2.0 / (cin * 9)) ** 0.5)
        scale = torch.rand(cout, generator=torch.Generator().manual_seed(33)) + 0.5
        shift = _rand((cout,), 34, 0.1)
        ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
        for wn in range(1, 8):
            rc, y = _run_variant(dev, x, wt, scale, shift, (wm, sc, wn))
            if rc != 0:
                continue                  # not on the menu for this Cout (WN must divide the padded Cout; WM 2 needs WN <= 4)
            _close(y.nchw(), ref)
            ran += 1
    assert ran >= 6, ran


def test_conv_stride2_and_1x1_variants(dev):
    for (n, h, w, cin, cout, k, stride, tvs) in [(1, 21, 35, 64, 128, 3, 2, [(1, 16, 1), (1, 16, 2), (1, 16, 4)]),
                                                 (2, 13, 19, 96, 256, 1, 1, [(1, 32, 1), (1, 32, 2), (1, 32, 4), (2, 32, 2), (2, 32, 4)])]:
        x = _rand((n, cin, h, w), 35)
        wt = _rand((cout, cin, k, k), 36, (2.0 / (cin * k * k)) ** 0.5)
        ref = F.relu(F.conv2d(x, wt, None, stride=stride, padding=k // 2))
        for tv in tvs:
            rc, y = _run_variant(dev, x, wt, None, None, tv, stride=stride)
            assert rc == 0, tv
            _close(y.nchw(), ref)


PW_CASES = [(2, 13, 19, 96, 256), (1, 25, 40, 2144, 1024), (1, 16, 16, 64, 320), (3, 9, 7, 32, 225), (1, 64, 64, 768, 256), (2, 10, 13, 160, 512)]


@pytest.mark.parametrize("mt", [4, 2])
@pytest.mark.parametrize("case", PW_CASES)
def test_conv_pointwise_kernel(dev, case, mt):
    """conv_pw (tune_wm 8): the 1x1 GEMM kernel with the weights fetched straight into registers, both workgroup heights; ragged pixel
    counts, Cout that is not a multiple of the 128-cout tile; equal to conv_igemm bit for bit (same K and accumulation order)."""
    n, h, w, cin, cout = case
    x = _rand((n, cin, h, w), 61)
    wt = _rand((cout, cin, 1, 1), 62, (2.0 / cin) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(63)) + 0.5
    shift = _rand((cout,), 64, 0.1)
    ref = F.relu(F.conv2d(x, wt, None) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, (8, 32, mt))
    assert rc == 0
    _close(y.nchw(), ref)
    rc0, y0 = _run_variant(dev, x, wt, scale, shift, (1, 32, 2))
    assert rc0 == 0 and torch.equal(y.t, y0.t)


def test_conv_pointwise_kernel_views_residual_partial_relu(dev, cmk_lib):
    """Channel views on both sides, a same-size residual, ReLU on the first channels only; then the requests the kernel refuses."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout = 2, 17, 19, 64, 288
    big = _rand((n, h, w, 160), 65).to(dev)
    wt = _rand((cout, cin, 1, 1), 66, 0.1)
    res = _rand((n, h, w, cout + 16), 67).to(dev)
    pc = ops.PackedConv(wt, torch.rand(cout, generator=torch.Generator().manual_seed(68)) + 0.5, _rand((cout,), 69, 0.1), dev)
    xin = big[..., 32:96].permute(0, 3, 1, 2).cpu()
    ref = F.conv2d(xin, wt) * pc.scale.cpu().view(1, -1, 1, 1) + pc.shift.cpu().view(1, -1, 1, 1) + res[..., 16:].permute(0, 3, 1, 2).cpu()
    ref[:, :100] = F.relu(ref[:, :100])
    for mt in (4, 2):
        out = torch.full((n, h, w, 400), -7.0, device=dev)
        d = (_lib.ConvDesc * 1)()
        ops._fill_desc(d[0], View(big, 32, cin), pc, View(out, 64, cout), False, 100, View(res, 16, cout), False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 8, 32, mt
        assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, cmk_lib.cmk_last_error()
        torch.cuda.synchronize()
        _close(out[..., 64:64 + cout].permute(0, 3, 1, 2), ref)
        assert float(out[..., :64].max()) == -7.0 and float(out[..., 64 + cout:].min()) == -7.0   # neighbours untouched
    # refused: a 3x3 conv, Cout <= 224, Cin % 32 != 0, a tile height that does not exist
    x = _rand((1, 48, 8, 8), 70)
    for wt_, tv in ((_rand((256, 48, 3, 3), 71), (8, 32, 4)), (_rand((128, 64, 1, 1), 72), (8, 32, 4)), (_rand((256, 48, 1, 1), 73), (8, 32, 4)),
                    (_rand((256, 64, 1, 1), 74), (8, 32, 3))):
        xx = x if wt_.shape[1] == 48 else _rand((1, 64, 8, 8), 75)
        rc, _ = _run_variant(dev, xx, wt_, None, None, tv)
        assert rc != 0, (tuple(wt_.shape), tv)


@pytest.mark.parametrize("case", [(2, 16, 16, 64, 256, 4), (3, 10, 13, 96, 512, 2), (3, 10, 13, 96, 512, 4), (2, 9, 15, 64, 320, 2), (5, 8, 9, 32, 256, 2)])
def test_conv_pointwise_pooled_sums_and_ese_gate(dev, cmk_lib, case):
    """The aggregation conv's fused average-pool partial sums (cmk_conv_desc.pool_ws) and the eSE gate made from them: images whose pixel
    count is not a multiple of the block (records split at the image boundary), a ragged last block, Cout that is not a multiple of 128.
    Reference: torch mean over the conv output and the reference's hsigmoid(fc(mean)) (vovnet.py:247-260)."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout, mt = case
    x = _rand((n, cin, h, w), 81)
    wt = _rand((cout, cin, 1, 1), 82, (2.0 / cin) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(83)) + 0.5
    shift = _rand((cout,), 84, 0.1)
    fc_w, fc_b = _rand((cout, cout), 85, 0.05), _rand((cout,), 86, 0.5)
    ref = F.relu(F.conv2d(x, wt, None) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref_gate = F.relu6(F.linear(ref.mean(dim=(2, 3)), fc_w, fc_b) + 3.0) / 6.0
    pc = ops.PackedConv(wt, scale, shift, dev)
    y = View(torch.full((n, h, w, cout), -5.0, device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], ops.as_view(x.to(dev)), pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 8, 32, mt
    rows = cmk_lib.cmk_conv_pool_rows(ctypes.byref(d[0]))
    assert rows == 32 * mt
    pws = torch.full((2 * (-(-(n * h * w) // rows)), cout), float("nan"), device=dev)
    d[0].pool_ws = pws.data_ptr()
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, cmk_lib.cmk_last_error()
    torch.cuda.synchronize()
    _close(y.nchw(), ref)
    assert bool(torch.isfinite(pws).all())                       # every record is written
    _close(pws.sum(0) / (n * h * w), ref.mean(dim=(0, 2, 3)), rel=1e-5)
    gate = ops.ese_gate_pooled((pws, rows), fc_w.to(dev), fc_b.to(dev), n, h * w)
    torch.cuda.synchronize()
    _close(gate, ref_gate, rel=1e-5)
    # a conv the pointwise kernel does not run: no pooled sums on offer, and asking for them anyway is an error
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 1, 32, 2
    assert cmk_lib.cmk_conv_pool_rows(ctypes.byref(d[0])) == 0
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0


@pytest.mark.parametrize("mt", [4, 2])
@pytest.mark.parametrize("case", [(1, 21, 35, 64, 128, 2), (2, 14, 14, 256, 256, 2), (1, 17, 23, 32, 128, 1), (3, 9, 11, 64, 320, 2), (1, 40, 64, 64, 128, 2)])
def test_conv_pointwise_gather_form(dev, case, mt):
    """tune_wm 9: a 3x3 conv (stride 1 | 2) as the pointwise kernel's GEMM over 9 taps x Cin/16 chunks, rows gathered per tap with
    bounds-checked loads — odd sizes (every border case of the padding), ragged pixel counts, Cout that is not a multiple of 128.  Same K
    order (tap-major) as conv_igemm's gather form: bit-identical to it."""
    n, h, w, cin, cout, stride = case
    x = _rand((n, cin, h, w), 91)
    wt = _rand((cout, cin, 3, 3), 92, (2.0 / (9 * cin)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(93)) + 0.5
    shift = _rand((cout,), 94, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, stride=stride, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, (9, 32, mt), stride=stride)
    assert rc == 0
    _close(y.nchw(), ref)
    rc0, y0 = _run_variant(dev, x, wt, scale, shift, (7, 32, 1), stride=stride)
    assert rc0 == 0 and torch.equal(y.t, y0.t)


@pytest.mark.parametrize("mt", [4, 2])
@pytest.mark.parametrize("case", [(2, 20, 36, 64, 256), (3, 10, 14, 96, 256), (1, 50, 80, 128, 288), (2, 6, 10, 32, 256)])
def test_conv_pointwise_upsampled_residual(dev, cmk_lib, case, mt):
    """The FPN lateral with the top-down add on the pointwise kernel (tune_wm 8 with res_mode 2): residual = nearest 2x upsampling of a
    map of half the size (odd heights: the coarse map has ceil(H/2) rows), ragged last tile, Cout that is not a multiple of 128; an odd
    width is refused (conv_igemm keeps it)."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout = case
    x = _rand((n, cin, h, w), 101)
    wt = _rand((cout, cin, 1, 1), 102, (2.0 / cin) ** 0.5)
    shift = _rand((cout,), 103, 0.1)
    hr, wr = (h + 1) // 2, (w + 1) // 2
    res = _rand((n, hr, wr, cout + 8), 104).to(dev)
    up = F.interpolate(res[..., 8:].permute(0, 3, 1, 2).cpu(), scale_factor=2.0, mode="nearest")[:, :, :h, :w]
    ref = F.conv2d(x, wt, shift) + up
    pc = ops.PackedConv(wt, None, shift, dev)
    y = View(torch.full((n, h, w, cout), -5.0, device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], ops.as_view(x.to(dev)), pc, y, False, None, View(res, 8, cout), True, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 8, 32, mt
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) == 0, cmk_lib.cmk_last_error()
    torch.cuda.synchronize()
    _close(y.nchw(), ref)
    # odd width: refused
    x2 = _rand((1, 32, 6, 9), 105)
    y2 = View(torch.empty((1, 6, 9, 256), device=dev))
    pc2 = ops.PackedConv(_rand((256, 32, 1, 1), 106), None, None, dev)
    res2 = View(torch.zeros((1, 3, 5, 256), device=dev))
    d2 = (_lib.ConvDesc * 1)()
    ops._fill_desc(d2[0], ops.as_view(x2.to(dev)), pc2, y2, False, None, res2, True, False)
    d2[0].tune_wm, d2[0].tune_sc, d2[0].tune_wn = 8, 32, mt
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d2[0]), ops._stream()) != 0


@pytest.mark.parametrize("case", [(1, 1, 37, 12544, 1024, 1, 1, (8, 32, 2, 8)), (1, 1, 400, 1024, 1024, 1, 1, (8, 32, 4, 2)), (6, 14, 14, 256, 256, 3, 2, (9, 32, 2, 4)),
                                  (2, 13, 20, 256, 256, 3, 2, (9, 32, 2, 8)), (2, 9, 11, 64, 320, 3, 1, (9, 32, 4, 2))])
def test_conv_pointwise_split_k(dev, case):
    """Split-K on the pointwise kernel (both the 1x1 and the gather form): raw partial sums + conv_igemm's reduce kernel."""
    n, h, w, cin, cout, k, stride, tv = case
    x = _rand((n, cin, h, w), 111)
    wt = _rand((cout, cin, k, k), 112, (2.0 / (cin * k * k)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(113)) + 0.5
    shift = _rand((cout,), 114, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, stride=stride, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, tv, stride=stride)
    assert rc == 0
    _close(y.nchw(), ref)


WINO6_CASES = [(2, 37, 45, 64, 128), (1, 16, 16, 256, 256), (1, 25, 40, 224, 224), (2, 14, 14, 256, 80), (1, 100, 160, 32, 5),
               (1, 12, 40, 128, 32), (1, 13, 41, 48, 33), (3, 5, 3, 32, 64), (1, 50, 80, 192, 192)]


@pytest.mark.parametrize("case", WINO6_CASES)
def test_conv_winograd6_variant(dev, case):
    """Fused Winograd F(4x4,3x3) kernel (tune_wm 6) against torch: exact tiles, ragged edges in both directions, maps smaller than a
    tile, Cout not a multiple of 32, Cin not a multiple of 32.  fp32 rounding differences only (tolerance as for every conv here)."""
    n, h, w, cin, cout = case
    x = _rand((n, cin, h, w), 81)
    wt = _rand((cout, cin, 3, 3), 82, (2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(83)) + 0.5
    shift = _rand((cout,), 84, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, (6, 16, 1))
    assert rc == 0
    _close(y.nchw(), ref)
    rc, y2 = _run_variant(dev, x, wt, scale, shift, (6, 16, 1))         # a second, warm launch agrees bit for bit with the first
    assert rc == 0 and torch.equal(y.t, y2.t)
    # the shared-V form (tune_sc 64: 64 couts per workgroup from one frequency image in LDS, conv_wino6s.hip) does the same arithmetic
    # in the same order: identical bits, also where the number of 32-cout tiles is odd (the idle wave group) and on ragged edges
    rc, y3 = _run_variant(dev, x, wt, scale, shift, (6, 64, 1))
    assert rc == 0
    _close(y3.nchw(), ref)
    assert torch.equal(y.t, y3.t), "conv_wino6s differs from conv_wino6"


@pytest.mark.parametrize("case", [(2, 25, 40, 768, 224, 4), (1, 50, 80, 512, 192, 2), (2, 13, 41, 64, 33, 2), (1, 12, 40, 128, 256, 8)])
def test_conv_winograd6_split_k(dev, case):
    """tune_wm 6 with split-K: 2 / 4 / 8 workgroups share the chunk loop of one (spatial tile, cout tile), raw partial sums go to the
    workspace, conv_igemm's reduce kernel sums them in a fixed order and applies the epilogue — for launches of about one round of
    workgroups with hundreds of input channels (the first conv of a stage-4 / stage-5 OSA block).  fp32 rounding differences only;
    chunk counts that do not split evenly and the shared-V form are refused."""
    n, h, w, cin, cout, sk = case
    x = _rand((n, cin, h, w), 181)
    wt = _rand((cout, cin, 3, 3), 182, (2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(183)) + 0.5
    shift = _rand((cout,), 184, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, (6, 16, 1, sk))
    assert rc == 0
    _close(y.nchw(), ref)
    rc2, y2 = _run_variant(dev, x, wt, scale, shift, (6, 16, 1, sk))      # deterministic: a second launch gives the same bits
    assert rc2 == 0 and torch.equal(y.t, y2.t)
    rc, _ = _run_variant(dev, x, wt, scale, shift, (6, 64, 1, sk))        # the shared-V form has no split-K
    assert rc != 0
    rc, _ = _run_variant(dev, _rand((1, 48, 12, 40), 185), _rand((32, 48, 3, 3), 186, 0.05), None, None, (6, 16, 1, 4))      # 6 chunks % 8 != 0
    assert rc != 0


@pytest.mark.parametrize("case", [(6, 14, 14, 256, 80), (5, 14, 14, 272, 256), (3, 16, 14, 32, 64), (2, 7, 7, 64, 32), (1, 14, 14, 64, 33)])
def test_conv_winograd6_roi_pair_geometry(dev, case):
    """tune_wm 6 / tune_wn 2: two whole RoI maps (at most 16 rows x 14 columns) per workgroup — even and odd batch, the 272-channel
    mask-IoU input, maps smaller than 14x14, Cout not a multiple of 32."""
    n, h, w, cin, cout = case
    x = _rand((n, cin, h, w), 91)
    wt = _rand((cout, cin, 3, 3), 92, (2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(93)) + 0.5
    shift = _rand((cout,), 94, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    rc, y = _run_variant(dev, x, wt, scale, shift, (6, 16, 2))
    assert rc == 0
    _close(y.nchw(), ref)
    rc, y3 = _run_variant(dev, x, wt, scale, shift, (6, 64, 2))         # the shared-V form of the same geometry: identical bits
    assert rc == 0
    _close(y3.nchw(), ref)
    assert torch.equal(y.t, y3.t), "conv_wino6s differs from conv_wino6"
    # maps wider than 14 columns are refused by this geometry
    for sc in (16, 64):
        rc, _ = _run_variant(dev, _rand((2, 32, 14, 15), 95), _rand((32, 32, 3, 3), 96, 0.05), None, None, (6, sc, 2))
        assert rc != 0


@pytest.mark.parametrize("sc", [16, 64])
def test_conv_winograd6_channel_views(dev, sc):
    """tune_wm 6 reading a channel slice of a wider buffer (an OSA concat buffer) and writing into a slice of another; partial ReLU."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout = 2, 23, 47, 64, 48
    big = _rand((n, h, w, 160), 5).to(dev)
    wt = _rand((cout, cin, 3, 3), 6, 0.06)
    out = torch.full((n, h, w, 96), -7.0, device=dev)
    pc = ops.PackedConv(wt, None, _rand((cout,), 8, 0.1), dev)
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], View(big, 32, cin), pc, View(out, 16, cout), False, 4, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 6, sc, 1
    _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "wino6 views")
    torch.cuda.synchronize()
    ref = F.conv2d(big[..., 32:96].permute(0, 3, 1, 2).cpu(), wt, pc.shift.cpu(), padding=1)
    ref[:, :4] = F.relu(ref[:, :4])
    _close(out[..., 16:64].permute(0, 3, 1, 2), ref)
    assert float(out[..., :16].max()) == -7.0 and float(out[..., 64:].min()) == -7.0   # neighbours untouched


@pytest.mark.parametrize("case", [(2, 37, 45, 64, 128), (1, 16, 16, 256, 256), (1, 25, 40, 224, 224), (2, 14, 14, 256, 80), (1, 100, 160, 32, 5)])
def test_conv_winograd_variant(dev, case):
    """Fused Winograd F(2x2,3x3) kernel (tune_wm 5) against torch; fp32 rounding differences only."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout = case
    x = _rand((n, cin, h, w), 81)
    wt = _rand((cout, cin, 3, 3), 82, (2.0 / (cin * 9)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(83)) + 0.5
    shift = _rand((cout,), 84, 0.1)
    ref = F.relu(F.conv2d(x, wt, None, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    pc = ops.PackedConv(wt, scale, shift, dev)
    xv = ops.as_view(x.to(dev))
    y = View(torch.full((n, h, w, cout), -5.0, device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], xv, pc, y, True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 5, 16, 2
    _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "wino")
    torch.cuda.synchronize()
    _close(y.nchw(), ref)


GATHER_SPLITK_CASES = [
    # N, H, W, Cin, Cout, k, stride, (wm, sc, wn, splitk), residual
    (3, 14, 14, 272, 256, 3, 2, (7, 32, 4, 1), False),    # maskiou conv4 class, gather form
    (3, 14, 14, 256, 256, 3, 2, (7, 32, 2, 2), False),    # + split-K
    (2, 13, 20, 256, 256, 3, 2, (7, 32, 1, 4), False),    # P7 class
    (2, 9, 11, 64, 96, 3, 1, (7, 32, 1, 1), True),        # gather form at stride 1 with a same-size residual
    (2, 14, 14, 272, 256, 3, 2, (0, 0, 0, 1), False),     # library default picks the gather form for tiny stride-2 maps
    (1, 1, 37, 12544, 1024, 1, 1, (1, 32, 1, 4), False),  # maskiou_fc1 as split-K GEMM
    (1, 1, 400, 1024, 80, 1, 1, (0, 0, 0, 2), True),      # split-K with the cost-model variant + residual
    (1, 16, 16, 64, 64, 3, 1, (1, 16, 2, 2), False),      # split-K on the tiled 3x3 kernel
]


@pytest.mark.parametrize("case", GATHER_SPLITK_CASES)
def test_conv_gather_form_and_split_k(dev, case):
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout, k, stride, tv, with_res = case
    x = _rand((n, cin, h, w), 61)
    wt = _rand((cout, cin, k, k), 62, (2.0 / (cin * k * k)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(63)) + 0.5
    shift = _rand((cout,), 64, 0.1)
    ref = F.conv2d(x, wt, None, stride=stride, padding=k // 2) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    res = _rand(tuple(ref.shape), 65) if with_res else None
    if with_res:
        ref = ref + res
    ref = F.relu(ref)
    pc = ops.PackedConv(wt, scale, shift, dev, stride=stride)
    pc.w_wino = None                                   # keep the default on the direct kernels
    xv = ops.as_view(x.to(dev))
    y = View(torch.full((n, ref.shape[2], ref.shape[3], cout), -5.0, device=dev))
    d = (_lib.ConvDesc * 1)()
    resv = ops.as_view(res.to(dev)) if with_res else None      # the descriptor only borrows the pointer: keep the tensor alive
    ops._fill_desc(d[0], xv, pc, y, True, None, resv, False, False)
    ws = ops._set_variant(d, 1, tv)
    _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "gather/split-K")
    torch.cuda.synchronize()
    del ws
    _close(y.nchw(), ref)


def test_conv_split_k_rejects_bad_requests(dev, cmk_lib):
    import ctypes
    from centermask2_amd import _lib
    x = ops.as_view(_rand((1, 48, 8, 8), 1).to(dev))
    pc = ops.PackedConv(_rand((32, 48, 1, 1), 2), None, None, dev)
    y = View(torch.empty((1, 8, 8, 32), device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], x, pc, y, False, None, None, False, False)
    ws = ops._set_variant(d, 1, (0, 0, 0, 2))          # 3 K chunks are not divisible by 2*2
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0 and b"split-K" in cmk_lib.cmk_last_error()
    d[0].splitk, d[0].splitk_ws = 2, None              # no workspace
    assert cmk_lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0
    del ws


@pytest.mark.parametrize("variant", [(0, 0, 0), (5, 16, 2), (6, 16, 1), (6, 64, 1), (1, 16, 1)])
def test_conv_fused_groupnorm_relu_input(dev, variant):
    """conv(relu(GroupNorm(x))) with the GN apply fused into the conv's input staging (direct kernels and Winograd form 6)."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, c, cout = 2, 25, 40, 256, 80
    x = _rand((n, c, h, w), 91, 2.0) + 0.7
    gamma = torch.rand(c, generator=torch.Generator().manual_seed(92)) + 0.5
    beta = _rand((c,), 93, 0.2)
    wt = _rand((cout, c, 3, 3), 94, 0.03)
    bias = _rand((cout,), 95, 0.1)
    ref = F.conv2d(F.relu(F.group_norm(x, 32, gamma, beta, eps=1e-5)), wt, bias, padding=1)
    xv = ops.as_view(x.to(dev))
    aff = ops.groupnorm_affine(xv.t, gamma.to(dev), beta.to(dev))
    pc = ops.PackedConv(wt, None, bias, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], xv, pc, y, False, None, None, False, False, aff)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = variant
    _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "conv+gn")
    torch.cuda.synchronize()
    _close(y.nchw(), ref)
    d[0].tune_wm = 4          # the Winograd form without the fused affine must refuse rather than ignore it
    assert _lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0


@pytest.mark.parametrize("form", [5, 6, 664])
@pytest.mark.parametrize("cout,groups", [(256, 32), (64, 32), (96, 3)])
def test_conv_with_fused_groupnorm_statistics(dev, cout, groups, form, monkeypatch):
    """The Winograd epilogues' {sum, sumsq} records -> the same per-(image, channel) affine as a pass over the output
    (664: the shared-V form of F(4x4), tune_wm 6 / tune_sc 64)."""
    monkeypatch.setattr(ops, "FORCE_VARIANT", (6, 64, 1) if form == 664 else (form, 16, 2 if form == 5 else 1))
    g = torch.Generator().manual_seed(21)
    shapes = [(2, 20, 36), (2, 9, 17), (2, 5, 3)]
    cin = 64
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    bias = torch.randn((cout,), generator=g) * 0.1
    gamma = (torch.rand((cout,), generator=g) + 0.5).to(dev)
    beta = (torch.randn((cout,), generator=g) * 0.1).to(dev)
    pc = ops.PackedConv(wt, None, bias, dev)
    xs = [torch.randn((n, cin, h, w), generator=g) for n, h, w in shapes]
    ys, aff = ops.conv_gn_multi([ops.as_view(x.to(dev)) for x in xs], [pc] * 3, gamma, beta, groups, 1e-5)
    torch.cuda.synchronize()
    for x, y, (sc, sh) in zip(xs, ys, aff):
        ref = F.conv2d(x, wt, bias, padding=1)
        _close(y.nchw(), ref)
        n, c = ref.shape[:2]
        r = ref.reshape(n, groups, -1).double()
        mean, var = r.mean(2), r.var(2, unbiased=False)
        rstd = (1.0 / torch.sqrt(var + 1e-5)).repeat_interleave(c // groups, 1).float()
        mean = mean.repeat_interleave(c // groups, 1).float()
        ref_sc = rstd * gamma.cpu()[None]
        ref_sh = beta.cpu()[None] - mean * ref_sc
        _close(sc, ref_sc, 1e-4)
        _close(sh, ref_sh, 1e-4)


@pytest.mark.parametrize("sc", [16, 64])
@pytest.mark.parametrize("with_affine", [False, True])
def test_tower_pair_launch_equals_two_launches(dev, sc, with_affine, monkeypatch):
    """ops.conv_gn_multi_pair: conv k of two towers with different weights as ONE launch of 2 x levels problems (per-problem packed weights
    in the F(4x4) kernels) — outputs and GroupNorm affines bit-identical to two conv_gn_multi launches, with and without the fused input
    affine of the previous GroupNorm."""
    monkeypatch.setattr(ops, "FORCE_VARIANT", (6, sc, 1))
    g = torch.Generator().manual_seed(77)
    shapes = [(2, 20, 36), (2, 9, 17), (2, 5, 3), (2, 3, 2), (2, 1, 1)]
    cin = cout = 64
    groups = 32
    wa = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    wb = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    pca = ops.PackedConv(wa, None, torch.randn((cout,), generator=g) * 0.1, dev)
    pcb = ops.PackedConv(wb, None, torch.randn((cout,), generator=g) * 0.1, dev)
    gn = [((torch.rand((cout,), generator=g) + 0.5).to(dev), (torch.randn((cout,), generator=g) * 0.1).to(dev)) for _ in range(2)]
    xs = [ops.as_view(torch.randn((n, cin, h, w), generator=g).to(dev)) for n, h, w in shapes]
    aff_a = aff_b = None
    if with_affine:
        aff_a = [((torch.rand((n, cin), generator=g) + 0.5).to(dev), (torch.randn((n, cin), generator=g) * 0.2).to(dev)) for n, _, _ in shapes]
        aff_b = [((torch.rand((n, cin), generator=g) + 0.5).to(dev), (torch.randn((n, cin), generator=g) * 0.2).to(dev)) for n, _, _ in shapes]
    pair = ops.conv_gn_multi_pair(xs, pca, gn[0], xs, pcb, gn[1], groups, 1e-5, in_affine_a=aff_a, in_affine_b=aff_b)
    assert pair is not None
    ya, affa = ops.conv_gn_multi(xs, [pca] * 5, gn[0][0], gn[0][1], groups, 1e-5, in_affine=aff_a)
    yb, affb = ops.conv_gn_multi(xs, [pcb] * 5, gn[1][0], gn[1][1], groups, 1e-5, in_affine=aff_b)
    torch.cuda.synchronize()
    for (ys, affs), (ys_ref, affs_ref) in zip(pair, ((ya, affa), (yb, affb))):
        for y, yr, (sc_, sh_), (scr, shr) in zip(ys, ys_ref, affs, affs_ref):
            assert torch.equal(y.t, yr.t) and torch.equal(sc_, scr) and torch.equal(sh_, shr)
    # the library refuses different weights per problem on the kernels that take one weight pointer per launch
    monkeypatch.setattr(ops, "FORCE_VARIANT", (5, 16, 2))
    assert ops.conv_gn_multi_pair(xs, pca, gn[0], xs, pcb, gn[1], groups, 1e-5) is None


@pytest.mark.parametrize("case", [(2, 13, 19, 96, 256, False), (1, 25, 40, 2144, 1024, True), (1, 16, 16, 64, 320, False), (1, 64, 64, 768, 256, True),
                                  (2, 10, 13, 160, 512, True)])
@pytest.mark.parametrize("form", [10, 12])
def test_conv_pointwise_split_bf16_form(dev, case, form, monkeypatch):
    """OPT-IN tune_wm 10 / 12: the pointwise GEMM with every fp32 product rebuilt from three bf16 pieces per operand (six products) or from two
    fp16 pieces (22-bit operands, three products), fp32 accumulation — checked against a FLOAT64 convolution at 2e-5 absolute (the fp32-MFMA
    kernel's own distance from float64 on these shapes), ragged pixel counts, cout padding, channel views via the packed conv, and the pooled
    sums of the eSE gate; refused without the split packing."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout, pool = case
    flag = "ALLOW_SPLIT_BF16" if form == 10 else "ALLOW_SPLIT_F16"
    monkeypatch.setattr(ops, flag, True)
    x = _rand((n, cin, h, w), 301).abs()                      # post-ReLU-like inputs
    wt = _rand((cout, cin, 1, 1), 302, (2.0 / cin) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(303)) + 0.5
    shift = _rand((cout,), 304, 0.1)
    ref = F.relu(F.conv2d(x.double(), wt.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    pc = ops.PackedConv(wt, scale, shift, dev)
    assert (pc.w_split if form == 10 else pc.w_splith) is not None
    xv = ops.as_view(x.to(dev))
    ys = {}
    for tv in ((form, 32, 4), (8, 32, 4)):
        y = View(torch.full((n, h, w, cout), -5.0, device=dev))
        d = (_lib.ConvDesc * 1)()
        ops._fill_desc(d[0], xv, pc, y, True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = tv
        pws = None
        if pool:
            rows = cmk_rows = _lib.load().cmk_conv_pool_rows(ctypes.byref(d[0]))
            assert rows == 128
            pws = torch.zeros((2 * (-(-(n * h * w) // rows)), cout), dtype=torch.float32, device=dev)
            d[0].pool_ws = pws.data_ptr()
        _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "pointwise " + str(tv))
        torch.cuda.synchronize()
        ys[tv[0]] = (y, pws)
    err_split = float((ys[form][0].nchw().cpu().double() - ref).abs().max())
    err_f32 = float((ys[8][0].nchw().cpu().double() - ref).abs().max())
    assert err_split <= 2e-5 and err_split <= 4.0 * err_f32 + 1e-6, (err_split, err_f32)
    if pool:                                                   # the same records (sums of the stored values) up to the values' own rounding
        a, b = ys[form][1].cpu().double(), ys[8][1].cpu().double()
        assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(b.abs().max()))
    # without the split packing the variant is refused, not silently replaced
    monkeypatch.setattr(ops, flag, False)
    pc2 = ops.PackedConv(wt, scale, shift, dev)
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], xv, pc2, View(torch.empty((n, h, w, cout), device=dev)), True, None, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = form, 32, 4
    assert _lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0


@pytest.mark.parametrize("form", [10, 12])
def test_conv_split_bf16_gather_and_upsampled_residual_forms(dev, form, monkeypatch):
    """OPT-IN tune_wm 10 / 12 (three bf16 pieces / two fp16 pieces per operand) beyond the plain 1x1 conv: a 3x3 conv (stride 2 and stride 1) in the gather form — K walks 9 taps x Cin / 16 chunks, the split
    weights packed tap-major — and a 1x1 lateral with the nearest-2x upsampled residual in the epilogue (d2 FPN top-down add); against float64."""
    import ctypes
    from centermask2_amd import _lib
    monkeypatch.setattr(ops, "ALLOW_SPLIT_BF16" if form == 10 else "ALLOW_SPLIT_F16", True)
    for (n, h, w, cin, cout, stride) in ((2, 21, 35, 64, 128, 2), (1, 17, 23, 32, 256, 1)):
        x = _rand((n, cin, h, w), 311).abs()
        wt = _rand((cout, cin, 3, 3), 312, (2.0 / (cin * 9)) ** 0.5)
        bias = _rand((cout,), 313, 0.1)
        ref = F.relu(F.conv2d(x.double(), wt.double(), bias.double(), stride=stride, padding=1))
        pc = ops.PackedConv(wt, None, bias, dev, stride=stride)
        assert (pc.w_split if form == 10 else pc.w_splith) is not None
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        y = View(torch.full((n, ho, wo, cout), -5.0, device=dev))
        d = (_lib.ConvDesc * 1)()
        ops._fill_desc(d[0], ops.as_view(x.to(dev)), pc, y, True, None, None, False, False)
        d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = form, 32, 4
        _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "split gather form")
        torch.cuda.synchronize()
        assert float((y.nchw().cpu().double() - ref).abs().max()) <= 2e-5
    # lateral + nearest-2x upsampled residual
    n, h, w, cin, cout = 2, 12, 20, 512, 256
    x = _rand((n, cin, h, w), 321).abs()
    wt = _rand((cout, cin, 1, 1), 322, (2.0 / cin) ** 0.5)
    bias = _rand((cout,), 323, 0.1)
    coarse = _rand((n, cout, h // 2, w // 2), 324)
    ref = F.conv2d(x.double(), wt.double(), bias.double()) + F.interpolate(coarse.double(), scale_factor=2, mode="nearest")
    pc = ops.PackedConv(wt, None, bias, dev)
    y = View(torch.empty((n, h, w, cout), device=dev))
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], ops.as_view(x.to(dev)), pc, y, False, None, ops.as_view(coarse.to(dev)), True, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = form, 32, 4
    _lib.check(_lib.load().cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "split lateral + upsampled residual")
    torch.cuda.synchronize()
    assert float((y.nchw().cpu().double() - ref).abs().max()) <= 2e-5


def test_groupnorm_affine_multi_level(dev):
    shapes = [(100, 160), (13, 20), (7, 10), (1, 2)]
    gamma = torch.rand(256, generator=torch.Generator().manual_seed(42)) + 0.5
    beta = _rand((256,), 43, 0.1)
    xs = [_rand((2, 256, h, w), 100 + i, 3.0) + 1.5 for i, (h, w) in enumerate(shapes)]
    outs = ops.groupnorm_affine_multi([x.permute(0, 2, 3, 1).contiguous().to(dev) for x in xs], gamma.to(dev), beta.to(dev))
    torch.cuda.synchronize()
    for x, (sc, sh) in zip(xs, outs):
        ref = F.group_norm(x, 32, gamma, beta, eps=1e-5)
        got = x * sc.cpu()[:, :, None, None] + sh.cpu()[:, :, None, None]
        _close(got, ref, 2e-5)


def test_maxpool_with_folded_ese_gate(dev):
    x = _rand((2, 64, 21, 30), 111)
    gate = torch.rand((2, 64), generator=torch.Generator().manual_seed(112))
    ref = F.max_pool2d(x * gate[:, :, None, None], 3, 2, ceil_mode=True)
    y = ops.maxpool3x3s2_ceil(ops.as_view(x.to(dev)), gate=gate.to(dev))
    torch.cuda.synchronize()
    _close(y.nchw(), ref, 1e-6)


SP3_CASES = [  # (n, h, w, cin, cout, geo, pieces): ragged maps, cout padding, every tile geometry; three pieces are refused
    (2, 37, 53, 64, 128, 0, 2), (1, 21, 70, 48, 80, 1, 2), (3, 14, 14, 256, 256, 2, 2), (2, 50, 80, 128, 256, 3, 2), (2, 13, 20, 32, 33, 0, 2),
    (1, 9, 33, 272, 160, 3, 2), (2, 8, 32, 64, 128, 0, 3), (1, 19, 45, 96, 224, 1, 2), (2, 21, 37, 32, 64, 0, 2), (1, 12, 40, 64, 48, 0, 2),      # <= 64 couts: one cout tile per wave
    (1, 19, 45, 96, 160, 0, 21), (2, 9, 33, 48, 192, 1, 21), (2, 14, 14, 64, 256, 2, 21), (1, 25, 40, 224, 224, 3, 21), (1, 9, 9, 32, 64, 0, 3)]     # one cout tile per wave on request (tune_sc 21); other values refused


@pytest.mark.parametrize("case", SP3_CASES)
def test_conv_direct_split_f16_form(dev, case, monkeypatch):
    """OPT-IN tune_wm 11 (conv_sp3.hip): 3x3 stride-1 conv as a direct implicit GEMM on fp16 pieces of the fp32 operands — two pieces (22-bit
    operands), three products, fp32 accumulation: within 1e-5 of a FLOAT64 conv on unit-variance outputs (an fp32 accumulation's distance; the
    fp32 F(4x4) Winograd form sits at 1e-4); inputs spanning 1e-4 .. 1e3 in magnitude (the residual scaling); output channel views, partial
    ReLU, every geometry; refused without the packing or with another piece count."""
    import ctypes
    from centermask2_amd import _lib
    n, h, w, cin, cout, geo, pieces = case
    monkeypatch.setattr(ops, "ALLOW_SPLIT_BF16", True)
    monkeypatch.setattr(ops, "ALLOW_SPLIT_F16", True)
    x = _rand((n, cin, h, w), 401)
    mags = (1.0, 1e-4, 1e3)                                   # images of very different magnitude: fp16's exponent range is met by scaling, not by luck
    for i in range(n):
        x[i] *= mags[i % 3]
    wt = _rand((cout, cin, 3, 3), 402, (1.0 / (9 * cin)) ** 0.5)
    scale = torch.rand(cout, generator=torch.Generator().manual_seed(403)) + 0.5
    shift = torch.zeros(cout)                                 # (a bias would hide the small image's error behind its own magnitude)
    pc = ops.PackedConv(wt, scale, shift, dev)
    assert pc.w_splith is not None
    relu_upto = cout // 2
    ref = F.conv2d(x.double(), wt.double(), padding=1) * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref[:, :relu_upto] = ref[:, :relu_upto].relu()
    xv = ops.as_view(x.to(dev))
    big = torch.full((n, h, w, cout + 24), float("nan"), device=dev)         # the output is a channel slice of a wider buffer (the OSA concat)
    y = View(big, 16, cout)
    d = (_lib.ConvDesc * 1)()
    ops._fill_desc(d[0], xv, pc, y, False, relu_upto, None, False, False)
    d[0].tune_wm, d[0].tune_sc, d[0].tune_wn = 11, pieces, geo
    lib = _lib.load()
    if pieces not in (2, 21):
        assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0
        return
    _lib.check(lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()), "conv_sp3")
    torch.cuda.synchronize()
    got = big[..., 16:16 + cout].permute(0, 3, 1, 2).double().cpu()
    assert bool(torch.isnan(big[..., :16]).all()) and bool(torch.isnan(big[..., 16 + cout:]).all())       # nothing written outside the view
    for i in range(n):                                        # per image, relative to that image's own output magnitude
        err, mag = float((got[i] - ref[i]).abs().max()), float(ref[i].abs().max())
        assert err <= 1e-5 * mag, (i, err, mag)
    d[0].w_splith = None
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d[0]), ops._stream()) != 0                                    # no packing: refused, not emulated


@pytest.mark.parametrize("geo", [0, 1, 2, 3])
@pytest.mark.parametrize("with_affine", [False, True])
def test_conv_direct_split_tower_launches(dev, geo, with_affine, monkeypatch):
    """The FCOS tower launches on the opt-in direct split form: five levels in one launch, the fused GroupNorm + ReLU of the previous layer
    applied while staging (padding stays zero), the {sum, sumsq} records of the next GroupNorm from the epilogue, and two towers with
    different weights as ONE launch — against float64 convolutions / statistics."""
    monkeypatch.setattr(ops, "ALLOW_SPLIT_F16", True)
    monkeypatch.setattr(ops, "FORCE_VARIANT", (11, 2, geo))
    g = torch.Generator().manual_seed(88)
    shapes = [(2, 20, 36), (2, 9, 17), (2, 5, 3), (2, 3, 2), (2, 1, 1)]
    cin = cout = 64
    groups = 32
    ws_ = [torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5 for _ in range(2)]
    bs_ = [torch.randn((cout,), generator=g) * 0.1 for _ in range(2)]
    pcs = [ops.PackedConv(w_, None, b_, dev) for w_, b_ in zip(ws_, bs_)]
    gn = [((torch.rand((cout,), generator=g) + 0.5).to(dev), (torch.randn((cout,), generator=g) * 0.1).to(dev)) for _ in range(2)]
    xs_cpu = [torch.randn((n, cin, h, w), generator=g) for n, h, w in shapes]
    xs = [ops.as_view(x.to(dev)) for x in xs_cpu]
    affs = [None, None]
    if with_affine:
        affs = [[((torch.rand((n, cin), generator=g) + 0.5).to(dev), (torch.randn((n, cin), generator=g) * 0.2).to(dev)) for n, _, _ in shapes] for _ in range(2)]
    pair = ops.conv_gn_multi_pair(xs, pcs[0], gn[0], xs, pcs[1], gn[1], groups, 1e-5, in_affine_a=affs[0], in_affine_b=affs[1])
    assert pair is not None
    single = ops.conv_gn_multi(xs, [pcs[0]] * 5, gn[0][0], gn[0][1], groups, 1e-5, in_affine=affs[0])
    torch.cuda.synchronize()
    for t, (ys, aff) in enumerate(pair):
        for l, (x, y, (sc, sh)) in enumerate(zip(xs_cpu, ys, aff)):
            xin = x.double()
            if with_affine:
                a_sc, a_sh = affs[t][l]
                xin = (xin * a_sc.cpu().double()[:, :, None, None] + a_sh.cpu().double()[:, :, None, None]).relu()
            ref = F.conv2d(xin, ws_[t].double(), bs_[t].double(), padding=1)
            assert float((y.nchw().double().cpu() - ref).abs().max()) <= 3e-4
            n, c = ref.shape[:2]
            r = ref.reshape(n, groups, -1)
            mean, var = r.mean(2), r.var(2, unbiased=False)
            rstd = (1.0 / torch.sqrt(var + 1e-5)).repeat_interleave(c // groups, 1)
            ref_sc = rstd * gn[t][0].cpu().double()[None]
            ref_sh = gn[t][1].cpu().double()[None] - mean.repeat_interleave(c // groups, 1) * ref_sc
            assert float((sc.cpu().double() - ref_sc).abs().max()) <= 2e-3 * float(ref_sc.abs().max())
            assert float((sh.cpu().double() - ref_sh).abs().max()) <= 2e-3 * max(1.0, float(ref_sh.abs().max()))
    for y, yr, (sc_, sh_), (scr, shr) in zip(pair[0][0], single[0], pair[0][1], single[1]):      # the pair launch = the single launch, bit for bit
        assert torch.equal(y.t, yr.t) and torch.equal(sc_, scr) and torch.equal(sh_, shr)
