"""-m gpu: pre/post-processing kernels vs the oracle.  d2's ROIMasks is absent from the reference
tree, so the mask paste is "parity unpinned" beyond the oracle's restatement of the published algorithm."""
import pytest
import torch

from centermask2_amd import ops, postprocess
from centermask2_amd.structures import Boxes, Instances
from oracle import centermask_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.uint8, torch.float32])
def test_preprocess_matches_reference_padding(dev, dtype):
    g = torch.Generator().manual_seed(1)
    imgs = [torch.randint(0, 256, (3, 37, 53), generator=g).to(dtype), torch.randint(0, 256, (3, 64, 21), generator=g).to(dtype)]
    mean, std = (103.53, 116.28, 123.675), (1.0, 57.0, 2.0)
    out, sizes = ops.preprocess_images([i.to(dev) for i in imgs], mean, std, fixed_size=96)
    torch.cuda.synchronize()
    assert sizes == [(37, 53), (64, 21)] and tuple(out.shape) == (2, 3, 96, 96)
    for i, im in enumerate(imgs):
        ref = O.preprocess(im, mean, std, fixed_size=96)
        assert torch.allclose(out[i].cpu(), ref, rtol=0, atol=1e-5)
    out2, _ = ops.preprocess_images([i.to(dev) for i in imgs], mean, std, size_divisibility=32)
    assert tuple(out2.shape) == (2, 3, 64, 64)


def test_detector_postprocess_and_mask_paste(dev):
    g = torch.Generator().manual_seed(2)
    n, h, w = 9, 427, 640                      # a COCO-sized original image; model ran at 800x1199
    boxes = torch.rand((n, 4), generator=g) * torch.tensor([900.0, 600.0, 900.0, 600.0])
    boxes[:, 2:] = boxes[:, :2] + torch.rand((n, 2), generator=g) * 500 + 1
    boxes[0] = torch.tensor([-50.0, -20.0, 30.0, 40.0])             # partly outside
    boxes[1] = torch.tensor([1300.0, 900.0, 1400.0, 950.0])         # fully outside after rescale -> dropped
    masks = torch.rand((n, 1, 28, 28), generator=g)
    res = dict(boxes=boxes, scores=torch.rand(n, generator=g), classes=torch.randint(0, 80, (n,), generator=g), locations=boxes[:, :2],
               pred_masks=masks, mask_scores=torch.rand(n, generator=g))
    ref = O.detector_postprocess(res, h, w)
    inst = Instances((800, 1199), pred_boxes=Boxes(boxes.to(dev)), scores=res["scores"].to(dev), pred_classes=res["classes"].to(dev),
                     locations=res["locations"].to(dev), pred_masks=masks.to(dev), mask_scores=res["mask_scores"].to(dev))
    out = postprocess.detector_postprocess(inst, h, w)
    torch.cuda.synchronize()
    assert len(out) == ref["boxes"].shape[0] < n
    assert torch.allclose(out.pred_boxes.tensor.cpu(), ref["boxes"], rtol=0, atol=1e-4)
    assert torch.equal(out.pred_classes.cpu(), ref["classes"])
    got, want = out.pred_masks.cpu(), ref["pred_masks"]
    assert got.shape == want.shape and got.dtype == torch.bool
    mismatch = (got != want).float().mean().item()
    assert mismatch < 1e-5, mismatch          # pixels whose interpolated value sits within rounding of the 0.5 threshold
    assert postprocess.resize_scale(427, 640) == pytest.approx(800 / 427)


def test_wire_bin_sextuple_from_the_hip_path(dev, tmp_path):
    """SURVEY 8(f)3 on the GPU path: the model's 6-tuple (modified_class.py:28-40 order) written as the deployment flow's six .bin files,
    against the files the REFERENCE's own modules and tuple helper produced for the same image (tests/golden/bin, make_golden_bin.py)."""
    import os
    from centermask2_amd import synthetic as S, wire
    from .helpers import GOLDEN, build_gpu_model, close, golden
    exp = golden("bin_expected")
    model, _ = build_gpu_model()
    x = S.make_synthetic_images(1, 256, 320, seed0=4321).to(dev)
    t6 = model.forward_tensor(x, hw=[(256, 320)])
    torch.cuda.synchronize()
    paths = wire.to_bin(tuple(t[:5] for t in t6), str(tmp_path / "000000000139"))
    assert [os.path.getsize(p) for p in paths] == [os.path.getsize(os.path.join(GOLDEN, "bin", "000000000139_{}.bin".format(i + 1))) for i in range(6)]
    got = wire.from_bin(str(tmp_path / "000000000139"))
    names = ("locations", "mask_scores", "pred_boxes", "pred_classes", "pred_masks", "scores")
    assert torch.equal(got[0], exp["locations"]) and torch.equal(got[3], exp["pred_classes"]) and got[3].dtype == torch.int64
    for i, tol in ((1, 1e-3), (2, 2e-5), (4, 1e-3), (5, 1e-4)):
        close(got[i], exp[names[i]], tol, names[i])


def test_pack_records_kernel_matches_layout(dev):
    """cmk_pack_records (the all-gather send record, one launch) against the torch assembly of the same layout, and the round trip."""
    from centermask2_amd.dist import pack_records, record_width, unpack_records
    g = torch.Generator().manual_seed(9)
    n, k = 3, 50
    out = dict(box=torch.rand((n, k, 4), generator=g) * 900, score=torch.rand((n, k), generator=g), mask_scores=torch.rand((n, k), generator=g),
               loc=torch.rand((n, k, 2), generator=g) * 1280, cls=torch.randint(0, 80, (n, k), generator=g), pred_masks=torch.rand((n, k, 1, 28, 28), generator=g),
               counts=torch.tensor([50, 0, 17], dtype=torch.int32))
    want = pack_records(out)                                      # CPU tensors: torch assembly
    got = pack_records({kk: v.to(dev) for kk, v in out.items()})  # GPU tensors: the kernel
    torch.cuda.synchronize()
    assert tuple(got.shape) == (n, record_width(k)) and torch.equal(got.cpu(), want)
    back = unpack_records(got.cpu(), k)
    assert torch.equal(back["cls"], out["cls"]) and torch.equal(back["counts"], out["counts"]) and torch.equal(back["pred_masks"], out["pred_masks"])
