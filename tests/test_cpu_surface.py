"""not gpu: host logic — config, registries, structures, plugin construction, C-ABI export list, sharding + gloo gather."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from centermask2_amd import _lib
    header = open(os.path.join(ROOT, "include", "cmk.h")).read()
    declared = set(re.findall(r"\b(cmk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                      # loads on CPU; no compute call is made here
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.cmk_version() == 5 and lib.cmk_arch() == b"gfx950"
    assert lib.cmk_conv_cout_pad(80) == 96 and lib.cmk_conv_cout_pad(160) == 160 and lib.cmk_conv_cout_pad(1024) == 1024
    assert lib.cmk_conv_packed_floats(256, 257, 3) == 9 * 17 * 256 * 16


def test_c_abi_argument_validation_without_gpu():
    """Bad descriptors are rejected before any launch, with a message (error behaviour of the boundary)."""
    from centermask2_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d), None) == -1 and b"null" in lib.cmk_last_error()
    buf = (ctypes.c_float * 64)()
    d.x = d.w = d.scale = d.shift = d.y = ctypes.addressof(buf)
    d.N, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride, d.x_cs, d.y_cs = 1, 2, 2, 24, 8, 3, 1, 24, 8
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d), None) == -1 and b"multiple of 16" in lib.cmk_last_error()
    d.Cin, d.x_cs, d.ksize = 16, 16, 5
    assert lib.cmk_conv2d_nhwc(ctypes.byref(d), None) == -1 and b"ksize" in lib.cmk_last_error()
    assert lib.cmk_nms_topk(None, None, None, None, None, 1, 1, 0.6, 50, None, None, None, None, None, None, None, None) == -1


def test_config_matches_reference_recipe():
    from centermask2_amd.config import config_path, get_cfg
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.FCOS.NMS_TH", 0.5])
    cfg.freeze()
    m = cfg.MODEL
    assert m.BACKBONE.NAME == "build_fcos_vovnet_fpn_backbone" and m.PROPOSAL_GENERATOR.NAME == "FCOS" and m.ROI_HEADS.NAME == "CenterROIHeads"
    assert m.ROI_MASK_HEAD.NAME == "SpatialAttentionMaskHead" and m.ROI_MASKIOU_HEAD.NAME == "MaskIoUHead"
    assert m.FCOS.POST_NMS_TOPK_TEST == 50 and m.FCOS.INFERENCE_TH_TEST == 0.05 and m.FCOS.NMS_TH == 0.5
    assert m.ROI_MASK_HEAD.ASSIGN_CRITERION == "ratio" and m.ROI_MASK_HEAD.POOLER_RESOLUTION == 14 and m.MASKIOU_ON and m.MASK_ON
    assert m.VOVNET.CONV_BODY == "V-39-eSE" and m.VOVNET.OUT_FEATURES == ["stage3", "stage4", "stage5"]
    with pytest.raises(AttributeError):
        cfg.MODEL.DEVICE = "cuda"
    with pytest.raises(KeyError):
        get_cfg().merge_from_list(["MODEL.NOPE", 1])
    c2 = cfg.clone()
    c2.defrost()
    c2.MODEL.DEVICE = "cuda"
    assert cfg.MODEL.DEVICE == "cpu"


def test_plugins_register_and_build_with_reference_keys():
    from centermask2_amd import synthetic as S
    from centermask2_amd.config import config_path, get_cfg
    from centermask2_amd.modeling import build_model
    from centermask2_amd.registry import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY,
                                          ROI_MASK_HEAD_REGISTRY, ROI_MASKIOU_HEAD_REGISTRY)
    for reg, name in ((BACKBONE_REGISTRY, "build_fcos_vovnet_fpn_backbone"), (BACKBONE_REGISTRY, "build_vovnet_backbone"),
                      (PROPOSAL_GENERATOR_REGISTRY, "FCOS"), (ROI_HEADS_REGISTRY, "CenterROIHeads"), (META_ARCH_REGISTRY, "GeneralizedRCNN"),
                      (ROI_MASK_HEAD_REGISTRY, "SpatialAttentionMaskHead"), (ROI_MASKIOU_HEAD_REGISTRY, "MaskIoUHead")):
        assert name in reg and reg.get(name) is not None
    for body, nkeys in (("V-39-eSE", 293), ("V-99-eSE", 613)):
        cfg = get_cfg()
        cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.VOVNET.CONV_BODY", body])
        model = build_model(cfg)
        sd = model.state_dict()
        shapes = S.model_param_shapes(body)
        assert len(sd) == nkeys and set(sd) == set(shapes)
        assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in sd)
    shp = model.backbone.output_shape()
    assert list(shp) == ["p3", "p4", "p5", "p6", "p7"] and shp["p6"].stride == 64 and model.backbone.size_divisibility == 32
    # build_vovnet_fpn_backbone (vovnet.py:504-524): FPN + LastLevelMaxPool; the parameter names the reference's builder produced
    from centermask2_amd.structures import ShapeSpec
    from .helpers import golden
    assert "build_vovnet_fpn_backbone" in BACKBONE_REGISTRY
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    bb = BACKBONE_REGISTRY.get("build_vovnet_fpn_backbone")(cfg, ShapeSpec(channels=3))
    g = golden("vovnet_fpn_maxpool")
    assert set(bb.state_dict().keys()) == set(g["keys"])
    shp = bb.output_shape()
    assert list(shp) == ["p3", "p4", "p5", "p6"] and [shp[k].stride for k in shp] == g["strides"].tolist()


def test_product_path_fails_loudly_without_gpu():
    from centermask2_amd._lib import CmkError
    from centermask2_amd.config import config_path, get_cfg
    from centermask2_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    model = build_model(cfg).eval()
    with pytest.raises(CmkError):
        model.backbone(torch.zeros(1, 3, 64, 64))
    with pytest.raises(CmkError):
        model.proposal_generator.fcos_head([torch.zeros(1, 256, 8, 8)])


def test_product_never_imports_oracle():
    """The product package may mention the oracle in comments, but never imports, loads or links it."""
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle)|liboracle|oracle_ops\.so|oracle\._|CDLL\([^)]*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "centermask2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".sh")):
                assert not pat.search(open(os.path.join(dirpath, f)).read()), f


def test_shipped_variant_tables_name_existing_kernels(tmp_path):
    """The measured conv-variant tables bench.py loads must only name variants the library still has; stale entries are dropped."""
    import glob, json, os
    from centermask2_amd import ops
    tables = glob.glob(os.path.join(os.path.dirname(ops.__file__), "tuned", "*.json"))
    assert tables
    for path in tables:
        table = json.load(open(path))
        optin = "_split" in os.path.basename(path)           # measured tables of the opt-in split forms: on the menu only for a caller that opted in
        flags = (ops.ALLOW_SPLIT_BF16, ops.ALLOW_SPLIT_F16)
        try:
            if optin:
                assert not all(ops._variant_on_menu(v) for v in table.values()), path      # ... and refused entry by entry otherwise
                ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
            assert table and all(ops._variant_on_menu(v) for v in table.values()), path
        finally:
            ops.ALLOW_SPLIT_BF16, ops.ALLOW_SPLIT_F16 = flags
        assert all(ops._key_to_str(ops._str_to_key(k)) == k for k in table)
    stale = tmp_path / "stale.json"
    key = next(iter(json.load(open(sorted(tables)[0]))))
    stale.write_text(json.dumps({key: [4, 16, 2]}))          # a Winograd form that was removed
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        assert ops.load_tuned(str(stale)) == 0 and not ops._TUNED
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def test_bin_sextuple_fixture_from_the_reference(tmp_path):
    """tests/golden/bin/000000000139_{1..6}.bin: written from the tuple of the reference's own single_flatten_to_tuple and read back by
    its own reader steps (tests/golden/make_golden_bin.py).  wire.from_bin must parse them to the same tensors, wire.to_bin must write
    the same bytes."""
    from centermask2_amd import wire
    from .helpers import GOLDEN, golden
    exp = golden("bin_expected")
    prefix = os.path.join(GOLDEN, "bin", "000000000139")
    got = wire.from_bin(prefix)
    names = ("locations", "mask_scores", "pred_boxes", "pred_classes", "pred_masks", "scores")
    for k, t in zip(names, got):
        assert t.dtype == exp[k].dtype and torch.equal(t, exp[k]), k
    paths = wire.to_bin(tuple(exp[k] for k in names), str(tmp_path / "000000000139"))
    for i, pth in enumerate(paths):
        assert open(pth, "rb").read() == open("{}_{}.bin".format(prefix, i + 1), "rb").read(), pth


def test_structures():
    from centermask2_amd.structures import Boxes, FakeImageList, ImageList, Instances
    b = Boxes(torch.tensor([[0., 0., 10., 20.], [5., 5., 5., 9.]]))
    assert b.area().tolist() == [200.0, 0.0] and b.nonempty().tolist() == [True, False]
    i = Instances((4, 6), pred_boxes=b, scores=torch.tensor([0.3, 0.7]))
    j = i[torch.tensor([1])]
    assert len(j) == 1 and j.scores.item() == pytest.approx(0.7) and j.image_size == (4, 6)
    c = Instances.cat([i, i])
    assert len(c) == 4 and c.pred_boxes.tensor.shape == (4, 4)
    with pytest.raises(AssertionError):
        i.bad = torch.zeros(3)
    il = ImageList.from_tensors([torch.zeros(3, 5, 7), torch.zeros(3, 6, 4)], 32)
    assert tuple(il.tensor.shape) == (2, 3, 32, 32) and il.image_sizes == [(5, 7), (6, 4)]
    assert len(FakeImageList(torch.zeros(2, 3, 8, 8))) == 2 and FakeImageList(torch.zeros(1, 3, 8, 8)).image_sizes == [(1344, 1344)]


def test_shard_ranges_cover_batch():
    from centermask2_amd.dist import shard_range
    for n in (1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_record_roundtrip():
    from centermask2_amd.dist import pack_records, record_width, unpack_records
    g = torch.Generator().manual_seed(0)
    n, k = 3, 50
    out = dict(box=torch.rand((n, k, 4), generator=g), score=torch.rand((n, k), generator=g), mask_scores=torch.rand((n, k), generator=g),
               loc=torch.rand((n, k, 2), generator=g), cls=torch.randint(0, 80, (n, k), generator=g),
               pred_masks=torch.rand((n, k, 1, 28, 28), generator=g), counts=torch.tensor([50, 0, 17], dtype=torch.int32))
    rec = pack_records(out)
    assert rec.shape == (n, record_width(k))
    back = unpack_records(rec, k)
    for key in out:
        assert torch.equal(back[key], out[key]), key


_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from centermask2_amd.dist import all_gather_records, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n_images, width = 5, 11
full = torch.arange(n_images * width, dtype=torch.float32).reshape(n_images, width)
lo, hi = shard_range(n_images, rank, world)
per_rank = -(-n_images // world)
got = all_gather_records(full[lo:hi].clone(), per_rank, n_images)
assert torch.equal(got, full), (rank, got)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_all_gather_world_size_2_gloo(tmp_path):
    """The N>1 path (uneven shards, zero-padded all-gather) with two CPU processes over gloo."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


_BENCH_GLOO_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import bench
from centermask2_amd.dist import pack_records, record_width, unpack_records
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cpu")
B, K = 3, 4
g = torch.Generator().manual_seed(100 + rank)
def fake_out(r):
    gg = torch.Generator().manual_seed(100 + r)
    return dict(box=torch.rand((B, K, 4), generator=gg), score=torch.rand((B, K), generator=gg), mask_scores=torch.rand((B, K), generator=gg),
                loc=torch.rand((B, K, 2), generator=gg), cls=torch.randint(0, 80, (B, K), generator=gg), pred_masks=torch.rand((B, K, 1, 28, 28), generator=gg),
                counts=torch.tensor([K, 1 + r, 0], dtype=torch.int32))
exchange = bench.RecordExchange(world)
calls = []
def step():
    out = fake_out(rank)
    rec = pack_records(out)            # the CPU form of the same record layout
    calls.append(1)
    return out, exchange(rec)
elapsed, (out, gathered) = bench.timed_steps(step, 4, 2, world, dev)
assert len(calls) == 6 and elapsed > 0.0
assert tuple(gathered.shape) == (world * B, record_width(K))
for r in range(world):                                   # rank order, every rank's rows intact
    back = unpack_records(gathered[r * B:(r + 1) * B], K)
    want = fake_out(r)
    for key in ("box", "score", "mask_scores", "loc", "cls", "pred_masks", "counts"):
        assert torch.equal(back[key], want[key] if key not in ("cls",) else want[key].to(torch.int64)), (rank, r, key)
desc = bench.collective_description(record_width(K), world)
assert "over 2 ranks" in desc and "'gloo'" in desc, desc
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_bench_exchange_and_timed_region_world_size_2_gloo(tmp_path):
    """bench.py's own N > 1 code — RecordExchange (all_gather_into_tensor into a buffer allocated once) and timed_steps (barrier + sync on
    both sides, max over ranks) — driven by two CPU processes over gloo, as `bench.py --gpus 2` drives it over RCCL."""
    script = tmp_path / "bench_worker.py"
    script.write_text(_BENCH_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=180)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_average_precision_known_answers():
    from centermask2_amd.evaluation import average_precision, box_iou
    gt = [dict(boxes=torch.tensor([[0., 0., 10., 10.], [20., 20., 40., 40.]]), classes=torch.tensor([1, 2]))]
    perfect = [dict(boxes=gt[0]["boxes"].clone(), classes=gt[0]["classes"].clone(), scores=torch.tensor([0.9, 0.8]))]
    assert average_precision(perfect, gt, "box") == pytest.approx(1.0)
    # one box shifted so that IoU = 0.6: counted at thresholds .50-.60 only (3 of 10) for class 1; class 2 perfect
    shifted = [dict(boxes=torch.tensor([[2.5, 0., 12.5, 10.], [20., 20., 40., 40.]]), classes=torch.tensor([1, 2]), scores=torch.tensor([0.9, 0.8]))]
    assert float(box_iou(shifted[0]["boxes"][:1], gt[0]["boxes"][:1])) == pytest.approx(0.6)
    assert average_precision(shifted, gt, "box") == pytest.approx((0.3 + 1.0) / 2, abs=1e-6)
    # a higher-scored false positive in front of the true positive halves the precision of class 1
    fp = [dict(boxes=torch.tensor([[50., 50., 60., 60.], [0., 0., 10., 10.], [20., 20., 40., 40.]]), classes=torch.tensor([1, 1, 2]),
               scores=torch.tensor([0.95, 0.9, 0.8]))]
    assert average_precision(fp, gt, "box") == pytest.approx((0.5 + 1.0) / 2, abs=1e-6)
    m = torch.zeros((2, 8, 8), dtype=torch.bool); m[0, :4] = True; m[1, 4:] = True
    gm = [dict(boxes=gt[0]["boxes"], classes=torch.tensor([1, 1]), masks=m)]
    pm = [dict(boxes=gt[0]["boxes"], classes=torch.tensor([1, 1]), masks=m.clone(), scores=torch.tensor([0.1, 0.9]), mask_scores=torch.tensor([0.9, 0.1]))]
    assert average_precision(pm, gm, "mask") == pytest.approx(1.0)


def test_wire_formats_bin_and_coco_json(tmp_path):
    import numpy as np
    from centermask2_amd import wire
    from centermask2_amd.structures import Boxes, Instances
    g = torch.Generator().manual_seed(3)
    n = 5
    t6 = (torch.rand((n, 2), generator=g), torch.rand(n, generator=g), torch.rand((n, 4), generator=g) * 100,
          torch.randint(0, 80, (n,), generator=g), torch.rand((n, 1, 28, 28), generator=g), torch.rand(n, generator=g))
    paths = wire.to_bin(t6, str(tmp_path / "000000000139"))
    assert [os.path.getsize(p) for p in paths] == [n * 8, n * 4, n * 16, n * 8, n * 784 * 4, n * 4]      # dtypes of postprocess_bin_outputs.py:37
    back = wire.from_bin(str(tmp_path / "000000000139"))
    assert all(torch.equal(a, b) for a, b in zip(back, t6)) and back[3].dtype == torch.int64
    # RLE: round trip on random, empty, full and single-pixel masks; column-major run order; delta coding with negative deltas
    for m in (torch.rand((37, 53), generator=g) > 0.5, torch.zeros((4, 6), dtype=torch.bool), torch.ones((4, 6), dtype=torch.bool),
              torch.tensor([[0, 1], [0, 0]], dtype=torch.bool)):
        rle = wire.rle_encode(m)
        assert rle["size"] == list(m.shape) and isinstance(rle["counts"], str)
        assert np.array_equal(wire.rle_decode(rle), m.numpy())
    assert wire.rle_counts(np.array([[0, 1], [0, 0]])) == [2, 1, 1]               # column-major: 0,0,1,0
    assert wire.rle_counts(np.ones((2, 2))) == [0, 4]
    big = [0, 5000, 3, 100000, 2, 7]
    assert wire.rle_from_string(wire.rle_to_string(big)) == big
    inst = Instances((37, 53), pred_boxes=Boxes(torch.tensor([[1., 2., 11., 22.], [0., 0., 5., 5.]])), scores=torch.tensor([0.9, 0.8]),
                     pred_classes=torch.tensor([3, 7]), pred_masks=torch.rand((2, 37, 53), generator=g) > 0.5, mask_scores=torch.tensor([0.5, 0.6]))
    js = wire.instances_to_coco_json(inst, 139)
    assert js[0]["bbox"] == [1.0, 2.0, 10.0, 20.0] and js[0]["category_id"] == 3 and js[1]["mask_score"] == pytest.approx(0.6)
    seg = wire.segm_results_ranked_by_mask_score(js)
    assert "bbox" not in seg[0] and seg[0]["score"] == pytest.approx(0.5) and "mask_score" not in seg[0] and js[0]["score"] == pytest.approx(0.9)
    assert wire.instances_to_coco_json(inst[torch.zeros(2, dtype=torch.bool)], 1) == []


def test_device_mismatch_is_refused(monkeypatch):
    """The wrappers launch on torch's current stream: a tensor that lives on another GPU than the current one must be refused
    (it used to launch on GPU 0 with GPU-1 pointers).  No GPU needed: the check reads only attributes."""
    from centermask2_amd import ops
    from centermask2_amd._lib import CmkError

    class FakeCuda1:
        is_cuda, dtype, device = True, torch.float32, torch.device("cuda", 1)

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    with pytest.raises(CmkError, match="current device is cuda:0"):
        ops._need_gpu(FakeCuda1(), "conv2d")
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    ops._need_gpu(FakeCuda1(), "conv2d")                    # same device: accepted


def test_lazy_instances_between_the_plugins():
    """FCOS.forward hands out LazyInstances over its padded device buffers; CenterROIHeads attaches its outputs to the same batch; the
    first field access is the single host sync (counts, overflow re-run).  Plain tensors stand in for the device buffers here."""
    from centermask2_amd.modeling.centermask.center_heads import lazy_batch_of
    from centermask2_amd.modeling.fcos.fcos import LazyBatch, LazyInstances
    from centermask2_amd.structures import Boxes, Instances
    n, k = 2, 4
    det = dict(box=torch.arange(n * k * 4, dtype=torch.float32).reshape(n, k, 4), score=torch.rand(n, k), cls=torch.arange(n * k).reshape(n, k),
               loc=torch.rand(n, k, 2), counts=torch.tensor([3, 0], dtype=torch.int32), cand_counts=torch.tensor([10, 2], dtype=torch.int32), cand_capacity=16)

    class FakeFcos:
        calls = 0

        def resolve_overflow(self, d):
            FakeFcos.calls += 1
            d2 = dict(d)
            d2["cand_capacity"] = 64
            d2["counts"] = torch.tensor([4, 1], dtype=torch.int32)
            return d2

    batch = LazyBatch(det, [(8, 9), (8, 9)], FakeFcos())
    insts = [LazyInstances((8, 9), batch, i) for i in range(n)]
    assert isinstance(insts[0], Instances) and insts[0].image_size == (8, 9) and batch.counts is None
    assert lazy_batch_of(insts) is batch and lazy_batch_of(insts[::-1]) is None and lazy_batch_of(insts[:1]) is None
    batch.attach_roi(dict(det, pred_masks=torch.rand(n, k, 1, 28, 28), mask_scores=torch.rand(n, k)), lambda d2: None)
    assert batch.counts is None                                         # still nothing read
    assert len(insts[0]) == 3 and len(insts[1]) == 0 and batch.counts == [3, 0] and FakeFcos.calls == 0
    assert isinstance(insts[0].pred_boxes, Boxes) and torch.equal(insts[0].pred_boxes.tensor, det["box"][0, :3])
    assert tuple(insts[0].pred_masks.shape) == (3, 1, 28, 28) and tuple(insts[1].pred_masks.shape) == (0, 1, 28, 28)
    assert insts[0].has("mask_scores") and tuple(insts[1].mask_scores.shape) == (0,)
    assert type(insts[0][0:2]) is Instances and len(insts[0][0:2]) == 2
    import copy
    c = copy.copy(insts[1])                                             # copies behave like the original (materialised) object
    assert len(c) == 0 and c.image_size == (8, 9) and c.has("pred_masks")
    insts[0].pred_classes = torch.zeros(3, dtype=torch.int64)            # a caller's own value: the padded buffers no longer describe it
    assert lazy_batch_of(insts) is None and torch.equal(insts[0].pred_classes, torch.zeros(3, dtype=torch.int64))
    # overflow: the capacity was too small -> resolve() re-runs the tail (and the ROI heads) through the callbacks
    det2 = dict(det, cand_counts=torch.tensor([100, 2], dtype=torch.int32))
    redo = []
    b2 = LazyBatch(det2, [(8, 9), (8, 9)], FakeFcos())
    b2.attach_roi(dict(det2), lambda d2: redo.append(1) or dict(d2, pred_masks=torch.zeros(n, k, 1, 28, 28)))
    i2 = [LazyInstances((8, 9), b2, i) for i in range(n)]
    assert len(i2[0]) == 4 and len(i2[1]) == 1 and FakeFcos.calls == 1 and redo == [1] and i2[1].has("pred_masks")


def test_fp16_split_weight_packing_is_exact_to_22_bits():
    """Host logic of the opt-in fp16-split forms (cmk.h w_splith): S_w is the power of two that puts max |w S_w| in [2^14, 2^15); h + m
    reconstructs w S_w to 2^-22 relative for every weight within 2^17 of the layer's largest; the layout is [tap][chunk][tile][piece][lane][8]
    with lane = 32 * hh + li holding input channels 16 chunk + 8 hh + e of output channel 32 tile + li; zero padding of Cin and Cout."""
    import math
    from centermask2_amd import ops
    g = torch.Generator().manual_seed(5)
    for cout, cin, k in ((40, 48, 3), (130, 32, 1)):
        w = torch.randn((cout, cin, k, k), generator=g) * 0.07
        w[0, 0, 0, 0] = 0.9                                   # the layer's largest weight
        w[1, 1, 0, 0] = 0.9 * 2.0 ** -16                      # near the bottom of the fully represented range
        packed, inv_s = ops.pack_splith_weight(w)
        s_w = 1.0 / inv_s
        assert math.log2(s_w) == int(math.log2(s_w)) and 2.0 ** 14 <= 0.9 * s_w < 2.0 ** 15
        taps, cin_pad, cout_pad = k * k, (cin + 15) // 16 * 16, (cout + 127) // 128 * 128
        assert tuple(packed.shape) == (taps, cin_pad // 16, cout_pad // 32, 2, 64, 8) and packed.dtype == torch.float16
        # undo the layout by the documented index formula
        r = packed.reshape(taps, cin_pad // 16, cout_pad // 32, 2, 2, 32, 8)          # [tap][chunk][tile][piece][hh][li][e]
        full = r.permute(3, 2, 5, 0, 1, 4, 6).reshape(2, cout_pad, taps, cin_pad)       # [piece][cout][tap][cin]
        h, m = full[0].double(), full[1].double()
        want = torch.zeros((cout_pad, taps, cin_pad), dtype=torch.float64)
        want[:cout, :, :cin] = w.double().reshape(cout, cin, taps).permute(0, 2, 1) * s_w
        err = (h + m - want).abs()
        big = want.abs() >= 2.0 ** -2                          # both pieces normal fp16
        assert float((err[big] / want.abs()[big]).max()) <= 2.0 ** -22
        assert float(err[~big].max()) <= 2.0 ** -24 * 1.01     # below that: the absolute resolution of fp16's subnormals (2^-25 per piece)
        assert float(h[cout:].abs().max()) == 0.0 and float(h[:, :, cin:].abs().max() if cin_pad > cin else 0.0) == 0.0
        assert float((h - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -11      # h alone is fp16(w S_w)
