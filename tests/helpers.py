import os

import torch

GOLDEN_ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(GOLDEN_ROOT, "golden")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


OBSERVED = {}      # what -> largest absolute error seen this session (printed by conftest at the end of a run: the margin on record)


def _record(what, err):
    if what:
        OBSERVED[what] = max(OBSERVED.get(what, 0.0), float(err))


def close(got, ref, tol, what=""):
    """Quantities with a natural scale (pixels, features of a single random op): max abs err <= tol * max(1, max|ref|)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    if got.numel() == 0:
        return 0.0
    err = (got - ref).abs().max().item()
    bound = tol * max(1.0, ref.abs().max().item())
    _record(what + " (rel)", err / max(1.0, ref.abs().max().item()))
    assert err <= bound, "{}: max abs err {:.3e} > {:.3e}".format(what, err, bound)
    return err


def close_abs(got, ref, tol, what=""):
    """north_star's bar for logits / regression / centerness / mask logits / scores: max ABSOLUTE error <= tol, whatever the
    magnitude of the reference (no max|ref| factor)."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    if got.numel() == 0:
        return 0.0
    err = (got - ref).abs().max().item()
    _record(what + " (abs)", err)
    assert err <= tol, "{}: max abs err {:.3e} > {:.3e} (absolute)".format(what, err, tol)
    return err


def build_gpu_model(conv_body="V-39-eSE", seed=0):
    from centermask2_amd.config import get_cfg, config_path
    from centermask2_amd.modeling import build_model
    from centermask2_amd import synthetic as S
    cfg = get_cfg()
    name = "centermask_V_99_eSE_FPN_ms_3x.yaml" if conv_body == "V-99-eSE" else "centermask_V_39_eSE_FPN_ms_3x.yaml"
    cfg.merge_from_file(config_path(name))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda", "MODEL.VOVNET.CONV_BODY", conv_body])
    cfg.freeze()
    sd = S.make_synthetic_state_dict(conv_body, seed)
    model = build_model(cfg).eval()
    model.load_state_dict(sd)
    return model, sd


def load_shipped_variant_table():
    """The measured conv variant table bench.py uses (Winograd where it wins), so parity is checked on the same kernels."""
    from centermask2_amd import ops
    path = os.path.join(os.path.dirname(GOLDEN_ROOT), "centermask2_amd", "tuned", "mi355x_V-39-eSE_b8_800x1280.json")
    return ops.load_tuned(path) if os.path.exists(path) else 0


def match_detections(a_scores, a_classes, a_locations, b_scores, b_classes, b_locations, tol=1e-4):
    """perm with b[perm[i]] == a[i] as detections — identified by (class, location), which is unique after per-class NMS — or an
    AssertionError.  The two lists must hold the SAME detections; their descending-score ORDER may differ only where the scores
    are within `tol`: descending-score order is defined up to the numeric noise of the scores, and on the deep V-99 body the CPU
    oracle itself orders two detections 9e-6 apart differently on different hosts (conv reduction order depends on the CPU's
    vector width and thread count).  Everything else — which detections, their labels, their locations — is exact."""
    a_scores, b_scores = a_scores.detach().float().cpu(), b_scores.detach().float().cpu()
    a_classes, b_classes = a_classes.detach().cpu(), b_classes.detach().cpu()
    a_locations, b_locations = a_locations.detach().float().cpu(), b_locations.detach().float().cpu()
    assert a_scores.shape == b_scores.shape, (a_scores.shape, b_scores.shape)
    key = lambda c, l, k: (int(c[k]), float(l[k, 0]), float(l[k, 1]))
    where = {key(b_classes, b_locations, k): k for k in range(b_scores.shape[0])}
    assert len(where) == b_scores.shape[0], "(class, location) must identify a detection"
    perm = []
    for i in range(a_scores.shape[0]):
        k = where.get(key(a_classes, a_locations, i))
        assert k is not None, "detection {} (class {}, location {}) has no counterpart".format(i, int(a_classes[i]), a_locations[i].tolist())
        assert abs(float(b_scores[k]) - float(b_scores[i])) <= tol, \
            "detection {} sits at rank {} in the other list and the scores there differ by more than {}: a real order change".format(i, k, tol)
        perm.append(k)
    assert sorted(perm) == list(range(len(perm)))
    return torch.tensor(perm, dtype=torch.int64)
