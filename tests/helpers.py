import os

import torch

GOLDEN_ROOT = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(GOLDEN_ROOT, "golden")


def golden(name):
    return torch.load(os.path.join(GOLDEN, name + ".pt"), weights_only=True)


def close(got, ref, tol, what=""):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, tuple(got.shape), tuple(ref.shape))
    if got.numel() == 0:
        return 0.0
    err = (got - ref).abs().max().item()
    bound = tol * max(1.0, ref.abs().max().item())
    assert err <= bound, "{}: max abs err {:.3e} > {:.3e}".format(what, err, bound)
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
