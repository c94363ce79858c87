"""Golden vectors for the other VoVNet bodies, produced by the REFERENCE's own VoVNet/FPN.

    python tests/golden/make_golden_bodies.py        # needs /root/reference; writes tests/golden/vovnet_bodies.pt

For each body: (1) the reference's state-dict keys/shapes must equal synthetic.model_param_shapes(body),
(2) the reference backbone runs on seeded weights and a small odd-sized input, (3) the oracle must agree,
(4) input and outputs are stored as plain tensors.  Data only — no reference source.
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the d2 stand-ins and imports the reference package)

S, O = G.S, G.O
BODIES = ("V-19-slim-dw-eSE", "V-19-dw-eSE", "V-19-slim-eSE", "V-19-eSE", "V-57-eSE")
FORWARD = {"V-19-slim-dw-eSE": (2, 76, 108), "V-19-dw-eSE": (1, 64, 96), "V-19-slim-eSE": (1, 76, 108), "V-19-eSE": (1, 76, 108),
           "V-57-eSE": (1, 76, 108)}


def main():
    out = {}
    for body in BODIES:
        _, backbone, fcos, roi_heads = G.build_reference(body)
        ref_sd = G.full_state_dict(backbone, fcos, roi_heads)
        shapes = S.model_param_shapes(body)
        assert set(ref_sd.keys()) == set(shapes.keys()), (body, sorted(set(ref_sd.keys()) ^ set(shapes.keys()))[:8])
        for k, v in ref_sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), (body, k, tuple(v.shape), shapes[k])
        print("state dict: {} keys match for {}".format(len(shapes), body))
        if body not in FORWARD:
            continue
        sd = S.make_synthetic_state_dict(body, seed=0)
        G.load_synthetic(backbone, fcos, roi_heads, sd)
        n, h, w = FORWARD[body]
        with torch.no_grad():
            x = S.make_synthetic_images(n, h, w, seed0=301)      # odd size: bottom-up only (pins ceil_mode pooling)
            ref_bu = backbone.bottom_up(x)
            orc_bu = O.vovnet_forward(sd, x, conv_body=body)
            x32 = S.make_synthetic_images(1, 64, 96, seed0=302)  # /32 size: VoVNet + FPN + P6/P7
            ref = backbone(x32)
            orc = O.backbone_forward(sd, x32, conv_body=body)
        case = dict(x=x, x32=x32)
        for k in ref_bu:
            print(body, k, tuple(ref_bu[k].shape), "absmax %.3f" % float(ref_bu[k].abs().max()), G.close(orc_bu[k], ref_bu[k], 1e-5, body + " " + k))
            case[k] = ref_bu[k].clone()
        for k in ref:
            print(body, k, tuple(ref[k].shape), G.close(orc[k], ref[k], 1e-5, body + " " + k))
            case[k] = ref[k].clone()
        out[body] = case
    path = os.path.join(HERE, "vovnet_bodies.pt")
    torch.save(out, path)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
