#!/usr/bin/env python3
"""CPU emulation of fp32 Winograd forms for the 3x3 stride-1 convs of the path, to decide which forms can pass the parity
gates BEFORE a kernel is written (F(4x4,3x3), fp32 only, gated by parity).

The emulation does what a fused kernel would: U = G g G^T in fp64 rounded once to fp32; input transform, the per-frequency
channel contraction and the output transform all in fp32.  It is patched into the oracle in place of F.conv2d for every
3x3 / stride 1 / Cin >= 32 conv and the whole model is run on one 800x1280 image; the result is compared with the plain
oracle (direct fp32 convs).  Forms: "2x2" (what ships), "4x4", "2x4" (F(2,3) down the rows x F(4,3) along the columns).

    python tools/wino_numerics.py [--form 4x4] [--body V-39-eSE] [--images 1]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# Lavin & Gray's matrices: points 0, +-1, inf for F(2,3); 0, +-1, +-2, inf for F(4,3)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
MATS = {2: (BT2, G2, AT2), 4: (BT4, G4, AT4)}


def wino_conv(x, w, bias, mh, mw):
    """F(mh x mw, 3x3), pad 1, fp32 arithmetic."""
    n, c, h, wd = x.shape
    o = w.shape[0]
    bth, gh, ath = MATS[mh]
    btw, gw, atw = MATS[mw]
    th, tw = -(-h // mh), -(-wd // mw)
    xp = F.pad(x, (1, tw * mw - wd + 1, 1, th * mh - h + 1))
    u = torch.einsum("ik,ockl,jl->ijoc", gh, w.double(), gw).float()                        # (a, b, O, C), rounded once
    t = xp.unfold(2, mh + 2, mh).unfold(3, mw + 2, mw)                                      # (N, C, th, tw, a, b)
    v = torch.einsum("ia,nctsab,jb->ijncts", bth.float(), t, btw.float())                   # fp32 transform
    v = v.reshape(mh + 2, mw + 2, n, c, th * tw)
    m = torch.einsum("ijoc,ijncp->ijnop", u, v)                                            # fp32 contraction over channels
    y = torch.einsum("ia,abnop,jb->nopij", ath.float(), m, atw.float())                     # (N, O, P, mh, mw)
    y = y.reshape(n, o, th, tw, mh, mw).permute(0, 1, 2, 4, 3, 5).reshape(n, o, th * mh, tw * mw)[:, :, :h, :wd]
    if bias is not None:
        y = y + bias[None, :, None, None]
    return y.contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", default="4x4")
    ap.add_argument("--body", default="V-39-eSE")
    ap.add_argument("--images", type=int, default=1)
    ap.add_argument("--min-hw", type=int, default=0, help="use the form only on maps with at least this many pixels (else direct)")
    args = ap.parse_args()
    mh, mw = (int(v) for v in args.form.split("x"))
    from centermask2_amd import synthetic as S
    from oracle import centermask_oracle as O
    sd = S.make_synthetic_state_dict(args.body, 0)
    real_conv = F.conv2d
    stats = {"n": 0}

    def patched(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        if (w.shape[2] == 3 and w.shape[3] == 3 and stride in (1, (1, 1)) and groups == 1 and w.shape[1] >= 32 and padding in (1, (1, 1))
                and x.shape[2] * x.shape[3] >= args.min_hw):
            stats["n"] += 1
            outs = [wino_conv(x[i:i + 1], w, b, mh, mw) for i in range(x.shape[0])]
            return torch.cat(outs, 0)
        return real_conv(x, w, b, stride, padding, dilation, groups)

    class FP:                                  # the oracle calls F.<op>; only conv2d is redirected
        def __getattr__(self, k):
            return patched if k == "conv2d" else getattr(F, k)

    for i in range(args.images):
        x = S.make_synthetic_images(1, 800, 1280, seed0=1234, first=i)
        t0 = time.time()
        ref, ri = O.centermask_inference(sd, x, [(800, 1280)], args.body, return_intermediates=True)
        t1 = time.time()
        O.F = FP()
        try:
            got, gi = O.centermask_inference(sd, x, [(800, 1280)], args.body, return_intermediates=True)
        finally:
            O.F = F
        t2 = time.time()
        r, g = ref[0], got[0]
        n = r["scores"].shape[0]
        same_n = g["scores"].shape[0] == n
        print("image {}: form F({}x{},3x3) on {} convs; oracle {:.1f}s, emulation {:.1f}s".format(i, mh, mw, stats["n"], t1 - t0, t2 - t1))
        print("  detections {} vs {}".format(n, g["scores"].shape[0]))
        if same_n:
            print("  labels equal: {}   locations equal: {}".format(bool(torch.equal(r["classes"], g["classes"])), bool(torch.equal(r["locations"], g["locations"]))))
            for k in ("boxes", "scores", "pred_masks", "mask_scores"):
                print("  {:12s} max abs diff {:.3e}".format(k, float((r[k] - g[k]).abs().max())))
        for k in ("p3", "p5", "p7"):
            a, b = ri["features"][k], gi["features"][k]
            print("  feature {:4s} max abs diff {:.3e} rms rel {:.3e} (max abs {:.2f})".format(k, float((a - b).abs().max()), float((a - b).norm() / a.norm()), float(a.abs().max())))
        for nm in ("logits", "bbox_reg", "ctrness"):
            d = max(float((a - b).abs().max()) for a, b in zip(ri[nm], gi[nm]))
            rr = max(float((a - b).norm() / a.norm()) for a, b in zip(ri[nm], gi[nm]))
            print("  {:12s} max abs diff {:.3e} rms rel {:.3e}".format(nm, d, rr))
        print("  candidates {} vs {}".format(ri["candidates"][0]["scores"].shape[0], gi["candidates"][0]["scores"].shape[0]))
        if same_n and torch.equal(r["classes"], g["classes"]):
            a, b = ri["mask_logits"], gi["mask_logits"]
            idx = torch.arange(a.shape[0])
            print("  sel mask logits max abs diff {:.3e}; maskiou {:.3e}".format(float((a[idx, r["classes"]] - b[idx, r["classes"]]).abs().max()), float((ri["maskiou"] - gi["maskiou"]).abs().max())))
        stats["n"] = 0


if __name__ == "__main__":
    main()
