"""Seeded synthetic weights and inputs; data generation only, no compute path.

No checkpoint or dataset is reachable offline (the reference's weights are Dropbox URLs,
README.md:173,254), so parity and the benchmark run on random-init weights written under the
reference's state-dict key names and on uint8-valued BGR noise images preprocessed like
deploy_utils.py:76-83.  Every tensor is drawn from its own generator seeded by
crc32(key) ^ seed, so the values do not depend on the enumeration order or on who enumerates.
"""
import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import torch

# vovnet.py:30-108
STAGE_SPECS = {
    "V-19-slim-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                             layer_per_block=3, block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-dw-eSE": dict(stem=[64, 64, 64], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                        layer_per_block=3, block_per_stage=[1, 1, 1, 1], dw=True),
    "V-19-slim-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[64, 80, 96, 112], stage_out_ch=[112, 256, 384, 512],
                          layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-19-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=3, block_per_stage=[1, 1, 1, 1]),
    "V-39-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 2, 2]),
    "V-57-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 1, 4, 3]),
    "V-99-eSE": dict(stem=[64, 64, 128], stage_conv_ch=[128, 160, 192, 224], stage_out_ch=[256, 512, 768, 1024],
                     layer_per_block=5, block_per_stage=[1, 3, 9, 3]),
}

PIXEL_MEAN = (103.53, 116.28, 123.675)  # deploy_utils.py:81 (BGR), std 1


def model_param_shapes(conv_body: str = "V-39-eSE", num_classes: int = 80, fpn_ch: int = 256,
                       mask_dim: int = 256, pooler_res: int = 14) -> "OrderedDict[str, Tuple[int, ...]]":
    """Every state-dict entry of the full model in the reference's key names (SURVEY §5 'checkpoint')."""
    spec = STAGE_SPECS[conv_body]
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def conv_bn(prefix: str, cin: int, cout: int, k: int) -> None:
        s[prefix + "/conv.weight"] = (cout, cin, k, k)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[prefix + "/norm." + n] = (cout,)

    def dw_pw_bn(prefix: str, c: int) -> None:      # dw_conv3x3 vovnet.py:110-130
        s[prefix + "/dw_conv3x3.weight"] = (c, 1, 3, 3)
        s[prefix + "/pw_conv1x1.weight"] = (c, c, 1, 1)
        for n in ("weight", "bias", "running_mean", "running_var"):
            s[prefix + "/pw_norm." + n] = (c,)

    dw = spec.get("dw", False)
    bu = "backbone.bottom_up."
    stem = spec["stem"]
    conv_bn(bu + "stem.stem_1", 3, stem[0], 3)
    if dw:
        assert stem[0] == stem[1] == stem[2]
        dw_pw_bn(bu + "stem.stem_2", stem[1])
        dw_pw_bn(bu + "stem.stem_3", stem[2])
    else:
        conv_bn(bu + "stem.stem_2", stem[0], stem[1], 3)
        conv_bn(bu + "stem.stem_3", stem[1], stem[2], 3)
    in_ch = stem[2]
    for si in range(4):
        stage_ch, concat_ch = spec["stage_conv_ch"][si], spec["stage_out_ch"][si]
        for b in range(spec["block_per_stage"][si]):
            mod = "OSA{}_{}".format(si + 2, b + 1)
            p = bu + "stage{}.{}.".format(si + 2, mod)
            cin = in_ch
            if dw and in_ch != stage_ch:            # conv_reduction vovnet.py:284-288
                conv_bn(p + "conv_reduction.{}_reduction_0".format(mod), in_ch, stage_ch, 1)
            for i in range(spec["layer_per_block"]):
                if dw:
                    dw_pw_bn(p + "layers.{}.{}_{}".format(i, mod, i), stage_ch)
                else:
                    conv_bn(p + "layers.{}.{}_{}".format(i, mod, i), cin, stage_ch, 3)
                cin = stage_ch
            conv_bn(p + "concat.{}_concat".format(mod), in_ch + spec["layer_per_block"] * stage_ch, concat_ch, 1)
            s[p + "ese.fc.weight"] = (concat_ch, concat_ch, 1, 1)
            s[p + "ese.fc.bias"] = (concat_ch,)
            in_ch = concat_ch
    for lvl, cin in zip((3, 4, 5), spec["stage_out_ch"][1:]):
        s["backbone.fpn_lateral{}.weight".format(lvl)] = (fpn_ch, cin, 1, 1)
        s["backbone.fpn_lateral{}.bias".format(lvl)] = (fpn_ch,)
        s["backbone.fpn_output{}.weight".format(lvl)] = (fpn_ch, fpn_ch, 3, 3)
        s["backbone.fpn_output{}.bias".format(lvl)] = (fpn_ch,)
    for n in ("p6", "p7"):
        s["backbone.top_block.{}.weight".format(n)] = (fpn_ch, fpn_ch, 3, 3)
        s["backbone.top_block.{}.bias".format(n)] = (fpn_ch,)
    h = "proposal_generator.fcos_head."
    for tower in ("cls_tower", "bbox_tower"):
        for k in range(4):
            s[h + "{}.{}.weight".format(tower, 3 * k)] = (fpn_ch, fpn_ch, 3, 3)
            s[h + "{}.{}.bias".format(tower, 3 * k)] = (fpn_ch,)
            s[h + "{}.{}.weight".format(tower, 3 * k + 1)] = (fpn_ch,)
            s[h + "{}.{}.bias".format(tower, 3 * k + 1)] = (fpn_ch,)
    for n, c in (("cls_logits", num_classes), ("bbox_pred", 4), ("ctrness", 1)):
        s[h + n + ".weight"] = (c, fpn_ch, 3, 3)
        s[h + n + ".bias"] = (c,)
    for l in range(5):
        s[h + "scales.{}.scale".format(l)] = (1,)
    m = "roi_heads.mask_head."
    for k in range(4):
        s[m + "mask_fcn{}.weight".format(k + 1)] = (mask_dim, fpn_ch if k == 0 else mask_dim, 3, 3)
        s[m + "mask_fcn{}.bias".format(k + 1)] = (mask_dim,)
    s[m + "spatialAtt.conv.weight"] = (1, 2, 3, 3)
    s[m + "deconv.weight"] = (mask_dim, mask_dim, 2, 2)
    s[m + "deconv.bias"] = (mask_dim,)
    s[m + "predictor.weight"] = (num_classes, mask_dim, 1, 1)
    s[m + "predictor.bias"] = (num_classes,)
    q = "roi_heads.maskiou_head."
    for k in range(4):
        s[q + "maskiou_fcn{}.weight".format(k + 1)] = (mask_dim, fpn_ch + 1 if k == 0 else mask_dim, 3, 3)
        s[q + "maskiou_fcn{}.bias".format(k + 1)] = (mask_dim,)
    res = pooler_res // 2
    s[q + "maskiou_fc1.weight"] = (1024, mask_dim * res * res)
    s[q + "maskiou_fc1.bias"] = (1024,)
    s[q + "maskiou_fc2.weight"] = (1024, 1024)
    s[q + "maskiou_fc2.bias"] = (1024,)
    s[q + "maskiou.weight"] = (num_classes, 1024)
    s[q + "maskiou.bias"] = (num_classes,)
    return s


# Frozen fixture constants (chosen once with the oracle so that every fixture image yields a few
# thousand candidates > 0.05 and >= 50 post-NMS detections spread over the three ROI levels).
SYNTH = dict(cls_logits_std=0.035, cls_logits_bias=-5.6, bbox_pred_std=0.05, bbox_pred_bias=1.0,
             ctrness_std=0.03, predictor_std=0.5, maskiou_std=0.002, scale_lo=6.0, scale_hi=12.0)


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator()
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def synthetic_tensor(name: str, shape: Iterable[int], seed: int = 0) -> torch.Tensor:
    shape = tuple(shape)
    g = _gen(name, seed)
    randn = lambda std=1.0, mean=0.0: torch.randn(shape, generator=g) * std + mean
    rand = lambda lo, hi: torch.rand(shape, generator=g) * (hi - lo) + lo
    leaf = name.rsplit(".", 1)[-1]
    if name.endswith("stem_1/norm.running_var"):
        return rand(0.5, 1.5) * 400.0                 # pixel-scale inputs (std ~20) are normalised by the first BN
    if name.endswith("norm.weight") or name.endswith("norm.running_var"):     # '/norm.' and the dw bodies' '/pw_norm.'
        return rand(0.5, 1.5)
    if name.endswith("norm.bias") or name.endswith("norm.running_mean"):
        return randn(0.1)
    if name.endswith("/dw_conv3x3.weight"):
        return randn(1.0 / 3.0)                       # no ReLU between the dw and pw convs: unit gain
    if ".scales." in name:
        return rand(SYNTH["scale_lo"], SYNTH["scale_hi"])
    if "ese.fc.weight" in name:
        return randn(1.0 / math.sqrt(shape[1]))
    if "ese.fc.bias" in name:
        return randn(1.0)
    if "cls_logits" in name:
        if leaf == "weight":                          # zero-mean per class: the post-ReLU tower output has a
            w = randn(SYNTH["cls_logits_std"])        # positive mean that would otherwise favour one class
            return w - w.mean(dim=(1, 2, 3), keepdim=True)
        return randn(0.02, SYNTH["cls_logits_bias"])
    if "bbox_pred" in name:
        return randn(SYNTH["bbox_pred_std"]) if leaf == "weight" else randn(0.2, SYNTH["bbox_pred_bias"])
    if "ctrness" in name:
        return randn(SYNTH["ctrness_std"]) if leaf == "weight" else randn(0.1)
    if "mask_head.predictor" in name:
        return randn(SYNTH["predictor_std"]) if leaf == "weight" else randn(0.1)
    if name.endswith("maskiou.weight"):
        return randn(SYNTH["maskiou_std"])
    if name.endswith("maskiou.bias"):
        return randn(0.05, 0.5)
    if "_tower." in name and len(shape) == 1 and int(name.split(".")[-2]) % 3 == 1:   # GroupNorm affine
        return rand(0.5, 1.5) if leaf == "weight" else randn(0.1)
    if "fpn_lateral" in name and len(shape) == 4:
        return randn(0.35 * math.sqrt(1.0 / shape[1]))  # no ReLU follows: keep p-levels O(1)
    if ("fpn_output" in name or "top_block" in name) and len(shape) == 4:
        return randn(math.sqrt(1.0 / (shape[1] * 9)))
    if len(shape) == 4:
        fan_in = shape[1] * shape[2] * shape[3]
        if "deconv" in name:
            fan_in = shape[0]
        return randn(math.sqrt(2.0 / fan_in))       # Kaiming-normal(fan_in): activations neither vanish nor explode
    if len(shape) == 2:
        return randn(math.sqrt(2.0 / shape[1]))
    return randn(0.1)                                 # remaining biases


def make_synthetic_state_dict(conv_body: str = "V-39-eSE", seed: int = 0, shapes=None) -> Dict[str, torch.Tensor]:
    """Deep bodies (stages of >= 3 OSA blocks: V-57, V-99) get the FrozenBN affine of every identity block's 1x1 aggregation scaled by
    1/sqrt(blocks in the stage).  Each identity block computes x + eSE(concat(x)); with unit-gain random branches the activations grow
    geometrically over 9 blocks (V-99: |p3| up to 450) and the fp32 arithmetic of the REFERENCE itself then sits 1.6e-4 (features),
    2e-3 (logits) and 0.9 px (boxes) away from a float64 evaluation (tools/diag_v99.py) — no fp32 implementation could be compared
    with it at 1e-3.  A trained network keeps its residual branches small; so does this scaling (|p3| <= 33, fp32 vs fp64 7e-7)."""
    import re
    shapes = shapes if shapes is not None else model_param_shapes(conv_body)
    sd = OrderedDict((k, synthetic_tensor(k, v, seed).float().contiguous()) for k, v in shapes.items())
    blocks = STAGE_SPECS[conv_body]["block_per_stage"]
    for k in sd:
        m = re.search(r"stage(\d)\.OSA\d_(\d+)\.concat\..*norm\.(weight|bias)$", k)
        if m and int(m.group(2)) >= 2 and blocks[int(m.group(1)) - 2] >= 3:
            sd[k] = (sd[k] * blocks[int(m.group(1)) - 2] ** -0.5).contiguous()
    return sd


def make_synthetic_images(batch: int, height: int = 800, width: int = 1280, seed0: int = 1234,
                          first: int = 0) -> torch.Tensor:
    """(batch,3,H,W) float32: uint8-valued BGR noise minus PIXEL_MEAN (deploy_utils.py:81-83).
    Image i of the global job uses seed seed0+i; `first` is the global index of this batch's first image.
    Smooth blobs are blended in so the image is not pure white noise (keeps activations image-dependent)."""
    imgs = []
    mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    for i in range(batch):
        g = torch.Generator()
        g.manual_seed(seed0 + first + i)
        coarse = torch.randint(0, 256, (1, 3, max(height // 32, 1), max(width // 32, 1)), generator=g).float()
        low = torch.nn.functional.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=False)[0]
        noise = torch.randint(0, 256, (3, height, width), generator=g).float()
        img = torch.floor(0.75 * low + 0.25 * noise).clamp_(0, 255)
        imgs.append(img - mean)
    return torch.stack(imgs).contiguous()
