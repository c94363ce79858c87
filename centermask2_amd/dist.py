"""Multi-GPU data parallelism for inference: shard images over ranks, all-gather fixed-stride results.

Images are independent units: weights are replicated, rank r of W takes a contiguous slice of the batch,
and the only exchange is ONE all-gather of padded per-image records at the end of the batch — torch.distributed
backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  The reference's analogue is
`comm.gather(self._predictions, dst=0)` in its evaluator (evaluation/coco_evaluation.py:155-156).
"""
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist

RECORD_FIELDS = ("box", "score", "mask_scores", "loc", "cls", "pred_masks", "counts")


def shard_range(num_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of the batch for `rank` (first `num_images % world` ranks get one extra)."""
    base, extra = divmod(num_images, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def record_width(topk: int, mask_hw: int = 28) -> int:
    return topk * (4 + 1 + 1 + 2 + 1 + mask_hw * mask_hw) + 1


def pack_records(out: Dict[str, torch.Tensor], rec: torch.Tensor = None) -> torch.Tensor:
    """(n, record_width) float32: [box 4K | score K | mask_score K | loc 2K | cls K | mask 784K | count].
    On the GPU this is ONE kernel (cmk_pack_records) writing straight into `rec` — pass the all-gather send buffer to reuse it every
    step; on CPU tensors (the gloo tests of the exchange logic) the same layout is assembled with torch ops."""
    n, k = out["score"].shape
    hw = out["pred_masks"].shape[-1]
    if not out["score"].is_cuda:
        return torch.cat([out["box"].reshape(n, -1), out["score"], out["mask_scores"], out["loc"].reshape(n, -1),
                          out["cls"].to(torch.float32), out["pred_masks"].reshape(n, -1),
                          out["counts"].to(torch.float32).reshape(n, 1)], dim=1).contiguous()
    from . import _lib, ops
    if rec is None:
        rec = torch.empty((n, record_width(k, hw)), dtype=torch.float32, device=out["score"].device)
    assert rec.shape == (n, record_width(k, hw)) and rec.is_contiguous()
    ops._need_gpu(out["score"], "pack_records")
    _lib.check(_lib.load().cmk_pack_records(out["box"].data_ptr(), out["score"].data_ptr(), out["mask_scores"].data_ptr(), out["loc"].data_ptr(),
                                            out["cls"].data_ptr(), out["pred_masks"].data_ptr(), out["counts"].data_ptr(), n, k, hw,
                                            rec.data_ptr(), ops._stream()), "cmk_pack_records")
    return rec


def unpack_records(rec: torch.Tensor, topk: int, mask_hw: int = 28) -> Dict[str, torch.Tensor]:
    n = rec.shape[0]
    k, o = topk, 0
    out = {}
    for name, width, shape in (("box", 4 * k, (n, k, 4)), ("score", k, (n, k)), ("mask_scores", k, (n, k)), ("loc", 2 * k, (n, k, 2)),
                               ("cls", k, (n, k)), ("pred_masks", k * mask_hw * mask_hw, (n, k, 1, mask_hw, mask_hw)), ("counts", 1, (n,))):
        out[name] = rec[:, o:o + width].reshape(shape)
        o += width
    out["cls"] = out["cls"].round().to(torch.int64)
    out["counts"] = out["counts"].round().to(torch.int32)
    return out


def all_gather_records(rec: torch.Tensor, per_rank: int, num_images: int) -> torch.Tensor:
    """All-gather equally sized shards (`per_rank` rows, zero-padded) and drop the padding: (num_images, width)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return rec[:num_images]
    buf = torch.zeros((per_rank, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    buf[:rec.shape[0]] = rec
    gathered = torch.empty((world * per_rank, rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(gathered, buf)
    rows = []
    for r in range(world):
        lo, hi = shard_range(num_images, r, world)
        rows.append(gathered[r * per_rank: r * per_rank + (hi - lo)])
    return torch.cat(rows, 0)
