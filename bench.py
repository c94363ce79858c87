#!/usr/bin/env python3
"""CenterMask2 V2-39-eSE-FPN inference throughput on MI355X (BASELINE.json metric: images/sec at 3x800x1280).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the whole hot path (VoVNetV2-39-eSE + FPN -> FCOS head/decode/NMS/top-50 -> CenterROIHeads with
SAG-Mask + MaskIoU) over one batch of 8 synthetic 3x800x1280 images per GPU, inputs already resident in HBM,
results left as padded device buffers.  Images shard across ranks (weak scaling, no data-path collective inside the
model); each step ends with one RCCL all-gather of the fixed-stride per-image results.
Rank 0 prints ONE JSON line, with
  roofline     — the dominant kernel (the fp32 Winograd F(4x4,3x3) conv on the matrix pipe) measured live with HIP events on the launch stream;
  cpu_baseline — the oracle (CPU restatement of the reference path, kind "port") timed on this host's cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_HBM_GBS = 8000.0
PEAK_16BIT_MATRIX_TFLOPS = 2010.0  # v_mfma_f32_32x32x16_{bf16,f16} measured on random operands (tools/probe/mfma_power_bf16.hip; nominal dense 2516 at 2.4 GHz): the pipe of the opt-in split kernels
PROFILE_ROUND = "r03"            # prefix of the PMC summaries under profiles/ this bench quotes


def build(conv_body, device):
    from centermask2_amd import synthetic as S
    from centermask2_amd.config import config_path, get_cfg
    from centermask2_amd.modeling import build_model
    cfg = get_cfg()
    cfg.merge_from_file(config_path("centermask_V_39_eSE_FPN_ms_3x.yaml" if conv_body == "V-39-eSE" else "centermask_V_99_eSE_FPN_ms_3x.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", str(device), "MODEL.VOVNET.CONV_BODY", conv_body])
    cfg.freeze()
    sd = S.make_synthetic_state_dict(conv_body, 0)
    model = build_model(cfg).eval()
    model.load_state_dict(sd)
    return model, sd


def pack_results(out):
    """Fixed-stride per-image record for the all-gather (centermask2_amd/dist.py)."""
    from centermask2_amd.dist import pack_records
    return pack_records(out)


class RecordExchange:
    """The step's only collective: one all_gather_into_tensor of this rank's fixed-stride records (B rows) into a
    (world * B)-row buffer allocated once — "nccl" = RCCL over xGMI on the GPUs, gloo in the CPU rehearsal (tests/test_cpu_surface.py
    drives this class and timed_steps with two gloo ranks).  Weak scaling: every rank holds the same number of rows."""

    def __init__(self, world: int):
        self.world, self.gathered = world, None

    def __call__(self, rec: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return rec
        if self.gathered is None:
            self.gathered = torch.empty((self.world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
        dist.all_gather_into_tensor(self.gathered, rec)
        return self.gathered


def _sync(dev) -> None:
    if dev.type == "cuda":
        torch.cuda.synchronize()


def timed_steps(step, steps: int, warmup: int, world: int, dev):
    """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by a barrier + device synchronisation on both
    sides; the elapsed time is the MAX over ranks.  Returns (seconds, what the last step returned)."""
    out = None
    for _ in range(warmup):
        step()
    if world > 1:
        dist.barrier()
    _sync(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    if world > 1:
        dist.barrier()
    _sync(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def collective_description(rec_floats: int, world: int) -> str:
    if world == 1:
        return "none"
    be = dist.get_backend()
    return "{} all_gather_into_tensor of {} B/img records over {} ranks (torch.distributed backend '{}')".format(
        "RCCL" if be == "nccl" else be, rec_floats * 4, dist.get_world_size(), be)


def roofline_leg(model, x, sizes):
    """One instrumented step: HIP events (torch's current stream == the launch stream) around every conv launch.
    `achieved`/`frac` price the dominant kernel on the FLOPs the matrix pipe EXECUTES (Winograd-domain multiplies, tile padding
    included) against the fp32 MFMA peak, so frac <= 1; the direct-convolution (algorithmic) rate SURVEY §8(d) defines is reported
    beside it as alg_equiv_TFLOPs.  MFMA-busy and HBM traffic come from separate rocprofv3 --pmc passes of this same command
    (a process cannot read the PMCs of its own launches) kept under profiles/, stamped with the hash of the kernel sources:
    a summary of an older kernel is reported as null, never quoted."""
    from centermask2_amd import ops
    ops.PROFILE = []
    with torch.no_grad():
        model.inference_padded(x, sizes)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for key, fl, by, e0, e1, shape, ex in prof:
        a = agg.setdefault(key, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, exec=0.0))
        a["ms"] += e0.elapsed_time(e1)
        a["flops"] += fl
        a["bytes"] += by
        a["exec"] += ex
        a["launches"] += 1
    def pipe(k):      # (MFMA products per fp32 product, peak of that pipe): the opt-in split kernels run on the 16-bit matrix instructions
        if k.startswith("conv_sp3") or (k.startswith("conv_pw") and k.endswith(", 2>")):
            return 3, PEAK_16BIT_MATRIX_TFLOPS
        if k.startswith("conv_pw") and k.endswith(", 1>"):
            return 6, PEAK_16BIT_MATRIX_TFLOPS
        return 1, PEAK_F32_MATRIX_TFLOPS
    dom = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[dom]
    dprod, dpeak = pipe(dom)
    total_ms = sum(a["ms"] for a in agg.values())
    khash = ops.kernel_source_hash()
    roof = {
        "bound": "mfma", "kernel": dom, "achieved": round(dprod * d["exec"] / d["ms"] / 1e9, 2), "peak": dpeak,
        "unit": "TFLOP/s", "frac": round(dprod * d["exec"] / d["ms"] / 1e9 / dpeak, 4), "traffic": None, "mfma_busy": None,
        "definition": "achieved = executed MFMA FLOPs (Winograd-domain, tile padding included) / HIP-event time of the kernel's launches",
        "alg_equiv_TFLOPs": round(d["flops"] / d["ms"] / 1e9, 2),
        "launches_per_step": d["launches"], "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2),
        "exec_flops_per_launch": round(d["exec"] / d["launches"] / 1e9, 3), "alg_flops_per_launch": round(d["flops"] / d["launches"] / 1e9, 3),
        "alg_bytes_per_launch": round(d["bytes"] / d["launches"]), "alg_GBps": round(d["bytes"] / d["ms"] / 1e6, 1),
        "hbm_frac": round(d["bytes"] / d["ms"] / 1e6 / PEAK_HBM_GBS, 4), "kernel_source_hash": khash,
        "all_convs": {"ms_per_step": round(total_ms, 3), "exec_TFLOP/s": round(sum(a["exec"] for a in agg.values()) / total_ms / 1e9, 2),
                      "alg_equiv_TFLOPs": round(sum(a["flops"] for a in agg.values()) / total_ms / 1e9, 2),
                      "frac": round(sum(a["exec"] for a in agg.values()) / total_ms / 1e9 / PEAK_F32_MATRIX_TFLOPS, 4),
                      "alg_GBps": round(sum(a["bytes"] for a in agg.values()) / total_ms / 1e6, 1)},
        "per_kernel": {k: {"ms": round(a["ms"], 3), "exec_TFLOP/s": round(pipe(k)[0] * a["exec"] / a["ms"] / 1e9, 1), "frac": round(pipe(k)[0] * a["exec"] / a["ms"] / 1e9 / pipe(k)[1], 3),
                           "alg_equiv_TFLOPs": round(a["flops"] / a["ms"] / 1e9, 1), "launches": a["launches"]}
                       for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])},
    }
    for fname, field in (("pmc_traffic.json", "traffic"), ("pmc_mfma.json", "mfma_busy")):
        path = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_" + fname)
        if not os.path.exists(path):
            continue
        t = json.load(open(path))
        if t.get("_kernel_source_hash") != khash:
            roof[field + "_source"] = "profiles/{}_{} is of kernel sources {} (now {}): not quoted".format(PROFILE_ROUND, fname, t.get("_kernel_source_hash"), khash)
            continue
        e = t.get(dom)
        if not e:
            continue
        if field == "traffic":
            roof["traffic"] = round(e["hbm_bytes_per_launch"])
            roof["traffic_source"] = "profiles/{}_{} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction)".format(PROFILE_ROUND, fname)
        else:
            roof["mfma_busy"] = round(e["mfma_busy_frac"], 4)
            roof["mfma_busy_source"] = "profiles/{}_{} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; busy cycles / (4 SIMDs x busy CU cycles))".format(PROFILE_ROUND, fname)
    # the same kernel's average duration in the rocprofv3 --kernel-trace --stats run kept under profiles/ (quoted while that run was of
    # these kernel sources): the event pair above also times the eager launch (~10 us), the kernel trace does not
    try:
        import csv
        under = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_bench_under_rocprof.json")))
        if under.get("roofline", {}).get("kernel_source_hash") == khash:
            for row in csv.DictReader(open(os.path.join(ROOT, "profiles", PROFILE_ROUND + "_bench_kernel_stats.csv"))):
                if dom + "(" in row["Name"]:
                    us = float(row["AverageNs"]) / 1e3
                    roof["avg_launch_us_rocprof"] = round(us, 2)
                    roof["frac_rocprof"] = round(d["exec"] / d["launches"] / (us * 1e-6) / 1e12 / PEAK_F32_MATRIX_TFLOPS, 4)
                    break
    except Exception:
        pass
    return roof


def host_cores():
    """Threads for the CPU leg: the cgroup CPU quota if there is one, else the affinity mask, capped at the 16-core
    share a one-GPU box grants (CMK_CPU_THREADS overrides)."""
    if os.environ.get("CMK_CPU_THREADS"):
        return int(os.environ["CMK_CPU_THREADS"])
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return min(cores, 16)


def ap_delta_leg(hip_out, oracle_results, height=800, width=1280):
    """The "AP delta" half of the metric on fixed inputs: COCO-style AP@[.5:.95] of the HIP path's detections
    against the pseudo ground truth = the oracle's detections on the same images (whose own AP against itself is 1 by
    construction).  Masks are compared as pasted full-image bitmasks at 0.5; segm ranks by mask_scores."""
    from centermask2_amd import ops
    from centermask2_amd.evaluation import average_precision
    from oracle import centermask_oracle as O
    preds, gts = [], []
    for i, ref in oracle_results:
        k = int(hip_out["counts"][i])
        boxes = hip_out["box"][i, :k]
        pm = ops.paste_masks(hip_out["pred_masks"][i, :k, 0], boxes, height, width, 0.5).cpu()
        preds.append(dict(boxes=boxes.cpu(), classes=hip_out["cls"][i, :k].cpu(), scores=hip_out["score"][i, :k].cpu(),
                          mask_scores=hip_out["mask_scores"][i, :k].cpu(), masks=pm))
        gts.append(dict(boxes=ref["boxes"], classes=ref["classes"], masks=O.paste_masks(ref["pred_masks"][:, 0], ref["boxes"], height, width, 0.5)))
    box_ap, mask_ap = average_precision(preds, gts, "box"), average_precision(preds, gts, "mask")
    return {"box_ap": round(box_ap, 4), "mask_ap": round(mask_ap, 4), "box_ap_delta_vs_ref": round(box_ap - 1.0, 4),
            "mask_ap_delta_vs_ref": round(mask_ap - 1.0, 4), "images": len(gts),
            "ground_truth": "detections of the CPU restatement of the reference on the same synthetic images (no COCO/pycocotools offline)"}


def cpu_baseline_leg(sd, conv_body, budget_s=25.0):
    """The oracle (plain PyTorch CPU ops + C nms/roi_align) on the same seeded workload, bounded to ~budget_s.
    Returns (baseline dict, [(image index, oracle result)]) — the results double as pseudo ground truth for the AP leg."""
    from centermask2_amd import synthetic as S
    from oracle import centermask_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    x = S.make_synthetic_images(1, 800, 1280, seed0=1234)
    results = []
    t0 = time.time()
    results.append((0, O.centermask_inference(sd, x, [(800, 1280)], conv_body)[0]))        # warm-up (also pages the weights in)
    warm = time.time() - t0
    n, t_used = 0, 0.0
    while n < 7 and (n == 0 or t_used + t_used / n < budget_s - warm):
        xi = S.make_synthetic_images(1, 800, 1280, seed0=1234, first=n + 1)
        t0 = time.time()
        results.append((n + 1, O.centermask_inference(sd, xi, [(800, 1280)], conv_body)[0]))
        t_used += time.time() - t0
        n += 1
    return {"value": round(n / t_used, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "{} image(s) 3x800x1280 (batch 1, full model incl. NMS/ROI heads) after 1 warm-up, torch {} CPU fp32".format(n, torch.__version__)}, results


def other_body_leg(body, dev, B, steps=5, warmup=2, keep=None):
    """BASELINE config 5's per-GPU workload (the V-99-eSE body, 8 x 3x800x1280) through the same graph-replay step, a bounded number of
    steps — an extra key of the default run, so that the driver's bench record carries it; never `value`."""
    from centermask2_amd import ops, synthetic as S
    saved = dict(ops._TUNED)
    try:
        ops._TUNED.clear()
        table = os.path.join(ROOT, "centermask2_amd", "tuned", "mi355x_{}_b{}_800x1280.json".format(body, B))
        n_loaded = ops.load_tuned(table) if os.path.exists(table) else 0
        model, _ = build(body, dev)
        x = S.make_synthetic_images(B, 800, 1280, seed0=1234).to(dev)
        sizes = [(800, 1280)] * B
        with torch.no_grad():
            model.inference_padded(x, sizes)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.inference_padded(x, sizes)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model.inference_padded(x, sizes)
            for _ in range(warmup):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        assert not bool(out["overflow"].any())
        if keep is not None:                        # the detections of this body on the fp32 path: what its opt-in legs are compared with
            keep.update({k: out[k].clone() for k in ("counts", "cls", "score")})
        return {"images_per_sec": round(steps * B / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup,
                "workload": "Full CenterMask2 {} , bs={}, 3x800x1280, hip-graph (BASELINE configs[4] per-GPU share)".format(body, B),
                "conv_variants_loaded": n_loaded, "detections_per_image": out["counts"].cpu().tolist()}
    finally:
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def split_gemm_leg(body, dev, B, ref_out=None, steps=20, warmup=3, level=1):
    """OPT-IN path, reported beside `value`, never as it: the same step with the convs the pointwise GEMM kernel runs (OSA aggregation convs, FPN
    laterals, the mask head's deconv, stem_3 in the gather form) on its bf16-split form (cmk.h tune_wm 10: every fp32 product rebuilt from three bf16 pieces per operand, six MFMA
    products, fp32 accumulation — the error of an fp32 accumulation).  The detections are compared with the default
    path's (`ref_out`): same counts, same labels in the same order, scores within 1e-4.
    level 2 (key `split_direct3x3`): additionally the 3x3 convs the measured table `tuned/..._split3.json` names run as direct implicit GEMMs on TWO
    fp16 pieces per fp32 operand (cmk.h tune_wm 11, conv_sp3.hip: 22-bit operands, three products, fp32 accumulation: an fp32 accumulation's error)."""
    from centermask2_amd import ops, synthetic as S
    saved, saved_flag, saved_flag3 = dict(ops._TUNED), ops.ALLOW_SPLIT_BF16, ops.ALLOW_SPLIT_F16
    try:
        ops.ALLOW_SPLIT_BF16 = True
        ops.ALLOW_SPLIT_F16 = level >= 2
        ops._TUNED.clear()
        table = os.path.join(ROOT, "centermask2_amd", "tuned", "mi355x_{}_b{}_800x1280{}.json".format(body, B, "_split3" if level >= 2 else ""))
        if os.path.exists(table):
            ops.load_tuned(table)
        elif level >= 2:
            return {"status": "no measured table for this body"}
        moved = 0
        for k, v in list(ops._TUNED.items()):
            # every conv the pointwise GEMM kernel runs without split-K: plain 1x1 (8), 1x1 with the FPN top-down add (8, res 2), 3x3 in the gather form (9)
            if len(v) == 3 and ((k[0] == 1 and v[0] == 8 and k[6] in (0, 2)) or (k[0] == 3 and v[0] == 9 and k[6] == 0)):
                ops._TUNED[k] = (10, 32, 4)
                moved += 1
        model, _ = build(body, dev)                       # packs the split weights (ALLOW_SPLIT_BF16 is on)
        x = S.make_synthetic_images(B, 800, 1280, seed0=1234).to(dev)
        sizes = [(800, 1280)] * B
        with torch.no_grad():
            model.inference_padded(x, sizes)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.inference_padded(x, sizes)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = model.inference_padded(x, sizes)
            for _ in range(warmup):
                graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                graph.replay()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        same_counts = same_labels = score_diff = None
        if ref_out is not None:
            same_counts = bool(torch.equal(out["counts"], ref_out["counts"]))
            same_labels = bool(torch.equal(out["cls"], ref_out["cls"]))
            score_diff = float((out["score"] - ref_out["score"]).abs().max())
        moved = sum(1 for v in ops._TUNED.values() if v[0] in (10, 12))
        # one instrumented eager step: what the split kernels reach on the 16-bit matrix pipe (products per fp32 product: 3 on fp16 pieces, 6 on bf16)
        ops.PROFILE = []
        with torch.no_grad():
            model.inference_padded(x, sizes)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
        agg = {}
        for key, fl, by, e0, e1, shape, ex in prof:
            a = agg.setdefault(key, dict(ms=0.0, flops=0.0, exec=0.0, launches=0))
            a["ms"] += e0.elapsed_time(e1); a["flops"] += fl; a["exec"] += ex; a["launches"] += 1
        kern = {}
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            products = 3 if (k.startswith("conv_sp3") or k.endswith(", 2>")) else 6 if (k.startswith("conv_pw") and k.endswith(", 1>")) else 0
            if products:
                t16 = products * a["exec"] / a["ms"] / 1e9
                kern[k] = {"ms": round(a["ms"], 3), "launches": a["launches"], "alg_equiv_TFLOPs": round(a["flops"] / a["ms"] / 1e9, 1), "products_per_fp32_product": products,
                           "mfma16_TFLOPs_executed": round(t16, 1), "frac_of_measured_16bit_peak": round(t16 / PEAK_16BIT_MATRIX_TFLOPS, 3)}
        conv_ms = sum(a["ms"] for a in agg.values())
        res = {"images_per_sec": round(steps * B / dt, 2), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps, "warmup": warmup, "body": body,
               "convs_moved": moved, "dtype": "f32 results; the moved convs multiply bf16 pieces (3 per fp32 operand, 6 products per fp32 product) and accumulate in f32",
               "same_detection_counts": same_counts, "same_labels_same_order": same_labels, "max_score_diff_vs_default": score_diff,
               "status": "opt-in (CMK_ALLOW_SPLIT_BF16=1 + a variant table naming tune 10/32/4); not used by `value`"}
        res["split_kernels"] = kern
        res["split_kernels_peak"] = "{} TFLOP/s: v_mfma_f32_32x32x16_bf16/f16 measured on random operands (tools/probe/mfma_power_bf16.hip); nominal dense 2516".format(PEAK_16BIT_MATRIX_TFLOPS)
        res["all_convs_ms_instrumented"] = round(conv_ms, 3)
        if level >= 2:
            res["convs_3x3_moved"] = sum(1 for v in ops._TUNED.values() if v[0] == 11)
            res["dtype"] = ("f32 results; the convs named by the table (3x3 direct form, pointwise GEMMs) multiply 2 fp16 pieces per fp32 operand (22-bit operands, "
                            "3 products per fp32 product); f32 accumulation")
            res["status"] = "opt-in, second level (CMK_ALLOW_SPLIT_F16=1 + tuned/*_split3.json); not used by `value`"
        return res
    finally:
        ops.ALLOW_SPLIT_BF16 = saved_flag
        ops.ALLOW_SPLIT_F16 = saved_flag3
        ops._TUNED.clear()
        ops._TUNED.update(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step")
    ap.add_argument("--body", default="V-39-eSE")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured HIP graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the sustained (>= 5 s) and host-fed runs that follow the timed region")
    ap.add_argument("--no-autotune", action="store_true", help="use the library's cost model instead of the measured conv tile-variant table")
    ap.add_argument("--tune-file", default=None, help="conv variant table (default: centermask2_amd/tuned/mi355x_<body>_b<B>_800x1280.json)")
    ap.add_argument("--save-tuned", action="store_true", help="write the variant table after start-up tuning")
    ap.add_argument("--path", choices=("fp32", "split"), default="fp32",
                    help="fp32 (default): every conv on the fp32 matrix instruction.  split: the OPT-IN kernels — fp32 results from two fp16 pieces per operand on the "
                         "16-bit matrix instructions, table tuned/*_split3.json (DESIGN section 3 'Split products'); `dtype` and `config.path` of the line say so")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # one process per GPU.  CMK_DIST_BACKEND=gloo + CMK_SHARE_GPU=1 exist only to rehearse the N>1 code path on a one-GPU box
    # (every rank on cuda:0, gloo moving the records); the real runs use nccl = RCCL over xGMI, one device per rank.
    backend = os.environ.get("CMK_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("CMK_SHARE_GPU") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from centermask2_amd import synthetic as S
    from centermask2_amd import ops
    if args.path == "split":                       # before build(): the split weight copies are packed with the model
        ops.ALLOW_SPLIT_BF16 = ops.ALLOW_SPLIT_F16 = True
    model, sd = build(args.body, dev)
    B = args.batch
    x = S.make_synthetic_images(B, 800, 1280, seed0=1234, first=rank * B).to(dev)   # resident in HBM before timing
    sizes = [(800, 1280)] * B

    exchange = RecordExchange(world)

    def step_eager():
        out = model.inference_padded(x, sizes)
        rec = pack_results(out)
        exchange(rec)
        return out, rec

    use_graph = not args.no_graph
    graph = None
    from centermask2_amd import ops
    tune_file = args.tune_file or os.path.join(ROOT, "centermask2_amd", "tuned", "mi355x_{}_b{}_800x1280{}.json".format(args.body, B, "_split3" if args.path == "split" else ""))
    n_loaded = ops.load_tuned(tune_file) if (os.path.exists(tune_file) and not args.no_autotune) else 0
    with torch.no_grad():
        ops.set_autotune(not args.no_autotune)   # first call: packs weights, sets kernel attributes; conv problems missing from the
        out, rec = step_eager()                  # shipped variant table are timed here (outside the timed region)
        ops.set_autotune(False)
        torch.cuda.synchronize()
        n_tuned = len(ops.tuned_variants())
        if args.save_tuned and rank == 0:
            os.makedirs(os.path.dirname(tune_file), exist_ok=True)
            ops.save_tuned(tune_file)
        if use_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model.inference_padded(x, sizes)
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    g_out = model.inference_padded(x, sizes)
                    g_rec = pack_results(g_out)
                torch.cuda.synchronize()
            except Exception as e:      # capture is an optimisation; the eager path is the same kernels
                if rank == 0:
                    print("graph capture failed ({}: {}); running eagerly".format(type(e).__name__, str(e)[:200]), file=sys.stderr)
                graph = None

        def step():
            if graph is None:
                return step_eager()
            graph.replay()
            exchange(g_rec)                      # the all-gather stays outside the captured graph
            return g_out, g_rec

        elapsed, (out, rec) = timed_steps(step, args.steps, args.warmup, world, dev)

        # ---- beyond the contract's K steps (extra keys, never `value`): a sustained run and a run fed from the host -------------
        sustained = fed = plugin_api = None
        if rank == 0 and world == 1 and not args.no_extras:
            # (1) >= 5 s of back-to-back steps: the K-step window above is well under a second, too short for the chip to settle
            # into the clock it holds under this load
            n_s, t0s = 0, time.perf_counter()
            while True:
                for _ in range(10):
                    step()
                n_s += 10
                torch.cuda.synchronize()
                if time.perf_counter() - t0s >= 5.0:
                    break
            t_s = time.perf_counter() - t0s
            sustained = {"images_per_sec": round(n_s * B / t_s, 2), "seconds": round(t_s, 2), "steps": n_s}
            # (2) fresh host batches: pinned host memory -> a staging buffer on a copy stream while the previous batch computes,
            # then one device-to-device copy into the buffer the (captured) step reads.  98 MB per batch over PCIe.
            host = [S.make_synthetic_images(B, 800, 1280, seed0=1234, first=0).pin_memory() for _ in range(2)]
            stage = [torch.empty_like(x) for _ in range(2)]
            copy_stream = torch.cuda.Stream()
            ready = [torch.cuda.Event() for _ in range(2)]
            consumed = [torch.cuda.Event() for _ in range(2)]
            main = torch.cuda.current_stream()

            def upload(i):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(consumed[i & 1])
                    stage[i & 1].copy_(host[i & 1], non_blocking=True)
                    ready[i & 1].record(copy_stream)
            for ev in consumed:
                ev.record(main)
            n_f = 40
            upload(0)
            torch.cuda.synchronize()
            t0f = time.perf_counter()
            for i in range(n_f):
                if i + 1 < n_f:
                    upload(i + 1)
                main.wait_event(ready[i & 1])
                x.copy_(stage[i & 1], non_blocking=True)
                consumed[i & 1].record(main)
                step()
            torch.cuda.synchronize()
            t_f = time.perf_counter() - t0f
            fed = {"images_per_sec": round(n_f * B / t_f, 2), "steps": n_f, "h2d_bytes_per_step": int(x.numel() * 4),
                   "how": "pinned host batch -> staging buffer on a copy stream (overlaps the previous step) -> D2D into the step's input"}
            # (3) the drop-in plugin API as the reference's callers drive it (tester.py:94-104, modified_class.py:35-38):
            # backbone -> proposal_generator(images, features, None) -> roi_heads(images, features, proposals, None), launched eagerly.
            # The Instances FCOS hands to the ROI heads are lazy (nothing is read back between the two plugins); "unread" leaves them
            # so, like `value` leaves its padded buffers; "read" reads every image's fields after each call (the host sync a caller
            # that looks at the results pays).  eager_padded = inference_padded without the graph, for the gap to `value`.
            from centermask2_amd.structures import FakeImageList
            images = FakeImageList(x, sizes)

            def rate(fn, n_it=20):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                t0p = time.perf_counter()
                for _ in range(n_it):
                    fn()
                torch.cuda.synchronize()
                return round(n_it * B / (time.perf_counter() - t0p), 2)

            def api_read():
                for inst in model.inference(images, do_preprocess=False, do_postprocess=False):
                    len(inst)
            plugin_api = {"images_per_sec": rate(lambda: model.inference(images, do_preprocess=False, do_postprocess=False)),
                          "images_per_sec_results_read": rate(api_read),
                          "eager_padded_images_per_sec": rate(lambda: model.inference_padded(x, sizes)),
                          "call": "model.inference(FakeImageList(x, sizes), do_preprocess=False, do_postprocess=False), eager launches, bs {}".format(B)}

        cand = out["cand_counts"].cpu().tolist()
        assert not bool(out["overflow"].any()), "candidate overflow {} > capacity {} inside the timed region".format(max(cand), out["cand_capacity"])
        dets = out["counts"].cpu().tolist()

        result = None
        if rank == 0:
            roof = roofline_leg(model, x, sizes)
            cpu, ap = None, None
            if not (args.no_cpu_baseline or world > 1):
                cpu, oracle_results = cpu_baseline_leg(sd, args.body)
                ap = ap_delta_leg(out, [r for r in oracle_results if r[0] < B])
            total_images = world * B * args.steps
            result = {
                "metric": "images/sec whole-node (V2-39-eSE 3x800x1280)" if args.body == "V-39-eSE" else "images/sec whole-node ({} 3x800x1280)".format(args.body),
                "value": round(total_images / elapsed, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32" if args.path == "fp32" else "f32 results from 2 fp16 pieces per fp32 operand (22-bit operands, 3 products per fp32 product, f32 accumulation) on the convs the table names",
                "data": "synthetic",
                "config": {"workload": "Full CenterMask2 {} (VoVNetV2-FPN + FCOS + CenterROIHeads/SAG-Mask/MaskIoU + ROIAlignV2), bs={} per GPU, "
                                       "3x800x1280, end-to-end (BASELINE configs[3])".format(args.body, B),
                           "global_batch": world * B, "per_gpu_batch": B, "parallelism": "dp{}".format(world), "path": args.path,
                           "launch": "hip-graph" if graph is not None else "eager",
                           "conv_variants": "measured table: {} problems loaded from {}, {} timed at start-up".format(
                               n_loaded, os.path.relpath(tune_file, ROOT), n_tuned - n_loaded) if not args.no_autotune else "library cost model",
                           "collective": collective_description(rec.shape[1], world),
                           "candidates_per_image": cand, "detections_per_image": dets,
                           "weights": "seeded random-init, reference state-dict keys"},
                "roofline": roof,
            }
            if sustained is not None:
                result["sustained"] = sustained
                result["fed"] = fed
                result["plugin_api"] = plugin_api
            if args.body == "V-39-eSE" and world == 1 and not args.no_extras and args.path == "fp32":
                v99_out = {}
                result["v99"] = other_body_leg("V-99-eSE", dev, B, keep=v99_out)
            if world == 1 and not args.no_extras and args.path == "fp32":      # (--path split IS the opt-in path: nothing to report beside it)
                result["split_gemm"] = split_gemm_leg(args.body, dev, B, out)
                result["split_direct3x3"] = split_gemm_leg(args.body, dev, B, out, level=2)
                if args.body == "V-39-eSE":
                    result["split_gemm_v99"] = split_gemm_leg("V-99-eSE", dev, B, v99_out or None, steps=5, warmup=2)
                    result["split_direct3x3_v99"] = split_gemm_leg("V-99-eSE", dev, B, v99_out or None, steps=5, warmup=2, level=2)
            if cpu is not None:
                result["cpu_baseline"] = cpu
                result["ap_delta"] = ap
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
