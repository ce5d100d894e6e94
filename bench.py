#!/usr/bin/env python3
"""Headline benchmark: training images/sec on 512x384 fake-ESPI frames, Xception backbone
(BASELINE.json configs[1]: batch 32 per MI355X; weak scaling over N GPUs with one RCCL gradient
all-reduce per step).

  python bench.py --gpus 1 --steps 10 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One timed step = draw augmentation parameters (host, reference RNG order) -> cutout + salt&pepper on
device -> forward (batch-statistics BN, dropout) -> custom_loss -> backward -> [all-reduce] ->
fused Adam + l2 with the 1-cycle learning rate of that iteration.  Frames and labels are resident in
HBM before the timed region starts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, BATCH = 384, 512, 32
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA (the figure without 2:1 sparsity)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec
DW_TRAIN_BYTES_PER_IMAGE = 157.6e6  # depthwise stack, fwd + bwd, 384x512
DW_FWD_BYTES_PER_IMAGE = 63.1e6


# The arithmetic the path computes in (not a precision claim): fp32 tensors and accumulators everywhere; since round 4 the
# forward and data-gradient GEMMs of the pointwise convolutions with >= 256 output columns multiply on the bf16 matrix
# cores with every fp32 operand split exactly into three bf16 pieces (six bf16 MFMAs per product block, fp32 accumulate:
# error against float64 no larger than the fp32 fmaf chain's, tests/test_kernels_gpu.py); all other GEMMs, the weight
# gradients included, are fp32 MFMA chains.  `roofline_alt.train_step` carries the all-fp32-chain step beside it.
DTYPE = "f32 (bf16x3 MFMA for pointwise fwd/dgrad/wgrad GEMMs)"
ARITHMETIC = ("fp32 tensors, accumulators and results; the forward, data-gradient and weight-gradient GEMMs of the pointwise "
              "convolutions with >= 256 input and output channels: fp32 operands split exactly into three bf16 pieces (by the "
              "kernels that produce them), six bf16 MFMAs per product block, fp32 accumulation (error vs float64 no larger "
              "than the fp32 fmaf chain's); every other GEMM: fp32 MFMA chains; Engine(pointwise='f32') / "
              "cf.pointwise_gemm = 'f32' builds the all-chain plan (roofline_alt.train_step times it in the same run)")
X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0      # fp32-equivalent: six bf16 MFMAs per fp32 product block
X3_REAL_OPERAND_TFLOPS = 1247.0 / 6.0             # MI355X_MICROARCH.md, DVFS give-back: its tuned bf16 GEMM on random operands

# BASELINE.json's metric string, verbatim
METRIC = "training images/sec on 512\u00d7384 fake-ESPI, Xception backbone, 1/2/4/8 GPU"


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "hbm_traffic_current.json")


def kernel_source_hash():
    """Fingerprint of everything that decides which kernels run and how they touch memory: the HIP sources and the
    launch plan.  (The GPU box has no .git, so a commit id is not available at run time; the PMC file is stamped
    with this hash by tools/pmc_traffic.py and with the commit it was taken at.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "spnet_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "spnet_amd", "csrc", "*.h")) +
                   [os.path.join(ROOT, "spnet_amd", "engine.py"), os.path.join(ROOT, "spnet_amd", "gemm_tiles.json"),
                    os.path.join(ROOT, "include", "spnet_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def measured_traffic():
    """HBM bytes per family from the committed PMC pass (profiles/hbm_traffic_current.json, folded by
    tools/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this same command,
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC passes cannot run inside the timed
    benchmark, so the file is only used when it was taken with EXACTLY the kernels and launch plan being timed
    (same kernel_source_hash); otherwise traffic is reported as null."""
    if not os.path.exists(TRAFFIC_FILE):
        return {}, "no PMC file"
    with open(TRAFFIC_FILE) as f:
        tr = json.load(f)
    want = kernel_source_hash()
    if tr.get("kernel_source_hash") != want:
        return {}, "PMC file is stale: taken at kernel_source_hash %s, running %s" % (tr.get("kernel_source_hash"), want)
    return tr, "PMC passes at kernel_source_hash %s (commit %s)" % (want, tr.get("commit", "?"))


class FamilyTimes:
    """Sum of engine.KernelTimer readings over several steps: totals() / tagged() like the timer itself."""

    def __init__(self):
        self.tot, self.tag = {}, {}

    def add(self, t):
        for dst, src in ((self.tot, t.totals()), (self.tag, t.tagged())):
            for k, (n, ms, w) in src.items():
                a = dst.setdefault(k, [0, 0.0, 0.0])
                a[0] += n
                a[1] += ms
                a[2] += w

    def totals(self):
        return {k: tuple(v) for k, v in self.tot.items()}

    def tagged(self):
        return {k: tuple(v) for k, v in self.tag.items()}


def time_families(eng, step, steps, sync, cushions=2):
    """HIP events around every launch of the timed families during `steps` steps -> FamilyTimes.

    An event is stamped when the GPU REACHES it: with an empty queue the start event of a launch is stamped at once and
    the launch behind it arrives a host gap later, so a pass whose host enqueue falls behind the GPU reads too long (the
    GEMM family read 14.3 instead of 8.6 ms per step in one run, profiles/r03_g_bench.json; pass after pass of thousands
    of live events made it likelier).  Every timed step is therefore enqueued behind `cushions` untimed steps -- GPU work
    that the host needs a fraction of the time to enqueue -- and its events are read and dropped before the next one."""
    from spnet_amd.engine import KernelTimer
    agg = FamilyTimes()
    for _ in range(steps):
        eng.prof = None
        for _ in range(cushions):
            step()
        tp = KernelTimer()
        eng.prof = tp
        step()
        eng.prof = None
        sync()
        agg.add(tp)
    return agg


GEMM_KERNELS = (        # (kernel, tag prefix of Engine's KernelTimer, peak TFLOP/s, what it runs); first match wins
    ("gemm_bf16x3_pp_dwbwd_kernel<0>", "x3p ab+dwbwd", X3_PEAK_TFLOPS,
     "data-gradient GEMM on 192x96 tiles with the depthwise backward as its epilogue (FLOPs: the GEMM's alone; the time also "
     "holds the epilogue's one pass over x / dx, 8-16 B per element of HBM traffic)"),
    ("gemm_bf16x3_pp_dwbwd_kernel<1>", "x3p aB+dwfwd", X3_PEAK_TFLOPS,
     "inference forward GEMM on 192x96 tiles with BatchNorm affine + ReLU + the next layer's depthwise forward as its epilogue "
     "(FLOPs: the GEMM's alone)"),
    ("gemm_bf16x3_pp_kernel", "x3p ", X3_PEAK_TFLOPS, "pointwise forward (+ BatchNorm sums) and data gradient from bf16 planes"),
    ("gemm_bf16x3_wgrad_kernel", "x3w ", X3_PEAK_TFLOPS, "pointwise weight gradients from the same planes (transposing LDS reads)"),
    ("gemm_bf16x3_fwd_kernel", "x3 ", X3_PEAK_TFLOPS, "pointwise forward on an fp32 A operand split in the kernel (strided residual convolutions)"),
    ("fp32 MFMA kernels (gemm_f32_kernel, conv3x3_*_kernel)", None, FP32_MFMA_PEAK_TFLOPS,
     "entry-flow layers below 256 channels incl. the blended data gradients, block1_conv2, Dense head, their weight gradients"),
)


def gemm_family_by_kernel(tim, steps):
    """The GEMM family of a KernelTimer / FamilyTimes reading split by kernel (tag prefix), each against ITS OWN peak:
    {kernel: {launches_per_step, ms_per_step, achieved, peak, frac, ...}}, the dominant kernel's name, and the family's
    time-weighted fraction sum_k(flops_k / peak_k) / sum_k(time_k)."""
    tot_n, tot_ms, tot_fl = tim.totals()["gemm"]
    rows, seen, taken = {}, [0, 0.0, 0.0], set()
    for kern, prefix, peak, what in GEMM_KERNELS:
        alg_bytes = None
        if prefix is None:
            n, ms, fl = tot_n - seen[0], tot_ms - seen[1], tot_fl - seen[2]
        else:
            hit = [t for t in tim.tagged() if isinstance(t, tuple) and str(t[0]).startswith(prefix) and t not in taken]
            taken.update(hit)
            sel = [tim.tagged()[t] for t in hit]
            # HBM bytes the launches have to move at the least (planes 6 B per operand element, fp32 results): what the
            # PMC figure `traffic` is to be read against
            if kern == "gemm_bf16x3_pp_kernel":
                alg_bytes = sum(tim.tagged()[t][0] * (6.0 * (t[1] + t[2]) * t[3] + 4.0 * t[1] * t[2]) for t in hit)
            elif kern == "gemm_bf16x3_wgrad_kernel":          # tag: (name "... xN batched", cin, cout, M)
                nb = lambda t: int(str(t[0]).split(" x")[1].split()[0]) if " x" in str(t[0]) else 1
                alg_bytes = sum(tim.tagged()[t][0] * nb(t) * (6.0 * t[3] * (t[1] + t[2]) + 4.0 * t[1] * t[2]) for t in hit)
            else:
                alg_bytes = None
            n, ms, fl = (sum(v[i] for v in sel) for i in range(3))
            seen = [seen[0] + n, seen[1] + ms, seen[2] + fl]
        if n <= 0:
            continue
        tf = fl / max(ms, 1e-9) / 1e9
        rows[kern] = {"runs": what, "launches_per_step": n / steps, "ms_per_step": round(ms / steps, 3),
                      "avg_launch_us": round(1e3 * ms / n, 2), "algorithmic_flops_per_launch": round(fl / n),
                      "achieved": round(tf, 1), "peak": round(peak, 1), "unit": "TFLOP/s" + (" (fp32-equivalent)" if prefix else ""),
                      "frac": round(tf / peak, 4)}
        if alg_bytes:
            rows[kern]["algorithmic_bytes_per_launch"] = round(alg_bytes / n)
    ideal_ms = sum(r["algorithmic_flops_per_launch"] * r["launches_per_step"] / (r["peak"] * 1e9) for r in rows.values())
    fam_ms = sum(r["ms_per_step"] for r in rows.values())
    # the dominant KERNEL: a single kernel, not the group row (a dozen gemm_f32_kernel<...> instantiations and the three
    # conv3x3 kernels, the largest of them 0.27 ms per step in profiles/r05_*_step_table.txt)
    single = [k for k, pre, _, _ in GEMM_KERNELS if pre is not None and k in rows]
    dominant = max(single or rows, key=lambda k: rows[k]["ms_per_step"])
    return rows, dominant, round(ideal_ms / max(fam_ms, 1e-9), 4), (tot_n, tot_ms, tot_fl)


def gemm_roofline_block(tim, steps, traffic=None):
    """`roofline` for a GEMM-dominated plan: the dominant KERNEL against its own peak (recomputable from the committed
    rocprofv3 kernel stats: that kernel's calls x average duration), the family beside it."""
    rows, dom, fam_frac, (g_n, g_ms, g_fl) = gemm_family_by_kernel(tim, steps)
    d = rows[dom]
    fam_tf = g_fl / max(g_ms, 1e-9) / 1e9
    blk = {"kernel": "%s: %s" % (dom, d["runs"]), "bound": "mfma", "achieved": d["achieved"], "peak": d["peak"],
           "unit": d["unit"], "frac": d["frac"], "traffic": traffic,
           "algorithmic_flops_per_launch": d["algorithmic_flops_per_launch"], "avg_launch_us": d["avg_launch_us"],
           "launches_per_step": d["launches_per_step"], "ms_per_step": d["ms_per_step"],
           "peak_note": "dense bf16 MFMA peak 2500 TFLOP/s / 6 MFMAs per fp32 product block" if dom.startswith("gemm_bf16x3") else
                        "fp32 MFMA peak (v_mfma_f32_*_f32 at the vector rate)",
           "algorithmic_bytes_per_launch": d.get("algorithmic_bytes_per_launch"),
           "by_kernel": rows,
           "family": {"kernels": list(rows), "launches_per_step": g_n / steps, "ms_per_step": round(g_ms / steps, 3),
                      "frac_time_weighted": fam_frac,
                      "frac_time_weighted_note": "sum over kernels of (algorithmic FLOPs / that kernel's own peak) / family time",
                      "achieved_fp32_equivalent_tflops": round(fam_tf, 2),
                      "frac_vs_fp32_peak_legacy": round(fam_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                      "legacy_note": "family FLOPs / family time / 157.3: the yardstick of rounds 1-4, kept only so that the "
                                     "rounds compare -- most of these FLOPs run on the bf16 pipe, whose peak is 2.65x higher"}}
    if dom.startswith("gemm_bf16x3"):
        blk["frac_vs_real_operand_bf16_rate"] = round(d["achieved"] / X3_REAL_OPERAND_TFLOPS, 4)
        blk["real_operand_note"] = ("MI355X_MICROARCH.md (DVFS give-back) quotes its own tuned bf16 GEMM at 1247 TFLOP/s on random "
                                    "operands (1483 on zeros): / 6 = 207.8 fp32-equivalent is what the power limit leaves")
    return blk


FAMILY_OF = (("spnet_bn_", "bn"), ("spnet_maxpool", "pool"), ("spnet_avgpool", "pool"), ("spnet_gather_s2", "pool"),
             ("spnet_scatter_add_s2", "pool"), ("spnet_stem_head", "stem"), ("spnet_conv3x3_small", "stem"),
             ("spnet_dropout", "stem"), ("spnet_adam_step", "optimizer"), ("spnet_adam_part", "optimizer"),
             ("spnet_adam_l2_sum", "optimizer"), ("spnet_dwconv", "dw"), ("spnet_gemm", "gemm"),
             ("spnet_conv3x3_", "gemm"), ("spnet_reduce_slabs", "gemm"), ("spnet_split_bf16x3", "gemm"),
             ("spnet_transpose_batched", "gemm"), ("spnet_cutout", "augment"), ("spnet_saltpepper", "augment"),
             ("spnet_minmax", "augment"), ("spnet_gather_rows", "augment"), ("spnet_ellipse_loss", "loss"))


def algorithmic_bytes(name, a):
    """HBM bytes an entry point has to move at the least -- every operand tensor read once, every result written once, in
    the element sizes the plan really uses (fp32; 6 bytes per element for a bf16x3 planes output) -- or None where no
    model is written down (those launches count in `unmodelled_ms_per_step`).  SURVEY.md section 8(d): BatchNorm, pooling,
    the stem and the optimizer are HBM-bound; argument positions: include/spnet_hip.h."""
    f4 = 4
    if name in ("spnet_bn_bwd_from_partials", "spnet_bn_bwd", "spnet_bn_bwd_from_partials_x3", "spnet_bn_bwd_x3"):
        return a[2] * a[3] * (2 * f4 + (6 if name.endswith("_x3") else f4))          # read x, dy; write dx
    if name in ("spnet_bn_finalize_apply", "spnet_bn_finalize_apply_ld"):
        return a[3] * a[4] * f4 * (2 + (1 if a[13] else 0))                            # read x (+ residual); write y
    if name == "spnet_bn_apply":
        return a[1] * a[2] * f4 * (2 + (1 if (a[5] and not a[6]) else 0))
    if name in ("spnet_bn_fwd_train", "spnet_bn_fwd_train_ld"):
        return a[1] * a[2] * f4 * (2 + (1 if (a[11] and not a[12]) else 0))
    if name in ("spnet_bn_fwd_infer", "spnet_bn_fwd_infer_ld"):
        return a[1] * a[2] * f4 * (2 + (1 if (a[9] and not a[10]) else 0))
    if name == "spnet_bn_finalize_fwd":
        return a[1] * 2 * a[3] * f4
    if name == "spnet_bn_bwd_coeffs_from_partials":
        return a[0] * 2 * a[3] * f4
    if name == "spnet_bn_bwd_coeffs":
        return a[2] * a[3] * 2 * f4
    if name == "spnet_bn_infer_coeffs":
        return a[0] * 6 * f4
    if name == "spnet_maxpool3x3s2_add_fwd":
        B, Hh, Ww, C = a[4:8]
        o = B * ((Hh + 1) // 2) * ((Ww + 1) // 2) * C
        return f4 * (B * Hh * Ww * C + 2 * o) + (o if a[3] else 0)                    # x; residual + y; byte argmax per 4 channels x 4
    if name in ("spnet_maxpool3x3s2_bwd", "spnet_maxpool3x3s2_bwd_bnsums"):
        B, Hh, Ww, C = a[3:7]
        o = B * ((Hh + 1) // 2) * ((Ww + 1) // 2) * C
        return f4 * (o + B * Hh * Ww * C * (2 if name.endswith("bnsums") else 1)) + o  # g, idx; write dx (+ read yp)
    if name in ("spnet_gather_s2", "spnet_scatter_add_s2"):
        B, Hh, Ww, C = a[2:6]
        o = B * ((Hh + 1) // 2) * ((Ww + 1) // 2) * C
        return f4 * o * (2 if name == "spnet_gather_s2" else 3)
    if name in ("spnet_adam_step", "spnet_adam_part"):
        return 28 * a[4]                                                                # w, g, m, v read; w, m, v written
    if name == "spnet_adam_l2_sum":
        return 0
    if name == "spnet_dropout":
        return 2 * f4 * a[2]
    if name == "spnet_conv3x3_small":
        op, cin, cout, stride, same = a[0:5]
        B, Hh, Ww = a[8:11]
        oh = (Hh + stride - 1) // stride if same else (Hh - 3) // stride + 1
        ow = (Ww + stride - 1) // stride if same else (Ww - 3) // stride + 1
        return f4 * (B * Hh * Ww * cin + B * oh * ow * cout)
    if name == "spnet_stem_head":
        B, Hh, Ww = a[5:8]
        return f4 * (B * Hh * Ww + B * (Hh // 2) * (Ww // 2) * (4 if a[0] == 0 else 3))
    return None


def time_entry_points(eng, step, steps, sync, cushions=2):
    """HIP events around EVERY launching entry point of the C ABI during `steps` steps (spnet_amd._lib.set_tracer), one
    stream: {entry: [calls, ms, algorithmic bytes | None]}.  Same discipline as time_families: every timed step is
    enqueued behind untimed ones, its events are read and dropped before the next."""
    import torch
    from spnet_amd import _lib as L
    acc = {}
    for _ in range(steps):
        for _ in range(cushions):
            step()
        rec = []

        def tracer(name, args, tok):
            if tok is None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                return e0
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            rec.append((name, args, tok, e1))

        L.set_tracer(tracer)
        try:
            step()
        finally:
            L.set_tracer(None)
        sync()
        for name, args, e0, e1 in rec:
            b = algorithmic_bytes(name, args)
            a = acc.setdefault(name, [0, 0.0, 0.0, 0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1)
            if b is None:
                a[3] += 1
            else:
                a[2] += b
    return acc


def other_family_rooflines(acc, steps, skip=("gemm", "dw")):
    """`roofline_other`: the HBM-bound families that are neither GEMMs nor depthwise layers (BatchNorm passes, pooling, the
    stem, the optimizer, augmentation, loss), algorithmic bytes / HIP-event time against the 8 TB/s peak."""
    fams = {}
    for name, (n, ms, by, unm) in acc.items():
        fam = next((f for pre, f in FAMILY_OF if name.startswith(pre)), "other")
        if fam in skip:
            continue
        d = fams.setdefault(fam, {"n": 0, "ms": 0.0, "bytes": 0.0, "unmodelled_ms": 0.0, "entries": {}})
        d["n"] += n
        d["ms"] += ms
        if unm:
            d["unmodelled_ms"] += ms
        else:
            d["bytes"] += by
        d["entries"][name] = [n, ms, by, unm]
    out = {}
    for fam, d in sorted(fams.items(), key=lambda kv: -kv[1]["ms"]):
        mod_ms = d["ms"] - d["unmodelled_ms"]
        gbs = d["bytes"] / max(mod_ms, 1e-9) / 1e6 if mod_ms > 0 else None
        top = sorted(d["entries"].items(), key=lambda kv: -kv[1][1])[:4]
        out[fam] = {"bound": "hbm", "launching_calls_per_step": d["n"] / steps, "ms_per_step": round(d["ms"] / steps, 3),
                    "algorithmic_bytes_per_step": round(d["bytes"] / steps),
                    "achieved": None if gbs is None else round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if gbs is None else round(gbs / HBM_PEAK_GBS, 4),
                    "unmodelled_ms_per_step": round(d["unmodelled_ms"] / steps, 3),
                    "largest_entries": {k: {"calls_per_step": v[0] / steps, "ms_per_step": round(v[1] / steps, 3),
                                            "achieved_gbs": None if v[3] or v[1] <= 0 else round(v[2] / v[1] / 1e6, 1)}
                                        for k, v in top}}
    return out


def labels_to_Y(label_rows):
    """Generator rows (cx,cy,a,b,angle,rings) -> normalised grid targets [n,576] (utils.py:260-320 path)."""
    from spnet_amd import utils as U
    Y = np.zeros((len(label_rows), 576), np.float32)
    for i, rows in enumerate(label_rows):
        arr = []
        for cx, cy, a, b, ang, rings in rows:
            if b > a:
                a, b, ang = b, a, ang + 90
            if rings > 0:
                t = 2 * np.deg2rad(ang)
                arr.append([cx, cy, a, b, np.cos(t), np.sin(t), 0, rings])
        arr.sort(key=lambda r: (r[0], r[1]))
        try:
            Y[i] = U.true_to_pred_grid(np.array(arr), [6, 6, 2, 8]).flatten()
        except AssertionError:       # >2 ellipses in one cell: keep the first two (rare; generator quirk)
            keep, seen = [], {}
            for r in arr:
                cell = (min(max(int((r[0] - 40) / 71), 0), 5), min(max(int((r[1] - 40) / 51), 0), 5))
                if seen.get(cell, 0) < 2:
                    keep.append(r)
                    seen[cell] = seen.get(cell, 0) + 1
            Y[i] = U.true_to_pred_grid(np.array(keep), [6, 6, 2, 8]).flatten()
    U.setup_means_and_ranges([6, 6, 2, 8])
    return U.norm_Y(Y).astype(np.float32)


def host_cpu_share():
    """(threads to use, description): the cores this process may really run on -- its affinity mask capped by the
    cgroup CPU quota.  A 1-GPU box shows 100+ logical CPUs but grants about 16; a thread pool sized by the former
    slows the oracle several times over and makes the figure box-dependent."""
    try:
        aff = len(os.sched_getaffinity(0))
    except AttributeError:
        aff = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    n = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    n = min(n, int(os.environ.get("SPNET_CPU_BASELINE_THREADS", 32)))
    return n, dict(affinity_cpus=aff, cgroup_quota_cpus=quota, logical_cpus=os.cpu_count(), cpu_model=model)


def cpu_baseline(X_u8, Y, steps, batch):
    """The oracle (torch-CPU restatement of the reference path) timed on this host: numpy
    augmentation -> forward -> custom_loss + l2 -> backward -> Keras-form Adam.  Median of `steps` steps after
    one warm-up step, on the cores this process is really granted (host_cpu_share)."""
    import torch
    from oracle import numpy_ref as R
    from oracle import torch_ref as T
    from spnet_amd import fake_espi as F
    cores, share = host_cpu_share()
    torch.set_num_threads(cores)
    P = T.init_params(H, W, seed=0)
    tr = T.Trainer(P)
    lrs = R.one_cycle_table(4e-5, 40000, 100, batch)
    rs = np.random.RandomState(0)

    def one(i):
        idx = np.arange(i * batch, (i + 1) * batch) % X_u8.shape[0]
        xb = F.to_network_input(X_u8[idx]).copy()
        for j in range(batch):
            R.augment_image(xb[j], rng=rs)
        tr.step(torch.from_numpy(xb), torch.from_numpy(Y[idx]), float(lrs[i]))

    one(0)                                   # warm-up (allocator, thread pool)
    ts = []
    for i in range(1, steps + 1):
        t0 = time.perf_counter()
        one(i)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return dict(value=round(batch / med, 3), unit="images/sec", cores=cores, kind="port",
                sample="median of %d train steps of batch %d at %dx%d (after 1 warm-up step), oracle/torch_ref.py + "
                       "oracle/numpy_ref.py on %d threads, %.1f s in all; per-step s: %s" %
                       (steps, batch, W, H, cores, sum(ts), " ".join("%.2f" % t for t in ts)),
                host=share)


def secondary(args):
    """Secondary measurements (single GPU): other frame sizes / batch sizes, or inference throughput as
    predict_spnet.py measures it (model.predict over device-resident frames, FPS)."""
    import torch
    from spnet_amd.engine import Engine
    h, w, b = args.height, args.width, args.batch
    dev = torch.device("cuda", 0)
    eng = Engine(h, w, b, device="cuda:0", seed=0, train=(args.mode == "train"), backbone=args.backbone)
    X = torch.rand(b, h, w, 1, device=dev) * 2 - 1
    Y = torch.rand(b, 576, device=dev)

    def step():
        if args.mode == "train":
            eng.train_step(X, Y, 1e-5)
        else:
            eng.forward(X, training=False)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"metric": "%s images/sec, %s backbone (secondary measurement)" % (args.mode, args.backbone),
           "value": round(b * args.steps / dt, 2), "unit": "images/sec", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "dtype": DTYPE if args.backbone != "InceptionResNetV2" else "f32",
           "data": "synthetic (uniform noise)",
           "config": {"workload": "%s, %s, %dx%d frames, batch %d" % (args.mode, args.backbone, w, h, b)}}
    if not args.no_kernel_timers:       # the GEMM family of this configuration against the fp32 MFMA peak (HIP events)
        tim = time_families(eng, step, args.steps, torch.cuda.synchronize)
        if "gemm" in tim.totals():
            out["roofline"] = dict(gemm_roofline_block(tim, args.steps),
                                   measured="HIP events around every GEMM launch during %d extra steps, each enqueued behind two "
                                            "untimed steps (no host gap inside an event pair)" % args.steps)
    print(json.dumps(out))


def predict_measure(X_pool, dev, steps, warmup, host_frames=1024):
    """BASELINE configs[4] on frames already resident in HBM (X_pool [n,384,512,1] on `dev`, n a multiple of 128):
    hipGraph-captured forward of batch 128 (Engine.predict_step), the eager-launch rate, the family rooflines from
    HIP events, and Model.predict over HOST frames (pinned ring + copy stream: PCIe-inclusive, never `value`)."""
    import torch
    from spnet_amd.engine import Engine, KernelTimer
    from spnet_amd.models import Model
    PB = 128
    pool = X_pool.shape[0] // PB * PB
    eng = Engine(H, W, PB, device=str(dev), seed=0, train=False)
    it = [0]

    def step(graph=True):
        lo = (it[0] * PB) % pool
        it[0] += 1
        eng.x_in.copy_(X_pool[lo:lo + PB])
        return eng.predict_step(use_graph=graph)

    def timed(n, graph):
        for _ in range(warmup):
            step(graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(graph)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    dt = timed(steps, True)
    dt_eager = timed(steps, False)
    tim = time_families(eng, lambda: step(False), steps, torch.cuda.synchronize)
    tot = tim.totals()
    d_n, d_ms, d_bytes = tot["dw"]
    n_prof = steps
    roof = gemm_roofline_block(tim, n_prof)
    dw_gbs = d_bytes / (d_ms * 1e-3) / 1e9        # the launches that exist (8 B per element); the rest run in GEMM epilogues
    del eng
    # PCIe-inclusive: model.predict over host frames, as predict_spnet.py calls it
    nh = min(pool, host_frames) // PB * PB
    X_host = X_pool[:nh].cpu().numpy()
    m = Model((H, W, 1), Y0size=576, seed=0, device=str(dev))
    m.predict(X_host[:PB], batch_size=PB)
    t0 = time.perf_counter()
    m.predict(X_host, batch_size=PB)
    dt_host = time.perf_counter() - t0
    # ... and over uint8 host frames (4x fewer bytes over PCIe; the /255*2-1 scaling of utils.py:340-342 on the device)
    host_u8 = None
    if hasattr(m, "predict_u8"):
        U8 = np.clip(np.rint((X_host[..., 0] + 1.0) * 127.5), 0, 255).astype(np.uint8)
        m.predict_u8(U8[:PB], batch_size=PB)
        t0 = time.perf_counter()
        m.predict_u8(U8, batch_size=PB)
        host_u8 = round(nh / (time.perf_counter() - t0), 1)
    del m
    torch.cuda.empty_cache()
    note = "HIP events around every launch of the family during %d eager forwards after the timed region" % n_prof
    return {
        "frames_per_sec": round(PB * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
        "batch": PB, "pool_frames": pool, "eager_frames_per_sec": round(PB * steps / dt_eager, 1),
        "host_streamed_frames_per_sec": round(nh / dt_host, 1),
        "host_streamed_u8_frames_per_sec": host_u8,
        "host_frames_note": "Model.predict over %d host frames: pageable -> pinned ring -> HBM on a copy stream, "
                            "overlapped with the forward passes (PCIe-inclusive; not `value`)" % nh,
        "roofline": dict(roof, measured=note),
        "roofline_secondary": {"kernel": "dw3x3_stream_fwd_kernel (the depthwise layers that run as launches of their own; the "
                                         "others -- second / third units of the 12x16-plane blocks -- run in the producing GEMM's "
                                         "epilogue, gemm_bf16x3_pp_dwbwd_kernel<1>)", "bound": "hbm",
                               "achieved": round(dw_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(dw_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                               "algorithmic_bytes_per_step": round(d_bytes / n_prof),
                               "bytes_per_step_of_layers_fused_into_gemm_epilogues": round(max(DW_FWD_BYTES_PER_IMAGE * PB - d_bytes / n_prof, 0.0)),
                               "launches_per_step": d_n / n_prof, "ms_per_step": round(d_ms / n_prof, 3), "measured": note},
    }


def predict_bench(args):
    """`bench.py --mode predict`: BASELINE configs[4] as its own line -- predict_spnet.py's inference loop (Xception,
    batch 128, 512x384 frames) on one GPU.  A step = one batch of 128 frames through the hipGraph-captured forward;
    frames are resident in HBM when the timed region starts."""
    import torch
    from spnet_amd import fake_espi as F
    from spnet_amd import parallel
    PB = 128
    pool = max(PB, min(args.pool, 2048) // PB * PB)
    parallel.init_distributed()
    dev = parallel.local_device()
    torch.cuda.set_device(dev)
    X_pool, _ = F.generate_device(pool, seed=11, device=str(dev))
    r = predict_measure(X_pool, dev, args.steps, args.warmup)
    print(json.dumps({
        "metric": "inference frames/sec, predict_spnet.py path (model.predict), Xception, batch 128 over 512x384 frames "
                  "(BASELINE configs[4]; secondary to the training metric)",
        "value": r["frames_per_sec"], "unit": "frames/sec", "n_gpus": 1, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": "configs[4]: inference-only forward, Xception, 512x384 fake-ESPI frames, batch 128, "
                               "hipGraph-captured plan, frames resident in HBM", "batch": PB, "frame_hw": [H, W],
                   "pool_frames": pool, "eager_frames_per_sec": r["eager_frames_per_sec"],
                   "host_frames_streamed_frames_per_sec": r["host_streamed_frames_per_sec"],
                   "host_frames_u8_streamed_frames_per_sec": r["host_streamed_u8_frames_per_sec"],
                   "host_frames_note": r["host_frames_note"]},
        "roofline": r["roofline"], "roofline_secondary": r["roofline_secondary"],
    }), flush=True)


def layout_331_measure(dev, steps=20, warmup=3):
    """The reference's own layout (model_type 'monolithic': 331x331 frames, batch 16 -- what its published 96-131
    training images/s and 444-717 inference FPS on TITAN X / 2080 Ti refer to): train steps of
    batch 16 and inference forwards of batch 32 on uniform-noise frames."""
    import torch
    from spnet_amd.engine import Engine
    out = {}
    for mode, b in (("train", 16), ("predict", 32)):
        eng = Engine(331, 331, b, device=str(dev), seed=0, train=(mode == "train"))
        X = torch.rand(b, 331, 331, 1, device=dev) * 2 - 1
        Y = torch.rand(b, 576, device=dev)

        def step():
            if mode == "train":
                eng.train_step(X, Y, 1e-5)
            else:
                eng.x_in.copy_(X)
                eng.predict_step()

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = {"images_per_sec": round(b * steps / dt, 1), "batch": b, "ms_per_step": round(1e3 * dt / steps, 3),
                     "steps": steps}
        del eng
        torch.cuda.empty_cache()
    out["reference_published"] = "train 96-131 images/s, predict 444-717 FPS (TITAN X / RTX 2080 Ti, BASELINE.md section 1)"
    return out


def backbone_leg(dev, backbone, batch, steps=20, warmup=5, predict=False):
    """The other two backbones of BASELINE.json's configs in the driver-witnessed line: configs[3] (Inception-ResNet-v2,
    batch 16) and configs[0]'s model on the GPU (MobileNet, batch 8), 512x384 frames (spnet/models.py:346-359,
    spnet/config.py:50-52).  Full train steps (forward, custom_loss, backward, Adam + l2) on uniform-noise frames; the
    GEMM family against the fp32 MFMA peak from HIP events; optionally the inference forward of the same batch size."""
    import torch
    from spnet_amd.engine import Engine
    eng = Engine(H, W, batch, device=str(dev), seed=0, backbone=backbone)
    X = torch.rand(batch, H, W, 1, device=dev) * 2 - 1
    Y = torch.rand(batch, 576, device=dev)

    def step():
        eng.train_step(X, Y, 1e-5)

    for _ in range(warmup):
        step()
    dts = []
    for _ in range(3):       # (these small steps are close to host-bound: one slow repeat is the host's, not the plan's)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = min(dts)
    out = {"workload": "%s, 512x384 frames, batch %d, full train step, uniform-noise frames" % (backbone, batch),
           "train_images_per_sec": round(batch * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
           "batch": batch, "repeats_ms_per_step": [round(1e3 * t / steps, 3) for t in dts],
           "timing": "best of 3 repeats of %d steps (a secondary leg; the headline `value` is one timed region)" % steps}
    n_prof = max(5, steps // 2)
    tim = time_families(eng, step, n_prof, torch.cuda.synchronize)
    if "gemm" in tim.totals():
        out["roofline"] = dict(gemm_roofline_block(tim, n_prof),
                               measured="HIP events around every GEMM launch during %d extra steps, each behind two untimed ones" % n_prof)
    del eng
    torch.cuda.empty_cache()
    if predict:
        eng = Engine(H, W, batch, device=str(dev), seed=0, backbone=backbone, train=False)
        eng.x_in.copy_(X)
        for _ in range(warmup):
            eng.predict_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.predict_step()
        torch.cuda.synchronize()
        out["predict_frames_per_sec"] = round(batch * steps / (time.perf_counter() - t0), 1)
        del eng
        torch.cuda.empty_cache()
    return out


def bf16x3_alt_measure(dev, iters=200):
    """`roofline_alt`: the bf16x3 kernels (csrc/gemm_bf16x3.hip: six bf16 MFMAs with fp32 accumulation per product block,
    operands as bf16 planes in 1-KiB pieces) beside the exact fp32 MFMA kernel on the network's dominant shape, the Xception
    middle-flow pointwise layer at batch 32 (6144 pixels x 728 x 728), isolated, operands rotating over four buffers: the
    planes x planes kernel (forward / data gradient), round 4's kernel (fp32 A split while staged), the weight gradient from
    planes.  Error of the forward kernels against float64 (torch.float64 on the device), relative to |a_row| * |w_col|."""
    import torch
    from spnet_amd import _lib as L
    M, N, K = 6144, 728, 728
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    nb = 4
    a = [torch.randn(M, K, device=dev, generator=g) for _ in range(nb)]
    gy = torch.randn(M, N, device=dev, generator=g) * 0.1
    w = torch.randn(K, N, device=dev, generator=g) * 0.05
    pe = lambda r, k: 3 * int(L.spnet_bf16x3_plane_elems(r, k))
    planes = torch.zeros(pe(N, K), dtype=torch.int16, device=dev)
    apl = [torch.zeros(pe(M, K), dtype=torch.int16, device=dev) for _ in range(nb)]
    gpl = torch.zeros(pe(M, N), dtype=torch.int16, device=dev)
    c3, c1, cp = (torch.empty(M, N, device=dev) for _ in range(3))
    NW = 8                    # weight gradients per launch here (the step batches the middle flow's 24)
    dw = torch.empty(NW, K, N, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    split = lambda i=0: L.spnet_split_bf16x3(w.data_ptr(), planes.data_ptr(), K, N, st)
    for i in range(nb):
        L.spnet_split_rows_bf16x3(a[i].data_ptr(), K, apl[i].data_ptr(), M, K, st)
    L.spnet_split_rows_bf16x3(gy.data_ptr(), N, gpl.data_ptr(), M, N, st)
    jobs = torch.tensor([v for b in range(NW) for v in (apl[b % nb].data_ptr(), gpl.data_ptr(), dw[b].data_ptr())],
                        dtype=torch.int64, device=dev)
    fp = lambda i: L.spnet_gemm_bf16x3_pp(apl[i % nb].data_ptr(), planes.data_ptr(), cp.data_ptr(), N, M, N, K, None, None, st)
    f3 = lambda i: L.spnet_gemm_bf16x3_fwd(a[i % nb].data_ptr(), K, planes.data_ptr(), c3.data_ptr(), N, M, N, K, st)
    f1 = lambda i: L.spnet_gemm_f32(a[i % nb].data_ptr(), 0, K, w.data_ptr(), 1, N, c1.data_ptr(), N, M, N, K, 1, None, 0, None, 0, st)
    fw = lambda i: L.spnet_gemm_bf16x3_wgrad_batched(jobs.data_ptr(), NW, K, N, M, 1, st)

    def t(fn):
        for i in range(10):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    # yardstick only (the product never calls the library): the vendor's bf16 GEMM executing the same number of bf16 MFMAs
    # as one bf16x3 product -- [M][6K] x [N][6K]^T, fp32 accumulate (its operand bytes are twice the planes')
    a6 = [torch.randn(M, 6 * K, device=dev, generator=g).to(torch.bfloat16) for _ in range(nb)]
    w6 = (torch.randn(N, 6 * K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    fv = lambda i: torch.mm(a6[i % nb], w6.t())
    split()
    tp, t3, t1, ts, tw = t(fp), t(f3), t(f1), t(split), t(fw)
    try:
        tv = t(fv)
    except Exception:                    # (a library without a bf16 path: no yardstick)
        tv = None
    del a6, w6
    fp(0); f3(0); f1(0)
    ref = a[0].double() @ w.double()
    scale = a[0].double().norm(dim=1)[:, None] * w.double().norm(dim=0)[None, :]
    ep, e1 = (cp.double() - ref).abs() / scale, (c1.double() - ref).abs() / scale
    same = bool(torch.equal(cp, c3))
    fl = 2.0 * M * N * K
    return {"kernel": "gemm_bf16x3_pp_kernel, isolated on the dominant shape: both operands as bf16x3 planes (1-KiB pieces, "
                      "LDS-DMA), 6 bf16 MFMAs per product block, fp32 accumulate", "shape": {"M": M, "N": N, "K": K}, "bound": "mfma",
            "achieved": round(fl / tp / 1e6, 1), "peak": round(X3_PEAK_TFLOPS, 1), "unit": "TFLOP/s (fp32-equivalent)",
            "frac": round(fl / tp / 1e6 / X3_PEAK_TFLOPS, 4),
            "frac_vs_real_operand_bf16_rate": round(fl / tp / 1e6 / X3_REAL_OPERAND_TFLOPS, 4),
            "peak_note": "dense bf16 MFMA peak %.0f TFLOP/s / 6 MFMAs per product block" % BF16_MFMA_PEAK_TFLOPS,
            "avg_launch_us": round(tp, 1), "fp32_a_split_in_kernel_us": round(t3, 1), "exact_fp32_kernel_us": round(t1, 1),
            "speedup_vs_exact": round(t1 / tp, 3), "bit_identical_to_fp32_a_kernel": same,
            "vendor_bf16_gemm_same_mfma_work_us": None if tv is None else round(tv, 1),
            "vendor_note": "torch.mm on bf16 [M][6K] x [N][6K]^T (hipBLASLt / rocBLAS, fp32 accumulate): the matrix work of this "
                           "bf16x3 product done by the vendor's GEMM; a yardstick for `frac`, never on the product path",
            "wgrad_from_planes_us": round(tw / NW, 1), "wgrad_tflops_fp32_equivalent": round(NW * fl / tw / 1e6, 1),
            "wgrad_note": "per layer, %d layers of this shape per launch (gemm_bf16x3_wgrad_kernel)" % NW,
            "weight_split_us": round(ts, 1),
            "max_rel_err_vs_f64": float("%.3g" % ep.max().item()), "rms_rel_err_vs_f64": float("%.3g" % ep.pow(2).mean().sqrt().item()),
            "exact_max_rel_err_vs_f64": float("%.3g" % e1.max().item()),
            "exact_rms_rel_err_vs_f64": float("%.3g" % e1.pow(2).mean().sqrt().item()),
            "iters": iters}


def exact_chain_step_measure(dev, X_pool, steps=20, warmup=5):
    """`roofline_alt.train_step`: the benchmark train step (Xception, batch 32, 512x384, without the augmentation kernels)
    on two engines over the same frames, alternating: `pointwise="bf16x3"` (the product path: the three GEMMs of every
    pointwise convolution with >= 256 channels on csrc/gemm_bf16x3.hip, their operands written as bf16 planes by the
    producing kernels) and `pointwise="f32"` (every GEMM the k-ordered fp32 MFMA chain, the product path of rounds 1-3):
    images/s and the GEMM family's time from HIP events.  The blended data-gradient GEMMs of blocks 2-3, block1_conv2 and the
    Dense head are the exact fp32 kernels in both columns."""
    import torch
    from spnet_amd.engine import Engine
    engs = {"bf16x3": Engine(H, W, BATCH, device=str(dev), seed=0), "f32": Engine(H, W, BATCH, device=str(dev), seed=0, pointwise="f32")}
    X = X_pool[:BATCH].contiguous()
    Y = torch.rand(BATCH, 576, device=dev)
    out = {}
    for name, mode in (("bf16x3", "bf16x3"), ("exact_fp32", "f32"), ("bf16x3_again", "bf16x3")):
        eng = engs[mode]

        def step():
            eng.train_step(X, Y, 1e-6)

        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        side = eng.wgrad_stream
        eng.wgrad_stream = None                      # family times on one stream, as the main roofline leg
        step()
        n_prof = max(5, steps // 2)
        tim = time_families(eng, step, n_prof, torch.cuda.synchronize)
        eng.wgrad_stream = side
        g_n, g_ms, g_flop = tim.totals()["gemm"]
        x3 = [v for t, v in tim.tagged().items() if isinstance(t, tuple) and str(t[0]).startswith("x3")]
        out[name] = {"images_per_sec": round(BATCH * steps / dt, 1), "ms_per_step": round(1e3 * dt / steps, 3),
                     "gemm_family_ms_per_step": round(g_ms / n_prof, 3),
                     "gemm_family_tflops": round(g_flop / (g_ms * 1e-3) / 1e12, 1),
                     "gemm_launches_per_step": g_n / n_prof,
                     "bf16x3_launches_per_step": sum(v[0] for v in x3) / n_prof,
                     "bf16x3_ms_per_step": round(sum(v[1] for v in x3) / n_prof, 3)}
    out["step_speedup_over_exact"] = round(out["exact_fp32"]["ms_per_step"] / out["bf16x3"]["ms_per_step"], 4)
    out["gemm_family_speedup_over_exact"] = round(out["exact_fp32"]["gemm_family_ms_per_step"] /
                                                  out["bf16x3"]["gemm_family_ms_per_step"], 4)
    del eng, engs
    torch.cuda.empty_cache()
    return out


def rccl_version():
    try:
        import torch
        return ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001  (diagnostic field only)
        return "unknown (%s)" % type(e).__name__


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, the way the driver does
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`), as a CHILD process
    -- this parent has not imported torch.cuda or touched the GPU, and never exec()s -- relay its output (rank 0
    prints the one JSON line) and exit with its return code (reference capability: spnet/multi_gpu.py:35-88,
    train_spnet.py:55).  Every rank's stderr is also kept in a log directory (`--tee 2`); when the job fails, the last 40
    stderr lines of every rank that wrote any are repeated at the end, rank by rank, so that the tail of the driver's log
    shows WHY (an RCCL / HIP message of one rank) and not only torchrun's summary of who died."""
    import glob
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    logdir = tempfile.mkdtemp(prefix="spnet_bench_ranks_")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "--log-dir", logdir, "--tee", "2",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # the pool's host driver supports dmabuf IPC only: without this RCCL's peer-to-peer setup fails with
    # `hipIpcGetMemHandle: invalid argument` (it is exported on the GPU boxes already; kept for any other caller)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("NCCL_DEBUG", "WARN")         # a failed RCCL init / collective says why
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    rc = subprocess.run(cmd, env=env).returncode
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank child job failed with exit code %d\n" % (n, rc))
        for f in sorted(glob.glob(os.path.join(logdir, "**", "stderr.log"), recursive=True)):
            try:
                tail = open(f, errors="replace").read().splitlines()[-40:]
            except OSError:
                continue
            if tail:
                sys.stderr.write("---- last %d stderr lines of %s\n%s\n" % (len(tail), os.path.relpath(f, logdir), "\n".join(tail)))
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pool", type=int, default=4096,
                    help="synthetic frames resident per GPU (SURVEY 8d: >= 4,096; 3.2 GB of fp32 frames, far beyond "
                         "the 256 MiB Infinity Cache)")
    ap.add_argument("--pool-source", choices=["device", "host"], default="device",
                    help="device: frames rasterised in HBM by csrc/espi.hip (SURVEY 8f-2); host: the PIL generator")
    ap.add_argument("--sustained-seconds", type=float, default=6.0,
                    help="after the timed region keep stepping for this long and report the rate separately "
                         "(clocks under sustained load); 0 = skip")
    ap.add_argument("--cpu-baseline-steps", type=int, default=5)
    ap.add_argument("--cpu-baseline-batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the configs[4] inference leg and the 331x331 layout leg appended to the N=1 line")
    ap.add_argument("--mode", choices=["train", "predict"], default="train",
                    help="predict: inference-only forward (BASELINE configs[4]); secondary, not the headline metric")
    ap.add_argument("--height", type=int, default=H)
    ap.add_argument("--width", type=int, default=W)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--backbone", default="Xception", choices=["Xception", "MobileNet", "InceptionResNetV2"],
                    help="secondary measurements only (the headline metric is Xception)")
    ap.add_argument("--launch-check", action="store_true",
                    help="only start the ranks, join the process group, count them with one all-reduce and print that "
                         "(no kernels: runs on a CPU-only host over gloo; tests/test_host_cpu.py)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="run the weight-gradient GEMMs on the main stream (the roofline leg always does)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # (also under the driver's own launcher)
        try:
            return run(args)
        except BaseException as e:      # noqa: BLE001  (re-raised: the launcher must see the failure)
            if not isinstance(e, SystemExit) or e.code not in (0, None):
                import traceback
                sys.stderr.write("bench.py: rank %s of %s failed: %s: %s\n%s\n" % (
                    os.environ.get("RANK", "?"), os.environ.get("WORLD_SIZE", "?"), type(e).__name__, e,
                    "".join(traceback.format_exc().splitlines(True)[-12:])))
                sys.stderr.flush()
            raise
    return run(args)


def run(args):
    """One rank of the benchmark (the only rank at N = 1)."""
    import torch
    import torch.distributed as dist
    from spnet_amd import fake_espi as F
    from spnet_amd import parallel
    from spnet_amd.augmentation import DeviceAugmenter
    from spnet_amd.callbacks import get_1cycle_schedule
    from spnet_amd.engine import Engine, KernelTimer
    from spnet_amd import _lib as L

    if args.launch_check:
        rank, local_rank, world = parallel.init_distributed()
        if world != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
        seen = 1
        if world > 1:
            ones = torch.ones(1, device=parallel.local_device() if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(ones)
            seen = int(round(float(ones.item())))
        if os.environ.get("SPNET_BENCH_FAIL_RANK") == str(rank):      # test hook (tests/test_host_cpu.py)
            raise RuntimeError("SPNET_BENCH_FAIL_RANK=%d: this rank fails on purpose" % rank)
        backend = dist.get_backend() if world > 1 else None
        if world > 1:
            dist.barrier()          # (a rank that failed above never arrives: no result line from a failed job)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "n_ranks_seen": seen,
                              "collective_backend": backend}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.mode == "predict" and (args.height, args.width) == (H, W) and args.batch in (BATCH, 128) \
            and args.backbone == "Xception":
        return predict_bench(args)
    if (args.height, args.width, args.batch, args.mode, args.backbone) != (H, W, BATCH, "train", "Xception"):
        return secondary(args)
    # ---- synthetic data first: the generator forks worker processes, which must happen before this process
    # initialises the GPU or joins the process group (rank-specific frames: weak scaling)
    rank, local_rank, world = parallel.env_world()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    t_gen = time.perf_counter()
    # host frames: the whole pool, the CPU baseline's sample, or none at all (generate() forks only while no HIP
    # context can exist in this process -- not under a profiler's preloaded runtime, fake_espi.gpu_may_be_live)
    need_cpu = world == 1 and not args.no_cpu_baseline
    n_host = args.pool if args.pool_source == "host" else (64 if need_cpu else 0)
    X_u8 = Y_host = None
    if n_host:
        X_u8, labels = F.generate(n_host, seed=1 + rank, workers=max(2, min(16, (os.cpu_count() or 8) // world)))
        Y_host = labels_to_Y(labels)
    rank, local_rank, world = parallel.init_distributed()
    dev = parallel.local_device()                  # (>1 rank per GPU only in gloo rehearsals)
    torch.cuda.set_device(dev)
    # ---- resident in HBM before timing
    if args.pool_source == "device":
        X_pool, labels = F.generate_device(args.pool, seed=1 + rank, device=str(dev))
        Y_pool = torch.from_numpy(labels_to_Y(labels)).to(dev)
    t_gen = time.perf_counter() - t_gen
    if args.pool_source == "host":
        X_pool = torch.from_numpy(F.to_network_input(X_u8)).to(dev)
        Y_pool = torch.from_numpy(Y_host).to(dev)
    np.random.seed(1 + rank)

    eng = Engine(H, W, BATCH, device=str(dev), seed=0, rank=rank)
    aug = DeviceAugmenter(X_pool)
    reducer = eng.make_reducer() if world > 1 else None
    if reducer is not None:
        reducer.exposed = []      # Engine._step_body leaves an event pair around reducer.finish() per step
    # 1-cycle table of the reference's own run configuration (lr_max 4e-5, 40k frames, 100 epochs)
    lrs = get_1cycle_schedule(lr_max=4e-5, n_data_points=40000, epochs=100, batch_size=BATCH * world)
    order = np.random.RandomState(7).permutation(args.pool)
    it = [0]

    def batch_indices(i):
        return order[(np.arange(BATCH) + i * BATCH) % args.pool]

    # The input pipeline is software-pipelined like any prefetching loader (the reference augments a whole epoch ahead,
    # callbacks.py:272-341): the host-side parameter draw of batch i+1 (~1 ms of numpy RNG calls in the reference's
    # order) happens right after step i has been enqueued, so a step begins with uploads and launches.  Every step still
    # performs exactly one draw; what it saves is the GPU idling behind that draw in the first step after a fence.
    drawn = [aug.draw(batch_indices(0))]

    def step():
        i = it[0]
        it[0] += 1
        aug.apply(drawn[0], eng.x_in)
        L.gather_rows(Y_pool, aug.index_dev, eng.y_true)      # the indices travelled with the parameters
        out = eng.train_step(None, None, float(lrs[i % len(lrs)]), reducer=reducer)
        drawn[0] = aug.draw(batch_indices(i + 1))
        return out

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    side_stream = eng.wgrad_stream
    if args.no_overlap:
        eng.wgrad_stream = None
    trace = os.environ.get("SPNET_BENCH_TRACE") == "1"      # dev: per-step GPU / host times of the timed region
    evs, hts = [], []
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    if trace:
        evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
    for _ in range(args.steps):
        th = time.perf_counter()
        out = step()
        if trace:
            evs.append(torch.cuda.Event(enable_timing=True)); evs[-1].record()
            hts.append(1e3 * (time.perf_counter() - th))
    t_host = time.perf_counter() - t0          # host time to ENQUEUE the K steps (launches are asynchronous)
    fence()
    dt = time.perf_counter() - t0
    if trace and rank == 0:
        sys.stderr.write("trace: wall %.2f ms; GPU ms per step: %s\n" % (1e3 * dt, " ".join("%.2f" % evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps))))
        sys.stderr.write("trace: host ms per step: %s\n" % " ".join("%.1f" % v for v in hts))
    loss = float(out[5])
    # N > 1: what an 8-GPU shortfall would be made of -- the part of the gradient all-reduce the main stream WAITS for
    # between the end of backward and the optimizer (everything else ran underneath backward), the bytes, the bucket plan
    comm = None
    if reducer is not None:
        pairs = reducer.exposed[-args.steps:]
        reducer.exposed = None
        exposed = [e0.elapsed_time(e1) for e0, e1 in pairs]
        buckets, tail = eng.grad_buckets()
        comm = {"allreduce_exposed_ms_per_step": round(float(np.mean(exposed)), 3),
                "allreduce_exposed_ms_max": round(float(np.max(exposed)), 3),
                "bytes_allreduced_per_step_per_rank": 4 * eng.n_theta,
                "ring_bytes_sent_per_step_per_rank": round(2.0 * (world - 1) / world * 4 * eng.n_theta),
                "buckets": {"launched_during_backward": len(buckets), "dense_head_pieces": sum(1 for b in buckets if b[2] is eng.nodes[-1]),
                            "mb": [round(4 * (hi - lo) / 2 ** 20, 1) for lo, hi, _ in buckets],
                            "tail_at_end_of_backward_mb": round(4 * sum(hi - lo for lo, hi in tail) / 2 ** 20, 1)},
                "measured": "HIP events around GradReducer.finish() on the main stream, mean / max over the timed steps of rank 0"}

    # Host cost of enqueueing ONE step into an idle GPU (the figure above is taken under queue back-pressure: the
    # host runs ahead until the launch queue is full and then waits for the GPU, so it reads ~ the GPU step time).
    t_idle = []
    for _ in range(3):
        fence()
        t1 = time.perf_counter()
        step()
        t_idle.append(time.perf_counter() - t1)
    fence()
    t_host_idle = min(t_idle)

    n_ranks_seen, backend = 1, None
    if world > 1:                # every rank must agree on the timed region before anything is derived from it
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        dist.all_reduce(ones)                    # how many ranks really took part in the collectives
        n_ranks_seen = int(round(float(ones.item())))
        backend = dist.get_backend()

    # Sustained leg: the same step for several seconds (reported separately; `value` stays the K timed steps).  The
    # step COUNT is fixed from the timed region's rate, identically on every rank (a time-based exit would let ranks
    # leave the loop after different numbers of collectives).
    sustained = None
    if args.sustained_seconds > 0:
        n_s = max(20, int(args.sustained_seconds / (dt / args.steps) / 20 + 1) * 20)
        fence()
        t1 = time.perf_counter()
        for _ in range(n_s):
            step()
        fence()
        dts = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt[0].item())
        sustained = {"seconds": round(dts, 2), "steps": n_s, "images_per_sec": round(BATCH * world * n_s / dts, 1),
                     "ms_per_step": round(1e3 * dts / n_s, 3)}

    # Roofline leg: the same K steps replayed with every kernel on ONE stream (no weight-gradient overlap),
    # so that a kernel's HIP-event duration is its own and not that of two kernels sharing the chip.
    timer = entry_times = None
    if not args.no_kernel_timers:
        eng.wgrad_stream = None
        step()
        fence()
        timer = time_families(eng, step, args.steps, fence)
        entry_times = time_entry_points(eng, step, args.steps, fence)      # (every rank: a step contains the collectives)
        eng.wgrad_stream = None if args.no_overlap else side_stream

    result = None
    if rank == 0:
        ips = BATCH * world * args.steps / dt
        result = {
            "metric": METRIC,
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
            "config": {"workload": "configs[1]: Xception, fake-ESPI 512x384 (HxW 384x512x1, model_type 'big'), "
                                   "batch 32 per GPU, full train step (augment+fwd+custom_loss+bwd+Adam+l2)",
                       "global_batch": BATCH * world, "frame_hw": [H, W], "pool_frames_per_gpu": args.pool,
                       "parallelism": "dp%d" % world, "n_ranks_seen": n_ranks_seen,
                       "collective_backend": backend, "rccl_version": rccl_version() if backend == "nccl" else None,
                       "devices_visible": torch.cuda.device_count(), "final_loss": round(loss, 6),
                       "wgrad_overlap": not args.no_overlap, "arithmetic": ARITHMETIC,
                       "host_enqueue_ms_per_step_backpressured": round(1e3 * t_host / args.steps, 3),
                       "host_enqueue_ms_per_step_gpu_idle": round(1e3 * t_host_idle, 3),
                       "pool_source": args.pool_source, "pool_generation_s": round(t_gen, 1)},
        }
        if comm is not None:
            result["config"]["comm"] = comm
        if sustained is not None:
            result["sustained"] = sustained
        if timer is not None:
            tot = timer.totals()
            fam = {}
            for k, (n, ms, work) in tot.items():
                fam[k] = dict(launches_per_step=n / args.steps, ms_per_step=round(ms / args.steps, 3))
            tr, tr_note = measured_traffic()
            g_traffic = None
            if "gemm" in tr:    # PMC bytes of the family per step / this run's family launches per step
                g_traffic = (tr["gemm"]["hbm_read_bytes_per_step"] + tr["gemm"]["hbm_write_bytes_per_step"]) / \
                            (tot["gemm"][0] / args.steps)
            d_traffic = None
            if "dw_fwd" in tr and "dw_bwd" in tr:
                d_traffic = (tr["dw_fwd"]["hbm_bytes_per_launch"] * tr["dw_fwd"]["launches"] +
                             tr["dw_bwd"]["hbm_bytes_per_launch"] * tr["dw_bwd"]["launches"]) / \
                            (tr["dw_fwd"]["launches"] + tr["dw_bwd"]["launches"])
            g_n, g_ms, g_flop = tot["gemm"]
            d_n, d_ms, d_bytes = tot["dw"]
            # depthwise family: the algorithmic bytes of the launches that EXIST (8 B per element forward, 12 B backward:
            # SURVEY 8(d)); the backward of the layers fused into the data-gradient GEMM's epilogue moves no dz any more
            dw_gbs = d_bytes / (d_ms * 1e-3) / 1e9
            dw_fused_bytes = DW_TRAIN_BYTES_PER_IMAGE * BATCH - d_bytes / args.steps
            # the dominant kernel of the step against its own peak; the GEMM family (every kernel against its own) beside it
            roof_gemm = gemm_roofline_block(timer, args.steps)
            dom = roof_gemm["kernel"].split(":")[0]
            if dom in tr and roof_gemm["launches_per_step"] > 0:      # PMC bytes of that kernel per launch
                roof_gemm["traffic"] = round((tr[dom]["hbm_read_bytes_per_step"] + tr[dom]["hbm_write_bytes_per_step"]) /
                                             roof_gemm["launches_per_step"])
                if roof_gemm.get("algorithmic_bytes_per_launch"):
                    roof_gemm["traffic_over_algorithmic"] = round(roof_gemm["traffic"] / roof_gemm["algorithmic_bytes_per_launch"], 3)
                    roof_gemm["traffic_note"] = ("PMC bytes past the L2 per launch (FETCH_SIZE x 2 + WRITE_SIZE) against 6 B per operand "
                                                 "element + 4 B per result: the eight XCDs each fetch the weight planes (3.2 MB) into "
                                                 "their own L2, and Infinity-Cache hits count as fetches")
            if "gemm" in tr:
                roof_gemm["family"]["traffic_per_launch"] = round(g_traffic)
            roof_dw = {"kernel": "dw3x3_{stream,tile}_fwd_kernel + dw3x3_{stream,tile}_bwd_kernel (34 depthwise layers, fwd + fused bwd: streaming form on the entry-flow planes, LDS-tiled form with the folded BatchNorm finalize on the 12x16 / 6x8 planes)",
                       "bound": "hbm", "achieved": round(dw_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(dw_gbs / HBM_PEAK_GBS, 4),
                       "traffic": None if d_traffic is None else round(d_traffic),
                       "algorithmic_bytes_per_launch": round(d_bytes / d_n),
                       "algorithmic_bytes_per_step": round(d_bytes / args.steps),
                       "bytes_per_step_of_layers_fused_into_the_gemm_epilogue": round(max(dw_fused_bytes, 0.0)),
                       "fused_note": "SURVEY 8(d)'s 157.6 MB per image count every depthwise layer forward + backward; the backward "
                                     "of the 12x16 / 6x8-plane layers runs in gemm_bf16x3_pp_dwbwd_kernel's epilogue (dz never "
                                     "reaches HBM): their 12 B per element are not in `achieved`, their time is in the GEMM family",
                       "avg_launch_us": round(1e3 * d_ms / d_n, 2), "launches_per_step": d_n / args.steps,
                       "ms_per_step": round(d_ms / args.steps, 3)}
            # the depthwise family by sub-family: entry flow (blocks 2-4: 93x125 / 47x63 / 24x32 planes, large tensors),
            # middle flow (12x16x728 planes: 18 MB tensors, latency-bound launches), exit flow (6x8 planes); bytes =
            # the per-launch algorithmic figure (8 B per element forward, 12 B backward)
            subs = {}
            for tag, (n, ms, work) in timer.tagged().items():
                if not (isinstance(tag, tuple) and len(tag) == 4 and str(tag[0]).startswith("dw")):
                    continue
                ph, pw_ = tag[1], tag[2]
                key = "entry" if ph * pw_ > 12 * 16 else ("middle" if ph * pw_ == 12 * 16 else "exit")
                a = subs.setdefault(key, [0, 0.0, 0.0])
                a[0] += n; a[1] += ms; a[2] += work
            roof_dw["sub_families"] = {
                k: {"launches_per_step": v[0] / args.steps, "ms_per_step": round(v[1] / args.steps, 3),
                    "algorithmic_bytes_per_step": round(v[2] / args.steps),
                    "achieved": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "unit": "GB/s",
                    "frac": round(v[2] / (v[1] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)} for k, v in sorted(subs.items())}
            note = ("HIP events around every launch of the family during %d extra steps replayed on one stream "
                    "(weight-gradient overlap off) right after the timed region, each enqueued behind one untimed step so "
                    "that no event pair contains a host gap" % args.steps)
            roof_gemm["measured"] = roof_dw["measured"] = note
            roof_gemm["clock_note"] = ("peaks are at the 2.4 GHz maximum clock; under this family's load the chip holds 1.7-2.1 GHz "
                                       "on real operands (2.3-2.4 GHz on zeros): profiles/r02_c_diag_gemm_phases.txt, "
                                       "profiles/r04_k_diag_bf16x3_power_probe.txt, profiles/r05_x3pp_knockouts.txt")
            roof_gemm["traffic_source"] = roof_dw["traffic_source"] = tr_note
            result["roofline"] = roof_gemm if g_ms >= d_ms else roof_dw
            result["roofline_secondary"] = roof_dw if g_ms >= d_ms else roof_gemm
            result["kernel_families"] = fam
            if entry_times is not None:
                result["roofline_other"] = other_family_rooflines(entry_times, args.steps)
                result["roofline_other"]["measured"] = ("HIP events around every launching entry point of the C ABI (spnet_amd._lib.set_tracer) "
                                                        "during %d further steps on one stream; bytes: bench.algorithmic_bytes" % args.steps)
        if world == 1 and not args.no_secondary:
            # the secondary configurations, in the same driver-witnessed line (N = 1 only; ~10 s): BASELINE configs[4]
            # (predict_spnet.py:84-87's FPS) on the resident pool, and the reference's own 331x331 layout
            del eng, aug
            torch.cuda.empty_cache()
            result["predict"] = predict_measure(X_pool[:min(args.pool, 2048) // 128 * 128], dev, 10, 2)
            result["layout_331"] = layout_331_measure(dev)
            result["irv2"] = backbone_leg(dev, "InceptionResNetV2", 16, predict=True)          # BASELINE configs[3]
            result["mobilenet"] = backbone_leg(dev, "MobileNet", 8)                            # configs[0]'s model
            result["roofline_alt"] = bf16x3_alt_measure(dev)
            result["roofline_alt"]["train_step"] = exact_chain_step_measure(dev, X_pool)
            eng = aug = None
        if world == 1 and not args.no_cpu_baseline:
            del eng, aug, X_pool
            torch.cuda.empty_cache()
            result["cpu_baseline"] = cpu_baseline(X_u8, Y_host, args.cpu_baseline_steps, args.cpu_baseline_batch)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
