#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace) over
`bench.py --steps K --warmup W --no-cpu-baseline --no-kernel-timers` into profiles/<tag>_hbm_traffic.json:
HBM bytes per launch for the kernel families bench.py reports.

Corrections (MI355X_MICROARCH.md, 'HBM'): counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests as
64 bytes, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

The output is stamped with bench.kernel_source_hash() of the tree the passes ran on (and, when given, the commit):
bench.py refuses to quote the file for any other kernel/plan state.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps_in_trace> <out.json> [commit]"""
import collections, csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FAMILIES = {
    "gemm": ("gemm_f32_kernel", "gemm_bf16x3_", "reduce_slabs_kernel", "conv3x3_fwd_kernel", "conv3x3_dgrad_kernel", "conv3x3_wgrad_kernel"),
    "dw_fwd": ("dw3x3_tile_fwd_kernel", "dw3x3_stream_fwd_kernel"),
    "dw_bwd": ("dw3x3_tile_bwd_kernel", "dw3x3_stream_bwd_kernel"),
    # single kernels (bench.py quotes the dominant kernel's own traffic)
    "gemm_bf16x3_pp_kernel": ("gemm_bf16x3_pp_kernel",),
    "gemm_bf16x3_wgrad_kernel": ("gemm_bf16x3_wgrad_kernel",),
    "gemm_bf16x3_pp_dwbwd_kernel<0>": ("gemm_bf16x3_pp_dwbwd_kernel<0",),
    "gemm_bf16x3_pp_dwbwd_kernel<1>": ("gemm_bf16x3_pp_dwbwd_kernel<1",),
    "bn_bwd_fused_vec_kernel": ("bn_bwd_fused_vec_kernel",),
}


def families(name):
    return [fam for fam, keys in FAMILIES.items() if any(k in name for k in keys)]


def fold(path, counter):
    tot, launches = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for fam in families(r["Kernel_Name"]):
            tot[fam] += float(r["Counter_Value"])
            launches[fam].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in launches.items()}


fetch, nl = fold(sys.argv[1], "FETCH_SIZE")
write, _ = fold(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
out = {}
for fam in FAMILIES:
    rd = fetch.get(fam, 0.0) * 1024.0 * 2.0          # KB -> bytes, gfx950 correction
    wr = write.get(fam, 0.0) * 1024.0
    n = nl.get(fam, 0)
    # reduce_slabs is part of a split-K GEMM call, not a call of its own: launches = spnet_gemm-level calls
    out[fam] = {"launches": n, "launches_per_step": n / steps,
                "hbm_bytes_per_launch": (rd + wr) / max(n, 1),
                "hbm_read_bytes_per_step": rd / steps, "hbm_write_bytes_per_step": wr / steps}
from bench import kernel_source_hash
out["kernel_source_hash"] = kernel_source_hash()
out["commit"] = sys.argv[5] if len(sys.argv) > 5 else "?"
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
