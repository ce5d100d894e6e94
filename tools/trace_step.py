#!/usr/bin/env python3
"""Dev tool: per-kernel table of ONE train step out of a rocprofv3 --kernel-trace CSV (steps are delimited
by the optimizer's closing sum_partials_kernel launch: since round 5 a step holds two adam_l2_kernel launches -- the Dense
head's range and the rest).  usage: trace_step.py <kernel_trace.csv> [step_index]"""
import collections, csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sum_partials_kernel")]
if len(idx) < 2:
    idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_l2")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
seg = rows[idx[k] + 1: idx[k + 1] + 1]
span = (int(seg[-1]["End_Timestamp"]) - int(rows[idx[k]]["End_Timestamp"])) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print("step span %.3f ms, kernel time %.3f ms, %d launches" % (span, tot / 1e6, len(seg)))
fam = collections.defaultdict(float)
for name, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    f = ("gemm" if ("gemm_f32" in name or "gemm_bf16x3" in name or "split_bf16x3" in name or "reduce_slabs" in name or "conv3x3_fwd_k" in name or "conv3x3_dgrad" in name or "conv3x3_wgrad" in name)
         else "dw" if "dw3x3" in name else "bn" if "bn_" in name else "stem" if ("conv3x3" in name or "conv1_" in name)
         else "pool" if "pool" in name else "other")
    fam[f] += ns
    print("%-72s n=%3d avg %7.1f us tot %7.3f ms" % (name[:72], n, ns / n / 1e3, ns / 1e6))
print({k: round(v / 1e6, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
