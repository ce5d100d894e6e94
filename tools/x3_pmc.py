#!/usr/bin/env python3
"""Dev tool (GPU box): SQ / LDS / TCP counters of the bf16x3 kernel beside the exact fp32 MFMA kernel (one rocprofv3 --pmc
pass per counter group, --kernel-trace only; the names are taken from `rocprofv3 -L` so that an unknown one is skipped
instead of failing the pass).  usage: x3_pmc.py OUTDIR"""
import csv, glob, os, re, subprocess, sys

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout
names = set(re.findall(r"\b((?:SQ|TCP|TA|TCC|GRBM)_[A-Za-z0-9_]+)\b", avail))
groups = [
    ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES", "SQ_WAVES"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA",
     "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INSTS_LDS",
     "SQ_INSTS_VALU", "SQ_INSTS_MFMA"],
    ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_VMEM" , "SQ_INSTS_SMEM",
     "SQ_ACTIVE_INST_FLAT"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TA_BUSY_avr", "GRBM_GUI_ACTIVE"],
]
table = {}
for gi, g in enumerate(groups):
    g = [c for c in g if c in names]
    if not g:
        continue
    d = os.path.join(out, "pass%d" % gi)
    r = subprocess.run(["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
                        "tools/x3_kernel_run.py"], capture_output=True, text=True, timeout=300)
    print("pass %d (%s): rc %d" % (gi, " ".join(g), r.returncode), flush=True)
    if r.returncode:
        print(r.stderr[-800:], flush=True)
        continue
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            kn = "pp" if "gemm_bf16x3_pp_kernel" in row["Kernel_Name"] else "bf16x3" if "gemm_bf16x3_fwd" in row["Kernel_Name"] \
                else "f32" if "gemm_f32" in row["Kernel_Name"] else None
            if kn:
                e = table.setdefault((kn, row["Counter_Name"]), [0.0, set()])
                e[0] += float(row["Counter_Value"])
                e[1].add(row["Dispatch_Id"])
with open(os.path.join(out, "x3_pmc_table.txt"), "w") as fh:
    for line in ["%-32s %16s %16s %16s   (per launch, 6144 x 728 x 728)" % ("counter", "planes x planes", "fp32 A (round 4)", "exact fp32")] + [
            "%-32s %16.0f %16.0f %16.0f" % (c, *(table.get((k, c), [0, {0}])[0] / max(1, len(table.get((k, c), [0, {0}])[1])) for k in ("pp", "bf16x3", "f32")))
            for c in sorted({c for _, c in table})]:
        print(line)
        fh.write(line + "\n")
