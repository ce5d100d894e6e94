#!/usr/bin/env python3
"""Dev tool (no GPU needed): compile a .hip file of csrc/ to gfx950 assembly and print, for every inner loop of the kernels
whose name contains PATTERN, the instruction stream as one letter per instruction -- M MFMA, v other vector, R ds_read,
D ds_write, G global/buffer load, S global/buffer store, w s_waitcnt, B s_barrier, s other scalar -- plus the register /
LDS / spill figures of the kernel.  Shows at a glance whether the MFMAs are interleaved with the vector work as the
sched_group_barrier pipeline asked.
usage: isa_stream.py spnet_amd/csrc/gemm_bf16x3.hip gemm_bf16x3_fwd [-DNAME=VALUE ...]"""
import os, re, subprocess, sys, tempfile, textwrap

src, pat, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"),
                "-I" + os.path.join(root, "spnet_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src] + extra,
               check=True, stderr=subprocess.DEVNULL)
text = open(out).read()


def letter(op):
    if op.startswith("v_mfma"): return "M"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "R"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "D"
    if re.match(r"(global|buffer|flat|scratch)_load", op): return "G"
    if re.match(r"(global|buffer|flat|scratch)_store", op): return "S"
    if op == "s_waitcnt": return "w"
    if op == "s_barrier": return "B"
    return "v" if op.startswith("v_") else "s"


for m in re.finditer(r"^(\S*%s\S*):[^\n]*\n(.*?)\.Lfunc_end" % re.escape(pat), text, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    print("== " + name)
    for key in ("vgpr_count", "vgpr_spill_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size"):
        mm = re.search(r"\.%s:\s+(\d+)" % key, text[text.index(".name:           " + name) - 1500:text.index(".name:           " + name) + 1500]) \
            if (".name:           " + name) in text else None
        if mm:
            print("   %s %s" % (key, mm.group(1)))
    heads = [i for i, l in enumerate(body) if "Loop Header" in l]
    for h in heads:
        label = body[h].split(":")[0]
        try:
            end = next(i for i in range(h + 1, len(body)) if "s_cbranch" in body[i] and label in body[i])
        except StopIteration:
            continue
        ops = [l.split()[0] for l in body[h + 1:end] if l.strip() and not l.strip().startswith((";", "."))]
        seq = "".join(letter(o) for o in ops)
        print("   loop %s: %d instructions, %d MFMA, %d vector, %d ds_read, %d ds_write, %d loads" % (
            label, len(ops), seq.count("M"), seq.count("v"), seq.count("R"), seq.count("D"), seq.count("G")))
        print(textwrap.indent("\n".join(textwrap.wrap(seq, 140)), "     "))
