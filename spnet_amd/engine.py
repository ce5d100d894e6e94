"""Execution engine of the SPNet hot path on one MI355X: a static plan of HIP kernel launches.

The network the reference assembles through Keras (stem spnet/models.py:315-342, Xception backbone
:357-359, Flatten+Dense head :378-388) has a fixed shape once the frame size and batch are known, so
instead of a tracing framework the engine lays the whole forward/backward out ONCE as an ordered list
of nodes, each owning its device buffers, and replays it: every `fwd()` / `bwd()` is a handful of
launches of the hand-written kernels in csrc/ through the C ABI (include/spnet_hip.h) on the current
HIP stream.  torch is used for device memory and streams only.

Data layout in HBM
  activations  NHWC fp32, one buffer per node output (kept for backward when training)
  parameters   ONE flat fp32 buffer `theta` (+ same-shaped grad / Adam m / Adam v): the ten
               l2-regularised kernels first (so the fused optimizer applies weight decay to a prefix),
               every tensor aligned to 64 floats; Keras layouts (conv HWIO, depthwise [3,3,C],
               pointwise [Cin,Cout], dense [in,out]) so checkpoints map 1:1 onto Keras weight names
  BN moving statistics in a second flat buffer (non-trainable, not touched by the optimizer)
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _lib as L

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
L2_COEF = 1e-4
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_RELU6 = 0, 1, 2, 3
# keras.applications.mobilenet.MobileNet(alpha=1): (pointwise filters, depthwise stride) of conv_dw/pw_1..13
MOBILENET_BLOCKS = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1), (512, 1),
                    (512, 1), (1024, 2), (1024, 1)]
K_MAJOR, OUT_MAJOR = 0, 1
WS_FLOATS = 48 * 1024 * 1024       # shared scratch (192 MiB), carved into regions (float offsets):
WS_GEMM = (0, 16 * 1024 * 1024)                       # split-K slabs
WS_DW = (16 * 1024 * 1024, 16 * 1024 * 1024)          # depthwise weight-gradient partials
WS_BNP = (32 * 1024 * 1024, 8 * 1024 * 1024)          # BatchNorm partial sums from GEMM / depthwise epilogues
WS_MISC = (40 * 1024 * 1024, 8 * 1024 * 1024)         # stand-alone BN reductions, small-conv partials
WS_GEMM2 = (48 * 1024 * 1024, 16 * 1024 * 1024)       # split-K slabs of the weight-gradient stream
WS_BNP2 = (64 * 1024 * 1024, 4 * 1024 * 1024)         # BatchNorm partial sums produced on the side stream
WS_TOTAL = 68 * 1024 * 1024
ALIGN = 64


def _capture(fn):
    """fn() captured as a hipGraph.  Python's cyclic garbage collector must not run while the stream is capturing: it
    may finalise garbage of EARLIER plans (an engine's graph, streams, events -- reachable only through reference cycles
    until then), and destroying a graph or an event is not a capturable operation: the process aborts inside the
    destructor (seen once in the GPU suite: `Fatal Python error: Aborted ... Garbage-collecting` under predict_step).
    Collect first, then keep the collector off until the capture has ended."""
    import gc
    torch.cuda.synchronize()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
    finally:
        if was_enabled:
            gc.enable()
    return g


_stream = L.current_stream       # raw handle of the current HIP stream (resolved once in _lib, public-API fallback)


class KernelTimer:
    """HIP-event timing of selected kernel families on the launch stream (bench.py's roofline leg).
    Records (family, start, end, algorithmic work) per launch; totals are read after a sync."""

    def __init__(self):
        self.records = []
        self.tags = {}
        self.keys = {}          # record index -> (a_major, b_major, stats, M, N, K) of a plain GEMM launch (autotuner)

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, family, start, work, tag=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((family, start, e, work))
        if tag is not None:
            self.tags[len(self.records) - 1] = tag

    def tagged(self):
        """{tag: (launches, total_ms, total_work)} of the launches that carry a tag (tools/gemm_table.py)."""
        out = {}
        for i, tag in self.tags.items():
            fam, s, e, w = self.records[i]
            n, ms, tw = out.get(tag, (0, 0.0, 0.0))
            out[tag] = (n + 1, ms + s.elapsed_time(e), tw + w)
        return out

    def totals(self):
        """{family: (launches, total_ms, total_work)} -- call after torch.cuda.synchronize()."""
        out = {}
        for fam, s, e, w in self.records:
            n, ms, tw = out.get(fam, (0, 0.0, 0.0))
            out[fam] = (n + 1, ms + s.elapsed_time(e), tw + w)
        return out


def xception_plan():
    """Keras-2.1.3 Xception(include_top=False) as a list of block descriptions."""
    blocks = [("entry",)]
    blocks.append(("strided", 2, 64, 128, 128, False, "conv2d_4", "batch_normalization_4"))
    blocks.append(("strided", 3, 128, 256, 256, True, "conv2d_5", "batch_normalization_5"))
    blocks.append(("strided", 4, 256, 728, 728, True, "conv2d_6", "batch_normalization_6"))
    for b in range(5, 13):
        blocks.append(("middle", b, 728))
    blocks.append(("strided", 13, 728, 728, 1024, True, "conv2d_7", "batch_normalization_7"))
    blocks.append(("exit", 14, 1024, 1536, 2048))
    return blocks


def backbone_out_hw(H, W):
    h, w = H // 2, W // 2
    h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    h, w = h - 2, w - 2
    for _ in range(4):
        h, w = (h + 1) // 2, (w + 1) // 2
    return h, w


def irv2_program():
    """keras.applications.inception_resnet_v2.InceptionResNetV2(include_top=False) (Keras 2.1.3; call site
    spnet/models.py:357-359 with cf.basemodel = 'InceptionResNetV2') as a flat list of operations over named tensors:
      ("conv", conv_name, bn_name | None, src, dst, cin, cout, (kh, kw), stride, same, relu, bias)
      ("maxpool", src, dst) | ("avgpool", src, dst) | ("concat", [srcs], dst, name) | ("resadd", x, up, dst, scale, relu)
    Unnamed conv2d_bn layers get Keras' creation-order names conv2d_4.. / batch_normalization_4.. (the stem owns 1-3);
    their BatchNormalization has scale=False (no gamma)."""
    ops, ctr = [], [3]

    def conv(src, dst, cin, cout, k, stride=1, same=True, relu=True, name=None, bias=False):
        kk = (k, k) if isinstance(k, int) else tuple(k)
        if name is None:
            ctr[0] += 1
            cname, bname = "conv2d_%d" % ctr[0], "batch_normalization_%d" % ctr[0]
        else:
            cname, bname = name, name + "_bn"
        ops.append(("conv", cname, None if bias else bname, src, dst, cin, cout, kk, stride, same, relu, bias))

    # stem
    conv("in", "x", 3, 32, 3, 2, False)
    conv("x", "x", 32, 32, 3, 1, False)
    conv("x", "x", 32, 64, 3)
    ops.append(("maxpool", "x", "x"))
    conv("x", "x", 64, 80, 1, 1, False)
    conv("x", "x", 80, 192, 3, 1, False)
    ops.append(("maxpool", "x", "x"))
    # mixed_5b (Inception-A)
    conv("x", "b0", 192, 96, 1)
    conv("x", "b1", 192, 48, 1)
    conv("b1", "b1", 48, 64, 5)
    conv("x", "b2", 192, 64, 1)
    conv("b2", "b2", 64, 96, 3)
    conv("b2", "b2", 96, 96, 3)
    ops.append(("avgpool", "x", "bp"))
    conv("bp", "bp", 192, 64, 1)
    ops.append(("concat", ["b0", "b1", "b2", "bp"], "x", "mixed_5b"))

    def block(kind, idx, c, scale, relu=True):
        if kind == "block35":
            conv("x", "b0", c, 32, 1)
            conv("x", "b1", c, 32, 1)
            conv("b1", "b1", 32, 32, 3)
            conv("x", "b2", c, 32, 1)
            conv("b2", "b2", 32, 48, 3)
            conv("b2", "b2", 48, 64, 3)
            br, cm = ["b0", "b1", "b2"], 128
        elif kind == "block17":
            conv("x", "b0", c, 192, 1)
            conv("x", "b1", c, 128, 1)
            conv("b1", "b1", 128, 160, (1, 7))
            conv("b1", "b1", 160, 192, (7, 1))
            br, cm = ["b0", "b1"], 384
        else:
            conv("x", "b0", c, 192, 1)
            conv("x", "b1", c, 192, 1)
            conv("b1", "b1", 192, 224, (1, 3))
            conv("b1", "b1", 224, 256, (3, 1))
            br, cm = ["b0", "b1"], 448
        name = "%s_%d" % (kind, idx)
        ops.append(("concat", br, "m", name + "_mixed"))
        conv("m", "up", cm, c, 1, relu=False, name=name + "_conv", bias=True)
        ops.append(("resadd", "x", "up", "x", scale, relu))

    for i in range(1, 11):
        block("block35", i, 320, 0.17)
    # mixed_6a (Reduction-A)
    conv("x", "b0", 320, 384, 3, 2, False)
    conv("x", "b1", 320, 256, 1)
    conv("b1", "b1", 256, 256, 3)
    conv("b1", "b1", 256, 384, 3, 2, False)
    ops.append(("maxpool", "x", "bp"))
    ops.append(("concat", ["b0", "b1", "bp"], "x", "mixed_6a"))
    for i in range(1, 21):
        block("block17", i, 1088, 0.1)
    # mixed_7a (Reduction-B)
    conv("x", "b0", 1088, 256, 1)
    conv("b0", "b0", 256, 384, 3, 2, False)
    conv("x", "b1", 1088, 256, 1)
    conv("b1", "b1", 256, 288, 3, 2, False)
    conv("x", "b2", 1088, 256, 1)
    conv("b2", "b2", 256, 288, 3)
    conv("b2", "b2", 288, 320, 3, 2, False)
    ops.append(("maxpool", "x", "bp"))
    ops.append(("concat", ["b0", "b1", "b2", "bp"], "x", "mixed_7a"))
    for i in range(1, 10):
        block("block8", i, 2080, 0.2)
    block("block8", 10, 2080, 1.0, relu=False)
    conv("x", "x", 2080, 1536, 1, name="conv_7b")
    return ops


def irv2_sibling_groups():
    """Groups of conv2d_bn layers of the program that are 1x1 / stride-1 convolutions of ONE tensor (the first layer of
    every branch of an inception block reads the block input): [[conv names], ...] with their BatchNormalization names.
    The executor runs such a group as one GEMM over the concatenated kernels and one BatchNormalization over the
    concatenated channels (IRv2Backbone); the parameter layout keeps the group's beta / moving statistics adjacent."""
    ver, readers = {}, {}
    for op in irv2_program():
        kind = op[0]
        if kind == "conv":
            _, cname, bname, src, dst, cin, cout, kk, stride, same, relu, bias = op
            if kk == (1, 1) and stride == 1 and not bias:
                readers.setdefault((src, ver.get(src, 0)), []).append((cname, bname, cout))
            ver[dst] = ver.get(dst, 0) + 1
        elif kind in ("maxpool", "avgpool"):
            ver[op[2]] = ver.get(op[2], 0) + 1
        elif kind == "concat":
            ver[op[2]] = ver.get(op[2], 0) + 1
        else:
            ver[op[3]] = ver.get(op[3], 0) + 1
    return [g for g in readers.values() if len(g) >= 2]


def param_packs(backbone):
    """Lists of parameter names that must sit back to back in the flat buffers (no alignment padding between them), so
    that a slice over the first one's offset spans them all."""
    if backbone != "InceptionResNetV2":
        return []
    packs = []
    for g in irv2_sibling_groups():
        for suffix in ("/beta", "/moving_mean", "/moving_variance"):
            packs.append([bname + suffix for _, bname, _ in g])
    return packs


def irv2_out_hw(H, W):
    h, w = H // 2, W // 2
    for k, s_ in ((3, 2), (3, 1), (3, 2), (3, 1), (3, 2), (3, 2), (3, 2)):     # valid convs / pools that shrink the plane
        h, w = (h - k) // s_ + 1, (w - k) // s_ + 1
    return h, w


def mobilenet_out_hw(H, W):
    h, w = H // 2, W // 2
    h, w = (h + 1) // 2, (w + 1) // 2
    for _, s_ in MOBILENET_BLOCKS:
        if s_ == 2:
            h, w = (h + 1) // 2, (w + 1) // 2
    return h, w


def param_specs(H, W, n_out=576, backbone="Xception"):
    """[(keras_name, shape, trainable, l2)] -- l2-regularised kernels first (run-log line 98 of the
    reference: conv2d_1..3, block1_conv1, block1_conv2, conv2d_4..7, FinalOutput).  backbone 'MobileNet'
    (spnet/models.py:346-355): conv1 + 13 depthwise/pointwise pairs, l2 on the Conv2D / Dense kernels."""
    specs = []

    def bn(name, c):
        specs.append((name + "/gamma", (c,), True, False))
        specs.append((name + "/beta", (c,), True, False))
        specs.append((name + "/moving_mean", (c,), False, False))
        specs.append((name + "/moving_variance", (c,), False, False))

    specs.append(("conv2d_1/kernel", (3, 3, 1, 3), True, True))
    bn("batch_normalization_1", 3)
    specs.append(("conv2d_2/kernel", (3, 3, 3, 3), True, True))
    bn("batch_normalization_2", 3)
    specs.append(("conv2d_3/kernel", (3, 3, 3, 3), True, True))
    bn("batch_normalization_3", 3)
    if backbone == "InceptionResNetV2":
        for op in irv2_program():
            if op[0] != "conv":
                continue
            _, cname, bname, _, _, cin, cout, kk, _, _, _, bias = op
            specs.append((cname + "/kernel", (kk[0], kk[1], cin, cout), True, True))
            if bias:
                specs.append((cname + "/bias", (cout,), True, False))
            else:                  # BatchNormalization(scale=False)
                specs.append((bname + "/beta", (cout,), True, False))
                specs.append((bname + "/moving_mean", (cout,), False, False))
                specs.append((bname + "/moving_variance", (cout,), False, False))
        h, w = irv2_out_hw(H, W)
        if h < 1 or w < 1:
            raise ValueError("frames of %dx%d are too small for InceptionResNetV2 (needs >= 150x150)" % (H, W))
        specs.append(("FinalOutput/kernel", (h * w * 1536, n_out), True, True))
        specs.append(("FinalOutput/bias", (n_out,), True, False))
        return specs
    if backbone == "MobileNet":
        specs.append(("conv1/kernel", (3, 3, 3, 32), True, True))
        bn("conv1_bn", 32)
        cin = 32
        for i, (cout, _) in enumerate(MOBILENET_BLOCKS, 1):
            specs.append(("conv_dw_%d/depthwise_kernel" % i, (3, 3, cin), True, False))
            bn("conv_dw_%d_bn" % i, cin)
            specs.append(("conv_pw_%d/kernel" % i, (1, 1, cin, cout), True, True))
            bn("conv_pw_%d_bn" % i, cout)
            cin = cout
        h, w = mobilenet_out_hw(H, W)
        specs.append(("FinalOutput/kernel", (h * w * cin, n_out), True, True))
        specs.append(("FinalOutput/bias", (n_out,), True, False))
        return specs
    specs.append(("block1_conv1/kernel", (3, 3, 3, 32), True, True))
    bn("block1_conv1_bn", 32)
    specs.append(("block1_conv2/kernel", (3, 3, 32, 64), True, True))
    bn("block1_conv2_bn", 64)

    def sep(name, cin, cout):
        specs.append((name + "/depthwise_kernel", (3, 3, cin), True, False))
        specs.append((name + "/pointwise_kernel", (cin, cout), True, False))
        bn(name + "_bn", cout)

    for blk in xception_plan():
        if blk[0] == "strided":
            _, b, cin, c1, c2, _, cn, bnn = blk
            specs.append((cn + "/kernel", (1, 1, cin, c2), True, True))
            bn(bnn, c2)
            sep("block%d_sepconv1" % b, cin, c1)
            sep("block%d_sepconv2" % b, c1, c2)
        elif blk[0] == "middle":
            for k in (1, 2, 3):
                sep("block%d_sepconv%d" % (blk[1], k), 728, 728)
        elif blk[0] == "exit":
            sep("block14_sepconv1", blk[2], blk[3])
            sep("block14_sepconv2", blk[3], blk[4])
    h, w = backbone_out_hw(H, W)
    specs.append(("FinalOutput/kernel", (h * w * 2048, n_out), True, True))
    specs.append(("FinalOutput/bias", (n_out,), True, False))
    return specs


def param_layout(H, W, n_out=576, backbone="Xception"):
    """Offsets of every tensor in the flat parameter / statistics buffers (pure: no device needed).
    Flat layout: [l2-regularised kernels, the Dense head's first] [all depthwise kernels] [everything else in
    forward order].  The l2 set is a prefix (weight decay inside the fused optimizer); the head kernel (73 % of
    the bytes, produced first in backward) is one contiguous range at offset 0; the depthwise gradients, which
    one batched reduction finishes at the very END of backward, sit together in front of the forward-ordered
    rest, so that suffixes of the buffer are complete -- and can be all-reduced -- as backward walks up the net.
    Returns {p_off, s_off: name -> (offset, n, shape); l2_n, rest_lo, n_theta, n_stats, spec_order}."""
    specs = param_specs(H, W, n_out, backbone)
    tr = [s for s in specs if s[2]]
    l2 = sorted((s for s in tr if s[3]), key=lambda s: s[0] != "FinalOutput/kernel")
    dwk = [s for s in tr if not s[3] and s[0].endswith("depthwise_kernel")]
    order = l2 + dwk + [s for s in tr if not s[3] and not s[0].endswith("depthwise_kernel")]
    # (packs: names that sit back to back with no padding between them -- param_packs; a pack is placed where its
    # first member comes up in `order` and starts on an aligned offset)
    pack_of = {n: tuple(p) for p in param_packs(backbone) for n in p}
    shape_of = {s[0]: s[1] for s in specs}
    off = 0
    p_off = OrderedDict()
    l2_n = rest_lo = 0
    for name, shape, _, l2 in order:
        if name in p_off:
            continue                 # placed with its pack
        for member in pack_of.get(name, (name,)):
            mshape = shape_of[member]
            n = int(np.prod(mshape))
            if member in pack_of and (n & 3):
                raise ValueError("packed parameters must be multiples of 4 floats: %s" % member)
            p_off[member] = (off, n, mshape)
            off += n if member in pack_of else (n + ALIGN - 1) // ALIGN * ALIGN
        off = (off + ALIGN - 1) // ALIGN * ALIGN
        if l2:
            l2_n = off               # prefix (incl. alignment padding, which stays zero)
        if name.endswith("depthwise_kernel"):
            rest_lo = off            # first offset behind the depthwise kernels
    if not any(n.endswith("depthwise_kernel") for n, _, _, _ in order):
        rest_lo = l2_n               # no depthwise kernels (InceptionResNetV2): the rest starts behind the l2 prefix
    n_theta = off
    so = 0
    s_off = OrderedDict()
    for name, shape, trn, _ in specs:
        if not trn and name not in s_off:
            for member in pack_of.get(name, (name,)):
                mshape = shape_of[member]
                n = int(np.prod(mshape))
                s_off[member] = (so, n, mshape)
                so += n if member in pack_of else (n + ALIGN - 1) // ALIGN * ALIGN
            so = (so + ALIGN - 1) // ALIGN * ALIGN
    return dict(p_off=p_off, s_off=s_off, l2_n=l2_n, rest_lo=rest_lo, n_theta=n_theta, n_stats=so,
                spec_order=[s[0] for s in specs])


def xception_node_pnames():
    """Forward-ordered (key, parameter-name prefixes, is_middle_block) of the nodes Engine._build_graph creates for the
    Xception plan -- what plan_grad_buckets needs of them (a CPU test plans the 384 x 512 engine's all-reduce from
    this; tests/test_engine_gpu.py checks it against the live engine's nodes)."""
    out = [("conv2d_1", ["conv2d_1"], False)]
    for k in (1, 2, 3):
        if k > 1:
            out.append(("conv2d_%d" % k, ["conv2d_%d" % k], False))
        out.append(("batch_normalization_%d" % k, ["batch_normalization_%d" % k], False))
    out.append(("dropout", [], False))
    out += [("block1_conv1", ["block1_conv1"], False), ("block1_conv1_bn", ["block1_conv1_bn"], False),
            ("block1_conv2", ["block1_conv2"], False), ("block1_conv2_bn", ["block1_conv2_bn"], False)]
    for blk in xception_plan():
        if blk[0] == "strided":
            b = blk[1]
            out.append(("block%d" % b, [blk[6], blk[7]] + [n for k in (1, 2) for n in ("block%d_sepconv%d" % (b, k),
                                                                                        "block%d_sepconv%d_bn" % (b, k))], False))
        elif blk[0] == "middle":
            b = blk[1]
            out.append(("block%d" % b, [n for k in (1, 2, 3) for n in ("block%d_sepconv%d" % (b, k),
                                                                        "block%d_sepconv%d_bn" % (b, k))], True))
        elif blk[0] == "exit":
            out.append(("block14", ["block14_sepconv1", "block14_sepconv1_bn", "block14_sepconv2", "block14_sepconv2_bn"], False))
    out.append(("FinalOutput", ["FinalOutput"], False))
    return out


def plan_grad_buckets(p_off, rest_lo, n_theta, nodes, bucket_bytes=32 << 20):
    """Plan of the gradient all-reduce (~32 MB buckets in reverse layer order); pure.

    nodes: forward-ordered [(key, parameter-name prefixes, is_middle_block)], the last one owning the Dense head.
    Returns (buckets, tail): buckets = [(lo, hi, trigger key)] in launch order -- float ranges of the flat
    gradient that are COMPLETE once the node `trigger` has been back-propagated -- and tail = [(lo, hi)], what is
    only complete when backward ends.  The Dense-head kernel (produced first, 73 % of the bytes) goes first, cut
    into bucket-sized pieces; then suffixes of the forward-ordered region, cut at node boundaries.  The
    middle-flow pointwise gradients are produced by ONE deferred batched launch once block 5 is done
    (flush_deferred_wgrads), so every bucket inside the middle flow is triggered by the FIRST middle block.  The tail
    holds the small l2 kernels, all depthwise kernels (their batched reduction is the last launch of backward) and
    whatever forward-ordered parameters precede the first bucket cut."""
    per = max(int(bucket_bytes) // 4, ALIGN)
    head = nodes[-1][0]
    hoff, hn, _ = p_off["FinalOutput/kernel"]
    hlo, hhi = hoff, hoff + hn
    buckets = [(lo, min(lo + per, hhi), head) for lo in range(hlo, hhi, per)]
    first = {}                                   # node -> lowest offset of its parameters in the rest region
    for key, pnames, _ in nodes:
        offs = [off for name, (off, n, _s) in p_off.items() if off >= rest_lo and name.split("/")[0] in pnames]
        if offs:
            first[key] = min(offs)
    first_middle = next((key for key, _, mid in nodes if mid), None)
    owners = [(key, mid) for key, _, mid in nodes if key in first]
    cur_hi = n_theta
    for i in range(len(owners) - 1, -1, -1):
        key, mid = owners[i]
        if first[key] >= cur_hi:
            continue
        # cut here when the bucket is full, and on both edges of the middle flow (what lies behind it is complete
        # long before the deferred launch; what lies in front of it is not touched by it)
        edge = i > 0 and owners[i - 1][1] != mid
        if (cur_hi - first[key]) >= per or edge:
            buckets.append((first[key], cur_hi, first_middle if mid else key))
            cur_hi = first[key]
    tail = [(hhi, cur_hi)] if cur_hi > hhi else []
    if hlo > 0:
        tail.insert(0, (0, hlo))
    return buckets, tail


def _glorot_fans(name, shape):
    if name.endswith("depthwise_kernel"):
        return shape[2] * 9, 9
    if len(shape) == 4:
        rf = shape[0] * shape[1]
        return shape[2] * rf, shape[3] * rf
    return shape[0], shape[1]


class Engine:
    def __init__(self, H, W, batch, n_out=576, device="cuda:0", loss_type="same", seed=0,
                 train=True, adam_eps=1e-7, share_from=None, rank=0, sigmoid_cols=None, backbone="Xception",
                 pointwise="bf16x3", x3_min_tiles=192, fuse_dw_bwd=True, early_head=True):
        """rank: data-parallel rank, mixed into the initial dropout seed so that replicas draw different masks.
        pointwise: which kernel runs the forward and data-gradient GEMMs of the pointwise (1x1) convolutions with >= 256
        output columns -- "bf16x3" (csrc/gemm_bf16x3.hip: fp32 operands as three bf16 pieces on the bf16 matrix cores, fp32
        accumulation, fp32-accurate; the default since round 4) or "f32" (the k-ordered fp32 MFMA chain of spnet_gemm_f32
        for every GEMM, as in rounds 1-3).  x3_min_tiles: a layer goes to the bf16x3 kernel only if its GEMM has at least this
        many 96 x 96 tiles (192: measured, see Pointwise; the parity tests pass 0 so that small test plans run it too)."""
        if pointwise not in ("bf16x3", "f32"):
            raise ValueError("pointwise must be 'bf16x3' or 'f32'")
        self._pointwise = pointwise     # fixed for the life of the plan: the operand buffers depend on it (`pointwise` property)
        # the depthwise backward inside the pointwise data-gradient GEMM's epilogue where a 192-pixel tile holds whole
        # images AND the 192 x 96 tiles fill the chip (12 x 16 planes at batch 32: the middle flow and block 13;
        # spnet_gemm_bf16x3_pp_dwbwd) -- False: the two launches everywhere.  Measured (tools/dwfuse_bwd_time.py,
        # profiles/r05_dwfuse_bwd.txt): 41.3 us against 48.4 us for the pair on the middle-flow layer, 45.5 against 56.5 with
        # the residual-branch gradient; the 6 x 8 planes of the exit flow (128 tiles) lose (74 against 60 us) and keep the pair.
        self.fuse_dw_bwd = bool(fuse_dw_bwd)
        # inference counterpart: the consumer's depthwise forward inside the producer's forward GEMM (sepconv -> BN -> relu
        # -> sepconv chains of the middle flow / block 13-14 on 12 x 16 | 6 x 8 planes; spnet_gemm_bf16x3_pp_dwfwd)
        self.fuse_dw_fwd = bool(fuse_dw_bwd)
        # the Dense head's weight gradient on the weight-gradient stream and its optimizer step right behind it, underneath
        # the backbone's backward (adam_head_early) -- False: both on the dependency chain as in rounds 1-4 (A/B only)
        self.early_head = bool(early_head)
        self.x3_min_tiles = int(x3_min_tiles)
        self.fuse_min_tiles = 256 if self.x3_min_tiles else 0       # (x3_min_tiles = 0, the parity tests: small plans fuse too)
        if not torch.cuda.is_available():
            raise RuntimeError("spnet_amd.Engine needs a HIP device (no CPU fallback)")
        self.H, self.W, self.B, self.n_out = int(H), int(W), int(batch), int(n_out)
        # (start, step) of the output columns that pass through a sigmoid: the 'compound' head of the reference
        # (models.py:379-386) = one dense layer with sigmoid 'noobj' columns once InterleaveColumns has re-ordered them
        self.sigmoid_cols = sigmoid_cols
        if backbone not in ("Xception", "MobileNet", "InceptionResNetV2"):
            raise NotImplementedError("backbone %r: this build implements Xception, MobileNet and InceptionResNetV2"
                                      % (backbone,))
        self.backbone = backbone
        self.dev = torch.device(device)
        self.loss_type = loss_type
        self.train_capable = bool(train)
        self.adam_eps = adam_eps
        # Optimizer iteration count and dropout seed live beside theta/m/v: every plan over one weight set
        # (other batch sizes, the inference plans) advances the SAME counters, so a second fit() with another
        # batch size continues Adam's bias correction and the dropout seed sequence instead of restarting them.
        self._opt = share_from._opt if share_from is not None else {"t": 0, "drop_seed": 12345 + 7919 * int(rank)}
        self.prof = None                # KernelTimer or None
        self._pw_layers = []            # every Pointwise of this plan (their bf16x3 weight planes: _build_planes)
        self.deferred_wgrads = []       # (x, dy, gw, cin, cout, M) of layers whose dW waits for the batched launch
        self.dw_reduce_jobs = []        # (partials, grad, rows, 9*C) of every depthwise layer
        self._dw_reduce_table = None
        self._wgrad_table = None
        self._first_middle = None
        # The three run-time switches of the engine (DESIGN.md section 2 lists them with the tests that run the non-default
        # branch): SPNET_OVERLAP_WGRAD=0 -- weight gradients on the main stream instead of a side stream joined before
        # Adam; SPNET_TRAIN_GRAPH=1 -- the single-GPU train step as a hipGraph; SPNET_GEMM_TILES=0 (_load_tile_table) --
        # the library's own tile choice instead of the in-step autotuned table.
        self.overlap_wgrad = os.environ.get("SPNET_OVERLAP_WGRAD", "1") != "0"
        # Optional: capture the single-GPU train step as a hipGraph after one eager step.  Off by default:
        # measured on MI355X / ROCm 7.2 the replay of this ~400-node two-stream graph takes 13.8 ms against
        # 13.4 ms for the eager launches (single stream: 13.5 either way) -- the host enqueues a step in
        # 4.3 ms and runs ahead of the GPU, so launch overhead is not what limits the step.
        self.use_graph = os.environ.get("SPNET_TRAIN_GRAPH", "0") == "1"
        torch.cuda.set_device(self.dev)
        # Inference coefficients (scale|shift from the moving statistics, one tiny kernel per BatchNorm) are
        # recomputed only when the weights or statistics changed since this plan last did: [version] is shared
        # by all plans over one weight set and bumped by every training forward / optimizer step / load.
        self._wver = share_from._wver if share_from is not None else [0]
        # W^T of every pointwise kernel (name -> tensor), shared by all plans; [_tver] counts changes of theta,
        # [_wT_ver] is the value of _tver the transposes were last refreshed at.
        self._wT = share_from._wT if share_from is not None else {}
        self._tver = share_from._tver if share_from is not None else [0]
        self._wT_ver = share_from._wT_ver if share_from is not None else [-1]
        self._wT_jobs = None
        # bf16x3 operand planes of the pointwise kernels (name -> (forward planes, data-gradient planes)), shared by all
        # plans over one weight set; [_planes_ver] = the value of _tver they were last split at
        self._planes = share_from._planes if share_from is not None else {}
        self._planes_ver = share_from._planes_ver if share_from is not None else [-1]
        self._planes_gen = share_from._planes_gen if share_from is not None else [0]      # bumped by every allocation
        self._planes_jobs = None
        self._igraph = None             # captured inference forward (predict_step)
        self._coeff_ver = -1
        self._infer_fresh = True
        if share_from is not None:      # second plan (other batch size / inference) over the SAME weights
            for a in ("p_off", "s_off", "l2_n", "rest_lo", "n_theta", "theta", "stats", "spec_order"):
                setattr(self, a, getattr(share_from, a))
            if self.train_capable:
                for a in ("grad", "m", "v"):
                    setattr(self, a, getattr(share_from, a))
        else:
            self._build_params(seed)
        self.update_mask = None         # optional flat 0/1 tensor: frozen parameters are skipped by Adam
        self._build_graph()
        self._build_planes()

    @property
    def pointwise(self):
        """"bf16x3" | "f32": which kernels run the pointwise GEMMs.  Read-only -- a bf16x3 plan keeps the operands of those
        GEMMs as bf16 planes written by their producers (SepConvBN.x3p), there is no fp32 copy to fall back on; build
        another Engine (share_from=...) for the other arithmetic."""
        return self._pointwise

    def new_planes(self, R, K):
        """zeroed bf16x3 plane set of an [R][K] matrix (csrc/x3t.h): the pad rows / columns are never written"""
        return torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(R, K)), dtype=torch.int16, device=self.dev)

    @property
    def t(self):
        """optimizer iterations done (shared by all plans over this weight set)"""
        return self._opt["t"]

    @t.setter
    def t(self, v):
        self._opt["t"] = int(v)

    @property
    def drop_seed(self):
        return self._opt["drop_seed"]

    @drop_seed.setter
    def drop_seed(self, v):
        self._opt["drop_seed"] = int(v) & 0xFFFFFFFF

    # ------------------------------------------------------------------ parameters
    def _build_params(self, seed):
        lay = param_layout(self.H, self.W, self.n_out, self.backbone)
        for k in ("p_off", "s_off", "l2_n", "rest_lo", "n_theta", "spec_order"):
            setattr(self, k, lay[k])
        s_off = lay["n_stats"]
        z = lambda n: torch.zeros(n, device=self.dev, dtype=torch.float32)
        self.theta = z(self.n_theta)
        self.stats = z(s_off)
        if self.train_capable:
            self.grad, self.m, self.v = z(self.n_theta), z(self.n_theta), z(self.n_theta)
        self.init_weights(seed)

    def init_weights(self, seed=0):
        """Keras defaults: glorot_uniform kernels, zeros bias/beta, ones gamma, moving stats 0/1."""
        self._wver[0] += 1
        self._tver[0] += 1
        g = torch.Generator().manual_seed(seed)
        host = torch.zeros(self.n_theta, dtype=torch.float32)
        for name, (off, n, shape) in self.p_off.items():
            if name.endswith("/gamma"):
                host[off:off + n] = 1.0
            elif name.endswith("kernel"):
                fi, fo = _glorot_fans(name, shape)
                lim = math.sqrt(6.0 / (fi + fo))
                host[off:off + n] = ((torch.rand(n, generator=g, dtype=torch.float64) * 2 - 1) * lim).float()
        self.theta.copy_(host)
        hs = torch.zeros(self.stats.numel(), dtype=torch.float32)
        for name, (off, n, _) in self.s_off.items():
            if name.endswith("moving_variance"):
                hs[off:off + n] = 1.0
        self.stats.copy_(hs)
        if self.train_capable:
            self.m.zero_()
            self.v.zero_()
        self.t = 0

    def P(self, name):
        off, n, shape = self.p_off[name]
        return self.theta[off:off + n]

    def G(self, name):
        off, n, shape = self.p_off[name]
        return self.grad[off:off + n]

    def S(self, name):
        off, n, shape = self.s_off[name]
        return self.stats[off:off + n]

    def state_dict(self):
        """name -> CPU tensor in Keras layout (trainable + moving statistics)."""
        out = OrderedDict()
        for name in self.spec_order:
            if name in self.p_off:
                off, n, shape = self.p_off[name]
                out[name] = self.theta[off:off + n].reshape(shape).cpu().clone()
            else:
                off, n, shape = self.s_off[name]
                out[name] = self.stats[off:off + n].reshape(shape).cpu().clone()
        return out

    def load_state_dict(self, sd):
        self._wver[0] += 1
        self._tver[0] += 1
        for name in self.spec_order:
            t = torch.as_tensor(np.asarray(sd[name]), dtype=torch.float32).reshape(-1)
            if name in self.p_off:
                off, n, _ = self.p_off[name]
                dst = self.theta
            else:
                off, n, _ = self.s_off[name]
                dst = self.stats
            if t.numel() != n:
                raise ValueError("shape mismatch for %s: got %d values, need %d" % (name, t.numel(), n))
            dst[off:off + n].copy_(t.to(self.dev))

    def grad_dict(self):
        return OrderedDict((name, self.grad[off:off + n].reshape(shape).cpu().clone())
                           for name, (off, n, shape) in self.p_off.items())

    # ------------------------------------------------------------------ graph construction
    def new(self, *shape):
        return torch.empty(shape, device=self.dev, dtype=torch.float32)

    def ws_ptr(self, region):
        return self.ws.data_ptr() + 4 * region[0]

    def _build_graph(self):
        B, H, W = self.B, self.H, self.W
        tr = self.train_capable
        self.ws = self.new(WS_TOTAL)
        # weight-gradient GEMMs run on a second HIP stream, concurrently with the data-gradient chain
        self.wgrad_stream = torch.cuda.Stream(device=self.dev) if (tr and self.overlap_wgrad) else None
        self.small = self.new(8 * 4096)          # BN scale/shift + backward coefficient scratch
        self.nodes = []
        self.x_in = self.new(B, H, W, 1)

        # ---- stem (spnet/models.py:321-340)
        H2, W2 = H // 2, W // 2
        n = self.nodes
        sh = StemHead(self, self.x_in, "conv2d_1"); n.append(sh)      # conv2d_1 + both average pools, one kernel
        a1 = BatchNorm(self, sh.y, 3, "batch_normalization_1", ACT_LRELU); n.append(a1)
        c2 = SmallConv(self, a1.y, 3, 3, 1, True, "conv2d_2"); n.append(c2)
        a2 = BatchNorm(self, c2.y, 3, "batch_normalization_2", ACT_LRELU); n.append(a2)
        c3 = SmallConv(self, a2.y, 3, 3, 1, True, "conv2d_3"); n.append(c3)
        s = BatchNorm(self, c3.y, 3, "batch_normalization_3", ACT_NONE, residual=sh.px, res_bcast=True); n.append(s)
        d = Dropout(self, s.y, 0.1); n.append(d)
        self.stem_out = d.y
        if self.backbone == "MobileNet":
            return self._build_mobilenet(d.y)
        if self.backbone == "InceptionResNetV2":
            net = IRv2Backbone(self, d.y); n.append(net)
            self.backbone_out = net.y
            return self._build_head(net.y)
        # ---- Xception entry flow, block 1
        e1 = SmallConv(self, d.y, 3, 32, 2, False, "block1_conv1"); n.append(e1)
        b1 = BatchNorm(self, e1.y, 32, "block1_conv1_bn", ACT_RELU); n.append(b1)
        e2 = Conv3x3Gemm(self, b1.y, 32, 64, "block1_conv2"); n.append(e2)
        b2 = BatchNorm(self, e2.y, 64, "block1_conv2_bn", ACT_RELU); n.append(b2)
        x = b2.y
        prev_middle = None
        for blk in xception_plan():
            if blk[0] == "strided":
                _, b, cin, c1_, c2_, first_relu, cn, bnn = blk
                node = StridedBlock(self, x, b, cin, c1_, c2_, first_relu, cn, bnn)
            elif blk[0] == "middle":
                node = MiddleBlock(self, x, blk[1], blk[2], prev=prev_middle)
                prev_middle = node
                if self._first_middle is None:
                    self._first_middle = node
            elif blk[0] == "exit":
                node = ExitBlock(self, x, blk[2], blk[3], blk[4])
            else:
                continue
            n.append(node)
            prev_middle = node if blk[0] == "middle" else None
            x = node.y
        self.backbone_out = x
        self._build_head(x)

    def _build_mobilenet(self, x):
        """keras.applications.mobilenet.MobileNet(include_top=False) behind the stem (spnet/models.py:346-355)."""
        n = self.nodes
        pad = PadSame(self, x, 3, 2); n.append(pad)
        c1 = SmallConv(self, pad.y, 3, 32, 2, False, "conv1"); n.append(c1)
        b1 = BatchNorm(self, c1.y, 32, "conv1_bn", ACT_RELU6); n.append(b1)
        x, cin = b1.y, 32
        for i, (cout, stride) in enumerate(MOBILENET_BLOCKS, 1):
            blk = MobileBlock(self, x, i, cin, cout, stride); n.append(blk)
            x, cin = blk.y, cout
        self.backbone_out = x
        self._build_head(x)

    def _build_head(self, x):
        tr = self.train_capable
        B = self.B
        head = Dense(self, x, self.n_out, "FinalOutput"); self.nodes.append(head)
        self.out = head.y
        if tr:
            self.y_true = self.new(B, self.n_out)
            self.dout = self.new(B, self.n_out)
            self.loss_parts = self.new(B, 5)
            self.loss_out = self.new(8)            # center,size,angle,noobj,class,total, l2, total+l2
            self.sq_scratch = self.new(2 * 2048)        # sum-of-squares partials of the optimizer's two ranges
            off, n, _ = self.p_off["FinalOutput/kernel"]
            self._head_hi = n if (off == 0 and n % 4 == 0) else 0
            self._head_parts = int(L.spnet_adam_parts(self._head_hi))
            self._rest_parts = int(L.spnet_adam_parts(self.n_theta - self._head_hi))
            self._opt_stream = None
            self._head_done = False
            # per-step scalars the kernels read from device memory: [lr_t (f32), dropout seed (u32)]
            self.step_params = torch.zeros(4, device=self.dev, dtype=torch.int32)
            self._step_upload = L.AsyncUploader(self.dev, depth=8)
            self.lr_ptr = self.step_params.data_ptr()
            self.seed_ptr = self.step_params.data_ptr() + 4
            self._graph = None
            self._graph_warm = 0

    # ------------------------------------------------------------------ execution
    def forward(self, X=None, training=False):
        """X: [B,H,W,1] fp32 tensor on the device (or None if self.x_in was filled in place)."""
        if training and not self.train_capable:
            raise RuntimeError("engine was built with train=False")
        if X is not None:
            self.x_in.copy_(X.reshape(self.x_in.shape))
        if training:
            self._wver[0] += 1              # moving statistics move; this plan's scale|shift now hold batch statistics
            self._coeff_ver = -1
        else:
            self._infer_fresh = self._coeff_ver != self._wver[0]
        self.refresh_planes()               # (on THIS stream, before any branch reads them)
        for node in self.nodes:
            node.fwd(training)
        if self.sigmoid_cols is not None:
            L.spnet_selective_sigmoid(L.ptr(self.out), None, self.B, self.n_out, self.sigmoid_cols[0],
                                      self.sigmoid_cols[1], 0, _stream())
        if not training:
            self._coeff_ver = self._wver[0]
        return self.out

    def predict_step(self, use_graph=None):
        """Inference forward of self.x_in -> self.out as ONE hipGraph replay (BASELINE configs[4]: predict_spnet.py's
        model.predict loop).  The plan is static, so the ~110 launches of a forward are captured once per plan; the
        BatchNorm inference coefficients are refreshed by an eager forward whenever the weights or moving statistics
        changed since this plan last ran (the graph itself holds no weight-dependent host decisions).
        use_graph=False keeps the eager launches."""
        if use_graph is False or self.prof is not None:
            return self.forward(None, training=False)
        if self._coeff_ver != self._wver[0] or self._igraph is None:
            out = self.forward(None, training=False)          # eager: also recomputes scale|shift of every BatchNorm
            if self._igraph is None:
                self._igraph = _capture(lambda: self.forward(None, training=False))
            return out
        self._igraph.replay()
        return self.out

    def backward(self, on_node_done=None, after_head=None):
        """Back-propagates self.dout (filled by loss()) into self.grad.  on_node_done(node) is called
        after each node's launches are enqueued (used to start the gradient all-reduce early); after_head() once the
        Dense head -- the first node of backward -- has been back-propagated (and its buckets launched): _step_body
        starts the head's optimizer step there."""
        g = self.dout
        if self.sigmoid_cols is not None:      # through the sigmoid columns: dout *= y (1 - y)
            L.spnet_selective_sigmoid(L.ptr(self.out), L.ptr(self.dout), self.B, self.n_out, self.sigmoid_cols[0],
                                      self.sigmoid_cols[1], 1, _stream())
        self.deferred_wgrads = []
        if self._wT_ver[0] != self._tver[0]:    # weights were loaded / re-initialised since the last optimizer step
            self.refresh_transposes()
        self.refresh_planes()
        for node in reversed(self.nodes):
            g = node.bwd(g)
            if node is self._first_middle:      # (flushing in 2 or 4 smaller batches measured no faster)
                self.flush_deferred_wgrads()
            if on_node_done is not None:
                on_node_done(node)
            if after_head is not None and node is self.nodes[-1]:
                after_head()
        self.flush_deferred_wgrads()
        self.reduce_depthwise_wgrads()
        if self.wgrad_stream is not None:      # every weight gradient must have landed before the optimizer
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)

    def _build_planes(self):
        """bf16x3 operand planes for every pointwise kernel of this plan that a bf16x3 launch will read (allocated once per
        weight set: a later plan over the same weights finds them, and adds the data-gradient planes an inference plan
        had no use for)."""
        for pw in self._pw_layers:
            if not (pw.x3_fwd or pw.x3_dgrad):
                continue
            ent = self._planes.setdefault(pw.wname, [None, None])
            if pw.x3_fwd and ent[0] is None:
                ent[0] = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(pw.cout, pw.cin)), dtype=torch.int16, device=self.dev)
                self._planes_gen[0] += 1
            if pw.x3_dgrad and ent[1] is None:
                ent[1] = torch.zeros(3 * int(L.spnet_bf16x3_plane_elems(pw.cin, pw.cout)), dtype=torch.int16, device=self.dev)
                self._planes_gen[0] += 1
        if self._planes_jobs is None or self._planes_jobs[1] != self._planes_gen[0]:
            self._planes_ver[0] = -1

    def refresh_planes(self):
        """The weights split into their bf16 planes, both operand forms, ONE batched launch -- whenever theta changed since
        the last split (optimizer step, load, re-initialisation).  Called at the start of forward() / backward() on the
        main stream: the split is ordered in front of every consumer on every stream (a strided block's residual
        convolution runs on the side stream)."""
        if self.pointwise != "bf16x3" or not self._planes or self._planes_ver[0] == self._tver[0]:
            return
        if self._planes_jobs is None or self._planes_jobs[1] != self._planes_gen[0]:
            flat, mx = [], 1
            for wname, (pf, pd) in self._planes.items():
                off, n, shape = self.p_off[wname]
                cin, cout = int(shape[-2]), int(shape[-1])
                w = self.theta.data_ptr() + 4 * off
                if pf is not None:      # forward: B element (n = cout, k = cin) = W[k][n]
                    flat += [w, pf.data_ptr(), cin, cout, 1, cout]
                    mx = max(mx, pf.numel() // 3)
                if pd is not None:      # data gradient dX = dY W^T: (n = cin, k = cout) = W[n][k]
                    flat += [w, pd.data_ptr(), cout, cin, cout, 1]
                    mx = max(mx, pd.numel() // 3)
            if self._planes_jobs is not None:     # a captured graph may still hold the old table's address: keep it alive,
                self._planes_tables_kept = getattr(self, "_planes_tables_kept", []) + [self._planes_jobs[0]]
                self._graph = self._igraph = None     # ... and re-capture with the new one
            self._planes_jobs = (torch.tensor(flat, dtype=torch.int64, device=self.dev), self._planes_gen[0], len(flat) // 6, mx)
        table, _, nj, mx = self._planes_jobs
        L.spnet_split_bf16x3_batched(table.data_ptr(), nj, mx, _stream())
        self._planes_ver[0] = self._tver[0]

    def transposed(self, wname):
        """W^T buffer ([cout][cin]) of a pointwise kernel; kept current by refresh_transposes()."""
        if wname not in self._wT:
            off, n, shape = self.p_off[wname]
            self._wT[wname] = self.new(n)
            self._wT_ver[0] = -1
        return self._wT[wname]

    def refresh_transposes(self):
        """One batched launch: W^T of the pointwise kernels whose data-gradient GEMM blends the BatchNorm backward
        into its A operand (Pointwise.blend) -- that kernel exists in the forward operand form only.  (Round 2 also
        measured ALL data-gradient GEMMs on transposed copies: the forward form is no faster than the K-major-B
        form, 61.5 vs 69.2 us on 6144x728x728, so the other layers read W in place.)"""
        if not self._wT:
            self._wT_ver[0] = self._tver[0]
            return
        if self._wT_jobs is None or self._wT_jobs[1] != len(self._wT):
            flat, mr, mc = [], 1, 1
            for name, dst in self._wT.items():
                off, n, shape = self.p_off[name]
                R, C = int(shape[-2]), int(shape[-1])          # [cin][cout] (1x1 HWIO kernels: trailing two dims)
                flat += [self.theta.data_ptr() + 4 * off, dst.data_ptr(), R, C]
                mr, mc = max(mr, R), max(mc, C)
            self._wT_jobs = (torch.tensor(flat, dtype=torch.int64, device=self.dev), len(self._wT), mr, mc)
        table, nj, mr, mc = self._wT_jobs
        L.spnet_transpose_batched(table.data_ptr(), nj, mr, mc, _stream())
        self._wT_ver[0] = self._tver[0]

    def reduce_depthwise_wgrads(self):
        """Fold the partial sums every fused depthwise backward left behind into the 34 depthwise weight
        gradients: one launch (weight-gradient stream) instead of one small kernel per layer on the
        data-gradient chain."""
        jobs = self.dw_reduce_jobs
        if not jobs:
            return
        if self._dw_reduce_table is None:
            flat = [v for (part, grad, rows, Lr) in jobs for v in (part.data_ptr(), grad.data_ptr(), rows, Lr)]
            self._dw_reduce_table = torch.tensor(flat, dtype=torch.int64, device=self.dev)
        max_L = max(j[3] for j in jobs)
        side = self.wgrad_stream
        if side is None:
            L.spnet_reduce_rows_batched(self._dw_reduce_table.data_ptr(), len(jobs), max_L, _stream())
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                L.spnet_reduce_rows_batched(self._dw_reduce_table.data_ptr(), len(jobs), max_L, _stream())

    def flush_deferred_wgrads(self):
        """dW of every deferred pointwise layer in one batched launch (weight-gradient stream).  The operand
        table lives in device memory and is rebuilt only if a buffer address changed (never, in practice:
        the plan's buffers are static)."""
        todo = self.deferred_wgrads
        if not todo:
            return
        self.deferred_wgrads = []
        x3 = [t for t in todo if t[0] == "x3"]
        if x3:
            self._flush_x3_wgrads(x3)
            todo = [t for t in todo if t[0] != "x3"]
            if not todo:
                return
        x0, dy0, gw0, cin, cout, M = todo[0]
        if any((t[3], t[4], t[5]) != (cin, cout, M) for t in todo):
            raise RuntimeError("deferred weight gradients must share one shape")
        # element offsets of every problem's operands from problem 0's (kernel-argument base pointers keep the
        # operand fetches in the global address space)
        a0, b0, c0 = x0.data_ptr(), dy0.data_ptr(), gw0.data_ptr()
        offs = [v for (x, dy, gw, _, _, _) in todo
                for v in ((x.data_ptr() - a0) // 4, (dy.data_ptr() - b0) // 4, (gw.data_ptr() - c0) // 4)]
        if self._wgrad_table is None:
            self._wgrad_table = {}
        key = (a0, b0, c0) + tuple(offs)
        if key not in self._wgrad_table:
            self._wgrad_table[key] = torch.tensor(offs, dtype=torch.int64, device=self.dev)
        table = self._wgrad_table[key]
        nb = len(todo)

        def launch():
            prof = self.prof
            t0 = prof.start() if prof is not None else None
            L.spnet_gemm_f32_batched(a0, b0, c0, table.data_ptr(), nb, OUT_MAJOR, cin, OUT_MAJOR, cout, cout, cin, cout, M,
                                     5, _stream())
            if prof is not None:
                prof.stop("gemm", t0, 2.0 * nb * cin * cout * M, ("AB x%d batched" % nb, cin, cout, M))

        side = self.wgrad_stream
        if side is None:
            launch()
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch()

    def _flush_x3_wgrads(self, todo):
        """dW of the deferred layers whose operands are bf16x3 planes: ONE launch of spnet_gemm_bf16x3_wgrad_batched
        (weight-gradient stream); a K split (slabs in the weight-gradient stream's workspace + one ordered sum per layer)
        only if the batch alone does not fill the chip."""
        _, _, _, _, cin, cout, M = todo[0]
        if any((t[4], t[5], t[6]) != (cin, cout, M) for t in todo):
            raise RuntimeError("deferred weight gradients must share one shape")
        nb = len(todo)
        side = self.wgrad_stream
        region = WS_GEMM2 if side is not None else WS_GEMM
        ks = int(L.spnet_gemm_bf16x3_wgrad_ksplit(cin, cout, M, nb))
        if ks > 1 and nb * ks * cin * cout > region[1]:
            ks = max(1, region[1] // (nb * cin * cout))
        key = ("x3", ks) + tuple(v for t in todo for v in (t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr()))
        if self._wgrad_table is None:
            self._wgrad_table = {}
        if key not in self._wgrad_table:
            ws = self.ws_ptr(region)
            flat = [v for b, t in enumerate(todo)
                    for v in (t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr() if ks == 1 else ws + 4 * b * ks * cin * cout)]
            self._wgrad_table[key] = torch.tensor(flat, dtype=torch.int64, device=self.dev)
        table = self._wgrad_table[key]

        def launch():
            prof = self.prof
            t0 = prof.start() if prof is not None else None
            L.spnet_gemm_bf16x3_wgrad_batched(table.data_ptr(), nb, cin, cout, M, ks, _stream())
            if ks > 1:
                ws = self.ws_ptr(region)
                for b, t in enumerate(todo):
                    L.spnet_reduce_slabs(ws + 4 * b * ks * cin * cout, ks, cin, cout, t[3].data_ptr(), cout, _stream())
            if prof is not None:
                prof.stop("gemm", t0, 2.0 * nb * cin * cout * M, ("x3w AB x%d batched" % nb, cin, cout, M))

        if side is None:
            launch()
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                launch()

    def loss(self, Y=None, with_grad=True):
        if Y is not None:
            self.y_true.copy_(Y.reshape(self.y_true.shape))
        L.spnet_ellipse_loss(L.ptr(self.y_true), L.ptr(self.out), L.ptr(self.dout) if with_grad else None,
                             L.ptr(self.loss_parts), L.ptr(self.loss_out), self.B, self.n_out,
                             0 if self.loss_type == "same" else 1, _stream())
        return self.loss_out

    def _upload_step_params(self, lr):
        """Advance the optimizer / dropout counters on the host and publish this step's scalars."""
        self.t += 1
        self.drop_seed = (self.drop_seed * 1664525 + 1013904223) & 0xFFFFFFFF
        b1, b2 = 0.9, 0.999
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        self._publish_step_params(lr_t)

    def _publish_step_params(self, lr_t=None):
        """[lr_t, dropout seed] -> device.  The host runs several steps ahead of the GPU, so the two words travel
        through a ring of pinned staging slots (one slot per step in flight, reused only after its own DMA has
        completed) and are copied into step_params IN STREAM ORDER: a step's kernels can never see the values of
        a later step, which a single pinned buffer overwritten every step would allow."""
        if lr_t is not None:
            self._lr_bits = int(np.float32(lr_t).view(np.int32))
        h = np.array([getattr(self, "_lr_bits", 0), int(np.uint32(self.drop_seed).view(np.int32)), 0, 0], np.int32)
        self.step_params.copy_(self._step_upload("step", h), non_blocking=True)

    def set_drop_seed(self, seed):
        """Seed used by the next forward(training=True) called outside train_step (tests, smoke)."""
        self.drop_seed = int(seed) & 0xFFFFFFFF
        self._publish_step_params()

    def _adam_range(self, which, grad_scale):
        """The fused Keras-Adam + l2 kernel over one of the optimizer's two ranges of the flat buffers: 0 = the Dense
        head's kernel (offset 0, 73 % of the parameters at the benchmark geometry), 1 = everything else."""
        b1, b2 = 0.9, 0.999
        lo, hi = (0, self._head_hi) if which == 0 else (self._head_hi, self.n_theta)
        if hi <= lo:
            return
        mask = None if self.update_mask is None else self.update_mask.data_ptr() + 4 * lo
        L.spnet_adam_part(self.theta.data_ptr() + 4 * lo, self.grad.data_ptr() + 4 * lo, self.m.data_ptr() + 4 * lo,
                          self.v.data_ptr() + 4 * lo, hi - lo, max(0, min(self.l2_n, hi) - lo), 0.0, b1, b2, self.adam_eps,
                          L2_COEF, grad_scale, mask, self.sq_scratch.data_ptr() + 4 * (0 if which == 0 else self._head_parts),
                          self.lr_ptr, _stream())

    def adam_head_early(self, grad_scale=1.0):
        """The head range's optimizer step, launched from backward as soon as the head's gradient is final (the Dense
        node is the first of backward; single-process training only): on a stream of its own, underneath the backbone's
        backward.  Waits for the data-gradient GEMM that reads the head's weights (main stream) and for the
        weight-gradient GEMM (weight-gradient stream).  One-stream plans (SPNET_OVERLAP_WGRAD=0, bench.py's roofline
        leg) run it in line.  Measured (tools/ab_engine_flags.py early_head=True,False, one box): 10.000 against
        10.048 ms per step -- the 0.34 ms of launches that leave the dependency chain come back as slower neighbours,
        the step is bound by what its kernels move, not by their order."""
        self._head_done = True
        side = self.wgrad_stream
        if side is None:
            self._adam_range(0, grad_scale)
            return
        if self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=self.dev)
        opt = self._opt_stream
        opt.wait_stream(torch.cuda.current_stream())
        opt.wait_stream(side)
        with torch.cuda.stream(opt):
            self._adam_range(0, grad_scale)

    def adam_step(self, lr=None, grad_scale=1.0):
        """Fused Keras-Adam + l2 over the flat buffers, in two ranges (the Dense head's kernel; everything else) so that
        the first can run early (adam_head_early) -- the same two launches, the same bits, wherever they run.
        lr=None: the step size already sits in device memory (train_step); a float: stand-alone use."""
        if lr is not None:
            self._upload_step_params(lr)
        self._wver[0] += 1
        if self._head_done:
            if self._opt_stream is not None:
                torch.cuda.current_stream().wait_stream(self._opt_stream)
        else:
            self._adam_range(0, grad_scale)
        self._head_done = False
        self._adam_range(1, grad_scale)
        L.spnet_adam_l2_sum(L.ptr(self.sq_scratch), self._head_parts + self._rest_parts, L2_COEF, self.loss_out[6:].data_ptr(),
                            _stream())
        self._tver[0] += 1
        self.refresh_transposes()
        # ... and the bf16x3 planes of the new weights, here and not lazily at the next forward: a captured train step
        # (SPNET_TRAIN_GRAPH=1) then always contains the split, whatever ran between the warm step and the capture
        # (round-4 ADVICE: a predict / validation pass in between left the planes "fresh" and the graph without a split)
        self.refresh_planes()

    def train_step(self, X, Y, lr, reducer=None):
        """augmented batch X -> forward -> custom_loss -> backward -> (all-reduce) -> Adam(+l2).
        X / Y may be None when self.x_in / self.y_true were filled in place.  `reducer` is a
        parallel.GradReducer (data parallel) or None.  Returns the device tensor loss_out (no host
        sync): [center,size,angle,noobj,class,data_total,l2_penalty,_]."""
        if X is not None:
            self.x_in.copy_(X.reshape(self.x_in.shape))
        if Y is not None:
            self.y_true.copy_(Y.reshape(self.y_true.shape))
        self._upload_step_params(lr)
        if reducer is None and self.use_graph and self.update_mask is None and self.prof is None:
            # Single GPU: the step is a fixed sequence of ~500 launches on two streams -> captured ONCE as
            # a hipGraph and replayed (host enqueue 1.9 ms instead of 4.4 ms per step).
            if self._graph is None and self._graph_warm >= 1:
                self._graph = _capture(lambda: self._step_body(None, 1.0))
            if self._graph is not None:
                self._graph.replay()
                # the host bookkeeping _step_body does outside the launches: weights, moving statistics and the
                # transposed copies changed (the replay refreshed the latter itself), and this plan's BatchNorm
                # scale|shift hold batch statistics again -- predict_step must rebuild its coefficients
                self._wver[0] += 2
                self._coeff_ver = -1
                self._tver[0] += 1
                self._wT_ver[0] = self._tver[0]
                self._planes_ver[0] = self._tver[0]
                return self.loss_out
            self._graph_warm += 1
        self._step_body(reducer, None)
        return self.loss_out

    def _step_body(self, reducer, scale):
        self.forward(None, training=True)
        self.loss(None)
        if reducer is None:
            scale = 1.0 if scale is None else scale
            self.backward(after_head=(lambda: self.adam_head_early(scale)) if self.early_head else None)
        else:
            reducer.side_stream = self.wgrad_stream
            # (the head's optimizer step stays behind finish() here: launching it behind the head buckets' all-reduce is the
            # same two launches, but nothing on the build's one-GPU boxes can time it against RCCL's channel kernels)
            self.backward(on_node_done=reducer.on_node_done)
            # (measurement hook: reducer.exposed = [] makes every step leave an event pair around finish() -- what the
            # main stream waits for the collectives AFTER backward has ended, i.e. the all-reduce time not hidden)
            ev = getattr(reducer, "exposed", None)
            if ev is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            scale = reducer.finish()
            if ev is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                ev.append((e0, e1))
        self.adam_step(None, scale)

    def grad_buckets(self, bucket_bytes=32 << 20):
        """plan_grad_buckets over this plan's nodes: (buckets [(lo, hi, trigger node)], tail [(lo, hi)])."""
        nodes = [(node, tuple(getattr(node, "pnames", ())), isinstance(node, MiddleBlock)) for node in self.nodes]
        return plan_grad_buckets(self.p_off, self.rest_lo, self.n_theta, nodes, bucket_bytes)

    def make_reducer(self, group=None, force=False, bucket_bytes=32 << 20):
        """parallel.GradReducer over this plan's gradient buckets (force=True: run the collectives even with one
        rank, which exercises RCCL's stream hand-off on a single GPU)."""
        from . import parallel
        buckets, tail = self.grad_buckets(bucket_bytes)
        return parallel.GradReducer(self.grad, buckets, tail=tail, group=group, force=force)

    def head_grad_range(self):
        """[lo,hi) of FinalOutput/kernel inside the flat gradient: produced first in backward and 73 %
        of all gradient bytes, so its all-reduce overlaps the whole backbone backward."""
        off, n, _ = self.p_off["FinalOutput/kernel"]
        return off, off + n


# ======================================================================================= nodes
class Node:
    def fwd(self, training):
        raise NotImplementedError

    def bwd(self, g):
        raise NotImplementedError


def _load_tile_table():
    """Per-shape tile choices measured IN the train step (tools/autotune_gemm.py -> spnet_amd/gemm_tiles.json): the cost
    model inside spnet_gemm_f32 is fitted to isolated launches, and a few of the network's shapes run faster on another
    tile between their real neighbours.  Keys "form,M,N,K" (form: a_major b_major stats), values tile ids; shapes that
    are not listed (other batch sizes / geometries / backbones) keep the library's own choice."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tiles.json")
    if os.environ.get("SPNET_GEMM_TILES", "1") == "0" or not os.path.exists(path):
        return {}
    import json
    with open(path) as f:
        return {tuple(int(v) for v in k.split(",")): int(t) for k, t in json.load(f).get("tiles", {}).items()}


TILE_TABLE = _load_tile_table()
TILE_PROBE = {}          # tools/autotune_gemm.py: {(a_major, b_major, stats, M, N, K): tile} overrides for one measurement


def _tile_for(a_major, b_major, stats, M, N, K, tile):
    if tile:
        return tile
    key = (a_major, b_major, stats, M, N, K)
    return TILE_PROBE.get(key, TILE_TABLE.get(key, 0))


def _gemm(A, a_major, lda, Bm, b_major, ldb, C, ldc, M, N, K, eng, bias=None, split_k=0, tile=0, region=WS_GEMM):
    tile = _tile_for(a_major, b_major, 0, M, N, K, tile)
    prof = eng.prof
    if prof is not None:
        t0 = prof.start()
    L.spnet_gemm_f32(L.ptr(A), a_major, lda, L.ptr(Bm), b_major, ldb, L.ptr(C), ldc, M, N, K, split_k,
                     eng.ws_ptr(region), region[1], L.ptr(bias), tile, _stream())
    if prof is not None:
        prof.stop("gemm", t0, 2.0 * M * N * K, ("aA"[a_major] + "bB"[b_major], M, N, K))
        prof.keys[len(prof.records) - 1] = (a_major, b_major, 0, M, N, K)


_stat_rows = __import__("ctypes").c_int(0)


def _gemm_colstats(A, lda, Bm, ldb, C, ldc, M, N, K, eng, region=WS_BNP):
    """Forward-form GEMM whose epilogue also leaves the BatchNorm column sums of C in the WS_BNP region;
    returns the number of partial rows."""
    if (M + 31) // 32 * 2 * N > region[1]:
        raise RuntimeError("BatchNorm partial region too small for M=%d N=%d" % (M, N))
    prof = eng.prof
    if prof is not None:
        t0 = prof.start()
    L.spnet_gemm_f32_colstats(L.ptr(A), K_MAJOR, lda, L.ptr(Bm), OUT_MAJOR, ldb, L.ptr(C), ldc, M, N, K,
                              _tile_for(K_MAJOR, OUT_MAJOR, 1, M, N, K, 0),
                              eng.ws_ptr(region), __import__("ctypes").addressof(_stat_rows), _stream())
    if prof is not None:
        prof.stop("gemm", t0, 2.0 * M * N * K, ("aB+stats", M, N, K))
        prof.keys[len(prof.records) - 1] = (K_MAJOR, OUT_MAJOR, 1, M, N, K)
    return _stat_rows.value


class SmallConv(Node):
    """Direct 3x3 conv with <=3 input channels (stem convs, block1_conv1)."""

    def __init__(self, eng, x, cin, cout, stride, same, name, need_dx=True):
        self.e, self.x, self.cin, self.cout, self.stride, self.same = eng, x, cin, cout, stride, int(same)
        B, H, W, _ = x.shape
        self.Hin, self.Win = H, W
        self.OH = H if same else (H - 3) // stride + 1
        self.OW = W if same else (W - 3) // stride + 1
        self.y = eng.new(B, self.OH, self.OW, cout)
        self.w = eng.P(name + "/kernel")
        self.pnames = [name]
        self.need_dx = need_dx
        if eng.train_capable:
            self.gw = eng.G(name + "/kernel")
            self.dx = eng.new(*x.shape) if need_dx else None

    def _call(self, op, a, b, out, region=WS_MISC):
        e = self.e
        L.spnet_conv3x3_small(op, self.cin, self.cout, self.stride, self.same, L.ptr(a), L.ptr(b), L.ptr(out),
                              e.B, self.Hin, self.Win, e.ws_ptr(region), region[1], _stream())

    def fwd(self, training):
        self._call(0, self.x, self.w, self.y)

    def bwd(self, g):
        side = self.e.wgrad_stream
        if side is None:
            self._call(2, self.x, g, self.gw)
        else:           # weight gradient off the data-gradient chain (see Pointwise.bwd)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._call(2, self.x, g, self.gw, region=WS_GEMM2)
        if self.need_dx:
            self._call(1, g, self.w, self.dx)
            return self.dx
        return None


class StemHead(Node):
    """conv2d_1 (1 -> 3, 3x3 same) + AveragePooling2D(2) and the skip connection's AveragePooling2D(2) of the input
    (spnet/models.py:321-323, 337) in one kernel each way: the full-resolution 3-channel tensor is never materialised
    (75 MB per step at batch 32) and its pooling backward is folded into the weight-gradient sum."""

    def __init__(self, eng, x, name):
        self.e, self.x = eng, x
        self.pnames = [name]
        B, H, W, _ = x.shape
        self.H, self.W = H, W
        self.w = eng.P(name + "/kernel")
        self.y = eng.new(B, H // 2, W // 2, 3)
        self.px = eng.new(B, H // 2, W // 2, 1)
        self.gw = eng.G(name + "/kernel") if eng.train_capable else None

    def fwd(self, training):
        L.spnet_stem_head(0, L.ptr(self.x), L.ptr(self.w), L.ptr(self.y), L.ptr(self.px), self.e.B, self.H, self.W,
                          None, 0, _stream())

    def bwd(self, g):
        e = self.e
        side = e.wgrad_stream
        region = WS_MISC if side is None else WS_GEMM2

        def call():
            L.spnet_stem_head(2, L.ptr(self.x), L.ptr(g), L.ptr(self.gw), None, e.B, self.H, self.W, e.ws_ptr(region),
                              region[1], _stream())
        if side is None:
            call()
        else:           # weight gradient off the data-gradient chain (see Pointwise.bwd)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                call()
        return None


class AvgPool(Node):
    def __init__(self, eng, x, C, need_dx=True):
        self.e, self.x, self.C = eng, x, C
        B, H, W, _ = x.shape
        self.H, self.W = H, W
        self.y = eng.new(B, H // 2, W // 2, C)
        self.need_dx = need_dx
        self.dx = eng.new(*x.shape) if (eng.train_capable and need_dx) else None

    def fwd(self, training):
        L.spnet_avgpool2_fwd(L.ptr(self.x), L.ptr(self.y), self.e.B, self.H, self.W, self.C, _stream())

    def bwd(self, g):
        if not self.need_dx:
            return g      # side branch (the avg-pooled input): nothing upstream needs a gradient
        L.spnet_avgpool2_bwd(L.ptr(g), L.ptr(self.dx), self.e.B, self.H, self.W, self.C, _stream())
        return self.dx


class BatchNorm(Node):
    """y = act(BN(x)) (+ residual).  Backward runs in place on the incoming gradient buffer."""

    def __init__(self, eng, x, C, name, act, residual=None, res_bcast=False, bwd_inplace=True):
        self.e, self.x, self.C, self.act = eng, x, C, act
        self.M = x.numel() // C
        self.pnames = [name]
        self.gamma, self.beta = eng.P(name + "/gamma"), eng.P(name + "/beta")
        self.mm, self.mv = eng.S(name + "/moving_mean"), eng.S(name + "/moving_variance")
        self.residual, self.res_bcast = residual, res_bcast
        self.y = eng.new(*x.shape)
        if eng.train_capable:
            self.ggamma, self.gbeta = eng.G(name + "/gamma"), eng.G(name + "/beta")
            self.save = eng.new(2 * C)
            self.dx = None if bwd_inplace else eng.new(*x.shape)

    def fwd(self, training):
        e, C = self.e, self.C
        ss = e.small[:2 * C]
        if training:
            L.spnet_bn_fwd_train(L.ptr(self.x), self.M, C, L.ptr(self.gamma), L.ptr(self.beta), L.ptr(self.mm),
                                 L.ptr(self.mv), L.ptr(self.save), self.save[C:].data_ptr(), L.ptr(ss), self.act,
                                 L.ptr(self.residual), int(self.res_bcast), L.ptr(self.y), BN_EPS, BN_MOMENTUM,
                                 e.ws_ptr(WS_MISC), _stream())
        else:
            L.spnet_bn_fwd_infer(L.ptr(self.x), self.M, C, L.ptr(self.gamma), L.ptr(self.beta), L.ptr(self.mm),
                                 L.ptr(self.mv), L.ptr(ss), self.act, L.ptr(self.residual), int(self.res_bcast),
                                 L.ptr(self.y), BN_EPS, _stream())

    def bwd(self, g):
        e, C = self.e, self.C
        co = e.small[:3 * C]
        out = g if self.dx is None else self.dx
        L.spnet_bn_bwd(L.ptr(self.x), L.ptr(g), self.M, C, L.ptr(self.gamma), L.ptr(self.beta), L.ptr(self.save),
                       self.save[C:].data_ptr(), self.act, L.ptr(out), L.ptr(self.ggamma), L.ptr(self.gbeta),
                       L.ptr(co), e.ws_ptr(WS_MISC), _stream())
        return out


class Dropout(Node):
    def __init__(self, eng, x, rate):
        self.e, self.x, self.rate = eng, x, rate
        # an inference-only plan has no use for a second buffer: Dropout is the identity there (a 75 MB copy per forward
        # of the batch-128 predict plan otherwise)
        self.y = eng.new(*x.shape) if eng.train_capable else x
        self.seed = 0

    def fwd(self, training):
        if training:      # the seed of this step lives in device memory (engine.step_params[1]) -> graph-replayable
            L.spnet_dropout(L.ptr(self.x), L.ptr(self.y), self.x.numel(), 0, self.rate, self.e.seed_ptr, _stream())
        elif self.y is not self.x:
            self.y.copy_(self.x)

    def bwd(self, g):
        L.spnet_dropout(L.ptr(g), L.ptr(g), g.numel(), 0, self.rate, self.e.seed_ptr, _stream())
        return g


class Pointwise:
    """1x1 conv as GEMM over flattened pixels: y[M,cout] = x[M,cin] @ W[cin,cout]."""

    def __init__(self, eng, M, cin, cout, wname, defer_wgrad=False, allow_blend=True):
        self.e, self.M, self.cin, self.cout = eng, M, cin, cout
        self.w = eng.P(wname)
        self.wname = wname
        self.gw = eng.G(wname) if eng.train_capable else None
        # bf16x3 kernel (csrc/gemm_bf16x3.hip) for the forward / data-gradient GEMM where its 96-wide tile pays: from 256
        # output columns on (in-step, us: 6144x728x728 67.7 -> 49.6, 94752x256x256 124 -> 103, 1536x1536x1024 52 -> 42;
        # 372000x128x128 147 -> 169: stays on the exact kernel)
        # ... and while its 96 x 96 tiles still fill the chip: at least 192 of them (the reference's own 331 x 331 layout at
        # batch 16 has 1,600-row middle-flow GEMMs, 136 tiles: 3,120 -> 2,950 training images/s with the bf16x3 kernel
        # there, so those stay on the exact kernel's autotuned 32-row tiles; 1536x1024x728 (176 tiles): 31.6 -> 32.3 us)
        x3 = eng.pointwise == "bf16x3" and cin % 4 == 0 and cout % 4 == 0
        tiles = lambda n: ((M + 95) // 96) * ((n + 95) // 96)
        self.x3_fwd = x3 and cout >= 256 and tiles(cout) >= eng.x3_min_tiles
        self.x3_dgrad = x3 and cin >= 256 and tiles(cin) >= eng.x3_min_tiles and eng.train_capable
        eng._pw_layers.append(self)
        # BatchNorm backward blended into the data-gradient GEMM's A operand (bwd_blend) instead of an elementwise
        # pass of its own.  The blend reads BOTH g and yp once per COLUMN tile of dX, so it only pays while dX has
        # one or two column tiles; measured on MI355X (tools/gemm_sweep.py blend, us incl. the BN kernels):
        #   cin 128 (M 372,000): 242 -> 202     cin 256 (M 94,752): 163 -> 189     cin 728 (M 6,144): 85 -> 99
        self.blend = bool(allow_blend and eng.train_capable and cin <= 128)
        self.wT = eng.transposed(wname) if self.blend else None     # [cout][cin]: forward operand form for the blend
        self.defer_wgrad = defer_wgrad      # dW is left to Engine.flush_deferred_wgrads() (one batched launch)

    def _x3(self, tag, A, lda, planes, C, ldc, N, K, colstats_region=None):
        """One bf16x3 launch: C[M][N] = A[M][K] x planes (+ BatchNorm column sums of C); returns the partial row count."""
        e = self.e
        prof = e.prof
        t0 = prof.start() if prof is not None else None
        rows = 0
        if colstats_region is None:
            L.spnet_gemm_bf16x3_fwd(L.ptr(A), lda, L.ptr(planes), L.ptr(C), ldc, self.M, N, K, _stream())
        else:
            if (self.M + 95) // 96 * 2 * N > colstats_region[1]:
                raise RuntimeError("BatchNorm partial region too small for M=%d N=%d" % (self.M, N))
            L.spnet_gemm_bf16x3_fwd_colstats(L.ptr(A), lda, L.ptr(planes), L.ptr(C), ldc, self.M, N, K,
                                             e.ws_ptr(colstats_region), __import__("ctypes").addressof(_stat_rows), _stream())
            rows = _stat_rows.value
        if prof is not None:
            prof.stop("gemm", t0, 2.0 * self.M * N * K, (tag, self.M, N, K))
        return rows

    def _x3p(self, tag, a_planes, b_planes, C, N, K, colstats_region=None):
        """One planes x planes launch: C[M][N] = A planes x B planes^T (+ BatchNorm column sums); returns the partial row
        count."""
        e = self.e
        prof = e.prof
        t0 = prof.start() if prof is not None else None
        rows = 0
        if colstats_region is None:
            L.spnet_gemm_bf16x3_pp(L.ptr(a_planes), L.ptr(b_planes), L.ptr(C), N, self.M, N, K, None, None, _stream())
        else:
            if (self.M + 95) // 96 * 2 * N > colstats_region[1]:
                raise RuntimeError("BatchNorm partial region too small for M=%d N=%d" % (self.M, N))
            L.spnet_gemm_bf16x3_pp(L.ptr(a_planes), L.ptr(b_planes), L.ptr(C), N, self.M, N, K, e.ws_ptr(colstats_region),
                                   __import__("ctypes").addressof(_stat_rows), _stream())
            rows = _stat_rows.value
        if prof is not None:
            prof.stop("gemm", t0, 2.0 * self.M * N * K, (tag, self.M, N, K))
        return rows

    def fwd_p(self, zp, y, colstats_region=None):
        """Forward from the planes of x (written by the producing depthwise kernel); with colstats_region the BatchNorm
        column sums of y are left there and the partial row count is returned."""
        tag = "x3p aB" if colstats_region is None else "x3p aB+stats"
        return self._x3p(tag, zp, self.e._planes[self.wname][0], y, self.cout, self.cin, colstats_region)

    def bwd_p(self, zp, dyp, dx):
        """Backward from planes: dW = z^T dy by spnet_gemm_bf16x3_wgrad_batched (deferred into the engine's batched launch,
        or alone on the weight-gradient stream with a deterministic K split), dx = dy W^T by the planes x planes kernel."""
        e = self.e
        if self.defer_wgrad:
            e.deferred_wgrads.append(("x3", zp, dyp, self.gw, self.cin, self.cout, self.M))
        else:
            e._flush_x3_wgrads([("x3", zp, dyp, self.gw, self.cin, self.cout, self.M)])
        if dx is not None:
            self._x3p("x3p ab", dyp, e._planes[self.wname][1], dx, self.cin, self.cout)

    def fwd(self, x, y):
        if self.x3_fwd and self.e.pointwise == "bf16x3":
            self._x3("x3 aB", x, self.cin, self.e._planes[self.wname][0], y, self.cout, self.cout, self.cin)
            return
        _gemm(x, K_MAJOR, self.cin, self.w, OUT_MAJOR, self.cout, y, self.cout, self.M, self.cout, self.cin, self.e)

    def fwd_colstats(self, x, y, region=WS_BNP):
        """Forward + BatchNorm column sums of y left in `region`; returns the partial row count."""
        if self.x3_fwd and self.e.pointwise == "bf16x3":
            return self._x3("x3 aB+stats", x, self.cin, self.e._planes[self.wname][0], y, self.cout, self.cout, self.cin,
                            colstats_region=region)
        return _gemm_colstats(x, self.cin, self.w, self.cout, y, self.cout, self.M, self.cout, self.cin, self.e,
                              region=region)

    def bwd(self, x, dy, dx):
        """dW[cin,cout] = x^T dy ; dx[M,cin] = dy W^T.  The two products are independent: dW goes to the
        engine's weight-gradient stream (own split-K workspace) and overlaps everything that follows on
        the main stream until Engine.backward() joins the streams in front of the optimizer.  x and dy
        are not rewritten before that join (they are only produced once per step)."""
        e = self.e
        side = e.wgrad_stream
        if self.defer_wgrad:
            e.deferred_wgrads.append((x, dy, self.gw, self.cin, self.cout, self.M))
        elif side is None:
            _gemm(x, OUT_MAJOR, self.cin, dy, OUT_MAJOR, self.cout, self.gw, self.cout, self.cin, self.cout, self.M, e)
        else:
            side.wait_stream(torch.cuda.current_stream())      # dy has just been produced on the main stream
            with torch.cuda.stream(side):
                _gemm(x, OUT_MAJOR, self.cin, dy, OUT_MAJOR, self.cout, self.gw, self.cout, self.cin, self.cout,
                      self.M, e, region=WS_GEMM2)
        if dx is not None and self.x3_dgrad and e.pointwise == "bf16x3":
            self._x3("x3 ab", dy, self.cout, e._planes[self.wname][1], dx, self.cin, self.cin, self.cout)
        elif dx is not None:    # dx[M,cin] = dy[M,cout] @ W^T: W read in place as a K-major B operand
            _gemm(dy, K_MAJOR, self.cout, self.w, K_MAJOR, self.cout, dx, self.cin, self.M, self.cin, self.cout, e)


    def bwd_blend(self, x, g, yp, bn, dyb, dx):
        """Backward through BatchNorm + this 1x1 conv in two GEMMs and no elementwise pass: the data-gradient GEMM
        builds dy = BN'(g, yp) while staging its A operand (coefficients in bn.coef) and writes it to dyb once;
        the weight-gradient GEMM (side stream / deferred batched launch) then reads dyb."""
        e = self.e
        prof = e.prof
        t0 = prof.start() if prof is not None else None
        L.spnet_gemm_f32_bnblend(L.ptr(g), L.ptr(yp), L.ptr(bn.coef), bn.cld, self.cout, L.ptr(self.wT), self.cin,
                                 L.ptr(dx), self.cin, self.M, self.cin, self.cout, 0, L.ptr(dyb), _stream())
        if prof is not None:
            prof.stop("gemm", t0, 2.0 * self.M * self.cin * self.cout, ("aB blend", self.M, self.cin, self.cout))
        side = e.wgrad_stream
        if self.defer_wgrad:
            e.deferred_wgrads.append((x, dyb, self.gw, self.cin, self.cout, self.M))
        elif side is None:
            _gemm(x, OUT_MAJOR, self.cin, dyb, OUT_MAJOR, self.cout, self.gw, self.cout, self.cin, self.cout, self.M, e)
        else:
            side.wait_stream(torch.cuda.current_stream())      # dyb has just been produced on the main stream
            with torch.cuda.stream(side):
                _gemm(x, OUT_MAJOR, self.cin, dyb, OUT_MAJOR, self.cout, self.gw, self.cout, self.cin, self.cout,
                      self.M, e, region=WS_GEMM2)


class Conv3x3Gemm(Node):
    """block1_conv2 (3x3 VALID, 32 -> 64): three implicit GEMMs that gather their operand tiles straight
    from the NHWC tensors -- the 9x patch matrix (428 MB at batch 32, 512x384) is never built."""

    def __init__(self, eng, x, cin, cout, name):
        self.e, self.x, self.cin, self.cout = eng, x, cin, cout
        B, H, W, _ = x.shape
        self.H, self.W = H, W
        self.OH, self.OW = H - 2, W - 2
        self.y = eng.new(B, self.OH, self.OW, cout)
        self.w = eng.P(name + "/kernel")
        self.pnames = [name]
        if eng.train_capable:
            self.gw = eng.G(name + "/kernel")
            self.dx = eng.new(*x.shape)
            if L.spnet_conv3x3_wgrad_ws(B, H, W, cin, cout) > min(WS_GEMM[1], WS_GEMM2[1]):
                raise RuntimeError("split-K workspace too small for %s" % name)

    def _timed(self, flops, call):
        """Count the launch in the GEMM family of the kernel timers (same MFMA tile machinery)."""
        prof = self.e.prof
        if prof is None:
            return call()
        t0 = prof.start()
        call()
        prof.stop("gemm", t0, flops, ("conv3x3 implicit", int(flops / (2.0 * 9 * self.cin * self.cout)), self.cout, 9 * self.cin))

    def fwd(self, training):
        e = self.e
        self._timed(2.0 * e.B * self.OH * self.OW * 9 * self.cin * self.cout,
                    lambda: L.spnet_conv3x3_fwd(L.ptr(self.x), L.ptr(self.w), L.ptr(self.y), e.B, self.H, self.W,
                                                self.cin, self.cout, _stream()))

    def _wgrad(self, g, region):
        e = self.e
        self._timed(2.0 * e.B * self.OH * self.OW * 9 * self.cin * self.cout,
                    lambda: L.spnet_conv3x3_wgrad(L.ptr(self.x), L.ptr(g), L.ptr(self.gw), e.B, self.H, self.W,
                                                  self.cin, self.cout, e.ws_ptr(region), region[1], _stream()))

    def bwd(self, g):
        """dW on the weight-gradient stream (independent of dX, like every other layer's), then dX."""
        e = self.e
        side = e.wgrad_stream
        if side is None:
            self._wgrad(g, WS_GEMM)
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._wgrad(g, WS_GEMM2)
        self._timed(2.0 * e.B * self.H * self.W * 9 * self.cout * self.cin,
                    lambda: L.spnet_conv3x3_dgrad(L.ptr(g), L.ptr(self.w), L.ptr(self.dx), e.B, self.H, self.W,
                                                  self.cin, self.cout, _stream()))
        return self.dx


class BN:
    """Parameters, statistics and per-layer affine (scale|shift) of one BatchNormalization layer whose
    reductions run inside OTHER kernels' epilogues and whose affine is applied by its consumers on load
    (the normalised tensor itself is only written when `apply()` is called)."""

    def __init__(self, eng, C, M, name):
        self.e, self.C, self.M = eng, C, M
        self.gamma, self.beta = eng.P(name + "/gamma"), eng.P(name + "/beta")
        self.mm, self.mv = eng.S(name + "/moving_mean"), eng.S(name + "/moving_variance")
        self.ss = eng.new(2 * C)                 # [scale | shift], valid after finalize()/infer()
        if eng.train_capable:
            self.ggamma, self.gbeta = eng.G(name + "/gamma"), eng.G(name + "/beta")
            self.save = eng.new(2 * C)           # [batch mean | invstd]
            # backward as an operand blend: dx = k1*g + k2*x + k3 is built by the consuming GEMM while it stages its
            # A tile (spnet_gemm_f32_bnblend); [k1 | k2 | k3], cld floats apart, zero beyond C (whole K tiles)
            self.cld = (C + 63) // 64 * 64
            self.coef = torch.zeros(3 * self.cld, device=eng.dev, dtype=torch.float32)

    def coeffs_from_partials(self, rows):
        """dgamma, dbeta and the blend coefficients from the (sum g, sum g*xhat) partial rows a consumer's fused
        depthwise backward left in WS_BNP."""
        e = self.e
        L.spnet_bn_bwd_coeffs_from_partials(rows, e.ws_ptr(WS_BNP), self.M, self.C, L.ptr(self.gamma), self.mean_ptr,
                                            self.invstd_ptr, L.ptr(self.ggamma), L.ptr(self.gbeta), L.ptr(self.coef),
                                            self.cld, _stream())

    def coeffs_full(self, x, g):
        """The same from an own reduction pass over (x, g) (no activation behind this BatchNorm)."""
        e = self.e
        L.spnet_bn_bwd_coeffs(L.ptr(x), L.ptr(g), self.M, self.C, L.ptr(self.gamma), L.ptr(self.beta), self.mean_ptr,
                              self.invstd_ptr, L.ptr(self.ggamma), L.ptr(self.gbeta), L.ptr(self.coef), self.cld,
                              e.ws_ptr(WS_MISC), _stream())

    @property
    def scale_ptr(self):
        return self.ss.data_ptr()

    @property
    def shift_ptr(self):
        return self.ss.data_ptr() + 4 * self.C

    @property
    def mean_ptr(self):
        return self.save.data_ptr()

    @property
    def invstd_ptr(self):
        return self.save.data_ptr() + 4 * self.C

    def finalize(self, rows, region=WS_BNP):
        """Batch statistics from the `rows` partial rows waiting in `region` -> scale/shift, moving stats."""
        e = self.e
        L.spnet_bn_finalize_fwd(e.ws_ptr(region), rows, self.M, self.C, L.ptr(self.gamma), L.ptr(self.beta),
                                L.ptr(self.mm), L.ptr(self.mv), self.mean_ptr, self.invstd_ptr, L.ptr(self.ss),
                                BN_EPS, BN_MOMENTUM, _stream())

    def infer(self):
        if not self.e._infer_fresh:          # scale|shift of this plan are still valid for the current weights
            return
        L.spnet_bn_infer_coeffs(self.C, L.ptr(self.gamma), L.ptr(self.beta), L.ptr(self.mm), L.ptr(self.mv),
                                L.ptr(self.ss), BN_EPS, _stream())

    def apply(self, x, y, act, residual=None):
        L.spnet_bn_apply(L.ptr(x), self.M, self.C, L.ptr(self.ss), act, L.ptr(residual), 0, L.ptr(y), _stream())

    def bwd_full(self, x, g, out, act):
        """Stand-alone backward (own reduction pass) of y = act(BN(x)); `out` may alias g."""
        e, C = self.e, self.C
        L.spnet_bn_bwd(L.ptr(x), L.ptr(g), self.M, C, L.ptr(self.gamma), L.ptr(self.beta), self.mean_ptr,
                       self.invstd_ptr, act, L.ptr(out), L.ptr(self.ggamma), L.ptr(self.gbeta), L.ptr(e.small[:3 * C]),
                       e.ws_ptr(WS_MISC), _stream())
        return out

    def bwd_from_partials(self, x, g, out, rows):
        """Backward when the consumer's depthwise-backward epilogue already left (sum g, sum g*xhat)
        as `rows` partial rows in WS_BNP."""
        e, C = self.e, self.C
        L.spnet_bn_bwd_from_partials(L.ptr(x), L.ptr(g), self.M, C, L.ptr(self.gamma), L.ptr(self.beta),
                                     self.mean_ptr, self.invstd_ptr, rows, e.ws_ptr(WS_BNP), L.ptr(out),
                                     L.ptr(self.ggamma), L.ptr(self.gbeta), L.ptr(e.small[:3 * C]), _stream())
        return out


    def bwd_full_x3(self, x, g, out_planes, act):
        """bwd_full with the result written as bf16x3 planes (the operand of the pointwise layer's two backward GEMMs)"""
        e, C = self.e, self.C
        L.spnet_bn_bwd_x3(L.ptr(x), L.ptr(g), self.M, C, L.ptr(self.gamma), L.ptr(self.beta), self.mean_ptr,
                          self.invstd_ptr, act, L.ptr(out_planes), L.ptr(self.ggamma), L.ptr(self.gbeta),
                          L.ptr(e.small[:3 * C]), e.ws_ptr(WS_MISC), _stream())
        return out_planes

    def bwd_from_partials_x3(self, x, g, out_planes, rows):
        """bwd_from_partials with the result written as bf16x3 planes"""
        e, C = self.e, self.C
        L.spnet_bn_bwd_from_partials_x3(L.ptr(x), L.ptr(g), self.M, C, L.ptr(self.gamma), L.ptr(self.beta),
                                        self.mean_ptr, self.invstd_ptr, rows, e.ws_ptr(WS_BNP), L.ptr(out_planes),
                                        L.ptr(self.ggamma), L.ptr(self.gbeta), L.ptr(e.small[:3 * C]), _stream())
        return out_planes


class Ref:
    """A tensor as consumers see it: `t` in HBM, plus (optionally) the BatchNorm whose affine still has to
    be applied on load because the normalised tensor was never materialised."""

    def __init__(self, t, bn=None, stats_bn=None, stats_x=None):
        self.t, self.bn = t, bn
        # A consumer's fused depthwise backward emits the backward sums of `stats_bn` (default: bn), whose
        # pre-normalisation tensor is `stats_x` (default: t).  Set when t is a block OUTPUT BN(stats_x)+residual.
        self.stats_bn, self.stats_x = stats_bn, stats_x


def _dw_fwd(x, w, y, B, H, W, C, relu_in, scale_ptr, shift_ptr):
    """Stride-1 depthwise forward y = dw3x3(relu?(x * scale + shift)): the streaming kernel where the library prefers it
    (every plane: spnet_dwconv3x3_prefers_stream), else the LDS-tiled one -- same bits either way."""
    if L.spnet_dwconv3x3_prefers_stream(B, H, W, C, 0):
        L.spnet_dwconv3x3_stream_fwd(L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, C, relu_in, scale_ptr, shift_ptr, 0, _stream())
    else:
        L.spnet_dwconv3x3_tiled_fwd(L.ptr(x), L.ptr(w), L.ptr(y), B, H, W, C, relu_in, scale_ptr, shift_ptr, _stream())


def _dw_fwd_x3(x, w, zp, B, H, W, C, relu_in, scale_ptr, shift_ptr):
    """_dw_fwd with the output written as bf16x3 planes"""
    if L.spnet_dwconv3x3_prefers_stream(B, H, W, C, 0):
        L.spnet_dwconv3x3_stream_fwd_x3(L.ptr(x), L.ptr(w), L.ptr(zp), B, H, W, C, relu_in, scale_ptr, shift_ptr, 0, _stream())
    else:
        L.spnet_dwconv3x3_tiled_fwd_x3(L.ptr(x), L.ptr(w), L.ptr(zp), B, H, W, C, relu_in, scale_ptr, shift_ptr, _stream())


class _DwBwdPlan:
    """Which fused depthwise backward a layer runs (decided once, when the plan is built) and the sizes that follow from
    it: streaming for planes of >= 2,048 pixels, LDS-tiled below.  rows = partial rows of the [rows][9][C] weight-gradient
    and [rows][2][C] BatchNorm-sum buffers, ws_floats = floats of the weight-gradient workspace."""

    def __init__(self, B, H, W, C):
        self.B, self.H, self.W, self.C = B, H, W, C
        self.stream = bool(L.spnet_dwconv3x3_prefers_stream(B, H, W, C, 1))
        if self.stream:
            self.rows = int(L.spnet_dwconv3x3_stream_rows(B, H, W, C, 0))
            self.ws_floats = int(L.spnet_dwconv3x3_stream_bwd_ws(B, H, W, C, 0))
        else:
            self.rows = int(L.spnet_dwconv3x3_tiled_rows(B, H, W, C))
            self.ws_floats = int(L.spnet_dwconv3x3_tiled_bwd_ws(B, H, W, C))

    def run(self, dz, x, w, dx, relu_in, add, wpart, scale_ptr, shift_ptr, mean_ptr, invstd_ptr, bn_partial_ptr, bn_x):
        """dx, the weight-gradient partial rows (left in wpart for the batched reduction) and, with bn_partial_ptr, the
        producer BatchNorm's backward sums."""
        a = (L.ptr(dz), L.ptr(x), L.ptr(w), L.ptr(dx), None, self.B, self.H, self.W, self.C, relu_in, L.ptr(add),
             L.ptr(wpart), scale_ptr, shift_ptr, mean_ptr, invstd_ptr, bn_partial_ptr, L.ptr(bn_x))
        if self.stream:
            L.spnet_dwconv3x3_stream_bwd(*a, 0, _stream())
        else:
            L.spnet_dwconv3x3_tiled_bwd(*a, _stream())


class SepConvBN:
    """[affine on load] -> relu? -> depthwise 3x3 -> pointwise (+BN sums in the GEMM epilogue) -> BN.

    mode 'lazy'   the BN output is never written; the consumer applies scale/shift (+relu) on load and its
                  fused depthwise backward also produces this BN's backward sums
    mode 'pool'   like lazy for the forward (a max-pool consumes it), stand-alone BN backward
    mode 'apply'  y = act(BN(.)) (+residual) is materialised; stand-alone BN backward
    """

    def __init__(self, eng, src, cin, cout, name, relu_in, mode="lazy", act=ACT_NONE, residual=None,
                 bwd_inplace=True, defer_wgrad=False):
        self.e, self.src, self.cin, self.cout, self.relu_in = eng, src, cin, cout, int(relu_in)
        self.mode, self.act, self.residual, self.bwd_inplace = mode, act, residual, bwd_inplace
        B, H, W, _ = src.t.shape
        self.H, self.W = H, W
        self.M = B * H * W
        self.wd = eng.P(name + "/depthwise_kernel")
        self.yp = eng.new(B, H, W, cout)
        self.pw = Pointwise(eng, self.M, cin, cout, name + "/pointwise_kernel", defer_wgrad=defer_wgrad)
        # Planes mode (round 5): the depthwise kernel writes z, and the BatchNorm backward writes dy, as the bf16x3 planes the
        # pointwise GEMMs read (csrc/x3t.h) -- all three of them then run on the bf16 matrix cores from LDS-DMA'd pieces
        # and no fp32 z / dy exists.  Needs the forward AND (in a training plan) the data-gradient GEMM on the bf16x3 path;
        # a layer with only one of them (cin < 256: blocks 2-3) keeps fp32 operands and the kernel that splits A itself.
        self.x3p = eng.pointwise == "bf16x3" and self.pw.x3_fwd and (self.pw.x3_dgrad or not eng.train_capable)
        self.z = None if self.x3p else eng.new(B, H, W, cin)
        self.zp = eng.new_planes(self.M, cin) if self.x3p else None
        self.bn = BN(eng, cout, self.M, name + "_bn")
        self.y = eng.new(B, H, W, cout) if mode == "apply" else None
        self.dwb = _DwBwdPlan(B, H, W, cin)
        # planes-mode units on planes whose 192-pixel GEMM tiles hold whole images: depthwise backward fused into the
        # data-gradient GEMM (dz never written, one launch less on the dependency chain)
        self.fuse_bwd = bool(self.x3p and eng.train_capable and eng.fuse_dw_bwd and L.spnet_gemm_bf16x3_dwbwd_ok(H, W, cin)
                             and ((self.M + 191) // 192) * ((cin + 95) // 96) >= eng.fuse_min_tiles)
        if self.fuse_bwd:
            self.dwb.rows = int(L.spnet_gemm_bf16x3_dwbwd_rows(self.M))
            self.dwb.ws_floats = self.dwb.rows * 9 * cin
        self.rows_src = self.dwb.rows                                # partial rows this unit emits for src.bn
        if self.rows_src * 2 * cin > WS_BNP[1]:
            raise RuntimeError("workspace regions too small for %s" % name)
        self.consumer_rows = 0              # set by the consumer (lazy mode)
        self.consumer_unit = None           # the SepConvBN that reads this unit's un-materialised BatchNorm output, if any
        self._dw_done = False               # inference: this unit's depthwise output was written by the producer's GEMM epilogue
        self.fin_by_consumer = False        # set by a SepConvBN consumer: it finalizes this unit's BatchNorm forward
        self.pending_rows = 0               # > 0: partial rows waiting in WS_BNP for the consumer's prologue
        if eng.train_capable:
            self.gwd = eng.G(name + "/depthwise_kernel")
            self.dz = None if self.fuse_bwd else eng.new(B, H, W, cin)
            self.dx = eng.new(B, H, W, cin)
            # This unit's depthwise weight-gradient partial sums [rows][9][cin]: kept in a buffer of its own so
            # that ALL units' reductions run as one launch at the end of backward, off the dependency chain.
            self.wpart = eng.new(self.dwb.ws_floats)
            eng.dw_reduce_jobs.append((self.wpart, self.gwd, self.rows_src, 9 * cin))
            # BatchNorm-backward output dy: written once by the blending data-gradient GEMM for the weight gradient
            # (units with an activation behind their BN take the unfused path and use dbn / the incoming buffer)
            self.blend = (act == ACT_NONE or mode != "apply") and self.pw.blend
            self.dyb = eng.new(B, H, W, cout) if self.blend else None
            self.dbn = None if (bwd_inplace or self.blend or self.x3p) else eng.new(B, H, W, cout)
            self.dyp = eng.new_planes(self.M, cout) if self.x3p else None
        if src.bn is not None and hasattr(src, "owner"):
            src.owner.consumer_rows = self.rows_src
            src.owner.consumer_unit = self           # (inference: the producer's GEMM may run this unit's depthwise, below)
            # the producer's training-forward BatchNorm finalize runs inside this unit's depthwise prologue
            # (spnet_dwconv3x3_tiled_fwd_bnfin) whenever its GEMM leaves at most 128 partial rows
            src.owner.fin_by_consumer = eng.train_capable

    def ref(self):
        if self.mode == "apply":
            return Ref(self.y)
        r = Ref(self.yp, self.bn)
        r.owner = self
        return r

    def fwd(self, training):
        e, sb = self.e, self.src.bn
        prof = e.prof
        if prof is not None:
            t0 = prof.start()
        owner = getattr(self.src, "owner", None)
        ran_dw = True
        if self._dw_done and not training:      # the producer's forward GEMM ran this depthwise in its epilogue
            self._dw_done = ran_dw = False
        elif training and owner is not None and owner.pending_rows:
            bnfin = L.spnet_dwconv3x3_tiled_fwd_bnfin_x3 if self.x3p else L.spnet_dwconv3x3_tiled_fwd_bnfin
            bnfin(L.ptr(self.src.t), L.ptr(self.wd), L.ptr(self.zp if self.x3p else self.z), e.B, self.H, self.W,
                  self.cin, self.relu_in, e.ws_ptr(WS_BNP), owner.pending_rows, sb.M,
                  L.ptr(sb.gamma), L.ptr(sb.beta), L.ptr(sb.mm), L.ptr(sb.mv), sb.mean_ptr,
                  sb.invstd_ptr, L.ptr(sb.ss), BN_EPS, BN_MOMENTUM, _stream())
            owner.pending_rows = 0
        else:
            (_dw_fwd_x3 if self.x3p else _dw_fwd)(self.src.t, self.wd, self.zp if self.x3p else self.z, e.B, self.H, self.W,
                                                  self.cin, self.relu_in, sb.scale_ptr if sb else None,
                                                  sb.shift_ptr if sb else None)
        if prof is not None and ran_dw:
            prof.stop("dw", t0, 2.0 * 4 * self.M * self.cin, ("dw fwd", self.H, self.W, self.cin))   # read x + write z
        if training:
            rows = self.pw.fwd_p(self.zp, self.yp, WS_BNP) if self.x3p else self.pw.fwd_colstats(self.z, self.yp)
            if self.fin_by_consumer and rows <= 128:
                self.pending_rows = rows            # the consumer's depthwise is the next launch on this stream
            elif self.mode == "apply":
                bn = self.bn                        # finalize + y = act(BN(yp)) (+ residual) in one launch
                L.spnet_bn_finalize_apply(e.ws_ptr(WS_BNP), rows, L.ptr(self.yp), self.M, self.cout, L.ptr(bn.gamma),
                                          L.ptr(bn.beta), L.ptr(bn.mm), L.ptr(bn.mv), bn.mean_ptr, bn.invstd_ptr,
                                          L.ptr(bn.ss), self.act, L.ptr(self.residual), L.ptr(self.y), BN_EPS,
                                          BN_MOMENTUM, _stream())
                return
            else:
                self.bn.finalize(rows)
        else:
            nxt = self.consumer_unit
            if (self.x3p and self.mode == "lazy" and nxt is not None and nxt.x3p and e.fuse_dw_fwd
                    and L.spnet_gemm_bf16x3_dwbwd_ok(self.H, self.W, self.cout)
                    and ((self.M + 191) // 192) * ((self.cout + 95) // 96) >= e.fuse_min_tiles):
                # inference: pointwise GEMM + this BatchNorm's affine + the consumer's ReLU and depthwise in ONE launch,
                # the result written as the consumer's z planes (spnet_gemm_bf16x3_pp_dwfwd); yp is never written
                self.bn.infer()
                t0 = prof.start() if prof is not None else None
                L.spnet_gemm_bf16x3_pp_dwfwd(L.ptr(self.zp), L.ptr(e._planes[self.pw.wname][0]), e.B, self.H, self.W, self.cin,
                                             self.cout, L.ptr(self.bn.ss), nxt.relu_in, L.ptr(nxt.wd), L.ptr(nxt.zp), _stream())
                if prof is not None:
                    prof.stop("gemm", t0, 2.0 * self.M * self.cin * self.cout, ("x3p aB+dwfwd", self.M, self.cout, self.cin))
                nxt._dw_done = True
                return
            if self.x3p:
                self.pw.fwd_p(self.zp, self.yp)
            else:
                self.pw.fwd(self.z, self.yp)
            self.bn.infer()
        if self.mode == "apply":
            self.bn.apply(self.yp, self.y, self.act, self.residual)

    def bwd(self, g, add=None):
        """g: gradient wrt this unit's BN output (after `act` for mode 'apply').  Returns the gradient wrt
        the source as its consumers see it (wrt src's BN output when src is lazy), plus `add`."""
        e, sb = self.e, self.src.bn
        from_partials = self.mode == "lazy" or (self.consumer_rows and self.act == ACT_NONE)
        if self.blend:
            if from_partials:
                self.bn.coeffs_from_partials(self.consumer_rows)
            else:
                self.bn.coeffs_full(self.yp, g)
            self.pw.bwd_blend(self.z, g, self.yp, self.bn, self.dyb, self.dz)
        elif self.x3p:
            if from_partials:
                self.bn.bwd_from_partials_x3(self.yp, g, self.dyp, self.consumer_rows)
            else:
                self.bn.bwd_full_x3(self.yp, g, self.dyp, self.act if self.mode == "apply" else ACT_NONE)
            self.pw.bwd_p(self.zp, self.dyp, self.dz)        # (fuse_bwd: dz is None -> the weight gradient only)
            if self.fuse_bwd:
                st = self.src.stats_bn if self.src.stats_bn is not None else sb
                prof = e.prof
                t0 = prof.start() if prof is not None else None
                L.spnet_gemm_bf16x3_pp_dwbwd(L.ptr(self.dyp), L.ptr(e._planes[self.pw.wname][1]), e.B, self.H, self.W, self.cin,
                                             self.cout, L.ptr(self.src.t), L.ptr(self.wd), L.ptr(self.dx), self.relu_in,
                                             L.ptr(add), L.ptr(self.wpart), sb.scale_ptr if sb else None,
                                             sb.shift_ptr if sb else None, st.mean_ptr if st else None,
                                             st.invstd_ptr if st else None, e.ws_ptr(WS_BNP) if st else None,
                                             L.ptr(self.src.stats_x), _stream())
                if prof is not None:
                    prof.stop("gemm", t0, 2.0 * self.M * self.cin * self.cout, ("x3p ab+dwbwd", self.M, self.cin, self.cout))
                return self.dx
        else:
            out = g if self.bwd_inplace else self.dbn
            if from_partials:
                dy = self.bn.bwd_from_partials(self.yp, g, out, self.consumer_rows)
            else:
                dy = self.bn.bwd_full(self.yp, g, out, self.act if self.mode == "apply" else ACT_NONE)
            self.pw.bwd(self.z, dy, self.dz)
        st = self.src.stats_bn if self.src.stats_bn is not None else sb     # whose backward sums to emit
        prof = e.prof
        if prof is not None:
            t0 = prof.start()
        self.dwb.run(self.dz, self.src.t, self.wd, self.dx, self.relu_in, add, self.wpart,
                     sb.scale_ptr if sb else None, sb.shift_ptr if sb else None,
                     st.mean_ptr if st else None, st.invstd_ptr if st else None,
                     e.ws_ptr(WS_BNP) if st else None, self.src.stats_x)
        if prof is not None:
            prof.stop("dw", t0, 3.0 * 4 * self.M * self.cin, ("dw bwd", self.H, self.W, self.cin))   # read dz, read x, write dx
        return self.dx


class MiddleBlock(Node):
    """Xception blocks 5-12: x + [relu, sepconv, BN] x 3.  Only the block output is materialised."""

    def __init__(self, eng, x, b, C, prev=None):
        """prev: the middle block whose output is x -- its closing BatchNorm's backward sums then come out of
        this block's first depthwise backward (which reads yp of that BN beside x), saving a reduction pass."""
        self.x = x
        self.pnames = [n for k in (1, 2, 3) for n in ("block%d_sepconv%d" % (b, k), "block%d_sepconv%d_bn" % (b, k))]
        src = Ref(x)
        if prev is not None and eng.train_capable:
            src = Ref(x, stats_bn=prev.u3.bn, stats_x=prev.u3.yp)
        # The 24 middle-flow weight gradients (728 x 728, K = batch*12*16 pixels) are each too small to fill
        # the chip without a K split; their operands (z, dy) stay intact until the next step, so they are
        # collected here and run as ONE batched GEMM once block 5 has been back-propagated.
        d = eng.train_capable
        self.u1 = SepConvBN(eng, src, C, C, "block%d_sepconv1" % b, True, mode="lazy", defer_wgrad=d)
        if src.stats_bn is not None:
            prev.u3.consumer_rows = self.u1.rows_src
        self.u2 = SepConvBN(eng, self.u1.ref(), C, C, "block%d_sepconv2" % b, True, mode="lazy", defer_wgrad=d)
        # u3's BN backward writes to its own buffer: the incoming gradient is also the identity branch's
        # gradient and is added back in u1's depthwise backward.
        self.u3 = SepConvBN(eng, self.u2.ref(), C, C, "block%d_sepconv3" % b, True, mode="apply", residual=x,
                            bwd_inplace=False, defer_wgrad=d)
        self.y = self.u3.y

    def fwd(self, training):
        self.u1.fwd(training)
        self.u2.fwd(training)
        self.u3.fwd(training)

    def bwd(self, g):
        d = self.u3.bwd(g)                  # g itself survives (out-of-place BN backward)
        d = self.u2.bwd(d)
        return self.u1.bwd(d, add=g)


class StridedBlock(Node):
    """Xception blocks 2,3,4,13: maxpool(BN(sepconv x2)) + BN(conv1x1/s2); both BN affines are applied
    inside the pooling kernel."""

    def __init__(self, eng, x, b, cin, c1, c2, first_relu, conv_name, bn_name):
        self.e, self.x, self.cin, self.c2 = eng, x, cin, c2
        self.pnames = [conv_name, bn_name] + [n for k in (1, 2) for n in ("block%d_sepconv%d" % (b, k),
                                                                           "block%d_sepconv%d_bn" % (b, k))]
        B, H, W, _ = x.shape
        self.H, self.W = H, W
        OH, OW = (H + 1) // 2, (W + 1) // 2
        self.OH, self.OW = OH, OW
        self.Ms = B * OH * OW
        self.xs = eng.new(B, OH, OW, cin)
        self.yr = eng.new(B, OH, OW, c2)
        self.pwr = Pointwise(eng, self.Ms, cin, c2, conv_name + "/kernel")
        self.bnr = BN(eng, c2, self.Ms, bn_name)
        self.u1 = SepConvBN(eng, Ref(x), cin, c1, "block%d_sepconv1" % b, first_relu, mode="lazy")
        self.u2 = SepConvBN(eng, self.u1.ref(), c1, c2, "block%d_sepconv2" % b, True, mode="pool")
        self.y = eng.new(B, OH, OW, c2)
        if eng.train_capable:
            self.idx = torch.empty(B * OH * OW * (c2 // 4), device=eng.dev, dtype=torch.int32)
            self.dxs = eng.new(B, OH, OW, cin)
            self.blend = self.pwr.blend
            self.dyr = eng.new(B, OH, OW, c2) if self.blend else None
            self.dpool = eng.new(B, H, W, c2)
            # the pooling backward produces dL/d(BN output of u2) and, in the same pass, that BatchNorm's two backward
            # sums (spnet_maxpool3x3s2_bwd_bnsums): u2 then takes the from-partials path, no reduction pass of its own
            # (the partial rows are the kernel's workgroup rows: 128 where u2's BatchNorm backward is the one-launch
            # finalize + apply, 1,024 where u2 only derives blend coefficients from the sums)
            rows = L.spnet_maxpool3x3s2_bwd_rows(B, H, W, c2, 1024 if self.u2.blend else 128)
            self.pool_stats = rows * 2 * c2 <= WS_BNP[1]      # (else: stand-alone reduction pass inside u2's backward)
            if self.pool_stats:
                self.u2.consumer_rows = self.pool_rows = rows
        else:
            self.idx = None

    def fwd(self, training):
        e = self.e
        side = e.wgrad_stream if training else None
        main = torch.cuda.current_stream()
        if side is not None:      # the residual branch only meets the main branch at the pooling kernel
            side.wait_stream(main)
            torch.cuda.set_stream(side)
        L.spnet_gather_s2(L.ptr(self.x), L.ptr(self.xs), e.B, self.H, self.W, self.cin, _stream())
        if training:
            region = WS_BNP2 if side is not None else WS_BNP
            self.bnr.finalize(self.pwr.fwd_colstats(self.xs, self.yr, region=region), region=region)
        else:
            self.pwr.fwd(self.xs, self.yr)
            self.bnr.infer()
        if side is not None:
            torch.cuda.set_stream(main)
        self.u1.fwd(training)
        self.u2.fwd(training)
        if side is not None:
            main.wait_stream(side)
        L.spnet_maxpool3x3s2_add_fwd(L.ptr(self.u2.yp), L.ptr(self.yr), L.ptr(self.y),
                                     L.ptr(self.idx) if training else None, e.B, self.H, self.W, self.c2,
                                     L.ptr(self.u2.bn.ss), L.ptr(self.bnr.ss), _stream())

    def bwd(self, g):
        e = self.e
        if self.pool_stats:
            bn2 = self.u2.bn
            L.spnet_maxpool3x3s2_bwd_bnsums(L.ptr(g), L.ptr(self.idx), L.ptr(self.dpool), e.B, self.H, self.W, self.c2,
                                            L.ptr(self.u2.yp), bn2.mean_ptr, bn2.invstd_ptr, e.ws_ptr(WS_BNP),
                                            self.pool_rows, _stream())
        else:
            L.spnet_maxpool3x3s2_bwd(L.ptr(g), L.ptr(self.idx), L.ptr(self.dpool), e.B, self.H, self.W, self.c2, _stream())
        if self.blend:
            self.bnr.coeffs_full(self.yr, g)
            self.pwr.bwd_blend(self.xs, g, self.yr, self.bnr, self.dyr, self.dxs)
        else:
            gr = self.bnr.bwd_full(self.yr, g, g, ACT_NONE)       # in place: g is dead after the pool backward
            self.pwr.bwd(self.xs, gr, self.dxs)
        d = self.u2.bwd(self.dpool)
        dx = self.u1.bwd(d)
        L.spnet_scatter_add_s2(L.ptr(self.dxs), L.ptr(dx), e.B, self.H, self.W, self.cin, _stream())
        return dx


class ExitBlock(Node):
    """Xception block 14: sepconv-BN-relu x 2 (no pre-activation); the first ReLU is applied by the second
    depthwise on load, the last one when the block output is materialised for the Dense head."""

    def __init__(self, eng, x, cin, c1, c2):
        self.pnames = ["block14_sepconv1", "block14_sepconv1_bn", "block14_sepconv2", "block14_sepconv2_bn"]
        self.u1 = SepConvBN(eng, Ref(x), cin, c1, "block14_sepconv1", False, mode="lazy")
        self.u2 = SepConvBN(eng, self.u1.ref(), c1, c2, "block14_sepconv2", True, mode="apply", act=ACT_RELU)
        self.y = self.u2.y

    def fwd(self, training):
        self.u1.fwd(training)
        self.u2.fwd(training)

    def bwd(self, g):
        return self.u1.bwd(self.u2.bwd(g))


class _T:
    """A tensor of the Inception-ResNet-v2 program: forward buffer + the gradient accumulated from its consumers."""

    def __init__(self, buf):
        self.buf, self.g = buf, None
        self.owner, self.consumers = None, 0      # producing op; number of ops that read the tensor


class IRv2Backbone(Node):
    """keras InceptionResNetV2(include_top=False) behind the stem (spnet/models.py:357-359, cf.basemodel =
    'InceptionResNetV2'): irv2_program() executed op by op.  Convolutions = [patch gather +] fp32 MFMA GEMM
    (spnet_patches + spnet_gemm_f32; 1x1 convs multiply the activation matrix directly; the 3-channel first conv runs
    the direct kernel), BatchNormalization(scale=False) + ReLU through the stand-alone BN kernels, Concatenate and
    the branch fan-out as channel-block copies / gradient accumulation (spnet_copy_cols), x + scale*up through
    spnet_resadd.  Backward walks the program in reverse; a tensor's gradient is complete when its producer is reached."""

    def __init__(self, eng, x):
        self.e = eng
        self.ops = []
        self.pnames = []
        cur = {"in": _T(x)}
        self.t_in = cur["in"]
        B = eng.B
        self.ones = {}
        max_dcol = 0
        for op in irv2_program():
            kind = op[0]
            if kind == "conv":
                _, cname, bname, src, dst, cin, cout, kk, stride, same, relu, bias = op
                o = _IRConv(eng, self, cur[src], cname, bname, cin, cout, kk, stride, same, relu, bias)
                max_dcol = max(max_dcol, o.col_floats)
                self.pnames += [cname] + ([bname] if bname else [])
                cur[dst] = o.out
            elif kind in ("maxpool", "avgpool"):
                o = _IRPool(eng, cur[op[1]], kind)
                cur[op[2]] = o.out
            elif kind == "concat":
                o = _IRConcat(eng, [cur[s_] for s_ in op[1]])
                cur[op[2]] = o.out
            else:
                _, xs, up, dst, scale, relu = op
                o = _IRResAdd(eng, cur[xs], cur[up], scale, relu)
                cur[dst] = o.out
            o.out.owner = o
            self.ops.append(o)
        self.y = cur["x"].buf
        self.t_out = cur["x"]
        for o in self.ops:
            for t in o.srcs():
                t.consumers += 1
        # Sibling 1x1 convolutions (the first layer of every branch of a block reads the block input) as ONE GEMM + ONE
        # BatchNormalization (_IRGroup).  In training every member needs its one consumer to be a k x k convolution or a
        # Concatenate (they leave the masked gradient and the BatchNorm sums in the group's blocks): true for every
        # group of the network.
        self.groups, self._wm_jobs, self._wm_ver = [], None, -1
        consumer = {}
        for o in self.ops:
            for t in o.srcs():
                consumer[id(t)] = o
        by_src = OrderedDict()
        for o in self.ops:
            if isinstance(o, _IRConv) and o.direct and not o.bias and not o.small and o.relu:
                by_src.setdefault(id(o.src), []).append(o)
        for ms in by_src.values():
            ok = len(ms) >= 2
            for m in ms:
                c = consumer.get(id(m.out))
                ok = ok and m.out.consumers == 1 and (isinstance(c, _IRConcat) or
                                                      (isinstance(c, _IRConv) and not c.direct and not c.small))
            if ok:
                self.groups.append(_IRGroup(eng, self, ms))
        # A conv2d_bn branch whose only consumer is a Concatenate writes its output straight into that buffer (its
        # BatchNorm apply pass gets the row stride of the concatenated tensor): no copy pass in forward.  The tensor
        # stays a strided view for everyone who looks at it (the Concatenate backward reads the ReLU mask from it).
        for o in self.ops:
            if not isinstance(o, _IRConcat):
                continue
            ct, off = sum(o.cs), 0
            for i, (t, c) in enumerate(zip(o.srcs_, o.cs)):
                p = t.owner
                if isinstance(p, _IRConv) and not p.bias and not p.small and t.consumers == 1 and p.group is None:
                    t.buf = o.out.buf[..., off:off + c]
                    p.ldy = ct
                    o.direct[i] = True
                off += c
        # A conv2d_bn output with ONE consumer that is a k x k convolution or a Concatenate gets its BatchNorm-backward
        # sums from that consumer's gradient pass (spnet_patches_bwd_bnsums / spnet_copy_cols_bnsums) instead of a
        # reduction pass of its own.
        if eng.train_capable:
            for o in self.ops:
                for t in o.srcs():
                    p = t.owner
                    ok = (isinstance(p, _IRConv) and not p.bias and t.consumers == 1 and p.group is None and
                          ((isinstance(o, _IRConv) and not o.direct and not o.small) or isinstance(o, _IRConcat)))
                    if ok:
                        # (up to 512 rows where the one-launch BatchNorm backward takes them: bn_fuse_ok in bn.hip)
                        p.sum_rows = int(L.spnet_grad_bnsums_rows(p.M, 512 if p.M * p.cout <= (4 << 20) else 128))
                        p.sum_part = eng.new(p.sum_rows * 2 * p.cout)
        # one scratch for every conv's patch-matrix gradient (used and consumed on the main stream, op by op)
        self.dcol = eng.new(max_dcol) if (eng.train_capable and max_dcol) else None
        # Weight gradients of the repeated blocks (10 x block35, 20 x block17, 10 x block8: 17 shapes cover 226 of the
        # 244 convolutions) are deferred to the end of backward and run as ONE batched GEMM per shape with a K split
        # (spnet_gemm_f32_batched_splitk) instead of a small split-K GEMM + slab reduce per layer: their operands --
        # the patch matrix / block input and the BatchNorm-backward output left in the gradient accumulator -- stay
        # untouched until the next step.
        self.wg_groups, self.wg_ws, self._wg_trigger = [], None, None
        if eng.train_capable:
            by_shape = {}
            for o in self.ops:
                if isinstance(o, _IRConv) and not o.small:
                    ldb = o.group.Ct if o.group is not None else o.cout      # dy of a sibling-group member: a column block
                    by_shape.setdefault((o.K, o.cout, o.M, ldb), []).append(o)
            need = 0
            for (K, C, M, ldb), members in by_shape.items():
                if len(members) < 2:
                    continue
                tile = __import__("ctypes").c_int(0)
                ksl = int(L.spnet_gemm_batched_ksplit(K, C, M, len(members), __import__("ctypes").addressof(tile)))
                for o in members:
                    o.deferred_wgrad = True
                self.wg_groups.append(dict(members=members, K=K, C=C, M=M, ldb=ldb, ksl=ksl, tile=tile.value, table=None,
                                           key=None))
                if ksl > 1:
                    need = max(need, len(members) * ksl * K * C)
            self.wg_ws = eng.new(need) if need else None

    def const_ones(self, C):
        if C not in self.ones:
            self.ones[C] = (torch.ones(C, device=self.e.dev, dtype=torch.float32), self.e.new(C))
        return self.ones[C]

    def refresh_wm(self):
        """The kernels of every sibling group gathered side by side into its GEMM operand: one launch for the network."""
        if not self.groups:
            return
        e = self.e
        if self._wm_jobs is None:
            flat, mx = [], 4
            for G in self.groups:
                for m in G.members:
                    flat += [m.w.data_ptr(), G.Wm.data_ptr() + 4 * m.c0, G.cin, m.cout, m.cout, G.Ct]
                    mx = max(mx, G.cin * m.cout)
            self._wm_jobs = (torch.tensor(flat, dtype=torch.int64, device=e.dev), len(flat) // 6, mx)
        table, nj, mx = self._wm_jobs
        L.spnet_copy_cols_batched(table.data_ptr(), nj, mx, _stream())
        self._wm_ver = e._tver[0]

    def fwd(self, training):
        # (training: every step follows an optimizer step -- and a captured step replays without host code, so the gather
        # is part of the step; inference: only when the weights changed since the last gather)
        if training or self._wm_ver != self.e._tver[0]:
            self.refresh_wm()
        for o in self.ops:
            o.fwd(training)

    def bwd(self, g):
        for o in self.ops:
            o.out.g = getattr(o, "g_view", None)      # members of a sibling group: their block of the group's gradient
        self.t_in.g = None
        self.t_out.g = g
        if self._wg_trigger is None:
            # a shape's batched weight gradient is launched as soon as its LAST member (the earliest layer; for a member
            # of a sibling group, the group's leader, whose bwd differentiates the whole group) has been back-propagated:
            # all of them at the end of backward left the weight-gradient stream a 1.9 ms tail with nothing beside it
            pos = {id(o): i for i, o in enumerate(self.ops)}
            lead = lambda m: m if m.group is None else m.group.members[0]
            self._wg_trigger = {}
            for g in self.wg_groups:
                i = min(pos[id(lead(m))] for m in g["members"])
                self._wg_trigger.setdefault(i, []).append(g)
        for i in range(len(self.ops) - 1, -1, -1):
            self.ops[i].bwd(self)
            if i in self._wg_trigger:
                self.flush_wgrads(self._wg_trigger[i])
        return self.t_in.g

    def flush_wgrads(self, groups=None):
        """Deferred weight gradients, one batched launch (+ one slab reduce) per shape, on the weight-gradient stream.
        The operand offset tables live in device memory and are rebuilt only if a buffer address changed."""
        groups = self.wg_groups if groups is None else groups
        if not groups:
            return
        e = self.e

        def run():
            for g in groups:
                ms = g["members"]
                for o in ms:
                    o.gather_patches()
                a0, b0, c0 = ms[0]._A().data_ptr(), ms[0].out.g.data_ptr(), ms[0].gw.data_ptr()
                offs = tuple(v for o in ms for v in ((o._A().data_ptr() - a0) // 4, (o.out.g.data_ptr() - b0) // 4,
                                                     (o.gw.data_ptr() - c0) // 4))
                if g["key"] != (a0, b0, c0) + offs:
                    g["key"] = (a0, b0, c0) + offs
                    g["table"] = torch.tensor(offs, dtype=torch.int64, device=e.dev)
                K, C, M, nb = g["K"], g["C"], g["M"], len(ms)
                prof = e.prof
                t0 = prof.start() if prof is not None else None
                L.spnet_gemm_f32_batched_splitk(a0, b0, c0, g["table"].data_ptr(), nb, OUT_MAJOR, K, OUT_MAJOR, g["ldb"], C, K, C, M,
                                                g["tile"], g["ksl"], L.ptr(self.wg_ws),
                                                self.wg_ws.numel() if self.wg_ws is not None else 0, _stream())
                if prof is not None:
                    prof.stop("gemm", t0, 2.0 * nb * K * C * M, ("AB x%d batched" % nb, K, C, M))

        side = e.wgrad_stream
        if side is None:
            run()
        else:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                run()


def _ir_acc(t, gbuf, eng):
    """Add an op's input gradient to the tensor's accumulator (the first contribution becomes the accumulator)."""
    if t.g is None:
        t.g = gbuf
    else:
        C = gbuf.shape[-1]
        L.spnet_copy_cols(L.ptr(gbuf), C, L.ptr(t.g), C, gbuf.numel() // C, C, 1, _stream())


class _IRGroup:
    """The 1x1 / stride-1 conv2d_bn layers that read ONE tensor (the first layer of every branch of an inception block) as
    one GEMM over their kernels side by side [cin][Ct] and one BatchNormalization over the Ct concatenated channels.
    The members' beta / moving statistics sit back to back in the flat parameter buffers (param_packs), so the merged
    arrays are plain slices; their kernels are gathered into Wm by IRv2Backbone.refresh_wm (one launch for the whole
    network).  Members keep their identity: output / pre-normalisation tensor / gradient are column blocks of the
    group's tensors, their consumers write the masked gradient and the BatchNorm sums into those blocks
    (spnet_*_bnsums_ld), and their weight gradients run with the rest of their shape in the batched launches."""

    def __init__(self, eng, net, members):
        self.e, self.members = eng, members
        m0 = members[0]
        self.src, self.cin, self.M = m0.src, m0.cin, m0.M
        self.Ct = Ct = sum(m.cout for m in members)
        B, OH, OW = m0.out.buf.shape[:3]
        self.yp, self.y = eng.new(B, OH, OW, Ct), eng.new(B, OH, OW, Ct)
        self.Wm = eng.new(self.cin, Ct)
        off, n, _ = eng.p_off[m0.bname + "/beta"]
        self.beta = eng.theta[off:off + Ct]
        so, _, _ = eng.s_off[m0.bname + "/moving_mean"]
        sv, _, _ = eng.s_off[m0.bname + "/moving_variance"]
        self.mm, self.mv = eng.stats[so:so + Ct], eng.stats[sv:sv + Ct]
        self.ones, self.gscr = net.const_ones(Ct)
        self.ss = eng.new(2 * Ct)
        tr = eng.train_capable
        if tr:
            self.gbeta = eng.grad[off:off + Ct]
            self.save = eng.new(2 * Ct)
            self.g = eng.new(B, OH, OW, Ct)
            self.rows = int(L.spnet_grad_bnsums_rows(self.M, 512 if self.M * Ct <= (4 << 20) else 128))
            self.sum_part = eng.new(self.rows * 2 * Ct)
            self.dx = eng.new(*self.src.buf.shape)
        c0 = 0
        for m in members:
            if not m.relu or m.bias:
                raise RuntimeError("sibling groups are conv2d_bn layers with a ReLU")
            po = eng.p_off[m.bname + "/beta"][0]
            if po != off + c0 or eng.s_off[m.bname + "/moving_mean"][0] != so + c0:
                raise RuntimeError("parameters of %s are not packed behind its siblings'" % m.bname)
            m.group, m.c0, m.ldy = self, c0, Ct
            m.out.buf = self.y[..., c0:c0 + m.cout]
            m.yp = self.yp[..., c0:c0 + m.cout]
            if tr:
                m.g_view = self.g[..., c0:c0 + m.cout]
            c0 += m.cout

    def fwd(self, training):
        e, Ct, x = self.e, self.Ct, self.src.buf
        act = ACT_RELU
        if training:
            rows = _gemm_colstats(x, self.cin, self.Wm, Ct, self.yp, Ct, self.M, Ct, self.cin, e)
            L.spnet_bn_finalize_apply_ld(e.ws_ptr(WS_BNP), rows, L.ptr(self.yp), self.M, Ct, L.ptr(self.ones), L.ptr(self.beta),
                                         L.ptr(self.mm), L.ptr(self.mv), L.ptr(self.save), self.save[Ct:].data_ptr(),
                                         L.ptr(self.ss), act, None, L.ptr(self.y), Ct, BN_EPS, BN_MOMENTUM, _stream())
            return
        _gemm(x, K_MAJOR, self.cin, self.Wm, OUT_MAJOR, Ct, self.yp, Ct, self.M, Ct, self.cin, e)
        L.spnet_bn_fwd_infer_ld(L.ptr(self.yp), self.M, Ct, L.ptr(self.ones), L.ptr(self.beta), L.ptr(self.mm),
                                L.ptr(self.mv), L.ptr(self.ss), act, None, 0, L.ptr(self.y), Ct, BN_EPS, _stream())

    def bwd(self, net):
        """Runs where the group's FIRST member stands in the program, i.e. last of the block in backward order: every
        member's consumer has left its masked gradient block in self.g and its two sums in self.sum_part by then."""
        e, Ct, g = self.e, self.Ct, self.g
        L.spnet_bn_bwd_from_partials(L.ptr(self.yp), L.ptr(g), self.M, Ct, L.ptr(self.ones), L.ptr(self.beta),
                                     L.ptr(self.save), self.save[Ct:].data_ptr(), self.rows, L.ptr(self.sum_part), L.ptr(g),
                                     L.ptr(self.gscr), L.ptr(self.gbeta), L.ptr(e.small[:3 * Ct]), _stream())
        side = e.wgrad_stream
        late = [m for m in self.members if not m.deferred_wgrad]      # (shapes that occur once: no batched launch)
        if late:
            def wgrads(region):
                for m in late:
                    _gemm(self.src.buf, OUT_MAJOR, self.cin, m.g_view, OUT_MAJOR, Ct, m.gw, m.cout, self.cin, m.cout,
                          self.M, e, region=region)
            if side is None:
                wgrads(WS_GEMM)
            else:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    wgrads(WS_GEMM2)
        src = self.src
        if src.g is not None:
            prof = e.prof
            t0 = prof.start() if prof is not None else None
            L.spnet_gemm_f32_accumulate(L.ptr(g), K_MAJOR, Ct, L.ptr(self.Wm), K_MAJOR, Ct, L.ptr(src.g), self.cin, self.M,
                                        self.cin, Ct, _tile_for(K_MAJOR, K_MAJOR, 0, self.M, self.cin, Ct, 0), _stream())
            if prof is not None:
                prof.stop("gemm", t0, 2.0 * self.M * self.cin * Ct, ("aA+=", self.M, self.cin, Ct))
        else:
            _gemm(g, K_MAJOR, Ct, self.Wm, K_MAJOR, Ct, self.dx, self.cin, self.M, self.cin, Ct, e)
            _ir_acc(src, self.dx, e)


class _IRConv:
    def __init__(self, eng, net, src, cname, bname, cin, cout, kk, stride, same, relu, bias):
        self.e, self.src, self.cin, self.cout = eng, src, cin, cout
        self.kh, self.kw, self.stride, self.same, self.relu, self.bias = kk[0], kk[1], stride, int(same), relu, bias
        self.cname, self.bname = cname, bname
        self.group, self.c0, self.g_view = None, 0, None      # set by _IRGroup: member of a sibling group
        B, H, W, _ = src.buf.shape
        self.H, self.W = H, W
        if same:
            OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        else:
            OH, OW = (H - self.kh) // stride + 1, (W - self.kw) // stride + 1
        self.M, self.K = B * OH * OW, self.kh * self.kw * cin
        self.direct = (self.kh == 1 and self.kw == 1 and stride == 1)
        self.small = (cin == 3)                      # the first conv: 3 -> 32, 3x3 / stride 2 / valid (stem.hip)
        self.deferred_wgrad = False                  # set by IRv2Backbone: dW comes out of a batched launch per shape
        self.sum_rows, self.sum_part = 0, None       # set by IRv2Backbone: my BatchNorm-backward sums come from my consumer
        self.ldy = cout                              # row stride of my output (IRv2Backbone: the Concatenate's width)
        # k x k convolutions with cin a multiple of the 32-deep K tile run forward on the TUNED GEMM kernel with its A
        # tiles gathered while they are staged (spnet_conv_gemm_f32: same tiles, same pipeline, same bits as patch gather
        # + GEMM) -- no patch gather launch on the main stream's chain; the weight gradient gathers its patch matrix
        # itself, on its own stream.  The rest (cin = 48, 80 ...) gathers a patch matrix first.  (A separate 64x64-tile
        # implicit kernel was measured slower than gather + GEMM in round 3 and removed in round 4.)
        self.gathered = not self.direct and not self.small and cin % 32 == 0
        self.w = eng.P(cname + "/kernel")
        # (an inference plan whose convolution gathers its A tiles never needs the patch matrix)
        no_col = self.direct or self.small or (not eng.train_capable and self.gathered)
        self.col = None if no_col else eng.new(self.M, self.K)
        self.col_floats = 0 if (self.direct or self.small) else self.M * self.K
        self.out = _T(eng.new(B, OH, OW, cout))
        tr = eng.train_capable
        if bias:
            self.b = eng.P(cname + "/bias")
            self.gb = eng.G(cname + "/bias") if tr else None
        else:
            self.yp = eng.new(B, OH, OW, cout)
            self.ones, self.gscr = net.const_ones(cout)
            self.beta = eng.P(bname + "/beta")
            self.mm, self.mv = eng.S(bname + "/moving_mean"), eng.S(bname + "/moving_variance")
            self.ss = eng.new(2 * cout)
            if tr:
                self.gbeta = eng.G(bname + "/beta")
                self.save = eng.new(2 * cout)
        if tr:
            self.gw = eng.G(cname + "/kernel")
            self.dx = eng.new(*src.buf.shape)

    def _A(self):
        return self.src.buf if self.direct else self.col

    def srcs(self):
        return [self.src]

    def has_sums(self):
        """My BatchNorm-backward sums (and the ReLU mask on my gradient) are produced by my one consumer."""
        return bool(self.sum_rows) or (self.group is not None and self.e.train_capable)

    def sum_views(self):
        """Where my consumer finds my pre-normalisation tensor / saved statistics and leaves the masked gradient and the
        partial sums: (yp, ldyp, mean, invstd, partial, ldp, rows, gradient destination or None, its row stride)."""
        G = self.group
        if G is None:
            C = self.cout
            return (L.ptr(self.yp), C, L.ptr(self.save), self.save[C:].data_ptr(), L.ptr(self.sum_part), C, self.sum_rows,
                    None, C)
        c4 = 4 * self.c0
        return (G.yp.data_ptr() + c4, G.Ct, G.save.data_ptr() + c4, G.save.data_ptr() + 4 * G.Ct + c4,
                G.sum_part.data_ptr() + c4, G.Ct, G.rows, G.g.data_ptr() + c4, G.Ct)

    def gather_patches(self):
        """The patch matrix of my input, for the weight-gradient GEMM (gathered-A convolutions skipped it in forward)."""
        if self.gathered:
            L.spnet_patches_ld(L.ptr(self.src.buf), self.src.buf.stride(2), L.ptr(self.col), self.e.B, self.H, self.W,
                               self.cin, self.kh, self.kw, self.stride, self.same, _stream())

    def fwd(self, training):
        if self.group is not None:             # one GEMM + one BatchNormalization for the whole sibling group
            if self.group.members[0] is self:
                self.group.fwd(training)
            return
        e, C = self.e, self.cout
        y = self.out.buf
        dst = y if self.bias else self.yp
        if self.small:
            L.spnet_conv3x3_small(0, 3, C, self.stride, 0, L.ptr(self.src.buf), L.ptr(self.w), L.ptr(dst), e.B, self.H,
                                  self.W, e.ws_ptr(WS_MISC), WS_MISC[1], _stream())
        elif self.gathered:
            stats = training and not self.bias
            prof = e.prof
            t0 = prof.start() if prof is not None else None
            L.spnet_conv_gemm_f32(L.ptr(self.src.buf), self.src.buf.stride(2), L.ptr(self.w), L.ptr(dst), C, e.B, self.H,
                                  self.W, self.cin, C, self.kh, self.kw, self.stride, self.same,
                                  L.ptr(self.b) if self.bias else None,
                                  _tile_for(K_MAJOR, OUT_MAJOR, 1 if stats else 0, self.M, C, self.K, 0),
                                  e.ws_ptr(WS_BNP) if stats else None,
                                  __import__("ctypes").addressof(_stat_rows) if stats else None, _stream())
            if prof is not None:
                prof.stop("gemm", t0, 2.0 * self.M * C * self.K, ("conv gathered", self.M, C, self.K))
                prof.keys[len(prof.records) - 1] = (K_MAJOR, OUT_MAJOR, 1 if stats else 0, self.M, C, self.K)
            if stats:
                L.spnet_bn_finalize_apply_ld(e.ws_ptr(WS_BNP), _stat_rows.value, L.ptr(self.yp), self.M, C, L.ptr(self.ones),
                                             L.ptr(self.beta), L.ptr(self.mm), L.ptr(self.mv), L.ptr(self.save),
                                             self.save[C:].data_ptr(), L.ptr(self.ss), ACT_RELU if self.relu else ACT_NONE,
                                             None, L.ptr(y), self.ldy, BN_EPS, BN_MOMENTUM, _stream())
                return
        else:
            if not self.direct:       # (the input may be one member's column block of a sibling group's output)
                L.spnet_patches_ld(L.ptr(self.src.buf), self.src.buf.stride(2), L.ptr(self.col), e.B, self.H, self.W,
                                   self.cin, self.kh, self.kw, self.stride, self.same, _stream())
            if training and not self.bias:
                # BatchNorm statistics out of the GEMM accumulators, finalize + normalise + ReLU in one more launch
                # (two when the GEMM leaves more than 128 partial rows): no reduction pass over yp
                rows = _gemm_colstats(self._A(), self.K, self.w, C, dst, C, self.M, C, self.K, e)
                L.spnet_bn_finalize_apply_ld(e.ws_ptr(WS_BNP), rows, L.ptr(self.yp), self.M, C, L.ptr(self.ones),
                                             L.ptr(self.beta), L.ptr(self.mm), L.ptr(self.mv), L.ptr(self.save),
                                             self.save[C:].data_ptr(), L.ptr(self.ss), ACT_RELU if self.relu else ACT_NONE,
                                             None, L.ptr(y), self.ldy, BN_EPS, BN_MOMENTUM, _stream())
                return
            _gemm(self._A(), K_MAJOR, self.K, self.w, OUT_MAJOR, C, dst, C, self.M, C, self.K, e,
                  bias=self.b if self.bias else None)
        if self.bias:
            return
        act = ACT_RELU if self.relu else ACT_NONE
        if training:      # (only the 3-channel first convolution gets here in training: its GEMM-free kernel leaves no sums)
            L.spnet_bn_fwd_train_ld(L.ptr(self.yp), self.M, C, L.ptr(self.ones), L.ptr(self.beta), L.ptr(self.mm),
                                    L.ptr(self.mv), L.ptr(self.save), self.save[C:].data_ptr(), L.ptr(self.ss), act, None, 0,
                                    L.ptr(y), self.ldy, BN_EPS, BN_MOMENTUM, e.ws_ptr(WS_MISC), _stream())
        else:
            L.spnet_bn_fwd_infer_ld(L.ptr(self.yp), self.M, C, L.ptr(self.ones), L.ptr(self.beta), L.ptr(self.mm),
                                    L.ptr(self.mv), L.ptr(self.ss), act, None, 0, L.ptr(y), self.ldy, BN_EPS, _stream())

    def bwd(self, net):
        if self.group is not None:
            if self.group.members[0] is self:
                self.group.bwd(net)
            return
        e, C, g = self.e, self.cout, self.out.g
        if self.bias:       # a weight gradient like the kernel's: off the data-gradient chain (two launches per block)
            if e.wgrad_stream is None:
                L.spnet_reduce_rows_ws(L.ptr(g), self.M, C, L.ptr(self.gb), e.ws_ptr(WS_MISC), WS_MISC[1], _stream())
            else:
                e.wgrad_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(e.wgrad_stream):
                    L.spnet_reduce_rows_ws(L.ptr(g), self.M, C, L.ptr(self.gb), e.ws_ptr(WS_GEMM2), WS_GEMM2[1], _stream())
        elif self.sum_rows:     # my one consumer left the masked gradient in g and the two sums in sum_part
            L.spnet_bn_bwd_from_partials(L.ptr(self.yp), L.ptr(g), self.M, C, L.ptr(self.ones), L.ptr(self.beta),
                                         L.ptr(self.save), self.save[C:].data_ptr(), self.sum_rows, L.ptr(self.sum_part),
                                         L.ptr(g), L.ptr(self.gscr), L.ptr(self.gbeta), L.ptr(e.small[:3 * C]), _stream())
        else:       # BatchNorm (+ReLU) backward in place on the accumulated gradient; gamma is the constant 1
            L.spnet_bn_bwd(L.ptr(self.yp), L.ptr(g), self.M, C, L.ptr(self.ones), L.ptr(self.beta), L.ptr(self.save),
                           self.save[C:].data_ptr(), ACT_RELU if self.relu else ACT_NONE, L.ptr(g), L.ptr(self.gscr),
                           L.ptr(self.gbeta), L.ptr(e.small[:3 * C]), e.ws_ptr(WS_MISC), _stream())
        side = e.wgrad_stream
        if self.small:
            def wgrad(region):
                L.spnet_conv3x3_small(2, 3, C, self.stride, 0, L.ptr(self.src.buf), L.ptr(g), L.ptr(self.gw), e.B, self.H,
                                      self.W, e.ws_ptr(region), region[1], _stream())
        else:
            def wgrad(region):
                self.gather_patches()
                _gemm(self._A(), OUT_MAJOR, self.K, g, OUT_MAJOR, C, self.gw, C, self.K, C, self.M, e, region=region)
        if self.deferred_wgrad:
            pass                                     # IRv2Backbone.flush_wgrads: g (= dy now) stays as it is until then
        elif side is None:
            wgrad(WS_GEMM)
        else:           # weight gradient off the data-gradient chain (see Pointwise.bwd)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                wgrad(WS_GEMM2)
        if self.small:
            L.spnet_conv3x3_small(1, 3, C, self.stride, 0, L.ptr(g), L.ptr(self.w), L.ptr(self.dx), e.B, self.H, self.W,
                                  e.ws_ptr(WS_MISC), WS_MISC[1], _stream())
        elif self.direct and self.src.g is not None:
            # another consumer of my input has already left its gradient in the accumulator: add mine in the GEMM's
            # epilogue (C += dY W^T) instead of a GEMM into dx + an accumulation pass -- same sum, same rounding
            prof = e.prof
            t0 = prof.start() if prof is not None else None
            L.spnet_gemm_f32_accumulate(L.ptr(g), K_MAJOR, C, L.ptr(self.w), K_MAJOR, C, L.ptr(self.src.g), self.K, self.M,
                                        self.K, C, _tile_for(K_MAJOR, K_MAJOR, 0, self.M, self.K, C, 0), _stream())
            if prof is not None:
                prof.stop("gemm", t0, 2.0 * self.M * self.K * C, ("aA+=", self.M, self.K, C))
            return
        elif self.direct:
            _gemm(g, K_MAJOR, C, self.w, K_MAJOR, C, self.dx, self.K, self.M, self.K, C, e)
        else:
            _gemm(g, K_MAJOR, C, self.w, K_MAJOR, C, net.dcol, self.K, self.M, self.K, C, e)
            p = self.src.owner
            if isinstance(p, _IRConv) and p.has_sums():          # + ReLU mask and BatchNorm sums of the producing layer
                yp, ldyp, mean, invstd, part, ldp, rows, gdst, ldg = p.sum_views()
                L.spnet_patches_bwd_bnsums_ld(L.ptr(net.dcol), gdst if gdst else L.ptr(self.dx), ldg, e.B, self.H, self.W,
                                              self.cin, self.kh, self.kw, self.stride, self.same, L.ptr(self.src.buf), p.ldy,
                                              yp, ldyp, mean, invstd, int(p.relu), part, ldp, rows, _stream())
                if gdst:
                    return                       # written into the sibling group's gradient block: nothing to accumulate
            else:
                L.spnet_patches(L.ptr(net.dcol), L.ptr(self.dx), e.B, self.H, self.W, self.cin, self.kh, self.kw,
                                self.stride, self.same, 1, _stream())
        _ir_acc(self.src, self.dx, e)


class _IRPool:
    def __init__(self, eng, src, kind):
        self.e, self.src, self.kind = eng, src, kind
        B, H, W, C = src.buf.shape
        self.H, self.W, self.C = H, W, C
        if kind == "maxpool":
            OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
            self.idx = torch.empty(B * OH * OW * (C // 4), device=eng.dev, dtype=torch.int32) if eng.train_capable else None
        else:
            OH, OW = H, W
        self.out = _T(eng.new(B, OH, OW, C))
        self.dx = eng.new(B, H, W, C) if eng.train_capable else None

    def srcs(self):
        return [self.src]

    def fwd(self, training):
        e = self.e
        if self.kind == "maxpool":
            L.spnet_maxpool3x3s2_valid_fwd(L.ptr(self.src.buf), L.ptr(self.out.buf), L.ptr(self.idx) if training else None,
                                           e.B, self.H, self.W, self.C, _stream())
        else:
            L.spnet_avgpool3x3s1_same(L.ptr(self.src.buf), L.ptr(self.out.buf), e.B, self.H, self.W, self.C, 0, _stream())

    def bwd(self, net):
        e, g = self.e, self.out.g
        if self.kind == "maxpool":
            L.spnet_maxpool3x3s2_valid_bwd(L.ptr(g), L.ptr(self.idx), L.ptr(self.dx), e.B, self.H, self.W, self.C, _stream())
        else:
            L.spnet_avgpool3x3s1_same(L.ptr(g), L.ptr(self.dx), e.B, self.H, self.W, self.C, 1, _stream())
        _ir_acc(self.src, self.dx, e)


class _IRConcat:
    def __init__(self, eng, srcs):
        self.e, self.srcs_ = eng, srcs
        B, H, W, _ = srcs[0].buf.shape
        self.cs = [t.buf.shape[-1] for t in srcs]
        self.rows = B * H * W
        self.out = _T(eng.new(B, H, W, sum(self.cs)))
        self.dparts = [eng.new(*t.buf.shape) for t in srcs] if eng.train_capable else None
        self.direct = [False] * len(srcs)            # IRv2Backbone: this branch writes into my buffer itself

    def srcs(self):
        return self.srcs_

    def fwd(self, training):
        ct, off = sum(self.cs), 0
        for t, c, direct in zip(self.srcs_, self.cs, self.direct):
            if not direct:            # (t may be a column block itself: a member of a sibling group)
                L.spnet_copy_cols(L.ptr(t.buf), t.buf.stride(2), self.out.buf.data_ptr() + 4 * off, ct, self.rows, c, 0,
                                  _stream())
            off += c

    def bwd(self, net):
        ct, off, g = sum(self.cs), 0, self.out.g
        for t, c, d in zip(self.srcs_, self.cs, self.dparts):
            p = t.owner
            if isinstance(p, _IRConv) and p.has_sums():          # + ReLU mask and BatchNorm sums of the branch's last layer
                yp, ldyp, mean, invstd, part, ldp, rows, gdst, ldg = p.sum_views()
                L.spnet_copy_cols_bnsums_ld(g.data_ptr() + 4 * off, ct, gdst if gdst else L.ptr(d), ldg, self.rows, c,
                                            L.ptr(t.buf), p.ldy, yp, ldyp, mean, invstd, int(p.relu), part, ldp, rows,
                                            _stream())
                if gdst:                         # written into the sibling group's gradient block
                    off += c
                    continue
            else:
                L.spnet_copy_cols(g.data_ptr() + 4 * off, ct, L.ptr(d), c, self.rows, c, 0, _stream())
            _ir_acc(t, d, self.e)
            off += c


class _IRResAdd:
    def __init__(self, eng, x, up, scale, relu):
        self.e, self.x, self.up, self.scale, self.relu = eng, x, up, float(scale), int(bool(relu))
        self.out = _T(eng.new(*x.buf.shape))
        if eng.train_capable:
            self.dx, self.dup = eng.new(*x.buf.shape), eng.new(*x.buf.shape)

    def srcs(self):
        return [self.x, self.up]

    def fwd(self, training):
        L.spnet_resadd(L.ptr(self.x.buf), L.ptr(self.up.buf), L.ptr(self.out.buf), self.out.buf.numel(), self.scale,
                       self.relu, _stream())

    def bwd(self, net):
        L.spnet_resadd_bwd(L.ptr(self.out.buf), L.ptr(self.out.g), L.ptr(self.dx), L.ptr(self.dup), self.out.buf.numel(),
                           self.scale, self.relu, _stream())
        _ir_acc(self.x, self.dx, self.e)
        _ir_acc(self.up, self.dup, self.e)


class PadSame(Node):
    """TF 'same' zero padding of a k x k / stride s convolution made explicit (extra row / column at the bottom /
    right), so that the VALID direct-conv kernels serve MobileNet's conv1 (Conv2D(32, 3, strides 2, padding='same'))."""

    def __init__(self, eng, x, k, s):
        self.e, self.x = eng, x
        self.pnames = []
        B, H, W, C = x.shape
        oh, ow = (H + s - 1) // s, (W + s - 1) // s
        ph, pw = max((oh - 1) * s + k - H, 0), max((ow - 1) * s + k - W, 0)
        self.pt, self.pl, self.H, self.W = ph // 2, pw // 2, H, W
        self.y = torch.zeros(B, H + ph, W + pw, C, device=eng.dev, dtype=torch.float32)   # the border stays zero
        self.dx = eng.new(*x.shape) if eng.train_capable else None

    def _inner(self, t):
        return t[:, self.pt:self.pt + self.H, self.pl:self.pl + self.W, :]

    def fwd(self, training):
        self._inner(self.y).copy_(self.x)

    def bwd(self, g):
        self.dx.copy_(self._inner(g))
        return self.dx


class MobileBlock(Node):
    """One depthwise-separable block of keras MobileNet (_depthwise_conv_block): DepthwiseConv2D 3x3 (stride 1|2,
    'same') - BN - relu6 - Conv2D 1x1 - BN - relu6.  Stride-1 depthwise layers run the LDS-tiled kernels (fused data +
    weight gradient backward), stride-2 ones the strided gather kernels; both BatchNorms are materialised."""

    def __init__(self, eng, x, i, cin, cout, stride):
        self.e, self.x, self.cin, self.cout, self.stride = eng, x, cin, cout, stride
        self.pnames = ["conv_dw_%d" % i, "conv_dw_%d_bn" % i, "conv_pw_%d" % i, "conv_pw_%d_bn" % i]
        B, H, W, _ = x.shape
        self.H, self.W = H, W
        OH, OW = (H + stride - 1) // stride, (W + stride - 1) // stride
        self.M = B * OH * OW
        self.wd = eng.P("conv_dw_%d/depthwise_kernel" % i)
        self.z = eng.new(B, OH, OW, cin)
        self.bn_dw = BatchNorm(eng, self.z, cin, "conv_dw_%d_bn" % i, ACT_RELU6)
        self.pw = Pointwise(eng, self.M, cin, cout, "conv_pw_%d/kernel" % i, allow_blend=False)
        self.yp = eng.new(B, OH, OW, cout)
        self.bn = BN(eng, cout, self.M, "conv_pw_%d_bn" % i)
        self.y = eng.new(B, OH, OW, cout)
        if eng.train_capable:
            self.gwd = eng.G("conv_dw_%d/depthwise_kernel" % i)
            self.da = eng.new(B, OH, OW, cin)
            self.dx = eng.new(B, H, W, cin)
            if stride == 1:
                self.dwb = _DwBwdPlan(B, H, W, cin)
                self.wpart = eng.new(self.dwb.ws_floats)
                eng.dw_reduce_jobs.append((self.wpart, self.gwd, self.dwb.rows, 9 * cin))
            else:
                self.ws = eng.new(L.spnet_dwconv3x3_strided_ws(B, H, W, cin, stride))

    def fwd(self, training):
        e = self.e
        if self.stride == 1:
            _dw_fwd(self.x, self.wd, self.z, e.B, self.H, self.W, self.cin, 0, None, None)
        else:
            L.spnet_dwconv3x3_strided(0, L.ptr(self.x), L.ptr(self.wd), L.ptr(self.z), e.B, self.H, self.W, self.cin,
                                      self.stride, None, _stream())
        self.bn_dw.fwd(training)
        if training:
            self.bn.finalize(self.pw.fwd_colstats(self.bn_dw.y, self.yp))
        else:
            self.pw.fwd(self.bn_dw.y, self.yp)
            self.bn.infer()
        self.bn.apply(self.yp, self.y, ACT_RELU6)

    def bwd(self, g):
        e = self.e
        dy = self.bn.bwd_full(self.yp, g, g, ACT_RELU6)           # in place: g is not needed again
        self.pw.bwd(self.bn_dw.y, dy, self.da)
        dz = self.bn_dw.bwd(self.da)
        if self.stride == 1:
            self.dwb.run(dz, self.x, self.wd, self.dx, 0, None, self.wpart, None, None, None, None, None, None)
        else:
            L.spnet_dwconv3x3_strided(1, L.ptr(dz), L.ptr(self.wd), L.ptr(self.dx), e.B, self.H, self.W, self.cin,
                                      self.stride, None, _stream())
            L.spnet_dwconv3x3_strided(2, L.ptr(self.x), L.ptr(dz), L.ptr(self.gwd), e.B, self.H, self.W, self.cin,
                                      self.stride, L.ptr(self.ws), _stream())
        return self.dx


class Dense(Node):
    """Flatten (NHWC order) + Dense(n_out) with bias (spnet/models.py:378,388)."""

    def __init__(self, eng, x, n_out, name):
        self.e, self.x, self.n_out = eng, x, n_out
        self.pnames = [name]
        self.K = x.numel() // eng.B
        self.w, self.b = eng.P(name + "/kernel"), eng.P(name + "/bias")
        self.y = eng.new(eng.B, n_out)
        if eng.train_capable:
            self.gw, self.gb = eng.G(name + "/kernel"), eng.G(name + "/bias")
            self.dx = eng.new(*x.shape)

    def fwd(self, training):
        e = self.e
        _gemm(self.x, K_MAJOR, self.K, self.w, OUT_MAJOR, self.n_out, self.y, self.n_out, e.B, self.n_out, self.K,
              e, bias=self.b)

    def bwd(self, g):
        e = self.e
        side = e.wgrad_stream if e.early_head else None
        if side is None:
            _gemm(self.x, OUT_MAJOR, self.K, g, OUT_MAJOR, self.n_out, self.gw, self.n_out, self.K, self.n_out, e.B, e)
        else:       # the 226 MB weight gradient (an outer product over the batch: pure HBM writes) leaves the dependency chain
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _gemm(self.x, OUT_MAJOR, self.K, g, OUT_MAJOR, self.n_out, self.gw, self.n_out, self.K, self.n_out, e.B, e,
                      region=WS_GEMM2)
        L.spnet_reduce_rows(L.ptr(g), e.B, self.n_out, L.ptr(self.gb), _stream())
        _gemm(g, K_MAJOR, self.n_out, self.w, K_MAJOR, self.n_out, self.dx, self.K, e.B, self.K, self.n_out, e)
        return self.dx
