"""torch-CPU restatement of the network the reference builds through Keras (TEST INFRASTRUCTURE).

PARITY UNPINNED at the Keras/TensorFlow boundary (see oracle/__init__.py): this file restates
  * the stem                       spnet/models.py:315-342
  * keras.applications.Xception    call site spnet/models.py:357-359 (Keras 2.1.3, not vendored;
                                   layer list reproduced from the published architecture and checked
                                   against the reference's logged parameter counts and shapes)
  * Flatten + Dense('FinalOutput') spnet/models.py:378,388
  * custom_loss                    spnet/models.py:564-589
  * add_regularization l2(1e-4)    spnet/models.py:47-71 (10 kernels, run-log line 98)
  * Adam                           spnet/models.py:494 (Keras form, eps outside bias correction)
with TensorFlow semantics: NHWC tensors, SAME padding (extra pad bottom/right), BatchNorm eps 1e-3 /
momentum 0.99 with batch statistics in training, NHWC flatten order.

All tensors at this module's boundary are NHWC; weights are kept in Keras layouts
(conv HWIO, depthwise [3,3,C], pointwise [Cin,Cout], dense [in,out]).
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99
L2 = 1e-4
LAMBDAS = dict(center=2.0, size=1.0, angle=3.0, noobj=0.3, cls=5.0)


# ------------------------------------------------------------------ primitive ops (NHWC in / out)
def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, w_hwio, stride=1, padding="valid"):
    """Dense conv, TF semantics.  'same' pads total=max((out-1)*s+k-in,0), floor(total/2) before."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    xc = _nchw(x)
    if padding == "same":
        H, W = x.shape[1], x.shape[2]
        oh, ow = -(-H // stride), -(-W // stride)
        ph = max((oh - 1) * stride + kh - H, 0)
        pw = max((ow - 1) * stride + kw - W, 0)
        xc = F.pad(xc, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return _nhwc(F.conv2d(xc, w_hwio.permute(3, 2, 0, 1), stride=stride))


def dwconv3x3(x, w_33c, stride=1):
    """Depthwise 3x3, SAME, no bias (tf.nn.depthwise_conv2d inside SeparableConv2D / MobileNet's DepthwiseConv2D);
    stride 2 pads the TF way (extra row / column at the bottom / right).  Written as nine shifted multiply-adds on the
    NHWC tensor (y[b,i,j,c] = sum_{kh,kw} xpad[b, i*s+kh, j*s+kw, c] * w[kh,kw,c]): identical to a grouped conv2d, but
    it stays fast in float64, where grouped convolutions have no optimised CPU path."""
    H, W = x.shape[1], x.shape[2]
    oh, ow = -(-H // stride), -(-W // stride)
    ph = max((oh - 1) * stride + 3 - H, 0)
    pw = max((ow - 1) * stride + 3 - W, 0)
    xp = F.pad(x, (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    y = None
    for kh in range(3):
        for kw in range(3):
            t = xp[:, kh:kh + (oh - 1) * stride + 1:stride, kw:kw + (ow - 1) * stride + 1:stride, :] * w_33c[kh, kw]
            y = t if y is None else y + t
    return y


def pwconv(x, w_io):
    """Pointwise 1x1 conv == GEMM over the flattened pixels."""
    return x @ w_io


def maxpool3x3s2_same(x):
    H, W = x.shape[1], x.shape[2]
    oh, ow = -(-H // 2), -(-W // 2)
    ph = max((oh - 1) * 2 + 3 - H, 0)
    pw = max((ow - 1) * 2 + 3 - W, 0)
    xc = F.pad(_nchw(x), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    return _nhwc(F.max_pool2d(xc, 3, 2))


class Decisions:
    """The DISCRETE decisions of one forward pass, in application order: sign masks of every LeakyReLU / ReLU input
    and the arg-max tap (kh*3+kw) of every max-pool window.

    A network with ReLUs and max-pools is piecewise linear in its activations: two correct fp32 evaluations that
    round an input sitting within ~1e-6 of zero (or two window entries within ~1e-6 of each other) to different sides
    take different linear pieces, and the GRADIENTS then differ by whole terms, not by rounding (measured: the fp32
    CPU evaluation of this very oracle deviates from its fp64 evaluation by up to 20 % in single tensors at 384x512).
    Decisions() records the pieces a forward pass took; Decisions(relu, pool) makes forward() take the given ones,
    and notes for each site how far from a tie the overridden elements were (`flips`).  Parity tests evaluate the
    fp64 oracle on the DEVICE's pieces: everything continuous must then agree to rounding, and every overridden
    decision must have been a tie at fp32 resolution."""

    def __init__(self, relu=None, pool=None):
        self.force = relu is not None
        self.relu = list(relu) if relu is not None else []
        self.pool = list(pool) if pool is not None else []
        self._ri = self._pi = 0
        self.n_decisions = 0     # elements / windows decided so far in force mode (the denominator of the override rate)
        self.flips = []          # (site, n_overridden, largest |x| (or window gap) among them / largest |x| of the tensor)

    def act(self, x, slope):
        if not self.force:
            self.relu.append((x > 0).detach())
            return F.leaky_relu(x, slope) if slope else F.relu(x)
        m = self.relu[self._ri].to(x.device)
        self._ri += 1
        self.n_decisions += m.numel()
        diff = m != (x > 0)
        n = int(diff.sum())
        if n:
            self.flips.append(("relu%d" % (self._ri - 1), n, float(x.detach()[diff].abs().max() / x.detach().abs().max())))
        return torch.where(m, x, slope * x)

    def act6(self, x):
        """ReLU6 (two kinks): the recorded decision is a byte per element, 0 = clipped to 0, 1 = linear, 2 = clipped to 6."""
        own = (x > 0).to(torch.uint8) + (x >= 6).to(torch.uint8)
        if not self.force:
            self.relu.append(own.detach())
            return torch.clamp(x, 0.0, 6.0)
        m = self.relu[self._ri].to(x.device)
        self._ri += 1
        self.n_decisions += m.numel()
        diff = m != own
        n = int(diff.sum())
        if n:
            xd = x.detach()[diff]
            self.flips.append(("relu6_%d" % (self._ri - 1), n, float(torch.minimum(xd.abs(), (xd - 6).abs()).max() / x.detach().abs().max())))
        return torch.where(m == 1, x, torch.where(m == 2, torch.full_like(x, 6.0), torch.zeros_like(x)))

    def pool_taps(self, cols):
        """cols [B,C,9,OH,OW] window entries (-inf padding) -> taps [B,C,1,OH,OW] to gather."""
        own = cols.detach().argmax(2, keepdim=True)
        if not self.force:
            self.pool.append(own)
            return own
        t = self.pool[self._pi].to(cols.device)
        self._pi += 1
        self.n_decisions += t.numel()
        diff = t != own
        n = int(diff.sum())
        if n:
            gap = (cols.detach().gather(2, own) - cols.detach().gather(2, t))[diff]
            self.flips.append(("pool%d" % (self._pi - 1), n, float(gap.max() / cols.detach()[torch.isfinite(cols.detach())].abs().max())))
        return t


def _act(x, slope, decisions):
    if decisions is None:
        return F.leaky_relu(x, slope) if slope else F.relu(x)
    return decisions.act(x, slope)


def _act6(x, decisions):
    return torch.clamp(x, 0.0, 6.0) if decisions is None else decisions.act6(x)


def maxpool3x3s2_same_decided(x, decisions):
    """maxpool3x3s2_same with the window arg-max recorded in / taken from `decisions`."""
    B, H, W, C = x.shape
    oh, ow = -(-H // 2), -(-W // 2)
    ph = max((oh - 1) * 2 + 3 - H, 0)
    pw = max((ow - 1) * 2 + 3 - W, 0)
    xc = F.pad(_nchw(x), (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2), value=float("-inf"))
    cols = F.unfold(xc, 3, stride=2).reshape(B, C, 9, oh, ow)
    return _nhwc(cols.gather(2, decisions.pool_taps(cols)).squeeze(2))


def avgpool2(x):
    return _nhwc(F.avg_pool2d(_nchw(x), 2))


def batchnorm(x, gamma, beta, mean, var, training, update=True):
    """Keras BatchNormalization(axis=-1): training -> batch mean / biased variance for the
    normalisation; moving stats <- 0.99*moving + 0.01*batch (variance Bessel-corrected, as the TF
    fused op reports it).  ``mean``/``var`` are updated in place when training and update."""
    C = x.shape[-1]
    if training:
        flat = x.reshape(-1, C)
        mu = flat.mean(0)
        v = flat.var(0, unbiased=False)
        if update:
            n = flat.shape[0]
            with torch.no_grad():
                mean.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * mu.detach())
                var.mul_(BN_MOMENTUM).add_((1 - BN_MOMENTUM) * v.detach() * (n / max(n - 1, 1)))
        return (x - mu) * torch.rsqrt(v + BN_EPS) * gamma + beta
    return (x - mean) * torch.rsqrt(var + BN_EPS) * gamma + beta


# ------------------------------------------------------------------ architecture description
def xception_layers():
    """Ordered description of keras.applications.Xception(include_top=False) as (kind, name, ...)."""
    L = [("conv", "block1_conv1", 3, 32, 2, "valid"), ("bn", "block1_conv1_bn", 32), ("relu",),
         ("conv", "block1_conv2", 32, 64, 1, "valid"), ("bn", "block1_conv2_bn", 64), ("relu",)]
    res_id = [4]      # residual convs are auto-named conv2d_4.. after the stem's conv2d_1..3

    def strided_block(b, cin, c1, c2, first_relu):
        r = res_id[0]
        res_id[0] += 1
        blk = [("res_conv", "conv2d_%d" % r, "batch_normalization_%d" % r, cin, c2)]
        if first_relu:
            blk.append(("relu",))
        blk += [("sep", "block%d_sepconv1" % b, cin, c1), ("bn", "block%d_sepconv1_bn" % b, c1), ("relu",),
                ("sep", "block%d_sepconv2" % b, c1, c2), ("bn", "block%d_sepconv2_bn" % b, c2),
                ("pool",), ("add_res",)]
        return blk

    L += strided_block(2, 64, 128, 128, False)
    L += strided_block(3, 128, 256, 256, True)
    L += strided_block(4, 256, 728, 728, True)
    for b in range(5, 13):
        L.append(("res_id",))
        for k in (1, 2, 3):
            L += [("relu",), ("sep", "block%d_sepconv%d" % (b, k), 728, 728), ("bn", "block%d_sepconv%d_bn" % (b, k), 728)]
        L.append(("add_res",))
    L += strided_block(13, 728, 728, 1024, True)
    L += [("sep", "block14_sepconv1", 1024, 1536), ("bn", "block14_sepconv1_bn", 1536), ("relu",),
          ("sep", "block14_sepconv2", 1536, 2048), ("bn", "block14_sepconv2_bn", 2048), ("relu",)]
    return L


MOBILENET_BLOCKS = [(64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1), (512, 1), (512, 1), (512, 1),
                    (512, 1), (1024, 2), (1024, 1)]       # (pointwise filters, depthwise stride) of conv_dw/pw_1..13


def mobilenet_out_hw(H, W):
    h, w = H // 2, W // 2                      # stem avgpool
    h, w = -(-h // 2), -(-w // 2)              # conv1 3x3/s2 SAME
    for _, s_ in MOBILENET_BLOCKS:
        if s_ == 2:
            h, w = -(-h // 2), -(-w // 2)
    return h, w


# ------------------------------------------------------------------ Inception-ResNet-v2 (keras.applications 2.1.3)
def irv2_layers():
    """keras.applications.inception_resnet_v2.InceptionResNetV2(include_top=False) as a flat program.

    Ops (executed in order; `src` / `dst` name tensors in a small register file):
      ("conv", name, bn_name|None, src, dst, cin, cout, (kh, kw), stride, padding, relu, bias)
      ("maxpool", src, dst)                       MaxPooling2D(3, strides=2, 'valid')
      ("avgpool", src, dst)                       AveragePooling2D(3, strides=1, 'same')
      ("concat", [srcs], dst, name)
      ("resadd", x, up, dst, scale, relu)         x + scale * up, optional ReLU (the Lambda + Activation of a block)
    conv2d_bn layers are auto-named by Keras in creation order: conv2d_4.. / batch_normalization_4.. behind the stem's
    three; BatchNormalization(scale=False): no gamma."""
    ops = []
    ctr = [3]

    def conv(src, dst, cin, cout, k, stride=1, padding="same", relu=True, name=None, bias=False):
        kk = (k, k) if isinstance(k, int) else tuple(k)
        if name is None:
            ctr[0] += 1
            cname, bname = "conv2d_%d" % ctr[0], "batch_normalization_%d" % ctr[0]
        else:
            cname, bname = name, name + "_bn"
        ops.append(("conv", cname, None if bias else bname, src, dst, cin, cout, kk, stride, padding, relu, bias))
        return cout

    c = conv("in", "x", 3, 32, 3, 2, "valid")
    c = conv("x", "x", c, 32, 3, 1, "valid")
    c = conv("x", "x", c, 64, 3)
    ops.append(("maxpool", "x", "x"))
    c = conv("x", "x", c, 80, 1, 1, "valid")
    c = conv("x", "x", c, 192, 3, 1, "valid")
    ops.append(("maxpool", "x", "x"))
    # mixed_5b
    conv("x", "b0", 192, 96, 1)
    conv("x", "b1", 192, 48, 1); conv("b1", "b1", 48, 64, 5)
    conv("x", "b2", 192, 64, 1); conv("b2", "b2", 64, 96, 3); conv("b2", "b2", 96, 96, 3)
    ops.append(("avgpool", "x", "bp")); conv("bp", "bp", 192, 64, 1)
    ops.append(("concat", ["b0", "b1", "b2", "bp"], "x", "mixed_5b"))
    c = 320

    def block(kind, idx, c, scale, relu=True):
        if kind == "block35":
            conv("x", "b0", c, 32, 1)
            conv("x", "b1", c, 32, 1); conv("b1", "b1", 32, 32, 3)
            conv("x", "b2", c, 32, 1); conv("b2", "b2", 32, 48, 3); conv("b2", "b2", 48, 64, 3)
            br, cm = ["b0", "b1", "b2"], 128
        elif kind == "block17":
            conv("x", "b0", c, 192, 1)
            conv("x", "b1", c, 128, 1); conv("b1", "b1", 128, 160, (1, 7)); conv("b1", "b1", 160, 192, (7, 1))
            br, cm = ["b0", "b1"], 384
        else:
            conv("x", "b0", c, 192, 1)
            conv("x", "b1", c, 192, 1); conv("b1", "b1", 192, 224, (1, 3)); conv("b1", "b1", 224, 256, (3, 1))
            br, cm = ["b0", "b1"], 448
        name = "%s_%d" % (kind, idx)
        ops.append(("concat", br, "m", name + "_mixed"))
        conv("m", "up", cm, c, 1, relu=False, name=name + "_conv", bias=True)
        ops.append(("resadd", "x", "up", "x", scale, relu))

    for i in range(1, 11):
        block("block35", i, 320, 0.17)
    # mixed_6a
    conv("x", "b0", 320, 384, 3, 2, "valid")
    conv("x", "b1", 320, 256, 1); conv("b1", "b1", 256, 256, 3); conv("b1", "b1", 256, 384, 3, 2, "valid")
    ops.append(("maxpool", "x", "bp"))
    ops.append(("concat", ["b0", "b1", "bp"], "x", "mixed_6a"))
    for i in range(1, 21):
        block("block17", i, 1088, 0.1)
    # mixed_7a
    conv("x", "b0", 1088, 256, 1); conv("b0", "b0", 256, 384, 3, 2, "valid")
    conv("x", "b1", 1088, 256, 1); conv("b1", "b1", 256, 288, 3, 2, "valid")
    conv("x", "b2", 1088, 256, 1); conv("b2", "b2", 256, 288, 3); conv("b2", "b2", 288, 320, 3, 2, "valid")
    ops.append(("maxpool", "x", "bp"))
    ops.append(("concat", ["b0", "b1", "b2", "bp"], "x", "mixed_7a"))
    for i in range(1, 10):
        block("block8", i, 2080, 0.2)
    block("block8", 10, 2080, 1.0, relu=False)
    conv("x", "x", 2080, 1536, 1, name="conv_7b")
    return ops


def irv2_out_hw(H, W):
    h, w = H // 2, W // 2                      # SPNet stem avgpool
    v = lambda n, k, s: (n - k) // s + 1
    h, w = v(h, 3, 2), v(w, 3, 2)
    h, w = v(h, 3, 1), v(w, 3, 1)
    h, w = v(h, 3, 2), v(w, 3, 2)
    h, w = v(h, 3, 1), v(w, 3, 1)
    h, w = v(h, 3, 2), v(w, 3, 2)              # 35x35 stage for a 299x299 network input
    h, w = v(h, 3, 2), v(w, 3, 2)              # mixed_6a
    h, w = v(h, 3, 2), v(w, 3, 2)              # mixed_7a
    return h, w


def maxpool3x3s2_valid(x, decisions=None):
    B, H, W, C = x.shape
    if decisions is None:
        return _nhwc(F.max_pool2d(_nchw(x), 3, 2))
    oh, ow = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    cols = F.unfold(_nchw(x), 3, stride=2).reshape(B, C, 9, oh, ow)
    return _nhwc(cols.gather(2, decisions.pool_taps(cols)).squeeze(2))


def avgpool3x3s1_same(x):
    """AveragePooling2D(3, strides=1, padding='same'): TF averages over the VALID window entries only."""
    return _nhwc(F.avg_pool2d(_nchw(x), 3, 1, 1, count_include_pad=False))


def conv2d_general(x, w_hwio, stride, padding):
    """Conv2D with a rectangular kernel, TF 'same' / 'valid' (conv2d above is the square-kernel case)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    xc = _nchw(x)
    if padding == "same":
        H, W = x.shape[1], x.shape[2]
        oh, ow = -(-H // stride), -(-W // stride)
        ph = max((oh - 1) * stride + kh - H, 0)
        pw = max((ow - 1) * stride + kw - W, 0)
        xc = F.pad(xc, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    return _nhwc(F.conv2d(xc, w_hwio.permute(3, 2, 0, 1), stride=stride))


def backbone_irv2(P, x, training, taps=None, decisions=None):
    reg = {"in": x}
    for op in irv2_layers():
        kind = op[0]
        if kind == "conv":
            _, cname, bname, src, dst, cin, cout, kk, stride, padding, relu, bias = op
            t = conv2d_general(reg[src], P[cname + "/kernel"], stride, padding)
            if bias:
                t = t + P[cname + "/bias"]
            else:
                t = batchnorm(t, 1.0, P[bname + "/beta"], P[bname + "/moving_mean"], P[bname + "/moving_variance"], training)
            if relu:
                t = _act(t, 0.0, decisions)
            reg[dst] = t
        elif kind == "maxpool":
            reg[op[2]] = maxpool3x3s2_valid(reg[op[1]], decisions)
        elif kind == "avgpool":
            reg[op[2]] = avgpool3x3s1_same(reg[op[1]])
        elif kind == "concat":
            reg[op[2]] = torch.cat([reg[s_] for s_ in op[1]], dim=-1)
            if taps is not None:
                taps[op[3]] = reg[op[2]]
        elif kind == "resadd":
            _, xs, up, dst, scale, relu = op
            t = reg[xs] + scale * reg[up]
            reg[dst] = _act(t, 0.0, decisions) if relu else t
    if taps is not None:
        taps["backbone"] = reg["x"]
    return reg["x"]


def backbone_out_hw(H, W):
    h, w = H // 2, W // 2                      # stem avgpool
    h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1  # block1_conv1 3x3/s2 valid
    h, w = h - 2, w - 2                        # block1_conv2 3x3 valid
    for _ in range(4):                         # blocks 2,3,4,13: /2 SAME
        h, w = -(-h // 2), -(-w // 2)
    return h, w


def _glorot(shape, fan_in, fan_out, gen, dtype):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


def init_params(H, W, n_out=576, seed=0, dtype=torch.float32, basemodel="Xception"):
    """Keras default initialisers: glorot_uniform kernels, zero bias, BN gamma 1 / beta 0 /
    moving_mean 0 / moving_variance 1.  Returns OrderedDict name -> tensor (Keras weight names).
    basemodel 'MobileNet': keras.applications.mobilenet.MobileNet(alpha=1, include_top=False) (spnet/models.py:346-355;
    random initialisation -- the reference's imagenet download is not available)."""
    g = torch.Generator().manual_seed(seed)
    P = OrderedDict()

    def bn(name, c):
        P[name + "/gamma"] = torch.ones(c, dtype=dtype)
        P[name + "/beta"] = torch.zeros(c, dtype=dtype)
        P[name + "/moving_mean"] = torch.zeros(c, dtype=dtype)
        P[name + "/moving_variance"] = torch.ones(c, dtype=dtype)

    def conv(name, k, cin, cout):
        P[name + "/kernel"] = _glorot((k, k, cin, cout), cin * k * k, cout * k * k, g, dtype)

    conv("conv2d_1", 3, 1, 3)
    bn("batch_normalization_1", 3)
    conv("conv2d_2", 3, 3, 3)
    bn("batch_normalization_2", 3)
    conv("conv2d_3", 3, 3, 3)
    bn("batch_normalization_3", 3)
    if basemodel == "InceptionResNetV2":
        for op in irv2_layers():
            if op[0] != "conv":
                continue
            _, cname, bname, src, dst, cin, cout, kk, stride, padding, relu, bias = op
            P[cname + "/kernel"] = _glorot((kk[0], kk[1], cin, cout), cin * kk[0] * kk[1], cout * kk[0] * kk[1], g, dtype)
            if bias:
                P[cname + "/bias"] = torch.zeros(cout, dtype=dtype)
            else:                                  # BatchNormalization(scale=False): beta + moving statistics only
                P[bname + "/beta"] = torch.zeros(cout, dtype=dtype)
                P[bname + "/moving_mean"] = torch.zeros(cout, dtype=dtype)
                P[bname + "/moving_variance"] = torch.ones(cout, dtype=dtype)
        h, w = irv2_out_hw(H, W)
        nin = h * w * 1536
        P["FinalOutput/kernel"] = _glorot((nin, n_out), nin, n_out, g, dtype)
        P["FinalOutput/bias"] = torch.zeros(n_out, dtype=dtype)
        return P
    if basemodel == "MobileNet":
        conv("conv1", 3, 3, 32)
        bn("conv1_bn", 32)
        cin = 32
        for i, (cout, _) in enumerate(MOBILENET_BLOCKS, 1):
            P["conv_dw_%d/depthwise_kernel" % i] = _glorot((3, 3, cin), cin * 9, 9, g, dtype)
            bn("conv_dw_%d_bn" % i, cin)
            P["conv_pw_%d/kernel" % i] = _glorot((1, 1, cin, cout), cin, cout, g, dtype)
            bn("conv_pw_%d_bn" % i, cout)
            cin = cout
        h, w = mobilenet_out_hw(H, W)
        nin = h * w * cin
        P["FinalOutput/kernel"] = _glorot((nin, n_out), nin, n_out, g, dtype)
        P["FinalOutput/bias"] = torch.zeros(n_out, dtype=dtype)
        return P
    for lay in xception_layers():
        kind = lay[0]
        if kind == "conv":
            conv(lay[1], 3, lay[2], lay[3])
        elif kind == "bn":
            bn(lay[1], lay[2])
        elif kind == "res_conv":
            conv(lay[1], 1, lay[3], lay[4])
            bn(lay[2], lay[4])
        elif kind == "sep":
            _, name, cin, cout = lay
            P[name + "/depthwise_kernel"] = _glorot((3, 3, cin), cin * 9, 9, g, dtype)
            P[name + "/pointwise_kernel"] = _glorot((cin, cout), cin, cout, g, dtype)
    h, w = backbone_out_hw(H, W)
    nin = h * w * 2048
    P["FinalOutput/kernel"] = _glorot((nin, n_out), nin, n_out, g, dtype)
    P["FinalOutput/bias"] = torch.zeros(n_out, dtype=dtype)
    return P


L2_KERNELS = ("conv2d_1", "conv2d_2", "conv2d_3", "block1_conv1", "block1_conv2",
              "conv2d_4", "conv2d_5", "conv2d_6", "conv2d_7", "FinalOutput")
# MobileNet: every layer that still carries a kernel_regularizer after add_regularization's JSON round trip
# (Conv2D / Dense; DepthwiseConv2D serialises a depthwise_regularizer instead and loses it, like SeparableConv2D)
L2_KERNELS_MOBILENET = ("conv2d_1", "conv2d_2", "conv2d_3", "conv1") + tuple("conv_pw_%d" % i for i in range(1, 14)) + \
    ("FinalOutput",)


def is_trainable(name):
    return not (name.endswith("/moving_mean") or name.endswith("/moving_variance"))


def count_params(P):
    tr = sum(v.numel() for k, v in P.items() if is_trainable(k))
    nt = sum(v.numel() for k, v in P.items() if not is_trainable(k))
    return tr + nt, tr, nt


# ------------------------------------------------------------------ forward
def stem(P, x, training, drop_mask=None, taps=None, decisions=None):
    """spnet/models.py:321-340.  x [B,H,W,1] -> [B,H/2,W/2,3]."""
    def bn(name, t):
        return batchnorm(t, P[name + "/gamma"], P[name + "/beta"], P[name + "/moving_mean"],
                         P[name + "/moving_variance"], training)
    t = conv2d(x, P["conv2d_1/kernel"], 1, "same")
    t = avgpool2(t)
    t = _act(bn("batch_normalization_1", t), 0.1, decisions)
    t = conv2d(t, P["conv2d_2/kernel"], 1, "same")
    t = _act(bn("batch_normalization_2", t), 0.1, decisions)
    t = conv2d(t, P["conv2d_3/kernel"], 1, "same")
    t = bn("batch_normalization_3", t)
    t = t + avgpool2(x)                      # [.,.,.,1] broadcasts over the 3 channels
    if training and drop_mask is not None:   # Dropout(0.1): mask holds 0 or 1/0.9
        t = t * drop_mask
    if taps is not None:
        taps["stem"] = t
    return t


def backbone(P, x, training, taps=None, decisions=None):
    res = None
    for lay in xception_layers():
        kind = lay[0]
        if kind == "conv":
            x = conv2d(x, P[lay[1] + "/kernel"], lay[4], lay[5])
        elif kind == "bn":
            n = lay[1]
            x = batchnorm(x, P[n + "/gamma"], P[n + "/beta"], P[n + "/moving_mean"], P[n + "/moving_variance"], training)
            if taps is not None:
                taps[n] = x
        elif kind == "relu":
            x = _act(x, 0.0, decisions)
        elif kind == "res_conv":
            _, cn, bnn, cin, cout = lay
            r = conv2d(x, P[cn + "/kernel"], 2, "same")
            res = batchnorm(r, P[bnn + "/gamma"], P[bnn + "/beta"], P[bnn + "/moving_mean"], P[bnn + "/moving_variance"], training)
        elif kind == "res_id":
            res = x
        elif kind == "sep":
            n = lay[1]
            x = pwconv(dwconv3x3(x, P[n + "/depthwise_kernel"]), P[n + "/pointwise_kernel"])
        elif kind == "pool":
            x = maxpool3x3s2_same(x) if decisions is None else maxpool3x3s2_same_decided(x, decisions)
        elif kind == "add_res":
            x = x + res
    if taps is not None:
        taps["backbone"] = x
    return x


def backbone_mobilenet(P, x, training, taps=None, decisions=None):
    """keras.applications.mobilenet.MobileNet(alpha=1, depth_multiplier=1, include_top=False): conv1 3x3/s2 SAME + BN +
    relu6, then 13 x [depthwise 3x3 (stride 1|2, SAME) + BN + relu6 + pointwise 1x1 + BN + relu6]."""
    def bn(n, t):
        t = batchnorm(t, P[n + "/gamma"], P[n + "/beta"], P[n + "/moving_mean"], P[n + "/moving_variance"], training)
        if taps is not None:
            taps[n] = t
        return t
    x = _act6(bn("conv1_bn", conv2d(x, P["conv1/kernel"], 2, "same")), decisions)
    for i, (cout, s_) in enumerate(MOBILENET_BLOCKS, 1):
        x = _act6(bn("conv_dw_%d_bn" % i, dwconv3x3(x, P["conv_dw_%d/depthwise_kernel" % i], s_)), decisions)
        x = _act6(bn("conv_pw_%d_bn" % i, pwconv(x, P["conv_pw_%d/kernel" % i][0, 0])), decisions)
    if taps is not None:
        taps["backbone"] = x
    return x


def forward(P, X, training=False, drop_mask=None, taps=None, decisions=None, sigmoid_cols=None):
    """X [B,H,W,1] -> [B,n_out] in normalised units (linear Dense output).  decisions: see Decisions.
    sigmoid_cols=(start, step): the 'compound' head (spnet/models.py:379-386) -- Dense(n_preds, sigmoid) and
    Dense(rest) re-ordered by InterleaveColumns == one dense layer whose columns start::step pass through a sigmoid."""
    x = stem(P, X, training, drop_mask, taps, decisions)
    net = backbone_mobilenet if "conv1/kernel" in P else (backbone_irv2 if "conv_7b/kernel" in P else backbone)
    x = net(P, x, training, taps, decisions)
    flat = x.reshape(x.shape[0], -1)          # NHWC flatten order (Keras Flatten on channels_last)
    y = flat @ P["FinalOutput/kernel"] + P["FinalOutput/bias"]
    if sigmoid_cols is not None:
        cols = torch.zeros(y.shape[1], dtype=torch.bool)
        cols[sigmoid_cols[0]::sigmoid_cols[1]] = True
        y = torch.where(cols, torch.sigmoid(y), y)
    return y


# ------------------------------------------------------------------ loss / optimiser
def custom_loss(y_true, y_pred, loss_type="same"):
    """spnet/models.py:564-589 on torch tensors (differentiable)."""
    e = (y_true - y_pred) ** 2
    o = 1 - y_true[:, 6::8]
    if loss_type == "same":
        loss = LAMBDAS["noobj"] * e[:, 6::8].sum(-1)
    else:
        z = y_pred[:, 6::8]
        loss = LAMBDAS["noobj"] * (torch.clamp(z, min=0) - z * y_true[:, 6::8] + torch.log1p(torch.exp(-z.abs()))).sum(-1)
    loss = loss + LAMBDAS["center"] * ((o * e[:, 0::8]).sum(-1) + (o * e[:, 1::8]).sum(-1))
    loss = loss + LAMBDAS["size"] * ((o * e[:, 2::8]).sum(-1) + (o * e[:, 3::8]).sum(-1))
    d2 = (y_true[:, 2::8] - y_true[:, 3::8]) ** 2
    loss = loss + LAMBDAS["angle"] * ((o * e[:, 4::8] * d2).sum(-1) + (o * e[:, 5::8] * d2).sum(-1))
    loss = loss + LAMBDAS["cls"] * (o * e[:, 7::8]).sum(-1)
    return (loss / y_pred.shape[-1]).mean()


def l2_penalty(P):
    if "conv_7b/kernel" in P:      # Inception-ResNet-v2: every Conv2D / Dense kernel (all are plain Conv2D layers)
        return sum(L2 * (v ** 2).sum() for k, v in P.items() if k.endswith("/kernel"))
    names = L2_KERNELS_MOBILENET if "conv1/kernel" in P else L2_KERNELS
    return sum(L2 * (P[n + "/kernel"] ** 2).sum() for n in names)


class Trainer:
    """Keras-style training loop state: params, Adam moments, iteration counter."""

    def __init__(self, P, loss_type="same", eps=1e-7, sigmoid_cols=None):
        self.P = P
        self.loss_type = loss_type
        self.sigmoid_cols = sigmoid_cols
        self.eps = eps
        self.t = 0
        self.names = [k for k in P if is_trainable(k)]
        self.m = {k: torch.zeros_like(P[k]) for k in self.names}
        self.v = {k: torch.zeros_like(P[k]) for k in self.names}

    def grads(self, X, Y, drop_mask=None, include_l2=True, decisions=None):
        """Returns (data_loss, total_loss_with_l2, {name: grad}, y_pred) and updates BN moving stats.
        include_l2=False differentiates the data term only (the product folds the l2 gradient into
        its optimizer kernel)."""
        leaves = {k: self.P[k].detach().clone().requires_grad_(True) for k in self.names}
        Pg = OrderedDict((k, leaves.get(k, self.P[k])) for k in self.P)
        yp = forward(Pg, X, training=True, drop_mask=drop_mask, decisions=decisions, sigmoid_cols=self.sigmoid_cols)
        data = custom_loss(Y, yp, self.loss_type)
        total = data + l2_penalty(Pg)
        gs = torch.autograd.grad(total if include_l2 else data, [leaves[k] for k in self.names])
        return float(data.detach()), float(total.detach()), dict(zip(self.names, gs)), yp.detach()

    def step(self, X, Y, lr, drop_mask=None):
        data, total, gs, _ = self.grads(X, Y, drop_mask)
        self.t += 1
        b1, b2 = 0.9, 0.999
        lr_t = lr * math.sqrt(1.0 - b2 ** self.t) / (1.0 - b1 ** self.t)
        with torch.no_grad():
            for k in self.names:
                g = gs[k]
                # (1. - beta) evaluated in the parameters' precision, as the Keras/TF graph does (see numpy_ref.adam_step)
                one = torch.ones((), dtype=g.dtype)
                c1, c2 = one - torch.tensor(b1, dtype=g.dtype), one - torch.tensor(b2, dtype=g.dtype)
                self.m[k].mul_(b1).add_(c1 * g)
                self.v[k].mul_(b2).add_(c2 * g * g)
                self.P[k].sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))
        return data, total
