"""Shared helpers of the engine-vs-oracle parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from oracle import torch_ref as T

# Tolerances of the end-to-end parity tests, set from what is OBSERVED (round 4): every case logs its worst figure
# through observe() (SPNET_PARITY_LOG=<file>: one JSON line per figure; profiles/r04_parity_observed.txt is the GPU
# suite's log), and each bound below is about 10x the worst value any case of the suite produced.
# Observed over the 16 end-to-end cases of the GPU suite (three backbones, 9 geometries, batch 1-3 + 384x512):
#   worst per-tensor gradient error 1.03e-4 (MobileNet conv_pw_1_bn/gamma; Xception <= 9.4e-5, typical 1e-5)
#   worst training-output error 3.3e-5, worst inference MSE / scale 1.45e-12
GRAD_TOL = 1e-3          # max|dgrad - ref| / max|ref| per parameter tensor (fp64 oracle on the device's decisions)
GRAD_TOL_VANISHING = 5e-3  # tensors whose true gradient vanishes identically (max|ref| < 1e-6 of the model's largest
#                            gradient entry: a bias in front of a training-mode BatchNorm), measured against that floor:
#                            observed 1.9e-3 (Inception-ResNet's block8_10_conv/bias), i.e. 2e-9 of the largest entry
OUT_REL_TOL = 3.5e-4     # the same measure on the network output of a training forward
FWD_MSE_FACTOR = 2e-11   # inference forward: MSE <= factor * max(mean(ref^2), 1)   (north-star tolerance: 1e-4)


def observe(kind, value, detail=""):
    """Append one observed parity figure to $SPNET_PARITY_LOG (no-op when unset)."""
    path = os.environ.get("SPNET_PARITY_LOG")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps({"kind": kind, "value": float(value), "detail": detail,
                                "test": os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]}) + "\n")


def assert_forward_mse(got, want, factor=None):
    """Inference forward against the oracle: MSE <= factor * max(mean(want^2), 1)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    mse = float(((got - want) ** 2).mean())
    scale = max(float((want ** 2).mean()), 1.0)
    observe("fwd_mse_over_scale", mse / scale)
    f = FWD_MSE_FACTOR if factor is None else factor
    assert mse <= f * scale, (mse, scale)
    return mse


def assert_output_close(got, ref64, tol=None):
    """Network output of a training forward against the fp64 oracle's (rel_err measure)."""
    e = rel_err(got, ref64)
    observe("train_out_rel_err", e)
    assert e <= (OUT_REL_TOL if tol is None else tol), e
    return e


def dropout_mask(n, seed, rate=0.1):
    """numpy restatement of csrc/augment.hip dropout_kernel's counter hash (test-side oracle)."""
    i = np.arange(n, dtype=np.uint64)
    x = (i * np.uint64(0x9e3779b9) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7feb352d)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846ca68b)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    thresh = np.uint64(int(float(np.float32(rate)) * 4294967296.0))
    return np.where(x >= thresh, np.float32(1.0) / (np.float32(1.0) - np.float32(rate)), np.float32(0)).astype(np.float32)


def randomize_bn(P, gen):
    """Non-trivial BatchNorm parameters / moving statistics / biases so that every term is exercised."""
    for k in P:
        if k.endswith("/gamma"):
            P[k] = 0.5 + torch.rand(P[k].shape, generator=gen)
        elif k.endswith("/beta") or k.endswith("/moving_mean") or k.endswith("/bias"):
            P[k] = 0.2 * torch.randn(P[k].shape, generator=gen)
        elif k.endswith("/moving_variance"):
            P[k] = 0.5 + torch.rand(P[k].shape, generator=gen)
    return P


def make_case(H, W, B, seed, basemodel="Xception"):
    """Seeded weights (Keras initialisers + randomised BatchNorm state), frames, targets and the engine's
    dropout mask for one (geometry, seed).  Returns P, X, Y, mask, drop_seed."""
    P = randomize_bn(T.init_params(H, W, seed=1000 + seed, basemodel=basemodel), torch.Generator().manual_seed(seed))
    rs = np.random.RandomState(seed)
    X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32)
    Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    dseed = 777 + seed
    mask = torch.tensor(dropout_mask(B * (H // 2) * (W // 2) * 3, dseed).reshape(B, H // 2, W // 2, 3))
    return P, X, Y, mask, dseed


def rel_err(got, ref):
    """max|got - ref| / max|ref| -- the per-tensor measure every gradient comparison uses."""
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max()) / max(float(np.abs(ref).max()), 1e-12)


def oracle_grads(P, X, Y, mask, double, decisions=None, loss_type="same", sigmoid_cols=None):
    """(data_loss, {name: grad}, y_pred, params-with-updated-moving-stats) of the oracle in fp32 or fp64."""
    if double:
        tr = T.Trainer({k: v.double() for k, v in P.items()}, loss_type=loss_type, sigmoid_cols=sigmoid_cols)
        data, _, g, yp = tr.grads(X.double(), Y.double(), drop_mask=mask.double(), include_l2=False, decisions=decisions)
    else:
        tr = T.Trainer({k: v.clone() for k, v in P.items()}, loss_type=loss_type, sigmoid_cols=sigmoid_cols)
        data, _, g, yp = tr.grads(X, Y, drop_mask=mask, include_l2=False, decisions=decisions)
    return data, g, yp, tr.P


def device_decisions(eng):
    """The discrete decisions (T.Decisions) the engine's last TRAINING forward took, rebuilt on the host from the
    tensors it keeps for backward, in the oracle's application order:
      stem LeakyReLUs / block1 ReLUs   sign of the materialised activation (y > 0  <=>  pre-activation > 0)
      ReLU in front of a depthwise     the engine applies relu(fmaf(yp, scale, shift)) on load and masks the gradient
                                       with the same expression; yp*scale+shift in float64 has the sign of the exact
                                       value, which is the sign fmaf rounds
      max-pools                        the byte arg-max taps the pooling kernel saved for its backward pass."""
    from spnet_amd import engine as E

    def cpu(t):
        return t.detach().cpu()

    def lazy_mask(unit):                 # sign of BN(unit.yp) as the consumer computes it
        C = unit.cout
        ss = cpu(unit.bn.ss).double()
        return (cpu(unit.yp).double() * ss[:C] + ss[C:]) > 0

    def state6(y):                       # ReLU6: 0 = clipped to 0, 1 = linear, 2 = clipped to 6 (T.Decisions.act6)
        y = cpu(y)
        return (y > 0).to(torch.uint8) + (y >= 6).to(torch.uint8)

    relu, pool = [], []
    for node in eng.nodes:
        if isinstance(node, E.BatchNorm) and node.act == E.ACT_RELU6:
            relu.append(state6(node.y))
        elif isinstance(node, E.MobileBlock):
            relu += [state6(node.bn_dw.y), state6(node.y)]
        elif isinstance(node, E.IRv2Backbone):
            for o in node.ops:
                if getattr(o, "relu", False):                       # conv + BN + ReLU, or a block's closing ReLU
                    relu.append(cpu(o.out.buf) > 0)
                elif getattr(o, "kind", None) == "maxpool":
                    B, OH, OW, C = o.out.buf.shape
                    taps = cpu(o.idx).view(torch.uint8).reshape(B, OH, OW, C).long()
                    pool.append(taps.permute(0, 3, 1, 2).unsqueeze(2))
        elif isinstance(node, E.BatchNorm) and node.act != E.ACT_NONE:
            relu.append(cpu(node.y) > 0)
        elif isinstance(node, E.StridedBlock):
            if node.u1.relu_in:
                relu.append(cpu(node.x) > 0)
            relu.append(lazy_mask(node.u1))
            B, OH, OW, C = node.y.shape
            taps = cpu(node.idx).view(torch.uint8).reshape(B, OH, OW, C).long()
            pool.append(taps.permute(0, 3, 1, 2).unsqueeze(2))
        elif isinstance(node, E.MiddleBlock):
            relu += [cpu(node.x) > 0, lazy_mask(node.u1), lazy_mask(node.u2)]
        elif isinstance(node, E.ExitBlock):
            relu += [lazy_mask(node.u1), cpu(node.u2.y) > 0]
    return T.Decisions(relu, pool)


def assert_gradients_match(eng, P, X, Y, mask, tol=None, tie=1e-5, loss_type="same", sigmoid_cols=None,
                           max_override_rate=1e-5):
    """Every parameter gradient of the engine's last forward/backward against the fp64 oracle evaluated on the
    device's own discrete decisions: max|diff| <= tol * max|ref| on EVERY tensor, and every decision in which the
    device departs from the oracle's own must have been a tie (|pre-activation| or window gap <= tie x the largest
    value of that tensor), and at most max_override_rate of all decisions may be overridden at all.  tol=None: GRAD_TOL.
    Returns (data_loss64, y_pred64, params64, decisions); the worst per-tensor error is logged through observe()."""
    tol = GRAD_TOL if tol is None else tol
    dec = device_decisions(eng)
    data64, g64, yp64, P64 = oracle_grads(P, X, Y, mask, double=True, decisions=dec, loss_type=loss_type,
                                          sigmoid_cols=sigmoid_cols)
    assert dec._ri == len(dec.relu) and dec._pi == len(dec.pool), "decision sites out of step with the oracle"
    far = [f for f in dec.flips if f[2] > tie]
    assert not far, "device decisions differ from the oracle's away from ties: %s" % far[:8]
    # ... and ties are rare: observed 10 overrides in 30 M decisions at 384x512 (every one within 4e-7 of the kink)
    n_over = sum(f[1] for f in dec.flips)
    assert n_over <= max(2, max_override_rate * dec.n_decisions), \
        "%d of %d decisions overridden: %s" % (n_over, dec.n_decisions, dec.flips[:8])
    gd = eng.grad_dict()
    # a tensor whose true gradient vanishes identically (a bias in front of a training-mode BatchNorm: Inception-
    # ResNet's block8_10_conv/bias, 1e-17 in fp64) is measured against 1e-6 of the model's largest gradient entry
    floor = 1e-6 * max(float(np.abs(v.numpy()).max()) for v in g64.values())
    bad, worst = {}, (0.0, "")
    for k in g64:
        ref = g64[k].numpy()
        e = float(np.abs(gd[k].numpy().astype(np.float64) - ref).max()) / max(float(np.abs(ref).max()), floor)
        vanishing = float(np.abs(ref).max()) < floor
        if not vanishing:
            worst = max(worst, (e, k))
        if e > (max(tol, GRAD_TOL_VANISHING) if vanishing else tol):
            bad[k] = e
    observe("grad_rel_err_worst_tensor", worst[0], worst[1])
    observe("decisions_overridden", n_over, "of %d; largest gap %.3g" % (dec.n_decisions, max([f[2] for f in dec.flips] or [0.0])))
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    return data64, yp64, P64, dec
