"""End-to-end parity of the HIP engine against the torch-CPU oracle on identical weights and frames:
inference forward, training forward (batch statistics + dropout), every parameter gradient, several
optimizer steps; plus the device augmentation against golden frames produced by the reference."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import numpy_ref as R
from oracle import torch_ref as T

H, W, B = 96, 128, 2
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


from tests.parity_util import assert_forward_mse, dropout_mask  # noqa: E402,F401  (dropout_mask re-exported for the other test modules)


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd.engine import Engine
    eng = Engine(H, W, B, device="cuda:0", seed=11)
    P = T.init_params(H, W, seed=5)
    # non-trivial BN parameters / moving statistics so that every term is exercised
    g = torch.Generator().manual_seed(1)
    for k in P:
        if k.endswith("/gamma"):
            P[k] = 0.5 + torch.rand(P[k].shape, generator=g)
        elif k.endswith("/beta") or k.endswith("/moving_mean") or k.endswith("/bias"):
            P[k] = 0.2 * torch.randn(P[k].shape, generator=g)
        elif k.endswith("/moving_variance"):
            P[k] = 0.5 + torch.rand(P[k].shape, generator=g)
    rs = np.random.RandomState(0)
    X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32)
    Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    return eng, P, X, Y


def test_structure(setup):
    eng, P, X, Y = setup
    sd = eng.state_dict()
    assert list(sd.keys()) == list(P.keys())
    assert all(tuple(sd[k].shape) == tuple(P[k].shape) for k in P)
    from spnet_amd.engine import param_specs
    n331 = sum(int(np.prod(s[1])) for s in param_specs(331, 331))
    assert n331 == 50353481                       # reference run log: total params


def test_inference_forward(setup):
    eng, P, X, Y = setup
    eng.load_state_dict(P)
    taps = {}
    want = T.forward(P, X, training=False, taps=taps)
    got = eng.forward(X.cuda(), training=False).cpu()
    np.testing.assert_allclose(eng.stem_out.cpu().numpy(), taps["stem"].numpy(), rtol=1e-4, atol=1e-5)
    scale = float(taps["backbone"].abs().max())
    np.testing.assert_allclose(eng.backbone_out.cpu().numpy(), taps["backbone"].numpy(), rtol=1e-3, atol=1e-4 * scale)
    assert_forward_mse(got, want)      # BASELINE tolerance is 1e-4; fp32 vs fp32 is far tighter


def test_training_forward_and_gradients(setup):
    eng, P, X, Y = setup
    eng.load_state_dict(P)
    seed = 424242
    eng.set_drop_seed(seed)
    h2, w2 = H // 2, W // 2
    mask = torch.tensor(dropout_mask(B * h2 * w2 * 3, seed).reshape(B, h2, w2, 3))
    Pc = {k: v.clone() for k, v in P.items()}
    tr = T.Trainer(Pc)
    data, total, grads, yp = tr.grads(X, Y, drop_mask=mask, include_l2=False)

    out = eng.forward(X.cuda(), training=True)
    loss = eng.loss(Y.cuda())
    eng.backward()
    torch.cuda.synchronize()
    yscale = float(yp.abs().max())
    np.testing.assert_allclose(out.cpu().numpy(), yp.numpy(), rtol=2e-3, atol=2e-4 * yscale)
    np.testing.assert_allclose(float(loss[5]), data, rtol=1e-4)
    # BN moving statistics after one training forward
    sd = eng.state_dict()
    for k in P:
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            np.testing.assert_allclose(sd[k].numpy(), Pc[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # every parameter gradient, tensor by tensor, relative to that tensor's largest entry: against the oracle in
    # fp32 as is, and against the fp64 oracle on the device's discrete decisions (tests/test_shapes_gpu.py)
    gd = eng.grad_dict()
    worst = {}
    for k, g in grads.items():
        ref = g.numpy()
        got = gd[k].numpy()
        denom = max(float(np.abs(ref).max()), 1e-12)
        worst[k] = float(np.abs(got - ref).max()) / denom
    bad = {k: v for k, v in worst.items() if v > 5e-3}
    assert not bad, "gradient mismatch (max|diff|/max|ref|): %s" % sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    from tests.parity_util import assert_gradients_match
    assert_gradients_match(eng, P, X, Y, mask)


def test_train_steps_follow_oracle(setup):
    eng, P, X, Y = setup
    eng.load_state_dict(P)
    eng.m.zero_()
    eng.v.zero_()
    eng.t = 0
    Pc = {k: v.clone() for k, v in P.items()}
    tr = T.Trainer(Pc)
    lr, steps = 1e-4, 3
    h2, w2 = H // 2, W // 2
    eng.drop_seed = 99
    losses_dev, losses_ref = [], []
    for s in range(steps):
        nxt = (eng.drop_seed * 1664525 + 1013904223) & 0xFFFFFFFF      # the engine's per-step seed sequence
        mask = torch.tensor(dropout_mask(B * h2 * w2 * 3, nxt).reshape(B, h2, w2, 3))
        d, t = tr.step(X, Y, lr, drop_mask=mask)
        out = eng.train_step(X.cuda(), Y.cuda(), lr)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        losses_dev.append((float(o[5]), float(o[5] + o[6])))
        losses_ref.append((d, t))
    np.testing.assert_allclose(np.array(losses_dev), np.array(losses_ref), rtol=2e-3)
    sd = eng.state_dict()
    # Adam normalises by sqrt(v): an element whose gradient is at rounding level can move by up to lr per step in
    # either direction, and from the second step on every element sees the (slightly) different weights of the first,
    # so weights are compared in units of lr*steps.  The fraction beyond 0.1 lr*steps is a noise statistic, not a
    # rounding bound: builds that differ ONLY in fp32 rounding (the BatchNorm shift evaluated with or without a fused
    # multiply-add) measure 0.0019 and 0.0023 here; the single-step checks -- every gradient against the oracle above,
    # the Adam kernel against numpy_ref.adam_step in test_kernels_gpu.py -- are the ones that bound errors.
    frac_bad, worst = 0.0, 0.0
    n = 0
    for k in P:
        if T.is_trainable(k):
            d = np.abs(sd[k].numpy() - Pc[k].numpy()) / (lr * steps)
            worst = max(worst, float(d.max()))
            frac_bad += float((d > 0.1).sum())
            n += d.size
    assert worst <= 2.0 + 1e-3, worst
    assert frac_bad / n < 3e-3, frac_bad / n


def test_device_augmentation_matches_reference_golden(golden):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd.augmentation import DeviceAugmenter
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in golden[f"aug_{tag}_shape"])
        X = (np.random.RandomState(5).rand(*shape).astype(np.float32) * 2 - 1)
        want = X.copy().ravel()
        want[golden[f"aug_{tag}_changed_idx"]] = golden[f"aug_{tag}_changed_val"]
        want = want.reshape(shape)
        aug = DeviceAugmenter(torch.from_numpy(X).cuda())
        out = torch.empty(shape, device="cuda")
        np.random.seed(1234)
        random.seed(1234)
        aug.augment(list(range(shape[0])), out)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), want)          # bit-exact with the reference's numpy code
        assert np.random.rand() == float(golden[f"aug_{tag}_rng_after"])   # same RNG consumption


def test_graph_replayed_training_invalidates_inference_coefficients():
    """SPNET_TRAIN_GRAPH=1: replayed steps skip _step_body's host code, so train_step must itself note that the
    weights and moving statistics moved -- otherwise predict_step replays its inference graph with BatchNorm
    scale|shift built from the OLD gamma/beta/moving statistics (round-2 ADVICE).  Twin engines, one eager and one
    replaying, must predict the same."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd.engine import Engine
    h, w, b = 64, 96, 2
    rs = np.random.RandomState(3)
    X = torch.tensor(rs.rand(b, h, w, 1) * 2 - 1, dtype=torch.float32).cuda()
    Y = torch.tensor(rs.rand(b, 576), dtype=torch.float32).cuda()
    outs = []
    for graph in (False, True):
        eng = Engine(h, w, b, device="cuda:0", seed=21)
        eng.use_graph = graph
        eng.x_in.copy_(X)
        first = eng.predict_step(use_graph=True).clone()      # builds the inference graph and its coefficients
        for _ in range(4):                                     # step 1 eager (warm), 2 captures, 3-4 replay
            eng.train_step(X, Y, 1e-3)
        assert (eng._graph is not None) == graph
        eng.x_in.copy_(X)
        out = eng.predict_step(use_graph=True)
        torch.cuda.synchronize()
        out = (eng.out if out is None else out).clone()
        assert float((out - first).abs().max()) > 1e-6        # four steps at lr 1e-3 moved the prediction
        outs.append(out.cpu())
    np.testing.assert_allclose(outs[1].numpy(), outs[0].numpy(), rtol=0, atol=1e-5)


def test_captured_train_step_contains_the_weight_split():
    """SPNET_TRAIN_GRAPH=1 with the bf16x3 planes kernels on (x3_min_tiles=0): a predict pass between the warm eager step
    and the capturing step used to leave the weight planes "fresh", so the captured graph held no split and every replay
    multiplied with the weights of the capture step (round-4 ADVICE).  The split now runs inside the optimizer step: twin
    engines, eager and replaying, step -> predict -> three more steps, must end with the same weights."""
    _need_gpu()
    from spnet_amd.engine import Engine
    h, w, b = 64, 96, 2
    rs = np.random.RandomState(4)
    X = torch.tensor(rs.rand(b, h, w, 1) * 2 - 1, dtype=torch.float32).cuda()
    Y = torch.tensor(rs.rand(b, 576), dtype=torch.float32).cuda()
    res = []
    for graph in (False, True):
        eng = Engine(h, w, b, device="cuda:0", seed=5, x3_min_tiles=0)
        assert any(u.x3_fwd for u in eng._pw_layers)
        eng.use_graph = graph
        eng.train_step(X, Y, 1e-3)                 # warm (eager)
        eng.x_in.copy_(X)
        eng.predict_step(use_graph=False)          # a forward of a plan over the same weights: refreshes the planes
        losses = [eng.train_step(X, Y, 1e-3)[:7].clone() for _ in range(3)]      # capture, replay, replay
        torch.cuda.synchronize()
        assert (eng._graph is not None) == graph
        res.append((eng.theta.clone(), torch.stack(losses)))
    np.testing.assert_allclose(res[1][1].cpu().numpy(), res[0][1].cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res[1][0].cpu().numpy(), res[0][0].cpu().numpy(), rtol=0, atol=2e-6)


def test_one_stream_step_equals_two_stream_step(monkeypatch):
    """SPNET_OVERLAP_WGRAD=0 (weight gradients on the main stream instead of the side stream that is joined before
    Adam; what bench.py's roofline leg runs): the same arithmetic in another launch order -- weights, Adam moments and
    losses of three steps are bit-identical to the two-stream plan's."""
    _need_gpu()
    from spnet_amd.engine import Engine
    h, w, b = 96, 128, 2
    rs = np.random.RandomState(8)
    X = torch.tensor(rs.rand(b, h, w, 1) * 2 - 1, dtype=torch.float32).cuda()
    Y = torch.tensor(rs.rand(b, 576), dtype=torch.float32).cuda()
    res = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("SPNET_OVERLAP_WGRAD", overlap)
        eng = Engine(h, w, b, device="cuda:0", seed=13)
        assert (eng.wgrad_stream is not None) == (overlap == "1")
        losses = [eng.train_step(X, Y, 1e-3)[:7].clone() for _ in range(3)]      # (slot 7 of loss_out is unused)
        torch.cuda.synchronize()
        res.append((eng.theta.clone(), eng.m.clone(), eng.v.clone(), torch.stack(losses)))
    for a, c in zip(res[0], res[1]):
        assert torch.equal(a, c)


def test_early_head_optimizer_step_equals_the_step_at_the_end():
    """Engine(early_head=True) -- the Dense head's weight gradient on the weight-gradient stream and its range of the
    optimizer step on a third stream right behind it, underneath the backbone's backward -- against early_head=False (both
    on the main stream, the optimizer's two launches after backward): the same launches in another order, so weights, Adam
    moments, losses (incl. the l2 penalty the optimizer kernel sums) of three steps are bit-identical; eager and as a
    captured graph."""
    _need_gpu()
    from spnet_amd.engine import Engine
    h, w, b = 96, 128, 2
    rs = np.random.RandomState(9)
    X = torch.tensor(rs.rand(b, h, w, 1) * 2 - 1, dtype=torch.float32).cuda()
    Y = torch.tensor(rs.rand(b, 576), dtype=torch.float32).cuda()
    res = []
    for early, graph in ((True, False), (False, False), (True, True)):
        eng = Engine(h, w, b, device="cuda:0", seed=13, early_head=early)
        eng.use_graph = graph
        assert eng._head_hi > 0 and eng._head_hi % 4 == 0 and eng._head_hi == eng.p_off["FinalOutput/kernel"][1]
        losses = [eng.train_step(X, Y, 1e-3)[:7].clone() for _ in range(4)]
        torch.cuda.synchronize()
        assert (eng._graph is not None) == graph
        assert (eng._opt_stream is not None) == early
        res.append((eng.theta.clone(), eng.m.clone(), eng.v.clone(), torch.stack(losses)))
    for other in res[1:]:
        for a, c in zip(res[0], other):
            assert torch.equal(a, c)
    assert float(res[0][3][-1, 6]) > 0          # the l2 penalty of the regularised kernels, folded over both ranges


def test_graph_capture_survives_garbage_of_an_earlier_plan():
    """The abort of round 3 (`Fatal Python error: Aborted ... Garbage-collecting` under predict_step), reproduced on
    purpose in a child process: an earlier plan's graph / streams / events reachable only through reference cycles, the
    cyclic collector set to run on nearly every allocation, then a capture.  tests/helpers/capture_gc.py."""
    _need_gpu()
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "capture_gc.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURE_GC_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_stream_accessor_matches_the_public_handle():
    """_lib.current_stream (torch's private raw accessor, resolved once) returns the handle of the public API, on the
    default stream and inside a `with torch.cuda.stream(...)` block."""
    _need_gpu()
    from spnet_amd import _lib as L
    assert L.current_stream() == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert L.current_stream() == side.cuda_stream == torch.cuda.current_stream().cuda_stream
    assert L.current_stream() == torch.cuda.current_stream().cuda_stream


def test_exact_chain_mode_and_bf16x3_mode_agree(setup):
    """Engine(pointwise=...): "bf16x3" (the default: pointwise forward / data-gradient GEMMs with >= 256 output columns on
    csrc/gemm_bf16x3.hip) and "f32" (every GEMM the k-ordered fp32 MFMA chain, rounds 1-3).  Not the same bits -- the same
    results to fp32 accuracy: a training forward / backward in either mode meets the suite's gradient tolerance against
    the fp64 oracle on the device's decisions, the outputs are within 1e-5 of each other, and the default plan really
    launches the bf16x3 kernel."""
    _need_gpu()
    from tests.parity_util import assert_gradients_match, rel_err
    from spnet_amd.engine import Engine, KernelTimer
    _, P, X, Y = setup
    seed = 99
    h2, w2 = H // 2, W // 2
    mask = torch.tensor(dropout_mask(B * h2 * w2 * 3, seed).reshape(B, h2, w2, 3))
    outs = []
    for mode in ("bf16x3", "f32"):
        eng = Engine(H, W, B, device="cuda:0", seed=11, pointwise=mode, x3_min_tiles=0)     # (0: also at this small M)
        eng.load_state_dict(P)
        eng.set_drop_seed(seed)
        eng.prof = KernelTimer()
        out = eng.forward(X.cuda(), training=True).clone()
        eng.loss(Y.cuda())
        eng.backward()
        torch.cuda.synchronize()
        n_x3 = sum(1 for t in eng.prof.tags.values() if str(t[0]).startswith("x3"))
        assert (n_x3 > 40) == (mode == "bf16x3"), n_x3
        eng.prof = None
        outs.append(out.cpu())
        assert_gradients_match(eng, P, X, Y, mask)
        del eng
    assert not torch.equal(outs[0], outs[1])          # another summation: not the same bits ...
    assert rel_err(outs[1].numpy(), outs[0].numpy()) <= 1e-5      # ... the same numbers
    with pytest.raises(ValueError):
        Engine(H, W, B, device="cuda:0", pointwise="bf16")
