"""Checks at the BASELINE sizes and layouts, through size-independent properties where the oracle is too
slow: the reference 331x331 layout (structural known-answers of the run log + forward parity), the
benchmark configuration 384x512 / batch 32 (determinism, learning, linearity), the hybrid loss, ragged
predict batches."""
import numpy as np
import pytest
import torch

from tests.parity_util import assert_forward_mse  # noqa: E402

pytestmark = pytest.mark.gpu

from oracle import torch_ref as T


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_reference_331_layout_shapes_and_forward():
    _need_gpu()
    from spnet_amd.engine import Engine
    eng = Engine(331, 331, 2, device="cuda:0", seed=1, train=False)
    assert tuple(eng.stem_out.shape) == (2, 165, 165, 3)            # run log: "(?, 165, 165, 3)"
    assert tuple(eng.backbone_out.shape) == (2, 5, 5, 2048)         # run log: "(None, 5, 5, 2048)"
    assert eng.n_theta >= 50298935
    P = T.init_params(331, 331, seed=4)
    eng.load_state_dict(P)
    X = torch.tensor(np.random.RandomState(0).rand(2, 331, 331, 1) * 2 - 1, dtype=torch.float32)
    want = T.forward(P, X, training=False)
    got = eng.forward(X.cuda(), training=False).cpu()
    assert_forward_mse(got, want)      # north-star tolerance: 1e-4; fp32 against fp32 is far tighter


@pytest.fixture(scope="module")
def full():
    _need_gpu()
    from spnet_amd.engine import Engine
    eng = Engine(384, 512, 32, device="cuda:0", seed=0)
    assert tuple(eng.backbone_out.shape) == (32, 6, 8, 2048)
    rs = np.random.RandomState(0)
    X = torch.tensor(rs.rand(32, 384, 512, 1) * 2 - 1, dtype=torch.float32).cuda()
    Y = torch.tensor(rs.rand(32, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    return eng, X, Y.cuda()


def test_full_config_is_deterministic_and_learns(full):
    eng, X, Y = full
    runs = []
    for rep in range(2):
        eng.use_graph = (rep == 1)       # second run: steps 3.. are hipGraph replays -> must be bit-identical to eager
        eng.init_weights(0)
        eng.drop_seed = 7
        losses = []
        for s in range(4):
            out = eng.train_step(X, Y, 1e-4)
            torch.cuda.synchronize()
            losses.append(out.cpu().numpy()[:7].copy())
        runs.append((np.array(losses), eng.theta.double().sum().item(), eng.theta[:1000].cpu().clone()))
    eng.use_graph = False
    assert np.array_equal(runs[0][0], runs[1][0])                    # bit-identical losses (no float atomics anywhere)
    assert runs[0][1] == runs[1][1] and torch.equal(runs[0][2], runs[1][2])
    data = runs[0][0][:, 5]
    assert np.all(np.isfinite(runs[0][0])) and data[-1] < data[0]    # the same batch is learned
    parts = runs[0][0][:, :5].sum(1)
    np.testing.assert_allclose(parts, data, rtol=1e-5)              # five terms add up to custom_loss


def test_host_running_ahead_equals_synchronous_stepping(full):
    """Per-step scalars (Adam's bias-corrected step size, the dropout seed) reach the device through a ring of pinned
    slots copied in stream order: six steps enqueued back to back (the host runs several steps ahead of the GPU, every
    step with another learning rate) must leave exactly the weights of six steps with a device sync after each."""
    eng, X, Y = full
    lrs = [1e-4, 3e-4, 5e-5, 2e-4, 7e-5, 1e-4]
    finals = []
    for sync in (False, True):
        eng.init_weights(0)
        eng.drop_seed = 11
        for lr in lrs:
            out = eng.train_step(X, Y, lr)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        finals.append((eng.theta.clone(), out.clone(), eng.t))
    assert finals[0][2] == finals[1][2] == 6
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])


def test_full_size_linearity_of_gemm_and_depthwise(full):
    eng, X, Y = full
    from spnet_amd import _lib as L
    st = torch.cuda.current_stream().cuda_stream
    rs = torch.Generator(device="cuda").manual_seed(1)
    M, N, K = 32 * 12 * 16, 728, 728                                # the network's dominant GEMM
    A = torch.randn(M, K, device="cuda", generator=rs)
    B1, B2 = torch.randn(K, N, device="cuda", generator=rs), torch.randn(K, N, device="cuda", generator=rs)
    ws = torch.empty(1 << 20, device="cuda")
    outs = []
    for Bm in (B1, B2, B1 + B2):
        C = torch.empty(M, N, device="cuda")
        L.spnet_gemm_f32(A.data_ptr(), 0, K, Bm.data_ptr(), 1, N, C.data_ptr(), N, M, N, K, 0, ws.data_ptr(), ws.numel(), None, 0, st)
        outs.append(C)
    torch.testing.assert_close(outs[0] + outs[1], outs[2], rtol=1e-4, atol=2e-3)
    Bq, Hh, Ww, Cc = 32, 93, 125, 128                               # the largest depthwise plane
    x1, x2 = torch.randn(Bq, Hh, Ww, Cc, device="cuda", generator=rs), torch.randn(Bq, Hh, Ww, Cc, device="cuda", generator=rs)
    w = torch.randn(3, 3, Cc, device="cuda", generator=rs)
    ys = []
    for xx in (x1, x2, 2.0 * x1 - x2):
        y = torch.empty_like(xx)
        L.spnet_dwconv3x3_tiled_fwd(xx.data_ptr(), w.data_ptr(), y.data_ptr(), Bq, Hh, Ww, Cc, 0, None, None, st)
        ys.append(y)
    torch.testing.assert_close(2.0 * ys[0] - ys[1], ys[2], rtol=1e-4, atol=1e-4)
    # a constant image through a depthwise conv: interior = sum of taps, corners see 4 taps (SAME zero padding)
    ones = torch.ones(1, Hh, Ww, Cc, device="cuda")
    y = torch.empty_like(ones)
    L.spnet_dwconv3x3_tiled_fwd(ones.data_ptr(), w.data_ptr(), y.data_ptr(), 1, Hh, Ww, Cc, 0, None, None, st)
    torch.testing.assert_close(y[0, 40, 60], w.sum((0, 1)), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y[0, 0, 0], w[1:, 1:].sum((0, 1)), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(y[0, Hh - 1, Ww - 1], w[:2, :2].sum((0, 1)), rtol=1e-5, atol=1e-5)


def test_hybrid_loss_gradients_small():
    _need_gpu()
    from spnet_amd.engine import Engine
    H, W, B = 96, 128, 2
    eng = Engine(H, W, B, device="cuda:0", seed=2, loss_type="hybrid")
    P = T.init_params(H, W, seed=6)
    eng.load_state_dict(P)
    rs = np.random.RandomState(3)
    X = torch.tensor(rs.rand(B, H, W, 1) * 2 - 1, dtype=torch.float32)
    Y = torch.tensor(rs.rand(B, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    tr = T.Trainer({k: v.clone() for k, v in P.items()}, loss_type="hybrid")
    eng.set_drop_seed(1)
    # dropout mask of the engine for this seed (same hash as tests/test_engine_gpu.py)
    from tests.test_engine_gpu import dropout_mask
    mask = torch.tensor(dropout_mask(B * (H // 2) * (W // 2) * 3, 1).reshape(B, H // 2, W // 2, 3))
    data, total, grads, yp = tr.grads(X, Y, drop_mask=mask, include_l2=False)
    eng.forward(X.cuda(), training=True)
    loss = eng.loss(Y.cuda())
    eng.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss[5]), data, rtol=1e-4)
    # every tensor, on the device's own ReLU / max-pool decisions, fp64 reference (tests/test_shapes_gpu.py)
    from tests.parity_util import assert_gradients_match
    assert_gradients_match(eng, P, X, Y, mask, loss_type="hybrid")


def test_predict_ragged_batches_match():
    _need_gpu()
    from spnet_amd import config as cf
    from spnet_amd.models import Model
    old = cf.model_type
    try:
        m = Model((64, 96, 1), Y0size=576, seed=5)
        X = (np.random.RandomState(1).rand(7, 64, 96, 1).astype(np.float32) * 2 - 1)
        a = m.predict(X, batch_size=7)
        b = m.predict(X, batch_size=3)        # 3 + 3 + 1 (padded tail)
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
        assert a.shape == (7, 576) and np.isfinite(a).all()
    finally:
        cf.model_type = old
