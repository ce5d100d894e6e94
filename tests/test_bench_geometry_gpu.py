"""Oracle parity AT the benchmark geometry (BASELINE.json configs[1] and configs[4]: 384x512 frames).

The plane sizes of the benchmark step -- 192x256 (stem), 95x127, 93x125, 47x63, 24x32, 12x16, 6x8 -- pick kernel
paths the small-geometry tests never reach: the XCD-remapped depthwise tiles (>= 8 channel chunks per tile),
ragged 93x125 / 47x63 tiles, the 95x127 implicit GEMM of block1_conv2.  Batch 2 keeps the CPU oracle at a few
seconds; the batch dimension only multiplies the pixel count M of every kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as T
from tests.parity_util import assert_forward_mse, assert_gradients_match, assert_output_close, make_case, rel_err

H, W = 384, 512


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def case():
    _need_gpu()
    return make_case(H, W, 2, 0)


def test_inference_forward_at_benchmark_geometry(case):
    from spnet_amd.engine import Engine
    P, X, Y, mask, dseed = case
    eng = Engine(H, W, 2, device="cuda:0", seed=1, train=False)
    eng.load_state_dict(P)
    taps = {}
    want = T.forward(P, X, training=False, taps=taps)
    got = eng.forward(X.cuda(), training=False).cpu()
    np.testing.assert_allclose(eng.stem_out.cpu().numpy(), taps["stem"].numpy(), rtol=1e-4, atol=1e-5)
    scale = float(taps["backbone"].abs().max())
    np.testing.assert_allclose(eng.backbone_out.cpu().numpy(), taps["backbone"].numpy(), rtol=1e-3, atol=1e-4 * scale)
    assert_forward_mse(got, want)      # north-star tolerance: 1e-4; fp32 against fp32 is far tighter


@pytest.mark.parametrize("fuse_dw_bwd", [False, True])
def test_training_forward_and_every_gradient_at_benchmark_geometry(case, fuse_dw_bwd):
    """(fuse_dw_bwd: the depthwise backward of the 12x16 / 6x8-plane layers inside the data-gradient GEMM's epilogue,
    spnet_gemm_bf16x3_pp_dwbwd -- an option of the plan, off by default; the same gradients either way.)
    Gradients against the fp64 oracle on the device's own ReLU / max-pool decisions (see tests/test_shapes_gpu.py:
    at this size some of the ~30 million ReLU inputs always sit within fp32 rounding of zero, and the fp32 CPU oracle
    itself is 2e-1 away from its fp64 evaluation in single tensors); every decision that differs must be a tie."""
    from spnet_amd.engine import Engine
    P, X, Y, mask, dseed = case
    # x3_min_tiles = 0: the pointwise forward / data-gradient GEMMs on the bf16x3 kernel as in the batch-32 benchmark plan
    # (batch 2 has too few rows for the engine's own 192-tile rule)
    eng = Engine(H, W, 2, device="cuda:0", seed=1, x3_min_tiles=0, fuse_dw_bwd=fuse_dw_bwd)
    assert sum(p.x3_fwd for p in eng._pw_layers) >= 30 and sum(p.x3_dgrad for p in eng._pw_layers) >= 28
    from spnet_amd.engine import MiddleBlock
    assert all(u.fuse_bwd == fuse_dw_bwd for n in eng.nodes if isinstance(n, MiddleBlock) for u in (n.u1, n.u2, n.u3))
    assert eng.nodes[-2].u1.fuse_bwd == fuse_dw_bwd           # (x3_min_tiles = 0: the exit flow's 6 x 8 planes fuse as well)
    eng.load_state_dict(P)
    eng.set_drop_seed(dseed)
    out = eng.forward(X.cuda(), training=True)
    loss = eng.loss(Y.cuda())
    eng.backward()
    torch.cuda.synchronize()
    data64, yp64, P64, dec = assert_gradients_match(eng, P, X, Y, mask)
    print("decisions overridden (site, count, distance from the tie):", dec.flips)
    assert_output_close(out.cpu().numpy(), yp64.numpy())
    np.testing.assert_allclose(float(loss[5]), data64, rtol=1e-4)
    sd = eng.state_dict()
    for k in P:                          # BatchNorm moving statistics after one training forward
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            np.testing.assert_allclose(sd[k].numpy(), P64[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_predict_config4_batch128(case):
    """BASELINE configs[4]: model.predict over 512x384 frames at batch 128 (predict_spnet.py:84-87).  Frames 0-1
    against the oracle; all 128 against the batch-2 plan of the same weights (another tile choice for every GEMM,
    so equal to rounding, not to the bit)."""
    from spnet_amd.models import Model
    P, X2, _, _, _ = case
    m = Model((H, W, 1), Y0size=576, seed=3)
    m.load_state_dict(P)
    rs = np.random.RandomState(4)
    X = (rs.rand(128, H, W, 1).astype(np.float32) * 2 - 1)
    X[:2] = X2.numpy()
    y128 = m.predict(X, batch_size=128)
    assert y128.shape == (128, 576) and np.isfinite(y128).all()
    want = T.forward(P, X2, training=False).numpy()
    assert_forward_mse(y128[:2], want)
    y2 = m.predict(X, batch_size=2)
    scale = float(np.abs(y2).max())
    np.testing.assert_allclose(y128, y2, rtol=1e-4, atol=1e-5 * scale)
    # streamed twice through the same plan: bit-identical
    assert np.array_equal(m.predict(X, batch_size=128), y128)
