"""MobileNet backbone (BASELINE configs[0]; spnet/models.py:346-355: keras.applications.mobilenet.MobileNet behind the
stem): structure known-answers, strided depthwise kernels, whole-network parity with the oracle, and the reference's
plumbing run -- train_spnet.py on a small fake-ESPI set at batch 8."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as T
from tests.parity_util import assert_forward_mse, assert_gradients_match, assert_output_close, make_case, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_structure_matches_keras_mobilenet():
    _need_gpu()
    from spnet_amd.engine import Engine, mobilenet_out_hw, param_specs
    specs = param_specs(448, 448, backbone="MobileNet")
    body = [s for s in specs if not s[0].startswith(("conv2d_", "batch_normalization_", "FinalOutput"))]
    total = sum(int(np.prod(s[1])) for s in body)
    trainable = sum(int(np.prod(s[1])) for s in body if s[2])
    # keras.applications.MobileNet(alpha=1, include_top=False): 3,228,864 parameters, 3,206,976 trainable
    assert (total, trainable, total - trainable) == (3228864, 3206976, 21888)
    assert mobilenet_out_hw(448, 448) == (7, 7)          # 224x224 behind the stem -> the canonical 7x7x1024
    eng = Engine(96, 128, 2, device="cuda:0", train=False, backbone="MobileNet")
    assert tuple(eng.backbone_out.shape) == (2, 2, 2, 1024)
    assert list(eng.state_dict().keys()) == list(T.init_params(96, 128, basemodel="MobileNet").keys())


@pytest.mark.parametrize("B,H,W,C,stride", [(2, 12, 16, 64, 2), (3, 13, 17, 32, 2), (2, 47, 63, 128, 2), (2, 9, 11, 24, 1)])
def test_strided_depthwise_kernels(B, H, W, C, stride):
    _need_gpu()
    from spnet_amd import _lib as L
    st = torch.cuda.current_stream().cuda_stream
    rs = np.random.RandomState(B * H + C)
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(rs.randn(3, 3, C), dtype=torch.float64, requires_grad=True)
    y = T.dwconv3x3(x, w, stride)
    dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float64)
    y.backward(dy)
    xd, wd, dyd = x.detach().float().cuda(), w.detach().float().cuda(), dy.float().cuda()
    yd = torch.full(tuple(y.shape), float("nan"), device="cuda")
    L.spnet_dwconv3x3_strided(0, xd.data_ptr(), wd.data_ptr(), yd.data_ptr(), B, H, W, C, stride, None, st)
    np.testing.assert_allclose(yd.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-5)
    dxd = torch.full((B, H, W, C), float("nan"), device="cuda")
    L.spnet_dwconv3x3_strided(1, dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), B, H, W, C, stride, None, st)
    np.testing.assert_allclose(dxd.cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=1e-5)
    ws = torch.empty(L.spnet_dwconv3x3_strided_ws(B, H, W, C, stride), device="cuda")
    dwd = torch.full((3, 3, C), float("nan"), device="cuda")
    L.spnet_dwconv3x3_strided(2, xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), B, H, W, C, stride, ws.data_ptr(), st)
    np.testing.assert_allclose(dwd.cpu().numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W))


@pytest.mark.parametrize("H,W,B,seed", [(96, 128, 2, 0), (75, 131, 2, 1), (224, 224, 1, 2)])
def test_mobilenet_forward_and_gradients(H, W, B, seed):
    _need_gpu()
    from spnet_amd.engine import Engine
    P, X, Y, mask, dseed = make_case(H, W, B, seed, basemodel="MobileNet")
    eng = Engine(H, W, B, device="cuda:0", seed=1, backbone="MobileNet")
    eng.load_state_dict(P)
    want = T.forward(P, X, training=False)
    got = eng.forward(X.cuda(), training=False).cpu()
    assert_forward_mse(got, want)
    eng.set_drop_seed(dseed)
    out = eng.forward(X.cuda(), training=True)
    loss = eng.loss(Y.cuda())
    eng.backward()
    torch.cuda.synchronize()
    data64, yp64, P64, _ = assert_gradients_match(eng, P, X, Y, mask)
    assert_output_close(out.cpu().numpy(), yp64.numpy())
    np.testing.assert_allclose(float(loss[5]), data64, rtol=1e-4)
    sd = eng.state_dict()
    for k in P:
        if k.endswith("moving_mean") or k.endswith("moving_variance"):
            np.testing.assert_allclose(sd[k].numpy(), P64[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_mobilenet_l2_set_and_training_steps():
    _need_gpu()
    from spnet_amd.engine import Engine
    H, W, B = 96, 128, 4
    P, X, Y, mask, dseed = make_case(H, W, B, 3, basemodel="MobileNet")
    eng = Engine(H, W, B, device="cuda:0", seed=1, backbone="MobileNet")
    eng.load_state_dict(P)
    l2_names = [n for n, (off, cnt, _) in eng.p_off.items() if off < eng.l2_n]
    assert sorted(l2_names) == sorted(n + "/kernel" for n in T.L2_KERNELS_MOBILENET)
    losses = []
    for _ in range(6):
        out = eng.train_step(X.cuda(), Y.cuda(), 1e-3)
        torch.cuda.synchronize()
        losses.append(out.cpu().numpy()[:7].copy())
    losses = np.array(losses)
    want_l2 = float(T.l2_penalty(P))
    assert abs(losses[0, 6] - want_l2) <= 1e-4 * want_l2            # penalty reported with the pre-step weights
    assert np.all(np.isfinite(losses)) and losses[-1, 5] < losses[0, 5]


def test_config0_plumbing_run_with_mobilenet(tmp_path):
    """BASELINE configs[0]: small fake-ESPI set, MobileNet backbone, batch 8, through train_spnet.py."""
    _need_gpu()
    from spnet_amd import fake_espi as F
    data = tmp_path / "data"
    F.write_dataset(str(data / "Train"), 48, seed=1)
    F.write_dataset(str(data / "Val"), 16, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train_spnet.py"), "-d", str(data), "-b", "8", "-e", "3",
                        "--name", "mb", "--backbone", "MobileNet"], cwd=str(work), env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "cf.basemodel = MobileNet" in r.stdout and "SPNet execution completed." in r.stdout
    logs = [d for d in os.listdir(work / "logs") if d.startswith("mb_")]
    rows = [l for l in open(work / "logs" / logs[0] / "losses.dat") if not l.startswith("#")]
    losses = [float(x.split()[1]) for x in rows]
    assert len(rows) == 3 and all(np.isfinite(losses)) and losses[-1] < losses[0]
    from safetensors import safe_open
    with safe_open(str(work / "full_model.h5"), framework="pt") as f:
        assert f.metadata()["basemodel"] == "MobileNet" and "conv_pw_13/kernel" in f.keys()
