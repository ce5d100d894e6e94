"""The Keras-like Model surface on the GPU: two-phase training (freeze_fac then unfreeze_model,
train_spnet.py:71-78 of the reference), weight round trips, predict()."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rs = np.random.RandomState(3)
    X = (rs.rand(16, 96, 128, 1).astype(np.float32) * 2 - 1)
    Y = rs.rand(16, 576).astype(np.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5)
    return X, Y


def _by_prefix(sd, frozen_prefixes):
    fro = {k: v for k, v in sd.items() if k.split("/")[0] in frozen_prefixes}
    rest = {k: v for k, v in sd.items() if k.split("/")[0] not in frozen_prefixes}
    return fro, rest


def test_frozen_layers_stay_put_and_unfreeze_trains_them(data):
    from spnet_amd import models as M
    X, Y = data
    np.random.seed(1)
    model = M.create_model_functional(X, Y0size=576, freeze_fac=0.75)
    assert model._mask is not None and 0 < int((model._mask == 0).sum()) < model._mask.numel()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    model.optimizer.lr = 1e-3
    model.fit(X, Y, batch_size=8, epochs=2, shuffle=False, verbose=0)
    after = model.state_dict()
    trainable = lambda k: not (k.endswith("moving_mean") or k.endswith("moving_variance"))
    fro_b, rest_b = _by_prefix(before, model._frozen_prefixes)
    fro_a, rest_a = _by_prefix(after, model._frozen_prefixes)
    assert fro_b and rest_b
    for k in fro_b:
        if trainable(k):                     # layer.trainable=False: kernel/gamma/beta untouched, bit for bit
            assert np.array_equal(fro_b[k].numpy(), fro_a[k].numpy()), k
    moved = [k for k in rest_b if trainable(k) and not np.array_equal(rest_b[k].numpy(), rest_a[k].numpy())]
    assert len(moved) >= 0.9 * sum(trainable(k) for k in rest_b)
    # BatchNorm moving statistics follow the batches even in frozen layers (Keras 2.1.3 semantics: `trainable`
    # only removes the weights from the optimizer)
    assert any(not np.array_equal(before[k].numpy(), after[k].numpy()) for k in before if k.endswith("moving_mean"))

    # phase 2: a fresh, fully trainable model carrying the same weights
    new = M.unfreeze_model(model, X, Y)
    assert new._mask is None
    for k, v in new.state_dict().items():
        assert np.array_equal(v.numpy(), after[k].numpy()), k
    new.optimizer.lr = 1e-3
    new.fit(X, Y, batch_size=8, epochs=1, shuffle=False, verbose=0)
    final = new.state_dict()
    fro_f, _ = _by_prefix(final, model._frozen_prefixes)
    moved = [k for k in fro_f if trainable(k) and not np.array_equal(fro_f[k].numpy(), fro_a[k].numpy())]
    assert len(moved) >= 0.9 * sum(trainable(k) for k in fro_f)


def test_optimizer_state_is_shared_by_all_plans(data, tmp_path):
    """Adam's iteration count and the dropout seed sequence belong to the weights, not to a (batch size) launch plan: a
    second fit() with another batch size continues both, and the whole-model file records the real iteration count."""
    from safetensors import safe_open
    from spnet_amd import models as M
    X, Y = data
    model = M.build_model(X, Y0size=576, freeze_fac=0.0)
    model.fit(X, Y, batch_size=8, epochs=1, shuffle=False, verbose=0)           # 2 iterations
    seed_after_first = model._root.drop_seed
    assert model._root.t == 2
    model.fit(X, Y, batch_size=4, epochs=1, shuffle=False, verbose=0)           # 4 more, on another plan
    assert model._root.t == 6 and model._engine(4, True).t == 6 and model._engine(8, True).t == 6
    assert model._root.drop_seed != seed_after_first
    path = str(tmp_path / "m.h5")
    model.save(path)
    with safe_open(path, framework="pt") as f:
        assert f.metadata()["optimizer_iterations"] == "6"


def test_weights_round_trip_and_predict_is_deterministic(data, tmp_path):
    from spnet_amd import models as M
    X, Y = data
    model = M.build_model(X, Y0size=576, freeze_fac=0.0)
    p1 = model.predict(X, batch_size=5)                     # ragged last batch
    assert p1.shape == (16, 576) and p1.dtype == np.float32 and np.isfinite(p1).all()
    path = str(tmp_path / "w.hdf5")
    model.save_weights(path)
    other = M.build_model(X, Y0size=576, freeze_fac=0.0)
    other.load_weights(path)
    np.testing.assert_array_equal(other.predict(X, batch_size=8), model.predict(X, batch_size=8))
    np.testing.assert_allclose(p1, model.predict(X, batch_size=8), rtol=0, atol=1e-5)
    w = model.get_weights()
    other.set_weights([a * 0 for a in w])
    assert float(np.abs(other.predict(X[:4], batch_size=4)).max()) < 1e-3
    assert abs(model.evaluate(X, Y, batch_size=8) - M.custom_loss(Y, p1)) < 1e-5
    empty = model.predict(X[:0], batch_size=8)              # no frames: an empty [0, 576] result, no launch
    assert empty.shape == (0, 576) and empty.dtype == np.float32
    one = model.predict(X[:1], batch_size=32)               # fewer frames than the batch size
    np.testing.assert_allclose(one, p1[:1], rtol=0, atol=1e-5)


def test_device_map_matches_host_metric():
    """spnet_ellipse_iou (all 72 x N pairs in one launch) against the host raster of diagnostics.py, the
    reference's IoU known-answer, and mAP on a noisy copy of random labels."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import diagnostics as D

    def tup(cx, cy, a_, b_, ang, noobj=0.0):
        return [cx, cy, a_, b_, np.cos(2 * np.deg2rad(ang)), np.sin(2 * np.deg2rad(ang)), noobj, 1.0]

    # reference tests/test_diagnostics.py pins 0.44227983 for this pair under cv2's rasteriser
    kp, kt = np.array([tup(100, 140, 120, 60, 90)], np.float32), np.array([tup(120, 123, 120, 60, 149.97)], np.float32)
    dev_pairs = D._pair_ious_device(kp, kt)
    assert len(dev_pairs) == 1 and abs(dev_pairs[0][0] - 0.44227983107795693) < 0.01
    assert abs(dev_pairs[0][0] - D.compute_iou(kp[0], kt[0])) < 1e-3

    rs = np.random.RandomState(8)
    N = 6
    Yt = np.zeros((N, 576), np.float32)
    for i in range(N):
        for k in range(72):
            empty = rs.rand() < 0.7
            Yt[i, 8 * k:8 * k + 8] = tup(rs.uniform(30, 480), rs.uniform(30, 350), rs.uniform(15, 140), rs.uniform(10, 100),
                                         rs.uniform(0, 180), 1.0 if empty else 0.0)
    Yp = Yt.copy()
    Yp[:, 0::8] += rs.randn(N, 72) * 6
    Yp[:, 1::8] += rs.randn(N, 72) * 6
    Yp[:, 2::8] *= 1 + rs.randn(N, 72) * 0.1
    Yp[:, 6::8] = np.clip(Yp[:, 6::8] + rs.randn(N, 72) * 0.3, 0, 1)
    Yp[0, 2] = -3.0                                  # a degenerate prediction (a <= 0): empty raster
    host = D._pair_ious(Yp, Yt)
    dev = D._pair_ious_device(Yp, Yt)
    assert len(host) == len(dev) > 50
    h, d = np.array([p[0] for p in host]), np.array([p[0] for p in dev])
    assert np.abs(h - d).max() < 2e-3, np.abs(h - d).max()
    assert [(p[1], p[2]) for p in host] == [(p[1], p[2]) for p in dev]
    m_host, m_dev = D.calc_map(Yp, Yt), D.calc_map(Yp, Yt, device=True)
    assert abs(m_host - m_dev) < 0.02 and 0.0 < m_dev < 1.0
    assert D.calc_map(Yt, Yt, device=True) == 1.0


def test_setup_model_checkpoint_semantics_and_whole_model_file(data, tmp_path, monkeypatch):
    """setup_model (models.py:461-507 of the reference): missing checkpoint -> fresh start or plain Exception
    with no_cp_fatal; present -> loaded.  Model.save / load_model round trip of the 'whole model' file."""
    from spnet_amd import models as M
    X, Y = data
    monkeypatch.chdir(tmp_path)
    with pytest.raises(Exception, match="No weights file"):
        M.setup_model(X, try_checkpoint=True, no_cp_fatal=True, weights_file="missing.hdf5", freeze_fac=0.0)
    model, serial = M.setup_model(X, try_checkpoint=True, weights_file="missing.hdf5", freeze_fac=0.0)
    assert model is serial and model.optimizer.lr == pytest.approx(1e-5)
    model.optimizer.lr = 1e-3
    model.fit(X, Y, batch_size=8, epochs=1, shuffle=False, verbose=0)
    want = model.predict(X, batch_size=8)
    model.save_weights("weights.hdf5")
    again, _ = M.setup_model(X, try_checkpoint=True, no_cp_fatal=True, weights_file="weights.hdf5", freeze_fac=0.0)
    np.testing.assert_array_equal(again.predict(X, batch_size=8), want)
    model.save("full_model.h5")
    whole = M.load_model("full_model.h5", custom_objects={"custom_loss": M.custom_loss})
    np.testing.assert_array_equal(whole.predict(X, batch_size=8), want)
    assert whole.count_params() == model.count_params()


def test_inference_coefficients_follow_weight_changes(data):
    """The predict plan caches its BatchNorm scale|shift between batches; every way of changing the weights or
    the moving statistics (fit, load_weights, set_weights) must invalidate that cache."""
    from spnet_amd import models as M
    X, Y = data
    model = M.build_model(X, Y0size=576, freeze_fac=0.0)
    p0 = model.predict(X, batch_size=8)
    np.testing.assert_array_equal(model.predict(X, batch_size=8), p0)       # cached coefficients, same result
    model.optimizer.lr = 1e-3
    model.fit(X, Y, batch_size=8, epochs=1, shuffle=False, verbose=0)       # weights AND moving statistics move
    p1 = model.predict(X, batch_size=8)
    assert np.abs(p1 - p0).max() > 1e-6
    fresh = M.build_model(X, Y0size=576, freeze_fac=0.0)
    fresh.set_weights(model.get_weights())
    np.testing.assert_array_equal(fresh.predict(X, batch_size=8), p1)       # a plan without any cache agrees
    w = model.get_weights()
    fresh.predict(X, batch_size=8)
    fresh.set_weights([a * 0.5 for a in w])
    q = fresh.predict(X, batch_size=8)
    ref = M.build_model(X, Y0size=576, freeze_fac=0.0)
    ref.set_weights([a * 0.5 for a in w])
    np.testing.assert_array_equal(q, ref.predict(X, batch_size=8))


def test_compound_head_model(tmp_path):
    """model_type 'compound' (models.py:379-386): sigmoid 'noobj' outputs through SigmoidOutput + InterleaveColumns.
    Forward and every gradient against the oracle's restatement, the reference's two head layers in the checkpoint,
    and the whole-model file brings the head variant back."""
    import torch
    from oracle import torch_ref as T
    from spnet_amd import config as cf
    from spnet_amd import models as M
    from tests.parity_util import assert_forward_mse, assert_gradients_match, assert_output_close, make_case, rel_err
    H, W, B = 96, 128, 2
    P, X, Y, mask, dseed = make_case(H, W, B, 1)
    old = cf.model_type
    try:
        cf.model_type = 'compound'
        model = M.create_model_functional(X.numpy(), Y0size=576, freeze_fac=0.0)
    finally:
        cf.model_type = old
    sd = model.state_dict()
    assert "FinalOutput/kernel" not in sd and tuple(sd["SigmoidOutput/kernel"].shape)[1] == 72
    assert tuple(sd["DenseOutput/kernel"].shape)[1] == 504 and tuple(sd["DenseOutput/bias"].shape) == (504,)
    model._root.load_state_dict(P)                      # engine-level names (one FinalOutput kernel in final column order)
    sc = (cf.ind_noobj, cf.vars_per_pred)
    want = T.forward(P, X, training=False, sigmoid_cols=sc).numpy()
    got = model.predict(X.numpy(), batch_size=B)
    assert_forward_mse(got, want)
    assert got[:, 6::8].min() > 0 and got[:, 6::8].max() < 1      # sigmoid columns where InterleaveColumns puts them
    # the interleaving itself: SigmoidOutput column p is output column 6 + 8p
    np.testing.assert_array_equal(model.state_dict()["SigmoidOutput/kernel"].numpy(), P["FinalOutput/kernel"].numpy()[:, 6::8])
    eng = model._engine(B, train=True)
    eng.set_drop_seed(dseed)
    out = eng.forward(X.cuda(), training=True)
    eng.loss(Y.cuda())
    eng.backward()
    torch.cuda.synchronize()
    _, yp64, _, _ = assert_gradients_match(eng, P, X, Y, mask, sigmoid_cols=sc)
    assert_output_close(out.cpu().numpy(), yp64.numpy())
    # checkpoint round trips keep the head variant
    path = str(tmp_path / "full_model.h5")
    model.save(path)
    now = model.predict(X.numpy(), batch_size=B)          # (the training forward above moved the BatchNorm statistics)
    again = M.load_model(path)
    assert again.compound and "SigmoidOutput/kernel" in again.state_dict()
    np.testing.assert_array_equal(again.predict(X.numpy(), batch_size=B), now)
    w = model.get_weights()
    again.set_weights([a * 0 for a in w])
    z = again.predict(X.numpy(), batch_size=B)
    assert np.allclose(z[:, 6::8], 0.5) and float(np.abs(np.delete(z, np.s_[6::8], axis=1)).max()) < 1e-3


def test_uint8_frames_are_scaled_on_the_device_bit_for_bit():
    """spnet_u8_to_input == the host codec's scaling (utils.py:340-342: /255, -0.5, *2 on float32), for every grey
    level and for lengths with a scalar tail; Model.predict over uint8 frames (a quarter of the PCIe bytes) ==
    Model.predict over the host-converted float frames, bit for bit, including a ragged last batch."""
    import torch
    from spnet_amd import _lib as L
    from spnet_amd import fake_espi as F
    from spnet_amd import models as M
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    rs = np.random.RandomState(9)
    for n in (256, 16, 5, 331 * 331 * 3, 4099):
        u = np.arange(n, dtype=np.int64) % 256 if n == 256 else rs.randint(0, 256, n)
        u = u.astype(np.uint8)
        want = F.to_network_input(u.reshape(1, 1, n))[0, 0, :, 0]
        ud = torch.from_numpy(u).cuda()
        out = torch.full((n,), float("nan"), device="cuda")
        L.spnet_u8_to_input(ud.data_ptr(), out.data_ptr(), n, torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(out.cpu().numpy(), want)
    H, W = 64, 96
    U = rs.randint(0, 256, (11, H, W)).astype(np.uint8)
    model = M.Model((H, W, 1), Y0size=576, seed=5)
    y_f = model.predict(F.to_network_input(U), batch_size=4)
    y_u = model.predict(U, batch_size=4)
    y_u2 = model.predict_u8(U[..., None], batch_size=4)
    assert np.array_equal(y_f, y_u) and np.array_equal(y_u, y_u2)
    # the same frames as a device-RESIDENT uint8 tensor take the same scaling (one meaning of uint8 for every container)
    y_r = model.predict(torch.from_numpy(U).cuda(), batch_size=4)
    assert np.array_equal(y_r, y_u)
    with pytest.raises(TypeError):
        model.predict_u8(U.astype(np.float32))


def test_config_selects_the_exact_gemm_chain():
    """cf.pointwise_gemm (additive): 'bf16x3' (default) or 'f32' = every GEMM on the k-ordered fp32 MFMA chain; the model's
    plans are built accordingly and both predict the same frames to fp32 accuracy."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import config as cf
    from spnet_amd import models as M
    rs = np.random.RandomState(2)
    X = (rs.rand(4, 96, 128, 1).astype(np.float32) * 2 - 1)
    old = cf.pointwise_gemm
    try:
        preds = {}
        for mode in ("bf16x3", "f32"):
            cf.pointwise_gemm = mode
            m = M.Model((96, 128, 1), Y0size=576, seed=7)
            assert m._root.pointwise == mode
            preds[mode] = m.predict(X, batch_size=4)
    finally:
        cf.pointwise_gemm = old
    scale = float(np.abs(preds["f32"]).max())
    np.testing.assert_allclose(preds["bf16x3"], preds["f32"], rtol=1e-4, atol=1e-5 * scale)
