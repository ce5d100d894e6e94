"""Inception-ResNet-v2 backbone (BASELINE configs[3]; cf.basemodel = 'InceptionResNetV2', spnet/models.py:357-359):
structure known-answers, the general-convolution building blocks, whole-network parity with the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import torch_ref as T
from tests.parity_util import assert_forward_mse, assert_gradients_match, assert_output_close, make_case, rel_err


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_structure_matches_keras_inception_resnet_v2():
    _need_gpu()
    from spnet_amd.engine import irv2_out_hw, irv2_program, param_specs
    # the product's program and the oracle's restatement are written independently: same ops, names and shapes
    got = [o[:9] + ("same" if o[9] else "valid",) + o[10:] if o[0] == "conv" else o for o in irv2_program()]
    assert got == T.irv2_layers()
    specs = param_specs(598, 598, backbone="InceptionResNetV2")
    body = [s for s in specs if not s[0].startswith(("conv2d_1/", "conv2d_2/", "conv2d_3/", "batch_normalization_1/",
                                                     "batch_normalization_2/", "batch_normalization_3/", "FinalOutput"))]
    total = sum(int(np.prod(s[1])) for s in body)
    trainable = sum(int(np.prod(s[1])) for s in body if s[2])
    # keras.applications.InceptionResNetV2(include_top=False): 54,336,736 parameters, 54,276,192 trainable
    assert (total, trainable, total - trainable) == (54336736, 54276192, 60544)
    assert irv2_out_hw(598, 598) == (8, 8)                # 299x299 behind the stem -> the canonical 8x8x1536
    assert sum(1 for o in irv2_program() if o[0] == "conv") == 244


@pytest.mark.parametrize("B,H,W,C,kh,kw,stride,same", [(2, 9, 11, 32, 3, 3, 1, 1), (2, 10, 14, 128, 1, 7, 1, 1),
                                                       (2, 10, 14, 160, 7, 1, 1, 1), (1, 21, 29, 48, 5, 5, 1, 1),
                                                       (2, 21, 29, 320, 3, 3, 2, 0), (3, 13, 12, 64, 3, 3, 1, 0),
                                                       (2, 6, 4, 192, 1, 3, 1, 1), (2, 6, 4, 224, 3, 1, 1, 1),
                                                       (2, 21, 29, 32, 3, 3, 1, 1), (1, 9, 11, 288, 3, 3, 2, 0)])
def test_general_convolution_through_patches_and_gemm(B, H, W, C, kh, kw, stride, same):
    _need_gpu()
    from spnet_amd import _lib as L
    st = torch.cuda.current_stream().cuda_stream
    rs = np.random.RandomState(C + kh * 7 + kw)
    cout = {288: 48, 160: 96, 224: 160, 48: 36}.get(C, 64)       # (also widths that are not whole 64-column tiles)
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float64, requires_grad=True)
    w = torch.tensor(rs.randn(kh, kw, C, cout) / np.sqrt(kh * kw * C), dtype=torch.float64, requires_grad=True)
    y = T.conv2d_general(x, w, stride, "same" if same else "valid")
    dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float64)
    y.backward(dy)
    M, K = y.shape[0] * y.shape[1] * y.shape[2], kh * kw * C
    xd, wd, dyd = x.detach().float().cuda(), w.detach().float().cuda(), dy.float().cuda().contiguous()
    col = torch.full((M, K), float("nan"), device="cuda")
    L.spnet_patches(xd.data_ptr(), col.data_ptr(), B, H, W, C, kh, kw, stride, same, 0, st)
    ws = torch.empty(1 << 22, device="cuda")
    yd = torch.empty(M, cout, device="cuda")
    L.spnet_gemm_f32(col.data_ptr(), 0, K, wd.data_ptr(), 1, cout, yd.data_ptr(), cout, M, cout, K, 0, ws.data_ptr(),
                     ws.numel(), None, 0, st)
    np.testing.assert_allclose(yd.cpu().numpy().reshape(y.shape), y.detach().numpy(), rtol=1e-4, atol=1e-4)
    dcol = torch.empty(M, K, device="cuda")
    L.spnet_gemm_f32(dyd.data_ptr(), 0, cout, wd.data_ptr(), 0, cout, dcol.data_ptr(), K, M, K, cout, 0, ws.data_ptr(),
                     ws.numel(), None, 0, st)
    dx = torch.full((B, H, W, C), float("nan"), device="cuda")
    L.spnet_patches(dcol.data_ptr(), dx.data_ptr(), B, H, W, C, kh, kw, stride, same, 1, st)
    np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), rtol=1e-4, atol=1e-4)
    gw = torch.empty(K, cout, device="cuda")
    L.spnet_gemm_f32(col.data_ptr(), 1, K, dyd.data_ptr(), 1, cout, gw.data_ptr(), cout, K, cout, M, 0, ws.data_ptr(),
                     ws.numel(), None, 0, st)
    np.testing.assert_allclose(gw.cpu().numpy().reshape(w.shape), w.grad.numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(M))
    # the forward convolution on the tuned GEMM kernel itself (gathered A tiles, spnet_conv_gemm_f32): every tile id, bit-identical to
    # patches + GEMM on that tile, statistics rows as the GEMM leaves them, bias, and an input that is a column block
    if C % 32 == 0:
        import ctypes
        wide = torch.tensor(rs.randn(B, H, W, C + 24), dtype=torch.float32).cuda()
        wide[..., 8:8 + C] = xd
        for tile in (0, 1, 2, 3, 5, 6, 7, 8, 9, 10):
            yt = torch.empty(M, cout, device="cuda")
            cst = torch.full(((M + 31) // 32 * 2 * cout,), float("nan"), device="cuda")
            nr = ctypes.c_int(0)
            L.spnet_gemm_f32_colstats(col.data_ptr(), 0, K, wd.data_ptr(), 1, cout, yt.data_ptr(), cout, M, cout, K, tile,
                                      cst.data_ptr(), ctypes.addressof(nr), st)
            yg = torch.full((M, cout), float("nan"), device="cuda")
            csg = torch.full_like(cst, float("nan"))
            ng = ctypes.c_int(0)
            L.spnet_conv_gemm_f32(xd.data_ptr(), C, wd.data_ptr(), yg.data_ptr(), cout, B, H, W, C, cout, kh, kw, stride, same,
                                  None, tile, csg.data_ptr(), ctypes.addressof(ng), st)
            assert torch.equal(yg, yt), tile
            assert ng.value == nr.value and torch.equal(csg[:ng.value * 2 * cout], cst[:nr.value * 2 * cout]), tile
            yw = torch.full((M, cout), float("nan"), device="cuda")
            L.spnet_conv_gemm_f32(wide.data_ptr() + 32, C + 24, wd.data_ptr(), yw.data_ptr(), cout, B, H, W, C, cout, kh, kw,
                                  stride, same, None, tile, None, None, st)
            assert torch.equal(yw, yt), tile
        bias2 = torch.tensor(rs.randn(cout), dtype=torch.float32).cuda()
        yb2 = torch.empty(M, cout, device="cuda")
        L.spnet_gemm_f32(col.data_ptr(), 0, K, wd.data_ptr(), 1, cout, yt.data_ptr(), cout, M, cout, K, 1, None, 0,
                         bias2.data_ptr(), 9, st)
        L.spnet_conv_gemm_f32(xd.data_ptr(), C, wd.data_ptr(), yb2.data_ptr(), cout, B, H, W, C, cout, kh, kw, stride, same,
                              bias2.data_ptr(), 9, None, None, st)
        assert torch.equal(yb2, yt)
        with pytest.raises(L.HipError):                                                      # C % 32 != 0 is refused
            L.spnet_conv_gemm_f32(xd.data_ptr(), C, wd.data_ptr(), yb2.data_ptr(), cout, B, H, W, C - 16, cout, kh, kw,
                                  stride, same, None, 0, None, None, st)
    # the adjoint gather that also masks with the producer's ReLU and leaves its BatchNorm-backward sums
    rows = int(L.spnet_grad_bnsums_rows(B * H * W, 512))
    assert 1 <= rows <= 512
    yprod = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32).cuda()           # producer output (sign = ReLU mask)
    ypre = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32).cuda()            # its pre-normalisation tensor
    mu, istd = torch.tensor(rs.randn(C) * 0.2, dtype=torch.float32).cuda(), torch.tensor(rs.rand(C) + 0.5, dtype=torch.float32).cuda()
    for relu in (1, 0):
        part = torch.full((rows, 2, C), float("nan"), device="cuda")
        dx2 = torch.full((B, H, W, C), float("nan"), device="cuda")
        L.spnet_patches_bwd_bnsums(dcol.data_ptr(), dx2.data_ptr(), B, H, W, C, kh, kw, stride, same, yprod.data_ptr(),
                                   ypre.data_ptr(), mu.data_ptr(), istd.data_ptr(), relu, part.data_ptr(), rows, st)
        want = dx * (yprod > 0) if relu else dx
        assert torch.equal(dx2, want)
        g64 = want.cpu().double().reshape(-1, C)
        xh = (ypre.cpu().double().reshape(-1, C) - mu.cpu().double()) * istd.cpu().double()
        sums = part.sum(0).cpu().double().numpy()
        np.testing.assert_allclose(sums[0], g64.sum(0).numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W) * float(g64.abs().max()))
        np.testing.assert_allclose(sums[1], (g64 * xh).sum(0).numpy(), rtol=1e-4,
                                   atol=3e-4 * np.sqrt(B * H * W) * float(g64.abs().max()))
    # ... and the Concatenate-backward form: a strided column block -> dense, same mask and sums
    cat = torch.tensor(rs.randn(B * H * W, C + 32), dtype=torch.float32).cuda()
    d = torch.full((B * H * W, C), float("nan"), device="cuda")
    part = torch.full((rows, 2, C), float("nan"), device="cuda")
    L.spnet_copy_cols_bnsums(cat.data_ptr() + 4 * 32, C + 32, d.data_ptr(), B * H * W, C, yprod.data_ptr(), C, ypre.data_ptr(),
                             mu.data_ptr(), istd.data_ptr(), 1, part.data_ptr(), rows, st)
    want = cat[:, 32:] * (yprod.reshape(-1, C) > 0)
    assert torch.equal(d, want)
    # the mask tensor may itself be a column block of the concatenated activation (row stride C + 32)
    ywide = torch.tensor(rs.randn(B * H * W, C + 32), dtype=torch.float32).cuda()
    ywide[:, 32:] = yprod.reshape(-1, C)
    d2, part2 = torch.full_like(d, float("nan")), torch.full_like(part, float("nan"))
    L.spnet_copy_cols_bnsums(cat.data_ptr() + 4 * 32, C + 32, d2.data_ptr(), B * H * W, C, ywide.data_ptr() + 4 * 32, C + 32,
                             ypre.data_ptr(), mu.data_ptr(), istd.data_ptr(), 1, part2.data_ptr(), rows, st)
    assert torch.equal(d2, d) and torch.equal(part2, part)
    g64 = want.cpu().double()
    xh = (ypre.cpu().double().reshape(-1, C) - mu.cpu().double()) * istd.cpu().double()
    sums = part.sum(0).cpu().double().numpy()
    np.testing.assert_allclose(sums[0], g64.sum(0).numpy(), rtol=1e-4, atol=1e-4 * np.sqrt(B * H * W) * 4)
    np.testing.assert_allclose(sums[1], (g64 * xh).sum(0).numpy(), rtol=1e-4, atol=3e-4 * np.sqrt(B * H * W) * 4)


def test_strided_patch_gather_and_batched_column_copies():
    """spnet_patches_ld on a column block of a wider tensor == spnet_patches on the dense copy; spnet_copy_cols_batched ==
    the copies one by one (the per-step gather of sibling kernels into their concatenated GEMM operand)."""
    _need_gpu()
    from spnet_amd import _lib as L
    st = torch.cuda.current_stream().cuda_stream
    rs = np.random.RandomState(4)
    B, H, W, C, Ct = 2, 9, 7, 32, 96
    wide = torch.tensor(rs.randn(B, H, W, Ct), dtype=torch.float32).cuda()
    dense = wide[..., 32:64].contiguous()
    for kh, kw, stride, same in ((3, 3, 1, 1), (1, 7, 1, 1), (3, 3, 2, 0)):
        OH = (H + stride - 1) // stride if same else (H - kh) // stride + 1
        OW = (W + stride - 1) // stride if same else (W - kw) // stride + 1
        a = torch.full((B * OH * OW, kh * kw * C), float("nan"), device="cuda")
        b = torch.full_like(a, float("nan"))
        L.spnet_patches(dense.data_ptr(), a.data_ptr(), B, H, W, C, kh, kw, stride, same, 0, st)
        L.spnet_patches_ld(wide.data_ptr() + 4 * 32, Ct, b.data_ptr(), B, H, W, C, kh, kw, stride, same, st)
        assert torch.equal(a, b)
    srcs = [torch.tensor(rs.randn(320, c), dtype=torch.float32).cuda() for c in (32, 48, 64)]
    dst = torch.full((320, 144), float("nan"), device="cuda")
    jobs, off = [], 0
    for t in srcs:
        jobs += [t.data_ptr(), dst.data_ptr() + 4 * off, 320, t.shape[1], t.shape[1], 144]
        off += t.shape[1]
    table = torch.tensor(jobs, dtype=torch.int64, device="cuda")
    L.spnet_copy_cols_batched(table.data_ptr(), 3, 320 * 64, st)
    assert torch.equal(dst, torch.cat(srcs, 1))


def test_irv2_pools_and_block_glue():
    _need_gpu()
    from spnet_amd import _lib as L
    st = torch.cuda.current_stream().cuda_stream
    rs = np.random.RandomState(4)
    B, H, W, C = 2, 21, 29, 64
    x = torch.tensor(rs.randn(B, H, W, C), dtype=torch.float64, requires_grad=True)
    for name, fn in (("max", T.maxpool3x3s2_valid), ("avg", T.avgpool3x3s1_same)):
        x.grad = None
        y = fn(x)
        dy = torch.tensor(rs.randn(*y.shape), dtype=torch.float64)
        y.backward(dy)
        xd, dyd = x.detach().float().cuda(), dy.float().cuda()
        yd, dxd = torch.empty(tuple(y.shape), device="cuda"), torch.empty(B, H, W, C, device="cuda")
        if name == "max":
            idx = torch.empty(y.numel() // 4, dtype=torch.int32, device="cuda")
            L.spnet_maxpool3x3s2_valid_fwd(xd.data_ptr(), yd.data_ptr(), idx.data_ptr(), B, H, W, C, st)
            L.spnet_maxpool3x3s2_valid_bwd(dyd.data_ptr(), idx.data_ptr(), dxd.data_ptr(), B, H, W, C, st)
        else:
            L.spnet_avgpool3x3s1_same(xd.data_ptr(), yd.data_ptr(), B, H, W, C, 0, st)
            L.spnet_avgpool3x3s1_same(dyd.data_ptr(), dxd.data_ptr(), B, H, W, C, 1, st)
        np.testing.assert_allclose(yd.cpu().numpy(), y.detach().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(dxd.cpu().numpy(), x.grad.numpy(), rtol=1e-5, atol=1e-6)
    # x + scale*up (+ReLU) and its gradient; channel-block copies with accumulation
    a, u, g = (torch.tensor(rs.randn(B, H, W, C), dtype=torch.float32).cuda() for _ in range(3))
    y = torch.empty_like(a)
    L.spnet_resadd(a.data_ptr(), u.data_ptr(), y.data_ptr(), a.numel(), 0.17, 1, st)
    want = torch.relu(a.double() + 0.17 * u.double())
    np.testing.assert_allclose(y.cpu().numpy(), want.cpu().numpy(), rtol=1e-6, atol=1e-6)
    dx, du = torch.empty_like(a), torch.empty_like(a)
    L.spnet_resadd_bwd(y.data_ptr(), g.data_ptr(), dx.data_ptr(), du.data_ptr(), a.numel(), 0.17, 1, st)
    gm = g * (y > 0)
    assert torch.equal(dx, gm) and torch.allclose(du, 0.17 * gm)
    cat = torch.zeros(B * H * W, 96, device="cuda")
    part = torch.tensor(rs.randn(B * H * W, 32), dtype=torch.float32).cuda()
    L.spnet_copy_cols(part.data_ptr(), 32, cat.data_ptr() + 4 * 64, 96, B * H * W, 32, 0, st)
    L.spnet_copy_cols(part.data_ptr(), 32, cat.data_ptr() + 4 * 64, 96, B * H * W, 32, 1, st)
    assert torch.equal(cat[:, 64:], 2 * part) and float(cat[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize("H,W,B,seed", [(288, 224, 2, 0), (235, 301, 3, 1)])
def test_irv2_forward_and_gradients(H, W, B, seed):
    """(Frames big enough that the last stage still has a dozen samples per channel: at 160x192 / batch 2 the 8x8-stage
    BatchNorms normalise over TWO samples, xhat = +-1 whatever the data, and nothing meaningful is left to compare.)"""
    _need_gpu()
    from spnet_amd.engine import Engine
    P, X, Y, mask, dseed = make_case(H, W, B, seed, basemodel="InceptionResNetV2")
    eng = Engine(H, W, B, device="cuda:0", seed=1, backbone="InceptionResNetV2")
    assert list(eng.state_dict().keys()) == list(P.keys())
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


def test_irv2_bench_geometry():
    """BASELINE configs[3] at the geometry bench.py runs it: Inception-ResNet-v2, 384x512 frames, batch 16.  The batch-16
    plans are the only ones that execute the in-step autotuned entries of spnet_amd/gemm_tiles.json (32x64 / 32x32 tiles,
    the gathered-A convolutions on them).  Inference: frames 0-1 against the oracle, all 16 against the batch-2 plan of
    the same weights (other tiles for most GEMMs: equal to rounding, not to the bit), captured replay == eager.
    Training (the fp64 oracle on the device's decisions checks every gradient at 288x224 / 235x301, batch 2-3; at this
    size it would take minutes): the batch-16 plan gives finite gradients, reproduces itself bit for bit from the same
    state, and reduces the loss over eight optimizer steps."""
    _need_gpu()
    from spnet_amd.engine import Engine, TILE_TABLE
    H, W = 384, 512
    P, X2, _, _, _ = make_case(H, W, 2, 5, basemodel="InceptionResNetV2")
    rs = np.random.RandomState(6)
    X = torch.tensor(rs.rand(16, H, W, 1) * 2 - 1, dtype=torch.float32)
    X[:2] = X2
    e16 = Engine(H, W, 16, device="cuda:0", seed=1, backbone="InceptionResNetV2", train=False)
    e16.load_state_dict(P)
    y16 = e16.forward(X.cuda(), training=False).cpu().clone()
    assert y16.shape == (16, 576) and bool(torch.isfinite(y16).all())
    want = T.forward(P, X2, training=False)
    assert_forward_mse(y16[:2], want)
    e2 = Engine(H, W, 2, device="cuda:0", seed=1, backbone="InceptionResNetV2", train=False)
    e2.load_state_dict(P)
    scale = float(y16.abs().max())
    for lo in range(0, 16, 2):
        y2 = e2.forward(X[lo:lo + 2].cuda(), training=False).cpu()
        np.testing.assert_allclose(y16[lo:lo + 2].numpy(), y2.numpy(), rtol=1e-4, atol=1e-5 * scale)
    # the batch-16 plan really runs autotuned tiles (the table is keyed by M = 16 x pixels: nothing at batch 2 hits it)
    net = [n for n in e16.nodes if hasattr(n, "ops")][0]
    Ms = {o.M for o in net.ops if hasattr(o, "M")}
    assert any(k[3] in Ms for k in TILE_TABLE), "no autotuned entry matches the batch-16 plan"
    # captured replay == eager, twice
    e16.x_in.copy_(X.cuda())
    a = e16.predict_step().clone()
    b = e16.predict_step().clone()
    assert torch.equal(a, b) and torch.equal(a.cpu(), y16)
    del e2, e16
    torch.cuda.empty_cache()
    # training plan at batch 16: finite loss / gradients, deterministic, and the loss falls over a few steps
    t16 = Engine(H, W, 16, device="cuda:0", seed=1, backbone="InceptionResNetV2")
    t16.load_state_dict(P)
    Y = torch.tensor(rs.rand(16, 576), dtype=torch.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    t16.set_drop_seed(11)
    t16.forward(X.cuda(), training=True)
    l0 = t16.loss(Y.cuda()).clone()
    t16.backward()
    torch.cuda.synchronize()
    g0 = t16.grad.clone()
    assert bool(torch.isfinite(g0).all()) and float(g0.abs().max()) > 0
    t16.load_state_dict(P)                      # moving statistics back to P: the second pass must reproduce the first
    t16.set_drop_seed(11)
    t16.forward(X.cuda(), training=True)
    l1 = t16.loss(Y.cuda()).clone()
    t16.backward()
    torch.cuda.synchronize()
    assert torch.equal(l0, l1) and torch.equal(g0, t16.grad)
    losses = [float(t16.train_step(X.cuda(), Y.cuda(), 1e-5)[5]) for _ in range(8)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_irv2_trains_through_the_model_api():
    """BASELINE configs[3] plumbing: cf.basemodel = 'InceptionResNetV2', batch 16 -- a few optimizer steps through
    Model.fit reduce the loss; predict is deterministic."""
    _need_gpu()
    from spnet_amd import config as cf
    from spnet_amd import models as M
    rs = np.random.RandomState(3)
    X = (rs.rand(16, 160, 192, 1).astype(np.float32) * 2 - 1)
    Y = rs.rand(16, 576).astype(np.float32)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5)
    old = cf.basemodel
    try:
        cf.basemodel = 'InceptionResNetV2'
        model = M.create_model_functional(X, Y0size=576, freeze_fac=0.0)
    finally:
        cf.basemodel = old
    assert model.basemodel == 'InceptionResNetV2'
    model.optimizer.lr = 3e-4
    h = model.fit(X, Y, batch_size=16, epochs=6, shuffle=False, verbose=0)
    assert np.all(np.isfinite(h["loss"])) and h["loss"][-1] < h["loss"][0]
    p1, p2 = model.predict(X, batch_size=16), model.predict(X, batch_size=16)
    assert p1.shape == (16, 576) and np.array_equal(p1, p2)


def test_config3_plumbing_run_with_inception_resnet_v2(tmp_path):
    """BASELINE configs[3]: Inception-ResNetV2 backbone, batch 16, through train_spnet.py (331x331 'monolithic' frames)."""
    import os
    import subprocess
    import sys
    _need_gpu()
    from spnet_amd import fake_espi as F
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = tmp_path / "data"
    F.write_dataset(str(data / "Train"), 32, seed=1)
    F.write_dataset(str(data / "Val"), 16, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    r = subprocess.run([sys.executable, os.path.join(root, "train_spnet.py"), "-d", str(data), "-b", "16", "-e", "2",
                        "--name", "ir", "--backbone", "InceptionResNetV2"], cwd=str(work), env=dict(os.environ, PYTHONPATH=root),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "cf.basemodel = InceptionResNetV2" in r.stdout and "SPNet execution completed." in r.stdout
    logs = [d for d in os.listdir(work / "logs") if d.startswith("ir_")]
    rows = [l for l in open(work / "logs" / logs[0] / "losses.dat") if not l.startswith("#")]
    assert len(rows) == 2 and all(np.isfinite(float(x.split()[1])) for x in rows)
