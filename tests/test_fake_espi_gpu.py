"""Device fake-ESPI generator (csrc/espi.hip) against the host generator (spnet_amd/fake_espi.py, PIL):
same parameters and labels by construction; pixels statistically (two rasterisers draw outlines a fraction of a pixel
apart), sensor model by its distribution."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_device_generator_matches_host_generator():
    _need_gpu()
    from spnet_amd import fake_espi as F
    n, seed = 24, 5
    Xh, labels_h = F.generate(n, seed=seed, workers=1)
    # noise-free canvas: analytic device raster vs PIL raster of the same parameters
    Xc, labels_d, Uc = F.generate_device(n, seed=seed, noise=False, want_u8=True)
    assert labels_d == labels_h
    host_canvas = np.stack([F.raster_host(*F.draw_params(s)[:2]) for s in F.frame_seeds(n, seed)])
    dev_canvas = Uc.cpu().numpy()
    assert set(np.unique(dev_canvas)) <= {0, 128, 138}
    differ = float((dev_canvas != host_canvas).mean())
    print("canvas pixels that differ between the two rasterisers: %.2f %%" % (100 * differ))
    assert differ < 0.06                      # outline edges only
    for v in (0, 128, 138):                   # the same area of black bands / grey canvas / bright rings
        assert abs(float((dev_canvas == v).mean()) - float((host_canvas == v).mean())) < 0.02, v
    np.testing.assert_allclose(Xc.cpu().numpy()[..., 0], (dev_canvas.astype(np.float32) / 255 - 0.5) * 2, atol=1e-6)
    # full sensor model: clipped N(40,40) noise + 50 % dropout
    Xd, _, Ud = F.generate_device(n, seed=seed, noise=True, want_u8=True)
    ud = Ud.cpu().numpy().astype(np.float64)
    assert abs(float((ud == 0).mean()) - float((Xh == 0).mean())) < 0.01          # dropout + black bands
    kept_d, kept_h = ud[ud > 0], Xh[Xh > 0].astype(np.float64)
    assert abs(kept_d.mean() - kept_h.mean()) < 2.0 and abs(kept_d.std() - kept_h.std()) < 2.0
    bg = (host_canvas == 128) & (dev_canvas == 128)    # flat background in both: 128 + clip(rint(N(40,40)), 0, 255), saturated
    for arr, name in ((ud, "device"), (Xh.astype(np.float64), "host")):
        px = arr[bg & (arr > 0)]
        assert abs(px.mean() - 171.1) < 1.0 and abs(px.std() - 34.1) < 1.0, (name, px.mean(), px.std())
    # determinism
    Xd2, _ = F.generate_device(n, seed=seed, noise=True)
    assert torch.equal(Xd, Xd2)
    Xd3, _ = F.generate_device(n, seed=seed + 1, noise=True)
    assert not torch.equal(Xd, Xd3)


def test_device_frames_train_the_network():
    """The frames + labels feed the engine like the host ones (grid encoding, one step, finite loss)."""
    _need_gpu()
    from bench import labels_to_Y
    from spnet_amd import fake_espi as F
    from spnet_amd.engine import Engine
    X, labels = F.generate_device(8, seed=2)
    Y = torch.from_numpy(labels_to_Y(labels)).cuda()
    assert X.shape == (8, 384, 512, 1) and float(X.min()) >= -1 and float(X.max()) <= 1
    eng = Engine(384, 512, 8, device="cuda:0", seed=0)
    out = eng.train_step(X, Y, 1e-4)
    torch.cuda.synchronize()
    assert np.isfinite(float(out[5]))
