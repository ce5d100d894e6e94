"""Device fake-ESPI generator (csrc/espi.hip + the host parameter draw of spnet_amd/fake_espi.py) against
oracle/espi_ref.py, the CPU restatement of gen_fake_espi.py:60-279 that tests/test_oracle_numpy.py pins to the reference's
own draws: parameters and labels exactly; pixels statistically (OpenCV's, the restatement's and the device's rasterisers
draw outlines a fraction of a pixel apart); the sensor model by its distribution."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_device_generator_matches_the_restated_reference_generator():
    _need_gpu()
    from oracle import espi_ref as E
    from spnet_amd import fake_espi as F
    n, seed = 12, 5
    seeds = F.frame_seeds(n, seed)
    # 1. parameters and labels: the product's draw == the restatement's, generator pair by generator pair
    ref = [E.frame_params(random.Random(s), np.random.RandomState(s)) for s in seeds]
    Xc, labels_d, Uc = F.generate_device(n, seed=seed, noise=False, want_u8=True)
    for k, s in enumerate(seeds):
        waves, nodes, _ = F.draw_params(s)
        rw, rows, calls = ref[k]
        assert tuple(waves) == tuple(rw[:5])
        assert [tuple(nd[:6]) for nd in nodes] == [tuple(r) for r in rows] == [tuple(r) for r in labels_d[k]]
        # ring_start of every antinode = the colour of its innermost ring in the restatement's call list
        j = 0
        for nd, r in zip(nodes, rows):
            assert (calls[j][3] == E.BLACK) == (nd[6] == 0)
            j += max(2 * r[5], 1)
    # 2. noise-free canvas: the analytic device raster against the restatement's numpy raster (and the host PIL raster)
    dev_canvas = Uc.cpu().numpy()
    ref_canvas = np.stack([E.raster(r[0], r[2]) for r in ref])
    host_canvas = np.stack([F.raster_host(*F.draw_params(s)[:2]) for s in seeds])
    assert set(np.unique(dev_canvas)) <= {0, 128, 138}
    for name, other in (("device vs restatement", dev_canvas), ("host PIL vs restatement", host_canvas)):
        differ = float((other != ref_canvas).mean())
        print("%s: canvas pixels that differ: %.2f %%" % (name, 100 * differ))
        assert differ < 0.06, name                  # outline edges only
        for v in (0, 128, 138):                     # the same area of black bands / grey canvas / bright rings
            assert abs(float((other == v).mean()) - float((ref_canvas == v).mean())) < 0.02, (name, v)
    np.testing.assert_allclose(Xc.cpu().numpy()[..., 0], (dev_canvas.astype(np.float32) / 255 - 0.5) * 2, atol=1e-6)
    # 3. full sensor model: clipped N(40,40) noise + 50 % dropout, against the restatement's
    Xd, _, Ud = F.generate_device(n, seed=seed, noise=True, want_u8=True)
    ud = Ud.cpu().numpy().astype(np.float64)
    ur = np.stack([E.sensor(ref_canvas[k], random.Random(s + 1), np.random.RandomState(s + 1)) for k, s in enumerate(seeds)]).astype(np.float64)
    assert abs(float((ud == 0).mean()) - float((ur == 0).mean())) < 0.01          # dropout + black bands
    kept_d, kept_r = ud[ud > 0], ur[ur > 0]
    assert abs(kept_d.mean() - kept_r.mean()) < 2.0 and abs(kept_d.std() - kept_r.std()) < 2.0
    bg = (ref_canvas == 128) & (dev_canvas == 128)    # flat background in both: 128 + clip(rint(N(40,40)), 0, 255), saturated
    for arr, name in ((ud, "device"), (ur, "restatement")):
        px = arr[bg & (arr > 0)]
        assert abs(px.mean() - 171.1) < 1.0 and abs(px.std() - 34.1) < 1.0, (name, px.mean(), px.std())
    # the host generator (PIL raster + numpy sensor model) shares the labels and the statistics
    Xh, labels_h = F.generate(n, seed=seed, workers=1)
    assert labels_h == labels_d
    assert abs(float((Xh == 0).mean()) - float((ur == 0).mean())) < 0.01
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
