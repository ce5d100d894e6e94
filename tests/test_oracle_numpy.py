"""The numpy oracle against golden vectors produced by the reference's own functions
(tests/golden/make_goldens.py) and against the reference's own known-answer tests."""
import random

import numpy as np

from oracle import numpy_ref as R


def test_loss_matches_reference_my_loss(golden):
    for tag in ("loss", "loss2"):
        yt, yp = golden[f"{tag}_yt"], golden[f"{tag}_yp"]
        for lt in ("same", "hybrid"):
            tot, parts = R.loss_terms(yt, yp, lt)
            np.testing.assert_allclose(tot, golden[f"{tag}_{lt}_total"], rtol=2e-6)
            np.testing.assert_allclose(parts, golden[f"{tag}_{lt}_parts"], rtol=2e-6)


def test_survey_known_answers(golden):
    # my_loss on RandomState(0)
    np.testing.assert_allclose(golden["loss_same_total"], 0.14024234, rtol=1e-6)
    np.testing.assert_allclose(golden["loss_hybrid_total"], 0.1551701, rtol=1e-6)


def test_loss_grad_is_gradient_of_loss(golden):
    yt, yp = golden["loss2_yt"].astype(np.float64), golden["loss2_yp"].astype(np.float64)
    for lt in ("same", "hybrid"):
        g = R.loss_grad(yt, yp, lt)
        rs = np.random.RandomState(0)
        for _ in range(20):
            i, j = rs.randint(yp.shape[0]), rs.randint(yp.shape[1])
            h = 1e-5
            a, b = yp.copy(), yp.copy()
            a[i, j] += h
            b[i, j] -= h
            fd = (R.loss_terms(yt, a, lt)[0] - R.loss_terms(yt, b, lt)[0]) / (2 * h)
            np.testing.assert_allclose(g[i, j], fd, rtol=1e-5, atol=1e-10)


def test_grid_constants(golden):
    g = R.grid_constants([6, 6, 2, 8])
    np.testing.assert_array_equal(golden["mr_scalars"], [g["cx_min"], g["cy_min"], g["cx_max"], g["cy_max"], g["xbin"], g["ybin"]])
    np.testing.assert_array_equal(golden["mr_gridYi"], g["defaults"])
    np.testing.assert_array_equal(golden["means"], g["means"])
    np.testing.assert_array_equal(golden["ranges"], g["ranges"])
    np.testing.assert_array_equal(g["means"][:8], [75.5, 65.5, 35.5, 25.5, 0, 0, 0, 5])
    np.testing.assert_array_equal(g["ranges"][:8], [71, 51, 71, 51, 2, 2, 1, 10])


def test_encode_norm_denorm(golden):
    grid = R.encode_grid(golden["grid_in"], [6, 6, 2, 8])
    np.testing.assert_array_equal(grid, golden["grid_out"])
    Yn = R.norm_y(grid.reshape(1, -1))
    np.testing.assert_array_equal(Yn, golden["grid_norm"])
    np.testing.assert_array_equal(R.denorm_y(Yn), golden["grid_denorm"])


def test_encode_overflow_asserts():
    rows = [[100, 140, 30, 20, 1, 0, 0, 3]] * 3
    try:
        R.encode_grid(rows)
    except AssertionError:
        return
    raise AssertionError("expected slot-overflow assertion (utils.py:240)")


def test_encode_empty():
    g = R.grid_constants()
    np.testing.assert_array_equal(R.encode_grid(np.array([])), g["defaults"])


def test_cleanup_and_denorm8(golden):
    sub = R.denorm_y(np.tile(np.array([.25, -.1, .5, .2, .6, -.8, .2, .13], np.float32), 72)[None, :])[0, :8]
    np.testing.assert_array_equal(sub, golden["denorm8"])
    np.testing.assert_allclose(R.cleanup_vars(sub), golden["cleanup8"], rtol=0, atol=0)
    for c, want in zip(golden["cleanup_in"], golden["cleanup_out"]):
        np.testing.assert_allclose(R.cleanup_vars(c), want, rtol=0, atol=0)


def test_parse_meta(golden):
    out = np.asarray(R.parse_meta_text(str(golden["meta_csv"])), np.float64)
    np.testing.assert_array_equal(out, golden["meta_out"])


def test_reference_unit_kats(golden):
    # tests/test_utils.py:6-14 of the reference
    assert R.nearest_multiple(720, 31) == 713 == int(golden["nearest_multiple_720_31"])


def test_one_cycle(golden):
    lrs = R.one_cycle_table(4e-5, 40000, 100, 16)
    assert len(lrs) == int(golden["lrs_len"]) == 250000
    np.testing.assert_array_equal(lrs[golden["lrs_idx"]], golden["lrs_val"])
    # paper/run_logs/log_DatasetA...:207,297 (printed with 7 significant digits)
    assert "%.6e" % lrs[2499] == "2.879505e-06"
    assert "%.6e" % lrs[12499] == "7.999573e-06"
    np.testing.assert_array_equal(R.one_cycle_table(1e-3, 1000, 3, 8), golden["lrs2_full"])


def test_count_errors(golden):
    out = R.count_errors(golden["ce_Yp"], golden["ce_Yt"])
    np.testing.assert_array_equal(out[:7], golden["ce_counts"])
    np.testing.assert_array_equal(out[7], golden["ce_pix_err"])
    assert out[8] == int(golden["ce_ipem"])


def test_cleanup_angle(golden):
    np.testing.assert_array_equal([R.cleanup_angle(a) for a in golden["angle_in"]], golden["angle_out"])


def test_augment_matches_reference_rng_order(golden):
    for tag in ("a", "b"):
        shape = tuple(golden[f"aug_{tag}_shape"])
        X = (np.random.RandomState(5).rand(*shape).astype(np.float32) * 2 - 1)
        want = X.copy().ravel()
        want[golden[f"aug_{tag}_changed_idx"]] = golden[f"aug_{tag}_changed_val"]
        want = want.reshape(shape)
        np.random.seed(1234)
        random.seed(1234)
        got = np.stack([R.augment_image(X[i].copy()) for i in range(shape[0])])
        np.testing.assert_array_equal(got, want)
        assert np.random.rand() == float(golden[f"aug_{tag}_rng_after"])


def test_fake_espi_restatement_reproduces_the_reference_draws(golden):
    """oracle/espi_ref.py against the reference's own draw_waves / draw_antinodes (gen_fake_espi.py:60-206), run from
    seeded generators by tests/golden/make_goldens.py with their OpenCV calls recorded: the same wave polylines (every
    point handed to cv2.polylines), the same draw_ellipse calls (centre, ring axes, angle, colour, thickness, in order),
    the same CSV caption, and the same NEXT outputs of both generators (total RNG consumption)."""
    from oracle import espi_ref as E
    for s in (0, 1, 2, 3, 11):
        rnd, nprnd = random.Random(s), np.random.RandomState(s)
        waves, rows, calls = E.frame_params(rnd, nprnd)
        lines = E.wave_polylines(waves)
        assert lines.shape[0] == int(golden["espi%d_n_lines" % s]) and waves[2] == int(golden["espi%d_line_thickness" % s])
        np.testing.assert_array_equal(lines[0], golden["espi%d_line_first" % s])
        np.testing.assert_array_equal(lines[-1], golden["espi%d_line_last" % s])
        np.testing.assert_array_equal(lines[:, 0, 1], golden["espi%d_line_y0" % s])
        got = np.array([(c[0][0], c[0][1], c[1][0], c[1][1], c[2], c[3], c[4]) for c in calls], np.float64).reshape(-1, 7)
        np.testing.assert_array_equal(got, golden["espi%d_ellipses" % s])
        assert E.caption(rows) == str(golden["espi%d_caption" % s])
        np.testing.assert_array_equal(np.array([rnd.random(), nprnd.rand()]), golden["espi%d_rng_after" % s])
    # draw_ellipse's fixed-point conversion (spnet/utils.py:41-52)
    assert E.cv_ellipse_args((261, 157), [139 / 3.0, 22.0], 150) == ((267264, 160768), (47445, 22528), -150)


def test_fake_espi_restated_raster_and_sensor_model():
    """The numpy rasteriser and sensor model of the restatement on one frame: three grey levels, band / ring areas in the
    range the parameters imply, saturated N(40, 40) noise on the flat background (mean 171.1, sigma 34.1 after clipping at
    255), half of the pixels dropped."""
    from oracle import espi_ref as E
    rnd, nprnd = random.Random(5), np.random.RandomState(5)
    waves, rows, calls = E.frame_params(rnd, nprnd)
    img = E.raster(waves, calls)
    assert set(np.unique(img)) <= {0, 128, 138} and (img == 128).mean() > 0.2 and (img == 0).mean() > 0.05
    for cx, cy, a, b, ang, rings in rows:                   # the centre of every antinode lies inside its innermost ring
        assert img[cy, cx] in (0, 128, 138)
    out = E.sensor(img, rnd, nprnd)
    assert abs((out == 0).mean() - (0.5 + 0.5 * (img == 0).mean() * 0.16)) < 0.02     # mask + black pixels with noise 0
    bg = out[(img == 128) & (out > 0)].astype(np.float64)
    assert abs(bg.mean() - 171.1) < 1.0 and abs(bg.std() - 34.1) < 1.0
