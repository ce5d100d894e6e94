"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/spnet_hip.h
declares; the product's host logic (grid codec, schedule, metrics, augmentation parameter draws,
parameter layout) against the reference-generated golden vectors; the N>1 path on gloo."""
import os
import random
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port that is free right now on 127.0.0.1 (fixed rendezvous ports collide when two suites share a host)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])


def test_library_exports_every_declared_symbol():
    from spnet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "spnet_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(spnet_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 27
    assert sorted(_lib.EXPORTS) == declared          # the binding table and the header list the same ABI
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (spnet_[a-z0-9_]+)", out))
    assert set(declared) <= exported


def test_product_has_no_oracle_imports():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spnet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
    for f in ("train_spnet.py", "predict_spnet.py", "evaluate_spnet.py"):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert not re.search(r"^\s*(from|import)\s+oracle\b", open(p).read(), re.M), f


def test_grid_codec_against_reference_golden(golden):
    from spnet_amd import utils as U
    ret = U.setup_means_and_ranges([6, 6, 2, 8])
    np.testing.assert_array_equal(golden["mr_scalars"], ret[:6])
    np.testing.assert_array_equal(golden["mr_gridYi"], ret[6])
    np.testing.assert_array_equal(golden["means"], U.means)
    np.testing.assert_array_equal(golden["ranges"], U.ranges)
    grid = U.true_to_pred_grid(golden["grid_in"], np.array([6, 6, 2, 8]))
    np.testing.assert_array_equal(grid, golden["grid_out"])
    Yn = U.norm_Y(grid.flatten()[None, :])
    np.testing.assert_array_equal(Yn, golden["grid_norm"])
    np.testing.assert_array_equal(U.denorm_Y(Yn), golden["grid_denorm"])
    for c, want in zip(golden["cleanup_in"], golden["cleanup_out"]):
        np.testing.assert_allclose(U.cleanup_antinode_vars(c), want, rtol=0, atol=0)
    assert U.nearest_multiple(720, 31) == 713                       # reference tests/test_utils.py
    assert U.add_to_stack(None, 5) == [5] and U.add_to_stack([5], 5) == [5, 5]
    with pytest.raises(AssertionError):
        U.true_to_pred_grid(np.array([[100, 140, 30, 20, 1, 0, 0, 3]] * 3), [6, 6, 2, 8])


def test_parse_meta_file(golden, tmp_path):
    from spnet_amd import utils as U
    p = tmp_path / "m.csv"
    p.write_text(str(golden["meta_csv"]))
    np.testing.assert_array_equal(np.asarray(U.parse_meta_file(str(p)), np.float64), golden["meta_out"])


def test_one_cycle_schedule(golden):
    from spnet_amd.callbacks import OneCycleScheduler, get_1cycle_schedule
    lrs = get_1cycle_schedule(lr_max=4e-5, n_data_points=40000, epochs=100, batch_size=16)
    assert len(lrs) == int(golden["lrs_len"])
    np.testing.assert_array_equal(lrs[golden["lrs_idx"]], golden["lrs_val"])
    np.testing.assert_array_equal(get_1cycle_schedule(1e-3, 1000, 3, 8), golden["lrs2_full"])

    class M:
        class optimizer:
            lr = 0.0
    s = OneCycleScheduler(lr_max=1e-3, n_data_points=1000, epochs=3, batch_size=8)
    s.set_model(M)
    for i in range(5):
        s.on_batch_begin(i)
    assert M.optimizer.lr == golden["lrs2_full"][4] and s.iteration == 5


def test_calc_errors(golden):
    from spnet_amd import diagnostics as D
    out = D.calc_errors(golden["ce_Yp"], golden["ce_Yt"])
    np.testing.assert_array_equal(out[:7], golden["ce_counts"])
    np.testing.assert_array_equal(out[7], golden["ce_pix_err"])
    assert out[8] == int(golden["ce_ipem"])


def test_iou_properties():
    from spnet_amd import diagnostics as D
    a = np.array([100, 140, 60, 30, np.cos(np.deg2rad(60)), np.sin(np.deg2rad(60)), 0, 5.0])
    assert D.compute_iou(a, a) == 1.0
    b = a.copy()
    b[0] += 400
    assert D.compute_iou(b, a) == 0.0
    c = a.copy()
    c[6] = 1.0
    assert D.compute_iou(a, c) == -1          # nothing supposed to be there
    area = np.count_nonzero(D.create_ellipse_image(a))
    assert abs(area - np.pi * 60 * 30) / (np.pi * 60 * 30) < 0.01
    # reference tests/test_diagnostics.py pins 0.44227983 for this pair under cv2's rasteriser (old 7-tuple
    # API: angle in degrees); the analytic raster must land within a boundary-pixel ring of it
    def tup(cx, cy, a_, b_, ang):
        return np.array([cx, cy, a_, b_, np.cos(2 * np.deg2rad(ang)), np.sin(2 * np.deg2rad(ang)), 0, 1.0])
    iou = D.compute_iou(tup(100, 140, 120, 60, 90), tup(120, 123, 120, 60, 149.97))
    assert abs(iou - 0.44227983107795693) < 0.01, iou


def test_augmentation_draws_follow_reference_rng_order(golden):
    """The HOST half of the device augmentation: parameter draws applied with numpy must reproduce the
    reference's augmented frames (the device half is checked bit-exactly in the GPU tests)."""
    from spnet_amd import augmentation as A
    for tag in ("a", "b"):
        shape = tuple(int(v) for v in golden[f"aug_{tag}_shape"])
        X = (np.random.RandomState(5).rand(*shape).astype(np.float32) * 2 - 1)
        want = X.copy().ravel()
        want[golden[f"aug_{tag}_changed_idx"]] = golden[f"aug_{tag}_changed_val"]
        want = want.reshape(shape)
        np.random.seed(1234)
        random.seed(1234)
        for i in range(shape[0]):
            img = X[i]
            for r0, r1, c0, c1, v in A.draw_cutout(img.shape, np.min(img), np.max(img)):
                img[r0:r1, c0:c1, :] = v
            sp = A.draw_saltpepper(img.shape)
            if sp is not None:
                hi, lo = np.max(img), np.min(img)
                img[sp[0], sp[1], :] = hi
                img[sp[2], sp[3], :] = lo
            A.draw_blur_gate()
        np.testing.assert_array_equal(X, want)
        assert np.random.rand() == float(golden[f"aug_{tag}_rng_after"])
    np.testing.assert_array_equal([A.cleanup_angle(a) for a in golden["angle_in"]], golden["angle_out"])


def test_parameter_layout_and_layer_table():
    from spnet_amd.engine import backbone_out_hw, param_specs
    from spnet_amd.models import InterleaveColumns, SelectiveSigmoid, keras_layer_table
    specs = param_specs(331, 331)
    total = sum(int(np.prod(s[1])) for s in specs)
    trainable = sum(int(np.prod(s[1])) for s in specs if s[2])
    assert (total, trainable, total - trainable) == (50353481, 50298935, 54546)     # reference run log :94-101
    assert [s[0].split("/")[0] for s in specs if s[3]] == ["conv2d_1", "conv2d_2", "conv2d_3", "block1_conv1", "block1_conv2",
                                                          "conv2d_4", "conv2d_5", "conv2d_6", "conv2d_7", "FinalOutput"]
    assert backbone_out_hw(331, 331) == (5, 5) and backbone_out_hw(384, 512) == (6, 8)
    assert sum(int(np.prod(s[1])) for s in param_specs(384, 512)) == 77485385
    assert len(keras_layer_table()) == 144                                            # "Freezing 0 / 144 layers"
    x = np.arange(16, dtype=np.float32)[None, :]
    y = SelectiveSigmoid()(x)
    assert np.flatnonzero(y != x).tolist() == [6, 14]       # reference tests/test_selectivesigmoid.py expectation
    cf_v = 3        # the docstring example of the reference's InterleaveColumns (models.py:228-236), vars_per_pred = 3
    from spnet_amd import config as cf
    old = cf.vars_per_pred
    cf.vars_per_pred = cf_v
    try:
        z = InterleaveColumns(start_index=2)(np.array([[10, 11, 12, 1, 2, 3, 4, 5, 6]], np.float32))
    finally:
        cf.vars_per_pred = old
    assert z.tolist() == [[1, 2, 10, 3, 4, 11, 5, 6, 12]]


def test_fake_espi_generator_is_deterministic_and_encodable():
    from spnet_amd import fake_espi as F
    X1, l1 = F.generate(4, seed=3, workers=1)
    X2, l2 = F.generate(4, seed=3, workers=1)
    assert X1.shape == (4, 384, 512) and X1.dtype == np.uint8
    np.testing.assert_array_equal(X1, X2)
    assert l1 == l2 and all(1 <= len(r) <= 7 for r in l1)
    x = F.to_network_input(X1)
    assert x.shape == (4, 384, 512, 1) and x.min() >= -1 and x.max() <= 1


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from spnet_amd import parallel
rank, local, world = parallel.init_distributed("gloo")
n = 1000
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
red = parallel.GradReducer(g, (300, 700))
red.launch_head()
scale = red.finish()
want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
assert torch.equal(g, want), (rank, g[:5])
assert scale == 1.0 / world
# bucketed form: pieces launched as their trigger nodes complete (reverse layer order), tail at finish()
g = torch.arange(n, dtype=torch.float32) * (rank + 1)
head, blk, stem = object(), object(), object()
red = parallel.GradReducer(g, [(0, 300, head), (300, 500, head), (800, 1000, blk), (650, 800, stem)], tail=[(500, 650)])
for node in (head, blk, object(), stem):
    red.on_node_done(node)
assert red.launched == 4
assert red.finish() == 1.0 / world and torch.equal(g, want)
try:
    parallel.GradReducer(g, [(0, 300, head)], tail=[(400, 1000)])
    raise SystemExit("a gap between buckets must be rejected")
except ValueError:
    pass
s1 = parallel.sample_seed(1, 3, 17)
assert s1 == parallel.sample_seed(1, 3, 17) != parallel.sample_seed(1, 3, 18) and 0 <= s1 < 2 ** 31
idx = parallel.shard_indices(103, epoch=2, rank=rank, world=world, seed=1, batch_size=4)
allidx = [None] * world
dist.all_gather_object(allidx, idx.tolist())
flat = sum(allidx, [])
assert len(flat) == len(set(flat)) and len(idx) % 4 == 0 and len(idx) == (103 // world) // 4 * 4
assert abs(parallel.all_reduce_scalar_mean(float(rank)) - (world - 1) / 2) < 1e-12
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("world", [2])
def test_gradient_allreduce_and_sharding_on_gloo(tmp_path, world):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = []
    port = _free_port()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


_PLAN_WORKER = r'''
import os, sys, time, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
torch.set_num_threads(1)
from spnet_amd import parallel
from spnet_amd import engine as E
rank, local, world = parallel.init_distributed("gloo")
lay = E.param_layout(384, 512)
nodes = E.xception_node_pnames()
buckets, tail = E.plan_grad_buckets(lay["p_off"], lay["rest_lo"], lay["n_theta"], nodes)
n = lay["n_theta"]
pattern = (torch.arange(n, dtype=torch.int64) % 7).float()
g = pattern * (rank + 1)
red = parallel.GradReducer(g, buckets, tail=tail, force=True)
assert red.active and red.world == world
launched = []
orig = red._launch
def spy(lo, hi):
    launched.append((lo, hi))
    return orig(lo, hi)
red._launch = spy
for key, _, _ in reversed(nodes):          # Engine.backward walks the nodes in reverse and reports each one
    red.on_node_done(key)
n_before_tail = len(launched)
scale = red.finish()
assert scale == 1.0 / world
assert torch.equal(g, pattern * sum(r + 1 for r in range(world)))
assert [tuple(b[:2]) for b in buckets] == launched[:n_before_tail] and [tuple(t) for t in tail] == launched[n_before_tail:]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", len(buckets), len(tail))
'''


def test_allreduce_plan_of_the_benchmark_engine():
    """The gradient all-reduce plan of the 384 x 512 Xception engine (replaces
    spnet/multi_gpu.py:35-88), computed without a device: it tiles the 310 MB flat gradient exactly once, sends the Dense
    head first in ~32 MB pieces, then suffixes of the forward-ordered region in reverse layer order with cuts on both edges
    of the middle flow (whose 24 weight gradients come out of one deferred launch, triggered by block 5), and leaves only
    the small prefix (l2 kernels, depthwise kernels, stem / entry-flow parameters) for the end of backward."""
    from spnet_amd import engine as E
    lay = E.param_layout(384, 512)
    nodes = E.xception_node_pnames()
    assert sum(n for _, n, _ in lay["p_off"].values()) == 77485385 - 54546          # trainable parameters
    buckets, tail = E.plan_grad_buckets(lay["p_off"], lay["rest_lo"], lay["n_theta"], nodes)
    per = (32 << 20) // 4
    cover = sorted([(lo, hi) for lo, hi, _ in buckets] + list(tail))
    assert cover[0][0] == 0 and cover[-1][1] == lay["n_theta"] and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    hoff, hn, _ = lay["p_off"]["FinalOutput/kernel"]
    head = [b for b in buckets if b[2] == "FinalOutput"]
    assert hoff == 0 and hn == 98304 * 576 and len(head) == -(-hn // per) and buckets[:len(head)] == head
    assert all(hi - lo <= per for lo, hi, _ in head) and head[0][0] == 0 and head[-1][1] == hn
    rest = buckets[len(head):]
    assert all(a[0] == b[1] for a, b in zip(rest, rest[1:]))                 # descending, contiguous suffixes
    assert rest[0][1] == lay["n_theta"]
    order = [k for k, _, _ in nodes]
    owner = {pn: k for k, pns, _ in nodes for pn in pns}
    trig = [order.index(t) for _, _, t in rest]
    assert trig == sorted(trig, reverse=True)                                # fired in backward order
    mid = [b for b in rest if b[2] == "block5"]
    names_in = lambda lo, hi: {n.split("/")[0] for n, (o, k, _) in lay["p_off"].items() if lo <= o < hi}
    for lo, hi, t in rest:          # a bucket is complete when its trigger is done: nothing in it belongs to an earlier node
        assert all(order.index(owner[nm]) >= order.index(t) for nm in names_in(lo, hi)), (lo, hi, t)
    mid_names = set().union(*[names_in(lo, hi) for lo, hi, _ in mid])
    assert mid and all(nm.startswith(("block5_", "block6_", "block7_", "block8_", "block9_", "block10_", "block11_", "block12_"))
                       for nm in mid_names) and len(mid_names) == 8 * 6
    assert sum(hi - lo for lo, hi in tail) * 4 < (16 << 20)                   # what waits for the end of backward: < 16 MB
    assert 4 * lay["n_theta"] == 309739008                                    # bytes all-reduced per step and rank


@pytest.mark.parametrize("world", [8])
def test_allreduce_plan_runs_on_eight_gloo_ranks(tmp_path, world):
    """... and eight gloo ranks execute that plan on full-size gradient buffers: every bucket is launched by the node that
    completes it, in the planned order, the tail at finish(), and every element comes out as the sum over ranks."""
    script = tmp_path / "w8.py"
    script.write_text(_PLAN_WORKER)
    procs = []
    port = _free_port()
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_prediction_csv_matches_reference_writer(golden, tmp_path):
    """hawley_spnet.csv: byte-for-byte the text the reference's own show_pred_ellipses wrote for the same
    de-normalised grids (utils.py:67-137; golden generated with the image / drawing calls stubbed out)."""
    from PIL import Image
    from spnet_amd import utils as U
    names = [str(n) for n in golden["csv_names"]]
    files = []
    for n in names:
        f = tmp_path / n
        Image.new("L", (512, 384), 128).save(str(f))
        files.append(str(f))
    out_csv = str(tmp_path / "hawley_spnet.csv")
    U.show_pred_ellipses(golden["csv_Yt"], golden["csv_Yp"], files, num_draw=len(files), log_dir=str(tmp_path),
                         out_csv=out_csv, show_true=False)
    assert open(out_csv).read() == str(golden["csv_text"])
    assert all((tmp_path / ("steelpan_pred_%05d.png" % j)).exists() for j in range(len(files)))


def test_gemm_tile_table_precedence_and_format():
    """spnet_amd/gemm_tiles.json (tools/autotune_gemm.py): keys parse to (a_major, b_major, stats, M, N, K), tile ids are
    ones the library knows, and an explicit tile beats a probe beats the table beats the library's own choice (0)."""
    import json
    import os
    from spnet_amd import engine as E
    path = os.path.join(os.path.dirname(E.__file__), "gemm_tiles.json")
    with open(path) as f:
        tiles = json.load(f)["tiles"]
    for k, t in tiles.items():
        parts = [int(v) for v in k.split(",")]
        assert len(parts) == 6 and parts[0] in (0, 1) and parts[1] in (0, 1) and parts[2] in (0, 1)
        assert 1 <= int(t) <= 10         # 9 = 32x64, 10 = 32x32 (few-row problems)
        assert E.TILE_TABLE[tuple(parts)] == int(t)
    key = next(iter(E.TILE_TABLE)) if E.TILE_TABLE else (0, 1, 0, 7, 7, 7)
    assert E._tile_for(*key, 5) == 5                                    # explicit
    E.TILE_PROBE[key] = 3
    try:
        assert E._tile_for(*key, 0) == 3                                # probe (the autotuner's override)
    finally:
        E.TILE_PROBE.clear()
    assert E._tile_for(*key, 0) == E.TILE_TABLE.get(key, 0)             # table
    assert E._tile_for(0, 1, 0, 12345, 4, 8, 0) == 0                    # unlisted shape: the library's cost model


def test_bench_self_launches_its_ranks_on_gloo():
    """`python bench.py --gpus 2` started the way the driver starts N=1 (no launcher, no WORLD_SIZE): the parent
    spawns the two ranks as a child torchrun job before touching torch.cuda, relays rank 0's one JSON line and
    returns the child's exit code (--launch-check: rendezvous + one all-reduce, no kernels, so it runs here)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["collective_backend"] == "gloo"
    # a rank that fails makes the parent fail (the WORLD_SIZE / --gpus mismatch inside a launched job)
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       capture_output=True, text=True, timeout=120, env=dict(env2, WORLD_SIZE="1"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_bench_launches_eight_ranks_and_reports_a_failing_rank():
    """What the first 8-GPU run will do before any kernel: `bench.py --gpus 8` starts eight ranks, they rendezvous and
    count each other with one all-reduce (gloo here; --launch-check runs no kernels).  And when ONE rank of a job dies,
    the parent's stderr ends with that rank's own last stderr lines (SPNET_BENCH_FAIL_RANK: a test hook that makes the
    named rank raise after the rendezvous), not only with torchrun's summary."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 8 and out["n_ranks_seen"] == 8 and out["collective_backend"] == "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check"],
                       capture_output=True, text=True, timeout=600, env=dict(env, SPNET_BENCH_FAIL_RANK="2"))
    assert r.returncode != 0
    tail = r.stderr[-6000:]
    assert "rank 2 of 3 failed" in tail and "SPNET_BENCH_FAIL_RANK" in tail and "last" in tail and "stderr lines of" in tail
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no result line from a failed job


def test_stream_accessor_resolves_with_a_public_fallback(monkeypatch):
    """Every kernel launch takes its stream from _lib.current_stream: torch's private raw accessor when it exists, the
    public torch.cuda.current_stream().cuda_stream otherwise (tests/test_engine_gpu.py: both give the same handle)."""
    import torch
    from spnet_amd import _lib
    assert _lib.STREAM_ACCESSOR in ("raw", "public")
    monkeypatch.delattr(torch._C, "_cuda_getCurrentRawStream", raising=False)
    fn, kind = _lib._resolve_current_stream()
    assert kind == "public" and fn is _lib._public_stream


def test_fork_is_refused_under_a_preloaded_profiler(monkeypatch):
    """rocprofv3 initialises the HIP runtime before Python starts (torch.cuda.is_initialized() stays False): the
    frame generator and the PNG loader must not fork then."""
    from spnet_amd import fake_espi as F
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("LD_PRELOAD", raising=False)
    monkeypatch.delenv("HSA_TOOLS_LIB", raising=False)
    assert not F.gpu_may_be_live()
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert F.gpu_may_be_live()
    monkeypatch.delenv("LD_PRELOAD")
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    assert F.gpu_may_be_live()
    X, lab = F.generate(16, seed=3, workers=4)        # serial under the "profiler"; same frames as the forked path
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    X2, lab2 = F.generate(16, seed=3, workers=4)
    assert np.array_equal(X, X2) and lab == lab2


def test_cpu_share_for_the_baseline():
    sys.path.insert(0, ROOT)
    import bench
    n, share = bench.host_cpu_share()
    assert 1 <= n <= share["affinity_cpus"] and n <= 32
    if share["cgroup_quota_cpus"]:
        assert n <= int(share["cgroup_quota_cpus"] + 0.5)


def test_irv2_sibling_groups_and_parameter_packs():
    """The 1x1 convolutions that read one tensor (first layer of every branch of an inception block) form 42 groups --
    10 x block35 (32+32+32), 20 x block17 (192+128), 10 x block8 (192+192), mixed_5b (96+48+64), mixed_7a (3 x 256) -- and
    their beta / moving statistics are the names the parameter layout packs back to back (engine.param_packs)."""
    from collections import Counter
    from spnet_amd import engine as E
    groups = E.irv2_sibling_groups()
    assert len(groups) == 42
    assert Counter(tuple(c for _, _, c in g) for g in groups) == {(32, 32, 32): 10, (192, 128): 20, (192, 192): 10,
                                                                 (96, 48, 64): 1, (256, 256, 256): 1}
    names = {s[0]: s for s in E.param_specs(384, 512, backbone="InceptionResNetV2")}
    packs = E.param_packs("InceptionResNetV2")
    assert len(packs) == 3 * 42 and E.param_packs("Xception") == []
    seen = set()
    for p in packs:
        assert len(p) >= 2
        for n in p:
            assert n in names and n not in seen and int(np.prod(names[n][1])) % 4 == 0
            seen.add(n)
        kinds = {n.split("/")[1] for n in p}
        assert len(kinds) == 1 and kinds <= {"beta", "moving_mean", "moving_variance"}


def test_fake_espi_object_density_of_the_published_dataset():
    """gen_fake_espi.py:250-251 draws 1-7 antinodes per frame since Nov 2020 (its own comment); the published Dataset-A
    run's sets hold 14,965 / 14,952 objects in 4,992 frames = 3.0 per frame, i.e. random.randint(0, 6).  count_range
    selects the generator version; the default stays the current reference's, frame by frame."""
    from spnet_amd import fake_espi as F
    n_new = [len(F.draw_params(s)[1]) for s in range(1500)]
    n_old = [len(F.draw_params(s, (0, 6))[1]) for s in range(1500)]
    assert min(n_new) >= 1 and max(n_new) <= 7 and 3.8 < np.mean(n_new) < 4.1
    assert min(n_old) == 0 and max(n_old) <= 6 and 2.85 < np.mean(n_old) < 3.1          # the log: 2.998 / 2.995
    assert [n[:6] for n in F.draw_params(5)[1]] == [n[:6] for n in F.draw_params(5, (1, 7))[1]]
