"""Data parallelism end to end on the GPU: two ranks (gloo, both on cuda:0 -- RCCL wants one device per rank,
the driver's 8-GPU run covers that) fed the SAME batch must end up with exactly the single-process weights:
sum of two equal gradients x 1/2 is exact, so any difference would be a synchronisation bug between the
weight-gradient stream, the deferred batched launches and the all-reduce."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port that is free right now on 127.0.0.1 (fixed rendezvous ports collide when two suites share a host)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return str(so.getsockname()[1])
HELPER = os.path.join(ROOT, "tests", "helpers", "dp_train_small.py")


def test_two_rank_data_parallel_equals_single_process(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PYTHONPATH=ROOT, SPNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    single, dp = str(tmp_path / "single.npz"), str(tmp_path / "dp.npz")
    r = subprocess.run([sys.executable, HELPER, single, "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), HELPER, dp, "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(single), np.load(dp)
    assert set(a.files) == set(b.files)
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, "data-parallel run differs from the single-process run in: %s" % bad[:8]


def test_bucketed_allreduce_through_rccl_on_one_rank(tmp_path):
    """The gradient all-reduce on the "nccl" backend (= RCCL): a forced one-rank group on the one GPU executes every
    bucket's collective on RCCL's stream, launched from the weight-gradient stream while backward is running.
    The sum over one rank is the identity, so the weights must equal the no-reducer run bit for bit -- any
    difference is an ordering bug between the two compute streams and RCCL's."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
    env.pop("SPNET_DIST_BACKEND", None)
    single, rccl = str(tmp_path / "single.npz"), str(tmp_path / "rccl.npz")
    r = subprocess.run([sys.executable, HELPER, single, "3"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([sys.executable, HELPER, rccl, "3", "rccl1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "backend nccl" in r.stdout, r.stdout
    a, b = np.load(single), np.load(rccl)
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, "RCCL run differs from the plain run in: %s" % bad[:8]


def test_two_rank_training_through_train_network(tmp_path):
    """train_spnet.py under torch.distributed.run with two ranks (gloo, both on the one GPU): Model.fit shards every
    epoch by rank, augments only its shard, all-reduces the gradient buckets, and rank 0 alone writes logs and
    checkpoints.  The replicas' weights must stay identical."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import fake_espi as F
    data = tmp_path / "data"
    F.write_dataset(str(data / "Train"), 64, seed=1)
    F.write_dataset(str(data / "Val"), 16, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, PYTHONPATH=ROOT, SPNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0",
               SPNET_DUMP_WEIGHT_SUM=str(work / "wsum"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", _free_port(), os.path.join(ROOT, "train_spnet.py"),
                        "-d", str(data), "-b", "8", "-e", "2", "--name", "dp"], cwd=str(work), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    assert "SPNet execution completed." in r.stdout
    logs = [d for d in os.listdir(work / "logs") if d.startswith("dp_")]
    assert len(logs) == 1                                   # one writer
    rows = [l for l in open(work / "logs" / logs[0] / "losses.dat") if not l.startswith("#")]
    assert len(rows) == 2 and all(np.isfinite(float(x.split()[1])) for x in rows)
    sums = [open(str(work / "wsum") + ".%d" % k).read().strip() for k in (0, 1)]
    assert sums[0] == sums[1], sums                         # replicas in lock-step


def test_bench_self_launched_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's N = 1 command): the parent starts
    two ranks itself, both on the one GPU over gloo here, and relays rank 0's single JSON line -- the full
    data-parallel benchmark step (sharded pools, bucketed all-reduce during backward, 1/world in Adam) at reduced pool
    and step counts; `n_ranks_seen` is counted by an all-reduce, `value` is the whole job's rate."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT, SPNET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--pool", "128", "--sustained-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["n_ranks_seen"] == 2 and out["config"]["global_batch"] == 64
    assert out["config"]["collective_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["value"] > 0 and np.isfinite(out["config"]["final_loss"])
    assert "cpu_baseline" not in out and "predict" not in out          # N = 1 only
    assert 0 < out["roofline"]["frac"] < 1 and out["kernel_families"]["gemm"]["launches_per_step"] == 97
    # what an N-GPU shortfall is attributable to: exposed all-reduce time, bytes, the bucket plan
    comm = out["config"]["comm"]
    assert comm["allreduce_exposed_ms_per_step"] >= 0 and comm["bytes_allreduced_per_step_per_rank"] == 309739008
    assert comm["ring_bytes_sent_per_step_per_rank"] == 309739008 and comm["buckets"]["dense_head_pieces"] == 7
    assert abs(sum(comm["buckets"]["mb"]) + comm["buckets"]["tail_at_end_of_backward_mb"] - 309739008 / 2 ** 20) < 1.0


def test_bench_default_line_carries_every_block():
    """The driver's N = 1 command with short step counts: one JSON line with the roofline of the dominant family, the
    secondary family by sub-family (and, inside `roofline`, by arithmetic), the predict / 331 x 331 / other-backbone legs,
    `roofline_alt` (the bf16x3 kernel isolated, and the all-fp32-chain train step beside the product's) and the CPU baseline
    (one short oracle step here) -- every leg of the default run executes, none is skipped by a flag."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--pool", "256",
                        "--sustained-seconds", "0", "--cpu-baseline-steps", "1", "--cpu-baseline-batch", "2"],
                       env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("roofline", "roofline_secondary", "roofline_other", "roofline_alt", "cpu_baseline", "predict", "layout_331",
                "kernel_families", "irv2", "mobilenet"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["dtype"].startswith("f32 (") and "bf16x3" in out["dtype"] and out["value"] > 0
    assert "fp32 MFMA" in out["config"]["arithmetic"]
    # `roofline` = the dominant KERNEL against its own peak (the planes x planes bf16x3 GEMM: bf16 MFMA peak / 6), every
    # GEMM kernel against its own beside it, the family's time-weighted fraction, the old fp32-peak figure only as legacy
    roof = out["roofline"]
    bk = roof["by_kernel"]
    assert roof["kernel"].startswith("gemm_bf16x3_") and abs(roof["peak"] - 416.7) < 0.1 and 0 < roof["frac"] < 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert {"gemm_bf16x3_pp_kernel", "gemm_bf16x3_wgrad_kernel"} <= set(bk) and len(bk) >= 3
    # the planes kernels carry the bytes they have to move per launch (what the PMC figure `traffic` is read against)
    assert bk["gemm_bf16x3_pp_kernel"]["algorithmic_bytes_per_launch"] > 4e7
    assert bk["gemm_bf16x3_wgrad_kernel"]["algorithmic_bytes_per_launch"] > 1e8
    if roof["traffic"] is not None and roof.get("algorithmic_bytes_per_launch"):
        assert 0.9 < roof["traffic_over_algorithmic"] < 3.0
    assert all(v["launches_per_step"] > 0 and v["ms_per_step"] > 0 and 0 < v["frac"] < 1 for v in bk.values())
    fam = roof["family"]
    assert 0 < fam["frac_time_weighted"] < 1 and "frac_vs_fp32_peak_legacy" in fam
    assert abs(fam["ms_per_step"] - sum(v["ms_per_step"] for v in bk.values())) < 0.01
    assert out["predict"]["roofline"]["kernel"].startswith("gemm_bf16x3_pp") and out["predict"]["roofline"]["peak"] > 400
    # the rest of the step: BatchNorm passes, pooling, the stem and the optimizer against the HBM peak
    other = out["roofline_other"]
    for famname in ("bn", "pool", "stem", "optimizer"):
        assert other[famname]["ms_per_step"] > 0 and 0 < other[famname]["frac"] < 1, famname
    # 28 bytes per trainable parameter (77,485,385 - 54,546 BatchNorm moving statistics, + alignment padding of the flat buffer)
    assert abs(other["optimizer"]["algorithmic_bytes_per_step"] / (28 * 77430839.0) - 1) < 1e-3
    ts = out["roofline_alt"]["train_step"]
    assert ts["exact_fp32"]["bf16x3_launches_per_step"] == 0 and ts["bf16x3"]["bf16x3_launches_per_step"] > 40
    assert set(out["roofline_secondary"].get("sub_families", out["roofline"].get("sub_families", {}))) >= {"entry", "middle", "exit"}
    alt = out["roofline_alt"]
    assert alt["max_rel_err_vs_f64"] < 5e-7 and alt["avg_launch_us"] > 0 and 0 < alt["frac"] < 1
    assert alt["bit_identical_to_fp32_a_kernel"] and alt["wgrad_from_planes_us"] > 0
    assert out["cpu_baseline"]["kind"] in ("port", "reference") and out["cpu_baseline"]["value"] > 0
