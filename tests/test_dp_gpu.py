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
                        "--master-addr", "127.0.0.1", "--master-port", "29713", HELPER, dp, "3"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(single), np.load(dp)
    assert set(a.files) == set(b.files)
    bad = [k for k in a.files if not np.array_equal(a[k], b[k])]
    assert not bad, "data-parallel run differs from the single-process run in: %s" % bad[:8]
