"""The reference's three entry points, end to end on a tiny fake-ESPI dataset (configs[0]-style
plumbing run: small set, batch 8, a few epochs), on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + "\n" + r.stderr[-3000:]
    return r.stdout


def test_train_evaluate_predict_cli(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from spnet_amd import fake_espi as F
    data = tmp_path / "data"
    F.write_dataset(str(data / "Train"), 48, seed=1)
    F.write_dataset(str(data / "Val"), 16, seed=2)
    work = tmp_path / "work"
    work.mkdir()
    out = run([os.path.join(ROOT, "train_spnet.py"), "-d", str(data), "-b", "8", "-e", "3", "--name", "t"], str(work))
    assert "SPNet execution completed." in out
    assert "mAP = " in out
    for f in ("final_weights.hdf5", "full_model.h5"):
        assert (work / f).exists(), f
    logs = [d for d in os.listdir(work / "logs") if d.startswith("t_")]
    assert logs and os.path.exists(work / "logs" / logs[0] / "losses.dat")
    rows = [l for l in open(work / "logs" / logs[0] / "losses.dat") if not l.startswith("#")]
    assert len(rows) == 3
    losses = [float(r.split()[1]) for r in rows]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]          # it learns
    # evaluate + predict from the saved files
    out = run([os.path.join(ROOT, "evaluate_spnet.py"), "-d", str(data / "Val"), "-b", "8", "-l", "logs/Testing/"], str(work))
    assert "mAP = " in out and "FPS = " in out
    out = run([os.path.join(ROOT, "predict_spnet.py"), "-w", "final_weights.hdf5", "-d", str(data / "Val"), "-b", "8",
               "-l", "logs/Predicting/"], str(work))
    assert "FPS = " in out
    csv = (work / "logs" / "Predicting" / "hawley_spnet.csv").read_text().strip().splitlines()
    assert len(csv) >= 16 and all(len(l.split(",")) == 7 for l in csv)
    # the same predictions from uint8 frames scaled on the device (additive flag): identical CSV
    out = run([os.path.join(ROOT, "predict_spnet.py"), "-w", "final_weights.hdf5", "-d", str(data / "Val"), "-b", "8",
               "-l", "logs/PredictingU8/", "--u8_frames"], str(work))
    assert "FPS = " in out
    assert (work / "logs" / "PredictingU8" / "hawley_spnet.csv").read_text().strip().splitlines() == csv
