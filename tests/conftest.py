import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle runs in this process: keep torch's thread pool at the share of cores a test box really has (a
    # 1-GPU box exposes 100+ logical CPUs but grants ~16; 128 oversubscribed threads make the fp64 oracle 20x slower).
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_numpy.npz"), allow_pickle=False)


@pytest.fixture(scope="session", autouse=True)
def _pw_alt_everywhere():
    """SPNET_TEST_PW_ALT=bf16x3: run the WHOLE suite with the bf16x3 measurement hook attached to every Xception engine
    built in this process (tools/probes/bf16x3_hook.py through Engine.pw_alt) -- the evidence VERDICT r3 item 8 asks for
    before that road could be offered as the product path: every end-to-end parity test at the suite's own tolerances."""
    if os.environ.get("SPNET_TEST_PW_ALT") != "bf16x3":
        yield
        return
    from spnet_amd import engine as E
    from tools.probes.bf16x3_hook import Bf16x3Pointwise
    orig = E.Engine._build_graph

    def build(self):
        orig(self)
        if self.backbone == "Xception":
            self.pw_alt = Bf16x3Pointwise(self)

    E.Engine._build_graph = build
    try:
        yield
    finally:
        E.Engine._build_graph = orig
