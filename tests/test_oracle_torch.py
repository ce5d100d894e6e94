"""Structural known-answers for the torch network oracle (the only pins the reference offers for the
Keras part of the path: paper/run_logs/log_DatasetA...txt:94-101)."""
import numpy as np
import torch

from oracle import torch_ref as T


def test_param_counts_match_reference_log():
    P = T.init_params(331, 331)
    total, trainable, frozen = T.count_params(P)
    assert (total, trainable, frozen) == (50353481, 50298935, 54546)
    assert sum(1 for l in T.xception_layers() if l[0] == "sep") == 34
    assert T.backbone_out_hw(331, 331) == (5, 5)
    assert T.backbone_out_hw(384, 512) == (6, 8)
    P2 = T.init_params(384, 512)
    assert T.count_params(P2)[0] == 77485385


def test_shapes_small_forward():
    torch.manual_seed(0)
    P = T.init_params(96, 128, seed=1)
    X = torch.rand(2, 96, 128, 1) * 2 - 1
    taps = {}
    y = T.forward(P, X, training=False, taps=taps)
    assert tuple(taps["stem"].shape) == (2, 48, 64, 3)
    assert tuple(taps["backbone"].shape) == (2,) + T.backbone_out_hw(96, 128) + (2048,)
    assert tuple(y.shape) == (2, 576)
    assert torch.isfinite(y).all()


def test_same_padding_asymmetry():
    # even size: TF pads 0 before / 1 after for 3x3/s2 -> first window starts at index 0
    x = torch.arange(16.0).reshape(1, 4, 4, 1)
    y = T.maxpool3x3s2_same(x)
    assert y.shape == (1, 2, 2, 1)
    np.testing.assert_array_equal(y[0, :, :, 0].numpy(), [[10, 11], [14, 15]])
    # odd size: pad 1 / 1
    x = torch.arange(25.0).reshape(1, 5, 5, 1)
    y = T.maxpool3x3s2_same(x)
    np.testing.assert_array_equal(y[0, :, :, 0].numpy(), [[6, 8, 9], [16, 18, 19], [21, 23, 24]])


def test_train_step_reduces_loss():
    torch.manual_seed(0)
    P = T.init_params(64, 64, seed=2)
    tr = T.Trainer(P)
    X = torch.rand(4, 64, 64, 1) * 2 - 1
    Y = torch.rand(4, 576)
    Y[:, 6::8] = (Y[:, 6::8] > 0.5).float()
    l0 = tr.step(X, Y, 1e-3)[0]
    for _ in range(5):
        l1 = tr.step(X, Y, 1e-3)[0]
    assert l1 < l0
