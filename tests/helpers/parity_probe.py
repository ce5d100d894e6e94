#!/usr/bin/env python3
"""Dev tool: separates device error from fp32 conditioning for one geometry.

For every seed it prints, per tensor group, the worst relative deviation (max|diff| / max|ref|) from the
oracle evaluated in DOUBLE precision of
  (a) the oracle evaluated in fp32 on the CPU      -> how much fp32 rounding alone moves that tensor
  (b) the HIP engine                               -> must be of the same order as (a)
A tensor where (b) >> (a) is a kernel bug; a tensor where both are large is ill-conditioned at this
seed (tiny BatchNorm populations, ReLU / max-pool near-ties) and says nothing about the kernels.

  python tests/helpers/parity_probe.py H W B [seed ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from oracle import torch_ref as T
from tests.parity_util import make_case, rel_err


def main():
    H, W, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    seeds = [int(s) for s in sys.argv[4:]] or [0, 1, 2, 3]
    from spnet_amd.engine import Engine
    for seed in seeds:
        P, X, Y, mask, dseed = make_case(H, W, B, seed)
        _, _, g64, yp64 = T.Trainer({k: v.double() for k, v in P.items()}).grads(
            X.double(), Y.double(), drop_mask=mask.double(), include_l2=False)
        _, _, g32, yp32 = T.Trainer({k: v.clone() for k, v in P.items()}).grads(X, Y, drop_mask=mask, include_l2=False)
        eng = Engine(H, W, B, device="cuda:0", seed=seed)
        eng.load_state_dict(P)
        eng.set_drop_seed(dseed)
        out = eng.forward(X.cuda(), training=True)
        eng.loss(Y.cuda())
        eng.backward()
        torch.cuda.synchronize()
        gd = eng.grad_dict()
        e32 = {k: rel_err(g32[k].numpy(), g64[k].numpy()) for k in g64}
        edev = {k: rel_err(gd[k].numpy(), g64[k].numpy()) for k in g64}
        w32 = max(e32, key=e32.get)
        wdev = max(edev, key=edev.get)
        ratio = max(edev[k] / max(e32[k], 1e-6) for k in g64)
        print("H%d W%d B%d seed %d | fp32-oracle worst %.2e (%s) | device worst %.2e (%s) | y_pred dev %.2e fp32 %.2e |"
              " max dev/max(fp32,1e-6) %.1f" % (H, W, B, seed, e32[w32], w32, edev[wdev], wdev,
                                               rel_err(out.cpu().numpy(), yp64.numpy()), rel_err(yp32.numpy(), yp64.numpy()),
                                               ratio), flush=True)
        top = sorted(edev.items(), key=lambda kv: -kv[1])[:4]
        print("    device top:", ["%s dev %.1e fp32 %.1e" % (k, v, e32[k]) for k, v in top], flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
