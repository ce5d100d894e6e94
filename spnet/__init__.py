"""Drop-in import path of the reference package: `from spnet import models, utils, callbacks, ...`
resolves to the MI355X implementation in spnet_amd/."""
import importlib
import sys

for _name in ("config", "utils", "augmentation", "diagnostics", "multi_gpu", "callbacks", "models"):
    _mod = importlib.import_module("spnet_amd." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
