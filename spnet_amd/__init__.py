"""spnet_amd: MI355X-native hot path of SPNet (ellipse / ring-count detection in ESPI frames).

HIP kernels (csrc/, gfx950) behind a C ABI (include/spnet_hip.h), driven by a static launch plan
(engine.py), exposed through the reference's own Python surface (models / utils / callbacks /
augmentation / diagnostics / config).
"""
__version__ = "0.1.0"
