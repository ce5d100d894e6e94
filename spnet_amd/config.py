"""Run-wide constants and mode switches -- same names and meanings as the reference's
spnet/config.py (:4, :30-38, :40, :48, :52); edited in place or overridden by the CLI scripts'
additive flags (--backbone / --model_type / --loss_type)."""
import numpy as np

dtype = np.float32
meta_extension = ".csv"

# layout of one predictor's 8 variables inside the flat 6x6x2x8 output (spnet/config.py:30-38)
vars_per_pred = 8
(ind_cx, ind_cy, ind_semi_a, ind_semi_b,
 ind_angle1, ind_angle2,          # cos(2 theta), sin(2 theta)
 ind_noobj, ind_rings) = range(vars_per_pred)

# 'same'  : MSE on every variable;  anything else ('hybrid'): BCE-with-logits on noobj (config.py:40)
loss_type = 'same'
# 'monolithic' (default): frames resized to 331x331; 'big': native 384x512 frames;
# 'simple' / 'compound' / 'ss': legacy head variants of the reference (config.py:42-48)
model_type = 'monolithic'
# backbone selector (config.py:50-52); this build implements 'Xception'
basemodel = 'Xception'


def _bgr(rgb):
    return tuple(rgb[::-1])


# drawing colours.  The reference stores them BGR for OpenCV; the overlay code here draws with PIL
# (RGB) and converts at the call site, so the BGR tuples are kept for API compatibility.
blue, red, green = (255, 0, 0), (0, 0, 255), (0, 200, 0)
white, black, grey, lightgrey = (255, 255, 255), (0, 0, 0), (128, 128, 128), (210, 210, 210)
yellow = _bgr((255, 255, 0))
cyan = _bgr((0, 220, 220))
mpl_blue = _bgr((31, 140, 200))
mpl_orange = _bgr((255, 127, 14))
veridis_purple = _bgr((72, 18, 84))
veridis_lightgreen = _bgr((97, 207, 99))
veridis_yellow = _bgr((254, 228, 76))
magma_light = _bgr((253, 252, 197))
truecolor = yellow
predcolor = veridis_purple

# (additive, no counterpart in the reference: dtype above stays float32 either way)  Which kernel multiplies the forward /
# data-gradient GEMMs of the wide pointwise convolutions: 'bf16x3' -- fp32 operands split exactly into three bf16 pieces
# on the bf16 matrix cores, fp32 accumulation, fp32-accurate (csrc/gemm_bf16x3.hip; the default) -- or 'f32': the
# k-ordered fp32 MFMA chain for every GEMM.
pointwise_gemm = 'bf16x3'
