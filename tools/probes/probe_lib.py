"""ctypes binding of tools/probes/lib/libspnet_probe.so: measured probes that are not on the product path.

  spnet_bf16x3_kp / spnet_split_bf16x3 / spnet_gemm_bf16x3_fwd   (gemm_bf16x3.hip)
      x = h + m + l (three bf16, exact), a*b ~ six piece products (error ~ one fp32 rounding per product), six
      v_mfma_f32_16x16x32_bf16 in place of eight fp32 MFMAs per 16x16x32 block.  Forward operand form only: A [M][K] fp32
      (split while staged), W as the three K-major bf16 planes spnet_split_bf16x3 makes of a Keras pointwise kernel
      [K][N] (3 * N * spnet_bf16x3_kp(K) bf16), C [M][N] fp32.  NOT the k-ordered fmaf chain of spnet_gemm_f32:
      bench.py reports it as `roofline_alt` only (pointwise convolutions of the Xception middle flow; call site
      spnet/models.py:357-359).
"""
import ctypes
import os
from ctypes import c_int, c_long, c_void_p

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libspnet_probe.so")


def available():
    return os.path.exists(LIB_PATH)


def load():
    """The probe library (raises if it has not been built: make -C tools/probes)."""
    import torch  # noqa: F401  (the HIP runtime torch has loaded must be the one this library binds to)
    lib = ctypes.CDLL(LIB_PATH)
    P = c_void_p
    lib.spnet_bf16x3_kp.restype, lib.spnet_bf16x3_kp.argtypes = c_long, [c_int]
    lib.spnet_split_bf16x3.restype, lib.spnet_split_bf16x3.argtypes = c_int, [P, P, c_int, c_int, P]
    lib.spnet_gemm_bf16x3_fwd.restype = c_int
    lib.spnet_gemm_bf16x3_fwd.argtypes = [P, c_int, P, P, c_int, c_int, c_int, c_int, P]
    return lib
