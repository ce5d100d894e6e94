"""ctypes binding of libspnet_hip.so (C ABI declared in include/spnet_hip.h).

The product path has NO CPU fallback: importing this module without the built library raises, and
every wrapper raises on a non-zero HIP status.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_float, c_int, c_long, c_uint, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPNET_HIP_LIB") or os.path.join(_HERE, "lib", "libspnet_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "spnet_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C spnet_amd/csrc` (hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)

# torch first: libspnet_hip.so must bind to the SAME HIP runtime (libamdhip64) that torch has loaded, or
# the two runtimes would not share devices / streams (kernels then fail with hipErrorNoDevice).
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

_lib = ctypes.CDLL(LIB_PATH)

P = c_void_p  # device pointers travel as integers

_SIGS = {
    "spnet_gemm_f32": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_long, P, c_int, P]),
    "spnet_bf16x3_kp": (c_long, [c_int]),
    "spnet_bf16x3_plane_elems": (c_long, [c_long, c_int]),
    "spnet_split_bf16x3": (c_int, [P, P, c_int, c_int, P]),
    "spnet_split_rows_bf16x3": (c_int, [P, c_long, P, c_long, c_int, P]),
    "spnet_gemm_bf16x3_pp": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_gemm_bf16x3_wgrad_ksplit": (c_long, [c_int, c_int, c_int, c_int]),
    "spnet_gemm_bf16x3_dwbwd_rows": (c_long, [c_long]),
    "spnet_gemm_bf16x3_dwbwd_ok": (c_long, [c_int, c_int, c_int]),
    "spnet_gemm_bf16x3_pp_dwbwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, P, P, P, P, P, P, P]),
    "spnet_gemm_bf16x3_pp_dwfwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P]),
    "spnet_gemm_bf16x3_wgrad_batched": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_split_bf16x3_batched": (c_int, [P, c_int, c_long, P]),
    "spnet_gemm_bf16x3_fwd": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_gemm_bf16x3_fwd_colstats": (c_int, [P, c_int, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_gemm_f32_accumulate": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_conv3x3_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_conv3x3_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_conv3x3_wgrad_ws": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_conv3x3_wgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_long, P]),
    "spnet_reduce_slabs": (c_int, [P, c_int, c_int, c_int, P, c_int, P]),
    "spnet_transpose_batched": (c_int, [P, c_int, c_int, c_int, P]),
    "spnet_gather_s2": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_scatter_add_s2": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_dwconv3x3_strided_ws": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_strided": (c_int, [c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "spnet_reduce_rows": (c_int, [P, c_int, c_int, P, P]),
    "spnet_reduce_rows_ws": (c_int, [P, c_int, c_int, P, P, c_long, P]),
    "spnet_reduce_rows_batched": (c_int, [P, c_int, c_int, P]),
    "spnet_dwconv3x3_tiled_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_dwconv3x3_tiled_fwd_bnfin": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_long, P, P, P, P, P, P, P,
                                                c_float, c_float, P]),
    "spnet_dwconv3x3_tiled_fwd_x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_dwconv3x3_tiled_fwd_bnfin_x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, c_int, c_long, P, P, P, P, P, P, P,
                                                   c_float, c_float, P]),
    "spnet_dwconv3x3_stream_fwd_x3": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "spnet_dwconv3x3_tiled_bwd_ws": (c_long, [c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_tiled_rows": (c_long, [c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_tiled_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P]),
    "spnet_dwconv3x3_prefers_stream": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_stream_rows": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_stream_bwd_ws": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_dwconv3x3_stream_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P]),
    "spnet_dwconv3x3_stream_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, c_int, P]),
    "spnet_bn_finalize_fwd": (c_int, [P, c_int, c_long, c_int, P, P, P, P, P, P, P, c_float, c_float, P]),
    "spnet_bn_infer_coeffs": (c_int, [c_int, P, P, P, P, P, c_float, P]),
    "spnet_bn_apply": (c_int, [P, c_long, c_int, P, c_int, P, c_int, P, P]),
    "spnet_bn_finalize_apply": (c_int, [P, c_int, P, c_long, c_int, P, P, P, P, P, P, P, c_int, P, P, c_float, c_float, P]),
    "spnet_bn_finalize_apply_ld": (c_int, [P, c_int, P, c_long, c_int, P, P, P, P, P, P, P, c_int, P, P, c_long, c_float, c_float, P]),
    "spnet_bn_bwd_from_partials": (c_int, [P, P, c_long, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "spnet_bn_bwd_from_partials_x3": (c_int, [P, P, c_long, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "spnet_bn_bwd_x3": (c_int, [P, P, c_long, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "spnet_gemm_f32_colstats": (c_int, [P, c_int, c_int, P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_gemm_f32_bnblend": (c_int, [P, P, P, c_int, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "spnet_bn_bwd_coeffs_from_partials": (c_int, [c_int, P, c_long, c_int, P, P, P, P, P, P, c_int, P]),
    "spnet_bn_bwd_coeffs": (c_int, [P, P, c_long, c_int, P, P, P, P, P, P, P, c_int, P, P]),
    "spnet_gemm_f32_batched": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_gemm_batched_ksplit": (c_long, [c_int, c_int, c_int, c_int, P]),
    "spnet_gemm_f32_batched_splitk": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                              c_int, P, c_long, P]),
    "spnet_bn_ws": (c_long, [c_long, c_int]),
    "spnet_bn_fwd_train": (c_int, [P, c_long, c_int, P, P, P, P, P, P, P, c_int, P, c_int, P, c_float, c_float, P, P]),
    "spnet_bn_fwd_infer": (c_int, [P, c_long, c_int, P, P, P, P, P, c_int, P, c_int, P, c_float, P]),
    "spnet_bn_fwd_train_ld": (c_int, [P, c_long, c_int, P, P, P, P, P, P, P, c_int, P, c_int, P, c_long, c_float, c_float, P, P]),
    "spnet_bn_fwd_infer_ld": (c_int, [P, c_long, c_int, P, P, P, P, P, c_int, P, c_int, P, c_long, c_float, P]),
    "spnet_bn_bwd": (c_int, [P, P, c_long, c_int, P, P, P, P, c_int, P, P, P, P, P, P]),
    "spnet_maxpool3x3s2_add_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_maxpool3x3s2_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_maxpool3x3s2_bwd_rows": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "spnet_maxpool3x3s2_bwd_bnsums": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P]),
    "spnet_maxpool3x3s2_valid_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_maxpool3x3s2_valid_bwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_avgpool3x3s1_same": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_patches": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_patches_ld": (c_int, [P, c_long, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_conv_gemm_f32": (c_int, [P, c_long, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int, P, P, P]),
    "spnet_grad_bnsums_rows": (c_long, [c_long, c_int]),
    "spnet_patches_bwd_bnsums": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_int, P,
                                         c_int, P]),
    "spnet_copy_cols_bnsums": (c_int, [P, c_int, P, c_long, c_int, P, c_long, P, P, P, c_int, P, c_int, P]),
    "spnet_patches_bwd_bnsums_ld": (c_int, [P, P, c_long, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_long, P,
                                            c_long, P, P, c_int, P, c_long, c_int, P]),
    "spnet_copy_cols_bnsums_ld": (c_int, [P, c_int, P, c_long, c_long, c_int, P, c_long, P, c_long, P, P, c_int, P, c_long,
                                          c_int, P]),
    "spnet_copy_cols_batched": (c_int, [P, c_int, c_long, P]),
    "spnet_resadd": (c_int, [P, P, P, c_long, c_float, c_int, P]),
    "spnet_resadd_bwd": (c_int, [P, P, P, P, c_long, c_float, c_int, P]),
    "spnet_copy_cols": (c_int, [P, c_int, P, c_int, c_long, c_int, c_int, P]),
    "spnet_avgpool2_fwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_avgpool2_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "spnet_conv3x3_small": (c_int, [c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P, c_long, P]),
    "spnet_stem_head": (c_int, [c_int, P, P, P, P, c_int, c_int, c_int, P, c_long, P]),
    "spnet_ellipse_loss": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "spnet_selective_sigmoid": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "spnet_decode": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "spnet_ellipse_iou": (c_int, [P, P, c_long, c_int, c_int, P, P]),
    "spnet_calc_errors": (c_int, [P, P, c_long, c_int, P, P, P]),
    "spnet_adam_step": (c_int, [P, P, P, P, c_long, c_long, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P, P, P]),
    "spnet_adam_parts": (c_long, [c_long]),
    "spnet_adam_part": (c_int, [P, P, P, P, c_long, c_long, c_float, c_float, c_float, c_float, c_float, c_float, P, P, P, P]),
    "spnet_adam_l2_sum": (c_int, [P, c_int, c_float, P, P]),
    "spnet_u8_to_input": (c_int, [P, P, c_long, P]),
    "spnet_gather_rows": (c_int, [P, c_long, P, c_int, P, c_int, c_long, P]),
    "spnet_minmax": (c_int, [P, c_int, c_long, P, P, P]),
    "spnet_cutout": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "spnet_saltpepper": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P, P, P]),
    "spnet_gaussian_blur": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "spnet_warp_affine": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P]),
    "spnet_warp_affine_fixed": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, P]),
    "spnet_fake_espi": (c_int, [P, P, P, c_int, c_int, c_int, c_uint, c_int, P, P, P]),
    "spnet_dropout": (c_int, [P, P, c_long, c_uint, c_float, P, P]),
}

EXPORTS = tuple(_SIGS)


class HipError(RuntimeError):
    pass


# Measurement hook (bench.py's per-family rooflines): while set, every launching entry point calls
# tracer(name, args, None) -> token before and tracer(name, args, token) after it has enqueued its kernels.  None (the
# default, and the only state outside a measurement pass): one list lookup per call.
_TRACER = [None]


def set_tracer(tracer):
    _TRACER[0] = tracer


def _bind(name, restype, argtypes):
    fn = getattr(_lib, name)      # AttributeError here = the library does not export the symbol
    fn.restype = restype
    fn.argtypes = argtypes
    if restype is not c_int:
        return fn

    def checked(*args):
        tr = _TRACER[0]
        if tr is None:
            rc = fn(*args)
        else:
            tok = tr(name, args, None)
            rc = fn(*args)
            tr(name, args, tok)
        if rc != 0:
            raise HipError("%s failed with hipError_t %d" % (name, rc))
    checked.__name__ = name
    return checked


for _n, (_r, _a) in _SIGS.items():
    globals()[_n] = _bind(_n, _r, _a)


def _public_stream():
    return torch.cuda.current_stream().cuda_stream


def _resolve_current_stream():
    """Raw handle of the current HIP stream of the current device, for the `stream` argument of every entry point.
    torch.cuda.current_stream().cuda_stream builds a Stream object per call (~2 us; a step is 330 ... 1,300 launches);
    torch's private accessors return the same handle without it.  They are not public API and have moved between torch
    releases, so they are resolved ONCE here and anything missing falls back to the public call
    (tests/test_host_cpu.py checks the resolution; tests/test_engine_gpu.py that both return the same handle)."""
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    dev = getattr(torch._C, "_cuda_getDevice", None)
    if raw is None or dev is None:
        return _public_stream, "public"
    return (lambda: raw(dev())), "raw"


current_stream, STREAM_ACCESSOR = _resolve_current_stream()


def gather_rows(src, index, dst):
    """dst[i] = src[index[i]] over the leading axis (device tensors; fp32 rows, int32 / int64 index) on the current stream."""
    n = int(index.numel())
    rows = int(src.shape[0])
    L_ = src.numel() // max(rows, 1)
    if dst.numel() != n * L_ or src.dtype != torch.float32 or dst.dtype != torch.float32 or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("gather_rows: shape / dtype / layout mismatch")
    ib = {torch.int32: 4, torch.int64: 8}[index.dtype]
    spnet_gather_rows(src.data_ptr(), rows, index.data_ptr(), ib, dst.data_ptr(), n, L_, current_stream())


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


class AsyncUploader:
    """numpy array -> device tensor WITHOUT synchronising the stream.

    A copy from pageable memory makes torch wait for everything queued before it -- the whole previous train
    step -- so the host could never run ahead and the GPU idled ~0.4 ms at every step boundary.  Per-step
    parameters therefore go through a small ring of pinned staging buffers (one ring per key / shape) and
    asynchronous copies into matching device buffers; a slot is reused only after its own copy has completed
    (event), and stream order protects the device buffer from being overwritten while kernels still read it.

    The ring depth also bounds how far the host runs AHEAD of the GPU (a key is uploaded once per step, and the host
    waits for the copy of `depth` steps ago before it stages the next one).  Default 1: one step of lead hides the
    host's 4-5 ms of enqueue work per 12.5 ms step completely, and the GPU itself runs faster that way -- with three
    or more steps queued behind the running one a train step takes 13.4-13.6 ms instead of 12.45 (bench.py
    SPNET_BENCH_TRACE per-step trace at depth 1 / 2 / 3 / 4 / 8: 12.73 / 12.83 / 12.85 / 12.95 / 12.98 ms per step
    over the first ten steps after a fence; the slow steps are exactly those during which the host is not blocked)."""

    def __init__(self, device, depth=None):
        self.device = device
        self.depth = int(os.environ.get("SPNET_UPLOAD_DEPTH", "1")) if depth is None else depth
        self.rings, self.count = {}, {}

    def __call__(self, key, array):
        array = np.ascontiguousarray(array)
        k = (key, array.shape, array.dtype.str)
        ring = self.rings.get(k)
        if ring is None:
            ring = []
            for _ in range(self.depth):
                host = torch.from_numpy(np.empty_like(array)).pin_memory()
                ring.append((host, torch.empty_like(host, device=self.device), torch.cuda.Event()))
            self.rings[k], self.count[k] = ring, 0
        host, dev, ev = ring[self.count[k] % self.depth]
        self.count[k] += 1
        ev.synchronize()                     # returns at once unless the GPU is `depth` uploads behind
        host.numpy()[...] = array
        dev.copy_(host, non_blocking=True)
        ev.record()
        return dev
