"""ctypes binding of libosmosis_hip.so (the C ABI declared in include/osmosis_hip.h).

PyTorch is plumbing here: it owns device memory and streams; every kernel is reached through a
plain C call with raw device pointers.  There is NO CPU / eager fallback: if the library is not
built, or a tensor is not a CUDA(HIP) fp32 tensor, the call raises.

A `Recorder` captures the exact sequence of C calls (function + marshalled arguments) issued while
it is active, so a whole UNet forward/backward or sampler step can be replayed with near-zero Python
overhead (and captured into a hipGraph through torch.cuda.CUDAGraph).
"""
import ctypes as C
import os
import threading
from typing import List, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSM_LIB") or os.path.join(_HERE, "libosmosis_hip.so")   # OSM_LIB: A/B builds

c_float_p = C.c_void_p  # device pointers travel as opaque addresses


class OsmosisHipError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("y", C.c_void_p), ("splitk_ws", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("ksize", C.c_int), ("splitk", C.c_int), ("accumulate", C.c_int),
                ("ldx", C.c_longlong), ("ldy", C.c_longlong), ("ldr", C.c_longlong), ("wfmt", C.c_int),
                ("gn_table", C.c_void_p), ("gn_silu", C.c_int),
                ("colsum", C.c_void_p), ("stat_mode", C.c_int), ("stat_silu", C.c_int), ("stat_x", C.c_void_p),
                ("ld_sx", C.c_longlong), ("stat_table", C.c_void_p), ("x_maxabs", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("Bm", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
                ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("b_kn", C.c_int),
                ("nb1", C.c_int), ("nb2", C.c_int), ("accumulate", C.c_int), ("alpha", C.c_float),
                ("lda", C.c_longlong), ("ldb", C.c_longlong), ("ldc", C.c_longlong), ("ldr", C.c_longlong),
                ("sA1", C.c_longlong), ("sB1", C.c_longlong), ("sC1", C.c_longlong),
                ("sA2", C.c_longlong), ("sB2", C.c_longlong), ("sC2", C.c_longlong),
                ("splitk", C.c_int), ("splitk_ws", C.c_void_p)]


class AttnDesc(C.Structure):
    _fields_ = [("qkv", C.c_void_p), ("ldqkv", C.c_longlong),
                ("q_off", C.c_int), ("k_off", C.c_int), ("v_off", C.c_int), ("head_stride", C.c_int),
                ("B", C.c_int), ("T", C.c_int), ("heads", C.c_int), ("ch", C.c_int), ("scale", C.c_float),
                ("out", C.c_void_p), ("ldout", C.c_longlong), ("dout", C.c_void_p), ("lddout", C.c_longlong),
                ("dqkv", C.c_void_p), ("lddqkv", C.c_longlong), ("ws", C.c_void_p), ("arith", C.c_int)]


class PhysDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("depth_type", C.c_int), ("dval", C.c_float * 3),
                ("weight_type", C.c_int), ("wdepth_type", C.c_int), ("wval", C.c_float * 3),
                ("loss_type", C.c_int), ("gamma_avrg", C.c_float), ("gamma_val", C.c_float),
                ("eta", C.c_float * 3), ("B", C.c_int), ("HW", C.c_int), ("optimizer", C.c_int)]


# name -> argtypes (restype is int unless listed in _SPECIAL)
_LL = C.c_longlong
_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_SIGS = {
    "osm_conv2d_nhwc": [C.POINTER(ConvDesc), _P],
    "osm_pack_conv_weight": [_P, _P, _P, _I, _I, _I, _P],
    "osm_gemm": [C.POINTER(GemmDesc), _P],
    "osm_pack_conv_weight_bf16s": [_P, _P, _P, _I, _I, _I, _I, _P],
    "osm_splitk_hint": [_I, _I, _I, _I, _I],
    "osm_conv_splitk": [_I, _I, _I, _I, _I, _I, _I, _I],
    "osm_conv_winograd_ok": [_I, _I, _I, _I, _I, _I],
    "osm_pack_conv_weight_winograd": [_P, _P, _P, _I, _I, _I, _P],
    "osm_conv_stat_chunks": [_I, _I, _I, _I, _I, _I, _I, _I, _I],
    "osm_gn_finalize_cols": [_P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _LL, _P, _P],
    "osm_gn_bwd_apply": [_P, _LL, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _I, _I, _I, _I, _P, _P, _P, _P, _P, _LL, _I, _P, _P],
    "osm_attn_small_supported": [_I, _I],
    "osm_attn_flash_supported": [_I, _I],
    "osm_attn_flash_fwd": [C.POINTER(AttnDesc), _P, _P],
    "osm_attn_flash_bwd": [C.POINTER(AttnDesc), _P, _LL, _P, _P, _P],
    "osm_attn_small_fwd": [C.POINTER(AttnDesc), _P],
    "osm_attn_small_bwd": [C.POINTER(AttnDesc), _P],
    "osm_gn_nchunk": [_I],
    "osm_gn_stats": [_P, _LL, _I, _I, _I, _I, _F, _P, _P, _P],
    "osm_gn_apply": [_P, _LL, _P, _LL, _I, _I, _I, _I, _P, _P, _P, _P, _LL, _I, _P, _P],
    "osm_gn_fwd": [_P, _LL, _P, _LL, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _LL, _I, _P, _P, _P],
    "osm_gn_prep": [_P, _LL, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P, _LL, _P, _P, _P],
    "osm_gn_bwd": [_P, _LL, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _I, _I, _I, _I, _P, _P, _P, _P, _LL, _I, _P, _P, _P, _P],
    "osm_pool2x2": [_P, _LL, _P, _LL, _I, _I, _I, _I, _F, _P],
    "osm_upsample2x": [_P, _LL, _P, _LL, _I, _I, _I, _I, _F, _P],
    "osm_resample_pair": [_I, _P, _LL, _P, _LL, _P, _LL, _P, _LL, _I, _I, _I, _I, _F, _P],
    "osm_softmax_rows": [_P, _P, _P, _I, _I, _P],
    "osm_softmax_rows_bwd": [_P, _P, _P, _P, _I, _I, _P],
    "osm_timestep_embedding": [_P, _P, _I, _I, _F, _P],
    "osm_linear": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "osm_nchw_to_nhwc": [_P, _P, _LL, _I, _I, _I, _P],
    "osm_nhwc_to_nchw": [_P, _LL, _P, _I, _I, _I, _P],
    "osm_copy2d": [_P, _LL, _P, _LL, _LL, _I, _I, _P],
    "osm_maxabs": [_P, _LL, _I, _LL, _I, _P, _P],
    "osm_maxabs_parts": [],
    "osm_conv_kernel_kind": [_I, _I, _I, _I, _I, _I, _I],
    "osm_stride2_pick": [_P, _LL, _P, _LL, _I, _I, _I, _I, _P],
    "osm_stride2_place": [_P, _LL, _P, _LL, _I, _I, _I, _I, _P],
    "osm_add_rowvec": [_P, _LL, _P, _LL, _I, _LL, _I, _P],
    "osm_posterior": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "osm_posterior_typed": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P],
    "osm_clamp_bwd": [_P, _P, _F, _F, _LL, _P],
    "osm_phys_nblk": [_I],
    "osm_phys_reduce": [C.POINTER(PhysDesc), _P, _P, _P, _P, _P],
    "osm_phys_finalize": [C.POINTER(PhysDesc), _P, _P, _P, _I, _P, _P, _P],
    "osm_phys_grad": [C.POINTER(PhysDesc), _P, _P, _P, _P, _P, _P],
    "osm_phys_optimize": [C.POINTER(PhysDesc), _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "osm_posterior_bwd": [_P, _P, _P, _I, _I, _P],
    "osm_guide_update": [_P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P],
    "osm_guide_update_rng": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _I, _I, C.c_ulonglong, _P, _I, _I, _I, _P],
    "osm_randn": [_P, _I, _LL, C.c_ulonglong, _P, _I, _I, _I, _P],
    "osm_philox_raw": [_P, _LL, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, _P],
    "osm_ddim_update": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _P],
    "osm_fetch_coefs": [_P, _I, _P, _I, _P, _P, _I, _P],
    "osm_ancestral_step": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "osm_version": [],
}
# fp16-storage family (activations as IEEE half, `_h` suffix): same argument lists
for _n in ("osm_conv2d_nhwc", "osm_gn_stats", "osm_gn_apply", "osm_gn_fwd", "osm_gn_prep", "osm_gn_bwd", "osm_gn_bwd_apply", "osm_pool2x2",
           "osm_resample_pair", "osm_upsample2x", "osm_stride2_pick", "osm_stride2_place", "osm_add_rowvec", "osm_nchw_to_nhwc", "osm_nhwc_to_nchw", "osm_copy2d"):
    _SIGS[_n + "_h"] = _SIGS[_n]
_SIGS["osm_half_to_f32"] = [_P, _LL, _P, _LL, _LL, _I, _P]
_SIGS["osm_f32_to_half"] = [_P, _LL, _P, _LL, _LL, _I, _P]
EXPORTS = sorted(list(_SIGS) + ["osm_last_error", "osm_packed_weight_elems", "osm_winograd_weight_elems"])

_lib = None
_lock = threading.Lock()


def load():
    """Load the shared library (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise OsmosisHipError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C osmosis_diffusion_code_amd/csrc` (hipcc, --offload-arch=gfx950). "
                    "There is no CPU fallback for the product path.")
            lib = C.CDLL(LIB_PATH)
            for name, argtypes in _SIGS.items():
                fn = getattr(lib, name)
                fn.argtypes = argtypes
                fn.restype = C.c_int
            lib.osm_packed_weight_elems.argtypes = [_I, _I, _I, _I, _I]
            lib.osm_packed_weight_elems.restype = C.c_longlong
            lib.osm_winograd_weight_elems.argtypes = [_I, _I, _I, _I]
            lib.osm_winograd_weight_elems.restype = C.c_longlong
            lib.osm_last_error.argtypes = []
            lib.osm_last_error.restype = C.c_char_p
            _lib = lib
    return _lib


# ----------------------------------------------------------------------------- recording
class Recorder:
    """Collects (cfunc, args) of every kernel call made while active; keeps referenced tensors alive."""

    def __init__(self):
        self.calls: List[Tuple] = []
        self.keep: List = []

    def __enter__(self):
        _state.recorders.append(self)
        return self

    def __exit__(self, *exc):
        _state.recorders.pop()
        return False

    @staticmethod
    def suspended():
        """Context manager: calls made inside run now and are NOT recorded (one-off work a recording pass triggers, e.g. a weight
        image packed on first use -- it must not be replayed with every step)."""
        class _Suspend:
            def __enter__(self_):
                self_.saved, _state.recorders = _state.recorders, []

            def __exit__(self_, *exc):
                _state.recorders = self_.saved
                return False
        return _Suspend()

    def replay(self):
        """Launch by launch on the stream the calls were recorded on."""
        for c in self.calls:
            fn, args = c[0], c[1]
            rc = fn(*args)
            if rc != 0:
                raise OsmosisHipError(f"{fn.__name__} failed ({rc}): {load().osm_last_error().decode()}")

    def to_graph(self):
        """Capture the recorded launches into a hipGraph (torch.cuda.CUDAGraph).  Every recorded call ends
        with its stream argument, which is re-targeted to the capture stream."""
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=cap):
            cs = torch.cuda.current_stream()
            for c in self.calls:
                fn, args = c[0], c[1]
                rc = fn(*args[:-1], cs.cuda_stream)
                if rc != 0:
                    raise OsmosisHipError(f"{fn.__name__} failed during capture ({rc}): "
                                          f"{load().osm_last_error().decode()}")
        return g

    def replay_timed(self, select):
        """Replay with HIP events around the selected launches (events are recorded on the stream
        the kernels run on).  `select(fn_name, args)` returns a tag or None.  Returns [(tag, ms)]."""
        marks = []
        for c in self.calls:
            fn, args = c[0], c[1]
            tag = select(fn.__name__, args)
            if tag is None:
                fn(*args)
                continue
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args)
            e1.record()
            marks.append((tag, e0, e1))
        torch.cuda.synchronize()
        return [(tag, e0.elapsed_time(e1)) for tag, e0, e1 in marks]

    def __len__(self):
        return len(self.calls)


class _State(threading.local):
    def __init__(self):
        self.recorders: List[Recorder] = []
        self.stream: Optional[int] = None


_state = _State()


def current_stream_ptr() -> int:
    if _state.stream is not None:
        return _state.stream
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args, keep=()):
    """Invoke a status-returning entry point; raise on failure; record if a Recorder is active."""
    lib = load()
    fn = getattr(lib, name)
    rc = fn(*args)
    if rc != 0:
        raise OsmosisHipError(f"{name} failed ({rc}): {lib.osm_last_error().decode()}")
    for r in _state.recorders:
        r.calls.append((fn, args))
        r.keep.extend(keep)
        r.keep.extend(a for a in args if isinstance(a, C.Structure) or hasattr(a, "_obj"))


def query(name: str, *args) -> int:
    """Value-returning helpers (osm_splitk_hint, osm_gn_nchunk, osm_phys_nblk, osm_version)."""
    return getattr(load(), name)(*args)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device address of a CUDA fp32/int32 tensor (None -> NULL).  Fails loudly on CPU tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise OsmosisHipError("osmosis_hip kernels need CUDA(HIP) tensors; got a CPU tensor "
                              "(there is no CPU fallback on the product path)")
    # int16 = packed bf16 / fp16 weight planes; float16 = activations of the fp16-storage family
    if t.dtype not in (torch.float32, torch.int32, torch.int16, torch.float16):
        raise OsmosisHipError(f"osmosis_hip kernels take fp32 (or fp16-family half) tensors; got {t.dtype}")
    return t.data_ptr()
