"""ctypes binding of libmsf2.so (declarations: include/msf2.h).

The shared library is the product's only compute path.  It is built in-tree by
``__graft_entry__.build()`` / ``speech_separation_amd/build.py``; when it is
missing or lacks a symbol the import fails loudly -- there is no CPU or
PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSF2_LIB", os.path.join(_HERE, "lib", "libmsf2.so"))   # override: A/B builds

TCN_DEPTH = 8
QK_DIM = 128
GROUP = 256
PREC_FP32 = 0
PREC_BF16 = 1
TIMED_TCN_POINTWISE = 0   # msf2_timing_begin_kernel: the two-product contraction of a TCN block
TIMED_HIDDEN_PROJ = 1     # ... the token-shifted to_hidden | to_qk projection of a FLASH layer
REC_TCN = 0      # GatedTCNBlock after each FLASH layer (what the reference revision instantiates)
REC_FSMN = 1     # Gated_FSMN_Block_Dilated (the upstream architecture's block, present in the reference)

fp = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [("dim", C.c_int32), ("num_layers", C.c_int32), ("enc_kernel", C.c_int32),
                ("num_spks", C.c_int32), ("precision", C.c_int32), ("recurrent", C.c_int32), ("fsmn_inner", C.c_int32)]


class FsmnWeights(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("prelu1", C.c_float), ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p),
                ("w_uv", C.c_void_p), ("c1_uv", C.c_void_p), ("c2_uv", C.c_void_p), ("dw_uv", C.c_void_p),
                ("w_lin", C.c_void_p), ("b_lin", C.c_void_p), ("w_proj", C.c_void_p), ("w_c1", C.c_void_p),
                ("in1_g", C.c_void_p), ("in1_b", C.c_void_p), ("pr1", C.c_void_p), ("w_c2", C.c_void_p),
                ("in2_g", C.c_void_p), ("in2_b", C.c_void_p), ("pr2", C.c_void_p), ("ln2_g", C.c_void_p),
                ("ln2_b", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p)]


class TcnWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dw_w", "dw_b", "w_gate", "b_gate", "w_pw", "c1", "c2")]


class LayerWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "w_hq", "b_hq", "dw_hq", "os_gamma", "os_beta", "rot_freqs", "w_out", "b_out", "dw_out",
        "w_gg", "b_gg", "w_in", "c1_in", "c2_in", "w_po", "b_po")] + [("tcn", TcnWeights * TCN_DEPTH),
                                                                           ("fsmn", C.POINTER(FsmnWeights))]


class Weights(C.Structure):
    _fields_ = [("enc_w", C.c_void_p), ("w_ce", C.c_void_p), ("c1_ce", C.c_void_p), ("c2_ce", C.c_void_p),
                ("pe_inv_freq", C.c_void_p), ("pe_scale", C.c_float), ("prelu_a", C.c_float),
                ("fin_ln_w", C.c_void_p), ("fin_ln_b", C.c_void_p), ("fin_gn_w", C.c_void_p),
                ("fin_gn_b", C.c_void_p), ("w_co", C.c_void_p), ("b_co", C.c_void_p), ("w_o", C.c_void_p),
                ("b_o", C.c_void_p), ("w_og", C.c_void_p), ("b_og", C.c_void_p), ("w_cd", C.c_void_p),
                ("dec_w", C.c_void_p), ("layers", C.POINTER(LayerWeights))]


class Taps(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("enc", "x0", "flash0", "tcn0", "mdl", "mask0", "mask1")]


# every symbol include/msf2.h declares: name -> (restype, argtypes)
_vp, _i, _sz, _f = C.c_void_p, C.c_int, C.c_size_t, C.c_float
SYMBOLS = {
    "msf2_strerror": (C.c_char_p, [_i]),
    "msf2_last_hip_error": (_i, []),
    "msf2_version": (_i, []),
    "msf2_sizeof": (_i, [_i]),
    "msf2_timing_begin": (_i, []),
    "msf2_timing_begin_kernel": (_i, [_i]),
    "msf2_timing_end": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
    "msf2_debug_kernel_choice": (_i, [C.c_char_p, C.c_char_p]),
    "msf2_num_frames": (_i, [_i]),
    "msf2_workspace_bytes": (_sz, [C.POINTER(Config), _i, _i]),
    "msf2_forward": (_i, [C.POINTER(Config), C.POINTER(Weights), _vp, _vp, _i, _i, _vp, _sz, C.POINTER(Taps), _vp]),
    "msf2_stats_ws_bytes": (_sz, [_i, _i]),
    "msf2_encoder": (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _i, _vp, _sz, _vp]),
    "msf2_dwconv17": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "msf2_linear_ws_bytes": (_sz, [_i, _i]),
    "msf2_linear": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "msf2_flash_attention_ws_bytes": (_sz, [_i, _i, _i]),
    "msf2_flash_attention": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "msf2_gated_tcn_ws_bytes": (_sz, [_i, _i, _i]),
    "msf2_gated_tcn": (_i, [_vp, C.POINTER(LayerWeights), _i, _i, _i, _i, _vp, _sz, _vp]),
    "msf2_tcn_pointwise": (_i, [_vp, _vp, _vp, C.POINTER(TcnWeights), _vp, _i, _i, _i, _i, _vp]),
    "msf2_decoder_ws_bytes": (_sz, [_i, _i]),
    "msf2_gated_fsmn_ws_bytes": (_sz, [_i, _i, _i, _i]),
    "msf2_gated_fsmn": (_i, [_vp, C.POINTER(FsmnWeights), _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "msf2_pit_si_snr_ws_bytes": (_sz, [_i, _i]),
    "msf2_pit_si_snr": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "msf2_pit_si_snr_noalloc": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "msf2_decoder": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
}


class Msf2Error(RuntimeError):
    pass


_lib = None


def load():
    """Load libmsf2.so once and type every entry point.  Raises if the library
    or any declared symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Msf2Error(f"{LIB_PATH} not found: build it with `python -m speech_separation_amd.build` "
                        "(hipcc, gfx950). There is no fallback compute path.")
    # torch ships its own libamdhip64.so.7; load it first so that libmsf2.so binds to
    # the SAME HIP runtime (streams and device pointers are shared with torch).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Msf2Error(f"libmsf2.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for which, st in enumerate((Config, TcnWeights, LayerWeights, Weights, Taps, FsmnWeights)):
        if lib.msf2_sizeof(which) != C.sizeof(st):
            raise Msf2Error(f"ABI mismatch: {st.__name__} is {C.sizeof(st)} bytes here, "
                            f"{lib.msf2_sizeof(which)} in libmsf2.so")
    _lib = lib
    return lib


def check(code: int, what: str = "msf2"):
    if code != 0:
        lib = load()
        msg = lib.msf2_strerror(code).decode()
        raise Msf2Error(f"{what}: {msg} (code {code}, hip error {lib.msf2_last_hip_error()})")
