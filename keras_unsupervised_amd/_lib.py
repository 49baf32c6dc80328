"""ctypes binding of ``libkurbm.so`` (C ABI: ``include/kurbm.h``).

There is no CPU fallback: if the shared library is missing or a call fails the
product raises.  ``load()`` only dlopens the library (works without a GPU, so the CPU
test tier can check the exported symbols); a context needs a gfx950 device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KURBM_LIB selects another build of the same ABI (the s_memtime diagnostic build); never a fallback
LIB_PATH = os.environ.get("KURBM_LIB") or os.path.join(_HERE, "csrc", "libkurbm.so")

KURBM_OK = 0
ACT_SIGMOID, ACT_RELU, ACT_LINEAR = 0, 1, 2
NOISE_NONE, NOISE_BERNOULLI, NOISE_GAUSSIAN = 0, 1, 2
WHICH_W, WHICH_BH, WHICH_BV, WHICH_ALL = 1, 2, 4, 7


class KurbmError(RuntimeError):
    pass


class Params(C.Structure):
    _fields_ = [("n_vis", C.c_int32), ("n_hid", C.c_int32), ("ldw", C.c_int32), ("_pad", C.c_int32),
                ("W", C.c_void_p), ("b_h", C.c_void_p), ("b_v", C.c_void_p)]


class Rng(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("row0", C.c_uint64), ("stream_id", C.c_uint32), ("step", C.c_uint32)]


class CdOpts(C.Structure):
    _fields_ = [("k", C.c_int32), ("mode", C.c_int32), ("lr", C.c_float), ("apply", C.c_int32),
                ("delta_out", C.c_void_p), ("v_chain", C.c_void_p),
                ("seed", C.c_uint64), ("row0", C.c_uint64), ("step", C.c_uint32), ("chain", C.c_uint32),
                ("v_planes", C.c_void_p), ("v_planes_stride", C.c_uint64)]


_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_PP, _RP, _OP = C.POINTER(Params), C.POINTER(Rng), C.POINTER(CdOpts)

# name -> (restype, argtypes); every symbol include/kurbm.h declares
SIGNATURES = {
    "kurbm_abi_version": (_i, []),
    "kurbm_last_error": (C.c_char_p, []),
    "kurbm_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "kurbm_ctx_destroy": (None, [_vp]),
    "kurbm_ctx_set_option": (_i, [_vp, C.c_char_p, _i]),
    "kurbm_ctx_status": (_i, [_vp, C.POINTER(_i)]),
    "kurbm_philox_uniform": (_i, [_vp, _vp, _i, _i, _i, _RP, _vp]),
    "kurbm_half_step_vh": (_i, [_vp, _PP, _vp, _i, _i, _i, _i, _RP, _vp, _vp, _i, _vp]),
    "kurbm_half_step_hv": (_i, [_vp, _PP, _vp, _i, _i, _i, _i, _RP, _vp, _vp, _i, _vp]),
    "kurbm_half_step_vh_dbg": (_i, [_vp, _PP, _vp, _i, _i, _i, _i, _RP, _vp, _vp, _vp, _i, _vp]),
    "kurbm_half_step_hv_dbg": (_i, [_vp, _PP, _vp, _i, _i, _i, _i, _RP, _vp, _vp, _vp, _i, _vp]),
    "kurbm_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "kurbm_cd_step": (_i, [_vp, _PP, _vp, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_cd_epoch": (_i, [_vp, _PP, _vp, _i, _i, _i, _OP, _vp, _sz, _vp]),
    "kurbm_apply_delta": (_i, [_vp, _PP, _vp, C.c_float, _i, _vp]),
    "kurbm_free_energy": (_i, [_vp, _PP, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "kurbm_outer_delta": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "kurbm_outer_partial": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "kurbm_bf16_mirror_bytes": (_sz, [_vp, _i, _i]),
    "kurbm_bf16_mirror_refresh": (_i, [_vp, _PP, _vp, _sz, _vp]),
    "kurbm_bf16_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "kurbm_cd_step_bf16": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_half_step_bf16": (_i, [_vp, _PP, _vp, _sz, _i, _vp, _i, _i, _i, _i, _RP, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "kurbm_x3_mirror_bytes": (_sz, [_vp, _i, _i]),
    "kurbm_x3_mirror_refresh": (_i, [_vp, _PP, _vp, _sz, _vp]),
    "kurbm_x3_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i]),
    "kurbm_cd_step_x3": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_cd_step_x3_stage": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _i, _i, _vp, _sz, _vp]),
    "kurbm_free_energy_x3": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "kurbm_x3_dump_plane": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i, _vp]),
    "kurbm_score_x3": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _vp, _vp, _vp, _sz, _vp]),
    "kurbm_cd_epoch_x3": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _i, _OP, _vp, _sz, _vp]),
    "kurbm_cd_chain_x3": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _vp, _sz, _vp]),
    "kurbm_x3_stats_rows": (_i, [_vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _i, _i, _vp, _sz, _vp]),
    "kurbm_x3_apply_delta": (_i, [_vp, _PP, _vp, _sz, _vp, C.c_float, _i, _vp]),
    "kurbm_half_step_x3": (_i, [_vp, _PP, _vp, _sz, _i, _vp, _i, _i, _i, _i, _i, _RP, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "kurbm_x3_planes_bytes": (_sz, [_vp, _i, _i, _i]),
    "kurbm_x3_convert_rows": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "kurbm_bf16_exact": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "kurbm_comm_unique_id": (_i, [_vp, _sz]),
    "kurbm_comm_init_rank": (_i, [_i, _i, _i, _vp, _sz, C.POINTER(_vp)]),
    "kurbm_comm_init_all": (_i, [_i, C.POINTER(_i), C.POINTER(_vp)]),
    "kurbm_comm_count": (_i, [_vp]),
    "kurbm_comm_rank": (_i, [_vp]),
    "kurbm_comm_destroy": (None, [_vp]),
    "kurbm_allreduce_sum_f32": (_i, [_vp, _vp, _sz, _vp]),
    "kurbm_cd_step_x3_dp": (_i, [_vp, _vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_cd_step_bf16_dp": (_i, [_vp, _vp, _PP, _vp, _sz, _vp, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_cd_step_small": (_i, [_vp, _PP, _vp, _i, _i, _OP, _i, _vp, _sz, _vp]),
    "kurbm_cd_epoch_small": (_i, [_vp, _PP, _vp, _i, _i, _i, _OP, _vp, _sz, _vp]),
    "kurbm_score_small": (_i, [_vp, _PP, _vp, _i, _i, _OP, _vp, _vp, _vp, _sz, _vp]),
    "kurbm_cd_epoch_small_scored": (_i, [_vp, _PP, _vp, _i, _i, _i, _OP, _i, _vp, _vp, _sz, _vp]),
    "kurbm_peer_handle_bytes": (_sz, []),
    "kurbm_peer_create": (_i, [_i, _i, _i, _i, _i, C.POINTER(_vp)]),
    "kurbm_peer_handle": (_i, [_vp, _vp, _sz]),
    "kurbm_peer_connect": (_i, [_vp, _vp, _sz]),
    "kurbm_peer_ranks": (_i, [_vp]),
    "kurbm_peer_allreduce_sum_f32": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "kurbm_cd_step_x3_peer": (_i, [_vp, _vp, _PP, _vp, _sz, _vp, _i, _i, _i, _OP, _vp, _sz, _vp]),
    "kurbm_peer_destroy": (None, [_vp]),
}

ABI_VERSION = 5
V_BINARY = 0x10          # KURBM_V_BINARY: OR into v_pieces = 1 for 0/1 data
PLANE_H_POS, PLANE_H_POS_T, PLANE_V_NEG, PLANE_V_NEG_T, PLANE_H_NEG_T = range(5)   # kurbm_x3_dump_plane
UNIQUE_ID_BYTES = 128

_lib = None


def load():
    """dlopen libkurbm.so and type its entry points.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KurbmError(
            "HIP library %s is missing; build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C keras_unsupervised_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    # the ABI guard FIRST: a stale build lacks the newer symbols, and a bare AttributeError would say nothing about rebuilding
    ver = getattr(lib, "kurbm_abi_version", None)
    if ver is None:
        raise KurbmError("%s does not export kurbm_abi_version: not a libkurbm build; rebuild it "
                         "(make -C keras_unsupervised_amd/csrc)" % LIB_PATH)
    ver.restype, ver.argtypes = SIGNATURES["kurbm_abi_version"]
    if ver() != ABI_VERSION:
        raise KurbmError("%s implements ABI %d, this package binds ABI %d: rebuild it (make -C keras_unsupervised_amd/csrc)"
                         % (LIB_PATH, ver(), ABI_VERSION))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise KurbmError("%s (ABI %d) does not export %s: rebuild it (make -C keras_unsupervised_amd/csrc)"
                             % (LIB_PATH, ABI_VERSION, name))
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(code):
    if code != KURBM_OK:
        msg = load().kurbm_last_error()
        raise KurbmError("kurbm error %d: %s" % (code, msg.decode() if msg else "?"))


class Context:
    """One kurbm_ctx per device."""

    _cache = {}

    def __init__(self, device_index):
        lib = load()
        h = _vp()
        check(lib.kurbm_ctx_create(int(device_index), C.byref(h)))
        self.handle = h
        self.device_index = int(device_index)
        self.lib = lib

    def status(self):
        """Sticky status bits of the context's kernels, read back and cleared (synchronises with the device): a device-side wait
        ran into its bound and skipped its work -- bit 1 the peer exchange, bit 2 the small step's grid barrier (include/kurbm.h)."""
        bits = _i(0)
        check(self.lib.kurbm_ctx_status(self.handle, C.byref(bits)))
        return int(bits.value)

    def set_option(self, name, value):
        """Set one experiment knob (its KURBM_* environment name) on this context; -1 = automatic."""
        check(self.lib.kurbm_ctx_set_option(self.handle, name.encode(), int(value)))

    @classmethod
    def get(cls, device_index):
        ctx = cls._cache.get(device_index)
        if ctx is None:
            ctx = cls._cache[device_index] = cls(device_index)
        return ctx
