"""ctypes binding of libwnoise_shard.so (include/wnoise_shard.h): z-slab bounds and the one RCCL gather of the sharded
dense-grid path.  Loaded on first use only (it pulls in librccl)."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libwnoise_shard.so")
WN_COMM_ID_BYTES = 128

_vp, _i, _sz = C.c_void_p, C.c_int, C.c_size_t
_ip = C.POINTER(C.c_int)
SIGNATURES = {
    "wn_shard_last_error": (C.c_char_p, []),
    "wn_shard_bounds": (_i, [_i, _i, _i, _ip, _ip]),
    "wn_comm_unique_id": (_i, [_vp]),
    "wn_comm_create": (_i, [C.POINTER(_vp), _i, _i, _vp]),
    "wn_comm_rank": (_i, [_vp, _ip, _ip]),
    "wn_comm_destroy": (None, [_vp]),
    "wn_gather_volume": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
}
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (the sharded path has no fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"wnoise_shard error {rc}: {load().wn_shard_last_error().decode()}")
