"""ctypes binding of libwnoise_hip.so (include/wnoise.h).

The library is the product: if it is missing or fails to load this module raises, it never
substitutes a CPU implementation.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# WN_HIP_LIBRARY: another build of the SAME library (the -DWN_TUNE_ENV development build, `make tune`); still no fallback
LIB_PATH = os.environ.get("WN_HIP_LIBRARY") or os.path.join(HERE, "libwnoise_hip.so")

WN_OK, WN_ERR_INVALID, WN_ERR_NO_DEVICE, WN_ERR_HIP, WN_ERR_ALLOC = range(5)
WN_Z_LATTICE, WN_Z_CONST = 0, 1
WN_GRID_DEFAULT, WN_GRID_EXACT = 0, 1


class WnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"wnoise error {code}: {msg}")
        self.code = code


class wn_grid(C.Structure):
    _fields_ = [("den", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32), ("z0", C.c_int32),
                ("z1", C.c_int32), ("base_range", C.c_float), ("octave_scale", C.c_float),
                ("post_scale", C.c_float), ("z_mode", C.c_int32), ("z_const", C.c_float),
                ("out_scale", C.c_float), ("flags", C.c_int32)]


# name -> (restype, argtypes); every symbol include/wnoise.h declares.
_vp, _sz, _i, _u32, _f, _d = C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)
_gp = C.POINTER(wn_grid)
SIGNATURES = {
    "wn_last_error": (C.c_char_p, []),
    "wn_version": (C.c_char_p, []),
    "wn_device_count": (_i, [C.POINTER(C.c_int)]),
    "wn_device_set": (_i, [_i]),
    "wn_device_get": (_i, [C.POINTER(C.c_int)]),
    "wn_device_info": (_i, [C.c_char_p, _sz, C.POINTER(C.c_int), C.POINTER(_sz)]),
    "wn_dev_alloc": (_i, [_pp, _sz]),
    "wn_dev_free": (_i, [_vp]),
    "wn_host_alloc_mapped": (_i, [_pp, _pp, _sz]),
    "wn_host_free_mapped": (_i, [_vp]),
    "wn_copy_h2d": (_i, [_vp, _vp, _sz, _vp]),
    "wn_copy_d2h": (_i, [_vp, _vp, _sz, _vp]),
    "wn_stream_sync": (_i, [_vp]),
    "wn_timer_create": (_i, [_pp]),
    "wn_timer_start": (_i, [_vp, _vp]),
    "wn_timer_stop": (_i, [_vp, _vp]),
    "wn_timer_elapsed_ms": (_i, [_vp, C.POINTER(C.c_float)]),
    "wn_timer_destroy": (None, [_vp]),
    "wn_gaussian_fill": (_i, [_u32, _sz, _vp]),
    "wn_perlin_permutation": (_i, [_u32, _vp]),
    "wn_tile_even_size": (_i, [_i]),
    "wn_tile_create": (_i, [_i, _i, _vp, _pp]),
    "wn_tile_generate": (_i, [_i, _i, _u32, _pp]),
    "wn_tile_generate_from_field": (_i, [_i, _i, _vp, _pp]),
    "wn_tile_size": (_i, [_vp]),
    "wn_tile_dims": (_i, [_vp]),
    "wn_tile_count": (_sz, [_vp]),
    "wn_tile_device_ptr": (_vp, [_vp]),
    "wn_tile_download": (_i, [_vp, _vp]),
    "wn_tile_destroy": (None, [_vp]),
    "wn_perm_create": (_i, [_vp, _pp]),
    "wn_perm_create_seeded": (_i, [_u32, _pp]),
    "wn_perm_download": (_i, [_vp, _vp]),
    "wn_perm_destroy": (None, [_vp]),
    "wn_eval3d_grid": (_i, [_vp, _gp, _vp, _vp]),
    "wn_eval2d_grid": (_i, [_vp, _gp, _vp, _vp]),
    "wn_eval3d_projected_grid": (_i, [_vp, _gp, C.POINTER(C.c_float), _vp, _vp]),
    "wn_multiband3d_grid": (_i, [_vp, _gp, _f, _i, _i, C.POINTER(C.c_float), _f, _vp, _vp]),
    "wn_perlin_grid": (_i, [_vp, _gp, _vp, _vp]),
    "wn_perlin_turb_grid": (_i, [_vp, _gp, _i, _vp, _vp]),
    "wn_perlin_fractal_grid": (_i, [_vp, _gp, _vp, _vp]),
    "wn_eval3d_points": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "wn_eval2d_points": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "wn_eval3d_projected_points": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "wn_multiband3d_points": (_i, [_vp, _vp, _sz, _f, _i, _i, C.POINTER(C.c_float), _f, _vp, _vp]),
    "wn_multiband3d_projected_points": (_i, [_vp, _vp, _vp, _i, _sz, _f, _i, _i, C.POINTER(C.c_float), _f, _vp, _vp]),
    "wn_perlin_points": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "wn_perlin_points_vec3": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "wn_perlin_turb_points": (_i, [_vp, _vp, _sz, _i, _vp, _vp]),
    "wn_perlin_fractal_points": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "wn_wavelet_texture_points": (_i, [_vp, _i, _d, _i, _vp, _vp, _sz, _vp, _vp]),
    "wn_noise_texture_points": (_i, [_vp, _d, _i, _vp, _vp, _sz, _vp, _vp]),
    "wn_scalar_eval3d": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wn_scalar_eval2d": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wn_scalar_eval3d_projected": (_i, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wn_scalar_perlin": (_i, [_vp, _d, _d, _d, C.POINTER(C.c_double)]),
    "wn_scalar_perlin_vec3": (_i, [_vp, C.POINTER(C.c_float), _i, _i, C.POINTER(C.c_double)]),
    "wn_scalar_wavelet_texture": (_i, [_vp, _i, _d, _i, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wn_scalar_noise_texture": (_i, [_vp, _d, _i, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "wn_scalar_stats": (_i, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "wn_scalar_shutdown": (_i, []),
}

_lib = None


def load():
    """Load libwnoise_hip.so or raise: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C {HERE}` (or "
            "__graft_entry__.build()).  This package has no CPU implementation.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != WN_OK:
        raise WnError(rc, load().wn_last_error().decode("utf-8", "replace"))
