"""ctypes loaders for the TEST-ONLY checkers under oracle/.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product package never does.

  * ``oracle.lib()``  -> liboracle.so, the plain-C restatement (oracle/wn_oracle.c)
  * ``oracle.ref()``  -> _ref/libwnref.so, the real reference compiled by oracle/Makefile
                         (None when it has not been built: /root/reference is absent on the
                         GPU box, so only a prebuilt copy can be there)
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

_lib = None
_ref = None
_ref_tried = False


def build(quiet=True):
    """(Re)build liboracle.so and, when /root/reference is present, _ref/libwnref.so."""
    out = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    sz, i, u32, f, d, vp = C.c_size_t, C.c_int, C.c_uint32, C.c_float, C.c_double, C.c_void_p

    def sig(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("wno_perlin_perm", None, u32, _i32p)
    sig("wno_tile_size", i, i)
    sig("wno_generate_tile2d", None, i, u32, _f32p)
    sig("wno_generate_tile3d", None, i, u32, _f32p)
    sig("wno_filter_tile2d", None, i, _f32p, _f32p)
    sig("wno_filter_tile3d", None, i, _f32p, _f32p)
    sig("wno_mod", i, i, i)
    sig("wno_evaluate2d", f, vp, sz, _f32p)
    sig("wno_evaluate3d", f, vp, sz, _f32p)
    sig("wno_evaluate3d_projected", f, vp, sz, _f32p, _f32p)
    sig("wno_perlin_noise", d, _i32p, d, d, d)
    sig("wno_perlin_fractal", d, _i32p, f, f, f)
    sig("wno_perlin_turb", d, _i32p, f, f, f, i)
    sig("wno_multiband3d", f, vp, sz, _f32p, f, i, i, _f32p, f)
    sig("wno_multiband3d_projected", f, vp, sz, _f32p, _f32p, f, i, i, _f32p, f)
    sig("wno_noise_texture_value", f, _i32p, d, i, _f32p)
    sig("wno_wavelet_texture_value", f, vp, sz, i, d, i, _f32p)
    sig("wno_grid_wavelet2d", None, _f32p, sz, i, i, _f32p)
    sig("wno_grid_wavelet3d_sliced", None, _f32p, sz, i, i, _f32p)
    sig("wno_grid_wavelet3d_projected", None, _f32p, sz, i, i, _f32p)
    sig("wno_grid_perlin2d", None, _i32p, i, i, _f32p)
    sig("wno_grid_perlin3d_sliced", None, _i32p, i, i, _f32p)
    sig("wno_grid_wavelet3d_volume", None, _f32p, sz, i, i, i, i, i, i, _f32p)
    sig("wno_grid_multiband3d_volume", None, _f32p, sz, i, i, i, i, i, f, i, i, _f32p, f, _f32p)
    sig("wno_grid_perlin_volume", None, _i32p, i, i, i, i, i, i, _f32p)
    sig("wno_grid_turb_volume", None, _i32p, i, i, i, i, i, i, _f32p)
    sig("wno_fnv1a64", C.c_uint64, vp, sz)
    _lib = L
    return L


def ref():
    """The compiled reference, or None when oracle/_ref/libwnref.so does not exist."""
    global _ref, _ref_tried
    if _ref_tried:
        return _ref
    _ref_tried = True
    path = os.path.join(HERE, "_ref", "libwnref.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/WaveletNoise.cpp"):
            build()
        if not os.path.exists(path):
            return None
    R = C.CDLL(path)
    sz, i, u, d, vp = C.c_size_t, C.c_int, C.c_uint, C.c_double, C.c_void_p

    def sig(name, res, *args):
        fn = getattr(R, name)
        fn.restype = res
        fn.argtypes = list(args)

    sig("ref_wn_new", vp, i, u)
    sig("ref_wn_delete", None, vp)
    sig("ref_wn_generate2d", None, vp)
    sig("ref_wn_generate3d", None, vp)
    sig("ref_wn_tile_size", i, vp)
    sig("ref_wn_coeff_count", sz, vp)
    sig("ref_wn_coeffs", None, vp, _f32p)
    sig("ref_wn_eval2d", None, vp, _f32p, sz, _f32p)
    sig("ref_wn_eval3d", None, vp, _f32p, sz, _f32p)
    sig("ref_wn_eval3d_projected", None, vp, _f32p, _f32p, sz, _f32p)
    sig("ref_wn_grid3d_volume", None, vp, i, i, i, i, i, i, _f32p)
    sig("ref_gaussian_stream", None, u, sz, _f32p)
    sig("ref_perlin_new", vp, u)
    sig("ref_perlin_new_default", vp)
    sig("ref_perlin_delete", None, vp)
    sig("ref_perlin_perm", None, vp, _i32p)
    sig("ref_perlin_noise", None, vp, _f64p, sz, _f64p)
    sig("ref_perlin_noise_vec3", None, vp, _f32p, sz, _f64p)
    sig("ref_perlin_fractal", None, vp, _f32p, sz, _f64p)
    sig("ref_PerlinNoise_new", vp, u)
    sig("ref_PerlinNoise_delete", None, vp)
    sig("ref_PerlinNoise_perm", None, vp, _i32p)
    sig("ref_PerlinNoise_noise", None, vp, _f64p, sz, _f64p)
    sig("ref_noise_texture_new", vp, d, i)
    sig("ref_wavelet_texture_new", vp, d, i, i)
    sig("ref_texture_delete", None, vp)
    sig("ref_texture_value", None, vp, _f32p, sz, _f32p)
    _ref = R
    return R


# ---- small numpy conveniences over liboracle (vectorised loops live in C where it matters) ----

def perlin_perm(seed):
    p = np.zeros(512, np.int32)
    lib().wno_perlin_perm(seed, p)
    return p


def tile2d(n, seed):
    n = lib().wno_tile_size(n)
    out = np.empty(n * n, np.float32)
    lib().wno_generate_tile2d(n, seed, out)
    return out


def tile3d(n, seed):
    n = lib().wno_tile_size(n)
    out = np.empty(n * n * n, np.float32)
    lib().wno_generate_tile3d(n, seed, out)
    return out


def _cptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def evaluate2d(coef, pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    L = lib()
    cnt = 0 if coef is None else coef.size
    return np.array([L.wno_evaluate2d(_cptr(coef), cnt, p) for p in pts], np.float32)


def evaluate3d(coef, pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    L = lib()
    cnt = 0 if coef is None else coef.size
    return np.array([L.wno_evaluate3d(_cptr(coef), cnt, p) for p in pts], np.float32)


def evaluate3d_projected(coef, pts, normals):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    L = lib()
    cnt = 0 if coef is None else coef.size
    return np.array([L.wno_evaluate3d_projected(_cptr(coef), cnt, p, n)
                     for p, n in zip(pts, normals)], np.float32)


def perlin_noise(perm, pts):
    pts = np.asarray(pts, np.float64).reshape(-1, 3)
    L = lib()
    return np.array([L.wno_perlin_noise(perm, *map(float, p)) for p in pts], np.float64)


def perlin_fractal(perm, pts):
    pts = np.asarray(pts, np.float32).reshape(-1, 3)
    L = lib()
    return np.array([L.wno_perlin_fractal(perm, *map(float, p)) for p in pts], np.float64)


def perlin_turb(perm, pts, depth):
    pts = np.asarray(pts, np.float32).reshape(-1, 3)
    L = lib()
    return np.array([L.wno_perlin_turb(perm, *map(float, p), depth) for p in pts], np.float64)


def multiband3d(coef, pts, s, first_band, nbands, w, var):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    w = np.ascontiguousarray(w, np.float32)
    L = lib()
    return np.array([L.wno_multiband3d(_cptr(coef), coef.size, p, s, first_band, nbands, w, var)
                     for p in pts], np.float32)


def multiband3d_projected(coef, pts, normals, s, first_band, nbands, w, var):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    normals = np.ascontiguousarray(np.broadcast_to(np.asarray(normals, np.float32).reshape(-1, 3), pts.shape))
    w = np.ascontiguousarray(w, np.float32)
    L = lib()
    return np.array([L.wno_multiband3d_projected(_cptr(coef), coef.size, p, n, s, first_band, nbands, w, var)
                     for p, n in zip(pts, normals)], np.float32)


def noise_texture_value(perm, scale, octave, pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    L = lib()
    return np.array([L.wno_noise_texture_value(perm, scale, octave, p) for p in pts], np.float32)


def wavelet_texture_value(coef, use_3d, scale, octave, pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 3)
    L = lib()
    cnt = 0 if coef is None else coef.size
    return np.array([L.wno_wavelet_texture_value(_cptr(coef), cnt, int(use_3d), scale, octave, p)
                     for p in pts], np.float32)


def grid_wavelet3d_volume(coef, den, nx, ny, z0, z1, octave):
    out = np.empty((z1 - z0) * ny * nx, np.float32)
    lib().wno_grid_wavelet3d_volume(coef, coef.size, den, nx, ny, z0, z1, octave, out)
    return out.reshape(z1 - z0, ny, nx)


def grid_multiband3d_volume(coef, den, nx, ny, z0, z1, s, first_band, nbands, w, var):
    out = np.empty((z1 - z0) * ny * nx, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    lib().wno_grid_multiband3d_volume(coef, coef.size, den, nx, ny, z0, z1, s, first_band,
                                      nbands, w, var, out)
    return out.reshape(z1 - z0, ny, nx)


def grid_perlin_volume(perm, den, nx, ny, z0, z1, octave):
    out = np.empty((z1 - z0) * ny * nx, np.float32)
    lib().wno_grid_perlin_volume(perm, den, nx, ny, z0, z1, octave, out)
    return out.reshape(z1 - z0, ny, nx)


def grid_turb_volume(perm, den, nx, ny, z0, z1, depth):
    out = np.empty((z1 - z0) * ny * nx, np.float32)
    lib().wno_grid_turb_volume(perm, den, nx, ny, z0, z1, depth, out)
    return out.reshape(z1 - z0, ny, nx)


def fnv1a64(arr):
    b = np.ascontiguousarray(arr)
    return int(lib().wno_fnv1a64(b.ctypes.data_as(C.c_void_p), b.nbytes))
