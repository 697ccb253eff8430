"""Python mirror of the reference's noise interface over the C ABI (include/wnoise.h).

Class and method names follow the reference (WaveletNoise.h:20-41, perlin.h:14-91,
experient/PerlinNoise.hpp:9-61, texture.h:14-115, experient/main.cpp:11-129) so the parity
tests read like calls into the reference.  Every evaluation runs on the GPU through
libwnoise_hip.so: scalar calls are batches of one.  torch is used for device memory and
streams only.
"""
import ctypes as C
import math
import sys
from dataclasses import dataclass

import numpy as np
import torch

from . import _capi
from ._capi import check, wn_grid, WN_GRID_DEFAULT, WN_GRID_EXACT, WN_Z_CONST, WN_Z_LATTICE

_lib = _capi.load()


# ---- plumbing -------------------------------------------------------------------------------
def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(x, dtype):
    """array-like / tensor -> contiguous CUDA tensor of `dtype` (no copy when already so)."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=_NP[dtype])))
    return t.to(device="cuda", dtype=dtype).contiguous()


_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.uint8: np.uint8}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _is_scalar_point(p, width):
    if isinstance(p, torch.Tensor):
        return p.dim() == 1 and p.numel() == width
    a = np.asarray(p)
    return a.ndim == 1 and a.size == width


def device_info():
    name = C.create_string_buffer(256)
    cus, hbm = C.c_int(0), C.c_size_t(0)
    check(_lib.wn_device_info(name, 256, C.byref(cus), C.byref(hbm)))
    return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}


class HipTimer:
    """HIP events recorded on the stream the kernels run on (torch's current stream)."""

    def __init__(self):
        self._h = C.c_void_p()
        check(_lib.wn_timer_create(C.byref(self._h)))

    def start(self):
        check(_lib.wn_timer_start(self._h, _stream()))

    def stop(self):
        check(_lib.wn_timer_stop(self._h, _stream()))

    def elapsed_ms(self):
        ms = C.c_float(0)
        check(_lib.wn_timer_elapsed_ms(self._h, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.wn_timer_destroy(self._h)
            self._h = None


# ---- WaveletNoise (WaveletNoise.h:20-59) ----------------------------------------------------
class WaveletNoise:
    def __init__(self, tileSize, seed=0):
        self.tileSizeN = _lib.wn_tile_even_size(int(tileSize))
        if self.tileSizeN != tileSize:  # WaveletNoise.cpp:22-25
            print(f"Warning: Tile size adjusted to {self.tileSizeN} (must be even)", file=sys.stderr)
        self.randomSeed = int(seed) & 0xFFFFFFFF
        self._drawn = 0      # Gaussian values consumed so far: the rng is a member (WaveletNoise.h:47)
        self._tile = None    # no coefficients yet: evaluate* return 0 (WaveletNoise.cpp:112,186,219)

    # -- tile management
    def _set_tile(self, handle):
        self._free()
        self._tile = handle

    def _free(self):
        if getattr(self, "_tile", None):
            _lib.wn_tile_destroy(self._tile)
        self._tile = None

    def __del__(self):
        self._free()

    def _handle(self, dims):
        """Tile handle for evaluation; an un-generated object evaluates an empty tile."""
        if self._tile is None:
            h = C.c_void_p()
            check(_lib.wn_tile_create(0, dims, None, C.byref(h)))
            self._tile = h
        return self._tile

    def _generate(self, dims):
        n = self.tileSizeN
        count = n ** dims
        # continue the member rng's stream: draw `_drawn + count` values, keep the tail
        field = np.empty(self._drawn + count, np.float32)
        check(_lib.wn_gaussian_fill(self.randomSeed, field.size, field.ctypes.data_as(C.c_void_p)))
        tail = np.ascontiguousarray(field[self._drawn:])
        self._drawn += count
        h = C.c_void_p()
        check(_lib.wn_tile_generate_from_field(n, dims, tail.ctypes.data_as(C.c_void_p), C.byref(h)))
        self._set_tile(h)

    def generateNoiseTile2D(self):
        self._generate(2)

    def generateNoiseTile3D(self):
        self._generate(3)

    @classmethod
    def from_coefficients(cls, coeffs, dims):
        """Adopt ready-made coefficients (n^dims floats, x fastest)."""
        c = np.ascontiguousarray(np.asarray(coeffs, np.float32).ravel())
        n = int(round(c.size ** (1.0 / dims))) if c.size else 0
        if n ** dims != c.size:
            raise ValueError("coefficient count is not n^dims")
        self = cls(n, 0)
        h = C.c_void_p()
        check(_lib.wn_tile_create(n, dims, c.ctypes.data_as(C.c_void_p) if c.size else None, C.byref(h)))
        self._set_tile(h)
        return self

    def getNoiseCoefficients(self):
        if self._tile is None:
            return np.empty(0, np.float32)
        out = np.empty(_lib.wn_tile_count(self._tile), np.float32)
        check(_lib.wn_tile_download(self._tile, out.ctypes.data_as(C.c_void_p)))
        return out

    def getTileSize(self):
        return self.tileSizeN

    # -- evaluation: a single point returns a float, an (N,k) batch returns a CUDA tensor
    def _scalar(self, fn, dims, p, width, normal=None):
        """One value through the resident scalar kernel (wn_scalar_*: no launch per call)."""
        a = (C.c_float * width)(*[float(v) for v in (p.tolist() if hasattr(p, "tolist") else p)])
        out = C.c_float(0)
        if normal is None:
            check(fn(self._handle(dims), a, C.byref(out)))
        else:
            nr = (C.c_float * 3)(*[float(v) for v in (normal.tolist() if hasattr(normal, "tolist") else normal)])
            check(fn(self._handle(dims), a, nr, C.byref(out)))
        return out.value

    def _points(self, fn, dims, p, width, extra=None):
        single = _is_scalar_point(p, width)
        pts = _dev(p, torch.float32).reshape(-1, width)
        out = torch.empty(pts.shape[0], dtype=torch.float32, device="cuda")
        if extra is None:
            check(fn(self._handle(dims), _ptr(pts), pts.shape[0], _ptr(out), _stream()))
        else:
            check(fn(self._handle(dims), _ptr(pts), _ptr(extra), pts.shape[0], _ptr(out), _stream()))
        return float(out.item()) if single else out

    def evaluate2D(self, p):
        if _is_scalar_point(p, 2):
            return self._scalar(_lib.wn_scalar_eval2d, 2, p, 2)
        return self._points(_lib.wn_eval2d_points, 2, p, 2)

    def evaluate3D(self, p):
        if _is_scalar_point(p, 3):
            return self._scalar(_lib.wn_scalar_eval3d, 3, p, 3)
        return self._points(_lib.wn_eval3d_points, 3, p, 3)

    def evaluate3DProjected(self, p, normal):
        if _is_scalar_point(p, 3) and _is_scalar_point(normal, 3):
            return self._scalar(_lib.wn_scalar_eval3d_projected, 3, p, 3, normal)
        pts = _dev(p, torch.float32).reshape(-1, 3)
        nr = _dev(normal, torch.float32).reshape(-1, 3)
        if nr.shape[0] == 1 and pts.shape[0] != 1:
            nr = nr.expand(pts.shape[0], 3).contiguous()
        return self._points(_lib.wn_eval3d_projected_points, 3, p, 3, extra=nr)

    def WMultibandNoise(self, p, s, firstBand, nbands, w, variance=None, normal=None):
        """Cook & DeRose Appendix 2; absent from the reference.  normal=None: bands are WNoise = evaluate3D
        (variance defaults to the reference's empirical 0.18402); with a normal (one for all points, or one per
        point) bands are WProjectedNoise = evaluate3DProjected (variance defaults to 0.296)."""
        single = _is_scalar_point(p, 3)
        pts = _dev(p, torch.float32).reshape(-1, 3)
        out = torch.empty(pts.shape[0], dtype=torch.float32, device="cuda")
        wa = (C.c_float * max(1, nbands))(*[float(x) for x in list(w)[:nbands]])
        if normal is not None:
            nr = _dev(normal, torch.float32).reshape(-1, 3)
            one = nr.shape[0] == 1
            if not one and nr.shape[0] != pts.shape[0]:
                raise ValueError("normal: one vector, or one per point")
            check(_lib.wn_multiband3d_projected_points(self._handle(3), _ptr(pts), _ptr(nr), int(one), pts.shape[0],
                                                       float(s), int(firstBand), int(nbands), wa,
                                                       float(0.296 if variance is None else variance), _ptr(out),
                                                       _stream()))
            return float(out.item()) if single else out
        variance = 0.18402 if variance is None else variance
        check(_lib.wn_multiband3d_points(self._handle(3), _ptr(pts), pts.shape[0], float(s),
                                         int(firstBand), int(nbands), wa, float(variance),
                                         _ptr(out), _stream()))
        return float(out.item()) if single else out


# ---- perlin / PerlinNoise (perlin.h:14-91, experient/PerlinNoise.hpp:9-61) -------------------
class perlin:
    def __init__(self, seed=5489):  # std::mt19937::default_seed
        self._h = C.c_void_p()
        check(_lib.wn_perm_create_seeded(int(seed) & 0xFFFFFFFF, C.byref(self._h)))

    @classmethod
    def from_table(cls, table512):
        self = cls.__new__(cls)
        t = np.ascontiguousarray(np.asarray(table512, np.int32))
        self._h = C.c_void_p()
        check(_lib.wn_perm_create(t.ctypes.data_as(C.c_void_p), C.byref(self._h)))
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            _lib.wn_perm_destroy(self._h)
            self._h = None

    @property
    def p(self):
        out = np.empty(512, np.int32)
        check(_lib.wn_perm_download(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def _run(self, fn, pts, *mid):
        out = torch.empty(pts.shape[0], dtype=torch.float64, device="cuda")
        check(fn(self._h, _ptr(pts), pts.shape[0], *mid, _ptr(out), _stream()))
        return out

    def noise(self, x, y=None, z=None):
        """noise(x,y,z) / noise(x,y) on doubles (perlin.h:42,65); noise(p) on a float vec3 or an
        (N,3) batch: float32 input follows noise(const point3&) (perlin.h:70), float64 input
        follows noise(double,double,double)."""
        if y is not None:
            out = C.c_double(0)
            check(_lib.wn_scalar_perlin(self._h, float(x), float(y), 0.0 if z is None else float(z), C.byref(out)))
            return out.value
        single = _is_scalar_point(x, 3)
        is64 = (x.dtype == torch.float64) if isinstance(x, torch.Tensor) else \
            (np.asarray(x).dtype == np.float64 and not single)
        if single and not is64:
            return self._scalar_vec3(x, 0)  # noise(const point3&), perlin.h:70
        if is64:
            out = self._run(_lib.wn_perlin_points, _dev(x, torch.float64).reshape(-1, 3))
        else:
            out = self._run(_lib.wn_perlin_points_vec3, _dev(x, torch.float32).reshape(-1, 3))
        return float(out.item()) if single else out

    def _scalar_vec3(self, p, kind, depth=0):
        a = (C.c_float * 3)(*[float(v) for v in (p.tolist() if hasattr(p, "tolist") else p)])
        out = C.c_double(0)
        check(_lib.wn_scalar_perlin_vec3(self._h, a, kind, depth, C.byref(out)))
        return out.value

    def fractal_noise(self, p):
        single = _is_scalar_point(p, 3)
        if single:
            return self._scalar_vec3(p, 2)
        out = self._run(_lib.wn_perlin_fractal_points, _dev(p, torch.float32).reshape(-1, 3))
        return float(out.item()) if single else out

    def turb(self, p, depth=7):
        """RTOW turb(p, depth); absent from the reference."""
        single = _is_scalar_point(p, 3)
        if single:
            return self._scalar_vec3(p, 1, int(depth))
        out = self._run(_lib.wn_perlin_turb_points, _dev(p, torch.float32).reshape(-1, 3), int(depth))
        return float(out.item()) if single else out


PerlinNoise = perlin  # experient/PerlinNoise.hpp is the same algorithm with an explicit seed


# ---- textures (texture.h) --------------------------------------------------------------------
class noise_texture:
    def __init__(self, scale, octave=4):
        self.noise = perlin()  # default-seeded member (texture.h:46)
        self.scale, self.octave_level = float(scale), int(octave)

    def grey(self, p, active=None, out=None):
        pts = _dev(p, torch.float32).reshape(-1, 3)
        if out is None:
            out = torch.zeros(pts.shape[0], dtype=torch.float32, device="cuda")
        act = _dev(active, torch.uint8) if active is not None else None
        check(_lib.wn_noise_texture_points(self.noise._h, self.scale, self.octave_level, _ptr(pts),
                                           _ptr(act), pts.shape[0], _ptr(out), _stream()))
        return out

    def value(self, u, v, p):
        if _is_scalar_point(p, 3):  # the reference's call shape: one request to the resident scalar kernel
            a = (C.c_float * 3)(*[float(x) for x in (p.tolist() if hasattr(p, "tolist") else p)])
            out = C.c_float(0)
            check(_lib.wn_scalar_noise_texture(self.noise._h, self.scale, self.octave_level, a, C.byref(out)))
            return (out.value,) * 3
        g = self.grey(p)
        return g[:, None].expand(-1, 3)


class wavelet_texture:
    def __init__(self, scale=1.0, octave=4, use_3d=True):
        self.scale, self.octave_level, self.use_3d_noise = float(scale), int(octave), bool(use_3d)
        TILE_SIZE, SEED = 128, 12345  # texture.h:55-56
        self.noise_2d = WaveletNoise(TILE_SIZE, SEED)
        self.noise_2d.generateNoiseTile2D()
        self.noise_3d = None
        if self.use_3d_noise:
            self.noise_3d = WaveletNoise(TILE_SIZE, SEED)
            self.noise_3d.generateNoiseTile3D()

    def grey(self, p, active=None, out=None):
        pts = _dev(p, torch.float32).reshape(-1, 3)
        if out is None:
            out = torch.zeros(pts.shape[0], dtype=torch.float32, device="cuda")
        act = _dev(active, torch.uint8) if active is not None else None
        use3d = self.use_3d_noise and self.noise_3d is not None
        src = self.noise_3d if use3d else self.noise_2d
        check(_lib.wn_wavelet_texture_points(src._handle(3 if use3d else 2), int(use3d), self.scale,
                                             self.octave_level, _ptr(pts), _ptr(act), pts.shape[0],
                                             _ptr(out), _stream()))
        return out

    def value(self, u, v, p):
        if _is_scalar_point(p, 3):
            use3d = self.use_3d_noise and self.noise_3d is not None
            src = self.noise_3d if use3d else self.noise_2d
            a = (C.c_float * 3)(*[float(x) for x in (p.tolist() if hasattr(p, "tolist") else p)])
            out = C.c_float(0)
            check(_lib.wn_scalar_wavelet_texture(src._handle(3 if use3d else 2), int(use3d), self.scale,
                                                 self.octave_level, a, C.byref(out)))
            return (out.value,) * 3
        g = self.grey(p)
        return g[:, None].expand(-1, 3)


# ---- dense grids (experient/main.cpp) -----------------------------------------------------------
@dataclass
class GridSpec:
    den: int
    nx: int
    ny: int
    z0: int = 0
    z1: int = 1
    base_range: float = 4.0          # experient/main.cpp:13
    octave_scale: float = 1.0
    post_scale: float = 1.0
    z_mode: int = WN_Z_LATTICE
    z_const: float = 0.0
    out_scale: float = 1.0
    flags: int = WN_GRID_DEFAULT

    def c(self):
        return wn_grid(self.den, self.nx, self.ny, self.z0, self.z1, self.base_range,
                       self.octave_scale, self.post_scale, self.z_mode, self.z_const,
                       self.out_scale, self.flags)

    @property
    def nz(self):
        return 1 if self.z_mode == WN_Z_CONST else self.z1 - self.z0

    def empty(self, out=None):
        n = self.nz * self.ny * self.nx
        if out is None:
            return torch.empty(n, dtype=torch.float32, device="cuda")
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= n
        return out


def _f32(x):
    return float(np.float32(x))


def _inv_stddev(var):
    return _f32(np.float32(1.0) / np.sqrt(np.float32(var)))  # 1.0f / std::sqrt(0.18402f)


def _octave_scale(octave):
    return _f32(math.pow(2.0, octave))  # std::pow(2.0f, octave) -> float


def _write(t, outputFile):
    if outputFile:
        t.cpu().numpy().astype("<f4").tofile(outputFile)  # raw float32, experient/main.cpp:32-34


def generate2DOctaveBandNoise(imageSize, octave, outputFile, noise, flags=WN_GRID_EXACT):
    g = GridSpec(imageSize, imageSize, imageSize, octave_scale=_octave_scale(octave), post_scale=2.0,
                 out_scale=_inv_stddev(0.19686), flags=flags)
    out = g.empty()
    gc = g.c()
    check(_lib.wn_eval2d_grid(noise._handle(2), C.byref(gc), _ptr(out), _stream()))
    out = out.view(imageSize, imageSize)
    _write(out, outputFile)
    return out


def generate3DSlicedOctaveBandNoise(imageSize, octave, outputFile, noise, flags=WN_GRID_EXACT):
    """Byte-identical to the reference's file by default (WN_GRID_EXACT); flags=WN_GRID_DEFAULT opts in
    to the separable brick kernel (within 1e-5)."""
    g = GridSpec(imageSize, imageSize, imageSize, octave_scale=_octave_scale(octave), post_scale=2.0,
                 z_mode=WN_Z_CONST, z_const=2.0, out_scale=_inv_stddev(0.18402), flags=flags)
    out = g.empty()
    gc = g.c()
    check(_lib.wn_eval3d_grid(noise._handle(3), C.byref(gc), _ptr(out), _stream()))
    out = out.view(imageSize, imageSize)
    _write(out, outputFile)
    return out


def generate3DProjectedOctaveBandNoise(imageSize, octave, outputFile, noise, normal=(0.0, 0.0, 1.0)):
    g = GridSpec(imageSize, imageSize, imageSize, octave_scale=_octave_scale(octave), post_scale=2.0,
                 z_mode=WN_Z_CONST, z_const=2.0, out_scale=_inv_stddev(0.296))
    out = g.empty()
    gc = g.c()
    nr = (C.c_float * 3)(*[float(v) for v in normal])
    check(_lib.wn_eval3d_projected_grid(noise._handle(3), C.byref(gc), nr, _ptr(out), _stream()))
    out = out.view(imageSize, imageSize)
    _write(out, outputFile)
    return out


def generatePerlinNoise2D(imageSize, octave, outputFile, perlin_obj):
    g = GridSpec(imageSize, imageSize, imageSize, octave_scale=_octave_scale(octave),
                 z_mode=WN_Z_CONST, z_const=0.0)
    out = g.empty()
    gc = g.c()
    check(_lib.wn_perlin_grid(perlin_obj._h, C.byref(gc), _ptr(out), _stream()))
    out = out.view(imageSize, imageSize)
    _write(out, outputFile)
    return out


def generatePerlinNoise3DSliced(imageSize, octave, outputFile, perlin_obj):
    os_ = _octave_scale(octave)
    g = GridSpec(imageSize, imageSize, imageSize, octave_scale=os_, z_mode=WN_Z_CONST,
                 z_const=_f32(np.float32(1.0) * np.float32(os_)))
    out = g.empty()
    gc = g.c()
    check(_lib.wn_perlin_grid(perlin_obj._h, C.byref(gc), _ptr(out), _stream()))
    out = out.view(imageSize, imageSize)
    _write(out, outputFile)
    return out


# ---- volumes ---------------------------------------------------------
def wavelet_volume(noise, den, nx, ny, z0, z1, octave, exact=False, out=None):
    """Config 2/5: q = ((i/den)*4)*2^octave*2 on all three axes, evaluate3D(q)/sqrt(0.18402)."""
    g = GridSpec(den, nx, ny, z0, z1, octave_scale=_octave_scale(octave), post_scale=2.0,
                 out_scale=_inv_stddev(0.18402), flags=WN_GRID_EXACT if exact else WN_GRID_DEFAULT)
    out = g.empty(out)
    gc = g.c()
    check(_lib.wn_eval3d_grid(noise._handle(3), C.byref(gc), _ptr(out), _stream()))
    return out[: g.nz * ny * nx].view(g.nz, ny, nx)


def wavelet_volume_launcher(noise, den, nx, ny, z0, z1, octave, out, exact=False):
    """The same call as wavelet_volume with its arguments marshalled once: returns launch(), one kernel launch into
    `out` on the stream that is current NOW (for long back-to-back runs, where a few tens of microseconds of
    argument handling per call would sit beside a 100 us kernel)."""
    g = GridSpec(den, nx, ny, z0, z1, octave_scale=_octave_scale(octave), post_scale=2.0,
                 out_scale=_inv_stddev(0.18402), flags=WN_GRID_EXACT if exact else WN_GRID_DEFAULT)
    out = g.empty(out)
    gc, h, p, st, fn = g.c(), noise._handle(3), _ptr(out), _stream(), _lib.wn_eval3d_grid
    ref = C.byref(gc)

    def launch():
        rc = fn(h, ref, p, st)
        if rc:
            check(rc)
    launch.keep = (gc, noise, out)
    return launch


def multiband_volume(noise, den, nx, ny, z0, z1, s=-16.0, firstBand=0, nbands=5, w=None,
                     variance=0.18402, exact=False, out=None):
    """Config 3(A): WMultibandNoise(p=(i/den)*4, s, NULL, firstBand, nbands, w)."""
    w = [1.0] * nbands if w is None else list(w)
    g = GridSpec(den, nx, ny, z0, z1, flags=WN_GRID_EXACT if exact else WN_GRID_DEFAULT)
    out = g.empty(out)
    gc = g.c()
    wa = (C.c_float * max(1, nbands))(*[float(x) for x in w[:nbands]])
    check(_lib.wn_multiband3d_grid(noise._handle(3), C.byref(gc), float(s), int(firstBand),
                                   int(nbands), wa, float(variance), _ptr(out), _stream()))
    return out[: g.nz * ny * nx].view(g.nz, ny, nx)


def perlin_volume(perlin_obj, den, nx, ny, z0, z1, octave, out=None):
    g = GridSpec(den, nx, ny, z0, z1, octave_scale=_octave_scale(octave))
    out = g.empty(out)
    gc = g.c()
    check(_lib.wn_perlin_grid(perlin_obj._h, C.byref(gc), _ptr(out), _stream()))
    return out[: g.nz * ny * nx].view(g.nz, ny, nx)


def turb_volume(perlin_obj, den, nx, ny, z0, z1, depth=7, out=None):
    """Config 3(B): turb(p=(i/den)*4, depth)."""
    g = GridSpec(den, nx, ny, z0, z1)
    out = g.empty(out)
    gc = g.c()
    check(_lib.wn_perlin_turb_grid(perlin_obj._h, C.byref(gc), int(depth), _ptr(out), _stream()))
    return out[: g.nz * ny * nx].view(g.nz, ny, nx)
