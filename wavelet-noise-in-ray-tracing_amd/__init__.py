"""wavelet-noise-in-ray-tracing_amd: MI355X-native evaluation of the wavelet / Perlin noise hot
path of Jason9339/Wavelet-Noise-in-ray-tracing.

Layout
  csrc/   hand-written HIP kernels + the C ABI of include/wnoise.h  -> libwnoise_hip.so
  host/   C++ host classes with the reference's names (WaveletNoise, perlin, PerlinNoise,
          noise_texture, wavelet_texture) forwarding to the C ABI
  tools/  grid generator (experient/main.cpp's role) over the batched ABI
  noise.py  Python mirror of the same interface (tests / bench harness; torch supplies device
          memory, streams and torch.distributed, nothing else)

The directory name is not a Python identifier: import it with
    importlib.import_module("wavelet-noise-in-ray-tracing_amd")
"""
from . import _capi
from ._capi import WnError, wn_grid, WN_GRID_DEFAULT, WN_GRID_EXACT, WN_Z_CONST, WN_Z_LATTICE

_capi.load()  # fail loudly when the HIP library is missing

from .noise import (  # noqa: E402
    WaveletNoise, perlin, PerlinNoise, noise_texture, wavelet_texture, GridSpec,
    generate2DOctaveBandNoise, generate3DSlicedOctaveBandNoise,
    generate3DProjectedOctaveBandNoise, generatePerlinNoise2D, generatePerlinNoise3DSliced,
    wavelet_volume, wavelet_volume_launcher, multiband_volume, perlin_volume, turb_volume, device_info, HipTimer,
)
from .shard import slab_bounds, gather_volume, NativeComm  # noqa: E402
from . import formats  # noqa: E402

__all__ = [
    "WnError", "wn_grid", "WN_GRID_DEFAULT", "WN_GRID_EXACT", "WN_Z_CONST", "WN_Z_LATTICE",
    "WaveletNoise", "perlin", "PerlinNoise", "noise_texture", "wavelet_texture", "GridSpec",
    "generate2DOctaveBandNoise", "generate3DSlicedOctaveBandNoise",
    "generate3DProjectedOctaveBandNoise", "generatePerlinNoise2D", "generatePerlinNoise3DSliced",
    "wavelet_volume", "wavelet_volume_launcher", "multiband_volume", "perlin_volume", "turb_volume", "device_info",
    "HipTimer", "slab_bounds", "gather_volume", "NativeComm", "formats",
]
