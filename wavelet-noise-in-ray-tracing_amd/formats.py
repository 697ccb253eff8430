"""On-disk formats of the reference and its spectral check.

  * raw:  float32[height*width], little-endian, row-major (experient/main.cpp:32-34)
  * JSON: {"width","height","original_range":{"min","max","mean","std"},"data":[0..1]} with
          separators (',', ':')  (threejs/convert_raw_to_json.py:12-90)
  * radial power spectrum of a grid (experient/analyze.py:88-136), computed with torch.fft -- rocFFT when
    the grid lives on the GPU -- used by the tests as an automated band-limit assertion.
"""
import json
import os

import numpy as np
import torch


def write_raw(grid, path):
    a = grid.detach().cpu().numpy() if isinstance(grid, torch.Tensor) else np.asarray(grid)
    a.astype("<f4").tofile(path)


def read_raw(path, size=None):
    a = np.fromfile(path, dtype="<f4")
    if size is None:
        size = int(round(np.sqrt(a.size)))
    return a.reshape(size, -1)


def raw_to_json_dict(grid):
    """The JSON document convert_raw_to_json.py writes for a square float32 grid."""
    a = grid.detach().cpu().numpy() if isinstance(grid, torch.Tensor) else np.asarray(grid)
    a = a.astype(np.float32).astype(np.float64)  # the converter unpacks floats into a float64 array
    lo, hi = float(a.min()), float(a.max())
    norm = (a - lo) / (hi - lo) if hi != lo else np.zeros_like(a)
    return {"width": int(a.shape[1]), "height": int(a.shape[0]),
            "original_range": {"min": lo, "max": hi, "mean": float(a.mean()), "std": float(a.std())},
            "data": norm.flatten().tolist()}


def convert_raw_to_json(raw_file_path, json_file_path, image_size=256):
    """threejs/convert_raw_to_json.py:12-90.  A missing input returns False (:23-25); a float count
    other than image_size^2 re-derives image_size = int(sqrt(count)) (:36-39) and writes that square;
    a count that is no perfect square cannot be reshaped there (:42, caught at :88-90) -> False; a file whose
    length is no multiple of 4 fails the reference's struct.unpack (:30-33) -> False."""
    if not os.path.exists(raw_file_path):
        return False
    if os.path.getsize(raw_file_path) % 4:  # struct.unpack(f'{len//4}f', data) raises on trailing bytes (:30-33) -> :88-90
        return False
    a = np.fromfile(raw_file_path, dtype="<f4")
    if a.size != image_size * image_size:
        image_size = int(np.sqrt(a.size))
    if a.size == 0 or a.size != image_size * image_size:
        return False
    grid = a.reshape(image_size, image_size)
    out_dir = os.path.dirname(json_file_path)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    with open(json_file_path, "w") as f:
        json.dump(raw_to_json_dict(grid), f, separators=(",", ":"))
    return True


def radial_power_spectrum(grid):
    """analyze.py:88-91,117-136: power spectrum of the mean-removed grid, averaged over integer
    radius bins [r, r+1) around the centre; returns (profile[r], peak_radius)."""
    g = grid if isinstance(grid, torch.Tensor) else torch.from_numpy(np.asarray(grid))
    g = g.to(torch.float64)
    g = g - g.mean()
    power = torch.fft.fftshift(torch.fft.fft2(g)).abs() ** 2
    h, w = power.shape
    cy, cx = h // 2, w // 2
    yy = torch.arange(h, device=g.device, dtype=torch.float64)[:, None] - cy
    xx = torch.arange(w, device=g.device, dtype=torch.float64)[None, :] - cx
    rbin = torch.sqrt(xx * xx + yy * yy).floor().long()
    max_r = min(cx, cy)
    keep = rbin < max_r
    sums = torch.zeros(max_r, dtype=torch.float64, device=g.device).index_add_(0, rbin[keep], power[keep])
    counts = torch.zeros(max_r, dtype=torch.float64, device=g.device).index_add_(
        0, rbin[keep], torch.ones_like(power[keep]))
    profile = torch.where(counts > 0, sums / counts.clamp(min=1), torch.zeros_like(sums))
    return profile, int(torch.argmax(profile))


def band_energy_fraction(profile, r_lo, r_hi):
    """Share of the ring-weighted radial power that falls in radii [r_lo, r_hi]."""
    r = torch.arange(profile.numel(), dtype=profile.dtype, device=profile.device)
    ring = profile * (2 * r + 1)
    return float(ring[r_lo:r_hi + 1].sum() / ring.sum())
