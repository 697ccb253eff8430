#!/usr/bin/env python3
"""Generate tests/golden/ from the REAL reference (oracle/_ref/libwnref.so + committed artefacts).

TEST INFRASTRUCTURE.  Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

Writes
  tests/golden/result_raw/*.raw      the 15 float32[256*256] grids the reference commits under
                                     experient/result_raw/ (data files, copied verbatim)
  tests/golden/json_stats.json       the `original_range` blocks of threejs/result_json/*.json
  tests/golden/artefacts.json        sha256 of the raws and of result_raytracing/*.png
  tests/golden/ref_vectors.npz       inputs + outputs of the compiled reference's own classes
                                     (seeded numpy inputs; every output is produced by calling
                                     the reference through oracle/ref_shim.cpp)
The inputs are data, not code; nothing of the reference's source text is stored.
"""
import glob
import hashlib
import json
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def decode_png_rgb(path):
    """Minimal 8-bit RGB, non-interlaced PNG decoder (the reference's stb_image_write output)."""
    import struct
    import zlib
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(d):
        ln = struct.unpack(">I", d[pos:pos + 4])[0]
        typ, body = d[pos + 4:pos + 8], d[pos + 8:pos + 8 + ln]
        if typ == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert depth == 8 and ctype == 2
        elif typ == b"IDAT":
            idat += body
        pos += 12 + ln
    rawb = zlib.decompress(idat)
    bpp, stride = 3, w * 3
    img = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int64)
    p = 0
    for y in range(h):
        f = rawb[p]
        line = np.frombuffer(rawb[p + 1:p + 1 + stride], np.uint8).astype(np.int64)
        p += 1 + stride
        if f == 0:
            cur = line
        elif f == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(stride, np.int64)
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 1:
                    pr = a
                elif f == 3:
                    pr = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (line[i] + pr) & 255
        img[y] = cur
        prev = cur
    return img.reshape(h, w, 3)


def special_points():
    """Coordinates the B-spline / floor logic is sensitive to: integers, half-integers,
    negatives, tile-period multiples, the scene extremes (lattice +-320)."""
    vals = [0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -1.5, 0.49999997, 0.50000006, 127.5, 128.0, 128.5,
            -127.5, -128.0, 255.9, 256.1, -0.001, 320.0, -320.0, 17.3, 64.0, 100.1, -3.75, 1.25,
            8.0, 2.0, 1e-7, -1e-7, 1023.75, -1024.25]
    pts = []
    for i, a in enumerate(vals):
        pts.append((a, vals[(i * 7 + 3) % len(vals)], vals[(i * 11 + 5) % len(vals)]))
        pts.append((a, a, a))
    # the SURVEY 8(c) probes
    pts += [(0, 0, 0), (0.5, 0.5, 0.5), (1.25, -3.75, 100.1), (-0.49, 127.6, 64),
            (320, -320, 17.3), (8, 8, 2), (255.9, 256.1, -0.001)]
    return np.array(pts, np.float32)


def main():
    R = oracle.ref()
    if R is None:
        sys.exit("oracle/_ref/libwnref.so missing and /root/reference absent: cannot generate")
    os.makedirs(os.path.join(GOLD, "result_raw"), exist_ok=True)
    rng = np.random.default_rng(20251004)
    out = {}

    # ---- committed artefacts ------------------------------------------------------------------
    art = {"raw": {}, "png": {}}
    for p in sorted(glob.glob(os.path.join(REF, "experient/result_raw/*.raw"))):
        shutil.copyfile(p, os.path.join(GOLD, "result_raw", os.path.basename(p)))
        art["raw"][os.path.basename(p)] = sha256(p)
    art["png_rgb_sha256"] = {}
    for p in sorted(glob.glob(os.path.join(REF, "result_raytracing/*.png"))):
        art["png"][os.path.basename(p)] = sha256(p)
        rgb = decode_png_rgb(p)  # pixel content of the committed render (the PNG container is stb's)
        art["png_rgb_sha256"][os.path.basename(p)] = {"shape": list(rgb.shape),
                                                      "sha256": hashlib.sha256(rgb.tobytes()).hexdigest()}
    # hit-point stream of the reference's render loop (oracle/_ref/raytrace_record: the reference's
    # unmodified main.cpp with oracle/ref_record_texture.h): count and FNV-1a64 of every p passed to
    # tex->value(), 1000x500x100 spp; identical for both noise types (noise never steers the paths)
    import subprocess
    import tempfile
    rec = os.path.join(ROOT, "oracle", "_ref", "raytrace_record")
    if os.path.exists(rec):
        with tempfile.TemporaryDirectory() as td:
            proc = subprocess.run([rec], input="1\n4\n", cwd=td, capture_output=True, text=True)
        line = [ln for ln in proc.stderr.splitlines() if ln.startswith("WN_RECORD")][-1]
        art["render_hit_stream"] = dict(kv.split("=") for kv in line.split()[1:])
    stats = {}
    for p in sorted(glob.glob(os.path.join(REF, "threejs/result_json/*.json"))):
        j = json.load(open(p))
        stats[os.path.basename(p)] = {"width": j["width"], "height": j["height"],
                                      "original_range": j["original_range"],
                                      "data_head": j["data"][:32], "data_len": len(j["data"]),
                                      "data_sum": float(np.sum(np.array(j["data"], np.float64)))}
    json.dump(stats, open(os.path.join(GOLD, "json_stats.json"), "w"), indent=1, sort_keys=True)

    # ---- libstdc++ streams ----------------------------------------------------------------------
    seeds = np.array([12345, 5489, 0, 1, 2024, 4294967295], np.uint32)
    out["perm_seeds"] = seeds
    perms = np.zeros((len(seeds), 512), np.int32)
    perms2 = np.zeros((len(seeds), 512), np.int32)
    for k, s in enumerate(seeds):
        h = R.ref_perlin_new(int(s)); R.ref_perlin_perm(h, perms[k]); R.ref_perlin_delete(h)
        h = R.ref_PerlinNoise_new(int(s)); R.ref_PerlinNoise_perm(h, perms2[k]); R.ref_PerlinNoise_delete(h)
    assert (perms == perms2).all()
    out["perm_tables"] = perms
    h = R.ref_perlin_new_default(); pd = np.zeros(512, np.int32); R.ref_perlin_perm(h, pd); R.ref_perlin_delete(h)
    out["perm_default"] = pd
    g = np.zeros(4096, np.float32); R.ref_gaussian_stream(12345, g.size, g); out["gauss_12345"] = g
    for s in (0, 1, 5489):
        g = np.zeros(64, np.float32); R.ref_gaussian_stream(s, g.size, g); out[f"gauss_{s}"] = g

    # ---- tiles --------------------------------------------------------------------------------------
    def ref_tile(n, seed, dims):
        h = R.ref_wn_new(n, seed)
        (R.ref_wn_generate2d if dims == 2 else R.ref_wn_generate3d)(h)
        c = np.zeros(R.ref_wn_coeff_count(h), np.float32)
        R.ref_wn_coeffs(h, c)
        return h, c

    h2, t2 = ref_tile(128, 12345, 2)
    h3, t3 = ref_tile(128, 12345, 3)
    out["tile2d_128_12345"] = t2
    idx = np.sort(rng.choice(t3.size, 8192, replace=False)).astype(np.int64)
    out["tile3d_128_12345_idx"] = idx
    out["tile3d_128_12345_val"] = t3[idx]
    art["tile3d_128_12345"] = {"fnv1a64": "%016x" % oracle.fnv1a64(t3),
                               "sha256": hashlib.sha256(t3.tobytes()).hexdigest(),
                               "var": float(t3.astype(np.float64).var()),
                               "mean": float(t3.astype(np.float64).mean())}
    art["tile2d_128_12345"] = {"fnv1a64": "%016x" % oracle.fnv1a64(t2),
                               "sha256": hashlib.sha256(t2.tobytes()).hexdigest(),
                               "var": float(t2.astype(np.float64).var())}
    small = {}
    for name, n, seed, dims in (("tile3d_8_7", 8, 7, 3), ("tile3d_16_12345", 16, 12345, 3),
                                ("tile2d_16_99", 16, 99, 2), ("tile2d_7odd_3", 7, 3, 2),
                                ("tile3d_5odd_11", 5, 11, 3)):
        hh, c = ref_tile(n, seed, dims)
        out[name] = c
        small[name] = (hh, c)
        art[name] = {"tile_size": int(R.ref_wn_tile_size(hh))}

    # ---- point probes, tile 128 ----------------------------------------------------------------------
    sp = special_points()
    pts = np.concatenate([sp, rng.uniform(-400, 400, (2048, 3)).astype(np.float32),
                          rng.uniform(-4, 4, (512, 3)).astype(np.float32)])
    pts = np.ascontiguousarray(pts, np.float32)
    out["probe_pts"] = pts
    e = np.zeros(len(pts), np.float32)
    R.ref_wn_eval2d(h2, np.ascontiguousarray(pts[:, :2]), len(pts), e); out["probe_e2d"] = e.copy()
    R.ref_wn_eval3d(h3, pts, len(pts), e); out["probe_e3d"] = e.copy()
    # projected: axis normals + random unit normals on a subset (it is ~40x slower)
    np_pts = np.ascontiguousarray(pts[:640])
    normals = np.zeros((len(np_pts), 3), np.float32)
    normals[:, 2] = 1.0
    normals[200:300] = (1, 0, 0)
    normals[300:400] = (0, 1, 0)
    rn = rng.normal(size=(240, 3)); rn /= np.linalg.norm(rn, axis=1, keepdims=True)
    normals[400:640] = rn.astype(np.float32)
    normals = np.ascontiguousarray(normals)
    ep = np.zeros(len(np_pts), np.float32)
    R.ref_wn_eval3d_projected(h3, np_pts, normals, len(np_pts), ep)
    out["probe_proj_pts"] = np_pts; out["probe_proj_normals"] = normals; out["probe_e3dp"] = ep

    # ---- point probes on the small tiles (wrap-heavy) ---------------------------------------------------
    spts = np.ascontiguousarray(np.concatenate([sp[:40], rng.uniform(-40, 40, (512, 3)).astype(np.float32)]))
    out["small_pts"] = spts
    for name in ("tile3d_8_7", "tile3d_16_12345"):
        hh, _ = small[name]
        e = np.zeros(len(spts), np.float32); R.ref_wn_eval3d(hh, spts, len(spts), e); out[name + "_e3d"] = e
    hh, _ = small["tile2d_16_99"]
    e = np.zeros(len(spts), np.float32); R.ref_wn_eval2d(hh, np.ascontiguousarray(spts[:, :2]), len(spts), e)
    out["tile2d_16_99_e2d"] = e

    # ---- Perlin ------------------------------------------------------------------------------------------------
    dp = np.concatenate([sp.astype(np.float64), rng.uniform(-600, 600, (2048, 3)),
                         rng.uniform(-3, 3, (512, 3))])
    dp = np.ascontiguousarray(dp)
    out["perlin_pts"] = dp
    for s in (12345, 5489):
        h = R.ref_perlin_new(s)
        o = np.zeros(len(dp)); R.ref_perlin_noise(h, dp, len(dp), o); out[f"perlin_noise_{s}"] = o
        fp = np.ascontiguousarray(dp.astype(np.float32))
        o = np.zeros(len(dp)); R.ref_perlin_noise_vec3(h, fp, len(fp), o); out[f"perlin_noise_vec3_{s}"] = o
        o = np.zeros(len(dp)); R.ref_perlin_fractal(h, fp, len(fp), o); out[f"perlin_fractal_{s}"] = o
        R.ref_perlin_delete(h)
    h = R.ref_PerlinNoise_new(12345)
    o = np.zeros(len(dp)); R.ref_PerlinNoise_noise(h, dp, len(dp), o)
    assert (o == out["perlin_noise_12345"]).all()
    R.ref_PerlinNoise_delete(h)

    # ---- textures (scene-like hit points: quad y=-0.5, sphere c=(1,0,-1.75) r=.5) -------------------------------
    q = np.stack([rng.uniform(-10, 10, 384), np.full(384, -0.5), rng.uniform(-10, 10, 384)], 1)
    d = rng.normal(size=(128, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    sph = np.array([1, 0, -1.75]) + 0.5 * d
    tp = np.ascontiguousarray(np.concatenate([q, sph, sp[:32] / 32.0]).astype(np.float32))
    out["tex_pts"] = tp
    cases = []
    for scale in (1.0, 0.37):
        for octave in (3, 4, 5):
            h = R.ref_noise_texture_new(scale, octave)
            rgb = np.zeros((len(tp), 3), np.float32); R.ref_texture_value(h, tp, len(tp), rgb)
            assert (rgb[:, 0] == rgb[:, 1]).all() and (rgb[:, 0] == rgb[:, 2]).all()
            out[f"tex_perlin_s{scale}_o{octave}"] = rgb[:, 0].copy()
            R.ref_texture_delete(h)
            cases.append(["perlin", scale, octave])
            for use3d in (1, 0):
                h = R.ref_wavelet_texture_new(scale, octave, use3d)
                rgb = np.zeros((len(tp), 3), np.float32); R.ref_texture_value(h, tp, len(tp), rgb)
                assert (rgb[:, 0] == rgb[:, 1]).all() and (rgb[:, 0] == rgb[:, 2]).all()
                out[f"tex_wavelet{'3d' if use3d else '2d'}_s{scale}_o{octave}"] = rgb[:, 0].copy()
                R.ref_texture_delete(h)
                cases.append(["wavelet3d" if use3d else "wavelet2d", scale, octave])
    art["texture_cases"] = cases

    # ---- dense 3-D volumes (SURVEY 8(d) config-2 mapping at small N; tile 128, seed 12345) ---------------------------
    vols = []
    for name, den, nx, ny, z0, z1, octave in (("vol_a", 32, 32, 32, 0, 32, 4),      # step 4
                                              ("vol_b", 64, 48, 40, 5, 9, 3),       # ragged slab, step 1
                                              ("vol_c", 512, 64, 16, 100, 104, 4),  # step .25 corner
                                              ("vol_d", 2048, 96, 8, 2040, 2044, 4),  # step 1/16
                                              ("vol_e", 100, 100, 12, 3, 5, 5)):    # non power of two
        v = np.zeros((z1 - z0) * ny * nx, np.float32)
        R.ref_wn_grid3d_volume(h3, den, nx, ny, z0, z1, octave, v)
        out[name] = v.reshape(z1 - z0, ny, nx)
        vols.append([name, den, nx, ny, z0, z1, octave])
    art["volumes"] = vols

    np.savez_compressed(os.path.join(GOLD, "ref_vectors.npz"), **out)
    json.dump(art, open(os.path.join(GOLD, "artefacts.json"), "w"), indent=1, sort_keys=True)
    sz = os.path.getsize(os.path.join(GOLD, "ref_vectors.npz"))
    print(f"wrote {len(out)} arrays, ref_vectors.npz = {sz / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
