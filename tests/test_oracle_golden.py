"""The oracle (oracle/wn_oracle.c) against every golden the reference offers.

CPU only.  Bar: BIT-EXACT: the 15 committed experient/result_raw grids, the vectors
the compiled reference produced (tests/golden/ref_vectors.npz) and the SURVEY fingerprints.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD, bits, raw

OCTAVES = (3, 4, 5)


def test_fnv_known_answers(ora):
    assert ora.fnv1a64(np.frombuffer(b"", np.uint8)) == 0xCBF29CE484222325
    assert ora.fnv1a64(np.frombuffer(b"a", np.uint8)) == 0xAF63DC4C8601EC8C
    assert ora.fnv1a64(np.frombuffer(b"foobar", np.uint8)) == 0x85944171F73967E8


def test_mod_matches_reference_semantics(ora):
    L = ora.lib()
    for x in (-257, -129, -128, -1, 0, 1, 127, 128, 129, 1000):
        for n in (1, 2, 6, 128):
            assert L.wno_mod(x, n) == x % n  # python % is the non-negative modulo for n > 0


def test_perm_tables(ora, gold):
    for s, table in zip(gold["perm_seeds"], gold["perm_tables"]):
        assert (ora.perlin_perm(int(s)) == table).all(), int(s)
    assert (ora.perlin_perm(5489) == gold["perm_default"]).all()  # perlin.h:34 default seed
    p = ora.perlin_perm(12345)
    # SURVEY 8(c)
    assert list(p[:16]) == [48, 218, 61, 238, 202, 125, 107, 148, 4, 16, 18, 151, 98, 34, 50, 197]
    assert int((p[:256].astype(np.int64) * np.arange(1, 257)).sum()) == 4332262
    q = ora.perlin_perm(5489)
    assert list(q[:16]) == [206, 21, 27, 124, 238, 156, 174, 113, 170, 81, 237, 12, 22, 241, 23, 141]
    assert int((q[:256].astype(np.int64) * np.arange(1, 257)).sum()) == 4338650
    assert sorted(p[:256]) == list(range(256)) and (p[:256] == p[256:]).all()


def test_gaussian_stream(ora, gold):
    import ctypes as C
    L = ora.lib()

    class Normal(C.Structure):
        _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int), ("saved", C.c_float),
                    ("avail", C.c_int)]
    L.wno_normal_seed.argtypes = [C.POINTER(Normal), C.c_uint32]
    L.wno_normal_next.argtypes = [C.POINTER(Normal)]
    L.wno_normal_next.restype = C.c_float
    for seed, key in ((12345, "gauss_12345"), (0, "gauss_0"), (1, "gauss_1"), (5489, "gauss_5489")):
        d = Normal()
        L.wno_normal_seed(C.byref(d), seed)
        got = np.array([L.wno_normal_next(C.byref(d)) for _ in range(len(gold[key]))], np.float32)
        assert (bits(got) == bits(gold[key])).all(), seed
    # SURVEY 8(c)
    np.testing.assert_allclose(gold["gauss_12345"][:6], [-0.785830259, -0.390740573, 0.528865635,
                                                         -0.478647768, 1.17944515, 2.49012256], rtol=1e-7)


def test_tiles(ora, gold, artefacts, tile2d_128, tile3d_128):
    assert (bits(tile2d_128) == bits(gold["tile2d_128_12345"])).all()
    assert hashlib.sha256(tile2d_128.tobytes()).hexdigest() == artefacts["tile2d_128_12345"]["sha256"]
    assert hashlib.sha256(tile3d_128.tobytes()).hexdigest() == artefacts["tile3d_128_12345"]["sha256"]
    assert (bits(tile3d_128[gold["tile3d_128_12345_idx"]]) == bits(gold["tile3d_128_12345_val"])).all()
    # first coefficients and variances
    np.testing.assert_allclose(tile2d_128[:3], [-0.38446945, -0.796465099, 0.133784413], rtol=1e-7)
    np.testing.assert_allclose(tile3d_128[:3], [-0.184339166, -0.354518235, 0.567728519], rtol=1e-7)
    assert abs(tile2d_128.astype(np.float64).var() - 0.742875) < 1e-5
    assert abs(tile3d_128.astype(np.float64).var() - 0.876291) < 1e-5
    assert abs(tile3d_128.astype(np.float64).mean()) < 1e-7
    # small tiles in full, odd sizes bumped to even (WaveletNoise.cpp:22-25)
    assert (bits(ora.tile3d(8, 7)) == bits(gold["tile3d_8_7"])).all()
    assert (bits(ora.tile3d(16, 12345)) == bits(gold["tile3d_16_12345"])).all()
    assert (bits(ora.tile2d(16, 99)) == bits(gold["tile2d_16_99"])).all()
    assert ora.lib().wno_tile_size(7) == 8 == artefacts["tile2d_7odd_3"]["tile_size"]
    assert (bits(ora.tile2d(7, 3)) == bits(gold["tile2d_7odd_3"])).all()
    assert (bits(ora.tile3d(5, 11)) == bits(gold["tile3d_5odd_11"])).all()


def test_point_probes(ora, gold, tile2d_128, tile3d_128):
    pts = gold["probe_pts"]
    assert (bits(ora.evaluate3d(tile3d_128, pts)) == bits(gold["probe_e3d"])).all()
    assert (bits(ora.evaluate2d(tile2d_128, pts[:, :2])) == bits(gold["probe_e2d"])).all()
    got = ora.evaluate3d_projected(tile3d_128, gold["probe_proj_pts"], gold["probe_proj_normals"])
    assert (bits(got) == bits(gold["probe_e3dp"])).all()
    # wrap-heavy small tiles
    sp = gold["small_pts"]
    assert (bits(ora.evaluate3d(gold["tile3d_8_7"], sp)) == bits(gold["tile3d_8_7_e3d"])).all()
    assert (bits(ora.evaluate3d(gold["tile3d_16_12345"], sp)) == bits(gold["tile3d_16_12345_e3d"])).all()
    assert (bits(ora.evaluate2d(gold["tile2d_16_99"], sp[:, :2])) == bits(gold["tile2d_16_99_e2d"])).all()


def test_survey_probe_table(ora, tile2d_128, tile3d_128):
    """SURVEY 8(c) point probes: (e2D on x,y; e3D; e3DP with normal (0,0,1))."""
    table = {
        (0, 0, 0): (-0.24726142, -0.220912978, -0.760986149),
        (0.5, 0.5, 0.5): (-0.0232076496, -0.467016131, -0.720798492),
        (1.25, -3.75, 100.1): (0.580647409, 0.355737954, 0.467617571),
        (-0.49, 127.6, 64): (-0.273921013, 0.0526246503, -0.136443913),
        (320, -320, 17.3): (0.0300284661, 0.0570624061, -0.163255632),
        (8, 8, 2): (-0.2139927, 0.760609925, 0.641458631),
    }
    for p, (e2, e3, e3p) in table.items():
        pt = np.array([p], np.float32)
        assert abs(ora.evaluate2d(tile2d_128, pt[:, :2])[0] - e2) < 2e-7
        assert abs(ora.evaluate3d(tile3d_128, pt)[0] - e3) < 2e-7
        assert abs(ora.evaluate3d_projected(tile3d_128, pt, [[0, 0, 1]])[0] - e3p) < 2e-7


def test_empty_tile_conventions(ora):
    """Empty tile -> 0.0f (WaveletNoise.cpp:112,186,219)."""
    assert ora.evaluate2d(None, [[1.5, 2.5]])[0] == 0.0
    assert ora.evaluate3d(None, [[1.5, 2.5, 3.5]])[0] == 0.0
    assert ora.evaluate3d_projected(None, [[1.5, 2.5, 3.5]], [[0, 0, 1]])[0] == 0.0
    assert ora.wavelet_texture_value(None, True, 1.0, 4, [[1, 2, 3]])[0] == 0.5  # texture.h:101,104


def test_perlin(ora, gold):
    pts = gold["perlin_pts"]
    for seed in (12345, 5489):
        perm = ora.perlin_perm(seed)
        assert (bits(ora.perlin_noise(perm, pts)) == bits(gold[f"perlin_noise_{seed}"])).all()
        f32 = pts.astype(np.float32)
        assert (bits(ora.perlin_noise(perm, f32.astype(np.float64)))
                == bits(gold[f"perlin_noise_vec3_{seed}"])).all()
        assert (bits(ora.perlin_fractal(perm, f32)) == bits(gold[f"perlin_fractal_{seed}"])).all()
    perm = ora.perlin_perm(12345)
    # SURVEY 8(c)
    probes = {(0, 0, 0): 0.0, (.5, .5, .5): 0.375, (1.25, -3.75, 100.1): -0.20131776773070792,
              (-0.49, 127.6, 64): -0.26518351475579793, (255.9, 256.1, -0.001): -0.18798900732392726}
    for p, v in probes.items():
        assert abs(ora.perlin_noise(perm, [p])[0] - v) < 1e-15


def test_turb_is_the_rtow_composition(ora, gold):
    """turb is absent from the reference: pinned by composing the (pinned) noise()."""
    perm = ora.perlin_perm(12345)
    pts = gold["perlin_pts"][:256].astype(np.float32)
    for depth in (1, 7):
        acc = np.zeros(len(pts))
        w, tp = 1.0, pts.copy()
        for _ in range(depth):
            acc += w * ora.perlin_noise(perm, tp.astype(np.float64))
            w *= 0.5
            tp = tp * np.float32(2)
        assert (bits(np.abs(acc)) == bits(ora.perlin_turb(perm, pts, depth))).all()


def test_multiband_is_the_paper_composition(ora, gold, tile3d_128):
    """WMultibandNoise is absent from the reference: pinned by composing evaluate3D."""
    pts = (gold["probe_pts"][-256:] * np.float32(0.5)).astype(np.float32)
    w = np.array([1, 0.5, 0.25, 2, 1], np.float32)
    for s, first, nb in ((-16.0, 0, 5), (-3.0, 0, 5), (-16.0, -2, 3), (0.0, 0, 5)):
        acc = np.zeros(len(pts), np.float32)
        for b in range(nb):
            if not (np.float32(s) + np.float32(first) + np.float32(b) < 0):
                break
            q = (np.float32(2) * pts * np.float32(2.0 ** (first + b))).astype(np.float32)
            acc = (acc + w[b] * ora.evaluate3d(tile3d_128, q)).astype(np.float32)
        var = np.float32(0)
        for b in range(nb):
            var = np.float32(var + w[b] * w[b])
        acc = (acc / np.sqrt(np.float32(var * np.float32(0.18402)))).astype(np.float32)
        got = ora.multiband3d(tile3d_128, pts, s, first, nb, w, 0.18402)
        assert (bits(got) == bits(acc)).all(), (s, first, nb)


def test_multiband_with_a_normal_is_the_paper_composition(ora, gold, tile3d_128):
    """The normal != NULL branch of Appendix 2: every band is evaluate3DProjected (pinned by the reference's
    vectors elsewhere in this file), normalised with 0.296.  Parity unpinned by any reference artefact."""
    pts = (gold["probe_pts"][-64:] * np.float32(0.5)).astype(np.float32)
    w = np.array([1, 0.5, 2], np.float32)
    for normal in ((0.0, 0.0, 1.0), (0.6, 0.0, 0.8)):
        nr = np.broadcast_to(np.asarray(normal, np.float32), pts.shape).copy()
        for s, first, nb in ((-16.0, 0, 3), (-1.0, 0, 3), (-16.0, -1, 2)):
            acc = np.zeros(len(pts), np.float32)
            for b in range(nb):
                if not (np.float32(s) + np.float32(first) + np.float32(b) < 0):
                    break
                q = (np.float32(2) * pts * np.float32(2.0 ** (first + b))).astype(np.float32)
                acc = (acc + w[b] * ora.evaluate3d_projected(tile3d_128, q, nr)).astype(np.float32)
            var = np.float32(0)
            for b in range(nb):
                var = np.float32(var + w[b] * w[b])
            acc = (acc / np.sqrt(np.float32(var * np.float32(0.296)))).astype(np.float32)
            got = ora.multiband3d_projected(tile3d_128, pts, nr, s, first, nb, w, 0.296)
            assert (bits(got) == bits(acc)).all(), (normal, s, first, nb)


def test_textures(ora, gold, artefacts, tile2d_128, tile3d_128):
    tp = gold["tex_pts"]
    perm = ora.perlin_perm(5489)  # noise_texture holds a default-seeded perlin (texture.h:46)
    for kind, scale, octave in artefacts["texture_cases"]:
        key = f"tex_{kind}_s{scale}_o{octave}"
        if kind == "perlin":
            got = ora.noise_texture_value(perm, scale, octave, tp)
        elif kind == "wavelet3d":
            got = ora.wavelet_texture_value(tile3d_128, True, scale, octave, tp)
        else:
            got = ora.wavelet_texture_value(tile2d_128, False, scale, octave, tp)
        assert (bits(got) == bits(gold[key])).all(), key


@pytest.mark.parametrize("octave", OCTAVES)
def test_committed_raw_grids(ora, octave, tile2d_128, tile3d_128):
    """The reference's own committed outputs, byte for byte (config 1 is wavelet_2D octave 4)."""
    L = ora.lib()
    perm = ora.perlin_perm(12345)
    sha = json.load(open(os.path.join(GOLD, "artefacts.json")))["raw"]
    jobs = (("wavelet_noise_2D", L.wno_grid_wavelet2d, (tile2d_128, tile2d_128.size)),
            ("wavelet_noise_3Dsliced", L.wno_grid_wavelet3d_sliced, (tile3d_128, tile3d_128.size)),
            ("wavelet_noise_3Dprojected", L.wno_grid_wavelet3d_projected, (tile3d_128, tile3d_128.size)),
            ("perlin_noise_2D", L.wno_grid_perlin2d, (perm,)),
            ("perlin_noise_3Dsliced", L.wno_grid_perlin3d_sliced, (perm,)))
    for name, fn, args in jobs:
        out = np.zeros(256 * 256, np.float32)
        fn(*args, 256, octave, out)
        fname = f"{name}_octave_{octave}.raw"
        assert hashlib.sha256(out.tobytes()).hexdigest() == sha[fname], fname
        assert (bits(out) == bits(raw(fname))).all()


def test_json_stats_of_committed_grids():
    """threejs/result_json `original_range` blocks describe the same raws (extra golden stats)."""
    stats = json.load(open(os.path.join(GOLD, "json_stats.json")))
    names = {"wavelet_noise_2d": "wavelet_noise_2D", "wavelet_noise_3d_sliced": "wavelet_noise_3Dsliced",
             "wavelet_noise_3d_projected": "wavelet_noise_3Dprojected", "perlin_noise_2d": "perlin_noise_2D",
             "perlin_noise_3d_sliced": "perlin_noise_3Dsliced"}
    for jname, st in stats.items():
        base, octv = jname[:-len("_octaveN.json")], jname[-6]
        a = raw(f"{names[base]}_octave_{octv}.raw").astype(np.float64)
        r = st["original_range"]
        assert abs(a.min() - r["min"]) < 1e-12 and abs(a.max() - r["max"]) < 1e-12
        assert abs(a.mean() - r["mean"]) < 1e-9 and abs(a.std() - r["std"]) < 1e-9


def test_dense_volumes(ora, gold, artefacts, tile3d_128):
    for name, den, nx, ny, z0, z1, octave in artefacts["volumes"]:
        got = ora.grid_wavelet3d_volume(tile3d_128, den, nx, ny, z0, z1, octave)
        assert (bits(got) == bits(gold[name])).all(), name


def test_reference_build_agrees_when_present(ora, gold, tile3d_128):
    """If oracle/_ref is there (build container, or prebuilt on the GPU box) run it live."""
    R = ora.ref()
    if R is None:
        pytest.skip("oracle/_ref/libwnref.so not built")
    h = R.ref_wn_new(128, 12345)
    R.ref_wn_generate3d(h)
    c = np.zeros(R.ref_wn_coeff_count(h), np.float32)
    R.ref_wn_coeffs(h, c)
    assert (bits(c) == bits(tile3d_128)).all()
    rng = np.random.default_rng(7)
    pts = np.ascontiguousarray(rng.uniform(-300, 300, (4096, 3)).astype(np.float32))
    e = np.zeros(len(pts), np.float32)
    R.ref_wn_eval3d(h, pts, len(pts), e)
    assert (bits(e) == bits(ora.evaluate3d(tile3d_128, pts))).all()
    R.ref_wn_delete(h)
