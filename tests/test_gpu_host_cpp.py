"""GPU: the C++ host classes (reference class names over the C ABI) and the tools built on them.

 * tools/gridgen               -- experient/main.cpp's role with batched launches
 * tools/scalar_api_check      -- the SCALAR members (evaluate*, noise, value): the host evaluators (default) and the
   resident scalar kernel (WN_SCALAR_ON_DEVICE=1), both against the reference's vectors and the batched kernels
 * build/linkcheck/experient_main -- the REFERENCE'S OWN experient/main.cpp compiled against this
   repo's headers (built by `make linkcheck` in the build container; the binary travels, the
   source does not): its 15 outputs must be byte-identical to the reference's committed raws.
"""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, bits, raw

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PKG = os.path.join(ROOT, "wavelet-noise-in-ray-tracing_amd")
NAMES = ("wavelet_noise_2D", "wavelet_noise_3Dsliced", "wavelet_noise_3Dprojected",
         "perlin_noise_2D", "perlin_noise_3Dsliced")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.fixture(scope="module")
def shas():
    return json.load(open(os.path.join(GOLD, "artefacts.json")))["raw"]


def test_gridgen_reproduces_all_15_committed_files_by_default(tmp_path, shas):
    exe = os.path.join(PKG, "tools", "gridgen")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = subprocess.run([exe, str(tmp_path / "raw")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for octave in (3, 4, 5):
        for n in NAMES:
            f = f"{n}_octave_{octave}.raw"
            assert sha(tmp_path / "raw" / f) == shas[f], f


def test_gridgen_fast_path(tmp_path, shas):
    exe = os.path.join(PKG, "tools", "gridgen")
    out = subprocess.run([exe, str(tmp_path / "raw"), "--fast"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    for octave in (3, 4, 5):
        for n in NAMES:
            f = f"{n}_octave_{octave}.raw"
            got = np.fromfile(tmp_path / "raw" / f, np.float32)
            if n == "wavelet_noise_3Dsliced":  # separable brick kernel where the step allows
                assert np.abs(got - raw(f)).max() <= 1e-5, f
            else:
                assert sha(tmp_path / "raw" / f) == shas[f], f


def scalar_env(on_device):
    env = dict(os.environ)
    env.pop("WN_SCALAR_ON_DEVICE", None)
    if on_device:
        env["WN_SCALAR_ON_DEVICE"] = "1"
    return env


@pytest.mark.parametrize("on_device", [False, True], ids=["host_evaluators", "resident_scalar_kernel"])
def test_scalar_members_match_reference_vectors(gold, on_device):
    exe = os.path.join(PKG, "tools", "scalar_api_check")
    pts = gold["probe_pts"][:200]
    text = f"{len(pts)}\n" + "\n".join(" ".join(repr(float(v)) for v in p) for p in pts) + "\n"
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=300, env=scalar_env(on_device))
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().split("\n")
    vals = np.array([[float(x) for x in ln.split()] for ln in lines[:len(pts)]])
    assert (bits(vals[:, 0].astype(np.float32)) == bits(gold["probe_e2d"][:200])).all()
    assert (bits(vals[:, 1].astype(np.float32)) == bits(gold["probe_e3d"][:200])).all()
    assert (bits(vals[:, 2].astype(np.float32)) == bits(gold["probe_e3dp"][:200])).all()  # normal (0,0,1)
    import oracle
    perm = oracle.perlin_perm(12345)
    f32 = pts.astype(np.float32)
    assert (bits(vals[:, 3]) == bits(oracle.perlin_noise(perm, f32.astype(np.float64)))).all()
    assert (bits(vals[:, 4]) == bits(oracle.perlin_fractal(perm, f32))).all()
    t3 = oracle.tile3d(128, 12345)
    assert (bits(vals[:, 5].astype(np.float32)) == bits(oracle.wavelet_texture_value(t3, True, 1.0, 4, f32))).all()
    assert (bits(vals[:, 6].astype(np.float32))
            == bits(oracle.noise_texture_value(oracle.perlin_perm(5489), 1.0, 4, f32))).all()
    assert lines[len(pts)].split() == ["batch_vs_scalar_mismatches", "0"]  # batched overloads, active mask
    masked = lines[len(pts) + 1].split()  # 200,000 points, ~59 % active, through wavelet_texture::values / noise_texture::values
    assert masked[:2] == ["masked_batch_mismatches", "0"] and 0.55 < int(masked[3]) / int(masked[5]) < 0.64, masked
    assert lines[len(pts) + 2].split() == ["empty", "0", "0"]
    assert lines[len(pts) + 3].split() == ["tile", "128", "coeffs", str(128 ** 3)]


@pytest.mark.parametrize("on_device", [False, True], ids=["host_evaluators", "resident_scalar_kernel"])
def test_reference_experient_main_linked_against_this_library(tmp_path, shas, on_device):
    exe = os.path.join(PKG, "build", "linkcheck", "experient_main")
    if not os.path.exists(exe):
        pytest.skip("build/linkcheck/experient_main not built (needs /root/reference at build time)")
    out = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=900, env=scalar_env(on_device))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for octave in (3, 4, 5):
        for n in NAMES:
            f = f"{n}_octave_{octave}.raw"
            assert sha(tmp_path / "result_raw" / f) == shas[f], f


def test_scalar_calls_through_the_resident_kernel_latency_and_parity():
    """With WN_SCALAR_ON_DEVICE=1 (the tool sets it) the reference's scalar members are served by a resident one-wave kernel
    (csrc/wn_mailbox.hip): every value equals the batched kernels' (bit for bit), a call costs microseconds instead of a
    launch + synchronise, and an idle gap (the kernel ends by itself after 2 ms) is survived by restarting it.  The host
    evaluator (the default) is timed and compared beside it."""
    exe = os.path.join(PKG, "tools", "scalar_latency")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["mismatches"] == 0
    assert line["resident_kernel_instances"] >= 10  # the ten idle gaps each ended an instance
    # latencies are recorded (profiles/r0*_scalar_latency.json), not asserted: wall-clock bounds on a shared box fail
    # without any code change (round-2 ADVICE); what must hold is an order of magnitude
    assert line["mailbox_us_per_call"] < 100.0, line
    assert line["host_evaluator_us_per_call"] < line["mailbox_us_per_call"], line
    print(line)
