"""The sharded dense-grid path on hardware: one rank's configs[4] slab (2048 x 2048 x 256 samples, 4 GiB),
bench.py's self-launched N-rank run, and the gather of the slabs on the RCCL backend (needs >= 2 GPUs;
skipped on a one-GPU box, where two gloo ranks sharing the GPU rehearse the same code path)."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, bits

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wn():
    assert torch.cuda.is_available(), "GPU tests need a GPU (the product has no CPU path)"
    return importlib.import_module("wavelet-noise-in-ray-tracing_amd")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_config5_shard_shape_2048x2048x256(wn, ora, tile3d_128):
    """Rank 3 of 8 of BASELINE configs[4]: planes [768, 1024) of the 2048^3 lattice (lattice step 1/16).
    Whole planes against the bit-exact kernel (itself oracle-checked in test_gpu_parity), rows against the oracle."""
    noise = wn.WaveletNoise(128, 12345)
    noise.generateNoiseTile3D()
    z0, z1 = wn.slab_bounds(2048, 8, 3)
    assert (z0, z1) == (768, 1024)
    slab = wn.wavelet_volume(noise, 2048, 2048, 2048, z0, z1, 4)
    assert slab.shape == (256, 2048, 2048)
    for z in (768, 769, 901, 1023):
        exact = wn.wavelet_volume(noise, 2048, 2048, 2048, z, z + 1, 4, exact=True)[0]
        err = float((slab[z - z0] - exact).abs().max())
        assert err <= 1e-5, (z, err)
    for z in (768, 1023):
        want = ora.grid_wavelet3d_volume(tile3d_128, 2048, 2048, 6, z, z + 1, 4)[0]  # rows y < 6 of plane z
        got = slab[z - z0, :6].cpu().numpy()
        assert np.abs(got - want).max() <= 1e-5, z
    # size-independent properties of the whole 4 GiB slab: the lattice spans 128 cells = one tile period in x
    # and y, so rows/columns 2048 samples apart would coincide; inside the slab, the mean is ~0 and finite
    assert bool(torch.isfinite(slab).all())
    assert abs(float(slab.double().mean())) < 5e-3


def _run_bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_two_gloo_ranks_on_one_gpu():
    """`python bench.py --gpus 2` started plainly: it starts two ranks itself; on a one-GPU box they share the GPU
    and exchange through gloo (WN_BENCH_BACKEND), exercising slabs + gather end to end."""
    line = _run_bench(["--gpus", "2", "--lattice", "512", "--steps", "2", "--warmup", "1", "--no-measured-peak"],
                      {"WN_BENCH_BACKEND": "gloo"})
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["lattice_total"] == [512, 512, 512] and line["config"]["lattice_per_gpu"] == [512, 512, 256]
    assert line["gather"]["peer_plane_matches_local_recompute"] is True
    assert line["gather"]["bytes_into_root"] == 4.0 * 512 * 512 * 256
    assert line["value"] > 0
    # the N > 1 line carries its own single-GPU baseline of the SAME lattice (here 512^3: the north star's scaling lattice)
    ss = line["strong_scaling"]
    assert ss["lattice"] == [512, 512, 512] and ss["n"] == 2 and ss["n1_ms"] > 0 and ss["nN_ms"] > 0
    assert 0 < ss["efficiency"] < 1.5  # two ranks sharing ONE GPU: ~0.5; the key exists and is sane


def test_native_gather_world_size_one_identity_and_two_forked_ranks_when_two_gpus(wn):
    """The C ABI's gather (include/wnoise_shard.h, libwnoise_shard.so: ncclCommInitRank + one grouped ncclSend / ncclRecv).
    On a one-GPU box: a communicator of world size 1 is created over RCCL and the gather is the identity (the root's own
    slab lands in its place by a device copy), in place and out of place, with a piece size that cuts the slab into
    several pieces.  The N > 1 leg of the same entry point is tools/gridgen --gpus N (unmeasured on hardware: one-GPU lease)."""
    noise = wn.WaveletNoise(128, 12345)
    noise.generateNoiseTile3D()
    slab = wn.wavelet_volume(noise, 256, 256, 256, 0, 24, 4).clone()
    comm = wn.NativeComm(world=1, rank=0, id_bytes=wn.NativeComm.unique_id())
    try:
        out = torch.full((24, 256, 256), -7.0, dtype=torch.float32, device="cuda")
        full = comm.gather_volume(slab, 24, dst=0, out=out, piece_bytes=5 * 256 * 256 * 4)
        torch.cuda.synchronize()
        assert full.data_ptr() == out.data_ptr() and torch.equal(full, slab)
        again = comm.gather_volume(slab, 24, dst=0)  # fresh output buffer
        torch.cuda.synchronize()
        assert torch.equal(again, slab)
        same = comm.gather_volume(slab, 24, dst=0, out=slab)  # the slab already is the volume: nothing to copy
        assert same.data_ptr() == slab.data_ptr()
    finally:
        comm.close()


def test_gridgen_shards_a_lattice_over_one_rank_with_the_native_gather(tmp_path):
    """tools/gridgen --gpus 1 --lattice 256: the C++ grid generator's sharded mode (fresh child per rank, the parent never
    touches a GPU) through wn_comm_create / wn_gather_volume; the gathered volume's planes are the plain call's."""
    exe = os.path.join(ROOT, "wavelet-noise-in-ray-tracing_amd", "tools", "gridgen")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = subprocess.run([exe, str(tmp_path), "--gpus", "1", "--lattice", "256", "--octave", "4"], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["lattice"] == 256 and line["ranks"] == 1 and line["gathered_planes"] == 256
    vol = np.fromfile(tmp_path / "wavelet_noise_3D_lattice_256_octave_4_plane_0.raw", dtype="<f4")
    wnm = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
    noise = wnm.WaveletNoise(128, 12345)
    noise.generateNoiseTile3D()
    want = wnm.wavelet_volume(noise, 256, 256, 256, 0, 1, 4)[0].cpu().numpy().ravel()
    assert (bits(vol) == bits(want)).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL gather needs two GPUs")
def test_gather_volume_on_rccl_two_gpus():
    """Two ranks, one per GPU, nccl (= RCCL) backend: bench.py's sharded run gathers the slabs on rank 0 and checks
    a peer's plane against a local recompute."""
    line = _run_bench(["--gpus", "2", "--lattice", "1024", "--steps", "2", "--warmup", "1"], {})
    assert line["n_gpus"] == 2
    assert line["gather"]["backend"] == "RCCL"
    assert line["gather"]["peer_plane_matches_local_recompute"] is True
