#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mnoise-samples/s of the dense 3-D wavelet grid
(BASELINE.json configs[1]: 512^3, tile 128, octave 4) on N MI355X, with the HBM roofline of the
dominant kernel and the reference's CPU path timed beside it.

  python bench.py --gpus 1 --steps 50 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of wn_eval3d_grid over this rank's z-slab: 512 x 512 x 512 samples per
GPU (weak scaling: rank r owns planes [512 r, 512 (r+1)) of a 512 x 512 x 512N lattice with
the same 0.25-cell step; the path shards with no data-path collective).
Output tensors live in HBM before the timed region starts; nothing crosses PCIe in it.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy rate
TILE, SEED, OCTAVE = 128, 12345, 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lattice", dest="n", type=int, default=512, help="lattice size per axis per GPU")
    ap.add_argument("--planes", type=int, default=0,
                    help="z-planes per GPU (default: --lattice); --lattice 2048 --planes 256 is one "
                         "GPU's slab of BASELINE configs[4] (2048^3 over 8 GPUs)")
    ap.add_argument("--workload", default="wavelet3d",
                    choices=["wavelet3d", "wavelet3d_exact", "multiband5", "turb7", "perlin",
                             "texture_points", "texture_points_perlin"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget")
    ap.add_argument("--gather", action="store_true",
                    help="also time one collection of the slabs on rank 0 (outside the timed region)")
    return ap.parse_args()


def cpu_threads_leg(run, n, seconds):
    """the same CPU evaluator on the box's CPU share, one z-slab stream per thread (the
    reference itself has no threading; its evaluate3D only reads the tile, and ctypes drops the GIL)."""
    import threading
    threads = max(1, min(16, os.cpu_count() or 1))
    planes_per_call = 2
    counts = [0] * threads
    stop_at = time.perf_counter() + seconds

    def worker(t):
        buf = np.empty(planes_per_call * n * n, np.float32)
        z = t * (n // threads)
        while time.perf_counter() < stop_at:
            run(z % (n - planes_per_call), z % (n - planes_per_call) + planes_per_call, buf)
            z += planes_per_call
            counts[t] += planes_per_call
    t0 = time.perf_counter()
    pool = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    dt = time.perf_counter() - t0
    planes = sum(counts)
    return {"value": planes * n * n / dt / 1e6, "unit": "Msamples/s", "cores": threads,
            "sample": f"{planes} z-planes in {dt:.1f} s on {threads} threads"}


def cpu_baseline(n, budget_s, gpu_slab):
    """The reference's CPU path (oracle/_ref = the real reference compiled by oracle/Makefile) or,
    when that .so is not there, the oracle restatement, on a bounded sample of the same workload:
    planes of the same 512^3 lattice, one thread (the reference is single-threaded)."""
    import oracle
    R = oracle.ref()
    planes_per_chunk, done, t_used = 8, 0, 0.0
    max_err = 0.0
    if R is not None:
        kind = "reference"
        h = R.ref_wn_new(TILE, SEED)
        R.ref_wn_generate3d(h)

        def run(z0, z1, out):
            R.ref_wn_grid3d_volume(h, n, n, n, z0, z1, OCTAVE, out)
    else:
        kind = "port"
        tile = oracle.tile3d(TILE, SEED)

        def run(z0, z1, out):
            oracle.lib().wno_grid_wavelet3d_volume(tile, tile.size, n, n, n, z0, z1, OCTAVE, out)
    buf = np.empty(planes_per_chunk * n * n, np.float32)
    while t_used < budget_s and done + planes_per_chunk <= gpu_slab.shape[0]:
        t0 = time.perf_counter()
        run(done, done + planes_per_chunk, buf)
        t_used += time.perf_counter() - t0
        got = gpu_slab[done:done + planes_per_chunk].cpu().numpy().ravel()
        max_err = max(max_err, float(np.abs(got - buf).max()))
        done += planes_per_chunk
    samples = done * n * n
    threaded = cpu_threads_leg(run, n, min(6.0, budget_s / 2))
    return {"value": samples / t_used / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind,
            "all_cores": threaded,
            "sample": f"{done} of {n} z-planes of the same {n}^3 lattice ({samples} samples, "
                      f"{t_used:.1f} s, 1 thread: the reference has no threading)",
            "host_cpus": os.cpu_count(), "gpu_vs_cpu_max_abs_err": max_err}


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("bytes_per_launch")
        except Exception:
            return None
    return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    # one rank per GPU; WN_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the
    # multi-rank path on a single-GPU box (the driver's runs use the default: nccl = RCCL)
    backend = os.environ.get("WN_BENCH_BACKEND", "nccl")
    device_index = local % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
    n = args.n
    planes = args.planes or n
    z0, z1 = rank * planes, (rank + 1) * planes  # this rank's slab of the n x n x (planes*world) lattice
    samples_per_rank = n * n * planes
    out = torch.empty(samples_per_rank, dtype=torch.float32, device="cuda")

    alg_bytes = 4 * samples_per_rank + 4 * TILE ** 3  # one fp32 store/sample + tile once
    dtype = "f32"
    if args.workload in ("wavelet3d", "wavelet3d_exact", "multiband5"):
        noise = wn.WaveletNoise(TILE, SEED)
        noise.generateNoiseTile3D()  # every rank regenerates the tile from the seed
        if args.workload == "multiband5":
            step = lambda: wn.multiband_volume(noise, n, n, n, z0, z1, -16.0, 0, 5, out=out)  # noqa: E731
            kernel, desc = "grid3d_sep_kernel<NB=5>", f"{n}^3 WMultibandNoise 5 bands (configs[2])"
        else:
            exact = args.workload == "wavelet3d_exact"
            step = lambda: wn.wavelet_volume(noise, n, n, n, z0, z1, OCTAVE, exact=exact, out=out)  # noqa: E731
            # the library's dispatch (csrc/wn_wavelet_strip.hip strip_try): rows of k*256 samples and
            # 0.18 <= planes per lattice step <= 1/3 go to the strip-march kernel, other lattices to the brick kernel
            lattice_step = 4.0 * 2.0 ** OCTAVE * 2.0 / n
            strip = n % 256 == 0 and 0.18 <= lattice_step < 1.0 / 3.0
            kernel = "grid3d_direct_kernel" if exact else ("grid3d_strip_kernel" if strip else "grid3d_sep_kernel<NB=1>")
            desc = (f"{n}^3 dense 3D WNoise grid, tile={TILE}, octave={OCTAVE} (configs[1])" if planes == n else
                    f"{n}x{n}x{planes} z-slab per GPU of a {n}^2 x {planes}*N lattice, tile={TILE}, octave={OCTAVE} (configs[4] shard)")
    elif args.workload in ("turb7", "perlin"):
        per = wn.perlin(SEED)
        dtype = "f64"
        if args.workload == "turb7":
            step = lambda: wn.turb_volume(per, n, n, n, z0, z1, 7, out=out)  # noqa: E731
            desc = f"{n}^3 perlin turb(depth=7) (configs[2])"
        else:
            step = lambda: wn.perlin_volume(per, n, n, n, z0, z1, OCTAVE, out=out)  # noqa: E731
            desc = f"{n}^3 perlin noise grid, octave={OCTAVE}"
        kernel = "perlin_grid_kernel"
        alg_bytes = 4 * samples_per_rank + 512
    else:  # texture_points: configs[3] stand-in: 85% quad / 15% sphere hits
        m = 80_000_000 // max(1, world)
        g = torch.Generator(device="cuda").manual_seed(1 + rank)
        pts = torch.empty((m, 3), dtype=torch.float32, device="cuda")
        pts[:, 0].uniform_(-10, 10, generator=g)
        pts[:, 1] = -0.5
        pts[:, 2].uniform_(-10, 10, generator=g)
        k = int(0.15 * m)
        d = torch.randn((k, 3), device="cuda", generator=g)
        pts[:k] = torch.tensor([1.0, 0.0, -1.75], device="cuda") + 0.5 * d / d.norm(dim=1, keepdim=True)
        perlin_tex = args.workload == "texture_points_perlin"
        tex = wn.noise_texture(1.0, OCTAVE) if perlin_tex else wn.wavelet_texture(1.0, OCTAVE, True)
        grey = torch.empty(m, dtype=torch.float32, device="cuda")
        step = lambda: tex.grey(pts, out=grey)  # noqa: E731
        samples_per_rank = m
        alg_bytes = 16 * m  # 12 B xyz in + 4 B out
        kernel = "noise_texture_kernel" if perlin_tex else "wavelet_texture_kernel"
        desc = f"{m} ray hit points, {'noise_texture (Perlin)' if perlin_tex else 'wavelet_texture'} octave {OCTAVE} (configs[3] stand-in)"
        dtype = "f64" if perlin_tex else "f32"

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    timer = wn.HipTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer.start()  # HIP events on the stream the kernels are launched on
    for _ in range(args.steps):
        step()
    timer.stop()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = timer.elapsed_ms()
    if dist is not None:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])

    gather = None
    if args.gather and dist is not None:
        slab = out.view(planes, n, n) if backend == "nccl" else out.view(planes, n, n).cpu()
        torch.cuda.synchronize(); barrier()
        g0 = time.perf_counter()
        wn.gather_volume(slab, planes * world, dst=0)
        torch.cuda.synchronize(); barrier()
        gs = time.perf_counter() - g0
        gather = {"ms": gs * 1e3, "GBps_into_root": 4.0 * samples_per_rank * (world - 1) / gs / 1e9}

    if rank == 0:
        launch_s = ev_ms / 1e3 / args.steps
        achieved = alg_bytes / launch_s / 1e9
        line = {
            "metric": "Mnoise-samples/sec (3D wavelet, octave=4)", "value": samples_per_rank * world * args.steps / dt / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "lattice_per_gpu": [n, n, planes], "tile": TILE, "seed": SEED,
                       "octave": OCTAVE, "sharding": "z-slabs, no data-path collective",
                       "device": wn.device_info()["name"]},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "avg_launch_us": launch_s * 1e6,
                         "traffic": pmc_traffic() if (args.workload == "wavelet3d" and n == 512 and planes == 512) else None},
        }
        if gather:
            line["gather"] = gather
        if world == 1 and not args.no_cpu_baseline and args.workload == "wavelet3d":
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds, out.view(planes, n, n))
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
