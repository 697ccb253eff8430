#!/usr/bin/env python3
"""bench.py -- Mnoise-samples/s of the hot path on N MI355X, with the roofline of the dominant
kernel and the reference's CPU path timed beside it.

  python bench.py                       N=1: BASELINE configs[1], 512^3 dense 3-D wavelet grid
  python bench.py --gpus 8              BASELINE configs[4]: 2048^3 in z-slabs of 2048/N planes,
                                        one gather of the slabs on rank 0 reported beside it
  python bench.py --workload turb7      the other BASELINE configs (multiband5, turb7, perlin,
                                        texture_points, texture_points_perlin), same evidence

`--gpus N` works both ways: started plainly it launches its N ranks itself (fresh children under
torch.distributed.run, before this process touches a GPU) and returns their exit code; started BY
torch.distributed.run (WORLD_SIZE set) it is one of the ranks.  One rank per GPU over RCCL.

A "step" is one pass of the hot path over this rank's share: one launch of the dense-grid kernel
over its z-slab (or of the texture kernel over its hit points).  Outputs live in HBM before the
timed region starts; nothing crosses PCIe in it.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TF = 78.6   # MI355X vector fp64 peak, FMA counted as 2 flops (MI355X_MICROARCH.md / SURVEY 7)
# Algorithmic fp64 flops of one perlin::noise(x,y,z) call (perlin.h:42-62): 3 x (x - floor(x)) = 3,
# 3 x fade (7 each, :18-20) = 21, x-1 / y-1 / z-1 = 3, 8 x grad (one add each, :26-31) = 8, 7 x lerp
# (3 each, :22-24) = 21  ->  56.  turb adds weight*noise and the accumulation: 2 per octave.
PERLIN_FLOPS = 56
TILE, SEED, OCTAVE = 128, 12345, 4
WORKLOADS = ["wavelet3d", "wavelet3d_exact", "multiband5", "turb7", "perlin", "texture_points",
             "texture_points_perlin"]


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lattice", dest="n", type=int, default=0,
                    help="lattice size per axis (default: 512 on one GPU, 2048 on several)")
    ap.add_argument("--planes", type=int, default=0,
                    help="z-planes per GPU (default: the whole lattice on one GPU, lattice/N on N)")
    ap.add_argument("--workload", default="wavelet3d", choices=WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline time budget (1 thread)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the gather of the slabs on rank 0")
    ap.add_argument("--no-measured-peak", action="store_true", help="skip the in-run fill / copy ceiling")
    ap.add_argument("--two-stream-probe", action="store_true",
                    help="also time the same launches alternating between two streams / two buffers (informational; off by "
                         "default so that a rocprofv3 trace of this command holds only the contract's launches)")
    ap.add_argument("--print-launch", action="store_true",
                    help="print the command --gpus N would start and exit (no GPU is touched)")
    return ap.parse_args(argv)


# ---- self-launch ---------------------------------------------------------------------------------
def launch_command(args, argv, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as fresh
    child processes and hand back their exit code.  This process has not touched a GPU (importing torch
    does not), and it never replaces itself with another program."""
    cmd = launch_command(args, argv, free_port())
    if args.print_launch:
        print(" ".join(cmd))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


# ---- CPU baseline ----------------------------------------------------------------------------------
def lattice_points(n, z0, z1, scale):
    """float32 lattice coordinates ((i / n) * 4) * scale of planes [z0, z1), as an [m, 3] array
    (experient/main.cpp:20-26: one float rounding per operation)."""
    i = np.arange(n, dtype=np.float32)
    c = ((i / np.float32(n)) * np.float32(4.0)) * np.float32(scale)
    zc = ((np.arange(z0, z1, dtype=np.float32) / np.float32(n)) * np.float32(4.0)) * np.float32(scale)
    pts = np.empty((z1 - z0, n, n, 3), np.float32)
    pts[..., 0] = c[None, None, :]
    pts[..., 1] = c[None, :, None]
    pts[..., 2] = zc[:, None, None]
    return pts.reshape(-1, 3)


def cpu_evaluator(workload, n, tex_points_host):
    """Returns (kind, unit_count, run) where run(k, scratch) evaluates unit k (one z-plane of the
    lattice, or one chunk of hit points) on the CPU and returns a float32 array comparable with the
    GPU's output for that unit.  kind "reference": the real reference compiled by oracle/Makefile
    (oracle/_ref/libwnref.so); "port": the oracle restatement when that library is absent."""
    import oracle
    R = oracle.ref()
    if workload in ("wavelet3d", "wavelet3d_exact"):
        if R is not None:
            h = R.ref_wn_new(TILE, SEED)
            R.ref_wn_generate3d(h)

            def run(z):
                out = np.empty(n * n, np.float32)
                R.ref_wn_grid3d_volume(h, n, n, n, z, z + 1, OCTAVE, out)  # WaveletNoise.cpp:185-215
                return out
            return "reference", n, run
        tile = oracle.tile3d(TILE, SEED)

        def run(z):
            out = np.empty(n * n, np.float32)
            oracle.lib().wno_grid_wavelet3d_volume(tile, tile.size, n, n, n, z, z + 1, OCTAVE, out)
            return out
        return "port", n, run
    if workload == "multiband5":
        # WMultibandNoise is absent from the reference: composed from its evaluate3D (paper App. 2)
        if R is not None:
            h = R.ref_wn_new(TILE, SEED)
            R.ref_wn_generate3d(h)
            norm = np.float32(np.sqrt(np.float32(np.float32(5.0) * np.float32(0.18402))))

            def run(z):
                p = lattice_points(n, z, z + 1, 1.0)
                acc = np.zeros(n * n, np.float32)
                val = np.empty(n * n, np.float32)
                for b in range(5):
                    q = np.ascontiguousarray((np.float32(2.0) * p) * np.float32(2.0 ** b))
                    R.ref_wn_eval3d(h, q, q.shape[0], val)
                    acc += np.float32(1.0) * val
                return acc / norm
            return "reference", n, run
        tile = oracle.tile3d(TILE, SEED)
        return "port", n, lambda z: oracle.grid_multiband3d_volume(tile, n, n, n, z, z + 1, -16.0, 0, 5,
                                                                   [1.0] * 5, 0.18402).ravel()
    if workload in ("perlin", "turb7"):
        depth, scale = (7, 1.0) if workload == "turb7" else (1, 2.0 ** OCTAVE)
        if R is not None:
            h = R.ref_perlin_new(SEED)

            def run(z):
                p = lattice_points(n, z, z + 1, scale)
                val = np.empty(n * n, np.float64)
                if depth == 1:
                    R.ref_perlin_noise_vec3(h, p, p.shape[0], val)   # perlin.h:42-72
                    return val.astype(np.float32)
                acc, w = np.zeros(n * n, np.float64), 1.0
                for _ in range(depth):                               # RTOW turb over perlin::noise
                    R.ref_perlin_noise_vec3(h, p, p.shape[0], val)
                    acc += w * val
                    w *= 0.5
                    p = p * np.float32(2.0)
                return np.abs(acc).astype(np.float32)
            return "reference", n, run
        perm = oracle.perlin_perm(SEED)
        if depth == 1:
            return "port", n, lambda z: oracle.grid_perlin_volume(perm, n, n, n, z, z + 1, OCTAVE).ravel()
        return "port", n, lambda z: oracle.grid_turb_volume(perm, n, n, n, z, z + 1, depth).ravel()
    # texture_points*: chunks of the same hit points (texture.h:37-43 / :67-107)
    chunk = 1 << 17
    units = tex_points_host.shape[0] // chunk
    perlin_tex = workload == "texture_points_perlin"
    if R is not None:
        h = R.ref_noise_texture_new(1.0, OCTAVE) if perlin_tex else R.ref_wavelet_texture_new(1.0, OCTAVE, 1)

        def run(k):
            p = np.ascontiguousarray(tex_points_host[k * chunk:(k + 1) * chunk])
            rgb = np.empty((chunk, 3), np.float32)
            R.ref_texture_value(h, p, chunk, rgb)
            return rgb[:, 0].copy()
        return "reference", units, run
    if perlin_tex:
        perm = oracle.perlin_perm(5489)
        return "port", units, lambda k: oracle.noise_texture_value(perm, 1.0, OCTAVE, tex_points_host[k * chunk:(k + 1) * chunk])
    tile = oracle.tile3d(TILE, SEED)
    return "port", units, lambda k: oracle.wavelet_texture_value(tile, True, 1.0, OCTAVE, tex_points_host[k * chunk:(k + 1) * chunk])


def cpu_baseline(workload, n, budget_s, gpu_unit, unit_samples, unit_name, tex_points_host=None):
    """The CPU path on a BOUNDED sample of the same workload: units (z-planes of the same lattice / chunks
    of the same hit points) on one thread for ~budget_s (the reference is single-threaded), compared with
    the GPU's output for those units, then the same evaluator on the box's CPU share (<= 16 threads)."""
    kind, units, run = cpu_evaluator(workload, n, tex_points_host)
    done, t_used, max_err = 0, 0.0, 0.0
    while t_used < budget_s and done < units:
        t0 = time.perf_counter()
        got_cpu = run(done)
        t_used += time.perf_counter() - t0
        max_err = max(max_err, float(np.abs(gpu_unit(done) - got_cpu).max()))
        done += 1
    threads = max(1, min(16, os.cpu_count() or 1))
    counts = [0] * threads
    stop_at = time.perf_counter() + min(6.0, budget_s / 2)

    def worker(t):
        k = (t * units) // threads
        while time.perf_counter() < stop_at:
            run(k % units)
            k += 1
            counts[t] += 1
    t0 = time.perf_counter()
    pool = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in pool:
        th.start()
    for th in pool:
        th.join()
    dt = time.perf_counter() - t0
    return {"value": done * unit_samples / t_used / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": f"{done} {unit_name} of the same workload ({done * unit_samples} samples, {t_used:.1f} s, "
                      "1 thread: the reference has no threading)",
            "all_cores": {"value": sum(counts) * unit_samples / dt / 1e6, "unit": "Msamples/s", "cores": threads,
                          "sample": f"{sum(counts)} {unit_name} in {dt:.1f} s on {threads} threads"},
            "host_cpus": os.cpu_count(), "gpu_vs_cpu_max_abs_err": max_err}


def pmc_traffic(kernel):
    """HBM bytes per launch of the committed rocprofv3 --pmc passes for `kernel` (profiles/), with the
    file they come from; bench.py cannot read PMC counters in-run."""
    for name in ("pmc_traffic.json", "pmc_traffic_workloads.json"):
        p = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(p):
            continue
        try:
            doc = json.load(open(p))
        except Exception:  # noqa: BLE001
            continue
        entries = doc if isinstance(doc, list) else [doc]
        for e in entries:
            if e.get("kernel") and e["kernel"] in kernel:
                return e.get("bytes_per_launch"), e.get("source", f"profiles/{name}")
    return None, None


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.print_launch):
        sys.exit(self_launch(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")
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
    # N = 1: configs[1] (512^3).  N > 1: configs[4] -- ONE 2048^3 lattice cut into z-slabs of 2048/N planes
    n = args.n or (512 if world == 1 else 2048)
    planes = args.planes or (n if world == 1 else -(-n // world))
    nz_total = n if (world > 1 and not args.planes) else planes * world
    z0 = min(rank * planes, nz_total)
    z1 = min(z0 + planes, nz_total)
    my_planes = z1 - z0
    samples_per_rank = n * n * my_planes
    sharded_volume = world > 1 and nz_total == n
    out = torch.empty(max(1, n * n * planes), dtype=torch.float32, device="cuda")

    alg_bytes = 4 * samples_per_rank + 4 * TILE ** 3  # one fp32 store/sample + the tile once
    alg_flops = None
    dtype, bound = "f32", "hbm"
    tex_pts = None
    wl = args.workload
    slab_desc = (f"{n}^3" if (world == 1 and planes == n) else
                 f"{n}x{n}x{planes} z-slab per GPU of " + (f"ONE {n}^3 lattice" if sharded_volume else f"a {n}^2 x {nz_total} lattice"))
    if wl in ("wavelet3d", "wavelet3d_exact", "multiband5"):
        noise = wn.WaveletNoise(TILE, SEED)
        noise.generateNoiseTile3D()  # every rank regenerates the tile from the seed
        if wl == "multiband5":
            step = lambda: wn.multiband_volume(noise, n, n, n, z0, z1, -16.0, 0, 5, out=out)  # noqa: E731
            kernel, desc = "grid3d_sep_kernel<5", f"{slab_desc} WMultibandNoise, 5 bands (configs[2])"
        else:
            exact = wl == "wavelet3d_exact"
            step = lambda: wn.wavelet_volume(noise, n, n, n, z0, z1, OCTAVE, exact=exact, out=out)  # noqa: E731
            # the library's dispatch (csrc/wn_wavelet_strip.hip strip_try): rows of k*256 samples and
            # 0.18 <= planes per lattice step <= 1/3 go to the strip-march kernel, other lattices to the brick kernel
            lattice_step = 4.0 * 2.0 ** OCTAVE * 2.0 / n
            strip = n % 256 == 0 and 0.18 <= lattice_step < 1.0 / 3.0
            kernel = "grid3d_exact_lds_kernel" if exact else ("grid3d_strip_kernel" if strip else "grid3d_sep_kernel<1")
            which = "configs[1]" if (world == 1 and n == 512 and planes == 512) else ("configs[4]" if sharded_volume and n == 2048 else "configs[4] shard shape" if (n, planes) == (2048, 256) else "custom lattice")
            desc = f"{slab_desc} dense 3D WNoise grid, tile={TILE}, octave={OCTAVE} ({which})"
    elif wl in ("turb7", "perlin"):
        per = wn.perlin(SEED)
        dtype, bound = "f64", "valu_fp64"
        if wl == "turb7":
            step = lambda: wn.turb_volume(per, n, n, n, z0, z1, 7, out=out)  # noqa: E731
            desc = f"{slab_desc} perlin turb(depth=7) (configs[2])"
            alg_flops = samples_per_rank * 7 * (PERLIN_FLOPS + 2)
        else:
            step = lambda: wn.perlin_volume(per, n, n, n, z0, z1, OCTAVE, out=out)  # noqa: E731
            desc = f"{slab_desc} perlin noise grid, octave={OCTAVE}"
            alg_flops = samples_per_rank * PERLIN_FLOPS
        kernel = "perlin_grid_run_kernel"
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
        perlin_tex = wl == "texture_points_perlin"
        tex = wn.noise_texture(1.0, OCTAVE) if perlin_tex else wn.wavelet_texture(1.0, OCTAVE, True)
        out = torch.empty(m, dtype=torch.float32, device="cuda")
        step = lambda: tex.grey(pts, out=out)  # noqa: E731
        tex_pts = pts
        samples_per_rank = m
        alg_bytes = 16 * m  # 12 B xyz in + 4 B out
        kernel = "noise_texture_kernel" if perlin_tex else "plane_sorted_points_kernel"
        desc = f"{m} ray hit points, {'noise_texture (Perlin)' if perlin_tex else 'wavelet_texture'} octave {OCTAVE} (configs[3] stand-in)"
        dtype = "f64" if perlin_tex else "f32"
        if perlin_tex:
            alg_flops = m * (PERLIN_FLOPS + 2)

    def barrier():
        if dist is not None:
            dist.barrier()

    def timed(fn, reps):
        timer = wn.HipTimer()
        torch.cuda.synchronize()
        timer.start()  # HIP events on the stream the kernels are launched on
        for _ in range(reps):
            fn()
        timer.stop()
        torch.cuda.synchronize()
        return timer.elapsed_ms() / reps

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    timer = wn.HipTimer()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timer.start()
    for _ in range(args.steps):
        step()
    timer.stop()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    ev_ms = timer.elapsed_ms()
    total_samples = samples_per_rank
    if dist is not None:
        t = torch.tensor([dt, ev_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, ev_ms = float(t[0]), float(t[1])
        s = torch.tensor([samples_per_rank], dtype=torch.float64, device=t.device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples = int(s[0])

    # ---- the one exchange of the sharded path: collect the slabs on rank 0 (outside the timed steps) ----
    gather = None
    if dist is not None and not args.no_gather and wl not in ("texture_points", "texture_points_perlin"):
        slab = out[: n * n * my_planes].view(my_planes, n, n)
        full = None
        if backend != "nccl":
            slab = slab.cpu()
        if rank == 0:
            full = torch.empty((nz_total, n, n), dtype=torch.float32, device=slab.device)
        times = []
        for _ in range(2):  # the first call also builds RCCL's connections
            torch.cuda.synchronize(); barrier()
            g0 = time.perf_counter()
            wn.gather_volume(slab, nz_total, dst=0, out=full)
            torch.cuda.synchronize(); barrier()
            times.append(time.perf_counter() - g0)
        moved = 4.0 * (total_samples - samples_per_rank) if rank == 0 else 0.0
        gather = {"ms": times[-1] * 1e3, "first_call_ms": times[0] * 1e3, "GBps_into_root": moved / times[-1] / 1e9,
                  "bytes_into_root": moved, "pattern": "grouped send/recv: every peer's slab lands in its place of the volume on rank 0",
                  "backend": "RCCL" if backend == "nccl" else backend}
        if rank == 0 and wl in ("wavelet3d",):  # the gathered volume is the volume: spot-check a peer's planes
            zc = min(nz_total - 1, planes)  # first plane of rank 1's slab
            chk = wn.wavelet_volume(noise, n, n, n, zc, zc + 1, OCTAVE)
            gather["peer_plane_matches_local_recompute"] = bool(torch.equal(chk[0].to(full.device), full[zc]))
        del full

    if rank == 0:
        launch_s = ev_ms / 1e3 / args.steps
        hbm = {"achieved": alg_bytes / launch_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": alg_bytes / launch_s / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": alg_bytes}
        traffic, traffic_src = pmc_traffic(kernel)
        roofline = {"bound": bound, "kernel": kernel, "avg_launch_us": launch_s * 1e6}
        if bound == "hbm":
            roofline.update(hbm)
        else:  # fp64 VALU bound: algorithmic flops / vector fp64 peak, the HBM view beside it
            tf = alg_flops / launch_s / 1e12
            roofline.update({"achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TF,
                             "algorithmic_flops_per_launch": alg_flops,
                             "note": "peak counts an FMA as 2 flops; the reference's unfused add/mul stream can reach half of it",
                             "hbm": hbm})
        roofline["traffic"] = traffic
        roofline["traffic_from"] = traffic_src  # a committed rocprofv3 --pmc summary, not measured in this run
        if not args.no_measured_peak:
            # the store / copy ceiling of THIS box, same buffer, same run (torch's fill and copy kernels)
            nbytes = out.numel() * 4
            out.zero_()
            fill_ms = timed(lambda: out.zero_(), 20)  # hipMemsetAsync-class fill: the store-stream ceiling
            src = torch.empty_like(out)
            copy_ms = timed(lambda: src.copy_(out), 10)
            del src
            roofline["measured_peak"] = {"fill_GBps": nbytes / fill_ms / 1e6, "copy_GBps": 2 * nbytes / copy_ms / 1e6,
                                         "bytes": nbytes, "how": "torch zero_ / copy_ of the output buffer, HIP events, same run"}
            if bound == "hbm":
                roofline["frac_of_measured_fill"] = roofline["achieved"] / roofline["measured_peak"]["fill_GBps"]
            if wl == "wavelet3d" and world == 1 and args.two_stream_probe:
                # beside the contract's single-stream figure: the same launches alternating between two streams and two
                # output buffers, so that the ramp and tail of consecutive launches overlap (informational, never `value`)
                second = torch.empty_like(out)
                lanes = [(torch.cuda.Stream(), out), (torch.cuda.Stream(), second)]

                def piped(k):
                    for i in range(k):
                        st, buf = lanes[i & 1]
                        with torch.cuda.stream(st):
                            wn.wavelet_volume(noise, n, n, n, z0, z1, OCTAVE, out=buf)

                piped(4)
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                piped(args.steps)
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
                roofline["two_stream_pipeline"] = {"ms_per_step": pdt * 1e3 / args.steps, "Msamples_per_s": samples_per_rank * args.steps / pdt / 1e6,
                                                   "how": "same launches alternating between 2 HIP streams / 2 output buffers; wall clock"}
                del second, lanes
            for _ in range(2):
                step()  # the buffer holds the workload's output again for the CPU comparison below
            torch.cuda.synchronize()
        line = {
            "metric": "Mnoise-samples/sec (3D wavelet, octave=4)", "value": total_samples * args.steps / dt / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong" if sharded_volume else "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": desc, "lattice_per_gpu": [n, n, planes], "lattice_total": [n, n, nz_total], "tile": TILE,
                       "seed": SEED, "octave": OCTAVE, "sharding": "z-slabs, no data-path collective inside a step",
                       "device": wn.device_info()["name"]},
            "per_gpu": {"Msamples_per_s": samples_per_rank / launch_s / 1e6, "avg_launch_us": launch_s * 1e6},
            "roofline": roofline,
        }
        if gather:
            line["gather"] = gather
            line["step_plus_gather_ms"] = dt * 1e3 / args.steps + gather["ms"]
        if world == 1 and not args.no_cpu_baseline:
            if tex_pts is not None:
                host_pts = tex_pts[: 1 << 22].cpu().numpy()
                chunk = 1 << 17
                line["cpu_baseline"] = cpu_baseline(wl, n, args.cpu_seconds, lambda k: out[k * chunk:(k + 1) * chunk].cpu().numpy(),
                                                    chunk, "chunks of 131072 hit points", host_pts)
            else:
                vol = out[: n * n * my_planes].view(my_planes, n, n)
                line["cpu_baseline"] = cpu_baseline(wl, n, args.cpu_seconds, lambda z: vol[z].cpu().numpy().ravel(),
                                                    n * n, f"z-planes of the same {n}^3 lattice")
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
