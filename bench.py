#!/usr/bin/env python3
"""bench.py -- Mnoise-samples/s of the hot path on N MI355X, with the roofline of the dominant
kernel and the reference's CPU path timed beside it.

  python bench.py                       N=1: BASELINE configs[1], 512^3 dense 3-D wavelet grid.  The line also carries
                                        `roofline.sustained` (>= 1 s of back-to-back headline launches) and `per_config`
                                        (short legs of every other BASELINE workload: multiband5, turb7, perlin, both
                                        texture stand-ins, the bit-exact grid), each with its kernel's roofline fraction
                                        and the reference's CPU path on a bounded sample of the same workload
  python bench.py --gpus 8              BASELINE configs[4]: ONE 2048^3 lattice in z-slabs of 2048/N planes.  Rank 0 also
                                        times the SAME lattice alone in the same run (`strong_scaling`), the 512^3 lattice
                                        of the north star sharded the same way (`strong_scaling_512`), and the one gather
                                        of the slabs on rank 0 (`gather`)
  python bench.py --workload turb7      one of the other workloads as the timed workload, same evidence

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
PER_CONFIG = ["multiband5", "turb7", "perlin", "texture_points", "texture_points_perlin", "wavelet3d_exact"]
TEX_POINTS = 80_000_000    # configs[3] stand-in: hit points of one 1920x1080x64 render, 85 % quad / 15 % sphere
TEX_CHUNK = 1 << 17


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
    ap.add_argument("--no-per-config", action="store_true",
                    help="N = 1 default workload: skip the short legs of the other BASELINE workloads (`per_config`)")
    ap.add_argument("--per-config-cpu-seconds", type=float, default=2.5)
    ap.add_argument("--no-sustained", action="store_true", help="skip `roofline.sustained` (>= 1 s of back-to-back launches)")
    ap.add_argument("--sustained-seconds", type=float, default=1.1)
    ap.add_argument("--no-strong-scaling", action="store_true",
                    help="N > 1: skip rank 0's single-GPU run of the same lattice and the 512^3 variant")
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
    """Returns (kind, unit_count, run) where run(k) evaluates unit k (one z-plane of the
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
    chunk = TEX_CHUNK
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


def cpu_baseline(workload, n, budget_s, gpu_unit, unit_samples, unit_name, tex_points_host=None, unit_order=None):
    """The CPU path on a BOUNDED sample of the same workload: units (z-planes of the same lattice / chunks
    of the same hit points) on one thread for ~budget_s (the reference is single-threaded), compared with
    the GPU's output for those units, then the same evaluator on the box's CPU share (<= 16 threads)."""
    kind, units, run = cpu_evaluator(workload, n, tex_points_host)
    order = list(unit_order(units)) if unit_order else list(range(units))
    done, t_used, max_err = 0, 0.0, 0.0
    while t_used < budget_s and done < len(order):
        t0 = time.perf_counter()
        got_cpu = run(order[done])
        t_used += time.perf_counter() - t0
        max_err = max(max_err, float(np.abs(gpu_unit(order[done]) - got_cpu).max()))
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


def spread_planes(units):
    """z-planes in an order that samples the whole lattice early: 0, n/2, n/4, 3n/4, ... (a short CPU budget then
    compares planes from all over the volume, not the first few)."""
    seen, out, step = set(), [], units
    while step >= 1 and len(out) < units:
        for z in range(0, units, max(1, step)):
            if z not in seen:
                seen.add(z)
                out.append(z)
        step //= 2
    return out


def traffic_key(wl, n, planes):
    """Key of a committed PMC pass: the workload, plus the lattice when it is not that workload's default 512^3
    (a kernel's HBM bytes belong to one lattice: the same kernel at 1024^3 writes 8x the bytes)."""
    if wl.startswith("texture_points") or (n == 512 and planes == 512):
        return wl
    return f"{wl}_{n}" if planes == n else f"{wl}_{n}x{n}x{planes}"


def pmc_traffic(key):
    """HBM bytes per launch of the committed rocprofv3 --pmc passes for the (workload, lattice) `key`
    (profiles/pmc_traffic_workloads.json), with the file they come from; bench.py cannot read PMC counters in-run."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic_workloads.json")
    try:
        entries = json.load(open(p))
    except Exception:  # noqa: BLE001
        return None, None
    for e in entries if isinstance(entries, list) else [entries]:
        if e.get("workload") == key:
            return e.get("bytes_per_launch"), e.get("source", "profiles/pmc_traffic_workloads.json")
    return None, None


# ---- workloads ---------------------------------------------------------------------------------------
class Workload:
    """One BASELINE workload on this rank's share: `step()` is one pass of the hot path (one kernel launch)."""

    def __init__(self, wn, wl, n, z0, z1, nz_total, world=1, rank=0, out=None, tex_points=TEX_POINTS):
        self.wl, self.n, self.z0, self.z1 = wl, n, z0, z1
        planes = z1 - z0
        self.planes = planes
        self.samples = n * n * planes
        self.alg_bytes = 4 * self.samples + 4 * TILE ** 3  # one fp32 store/sample + the tile once
        self.alg_flops = None
        self.dtype, self.bound = "f32", "hbm"
        self.tex_pts = None
        sharded = world > 1 and nz_total == n
        slab = (f"{n}^3" if (world == 1 and planes == n) else
                f"{n}x{n}x{planes} z-slab per GPU of " + (f"ONE {n}^3 lattice" if sharded else f"a {n}^2 x {nz_total} lattice"))
        if wl in ("wavelet3d", "wavelet3d_exact", "multiband5"):
            self.out = out if out is not None else torch.empty(max(1, self.samples), dtype=torch.float32, device="cuda")
            noise = self.noise = wn.WaveletNoise(TILE, SEED)
            noise.generateNoiseTile3D()  # every rank regenerates the tile from the seed
            if wl == "multiband5":
                self.step = lambda: wn.multiband_volume(noise, n, n, n, z0, z1, -16.0, 0, 5, out=self.out)
                self.kernel, self.desc = "grid3d_mbp_kernel<5", f"{slab} WMultibandNoise, 5 bands (configs[2])"
            else:
                exact = wl == "wavelet3d_exact"
                self.step = wn.wavelet_volume_launcher(noise, n, n, n, z0, z1, OCTAVE, self.out, exact=exact)
                # the library's dispatch (csrc/wn_wavelet_strip.hip strip_try): rows of k*256 samples and
                # 0.18 <= planes per lattice step <= 1/3 go to the strip-march kernel, other lattices to the brick kernel
                # and before both (csrc/wn_wavelet_grid.hip wn_eval3d_grid -> wn_wavelet_multiband.hip multiband_try): lattices
                # whose rows fill 512-wide bricks and whose step is below 2/7 of a cell go to the single-band plane pipeline
                lattice_step = 4.0 * 2.0 ** OCTAVE * 2.0 / n
                strip = n % 256 == 0 and 0.18 <= lattice_step < 1.0 / 3.0
                pipeline = n > 256 and n % 4 == 0 and lattice_step < 2.0 / 7.0 and -(-n // 512) * 512 * 10 <= n * 11
                self.kernel = ("grid3d_exact_lds_kernel" if exact else "grid3d_mbp_kernel<1" if pipeline else
                               "grid3d_strip_kernel" if strip else "grid3d_sep_kernel<1")
                which = ("configs[1]" if (world == 1 and n == 512 and planes == 512) else
                         "configs[4]" if sharded and n == 2048 else
                         "configs[4] shard shape" if (n, planes) == (2048, 256) else "custom lattice")
                self.desc = f"{slab} dense 3D WNoise grid, tile={TILE}, octave={OCTAVE} ({which})" + (", bit-exact kernel (WN_GRID_EXACT)" if exact else "")
        elif wl in ("turb7", "perlin"):
            self.out = out if out is not None else torch.empty(max(1, self.samples), dtype=torch.float32, device="cuda")
            per = self.per = wn.perlin(SEED)
            self.dtype, self.bound = "f64", "valu_fp64"
            if wl == "turb7":
                self.step = lambda: wn.turb_volume(per, n, n, n, z0, z1, 7, out=self.out)
                self.desc = f"{slab} perlin turb(depth=7) (configs[2])"
                self.alg_flops = self.samples * 7 * (PERLIN_FLOPS + 2)
            else:
                self.step = lambda: wn.perlin_volume(per, n, n, n, z0, z1, OCTAVE, out=self.out)
                self.desc = f"{slab} perlin noise grid, octave={OCTAVE}"
                self.alg_flops = self.samples * PERLIN_FLOPS
            self.kernel = "perlin_grid_run_kernel"
            self.alg_bytes = 4 * self.samples + 512
        else:  # texture_points: configs[3] stand-in: 85% quad / 15% sphere hits
            m = tex_points // max(1, world)
            g = torch.Generator(device="cuda").manual_seed(1 + rank)
            pts = torch.empty((m, 3), dtype=torch.float32, device="cuda")
            pts[:, 0].uniform_(-10, 10, generator=g)
            pts[:, 1] = -0.5
            pts[:, 2].uniform_(-10, 10, generator=g)
            k = int(0.15 * m)
            d = torch.randn((k, 3), device="cuda", generator=g)
            pts[:k] = torch.tensor([1.0, 0.0, -1.75], device="cuda") + 0.5 * d / d.norm(dim=1, keepdim=True)
            del d
            perlin_tex = wl == "texture_points_perlin"
            tex = self.tex = wn.noise_texture(1.0, OCTAVE) if perlin_tex else wn.wavelet_texture(1.0, OCTAVE, True)
            self.out = torch.empty(m, dtype=torch.float32, device="cuda")
            self.step = lambda: tex.grey(pts, out=self.out)
            self.tex_pts = pts
            self.samples = m
            self.alg_bytes = 16 * m  # 12 B xyz in + 4 B out
            self.kernel = "noise_texture_kernel" if perlin_tex else "row_slab_points_kernel"  # (after plane_sorted_points_kernel<defer>: two launches a call)
            self.desc = f"{m} ray hit points, {'noise_texture (Perlin)' if perlin_tex else 'wavelet_texture'} octave {OCTAVE} (configs[3] stand-in)"
            self.dtype = "f64" if perlin_tex else "f32"
            if perlin_tex:
                self.alg_flops = m * (PERLIN_FLOPS + 2)

    def roofline(self, launch_s):
        """algorithmic bytes (or flops) per launch / the average launch time, against the nominal peak."""
        hbm = {"achieved": self.alg_bytes / launch_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
               "frac": self.alg_bytes / launch_s / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": self.alg_bytes}
        r = {"bound": self.bound, "kernel": self.kernel, "avg_launch_us": launch_s * 1e6}
        if self.bound == "hbm":
            r.update(hbm)
        else:  # fp64 VALU bound: algorithmic flops / vector fp64 peak, the HBM view beside it
            tf = self.alg_flops / launch_s / 1e12
            r.update({"achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TF,
                      "algorithmic_flops_per_launch": self.alg_flops,
                      "note": "peak counts an FMA as 2 flops; the reference's unfused add/mul stream can reach half of it",
                      "hbm": hbm})
        r["traffic"], r["traffic_from"] = pmc_traffic(traffic_key(self.wl, self.n, self.planes))
        return r  # traffic: a committed rocprofv3 --pmc summary of this (workload, lattice), not measured in this run

    def cpu_baseline(self, budget_s):
        """The reference's CPU path on units of this workload, compared with what the GPU left in `self.out`."""
        if self.tex_pts is not None:
            host_pts = self.tex_pts[: 1 << 22].cpu().numpy()
            return cpu_baseline(self.wl, self.n, budget_s, lambda k: self.out[k * TEX_CHUNK:(k + 1) * TEX_CHUNK].cpu().numpy(),
                                TEX_CHUNK, f"chunks of {TEX_CHUNK} hit points", host_pts)
        vol = self.out[: self.samples].view(self.planes, self.n, self.n)
        return cpu_baseline(self.wl, self.n, budget_s, lambda z: vol[z].cpu().numpy().ravel(), self.n * self.n,
                            f"z-planes of the same {self.n}^3 lattice", unit_order=spread_planes)


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
    sharded_volume = world > 1 and nz_total == n
    wl = args.workload
    W = Workload(wn, wl, n, z0, z1, nz_total, world, rank)
    step, out = W.step, W.out
    samples_per_rank = W.samples
    red_device = "cuda" if backend == "nccl" else "cpu"

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

    def max_over_ranks(*vals):
        if dist is None:
            return vals
        t = torch.tensor(vals, dtype=torch.float64, device=red_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t)

    # The store / copy ceiling of THIS box, same buffer, same run (torch's fill and copy kernels), measured BEFORE the warm-up:
    # ~8 ms of memory-bound launches after the idle gap of the set-up, so that the W warm-up and K timed steps that follow
    # are not the chip's first work after idling (the first ~20 launches after an idle gap run at a lower clock: 110 us per
    # launch against 101.8 us from the 100th on, `roofline.sustained`).
    measured_peak = None
    if rank == 0 and not args.no_measured_peak:
        nbytes = out.numel() * 4
        out.zero_()
        fill_ms = timed(lambda: out.zero_(), 50)  # hipMemsetAsync-class fill: the store-stream ceiling
        src = torch.empty_like(out)
        copy_ms = timed(lambda: src.copy_(out), 20)
        del src
        measured_peak = {"fill_GBps": nbytes / fill_ms / 1e6, "copy_GBps": 2 * nbytes / copy_ms / 1e6, "bytes": nbytes,
                         "how": "torch zero_ / copy_ of the output buffer, HIP events, same run, before the warm-up steps"}
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
        dt, ev_ms = max_over_ranks(dt, ev_ms)
        s = torch.tensor([samples_per_rank], dtype=torch.float64, device=red_device)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        total_samples = int(s[0])

    # ---- the one exchange of the sharded path: collect the slabs on rank 0 (outside the timed steps) ----
    gather = None
    if dist is not None and not args.no_gather and W.tex_pts is None:
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
            chk = wn.wavelet_volume(W.noise, n, n, n, zc, zc + 1, OCTAVE)
            gather["peer_plane_matches_local_recompute"] = bool(torch.equal(chk[0].to(full.device), full[zc]))
        del full

    # ---- N > 1: the same lattice on ONE GPU in the same run, and the north star's 512^3 lattice sharded the same way ----
    # value(N) of this line is for ONE n^3 lattice cut into N slabs; the N = 1 line of the default command is for 512^3
    # (configs[1], strip kernel) -- two workloads.  So every N > 1 line carries its own single-GPU baseline: rank 0 computes
    # the whole lattice alone while the others wait, and both lattices are reported as {n1_ms, nN_ms, efficiency}.
    scaling = {}
    if dist is not None and sharded_volume and wl == "wavelet3d" and not args.no_strong_scaling:
        def scaling_leg(m, reps):
            zs = [min(r * -(-m // world), m) for r in range(world + 1)]
            a, b = zs[rank], zs[rank + 1]
            buf = out if m * m * (b - a) <= out.numel() else torch.empty(m * m * (b - a), dtype=torch.float32, device="cuda")
            fn = wn.wavelet_volume_launcher(W.noise, m, m, m, a, b, OCTAVE, buf) if b > a else (lambda: None)
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); barrier()
            (nN_ms,) = max_over_ranks(timed(fn, reps))  # HIP events on each rank's stream, the slowest rank counts
            del buf
            n1_ms = 0.0
            if rank == 0:  # the others wait at the barrier below; nothing else runs on this GPU meanwhile
                whole = torch.empty(m * m * m, dtype=torch.float32, device="cuda")
                fn1 = wn.wavelet_volume_launcher(W.noise, m, m, m, 0, m, OCTAVE, whole)
                for _ in range(2):
                    fn1()
                n1_ms = timed(fn1, reps)
                del whole
            barrier()
            return {"lattice": [m, m, m], "n1_ms": n1_ms, "nN_ms": nN_ms, "n": world, "reps": reps,
                    "efficiency": (n1_ms / nN_ms / world) if nN_ms > 0 else None,
                    "how": "HIP events around `reps` back-to-back launches; nN = the slowest rank of the N-slab run, n1 = rank 0 alone on the whole lattice, same run, same build"}
        scaling["strong_scaling"] = scaling_leg(n, max(3, min(args.steps, 10)))
        if n != 512:
            scaling["strong_scaling_512"] = scaling_leg(512, max(20, args.steps))

    if rank == 0:
        launch_s = ev_ms / 1e3 / args.steps
        roofline = W.roofline(launch_s)
        bound = W.bound
        if measured_peak is not None:
            roofline["measured_peak"] = measured_peak
            if bound == "hbm":
                roofline["frac_of_measured_fill"] = roofline["achieved"] / measured_peak["fill_GBps"]
            if wl == "wavelet3d" and world == 1 and args.two_stream_probe:
                # beside the contract's single-stream figure: the same launches alternating between two streams and two
                # output buffers, so that the ramp and tail of consecutive launches overlap (informational, never `value`)
                second = torch.empty_like(out)
                lanes = [(torch.cuda.Stream(), out), (torch.cuda.Stream(), second)]

                def piped(k):
                    for i in range(k):
                        st, buf = lanes[i & 1]
                        with torch.cuda.stream(st):
                            wn.wavelet_volume(W.noise, n, n, n, z0, z1, OCTAVE, out=buf)

                piped(4)
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                piped(args.steps)
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
                roofline["two_stream_pipeline"] = {"ms_per_step": pdt * 1e3 / args.steps, "Msamples_per_s": samples_per_rank * args.steps / pdt / 1e6,
                                                   "how": "same launches alternating between 2 HIP streams / 2 output buffers; wall clock"}
                del second, lanes
        if world == 1 and not args.no_sustained and wl in ("wavelet3d", "multiband5", "wavelet3d_exact", "perlin", "turb7"):
            # >= 1 s of back-to-back launches of the same step on the same stream, one HIP-event pair per segment of 100
            # launches: the mean, and the min / max of the per-segment means (does the figure hold beyond a 5 ms burst?)
            seg = 100 if launch_s < 5e-4 else 10
            nseg = max(3, min(400, int(np.ceil(args.sustained_seconds / (launch_s * seg)))))
            timers = [wn.HipTimer() for _ in range(nseg)]
            torch.cuda.synchronize()
            s0 = time.perf_counter()
            for t in timers:
                t.start()
                for _ in range(seg):
                    step()
                t.stop()
            torch.cuda.synchronize()
            wall = time.perf_counter() - s0
            means = [t.elapsed_ms() * 1e3 / seg for t in timers]
            mean_us = float(np.mean(means))
            roofline["sustained"] = {"launches": seg * nseg, "seconds": wall, "mean_us": mean_us, "min_us": min(means), "max_us": max(means),
                                     "first_segment_us": means[0], "last_segment_us": means[-1], "segment_launches": seg,
                                     "wall_us_per_launch": wall * 1e6 / (seg * nseg),
                                     "frac": (W.alg_bytes / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if bound == "hbm" else
                                             (W.alg_flops / (mean_us * 1e-6) / 1e12 / FP64_VALU_PEAK_TF),
                                     "how": f"{nseg} segments of {seg} back-to-back launches, one HIP-event pair per segment, same stream / buffer"}
            del timers
        if not args.no_measured_peak:
            for _ in range(2):
                step()  # the buffer holds the workload's output again for the CPU comparison below
            torch.cuda.synchronize()
        line = {
            "metric": "Mnoise-samples/sec (3D wavelet, octave=4)", "value": total_samples * args.steps / dt / 1e6,
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "strong" if sharded_volume else "weak",
            "vs_baseline": None, "dtype": W.dtype, "data": "synthetic",
            "config": {"workload": W.desc, "lattice_per_gpu": [n, n, planes], "lattice_total": [n, n, nz_total], "tile": TILE,
                       "seed": SEED, "octave": OCTAVE, "sharding": "z-slabs, no data-path collective inside a step",
                       "device": wn.device_info()["name"]},
            "per_gpu": {"Msamples_per_s": samples_per_rank / launch_s / 1e6, "avg_launch_us": launch_s * 1e6},
            "roofline": roofline,
        }
        if gather:
            line["gather"] = gather
            line["step_plus_gather_ms"] = dt * 1e3 / args.steps + gather["ms"]
        line.update(scaling)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = W.cpu_baseline(args.cpu_seconds)
        if world == 1 and wl == "wavelet3d" and not args.no_per_config:
            # the other BASELINE workloads, each as a short leg of the same evidence (5 warm-up + 20 launches, HIP events)
            del W, out, step
            torch.cuda.empty_cache()
            per = {}
            for name in PER_CONFIG:
                P = Workload(wn, name, 512, 0, 512, 512)
                # Untimed warm-up: at least 5 launches and at least 0.1 s of them.  The first ~20 launches after an idle
                # gap run at the clock the chip idles at (the setup and the CPU leg of the previous workload are such a gap):
                # the same kernel measures 179 us over launches 6-25 and 145 us from the 100th on (multiband5, round 3).
                warm, w0 = 0, time.perf_counter()
                while warm < 5 or time.perf_counter() - w0 < 0.1:
                    P.step()
                    warm += 1
                    if warm % 8 == 0:
                        torch.cuda.synchronize()
                ms = timed(P.step, 20)
                r = P.roofline(ms / 1e3)
                entry = {"workload": P.desc, "dtype": P.dtype, "launches": 20, "warmup": warm,
                         "warmup_rule": ">= 5 launches and >= 0.1 s of launches (clock ramp after the idle gap of the set-up)",
                         "Msamples_per_s": P.samples / (ms / 1e3) / 1e6}
                entry.update(r)
                reps = max(20, min(2000, int(0.3 / (ms / 1e3))))  # ~0.3 s more of the same launches, back to back
                entry["sustained_us"] = timed(P.step, reps) * 1e3
                entry["sustained_launches"] = reps
                if not args.no_cpu_baseline:
                    entry["cpu_baseline"] = P.cpu_baseline(args.per_config_cpu_seconds)
                    entry["gpu_vs_cpu_max_abs_err"] = entry["cpu_baseline"]["gpu_vs_cpu_max_abs_err"]
                per[name] = entry
                del P
                torch.cuda.empty_cache()
            line["per_config"] = per
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
