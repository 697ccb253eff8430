"""Worker for tests/test_shard_gloo.py: one rank of a gloo process group on the CPU.

Each rank produces its z-slab of a small config-2-style volume with the ORACLE (this is a test:
the GPU kernels cannot run here), then the product's gather_volume() collects the slabs on rank 0,
which checks the result against the oracle's full volume."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    nz, ny, nx, den, octave, dst = (int(v) for v in sys.argv[1:7])
    piece_bytes = int(sys.argv[7]) if len(sys.argv) > 7 else 1 << 30  # small: a slab travels in several messages
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
    import oracle

    tile = oracle.tile3d(16, 12345)
    z0, z1 = wn.slab_bounds(nz, world, rank)
    slab = torch.from_numpy(oracle.grid_wavelet3d_volume(tile, den, nx, ny, z0, z1, octave))
    full = wn.gather_volume(slab, nz, dst=dst, piece_bytes=piece_bytes)
    if rank == dst:
        want = oracle.grid_wavelet3d_volume(tile, den, nx, ny, 0, nz, octave)
        assert full.shape == (nz, ny, nx)
        assert (full.numpy().view(np.uint32) == want.view(np.uint32)).all()
        print("GATHER_OK", world, nz)
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
