"""Multi-GPU sharding of the dense-grid path: contiguous z-slabs, one process per GPU.

Samples are independent and the only shared state is the read-only tile (regenerated on each
rank from the seed), so the evaluation itself needs no collective.  The single exchange is the
optional collection of the slabs on one rank: because every slab is a contiguous block of the
final x-fastest volume, the root receives each peer's slab straight into its place with one
grouped send/recv (RCCL over xGMI on the GPU box: every peer pushes over its own link; gloo in
the CPU tests).  Nothing is re-packed.
"""
import torch
import torch.distributed as dist


def slab_bounds(nz, world_size, rank):
    """Planes [z0, z1) owned by `rank`: as even as possible, earlier ranks take the remainder."""
    base, rem = divmod(int(nz), int(world_size))
    z0 = rank * base + min(rank, rem)
    return z0, z0 + base + (1 if rank < rem else 0)


def gather_volume(slab, nz, dst=0, group=None, out=None, piece_bytes=1 << 30):
    """Collect z-slabs (each `slab` is [z1-z0, ny, nx], planes per slab_bounds) on rank `dst`.

    Returns the full [nz, ny, nx] volume on `dst`, None elsewhere.  Without an initialised
    process group (single process) the slab is the volume.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return slab
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return slab
    ny, nx = slab.shape[-2], slab.shape[-1]
    slab = slab.contiguous()
    # one grouped exchange; a slab travels in pieces of whole planes of at most `piece_bytes` (1 GiB) (a 2048^3 / 2 slab is 16 GiB:
    # no single message near the 32-bit byte counts some transports still carry)
    step = max(1, int(piece_bytes) // max(1, ny * nx * slab.element_size()))
    ops = []
    full = None
    if rank == dst:
        full = out if out is not None else torch.empty((nz, ny, nx), dtype=slab.dtype, device=slab.device)
        for r in range(world):
            z0, z1 = slab_bounds(nz, world, r)
            if z1 == z0:
                continue
            if r == dst:
                full[z0:z1].copy_(slab)
            else:
                src = dist.get_global_rank(group, r) if group is not None else r
                for z in range(z0, z1, step):
                    ops.append(dist.P2POp(dist.irecv, full[z:min(z + step, z1)], src, group))
    else:
        if slab.numel():
            peer = dist.get_global_rank(group, dst) if group is not None else dst
            for z in range(0, slab.shape[0], step):
                ops.append(dist.P2POp(dist.isend, slab[z:z + step], peer, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return full
