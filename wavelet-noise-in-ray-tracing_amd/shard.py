"""Multi-GPU sharding of the dense-grid path: contiguous z-slabs, one process per GPU.

Samples are independent and the only shared state is the read-only tile (regenerated on each
rank from the seed), so the evaluation itself needs no collective.  The single exchange is the
optional collection of the slabs on one rank: because every slab is a contiguous block of the
final x-fastest volume, the root receives each peer's slab straight into its place with one
grouped send/recv (RCCL over xGMI on the GPU box: every peer pushes over its own link; gloo in
the CPU tests).  Nothing is re-packed.
"""
import ctypes as C

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


class NativeComm:
    """The C ABI's communicator (include/wnoise_shard.h, libwnoise_shard.so: grouped ncclSend / ncclRecv over RCCL), for
    callers that do not go through torch.distributed's process group -- and to exercise from Python the entry points
    tools/gridgen --gpus N uses.  The id travels by torch.distributed when a process group exists (any backend),
    otherwise `id_bytes` must be handed in (rank 0: NativeComm.unique_id())."""

    def __init__(self, world=None, rank=None, id_bytes=None):
        from . import _capi_shard as cs
        self._cs = cs
        lib = cs.load()
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
        if id_bytes is None:
            buf = (C.c_ubyte * cs.WN_COMM_ID_BYTES)()
            if rank == 0:
                cs.check(lib.wn_comm_unique_id(buf))
            if world > 1:
                t = torch.tensor(list(buf), dtype=torch.uint8)
                if dist.get_backend() == "nccl":
                    t = t.cuda()
                dist.broadcast(t, src=0)
                buf = (C.c_ubyte * cs.WN_COMM_ID_BYTES)(*t.cpu().tolist())
            id_bytes = bytes(buf)
        self.world, self.rank = world, rank
        self._h = C.c_void_p()
        idbuf = (C.c_ubyte * cs.WN_COMM_ID_BYTES).from_buffer_copy(id_bytes)
        cs.check(lib.wn_comm_create(C.byref(self._h), world, rank, idbuf))

    @staticmethod
    def unique_id():
        from . import _capi_shard as cs
        buf = (C.c_ubyte * cs.WN_COMM_ID_BYTES)()
        cs.check(cs.load().wn_comm_unique_id(buf))
        return bytes(buf)

    def gather_volume(self, slab, nz, dst=0, out=None, piece_bytes=0):
        """wn_gather_volume: `slab` [z1-z0, ny, nx] float32 on the current device -> the [nz, ny, nx] volume on `dst`."""
        ny, nx = slab.shape[-2], slab.shape[-1]
        slab = slab.contiguous()
        full = None
        if self.rank == dst:
            full = out if out is not None else torch.empty((nz, ny, nx), dtype=torch.float32, device=slab.device)
        self._cs.check(self._cs.load().wn_gather_volume(
            self._h, C.c_void_p(slab.data_ptr()), nz, ny, nx, dst, C.c_void_p(full.data_ptr()) if full is not None else None,
            piece_bytes, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return full

    def close(self):
        if getattr(self, "_h", None):
            self._cs.load().wn_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
