/* wnoise_shard.h -- C ABI of libwnoise_shard.so: the ONE exchange of the sharded dense-grid path.
 *
 * The reference has no multi-GPU code; its experient/main.cpp:131-168 writes whole grids from one
 * process.  The sharded form of that role: every rank computes a contiguous z-slab of the lattice (wn_shard_bounds;
 * the slab is one contiguous block of the final x-fastest volume, experient/main.cpp:28 extended to z) with the
 * kernels of libwnoise_hip.so and no collective; the slabs are then collected on one rank with ONE grouped
 * ncclSend / ncclRecv exchange over RCCL (xGMI: every peer pushes over its own link straight into the slab's place in
 * the root's volume, nothing is re-packed).  Separate from libwnoise_hip.so so that only programs that shard load
 * librccl (570 MB).
 *
 * Conventions as in wnoise.h: int status (WN_OK = 0), message through wn_shard_last_error(), plain pointers,
 * `void *stream` = hipStream_t (NULL: the default stream), buffers caller-owned.
 */
#ifndef WNOISE_SHARD_H
#define WNOISE_SHARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WN_SHARD_API __attribute__((visibility("default")))
#define WN_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct wn_comm wn_comm;

WN_SHARD_API const char *wn_shard_last_error(void);

/* Planes [*z0, *z1) of an nz-plane lattice owned by `rank` of `world`: as even as possible, earlier ranks take the
 * remainder.  Pure arithmetic (no device needed). */
WN_SHARD_API int wn_shard_bounds(int nz, int world, int rank, int *z0, int *z1);

/* Rank 0 draws the communicator id (ncclGetUniqueId) and hands its WN_COMM_ID_BYTES bytes to the other ranks by any
 * means (a file, a pipe, MPI, torch.distributed); every rank then creates its communicator on its CURRENT device
 * (ncclCommInitRank: collective over the `world` ranks). */
WN_SHARD_API int wn_comm_unique_id(void *id_bytes);
WN_SHARD_API int wn_comm_create(wn_comm **out, int world, int rank, const void *id_bytes);
WN_SHARD_API int wn_comm_rank(const wn_comm *comm, int *rank, int *world);
WN_SHARD_API void wn_comm_destroy(wn_comm *comm);

/* Collect the z-slabs of an [nz][ny][nx] float32 volume on rank `dst`: rank r passes its planes wn_shard_bounds(nz,
 * world, r) as `slab_dev`; on `dst`, `out_dev` (nz*ny*nx floats) receives every peer's slab in its place (its own by a
 * device-to-device copy); elsewhere `out_dev` is ignored.  One ncclGroupStart ... ncclGroupEnd with pieces of whole
 * planes of at most `piece_bytes` (0: 1 GiB).  Enqueued on `stream`; returns without synchronising. */
WN_SHARD_API int wn_gather_volume(wn_comm *comm, const float *slab_dev, int nz, int ny, int nx, int dst,
                                  float *out_dev, size_t piece_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* WNOISE_SHARD_H */
