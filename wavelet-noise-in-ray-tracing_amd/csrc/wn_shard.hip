// wn_shard.hip -- libwnoise_shard.so (include/wnoise_shard.h): z-slab bounds and the one gather of the sharded
// dense-grid path over RCCL.  gfx950 / ROCm only; RCCL's API is NCCL's.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "wnoise.h"
#include "wnoise_shard.h"

struct wn_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define SHARD_NCCL(call)                                                                       \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) return fail(WN_ERR_HIP, "%s: %s", #call, ncclGetErrorString(r_)); \
    } while (0)
#define SHARD_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) return fail(WN_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

void bounds(int nz, int world, int rank, int *z0, int *z1)
{
    const int base = nz / world, rem = nz % world;
    *z0 = rank * base + (rank < rem ? rank : rem);
    *z1 = *z0 + base + (rank < rem ? 1 : 0);
}

} // namespace

extern "C" {

const char *wn_shard_last_error(void) { return g_err; }

int wn_shard_bounds(int nz, int world, int rank, int *z0, int *z1)
{
    if (!z0 || !z1) return fail(WN_ERR_INVALID, "wn_shard_bounds: NULL output");
    if (nz < 0 || world < 1 || rank < 0 || rank >= world)
        return fail(WN_ERR_INVALID, "wn_shard_bounds: nz=%d world=%d rank=%d", nz, world, rank);
    bounds(nz, world, rank, z0, z1);
    return WN_OK;
}

int wn_comm_unique_id(void *id_bytes)
{
    static_assert(sizeof(ncclUniqueId) == WN_COMM_ID_BYTES, "WN_COMM_ID_BYTES is sizeof(ncclUniqueId)");
    if (!id_bytes) return fail(WN_ERR_INVALID, "wn_comm_unique_id: NULL");
    ncclUniqueId id;
    SHARD_NCCL(ncclGetUniqueId(&id));
    std::memcpy(id_bytes, &id, sizeof(id));
    return WN_OK;
}

int wn_comm_create(wn_comm **out, int world, int rank, const void *id_bytes)
{
    if (!out || !id_bytes) return fail(WN_ERR_INVALID, "wn_comm_create: NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(WN_ERR_INVALID, "wn_comm_create: world=%d rank=%d", world, rank);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(WN_ERR_NO_DEVICE, "no HIP device: the sharded path has no CPU implementation");
    }
    int device = 0;
    SHARD_HIP(hipGetDevice(&device));
    wn_comm *c = new wn_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(WN_ERR_HIP, "ncclCommInitRank(world=%d, rank=%d): %s", world, rank, ncclGetErrorString(r));
    }
    *out = c;
    return WN_OK;
}

int wn_comm_rank(const wn_comm *comm, int *rank, int *world)
{
    if (!comm) return fail(WN_ERR_INVALID, "wn_comm_rank: NULL communicator");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return WN_OK;
}

void wn_comm_destroy(wn_comm *comm)
{
    if (!comm) return;
    if (comm->comm) (void)ncclCommDestroy(comm->comm);
    delete comm;
}

int wn_gather_volume(wn_comm *comm, const float *slab_dev, int nz, int ny, int nx, int dst, float *out_dev,
                     size_t piece_bytes, void *stream)
{
    if (!comm) return fail(WN_ERR_INVALID, "wn_gather_volume: NULL communicator");
    if (nz < 0 || ny < 0 || nx < 0 || dst < 0 || dst >= comm->world)
        return fail(WN_ERR_INVALID, "wn_gather_volume: nz=%d ny=%d nx=%d dst=%d world=%d", nz, ny, nx, dst, comm->world);
    int dev = -1;
    SHARD_HIP(hipGetDevice(&dev));
    if (dev != comm->device)
        return fail(WN_ERR_INVALID, "the communicator was created on device %d, the current device is %d", comm->device, dev);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t plane = (size_t)ny * nx; // floats
    if (piece_bytes == 0) piece_bytes = (size_t)1 << 30;
    // a slab travels in pieces of whole planes of at most piece_bytes: no single message near the 32-bit byte counts
    // some transports still carry (a 2048^3 / 2 slab is 16 GiB)
    size_t step = plane ? piece_bytes / (plane * sizeof(float)) : 1;
    if (step < 1) step = 1;
    int z0 = 0, z1 = 0;
    bounds(nz, comm->world, comm->rank, &z0, &z1);
    if (z1 > z0 && !slab_dev) return fail(WN_ERR_INVALID, "wn_gather_volume: slab_dev is NULL");
    if (comm->rank == dst) {
        if (!out_dev && nz > 0 && plane) return fail(WN_ERR_INVALID, "wn_gather_volume: out_dev is NULL on the destination rank");
        if (z1 > z0 && plane && out_dev + (size_t)z0 * plane != slab_dev)
            SHARD_HIP(hipMemcpyAsync(out_dev + (size_t)z0 * plane, slab_dev, (size_t)(z1 - z0) * plane * sizeof(float),
                                     hipMemcpyDeviceToDevice, s));
    }
    if (comm->world == 1 || plane == 0) return WN_OK;
    SHARD_NCCL(ncclGroupStart());
    if (comm->rank == dst) {
        for (int r = 0; r < comm->world; ++r) {
            if (r == dst) continue;
            int a = 0, b = 0;
            bounds(nz, comm->world, r, &a, &b);
            for (size_t z = (size_t)a; z < (size_t)b; z += step) {
                const size_t n = ((size_t)b - z < step ? (size_t)b - z : step) * plane;
                const ncclResult_t rr = ncclRecv(out_dev + z * plane, n, ncclFloat, r, comm->comm, s);
                if (rr != ncclSuccess) {
                    (void)ncclGroupEnd();
                    return fail(WN_ERR_HIP, "ncclRecv from rank %d: %s", r, ncclGetErrorString(rr));
                }
            }
        }
    } else {
        for (size_t z = 0; z < (size_t)(z1 - z0); z += step) {
            const size_t n = ((size_t)(z1 - z0) - z < step ? (size_t)(z1 - z0) - z : step) * plane;
            const ncclResult_t rr = ncclSend(slab_dev + z * plane, n, ncclFloat, dst, comm->comm, s);
            if (rr != ncclSuccess) {
                (void)ncclGroupEnd();
                return fail(WN_ERR_HIP, "ncclSend to rank %d: %s", dst, ncclGetErrorString(rr));
            }
        }
    }
    SHARD_NCCL(ncclGroupEnd());
    return WN_OK;
}

} // extern "C"
