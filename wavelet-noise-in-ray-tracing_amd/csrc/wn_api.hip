// wn_api.hip -- C-ABI plumbing of libwnoise_hip.so: errors, device memory, streams, timers,
// the libstdc++ setup streams and the tile / permutation handles.  See include/wnoise.h.
#include "wn_internal.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <numeric>
#include <random>
#include <utility>
#include <vector>

namespace wn {

static char g_private_rand_state[256];
static std::mutex g_rand_mu;
static int g_rand_depth = 0;         // ABI calls (of all threads, nested ones included) inside a guarded region
static char *g_rand_caller = nullptr; // the application's state, parked while g_rand_depth > 0

// The caller's state is parked by the first call that enters the library and put back by the last one
// that leaves, under a mutex: two host threads inside the library can no longer hand each other the
// private state as "caller state" (ADVICE round 1).  A thread that calls rand() itself while another
// thread is inside the library still draws from the private state -- rand() is one process-global
// stream; wnoise.h says so.
RandStateGuard::RandStateGuard()
{
    std::lock_guard<std::mutex> lock(g_rand_mu);
    if (g_rand_depth++ == 0) {
        static bool initialised = false;
        if (!initialised) {
            // initstate switches to the new state and returns the previous (= the caller's) one
            g_rand_caller = initstate(0x776e6f69u, g_private_rand_state, sizeof(g_private_rand_state));
            initialised = true;
        } else {
            g_rand_caller = setstate(g_private_rand_state);
        }
    }
}

RandStateGuard::~RandStateGuard()
{
    std::lock_guard<std::mutex> lock(g_rand_mu);
    if (--g_rand_depth == 0 && g_rand_caller) {
        setstate(g_rand_caller);
        g_rand_caller = nullptr;
    }
}

// ---- per-device facts (mutex-protected tables keyed by device ordinal) ---------------------------
static std::mutex g_dev_mu;
static std::map<int, int> g_dev_cus;
static std::map<std::pair<const void *, int>, size_t> g_lds_optin; // (kernel, device) -> bytes granted

int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return dev;
}

int device_compute_units(int dev)
{
    std::lock_guard<std::mutex> lock(g_dev_mu);
    auto it = g_dev_cus.find(dev);
    if (it != g_dev_cus.end()) return it->second;
    hipDeviceProp_t prop;
    int cus = 256;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    else
        (void)hipGetLastError();
    g_dev_cus[dev] = cus;
    return cus;
}

bool ensure_dynamic_lds(const void *kernel, int dev, size_t bytes)
{
    std::lock_guard<std::mutex> lock(g_dev_mu);
    const auto key = std::make_pair(kernel, dev);
    auto it = g_lds_optin.find(key);
    if (it != g_lds_optin.end() && it->second >= bytes) return true;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
        return false;
    }
    g_lds_optin[key] = bytes;
    return true;
}

int check_handle_device(int handle_device, const char *what)
{
    const int dev = current_device();
    if (dev != handle_device)
        return fail(WN_ERR_INVALID, "%s was created on device %d but the current device is %d", what,
                    handle_device, dev);
    return WN_OK;
}

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? WN_ERR_NO_DEVICE : WN_ERR_HIP;
}

// The product has no CPU path: every compute entry point starts here.
int require_device()
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return fail(WN_ERR_NO_DEVICE,
                    "no HIP device available (libwnoise_hip has no CPU fallback): %s",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    return WN_OK;
}

int check_grid(const wn_grid *g, bool needs_z, GridArgs *out)
{
    if (!g) return fail(WN_ERR_INVALID, "wn_grid is NULL");
    if (g->den <= 0) return fail(WN_ERR_INVALID, "wn_grid.den must be > 0 (got %d)", g->den);
    if (g->nx < 0 || g->ny < 0) return fail(WN_ERR_INVALID, "wn_grid.nx/ny must be >= 0");
    out->den = g->den;
    out->nx = g->nx;
    out->ny = g->ny;
    out->base_range = g->base_range;
    out->octave_scale = g->octave_scale;
    out->post_scale = g->post_scale;
    out->z_const_mode = (g->z_mode == WN_Z_CONST) ? 1 : 0;
    out->z_const = g->z_const;
    out->out_scale = g->out_scale;
    if (!needs_z || out->z_const_mode) {
        out->z0 = 0;
        out->nz = 1;
    } else {
        if (g->z1 < g->z0) return fail(WN_ERR_INVALID, "wn_grid.z1 < z0");
        out->z0 = g->z0;
        out->nz = g->z1 - g->z0;
    }
    return WN_OK;
}

} // namespace wn

using namespace wn;

extern "C" {

const char *wn_last_error(void) { return g_err; }
const char *wn_version(void) { return "wnoise-hip 0.1 (gfx950)"; }

int wn_device_count(int *count)
{
    WN_ENTRY();
    if (!count) return fail(WN_ERR_INVALID, "count is NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    return WN_OK;
}

int wn_device_set(int ordinal)
{
    WN_ENTRY();
    WN_HIP(hipSetDevice(ordinal));
    return WN_OK;
}

int wn_device_get(int *ordinal)
{
    WN_ENTRY();
    if (!ordinal) return fail(WN_ERR_INVALID, "ordinal is NULL");
    WN_HIP(hipGetDevice(ordinal));
    return WN_OK;
}

int wn_device_info(char *name, size_t name_len, int *compute_units, size_t *hbm_bytes)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    int dev = 0;
    WN_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    WN_HIP(hipGetDeviceProperties(&prop, dev));
    if (name && name_len) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return WN_OK;
}

int wn_dev_alloc(void **dptr, size_t bytes)
{
    WN_ENTRY();
    if (!dptr) return fail(WN_ERR_INVALID, "dptr is NULL");
    *dptr = nullptr;
    int rc = require_device();
    if (rc) return rc;
    if (bytes == 0) return WN_OK;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {
        hip_fail(e, "hipMalloc");
        return WN_ERR_ALLOC;
    }
    return WN_OK;
}

int wn_dev_free(void *dptr)
{
    WN_ENTRY();
    if (!dptr) return WN_OK;
    WN_HIP(hipFree(dptr));
    return WN_OK;
}

int wn_host_alloc_mapped(void **host_ptr, void **dev_alias, size_t bytes)
{
    WN_ENTRY();
    if (!host_ptr || !dev_alias) return fail(WN_ERR_INVALID, "host_ptr/dev_alias is NULL");
    *host_ptr = *dev_alias = nullptr;
    int rc = require_device();
    if (rc) return rc;
    if (bytes == 0) return WN_OK;
    hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocMapped);
    if (e != hipSuccess) {
        hip_fail(e, "hipHostMalloc");
        return WN_ERR_ALLOC;
    }
    e = hipHostGetDevicePointer(dev_alias, *host_ptr, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(*host_ptr);
        *host_ptr = nullptr;
        return hip_fail(e, "hipHostGetDevicePointer");
    }
    return WN_OK;
}

int wn_host_free_mapped(void *host_ptr)
{
    WN_ENTRY();
    if (!host_ptr) return WN_OK;
    WN_HIP(hipHostFree(host_ptr));
    return WN_OK;
}

int wn_copy_h2d(void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
    WN_ENTRY();
    if (bytes == 0) return WN_OK;
    if (!dst_dev || !src_host) return fail(WN_ERR_INVALID, "wn_copy_h2d: NULL pointer");
    WN_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return WN_OK;
}

int wn_copy_d2h(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
    WN_ENTRY();
    if (bytes == 0) return WN_OK;
    if (!dst_host || !src_dev) return fail(WN_ERR_INVALID, "wn_copy_d2h: NULL pointer");
    WN_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return WN_OK;
}

int wn_stream_sync(void *stream)
{
    WN_ENTRY();
    WN_HIP(hipStreamSynchronize(as_stream(stream)));
    return WN_OK;
}

// ---- timers ------------------------------------------------------------------------------------
int wn_timer_create(wn_timer **t)
{
    WN_ENTRY();
    if (!t) return fail(WN_ERR_INVALID, "t is NULL");
    *t = nullptr;
    int rc = require_device();
    if (rc) return rc;
    wn_timer *x = new wn_timer();
    hipError_t e = hipEventCreate(&x->start);
    if (e == hipSuccess) e = hipEventCreate(&x->stop);
    if (e != hipSuccess) {
        delete x;
        return hip_fail(e, "hipEventCreate");
    }
    *t = x;
    return WN_OK;
}
int wn_timer_start(wn_timer *t, void *stream)
{
    WN_ENTRY();
    if (!t) return fail(WN_ERR_INVALID, "timer is NULL");
    WN_HIP(hipEventRecord(t->start, as_stream(stream)));
    return WN_OK;
}
int wn_timer_stop(wn_timer *t, void *stream)
{
    WN_ENTRY();
    if (!t) return fail(WN_ERR_INVALID, "timer is NULL");
    WN_HIP(hipEventRecord(t->stop, as_stream(stream)));
    return WN_OK;
}
int wn_timer_elapsed_ms(wn_timer *t, float *ms)
{
    WN_ENTRY();
    if (!t || !ms) return fail(WN_ERR_INVALID, "timer/ms is NULL");
    WN_HIP(hipEventSynchronize(t->stop));
    WN_HIP(hipEventElapsedTime(ms, t->start, t->stop));
    return WN_OK;
}
void wn_timer_destroy(wn_timer *t)
{
    WN_ENTRY();
    if (!t) return;
    if (t->start) (void)hipEventDestroy(t->start);
    if (t->stop) (void)hipEventDestroy(t->stop);
    delete t;
}

// ---- setup streams (host, libstdc++ <random>) ---------------------------------------------------
int wn_gaussian_fill(uint32_t seed, size_t count, float *out_host)
{
    WN_ENTRY();
    if (count && !out_host) return fail(WN_ERR_INVALID, "out_host is NULL");
    std::mt19937 engine(seed);
    std::normal_distribution<float> gauss(0.0f, 1.0f);
    for (size_t i = 0; i < count; ++i) out_host[i] = gauss(engine);
    return WN_OK;
}

int wn_perlin_permutation(uint32_t seed, int out512_host[512])
{
    WN_ENTRY();
    if (!out512_host) return fail(WN_ERR_INVALID, "out512_host is NULL");
    std::vector<int> v(256);
    std::iota(v.begin(), v.end(), 0);
    std::shuffle(v.begin(), v.end(), std::mt19937(seed));
    for (int i = 0; i < 256; ++i) out512_host[i] = out512_host[i + 256] = v[i];
    return WN_OK;
}

// ---- tiles -------------------------------------------------------------------------------------------
int wn_tile_even_size(int requested) { return (requested % 2 != 0) ? requested + 1 : requested; }

static int tile_alloc(int n, int dims, wn_tile **out)
{
    if (!out) return fail(WN_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (dims != 2 && dims != 3) return fail(WN_ERR_INVALID, "dims must be 2 or 3 (got %d)", dims);
    if (n < 0) return fail(WN_ERR_INVALID, "tile size must be >= 0 (got %d)", n);
    if (n > 1024 || (dims == 3 && n > 1024)) return fail(WN_ERR_INVALID, "tile size %d too large", n);
    int rc = require_device();
    if (rc) return rc;
    wn_tile *t = new wn_tile();
    t->n = n;
    t->dims = dims;
    t->count = (dims == 2) ? (size_t)n * n : (size_t)n * n * n;
    (void)hipGetDevice(&t->device);
    if (t->count) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->dev), t->count * sizeof(float));
        if (e != hipSuccess) {
            delete t;
            hip_fail(e, "hipMalloc(tile)");
            return WN_ERR_ALLOC;
        }
    }
    *out = t;
    return WN_OK;
}

int wn_tile_create(int n, int dims, const float *coeffs_host, wn_tile **out)
{
    WN_ENTRY();
    if (!coeffs_host) n = 0; // empty tile: evaluates to 0 (WaveletNoise.cpp:112,186,219)
    if (n % 2 != 0)
        return fail(WN_ERR_INVALID, "wn_tile_create: coefficient tiles have even size (got %d)", n);
    int rc = tile_alloc(n, dims, out);
    if (rc) return rc;
    if ((*out)->count) {
        hipError_t e = hipMemcpy((*out)->dev, coeffs_host, (*out)->count * sizeof(float),
                                 hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            wn_tile_destroy(*out);
            *out = nullptr;
            return hip_fail(e, "hipMemcpy(tile)");
        }
        rc = tile_build_padded(*out, nullptr);
        if (rc) {
            wn_tile_destroy(*out);
            *out = nullptr;
            return rc;
        }
    }
    return WN_OK;
}

int wn_tile_generate_from_field(int n, int dims, const float *field_host, wn_tile **out)
{
    WN_ENTRY();
    if (!field_host && n > 0) return fail(WN_ERR_INVALID, "field_host is NULL");
    if (n % 2 != 0) return fail(WN_ERR_INVALID, "tile size must be even (got %d)", n);
    int rc = tile_alloc(n, dims, out);
    if (rc) return rc;
    wn_tile *t = *out;
    if (!t->count) return WN_OK;
    float *field_dev = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&field_dev), t->count * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(field_dev, field_host, t->count * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (field_dev) (void)hipFree(field_dev);
        wn_tile_destroy(t);
        *out = nullptr;
        return hip_fail(e, "tile field upload");
    }
    rc = wn::tilegen_filter(t, field_dev, nullptr); // wn_tilegen.hip
    hipError_t se = hipStreamSynchronize(nullptr);
    (void)hipFree(field_dev);
    if (rc == WN_OK && se != hipSuccess) rc = hip_fail(se, "tile generation");
    if (rc == WN_OK) rc = tile_build_padded(t, nullptr);
    if (rc) {
        wn_tile_destroy(t);
        *out = nullptr;
    }
    return rc;
}

int wn_tile_generate(int n, int dims, uint32_t seed, wn_tile **out)
{
    WN_ENTRY();
    if (n < 0) return fail(WN_ERR_INVALID, "tile size must be >= 0");
    if (dims != 2 && dims != 3) return fail(WN_ERR_INVALID, "dims must be 2 or 3 (got %d)", dims);
    const int even = wn_tile_even_size(n);
    if (even != n)
        fprintf(stderr, "Warning: Tile size adjusted to %d (must be even)\n", even); // :22-25
    const size_t count = (dims == 2) ? (size_t)even * even : (size_t)even * even * even;
    std::vector<float> field(count);
    wn_gaussian_fill(seed, count, field.data());
    return wn_tile_generate_from_field(even, dims, field.data(), out);
}

int wn_tile_size(const wn_tile *t) { return t ? t->n : 0; }
int wn_tile_dims(const wn_tile *t) { return t ? t->dims : 0; }
size_t wn_tile_count(const wn_tile *t) { return t ? t->count : 0; }
const float *wn_tile_device_ptr(const wn_tile *t) { return t ? t->dev : nullptr; }

int wn_tile_download(const wn_tile *t, float *out_host)
{
    WN_ENTRY();
    if (!t) return fail(WN_ERR_INVALID, "tile is NULL");
    if (!t->count) return WN_OK;
    if (!out_host) return fail(WN_ERR_INVALID, "out_host is NULL");
    WN_HIP(hipMemcpy(out_host, t->dev, t->count * sizeof(float), hipMemcpyDeviceToHost));
    return WN_OK;
}

void wn_tile_destroy(wn_tile *t)
{
    WN_ENTRY();
    if (!t) return;
    if (t->dev) (void)hipFree(t->dev);
    if (t->dev_padded) (void)hipFree(t->dev_padded);
    delete t;
}

// ---- permutation tables ----------------------------------------------------------------------------------
int wn_perm_create(const int table512_host[512], wn_perm **out)
{
    WN_ENTRY();
    if (!out) return fail(WN_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!table512_host) return fail(WN_ERR_INVALID, "table is NULL");
    uint8_t bytes[512];
    for (int i = 0; i < 512; ++i) {
        if (table512_host[i] < 0 || table512_host[i] > 255)
            return fail(WN_ERR_INVALID, "permutation entry %d out of 0..255", i);
        bytes[i] = (uint8_t)table512_host[i];
    }
    int rc = require_device();
    if (rc) return rc;
    wn_perm *p = new wn_perm();
    std::memcpy(p->host, table512_host, sizeof(p->host));
    (void)hipGetDevice(&p->device);
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->dev), 512);
    if (e == hipSuccess) e = hipMemcpy(p->dev, bytes, 512, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->dev) (void)hipFree(p->dev);
        delete p;
        return hip_fail(e, "perm upload");
    }
    *out = p;
    return WN_OK;
}

int wn_perm_create_seeded(uint32_t seed, wn_perm **out)
{
    WN_ENTRY();
    int table[512];
    wn_perlin_permutation(seed, table);
    return wn_perm_create(table, out);
}

int wn_perm_download(const wn_perm *p, int out512_host[512])
{
    WN_ENTRY();
    if (!p || !out512_host) return fail(WN_ERR_INVALID, "perm/out is NULL");
    uint8_t bytes[512];
    WN_HIP(hipMemcpy(bytes, p->dev, 512, hipMemcpyDeviceToHost));
    for (int i = 0; i < 512; ++i) out512_host[i] = bytes[i];
    return WN_OK;
}

void wn_perm_destroy(wn_perm *p)
{
    WN_ENTRY();
    if (!p) return;
    if (p->dev) (void)hipFree(p->dev);
    delete p;
}

} // extern "C"
