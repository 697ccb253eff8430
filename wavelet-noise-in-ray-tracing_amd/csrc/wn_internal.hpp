// wn_internal.hpp -- shared between the C-ABI translation units of libwnoise_hip.so.
// gfx950 only; compiled with -ffp-contract=off (fused multiply-adds appear only where a kernel
// asks for them with __builtin_fmaf).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "wnoise.h"

struct wn_tile {
    int n = 0;          // even tile size (0 = empty tile)
    int dims = 0;       // 2 or 3
    size_t count = 0;   // n^dims
    float *dev = nullptr;
    // 3-D tiles also keep a copy whose rows carry two wrap-around columns (row stride n+2,
    // padded[x] = tile[x mod n] for x in [0, n+2)): the three x taps of a scattered point are then
    // always adjacent and come with ONE 12-byte load (9 gathers per point instead of 27)
    float *dev_padded = nullptr;
    int device = 0;
};

struct wn_perm {
    int host[512];
    uint8_t *dev = nullptr; // 512 bytes: values 0..255 (perlin.h:35-38)
    int device = 0;
};

struct wn_timer {
    hipEvent_t start = nullptr, stop = nullptr;
};

namespace wn {

void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
int require_device();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// The HIP runtime draws from the C library's global rand() state while it launches kernels
// (measured on ROCm 7.2: 100 launches shift the caller's rand() sequence).  The reference's
// renderer is deterministic only through the unseeded global rand() (main.cpp:184-185,
// rtweekend.h:37-40), so every ABI entry point parks the caller's random state and lets HIP
// consume a private one; the caller's stream is exactly what it would be without the library.
// Depth-counted under a mutex: the state is parked by the first ABI call that enters the library
// (from any thread) and restored by the last one that leaves.
class RandStateGuard {
  public:
    RandStateGuard();
    ~RandStateGuard();
    RandStateGuard(const RandStateGuard &) = delete;
    RandStateGuard &operator=(const RandStateGuard &) = delete;
};
#define WN_ENTRY() ::wn::RandStateGuard wn_rand_state_guard_

#define WN_HIP(call)                                            \
    do {                                                        \
        hipError_t _e = (call);                                 \
        if (_e != hipSuccess) return ::wn::hip_fail(_e, #call); \
    } while (0)

#define WN_LAUNCH_CHECK(name)                                    \
    do {                                                         \
        hipError_t _e = hipGetLastError();                       \
        if (_e != hipSuccess) return ::wn::hip_fail(_e, name);   \
    } while (0)

// Grid coordinate arguments shared by all dense-grid kernels (mirror of wn_grid plus the
// derived slab extent).
struct GridArgs {
    int den, nx, ny, z0, nz; // nz = planes in this call
    float base_range, octave_scale, post_scale;
    int z_const_mode;
    float z_const;
    float out_scale;
};

int check_grid(const wn_grid *g, bool needs_z, GridArgs *out);

// Per-device facts, kept in mutex-protected tables keyed by the device ordinal (a host may drive
// several devices from several threads).
int current_device();
int device_compute_units(int dev);
// Opt `kernel` in to `bytes` of dynamic LDS on device `dev` (needed beyond 64 KiB), once per
// (kernel, device).  false = the runtime refused: the caller falls back to another kernel.
bool ensure_dynamic_lds(const void *kernel, int dev, size_t bytes);
// WN_ERR_INVALID unless the handle (tile / perm) lives on the current device.
int check_handle_device(int handle_device, const char *what);

// wn_wavelet_strip.hip: launches the strip-march kernel when the lattice is in its regime.
int strip_try(const wn_tile *tile, const GridArgs &g, float *out_dev, hipStream_t stream, bool *launched);

// wn_wavelet_multiband.hip: launches the plane-pipeline kernel when a lattice of 1..5 bands is in its regime (g carries the
// bands' common post_scale; oscale / weights per band; out_div = sqrt(sum w^2 * variance)) and has at least
// min_bricks_per_cu bricks of 512 x 8 x 8 samples per compute unit.
int multiband_try(const wn_tile *tile, const GridArgs &g, int nbands, const float *oscale, const float *weights,
                  float out_div, float *out_dev, hipStream_t stream, bool *launched, int min_bricks_per_cu);

// wn_wavelet_exact.hip: bit-exact dense 3-D grids with the coefficient box staged in LDS; *launched tells the caller.
int exact_lds_try(const wn_tile *tile, const GridArgs &g, float *out_dev, hipStream_t stream, bool *launched);

// wn_tilegen.hip: the filter half of generateNoiseTile2D/3D on the device.
int tilegen_filter(wn_tile *t, const float *field_dev, hipStream_t stream);
// wn_tilegen.hip: (re)build t->dev_padded from t->dev (no-op for 2-D tiles).
int tile_build_padded(wn_tile *t, hipStream_t stream);

} // namespace wn

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
namespace wn {

// Non-negative modulo (WaveletNoise.cpp:31-34); `mask` = n-1 when n is a power of two, else -1.
__device__ __forceinline__ int dmod(int x, int n, int mask)
{
    if (mask >= 0) return x & mask;
    int m = x % n;
    return m < 0 ? m + n : m;
}

// Lattice coordinate of index i (experient/main.cpp:20-26): ((float(i)/den)*range)*octave*post,
// one float rounding per operation, division IEEE-correct (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).
__device__ __forceinline__ float lattice_coord(int i, float den, float range, float oscale,
                                               float post)
{
    float c = ((float)i / den) * range;
    c = c * oscale;
    c = c * post;
    return c;
}

// Quadratic B-spline weights (WaveletNoise.cpp:194-200).
__device__ __forceinline__ void bspline(float p, int &mid, float &w0, float &w1, float &w2)
{
    const float pm = p - 0.5f;
    const float cm = ceilf(pm);
    mid = (int)cm;
    const float t = cm - pm;
    w0 = t * t / 2.0f;
    w2 = (1.0f - t) * (1.0f - t) / 2.0f;
    w1 = 1.0f - w0 - w2;
}

} // namespace wn
#endif
