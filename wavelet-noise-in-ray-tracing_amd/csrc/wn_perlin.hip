// wn_perlin.hip -- Ken Perlin "improved noise" (perlin.h:42-62, experient/PerlinNoise.hpp:36-60)
// for gfx950: dense grids, point lists, turb / fractal_noise, and the noise_texture adaptor
// (K4 / K6 of SURVEY.md 8).
//
// The 512-entry permutation table sits in LDS as bytes; all arithmetic is fp64 in the
// reference's order with contraction off, and the integer hash path is plain int arithmetic, so
// results are bit-identical to the CPU classes.  Grids write one float per sample (4 B/sample
// of HBM traffic); the work per sample is ~60 fp64 operations, so this path is fp64-VALU bound,
// not HBM bound.
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"

#include <cmath>

namespace {

using wn::GridArgs;

enum { kNoise = 0, kTurb = 1, kFractal = 2 };

struct PerlinGridArgs {
    const uint8_t *perm;
    float *out;
    GridArgs g;
    int kind, depth;
};

__global__ __launch_bounds__(256) void perlin_grid_kernel(const PerlinGridArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    const GridArgs &g = a.g;
    const float den = (float)g.den;
    const unsigned plane = (unsigned)g.nx * (unsigned)g.ny;
    const size_t total = (size_t)plane * g.nz;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const unsigned z = (unsigned)(e / plane);
        const unsigned r = (unsigned)(e - (size_t)z * plane);
        const unsigned y = r / (unsigned)g.nx, x = r - y * (unsigned)g.nx;
        const float px = wn::lattice_coord((int)x, den, g.base_range, g.octave_scale, g.post_scale);
        const float py = wn::lattice_coord((int)y, den, g.base_range, g.octave_scale, g.post_scale);
        const float pz = g.z_const_mode ? g.z_const
                                        : wn::lattice_coord(g.z0 + (int)z, den, g.base_range,
                                                            g.octave_scale, g.post_scale);
        double v;
        if (a.kind == kNoise) v = wn::perlin_exact(perm, (double)px, (double)py, (double)pz);
        else if (a.kind == kTurb) v = wn::perlin_turb(perm, px, py, pz, a.depth);
        else v = wn::perlin_fractal(perm, px, py, pz);
        a.out[e] = (float)v * g.out_scale;
    }
}

struct PerlinPointsArgs {
    const uint8_t *perm;
    const double *pts64;
    const float *pts32;
    double *out;
    size_t count;
    int kind, depth;
};

__global__ __launch_bounds__(256) void perlin_points_kernel(const PerlinPointsArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (a.pts64) {
            const double *p = a.pts64 + 3 * i;
            v = wn::perlin_exact(perm, p[0], p[1], p[2]);
        } else {
            const float *p = a.pts32 + 3 * i;
            if (a.kind == kNoise) v = wn::perlin_exact(perm, (double)p[0], (double)p[1], (double)p[2]);
            else if (a.kind == kTurb) v = wn::perlin_turb(perm, p[0], p[1], p[2], a.depth);
            else v = wn::perlin_fractal(perm, p[0], p[1], p[2]);
        }
        a.out[i] = v;
    }
}

// noise_texture::value (texture.h:37-43) with the same ballot compaction as the wavelet texture.
struct NoiseTexArgs {
    const uint8_t *perm;
    float fscale;       // (float)scale: vec3 * float (vec3.h:82-84)
    float octave_scale; // (float)pow(2, octave)
    const float *pts;
    const uint8_t *active;
    float *grey;
    size_t count;
    int points_per_wave;
};

__device__ __forceinline__ float noise_texture_value(const uint8_t *perm, const NoiseTexArgs &a,
                                                     float px, float py, float pz)
{
    const float sx = (a.fscale * px) * a.octave_scale;
    const float sy = (a.fscale * py) * a.octave_scale;
    const float sz = (a.fscale * pz) * a.octave_scale;
    double v = wn::perlin_exact(perm, (double)sx, (double)sy, (double)sz);
    v = 0.5 * (1.0 + v);
    return (float)v;
}

template <bool MASKED>
__global__ __launch_bounds__(256) void noise_texture_kernel(const NoiseTexArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    __shared__ float q_x[4][128], q_y[4][128], q_z[4][128];
    __shared__ unsigned q_i[4][128];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gwave = (size_t)blockIdx.x * 4 + wave;
    const size_t begin = gwave * (size_t)a.points_per_wave;
    if (begin >= a.count) return;
    const size_t end = min(a.count, begin + (size_t)a.points_per_wave);

    if (!MASKED) {
        for (size_t i = begin + lane; i < end; i += 64) {
            const float *p = a.pts + 3 * i;
            a.grey[i] = noise_texture_value(perm, a, p[0], p[1], p[2]);
        }
        return;
    }
    int queued = 0;
    for (size_t base = begin; base < end; base += 64) {
        const size_t i = base + lane;
        const bool hit = (i < end) && (a.active[i] != 0);
        const unsigned long long ballot = __ballot(hit);
        if (hit) {
            const int slot = queued + __popcll(ballot & ((1ull << lane) - 1ull));
            const float *p = a.pts + 3 * i;
            q_x[wave][slot] = p[0];
            q_y[wave][slot] = p[1];
            q_z[wave][slot] = p[2];
            q_i[wave][slot] = (unsigned)(i - begin);
        }
        queued += __popcll(ballot);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (queued >= 64) {
            const float x = q_x[wave][lane], y = q_y[wave][lane], z = q_z[wave][lane];
            const unsigned idx = q_i[wave][lane];
            const int rest = queued - 64;
            float cx = 0, cy = 0, cz = 0;
            unsigned ci = 0;
            if (lane < rest) {
                cx = q_x[wave][64 + lane];
                cy = q_y[wave][64 + lane];
                cz = q_z[wave][64 + lane];
                ci = q_i[wave][64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < rest) {
                q_x[wave][lane] = cx;
                q_y[wave][lane] = cy;
                q_z[wave][lane] = cz;
                q_i[wave][lane] = ci;
            }
            queued = rest;
            a.grey[begin + idx] = noise_texture_value(perm, a, x, y, z);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (lane < queued) {
        const unsigned idx = q_i[wave][lane];
        a.grey[begin + idx] =
            noise_texture_value(perm, a, q_x[wave][lane], q_y[wave][lane], q_z[wave][lane]);
    }
}

inline int blocks_for(size_t total)
{
    size_t b = (total + 255) / 256;
    const size_t cap = 256u * 8u * 8u;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int perlin_grid(const wn_perm *perm, const wn_grid *grid, int kind, int depth, float *out_dev,
                void *stream)
{
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    GridArgs g;
    rc = wn::check_grid(grid, true, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    if (total == 0) return WN_OK;
    if (!out_dev) return wn::fail(WN_ERR_INVALID, "out_dev is NULL");
    if ((size_t)g.nx * g.ny > 0xffffffffull) return wn::fail(WN_ERR_INVALID, "plane too large");
    PerlinGridArgs a{perm->dev, out_dev, g, kind, depth};
    hipLaunchKernelGGL(perlin_grid_kernel, dim3(blocks_for(total)), dim3(256), 0,
                       wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("perlin_grid_kernel");
    return WN_OK;
}

int perlin_points(const wn_perm *perm, const double *p64, const float *p32, size_t n, int kind,
                  int depth, double *out_dev, void *stream)
{
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    if (n == 0) return WN_OK;
    if ((!p64 && !p32) || !out_dev) return wn::fail(WN_ERR_INVALID, "points/out pointer is NULL");
    PerlinPointsArgs a{perm->dev, p64, p32, out_dev, n, kind, depth};
    hipLaunchKernelGGL(perlin_points_kernel, dim3(blocks_for(n)), dim3(256), 0,
                       wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("perlin_points_kernel");
    return WN_OK;
}

} // namespace

extern "C" {

int wn_perlin_grid(const wn_perm *perm, const wn_grid *g, float *out_dev, void *stream)
{
    WN_ENTRY();
    return perlin_grid(perm, g, kNoise, 0, out_dev, stream);
}
int wn_perlin_turb_grid(const wn_perm *perm, const wn_grid *g, int depth, float *out_dev,
                        void *stream)
{
    WN_ENTRY();
    if (depth < 0) return wn::fail(WN_ERR_INVALID, "depth must be >= 0");
    return perlin_grid(perm, g, kTurb, depth, out_dev, stream);
}
int wn_perlin_fractal_grid(const wn_perm *perm, const wn_grid *g, float *out_dev, void *stream)
{
    WN_ENTRY();
    return perlin_grid(perm, g, kFractal, 0, out_dev, stream);
}
int wn_perlin_points(const wn_perm *perm, const double *xyz_dev, size_t n, double *out_dev,
                     void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, xyz_dev, nullptr, n, kNoise, 0, out_dev, stream);
}
int wn_perlin_points_vec3(const wn_perm *perm, const float *xyz_dev, size_t n, double *out_dev,
                          void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, nullptr, xyz_dev, n, kNoise, 0, out_dev, stream);
}
int wn_perlin_turb_points(const wn_perm *perm, const float *xyz_dev, size_t n, int depth,
                          double *out_dev, void *stream)
{
    WN_ENTRY();
    if (depth < 0) return wn::fail(WN_ERR_INVALID, "depth must be >= 0");
    return perlin_points(perm, nullptr, xyz_dev, n, kTurb, depth, out_dev, stream);
}
int wn_perlin_fractal_points(const wn_perm *perm, const float *xyz_dev, size_t n, double *out_dev,
                             void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, nullptr, xyz_dev, n, kFractal, 0, out_dev, stream);
}

int wn_noise_texture_points(const wn_perm *perm, double scale, int octave, const float *xyz_dev,
                            const uint8_t *active_dev, size_t n, float *grey_dev, void *stream)
{
    WN_ENTRY();
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    if (n == 0) return WN_OK;
    if (!xyz_dev || !grey_dev) return wn::fail(WN_ERR_INVALID, "points/grey pointer is NULL");
    NoiseTexArgs a{};
    a.perm = perm->dev;
    a.fscale = (float)scale;
    a.octave_scale = (float)std::pow(2.0, (double)octave); // texture.h:38
    a.pts = xyz_dev;
    a.active = active_dev;
    a.grey = grey_dev;
    a.count = n;
    a.points_per_wave = 1024;
    const size_t waves = (n + a.points_per_wave - 1) / a.points_per_wave;
    const size_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffull) return wn::fail(WN_ERR_INVALID, "too many points");
    if (active_dev)
        hipLaunchKernelGGL(noise_texture_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                           wn::as_stream(stream), a);
    else
        hipLaunchKernelGGL(noise_texture_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                           wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("noise_texture_kernel");
    return WN_OK;
}

} // extern "C"
