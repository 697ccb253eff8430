// ref_shim.cpp -- TEST INFRASTRUCTURE.  extern "C" doorway onto the REAL reference classes.
//
// This file contains no noise arithmetic of its own: it instantiates the reference's
// WaveletNoise / perlin / PerlinNoise / noise_texture / wavelet_texture straight from the
// headers under $(REF) (default /root/reference) and loops their scalar entry points over
// arrays.  It is compiled together with $(REF)/WaveletNoise.cpp into oracle/_ref/libwnref.so
// by oracle/Makefile (g++ only; the reference's own build system is not run).  Nothing from
// the reference is copied into this repository; oracle/_ref/ is git-ignored.
//
// Used for: (a) validating oracle/wn_oracle.c, (b) generating tests/golden/ vectors
// (oracle/gen_golden.py), (c) bench.py's cpu_baseline leg (kind "reference").
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <vector>

#include "WaveletNoise.h"  // $(REF)/WaveletNoise.h

// The permutation table is a private member; the shim only needs to *read* it.
#define private public
#include "perlin.h"                  // $(REF)/perlin.h
#include "experient/PerlinNoise.hpp" // $(REF)/experient/PerlinNoise.hpp
#undef private
#include "texture.h"                 // $(REF)/texture.h

extern "C" {

// ---- WaveletNoise ---------------------------------------------------------------------------
void *ref_wn_new(int tile, unsigned seed) { return new WaveletNoise(tile, seed); }
void ref_wn_delete(void *h) { delete static_cast<WaveletNoise *>(h); }
void ref_wn_generate2d(void *h) { static_cast<WaveletNoise *>(h)->generateNoiseTile2D(); }
void ref_wn_generate3d(void *h) { static_cast<WaveletNoise *>(h)->generateNoiseTile3D(); }
int ref_wn_tile_size(void *h) { return static_cast<WaveletNoise *>(h)->getTileSize(); }
size_t ref_wn_coeff_count(void *h)
{
    return static_cast<WaveletNoise *>(h)->getNoiseCoefficients().size();
}
void ref_wn_coeffs(void *h, float *out)
{
    const auto &c = static_cast<WaveletNoise *>(h)->getNoiseCoefficients();
    std::memcpy(out, c.data(), c.size() * sizeof(float));
}
void ref_wn_eval2d(void *h, const float *xy, size_t n, float *out)
{
    const auto *w = static_cast<const WaveletNoise *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = w->evaluate2D(xy + 2 * i);
}
void ref_wn_eval3d(void *h, const float *xyz, size_t n, float *out)
{
    const auto *w = static_cast<const WaveletNoise *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = w->evaluate3D(xyz + 3 * i);
}
void ref_wn_eval3d_projected(void *h, const float *xyz, const float *normals, size_t n,
                             float *out)
{
    const auto *w = static_cast<const WaveletNoise *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = w->evaluate3DProjected(xyz + 3 * i, normals + 3 * i);
}
// the experient/main.cpp:41-56 loop body with z made a lattice axis.
void ref_wn_grid3d_volume(void *h, int den, int nx, int ny, int z0, int z1, int octave,
                          float *out)
{
    const auto *w = static_cast<const WaveletNoise *>(h);
    const float base_range = 4.0f;
    const float octave_scale = std::pow(2.0f, octave);
    const float inv_stddev_3d = 1.0f / std::sqrt(0.18402f);
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                float u = (static_cast<float>(x) / den) * base_range;
                float v = (static_cast<float>(y) / den) * base_range;
                float t = (static_cast<float>(z) / den) * base_range;
                float p[3] = {u * octave_scale, v * octave_scale, t * octave_scale};
                p[0] *= 2.0f;
                p[1] *= 2.0f;
                p[2] *= 2.0f;
                out[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)(z - z0))] =
                    w->evaluate3D(p) * inv_stddev_3d;
            }
}

// ---- libstdc++ streams as the reference draws them ----------------------------------------------
void ref_gaussian_stream(unsigned seed, size_t n, float *out)
{
    std::mt19937 rng(seed);                                  // WaveletNoise.cpp:21
    std::normal_distribution<float> gaussianDist(0.0f, 1.0f);
    for (size_t i = 0; i < n; ++i) out[i] = gaussianDist(rng);
}

// ---- Perlin ---------------------------------------------------------------------------------------
void *ref_perlin_new(unsigned seed) { return new perlin(seed); }
void *ref_perlin_new_default() { return new perlin(); }
void ref_perlin_delete(void *h) { delete static_cast<perlin *>(h); }
void ref_perlin_perm(void *h, int *out512)
{
    const auto &p = static_cast<perlin *>(h)->p;
    for (int i = 0; i < 512; ++i) out512[i] = p[i];
}
void ref_perlin_noise(void *h, const double *xyz, size_t n, double *out)
{
    const auto *pn = static_cast<const perlin *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = pn->noise(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}
void ref_perlin_noise_vec3(void *h, const float *xyz, size_t n, double *out)
{
    const auto *pn = static_cast<const perlin *>(h);
    for (size_t i = 0; i < n; ++i)
        out[i] = pn->noise(point3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}
void ref_perlin_fractal(void *h, const float *xyz, size_t n, double *out)
{
    const auto *pn = static_cast<const perlin *>(h);
    for (size_t i = 0; i < n; ++i)
        out[i] = pn->fractal_noise(point3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
}
void *ref_PerlinNoise_new(unsigned seed) { return new PerlinNoise(seed); }
void ref_PerlinNoise_delete(void *h) { delete static_cast<PerlinNoise *>(h); }
void ref_PerlinNoise_perm(void *h, int *out512)
{
    const auto &p = static_cast<PerlinNoise *>(h)->p;
    for (int i = 0; i < 512; ++i) out512[i] = p[i];
}
void ref_PerlinNoise_noise(void *h, const double *xyz, size_t n, double *out)
{
    const auto *pn = static_cast<const PerlinNoise *>(h);
    for (size_t i = 0; i < n; ++i) out[i] = pn->noise(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

// ---- textures -------------------------------------------------------------------------------------
void *ref_noise_texture_new(double scale, int octave) { return new noise_texture(scale, octave); }
void *ref_wavelet_texture_new(double scale, int octave, int use_3d)
{
    return new wavelet_texture(scale, octave, use_3d != 0);
}
void ref_texture_delete(void *h) { delete static_cast<texture *>(h); }
// out: 3 floats per point (the colour as the reference returns it).
void ref_texture_value(void *h, const float *xyz, size_t n, float *out_rgb)
{
    const auto *t = static_cast<const texture *>(h);
    for (size_t i = 0; i < n; ++i) {
        color c = t->value(0.0, 0.0, point3(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        out_rgb[3 * i] = c.x();
        out_rgb[3 * i + 1] = c.y();
        out_rgb[3 * i + 2] = c.z();
    }
}

} // extern "C"
