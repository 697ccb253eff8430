/*
 * wn_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's per-sample noise path, written so that
 * every function follows the reference's operation order exactly (fp32 for the
 * wavelet path, fp64 for the Perlin path, no FMA contraction: build with
 * -ffp-contract=off).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product (libwnoise_hip.so and the host
 * classes above it) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  The restatement is checked against
 *   (1) all 15 committed reference outputs experient/result_raw/ *.raw
 *       (byte-for-byte, tests/test_oracle_golden.py),
 *   (2) vectors produced by the real reference compiled from /root/reference
 *       (oracle/_ref, recipe in oracle/Makefile; vectors in tests/golden/),
 *   (3) the fingerprints recorded in SURVEY.md section 8(c).
 *
 * All file:line citations are relative to the reference checkout.
 */
#ifndef WN_ORACLE_H
#define WN_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libstdc++ <random> streams the reference depends on (GCC 11.4 libstdc++) ------------- */

/* std::mt19937 (standard-exact). */
typedef struct wno_mt19937 {
    uint32_t mt[624];
    int      idx;
} wno_mt19937;
void     wno_mt_seed(wno_mt19937 *g, uint32_t seed);
uint32_t wno_mt_next(wno_mt19937 *g);

/* std::normal_distribution<float>(0,1) driven by mt19937, libstdc++ polar method with its
 * one-value cache (bits/random.tcc normal_distribution::operator()).  Call sites:
 * WaveletNoise.cpp:21,76,147. */
typedef struct wno_normal {
    wno_mt19937 g;
    float       saved;
    int         saved_available;
} wno_normal;
void  wno_normal_seed(wno_normal *d, uint32_t seed);
float wno_normal_next(wno_normal *d);

/* perlin.h:34-39 / experient/PerlinNoise.hpp:29-34: iota(0..255), std::shuffle with a fresh
 * mt19937(seed), duplicated to 512 entries. */
void wno_perlin_perm(uint32_t seed, int p[512]);

/* ---- wavelet tile generation (WaveletNoise.cpp:20-26, 37-108, 142-183) --------------------- */

/* WaveletNoise.cpp:22-25: odd tile sizes are bumped to the next even number. */
int  wno_tile_size(int requested);
/* `out` holds n*n (2-D) or n*n*n (3-D) floats, x fastest; n must already be even. */
void wno_generate_tile2d(int n, uint32_t seed, float *out);
void wno_generate_tile3d(int n, uint32_t seed, float *out);
/* The filter half alone (everything after the Gaussian fill); r holds the Gaussian field. */
void wno_filter_tile2d(int n, const float *r, float *out);
void wno_filter_tile3d(int n, const float *r, float *out);

/* ---- per-sample evaluation (the hot path) --------------------------------------------------- */

int   wno_mod(int x, int n);                                               /* WaveletNoise.cpp:31-34  */
float wno_evaluate2d(const float *coef, size_t count, const float p[2]);   /* WaveletNoise.cpp:111-140 */
float wno_evaluate3d(const float *coef, size_t count, const float p[3]);   /* WaveletNoise.cpp:185-215 */
float wno_evaluate3d_projected(const float *coef, size_t count,            /* WaveletNoise.cpp:218-265 */
                               const float p[3], const float normal[3]);

double wno_perlin_noise(const int p[512], double x, double y, double z);   /* perlin.h:42-62 */
/* perlin.h:75-90 (point3 is a float vec3: the products p.x()*frequency are float*double). */
double wno_perlin_fractal(const int p[512], float x, float y, float z);
/* Absent from the reference; "Ray Tracing: The Next Week" turb(p, depth) on the reference's
 * float vec3: accum += weight*noise(temp_p); weight *= 0.5; temp_p *= 2 (float); fabs(accum). */
double wno_perlin_turb(const int p[512], float x, float y, float z, int depth);

/* Absent from the reference; Cook & DeRose 2005 Appendix 2 WMultibandNoise (normal == NULL
 * branch) composed over wno_evaluate3d.  variance = sum_b w[b]^2 (over ALL nbands), result
 * divided by sqrt(variance * var_per_band).  var_per_band: 0.210 in the paper; the reference's
 * empirical 3-D value is 0.18402 (texture.h:84). */
float wno_multiband3d(const float *coef, size_t count, const float p[3], float s,
                      int first_band, int nbands, const float *w, float var_per_band);
/* the normal != NULL branch of the same Appendix-2 function (bands are evaluate3DProjected) */
float wno_multiband3d_projected(const float *coef, size_t count, const float p[3], const float normal[3],
                                float s, int first_band, int nbands, const float *w, float var_per_band);

/* ---- texture adaptor (texture.h) ------------------------------------------------------------- */

/* noise_texture::value, texture.h:37-43.  Returns the grey level (all three channels equal). */
float wno_noise_texture_value(const int perm[512], double scale, int octave, const float p[3]);
/* wavelet_texture::value, texture.h:67-107.  use_3d selects the 3-D (evaluate3D, 0.18402) or
 * the 2-D (evaluate2D on x,y, 0.19686) branch; a NULL/empty tile gives the 0.0 branch. */
float wno_wavelet_texture_value(const float *coef, size_t count, int use_3d,
                                double scale, int octave, const float p[3]);

/* ---- dense-grid generators (experient/main.cpp) ---------------------------------------------- */

/* :11-36   */ void wno_grid_wavelet2d(const float *coef2d, size_t count, int image, int octave, float *out);
/* :38-64   */ void wno_grid_wavelet3d_sliced(const float *coef3d, size_t count, int image, int octave, float *out);
/* :66-93   */ void wno_grid_wavelet3d_projected(const float *coef3d, size_t count, int image, int octave, float *out);
/* :95-111  */ void wno_grid_perlin2d(const int perm[512], int image, int octave, float *out);
/* :113-129 */ void wno_grid_perlin3d_sliced(const int perm[512], int image, int octave, float *out);

/* the 3-D-sliced mapping extended to a full z axis,
 *   q_a = ((float(i_a)/den) * 4.0f) * 2^octave * 2.0f,  out[x + nx*(y + ny*(z-z0))] = evaluate3D(q)*inv_stddev
 * for x<nx, y<ny, z0<=z<z1 (den is the divisor of every axis: the reference's imageSize). */
void wno_grid_wavelet3d_volume(const float *coef3d, size_t count, int den, int nx, int ny,
                               int z0, int z1, int octave, float *out);
/* Same lattice, p = (float(i)/den)*4.0f un-scaled, fed to wno_multiband3d / wno_perlin_turb (config 3). */
void wno_grid_multiband3d_volume(const float *coef3d, size_t count, int den, int nx, int ny,
                                 int z0, int z1, float s, int first_band, int nbands,
                                 const float *w, float var_per_band, float *out);
void wno_grid_perlin_volume(const int perm[512], int den, int nx, int ny, int z0, int z1,
                            int octave, float *out);
void wno_grid_turb_volume(const int perm[512], int den, int nx, int ny, int z0, int z1,
                          int depth, float *out);

/* FNV-1a 64 over raw bytes (fingerprints of SURVEY 8(c)). */
uint64_t wno_fnv1a64(const void *data, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* WN_ORACLE_H */
