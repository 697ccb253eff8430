/*
 * wn_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See wn_oracle.h.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared wn_oracle.c -lm   (oracle/Makefile)
 * -ffp-contract=off matters: the reference is built by g++ for baseline x86-64 (no FMA), and
 * the byte-for-byte match with experient/result_raw depends on unfused multiply/add.
 */
#include "wn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================================
 * libstdc++ random streams (GCC 11.4: bits/random.h, bits/random.tcc, bits/uniform_int_dist.h,
 * bits/stl_algo.h).  Restated, not linked: the oracle is plain C.
 * ========================================================================================== */

void wno_mt_seed(wno_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) {
        uint32_t prev = g->mt[i - 1];
        g->mt[i] = 1812433253u * (prev ^ (prev >> 30)) + (uint32_t)i;
    }
    g->idx = 624;
}

static void mt_refill(wno_mt19937 *g)
{
    uint32_t *mt = g->mt;
    for (int i = 0; i < 624; ++i) {
        uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
        uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
        if (y & 1u) v ^= 0x9908b0dfu;
        mt[i] = v;
    }
    g->idx = 0;
}

uint32_t wno_mt_next(wno_mt19937 *g)
{
    if (g->idx >= 624) mt_refill(g);
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* std::generate_canonical<float, 24>(mt19937): one 32-bit draw, converted to float (round to
 * nearest even), divided by 2^32, clamped below 1 (random.tcc:3345-3380). */
static float canonical_float(wno_mt19937 *g)
{
    float sum = 0.0f;
    sum += (float)wno_mt_next(g) * 1.0f;
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}

void wno_normal_seed(wno_normal *d, uint32_t seed)
{
    wno_mt_seed(&d->g, seed);
    d->saved = 0.0f;
    d->saved_available = 0;
}

/* normal_distribution<float>::operator(), Marsaglia polar with one cached value
 * (random.tcc:1802-1835). */
float wno_normal_next(wno_normal *d)
{
    float ret;
    if (d->saved_available) {
        d->saved_available = 0;
        ret = d->saved;
    } else {
        float x, y, r2;
        do {
            x = (float)((double)(2.0f * canonical_float(&d->g)) - 1.0);
            y = (float)((double)(2.0f * canonical_float(&d->g)) - 1.0);
            r2 = x * x + y * y;
        } while ((double)r2 > 1.0 || (double)r2 == 0.0);
        const float mult = sqrtf(-2.0f * logf(r2) / r2);
        d->saved = x * mult;
        d->saved_available = 1;
        ret = y * mult;
    }
    return ret * 1.0f + 0.0f; /* * stddev + mean */
}

/* uniform_int_distribution<unsigned long>{0, range-1} on a 32-bit-range engine: Lemire's
 * nearly-divisionless method, uniform_int_dist.h:243-270 (_S_nd<uint64_t>). */
static uint32_t lemire_below(wno_mt19937 *g, uint32_t range)
{
    uint64_t product = (uint64_t)wno_mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)wno_mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

static void swap_int(int *a, int *b) { int t = *a; *a = *b; *b = t; }

/* std::shuffle(first, first+256, mt19937(seed)), stl_algo.h:3729-3792: the engine range
 * (2^32-1) is >= 256*256, so elements are swapped in pairs from one draw each. */
void wno_perlin_perm(uint32_t seed, int p[512])
{
    wno_mt19937 g;
    wno_mt_seed(&g, seed);
    const uint32_t count = 256;
    for (uint32_t i = 0; i < count; ++i) p[i] = (int)i;

    uint32_t i = 1;
    if ((count % 2u) == 0u) {
        uint32_t j = lemire_below(&g, 2u);
        swap_int(&p[i], &p[j]);
        ++i;
    }
    while (i != count) {
        const uint32_t swap_range = i + 1u;
        const uint32_t b1 = swap_range + 1u;
        const uint32_t x = lemire_below(&g, swap_range * b1);
        swap_int(&p[i], &p[x / b1]);
        ++i;
        swap_int(&p[i], &p[x % b1]);
        ++i;
    }
    for (uint32_t k = 0; k < count; ++k) p[count + k] = p[k];
}

/* ============================================================================================
 * Tile generation
 * ========================================================================================== */

/* Appendix-1 analysis filter as the reference holds it (WaveletNoise.cpp:11-16); note the
 * 0.003545 / 0.003546 asymmetry is the reference's. */
static const float k_analysis[32] = {
    0.000334f, -0.001528f, 0.000410f,  0.003545f, -0.000938f, -0.008233f, 0.002172f,  0.019120f,
    -0.005040f, -0.044412f, 0.011655f, 0.103311f, -0.025936f, -0.243780f, 0.033979f,  0.655340f,
    0.655340f,  0.033979f,  -0.243780f, -0.025936f, 0.103311f, 0.011655f, -0.044412f, -0.005040f,
    0.019120f,  0.002172f,  -0.008233f, -0.000938f, 0.003546f, 0.000410f, -0.001528f, 0.000334f};
/* WaveletNoise.cpp:18 */
static const float k_synthesis[4] = {0.25f, 0.75f, 0.75f, 0.25f};

int wno_tile_size(int requested) { return (requested % 2 != 0) ? requested + 1 : requested; }

int wno_mod(int x, int n)
{
    int m = x % n;
    return m < 0 ? m + n : m;
}

/* WaveletNoise.cpp:37-48: 32 taps, k = -16..15, accumulated in that order. */
static void downsample_line(const float *from, float *to, int n)
{
    for (int i = 0; i < n / 2; ++i) {
        float acc = 0.0f;
        for (int k = -16; k < 16; ++k) acc += k_analysis[16 + k] * from[wno_mod(2 * i + k, n)];
        to[i] = acc;
    }
}

/* WaveletNoise.cpp:51-66: k runs over i/2 and i/2+1, tap index i-2k must lie in [-2,1]. */
static void upsample_line(const float *from, float *to, int n)
{
    const int half = n / 2;
    for (int i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int k = i / 2; k <= i / 2 + 1; ++k) {
            const int tap = i - 2 * k;
            if (tap >= -2 && tap <= 1) acc += k_synthesis[2 + tap] * from[wno_mod(k, half)];
        }
        to[i] = acc;
    }
}

/* One separable pass: every line along `axis_stride` (line count = total/n) goes through
 * downsample then upsample.  lines are enumerated by (outer, inner) strides. */
static void lowpass_axis(const float *src, float *dst, int n, size_t line_stride,
                         size_t count_a, size_t stride_a, size_t count_b, size_t stride_b)
{
    float *in = (float *)malloc(sizeof(float) * (size_t)n);
    float *half = (float *)malloc(sizeof(float) * (size_t)(n / 2));
    float *outl = (float *)malloc(sizeof(float) * (size_t)n);
    for (size_t b = 0; b < count_b; ++b)
        for (size_t a = 0; a < count_a; ++a) {
            const size_t base = a * stride_a + b * stride_b;
            for (int i = 0; i < n; ++i) in[i] = src[base + (size_t)i * line_stride];
            downsample_line(in, half, n);
            upsample_line(half, outl, n);
            for (int i = 0; i < n; ++i) dst[base + (size_t)i * line_stride] = outl[i];
        }
    free(in);
    free(half);
    free(outl);
}

void wno_filter_tile2d(int n, const float *r, float *out)
{
    const size_t N = (size_t)n, total = N * N;
    float *t1 = (float *)malloc(sizeof(float) * total);
    float *t2 = (float *)malloc(sizeof(float) * total);
    lowpass_axis(r, t1, n, 1, N, N, 1, 0);  /* rows    (WaveletNoise.cpp:87-92)  */
    lowpass_axis(t1, t2, n, N, N, 1, 1, 0); /* columns (WaveletNoise.cpp:95-100) */
    for (size_t i = 0; i < total; ++i) out[i] = r[i] - t2[i]; /* :104-107 */
    free(t1);
    free(t2);
}

void wno_filter_tile3d(int n, const float *r, float *out)
{
    const size_t N = (size_t)n, total = N * N * N;
    float *t1 = (float *)malloc(sizeof(float) * total);
    float *t2 = (float *)malloc(sizeof(float) * total);
    lowpass_axis(r, t1, n, 1, N, N, N, N * N);      /* X lines, for z, y (:153-159) */
    lowpass_axis(t1, t2, n, N, N, 1, N, N * N);     /* Y lines, for z, x (:162-168) */
    lowpass_axis(t2, t1, n, N * N, N, 1, N, N);     /* Z lines, for y, x (:171-177) */
    for (size_t i = 0; i < total; ++i) out[i] = r[i] - t1[i]; /* :179-182 */
    free(t1);
    free(t2);
}

static void gaussian_fill(uint32_t seed, size_t total, float *r)
{
    wno_normal d;
    wno_normal_seed(&d, seed);
    for (size_t i = 0; i < total; ++i) r[i] = wno_normal_next(&d);
}

void wno_generate_tile2d(int n, uint32_t seed, float *out)
{
    const size_t total = (size_t)n * (size_t)n;
    float *r = (float *)malloc(sizeof(float) * total);
    gaussian_fill(seed, total, r); /* WaveletNoise.cpp:74-77 */
    wno_filter_tile2d(n, r, out);
    free(r);
}

void wno_generate_tile3d(int n, uint32_t seed, float *out)
{
    const size_t total = (size_t)n * (size_t)n * (size_t)n;
    float *r = (float *)malloc(sizeof(float) * total);
    gaussian_fill(seed, total, r); /* WaveletNoise.cpp:146-147 */
    wno_filter_tile3d(n, r, out);
    free(r);
}

/* ============================================================================================
 * Wavelet evaluation
 * ========================================================================================== */

/* Quadratic B-spline weights around p (WaveletNoise.cpp:122-128 / :194-200). */
static void bspline_axis(float p, int *mid, float w[3])
{
    const int m = (int)ceilf(p - 0.5f);
    const float t = (float)m - (p - 0.5f);
    w[0] = t * t / 2.0f;
    w[2] = (1.0f - t) * (1.0f - t) / 2.0f;
    w[1] = 1.0f - w[0] - w[2];
    *mid = m;
}

float wno_evaluate2d(const float *coef, size_t count, const float p[2])
{
    if (coef == NULL || count == 0) return 0.0f;             /* :112 */
    const int n = (int)round(sqrt((double)count));           /* :113 */
    if (n == 0) return 0.0f;
    int mid[2];
    float w[2][3];
    for (int a = 0; a < 2; ++a) bspline_axis(p[a], &mid[a], w[a]);
    float result = 0.0f;
    for (int fy = -1; fy <= 1; ++fy)
        for (int fx = -1; fx <= 1; ++fx) {
            const float weight = w[0][fx + 1] * w[1][fy + 1];
            const int idx = wno_mod(mid[0] + fx, n) + wno_mod(mid[1] + fy, n) * n;
            result += weight * coef[idx];
        }
    return result;
}

float wno_evaluate3d(const float *coef, size_t count, const float p[3])
{
    if (coef == NULL || count == 0) return 0.0f;             /* :186 */
    const int n = (int)round(cbrt((double)count));           /* :187 */
    if (n == 0) return 0.0f;
    int mid[3];
    float w[3][3];
    for (int a = 0; a < 3; ++a) bspline_axis(p[a], &mid[a], w[a]);
    float result = 0.0f;
    for (int fz = -1; fz <= 1; ++fz)
        for (int fy = -1; fy <= 1; ++fy)
            for (int fx = -1; fx <= 1; ++fx) {
                const float weight = w[0][fx + 1] * w[1][fy + 1] * w[2][fz + 1];
                const int idx = wno_mod(mid[0] + fx, n) + wno_mod(mid[1] + fy, n) * n +
                                wno_mod(mid[2] + fz, n) * n * n;
                result += weight * coef[idx];
            }
    return result;
}

float wno_evaluate3d_projected(const float *coef, size_t count, const float p[3],
                               const float normal[3])
{
    if (coef == NULL || count == 0) return 0.0f;             /* :219 */
    const int n = (int)round(cbrt((double)count));           /* :220 */
    if (n == 0) return 0.0f;

    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {                            /* :228-232 */
        const float support =
            3.0f * fabsf(normal[a]) + 3.0f * sqrtf((1.0f - normal[a] * normal[a]) / 2.0f);
        lo[a] = (int)ceilf(p[a] - support);
        hi[a] = (int)floorf(p[a] + support);
    }

    float result = 0.0f;
    int c[3];
    for (c[2] = lo[2]; c[2] <= hi[2]; ++c[2])
        for (c[1] = lo[1]; c[1] <= hi[1]; ++c[1])
            for (c[0] = lo[0]; c[0] <= hi[0]; ++c[0]) {
                float dot = 0.0f;                            /* :239-240 */
                for (int a = 0; a < 3; ++a) dot += normal[a] * (p[a] - (float)c[a]);
                float weight = 1.0f;                         /* :243-255 */
                for (int a = 0; a < 3; ++a) {
                    const float t = ((float)c[a] + normal[a] * dot / 2.0f) - (p[a] - 1.5f);
                    if (t <= 0.0f || t >= 3.0f) {
                        weight = 0.0f;
                        break;
                    }
                    const float t1 = t - 1.0f, t2 = 2.0f - t, t3 = 3.0f - t;
                    if (t < 1.0f)
                        weight *= (t * t / 2.0f);
                    else if (t < 2.0f)
                        weight *= (1.0f - (t1 * t1 + t2 * t2) / 2.0f);
                    else
                        weight *= (t3 * t3 / 2.0f);
                }
                if ((double)weight > 1e-6) {                 /* :257 (float vs double literal) */
                    const int idx =
                        wno_mod(c[0], n) + wno_mod(c[1], n) * n + wno_mod(c[2], n) * n * n;
                    result += weight * coef[idx];
                }
            }
    return result;
}

float wno_multiband3d(const float *coef, size_t count, const float p[3], float s, int first_band,
                      int nbands, const float *w, float var_per_band)
{
    float result = 0.0f, variance = 0.0f;
    for (int b = 0; b < nbands && s + (float)first_band + (float)b < 0.0f; ++b) {
        float q[3];
        /* q[i] = 2*p[i]*pow(2, firstBand+b): the power of two is exact, so float products. */
        const float band_scale = (float)ldexp(1.0, first_band + b);
        for (int a = 0; a < 3; ++a) q[a] = 2.0f * p[a] * band_scale;
        result += w[b] * wno_evaluate3d(coef, count, q);
    }
    for (int b = 0; b < nbands; ++b) variance += w[b] * w[b];
    if (variance != 0.0f) result /= sqrtf(variance * var_per_band);
    return result;
}

/* Cook & DeRose Appendix 2, the normal != NULL branch: every band is WProjectedNoise =
 * evaluate3DProjected (WaveletNoise.cpp:218-265); the paper normalises with 0.296, the constant the
 * reference also uses for its projected grids (experient/main.cpp:72).  Absent from the reference:
 * pinned only as this composition of the pinned evaluate3DProjected. */
float wno_multiband3d_projected(const float *coef, size_t count, const float p[3], const float normal[3],
                                float s, int first_band, int nbands, const float *w, float var_per_band)
{
    float result = 0.0f, variance = 0.0f;
    for (int b = 0; b < nbands && s + (float)first_band + (float)b < 0.0f; ++b) {
        float q[3];
        const float band_scale = (float)ldexp(1.0, first_band + b);
        for (int a = 0; a < 3; ++a) q[a] = 2.0f * p[a] * band_scale;
        result += w[b] * wno_evaluate3d_projected(coef, count, q, normal);
    }
    for (int b = 0; b < nbands; ++b) variance += w[b] * w[b];
    if (variance != 0.0f) result /= sqrtf(variance * var_per_band);
    return result;
}

/* ============================================================================================
 * Perlin (all fp64, perlin.h:18-31, 42-62)
 * ========================================================================================== */

static double pfade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
static double plerp(double t, double a, double b) { return a + t * (b - a); }
static double pgrad(int hash, double x, double y, double z)
{
    const int h = hash & 15;
    const double u = h < 8 ? x : y;
    const double v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

double wno_perlin_noise(const int p[512], double x, double y, double z)
{
    const int X = (int)floor(x) & 255;
    const int Y = (int)floor(y) & 255;
    const int Z = (int)floor(z) & 255;
    x -= floor(x);
    y -= floor(y);
    z -= floor(z);
    const double u = pfade(x), v = pfade(y), w = pfade(z);
    const int A = p[X] + Y, AA = p[A] + Z, AB = p[A + 1] + Z;
    const int B = p[X + 1] + Y, BA = p[B] + Z, BB = p[B + 1] + Z;
    const double x00 = plerp(u, pgrad(p[AA], x, y, z), pgrad(p[BA], x - 1, y, z));
    const double x10 = plerp(u, pgrad(p[AB], x, y - 1, z), pgrad(p[BB], x - 1, y - 1, z));
    const double x01 = plerp(u, pgrad(p[AA + 1], x, y, z - 1), pgrad(p[BA + 1], x - 1, y, z - 1));
    const double x11 =
        plerp(u, pgrad(p[AB + 1], x, y - 1, z - 1), pgrad(p[BB + 1], x - 1, y - 1, z - 1));
    return plerp(w, plerp(v, x00, x10), plerp(v, x01, x11));
}

double wno_perlin_fractal(const int p[512], float x, float y, float z)
{
    double result = 0.0, amplitude = 1.0, frequency = 1.0, max_value = 0.0;
    for (int i = 0; i < 6; ++i) {
        result += wno_perlin_noise(p, x * frequency, y * frequency, z * frequency) * amplitude;
        max_value += amplitude;
        amplitude *= 0.5;
        frequency *= 2.0;
    }
    return result / max_value;
}

double wno_perlin_turb(const int p[512], float x, float y, float z, int depth)
{
    double accum = 0.0, weight = 1.0;
    float tx = x, ty = y, tz = z;
    for (int i = 0; i < depth; ++i) {
        accum += weight * wno_perlin_noise(p, (double)tx, (double)ty, (double)tz);
        weight *= 0.5;
        tx *= 2.0f;
        ty *= 2.0f;
        tz *= 2.0f;
    }
    return fabs(accum);
}

/* ============================================================================================
 * Texture adaptor
 * ========================================================================================== */

float wno_noise_texture_value(const int perm[512], double scale, int octave, const float p[3])
{
    const float octave_scale = (float)pow(2.0, (double)octave);       /* texture.h:38 */
    const float fs = (float)scale;                                    /* vec3*float, vec3.h:82-84 */
    const float sx = (fs * p[0]) * octave_scale;                      /* texture.h:39 */
    const float sy = (fs * p[1]) * octave_scale;
    const float sz = (fs * p[2]) * octave_scale;
    double v = wno_perlin_noise(perm, (double)sx, (double)sy, (double)sz); /* :40 */
    v = 0.5 * (1.0 + v);                                              /* :41 */
    return (float)v;                                                  /* color(float...) :42 */
}

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (hi < v ? hi : v); }

float wno_wavelet_texture_value(const float *coef, size_t count, int use_3d, double scale,
                                int octave, const float p[3])
{
    double v;
    const float octave_scale = (float)pow(2.0, (double)octave);       /* texture.h:77 / :92 */
    if (use_3d && coef != NULL && count != 0) {
        float pos[3];
        for (int a = 0; a < 3; ++a) pos[a] = (float)((double)p[a] * scale); /* :71-75 */
        for (int a = 0; a < 3; ++a) pos[a] *= octave_scale * 2.0f;         /* :78-80 */
        v = (double)wno_evaluate3d(coef, count, pos);                       /* :82 */
        const float inv_stddev = 1.0f / sqrtf(0.18402f);                    /* :84 */
        v *= (double)inv_stddev;                                            /* :85 */
    } else if (!use_3d && coef != NULL && count != 0) {
        float pos[2];
        for (int a = 0; a < 2; ++a) pos[a] = (float)((double)p[a] * scale); /* :87-90 */
        for (int a = 0; a < 2; ++a) pos[a] *= octave_scale * 2.0f;         /* :93-94 */
        v = (double)wno_evaluate2d(coef, count, pos);                       /* :96 */
        const float inv_stddev = 1.0f / sqrtf(0.19686f);                    /* :98 */
        v *= (double)inv_stddev;
    } else {
        v = 0.0;                                                            /* :101 */
    }
    v = 0.5 * (1.0 + clampd(v / 4.0, -1.0, 1.0));                           /* :104 */
    return (float)v;
}

/* ============================================================================================
 * Dense grids
 * ========================================================================================== */

static float lattice_coord(int i, int den) { return ((float)i / (float)den) * 4.0f; }

void wno_grid_wavelet2d(const float *coef, size_t count, int image, int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    const float inv_stddev = 1.0f / sqrtf(0.19686f);
    for (int y = 0; y < image; ++y)
        for (int x = 0; x < image; ++x) {
            float p[2] = {lattice_coord(x, image) * octave_scale,
                          lattice_coord(y, image) * octave_scale};
            p[0] *= 2.0f;
            p[1] *= 2.0f;
            out[(size_t)y * image + x] = wno_evaluate2d(coef, count, p) * inv_stddev;
        }
}

void wno_grid_wavelet3d_sliced(const float *coef, size_t count, int image, int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    const float inv_stddev = 1.0f / sqrtf(0.18402f);
    for (int y = 0; y < image; ++y)
        for (int x = 0; x < image; ++x) {
            float p[3] = {lattice_coord(x, image) * octave_scale,
                          lattice_coord(y, image) * octave_scale, 1.0f};
            p[0] *= 2.0f;
            p[1] *= 2.0f;
            p[2] *= 2.0f;
            out[(size_t)y * image + x] = wno_evaluate3d(coef, count, p) * inv_stddev;
        }
}

void wno_grid_wavelet3d_projected(const float *coef, size_t count, int image, int octave,
                                  float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    const float normal[3] = {0.0f, 0.0f, 1.0f};
    const float inv_stddev = 1.0f / sqrtf(0.296f);
    for (int y = 0; y < image; ++y)
        for (int x = 0; x < image; ++x) {
            float p[3] = {lattice_coord(x, image) * octave_scale,
                          lattice_coord(y, image) * octave_scale, 1.0f};
            p[0] *= 2.0f;
            p[1] *= 2.0f;
            p[2] *= 2.0f;
            out[(size_t)y * image + x] =
                wno_evaluate3d_projected(coef, count, p, normal) * inv_stddev;
        }
}

void wno_grid_perlin2d(const int perm[512], int image, int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    for (int y = 0; y < image; ++y)
        for (int x = 0; x < image; ++x) {
            const float u = lattice_coord(x, image), v = lattice_coord(y, image);
            out[(size_t)y * image + x] =
                (float)wno_perlin_noise(perm, (double)(u * octave_scale),
                                        (double)(v * octave_scale), 0.0);
        }
}

void wno_grid_perlin3d_sliced(const int perm[512], int image, int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    for (int y = 0; y < image; ++y)
        for (int x = 0; x < image; ++x) {
            const float u = lattice_coord(x, image), v = lattice_coord(y, image);
            out[(size_t)y * image + x] =
                (float)wno_perlin_noise(perm, (double)(u * octave_scale),
                                        (double)(v * octave_scale),
                                        (double)(1.0f * octave_scale));
        }
}

void wno_grid_wavelet3d_volume(const float *coef, size_t count, int den, int nx, int ny, int z0,
                               int z1, int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    const float inv_stddev = 1.0f / sqrtf(0.18402f);
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                float p[3] = {lattice_coord(x, den) * octave_scale,
                              lattice_coord(y, den) * octave_scale,
                              lattice_coord(z, den) * octave_scale};
                p[0] *= 2.0f;
                p[1] *= 2.0f;
                p[2] *= 2.0f;
                out[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)(z - z0))] =
                    wno_evaluate3d(coef, count, p) * inv_stddev;
            }
}

void wno_grid_multiband3d_volume(const float *coef, size_t count, int den, int nx, int ny, int z0,
                                 int z1, float s, int first_band, int nbands, const float *w,
                                 float var_per_band, float *out)
{
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const float p[3] = {lattice_coord(x, den), lattice_coord(y, den),
                                    lattice_coord(z, den)};
                out[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)(z - z0))] =
                    wno_multiband3d(coef, count, p, s, first_band, nbands, w, var_per_band);
            }
}

void wno_grid_perlin_volume(const int perm[512], int den, int nx, int ny, int z0, int z1,
                            int octave, float *out)
{
    const float octave_scale = (float)pow(2.0, (double)octave);
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x)
                out[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)(z - z0))] =
                    (float)wno_perlin_noise(perm, (double)(lattice_coord(x, den) * octave_scale),
                                            (double)(lattice_coord(y, den) * octave_scale),
                                            (double)(lattice_coord(z, den) * octave_scale));
}

void wno_grid_turb_volume(const int perm[512], int den, int nx, int ny, int z0, int z1, int depth,
                          float *out)
{
    for (int z = z0; z < z1; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x)
                out[(size_t)x + (size_t)nx * ((size_t)y + (size_t)ny * (size_t)(z - z0))] =
                    (float)wno_perlin_turb(perm, lattice_coord(x, den), lattice_coord(y, den),
                                           lattice_coord(z, den), depth);
}

uint64_t wno_fnv1a64(const void *data, size_t nbytes)
{
    const unsigned char *b = (const unsigned char *)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < nbytes; ++i) {
        h ^= (uint64_t)b[i];
        h *= 0x100000001b3ull;
    }
    return h;
}
