// scalar_eval.cpp -- the reference's scalar members on the host, one sample per call (see scalar_eval.h).
// Built with -ffp-contract=off: every product and sum rounds once, as on the reference's baseline x86-64 build.
#include "scalar_eval.h"

#include <cmath>
#include <cstdlib>

namespace {

inline int wrap(int x, int n) // WaveletNoise::Mod, WaveletNoise.cpp:31-34
{
    const int m = x % n;
    return m < 0 ? m + n : m;
}

// the three quadratic B-spline weights around p and the index of the middle one (WaveletNoise.cpp:194-200)
inline void bspline(float p, int &mid, float w[3])
{
    const float pm = p - 0.5f;
    const float cm = std::ceil(pm);
    mid = (int)cm;
    const float t = cm - pm;
    w[0] = t * t / 2.0f;
    w[2] = (1.0f - t) * (1.0f - t) / 2.0f;
    w[1] = 1.0f - w[0] - w[2];
}

inline double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); } // perlin.h:18-20
inline double lerp(double t, double a, double b) { return a + t * (b - a); }  // perlin.h:22-24
inline double grad(int hash, double x, double y, double z)                    // perlin.h:26-31
{
    const int h = hash & 15;
    const double u = h < 8 ? x : y;
    const double v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

} // namespace

extern "C" {

float wnhost_eval2d(const float *coef, int n, const float p[2])
{
    if (!coef || n <= 0) return 0.0f; // :112-114
    int mid[2];
    float w[2][3];
    bspline(p[0], mid[0], w[0]);
    bspline(p[1], mid[1], w[1]);
    float result = 0.0f;
    for (int fy = -1; fy <= 1; ++fy)
        for (int fx = -1; fx <= 1; ++fx) {
            const float weight = w[0][fx + 1] * w[1][fy + 1];
            result += weight * coef[wrap(mid[0] + fx, n) + wrap(mid[1] + fy, n) * n];
        }
    return result;
}

float wnhost_eval3d(const float *coef, int n, const float p[3])
{
    if (!coef || n <= 0) return 0.0f; // :186-188
    int mid[3];
    float w[3][3];
    for (int i = 0; i < 3; ++i) bspline(p[i], mid[i], w[i]);
    float result = 0.0f;
    for (int f2 = -1; f2 <= 1; ++f2) // :202-213: z outermost, x innermost, weight = (wx * wy) * wz
        for (int f1 = -1; f1 <= 1; ++f1)
            for (int f0 = -1; f0 <= 1; ++f0) {
                const float weight = w[0][f0 + 1] * w[1][f1 + 1] * w[2][f2 + 1];
                const int idx = wrap(mid[0] + f0, n) + wrap(mid[1] + f1, n) * n + wrap(mid[2] + f2, n) * n * n;
                result += weight * coef[idx];
            }
    return result;
}

float wnhost_eval3d_projected(const float *coef, int n, const float p[3], const float nrm[3])
{
    if (!coef || n <= 0) return 0.0f; // :219-221
    int lo[3], hi[3];
    for (int i = 0; i < 3; ++i) { // the support box of the projected basis, :226-231
        const float support = 3.0f * std::fabs(nrm[i]) + 3.0f * std::sqrt((1.0f - nrm[i] * nrm[i]) / 2.0f);
        lo[i] = (int)std::ceil(p[i] - support);
        hi[i] = (int)std::floor(p[i] + support);
    }
    float result = 0.0f;
    for (int c2 = lo[2]; c2 <= hi[2]; ++c2)
        for (int c1 = lo[1]; c1 <= hi[1]; ++c1)
            for (int c0 = lo[0]; c0 <= hi[0]; ++c0) {
                const float cf[3] = {(float)c0, (float)c1, (float)c2};
                float dot = 0.0f;
                for (int i = 0; i < 3; ++i) dot += nrm[i] * (p[i] - cf[i]);
                float weight = 1.0f;
                for (int i = 0; i < 3; ++i) { // the first axis outside the support ends the product (`break`, :243-254)
                    const float t = (cf[i] + nrm[i] * dot / 2.0f) - (p[i] - 1.5f);
                    if (t <= 0.0f || t >= 3.0f) {
                        weight = 0.0f;
                        break;
                    }
                    const float t1 = t - 1.0f, t2 = 2.0f - t, t3 = 3.0f - t;
                    if (t < 1.0f) weight *= (t * t / 2.0f);
                    else if (t < 2.0f) weight *= (1.0f - (t1 * t1 + t2 * t2) / 2.0f);
                    else weight *= (t3 * t3 / 2.0f);
                }
                if ((double)weight > 1e-6) // :257 compares with a double literal
                    result += weight * coef[wrap(c0, n) + wrap(c1, n) * n + wrap(c2, n) * n * n];
            }
    return result;
}

double wnhost_perlin(const int *perm, double x, double y, double z)
{
    const double fx = std::floor(x), fy = std::floor(y), fz = std::floor(z);
    const int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
    x -= fx;
    y -= fy;
    z -= fz;
    const double u = fade(x), v = fade(y), w = fade(z);
    const int A = perm[X] + Y, AA = perm[A] + Z, AB = perm[A + 1] + Z;
    const int B = perm[X + 1] + Y, BA = perm[B] + Z, BB = perm[B + 1] + Z;
    return lerp(w,
                lerp(v, lerp(u, grad(perm[AA], x, y, z), grad(perm[BA], x - 1, y, z)),
                     lerp(u, grad(perm[AB], x, y - 1, z), grad(perm[BB], x - 1, y - 1, z))),
                lerp(v, lerp(u, grad(perm[AA + 1], x, y, z - 1), grad(perm[BA + 1], x - 1, y, z - 1)),
                     lerp(u, grad(perm[AB + 1], x, y - 1, z - 1), grad(perm[BB + 1], x - 1, y - 1, z - 1))));
}

double wnhost_perlin_fractal(const int *perm, const float q[3])
{
    double result = 0.0, amplitude = 1.0, frequency = 1.0, max_value = 0.0;
    for (int i = 0; i < 6; ++i) { // float point times double frequency, perlin.h:80-86
        result += wnhost_perlin(perm, q[0] * frequency, q[1] * frequency, q[2] * frequency) * amplitude;
        max_value += amplitude;
        amplitude *= 0.5;
        frequency *= 2.0;
    }
    return result / max_value;
}

double wnhost_perlin_turb(const int *perm, const float q[3], int depth)
{
    double accum = 0.0, weight = 1.0;
    float x = q[0], y = q[1], z = q[2]; // the point doubles in float (vec3 * float)
    for (int i = 0; i < depth; ++i) {
        accum += weight * wnhost_perlin(perm, (double)x, (double)y, (double)z);
        weight *= 0.5;
        x *= 2.0f;
        y *= 2.0f;
        z *= 2.0f;
    }
    return std::fabs(accum);
}

float wnhost_wavelet_texture_value(const float *coef, int n, int use_3d, double scale, int octave, const float xyz[3])
{
    double v;
    const float octave_scale = (float)std::pow(2.0, (double)octave); // std::pow(2.0f, int) is evaluated in double, :77
    const float mul = octave_scale * 2.0f;
    if (coef && use_3d) {
        float pos[3] = {(float)((double)xyz[0] * scale), (float)((double)xyz[1] * scale), (float)((double)xyz[2] * scale)};
        pos[0] *= mul;
        pos[1] *= mul;
        pos[2] *= mul;
        v = (double)wnhost_eval3d(coef, n, pos);
        v *= (double)(1.0f / std::sqrt(0.18402f)); // :84-85
    } else if (coef) {
        float pos[2] = {(float)((double)xyz[0] * scale), (float)((double)xyz[1] * scale)};
        pos[0] *= mul;
        pos[1] *= mul;
        v = (double)wnhost_eval2d(coef, n, pos);
        v *= (double)(1.0f / std::sqrt(0.19686f)); // :98-99
    } else {
        v = 0.0; // :100-102
    }
    const double q = v / 4.0;
    const double c = (q < -1.0) ? -1.0 : ((1.0 < q) ? 1.0 : q); // std::clamp
    return (float)(0.5 * (1.0 + c));                              // :104-106, narrowed by color
}

float wnhost_noise_texture_value(const int *perm, double scale, int octave, const float xyz[3])
{
    const float octave_scale = (float)std::pow(2.0, (double)octave);
    const float fscale = (float)scale; // vec3 * double narrows the factor (vec3.h:82-84)
    const float sx = (fscale * xyz[0]) * octave_scale, sy = (fscale * xyz[1]) * octave_scale, sz = (fscale * xyz[2]) * octave_scale;
    double v = wnhost_perlin(perm, (double)sx, (double)sy, (double)sz);
    v = 0.5 * (1.0 + v);
    return (float)v;
}

int wnhost_scalar_on_device(void)
{
    static const int on = [] {
        const char *e = std::getenv("WN_SCALAR_ON_DEVICE");
        return (e && *e && *e != '0') ? 1 : 0;
    }();
    return on;
}

} // extern "C"
