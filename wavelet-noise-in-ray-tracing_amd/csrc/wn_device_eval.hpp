// wn_device_eval.hpp -- per-sample device evaluators in the reference's exact operation order.
// Built with -ffp-contract=off: every product and sum below rounds once, as on the reference's
// baseline x86-64 build, so these return the same bits as the CPU classes.
#pragma once

#include "wn_internal.hpp"

namespace wn {

// WaveletNoise::evaluate2D, WaveletNoise.cpp:111-140.
__device__ __forceinline__ float eval2d_exact(const float *coef, int n, int nmask, float px,
                                              float py)
{
    if (n == 0) return 0.0f; // :112-114
    int mx, my;
    float wx[3], wy[3];
    bspline(px, mx, wx[0], wx[1], wx[2]);
    bspline(py, my, wy[0], wy[1], wy[2]);
    int cx[3], cy[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        cx[f] = dmod(mx + f - 1, n, nmask);
        cy[f] = dmod(my + f - 1, n, nmask) * n;
    }
    float result = 0.0f;
#pragma unroll
    for (int fy = 0; fy < 3; ++fy)
#pragma unroll
        for (int fx = 0; fx < 3; ++fx) {
            const float weight = wx[fx] * wy[fy];
            result += weight * coef[cx[fx] + cy[fy]];
        }
    return result;
}

// WaveletNoise::evaluate3D, WaveletNoise.cpp:185-215 (f2 outer, f0 inner; weight=(w0*w1)*w2).
// PADDED: `coef` is wn_tile::dev_padded (row stride n+2 with two wrap-around columns), so the
// three x taps of every (y,z) row are adjacent and fetched with one 12-byte load; the values, the
// arithmetic and its order are those of the linear layout.
// POW2: the tile size is a power of two and the wrap a mask.  The general modulo behind a run-time test per index splits the
// nine loads into basic blocks (a branch per wrap); eval3d_exact tests once and calls the form that has none.
template <bool PADDED, bool POW2>
__device__ __forceinline__ float eval3d_exact_impl(const float *coef, int n, int nmask, float px,
                                                   float py, float pz)
{
    int mx, my, mz;
    float wx[3], wy[3], wz[3];
    bspline(px, mx, wx[0], wx[1], wx[2]);
    bspline(py, my, wy[0], wy[1], wy[2]);
    bspline(pz, mz, wz[0], wz[1], wz[2]);
    const int stride = PADDED ? n + 2 : n;
    int cx[3], cy[3], cz[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        cx[f] = POW2 ? ((mx + f - 1) & nmask) : dmod(mx + f - 1, n, -1);
        cy[f] = (POW2 ? ((my + f - 1) & nmask) : dmod(my + f - 1, n, -1)) * stride;
        cz[f] = (POW2 ? ((mz + f - 1) & nmask) : dmod(mz + f - 1, n, -1)) * stride * n;
    }
    float result = 0.0f;
#pragma unroll
    for (int fz = 0; fz < 3; ++fz)
#pragma unroll
        for (int fy = 0; fy < 3; ++fy) {
            float c[3];
            if (PADDED) {
                __builtin_memcpy(c, coef + cx[0] + cy[fy] + cz[fz], sizeof(c)); // global_load_dwordx3
            } else {
#pragma unroll
                for (int fx = 0; fx < 3; ++fx) c[fx] = coef[cx[fx] + cy[fy] + cz[fz]];
            }
#pragma unroll
            for (int fx = 0; fx < 3; ++fx) {
                const float weight = wx[fx] * wy[fy] * wz[fz];
                result += weight * c[fx];
            }
        }
    return result;
}

template <bool PADDED = false>
__device__ __forceinline__ float eval3d_exact(const float *coef, int n, int nmask, float px,
                                              float py, float pz)
{
    if (n == 0) return 0.0f; // :186-188
    return nmask >= 0 ? eval3d_exact_impl<PADDED, true>(coef, n, nmask, px, py, pz)
                      : eval3d_exact_impl<PADDED, false>(coef, n, nmask, px, py, pz);
}

// evaluate3D on the padded tile with two of its y rows held in LDS: `slab` = [z][2][n + 2], the rows (ry - 1) mod n and ry of
// every z plane.  A point whose middle y row is ry takes its first two row triples (fy = 0, 1 of every fz) from the slab and
// the third from memory -- 3 scattered 12-byte gathers instead of 9; any other point takes all nine from memory.  Same
// values, products, sums and order as eval3d_exact<true>: the same bits.
// `third` = [z][n + 2]: row (ry + 1) mod n of the planes z < third_planes (0: none), also in LDS.
__device__ __forceinline__ float eval3d_exact_rowslab(const float *coef, int n, int nmask, float px, float py, float pz,
                                                      const float *slab, int ry, const float *third, int third_planes)
{
    if (n == 0) return 0.0f;
    int mx, my, mz;
    float wx[3], wy[3], wz[3];
    bspline(px, mx, wx[0], wx[1], wx[2]);
    bspline(py, my, wy[0], wy[1], wy[2]);
    bspline(pz, mz, wz[0], wz[1], wz[2]);
    const int stride = n + 2;
    // (the tile size is a power of two here: the wrap is a mask, and no branch on the kind of modulo splits the loads below
    // into basic blocks)
    const int cx0 = (mx - 1) & nmask;
    int ry3[3], rz3[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        ry3[f] = (my + f - 1) & nmask;
        rz3[f] = (mz + f - 1) & nmask;
    }
    const bool in_slab = ry3[1] == ry; // then ry3[0] is the slab's first row
    float c[3][3][3];
    // every load is requested before the first product; ONE divergent branch (the lanes off the slab's rows)
#pragma unroll
    for (int fz = 0; fz < 3; ++fz) {
        if (in_slab && rz3[fz] < third_planes) __builtin_memcpy(c[fz][2], third + rz3[fz] * stride + cx0, 12);
        else __builtin_memcpy(c[fz][2], coef + cx0 + ry3[2] * stride + rz3[fz] * stride * n, 12);
    }
    if (in_slab) {
#pragma unroll
        for (int fz = 0; fz < 3; ++fz)
#pragma unroll
            for (int fy = 0; fy < 2; ++fy) __builtin_memcpy(c[fz][fy], slab + (rz3[fz] * 2 + fy) * stride + cx0, 12);
    } else {
#pragma unroll
        for (int fz = 0; fz < 3; ++fz)
#pragma unroll
            for (int fy = 0; fy < 2; ++fy) __builtin_memcpy(c[fz][fy], coef + cx0 + ry3[fy] * stride + rz3[fz] * stride * n, 12);
    }
    float result = 0.0f;
#pragma unroll
    for (int fz = 0; fz < 3; ++fz)
#pragma unroll
        for (int fy = 0; fy < 3; ++fy)
#pragma unroll
            for (int fx = 0; fx < 3; ++fx) {
                const float weight = wx[fx] * wy[fy] * wz[fz];
                result += weight * c[fz][fy][fx];
            }
    return result;
}

// WaveletNoise::evaluate3DProjected, WaveletNoise.cpp:218-265: data-dependent support box,
// `break` on the first axis outside the basis support, contributions <= 1e-6 skipped.
__device__ __forceinline__ float projected_exact(const float *coef, int n, int nmask,
                                                 const float p[3], const float nrm[3])
{
    if (n == 0) return 0.0f; // :219-221
    int lo[3], hi[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float support =
            3.0f * fabsf(nrm[i]) + 3.0f * sqrtf((1.0f - nrm[i] * nrm[i]) / 2.0f);
        lo[i] = (int)ceilf(p[i] - support);
        hi[i] = (int)floorf(p[i] + support);
    }
    float result = 0.0f;
    for (int c2 = lo[2]; c2 <= hi[2]; ++c2)
        for (int c1 = lo[1]; c1 <= hi[1]; ++c1)
            for (int c0 = lo[0]; c0 <= hi[0]; ++c0) {
                const float cf[3] = {(float)c0, (float)c1, (float)c2};
                float dot = 0.0f;
#pragma unroll
                for (int i = 0; i < 3; ++i) dot += nrm[i] * (p[i] - cf[i]);
                float weight = 1.0f;
                bool outside = false;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (!outside) {
                        const float t = (cf[i] + nrm[i] * dot / 2.0f) - (p[i] - 1.5f);
                        if (t <= 0.0f || t >= 3.0f) {
                            weight = 0.0f;
                            outside = true;
                        } else {
                            const float t1 = t - 1.0f, t2 = 2.0f - t, t3 = 3.0f - t;
                            if (t < 1.0f) weight *= (t * t / 2.0f);
                            else if (t < 2.0f) weight *= (1.0f - (t1 * t1 + t2 * t2) / 2.0f);
                            else weight *= (t3 * t3 / 2.0f);
                        }
                    }
                }
                if ((double)weight > 1e-6) { // :257 compares against a double literal
                    const int idx =
                        dmod(c0, n, nmask) + dmod(c1, n, nmask) * n + dmod(c2, n, nmask) * n * n;
                    result += weight * coef[idx];
                }
            }
    return result;
}

// ---- Perlin improved noise, fp64 (perlin.h:18-31, 42-62) ------------------------------------------
__device__ __forceinline__ double pfade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
__device__ __forceinline__ double plerp(double t, double a, double b) { return a + t * (b - a); }
__device__ __forceinline__ double pgrad(int hash, double x, double y, double z)
{
    const int h = hash & 15;
    const double u = h < 8 ? x : y;
    const double v = h < 4 ? y : ((h == 12 || h == 14) ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

// `perm` is the 512-entry table as bytes (values 0..255; every index the algorithm forms is
// <= 511, perlin.h:55-61), in LDS or global memory.
template <typename Table>
__device__ __forceinline__ double perlin_exact(const Table perm, double x, double y, double z)
{
    const double fx = floor(x), fy = floor(y), fz = floor(z);
    const int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
    x -= fx;
    y -= fy;
    z -= fz;
    const double u = pfade(x), v = pfade(y), w = pfade(z);
    const int A = perm[X] + Y, AA = perm[A] + Z, AB = perm[A + 1] + Z;
    const int B = perm[X + 1] + Y, BA = perm[B] + Z, BB = perm[B + 1] + Z;
    const double x00 = plerp(u, pgrad(perm[AA], x, y, z), pgrad(perm[BA], x - 1, y, z));
    const double x10 = plerp(u, pgrad(perm[AB], x, y - 1, z), pgrad(perm[BB], x - 1, y - 1, z));
    const double x01 =
        plerp(u, pgrad(perm[AA + 1], x, y, z - 1), pgrad(perm[BA + 1], x - 1, y, z - 1));
    const double x11 = plerp(u, pgrad(perm[AB + 1], x, y - 1, z - 1),
                             pgrad(perm[BB + 1], x - 1, y - 1, z - 1));
    return plerp(w, plerp(v, x00, x10), plerp(v, x01, x11));
}

// RTOW turb on a float vec3 (absent from the reference): weight halves, point doubles in float.
template <typename Table>
__device__ __forceinline__ double perlin_turb(const Table perm, float x, float y, float z,
                                              int depth)
{
    double accum = 0.0, weight = 1.0;
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_exact(perm, (double)x, (double)y, (double)z);
        weight *= 0.5;
        x *= 2.0f;
        y *= 2.0f;
        z *= 2.0f;
    }
    return fabs(accum);
}

// perlin::fractal_noise, perlin.h:75-90 (float point times double frequency).
template <typename Table>
__device__ __forceinline__ double perlin_fractal(const Table perm, float x, float y, float z)
{
    double result = 0.0, amplitude = 1.0, frequency = 1.0, max_value = 0.0;
    for (int i = 0; i < 6; ++i) {
        result += perlin_exact(perm, x * frequency, y * frequency, z * frequency) * amplitude;
        max_value += amplitude;
        amplitude *= 0.5;
        frequency *= 2.0;
    }
    return result / max_value;
}

// Load the 512-byte permutation table into LDS (blockDim >= 128 lanes, 4 bytes each).
__device__ __forceinline__ void load_perm_lds(uint8_t *lds_perm, const uint8_t *gperm)
{
    for (int i = threadIdx.x; i < 128; i += blockDim.x)
        reinterpret_cast<uint32_t *>(lds_perm)[i] = reinterpret_cast<const uint32_t *>(gperm)[i];
    __syncthreads();
}

} // namespace wn
