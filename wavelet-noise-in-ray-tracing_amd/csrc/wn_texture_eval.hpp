// wn_texture_eval.hpp -- the texture adaptors' per-point arithmetic (texture.h:37-43, 67-107), shared by
// the batched texture kernels (wn_wavelet_points.hip, wn_perlin.hip) and the resident scalar kernel
// (wn_mailbox.hip).  Reference operation order, unfused (the library is built with -ffp-contract=off).
#pragma once

#include "wn_device_eval.hpp"

namespace wn {

// wavelet_texture::value, texture.h:67-107.  `A` carries coef, n, nmask, mode (3: evaluate3D branch,
// 2: evaluate2D branch, 0: no tile -> texture.h:100-102), scale (double), octave_mul (octave_scale * 2.0f,
// :77-80) and inv_stddev (1/sqrt(0.18402f) or 1/sqrt(0.19686f), :84,98).
template <bool PADDED, typename A>
__device__ __forceinline__ float wavelet_texture_value(const A &a, float px, float py, float pz)
{
    double v;
    if (a.mode == 3) {
        float pos[3] = {(float)((double)px * a.scale), (float)((double)py * a.scale),
                        (float)((double)pz * a.scale)};
        pos[0] *= a.octave_mul;
        pos[1] *= a.octave_mul;
        pos[2] *= a.octave_mul;
        v = (double)eval3d_exact<PADDED>(a.coef, a.n, a.nmask, pos[0], pos[1], pos[2]);
        v *= (double)a.inv_stddev;
    } else if (a.mode == 2) {
        float pos[2] = {(float)((double)px * a.scale), (float)((double)py * a.scale)};
        pos[0] *= a.octave_mul;
        pos[1] *= a.octave_mul;
        v = (double)eval2d_exact(a.coef, a.n, a.nmask, pos[0], pos[1]);
        v *= (double)a.inv_stddev;
    } else {
        v = 0.0;
    }
    const double q = v / 4.0;
    const double c = (q < -1.0) ? -1.0 : ((1.0 < q) ? 1.0 : q); // std::clamp
    return (float)(0.5 * (1.0 + c));                              // texture.h:104-106
}

// wavelet_texture::value on a 3-D padded tile with a two-row slab in LDS (eval3d_exact_rowslab): the same arithmetic.
template <typename A>
__device__ __forceinline__ float wavelet_texture_value_rowslab(const A &a, float px, float py, float pz, const float *slab, int ry,
                                                               const float *third, int third_planes)
{
    float pos[3] = {(float)((double)px * a.scale), (float)((double)py * a.scale), (float)((double)pz * a.scale)};
    pos[0] *= a.octave_mul;
    pos[1] *= a.octave_mul;
    pos[2] *= a.octave_mul;
    double v = (double)eval3d_exact_rowslab(a.coef, a.n, a.nmask, pos[0], pos[1], pos[2], slab, ry, third, third_planes);
    v *= (double)a.inv_stddev;
    const double q = v / 4.0;
    const double c = (q < -1.0) ? -1.0 : ((1.0 < q) ? 1.0 : q);
    return (float)(0.5 * (1.0 + c));
}

// noise_texture::value, texture.h:37-43: scaled_p = p * scale * octave_scale in float (vec3 * float,
// vec3.h:82-84), noise in fp64, 0.5 * (1 + n).
template <typename Table>
__device__ __forceinline__ float noise_texture_value(const Table perm, float fscale, float octave_scale, float px,
                                                     float py, float pz)
{
    const float sx = (fscale * px) * octave_scale;
    const float sy = (fscale * py) * octave_scale;
    const float sz = (fscale * pz) * octave_scale;
    double v = perlin_exact(perm, (double)sx, (double)sy, (double)sz);
    v = 0.5 * (1.0 + v);
    return (float)v;
}

} // namespace wn
