// scalar_eval.h -- host evaluators of the reference's SCALAR members only (one sample per call).
//
// "Scalar class methods stay on the CPU path".  A caller that asks for one sample and needs it before its
// next instruction (the reference's ray tracer: material.h:72 -> texture.h:37-43 / 67-107, 29.6 M calls a render) cannot
// be served by a device faster than the PCIe round trip (1.9 us measured, profiles/r03_scalar_latency.json); the
// reference's own call takes ~0.1 us.  These functions are that call: the reference's arithmetic in the reference's order
// (file:line beside each), written for this library, compiled with -ffp-contract=off into libwnoise_host.so, bit-identical
// to the reference and to the HIP kernels (tests/test_host_scalar.py, tools/scalar_api_check).
//
// They are not a fallback: nothing that takes more than one sample (points lists, textures' values(), dense grids, tile
// generation) has a host form, and the host classes still throw without a HIP device.  WN_SCALAR_ON_DEVICE=1 in the
// environment sends the scalar members through the resident scalar kernel instead (wn_scalar_*, csrc/wn_mailbox.hip).
#ifndef WN_HOST_SCALAR_EVAL_H
#define WN_HOST_SCALAR_EVAL_H

#include <stddef.h>

#if defined(__GNUC__)
#define WNHOST_API __attribute__((visibility("default")))
#else
#define WNHOST_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

// `coef`: the tile's n^2 / n^3 coefficients, x fastest (WaveletNoise::getNoiseCoefficients()); n == 0 or coef == NULL -> 0.0f
WNHOST_API float wnhost_eval2d(const float *coef, int n, const float p[2]);                                  // WaveletNoise.cpp:111-140
WNHOST_API float wnhost_eval3d(const float *coef, int n, const float p[3]);                                  // WaveletNoise.cpp:185-215
WNHOST_API float wnhost_eval3d_projected(const float *coef, int n, const float p[3], const float normal[3]); // WaveletNoise.cpp:218-265
// `perm`: the 512-entry table (perlin.h:34-39)
WNHOST_API double wnhost_perlin(const int *perm, double x, double y, double z);      // perlin.h:42-62
WNHOST_API double wnhost_perlin_fractal(const int *perm, const float q[3]);           // perlin.h:75-90
WNHOST_API double wnhost_perlin_turb(const int *perm, const float q[3], int depth);   // RTOW turb (absent from the reference)
// grey level of texture::value (texture.h); use_3d = 0: the 2-D tile and branch; coef == NULL: the no-tile grey
WNHOST_API float wnhost_wavelet_texture_value(const float *coef, int n, int use_3d, double scale, int octave,
                                              const float xyz[3]);                    // texture.h:67-107
WNHOST_API float wnhost_noise_texture_value(const int *perm, double scale, int octave, const float xyz[3]); // texture.h:37-43
// 1 when WN_SCALAR_ON_DEVICE is set (read once): the classes' scalar members then use the resident scalar kernel
WNHOST_API int wnhost_scalar_on_device(void);

#ifdef __cplusplus
}
#endif
#endif
