// noise_grid.h -- the five dense-grid generators of the reference's experient/main.cpp (:11-129),
// same names and argument order, as single batched launches through the C ABI instead of 65,536
// scalar calls each.  Output files are the reference's raw format: float32[imageSize*imageSize],
// row-major, index y*imageSize+x (experient/main.cpp:28,32-34).
#pragma once

#include <cmath>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "PerlinNoise.hpp"
#include "WaveletNoise.h"
#include "wn_host.hpp"

namespace wnhost {

inline wn_grid lattice2d(int imageSize, int octave, float post_scale, float out_scale, int flags)
{
    wn_grid g{};
    g.den = g.nx = g.ny = imageSize;
    g.z0 = 0;
    g.z1 = 1;
    g.base_range = 4.0f;                          // experient/main.cpp:13
    g.octave_scale = std::pow(2.0f, octave);      // :14
    g.post_scale = post_scale;
    g.z_mode = WN_Z_LATTICE;
    g.z_const = 0.0f;
    g.out_scale = out_scale;
    g.flags = flags;
    return g;
}

template <typename Launch>
inline void run_grid(int imageSize, const std::string &outputFile, Launch launch)
{
    const size_t count = static_cast<size_t>(imageSize) * imageSize;
    std::vector<float> image(count);
    if (count) {
        DeviceBuffer out(count * sizeof(float));
        launch(out.as<float>());
        out.download(image.data());
    }
    std::ofstream outFile(outputFile, std::ios::binary);
    outFile.write(reinterpret_cast<const char *>(image.data()), image.size() * sizeof(float));
}

} // namespace wnhost

// The reference-named generators reproduce the reference's files byte for byte: they ask for
// WN_GRID_EXACT.  (3-D sliced only: pass WN_GRID_DEFAULT to opt in to the separable brick kernel,
// within 1e-5 of the reference -- a single 256x256 plane gains nothing measurable from it.)
inline void generate2DOctaveBandNoise(int imageSize, int octave, const std::string &outputFile,
                                      WaveletNoise &noise) // experient/main.cpp:11-36
{
    wn_grid g = wnhost::lattice2d(imageSize, octave, 2.0f, 1.0f / std::sqrt(0.19686f), WN_GRID_DEFAULT);
    wnhost::run_grid(imageSize, outputFile, [&](float *out) {
        wnhost::check(wn_eval2d_grid(noise.tile(2), &g, out, nullptr), "wn_eval2d_grid");
    });
    std::cout << "Generated Wavelet 2D Octave " << octave << " noise: " << outputFile << std::endl;
}

inline void generate3DSlicedOctaveBandNoise(int imageSize, int octave, const std::string &outputFile,
                                            WaveletNoise &noise, int flags = WN_GRID_EXACT) // :38-64
{
    wn_grid g = wnhost::lattice2d(imageSize, octave, 2.0f, 1.0f / std::sqrt(0.18402f), flags);
    g.z_mode = WN_Z_CONST;
    g.z_const = 1.0f * 2.0f; // p[2] = 1.0f; p[2] *= 2.0f (:46,54)
    wnhost::run_grid(imageSize, outputFile, [&](float *out) {
        wnhost::check(wn_eval3d_grid(noise.tile(3), &g, out, nullptr), "wn_eval3d_grid");
    });
    std::cout << "Generated Wavelet 3D Sliced Octave " << octave << " noise: " << outputFile << std::endl;
}

inline void generate3DProjectedOctaveBandNoise(int imageSize, int octave, const std::string &outputFile,
                                               WaveletNoise &noise) // :66-93
{
    wn_grid g = wnhost::lattice2d(imageSize, octave, 2.0f, 1.0f / std::sqrt(0.296f), WN_GRID_DEFAULT);
    g.z_mode = WN_Z_CONST;
    g.z_const = 1.0f * 2.0f;
    const float normal[3] = {0.0f, 0.0f, 1.0f};
    wnhost::run_grid(imageSize, outputFile, [&](float *out) {
        wnhost::check(wn_eval3d_projected_grid(noise.tile(3), &g, normal, out, nullptr), "wn_eval3d_projected_grid");
    });
    std::cout << "Generated Wavelet 3D Projected Octave " << octave << " noise: " << outputFile << std::endl;
}

inline void generatePerlinNoise2D(int imageSize, int octave, const std::string &outputFile,
                                  const PerlinNoise &perlin) // :95-111
{
    wn_grid g = wnhost::lattice2d(imageSize, octave, 1.0f, 1.0f, WN_GRID_DEFAULT);
    g.z_mode = WN_Z_CONST;
    g.z_const = 0.0f; // noise(x, y) == noise(x, y, 0.0)
    wnhost::run_grid(imageSize, outputFile, [&](float *out) {
        wnhost::check(wn_perlin_grid(perlin.perm(), &g, out, nullptr), "wn_perlin_grid");
    });
    std::cout << "Generated Perlin 2D Octave " << octave << " noise: " << outputFile << std::endl;
}

inline void generatePerlinNoise3DSliced(int imageSize, int octave, const std::string &outputFile,
                                        const PerlinNoise &perlin) // :113-129
{
    wn_grid g = wnhost::lattice2d(imageSize, octave, 1.0f, 1.0f, WN_GRID_DEFAULT);
    g.z_mode = WN_Z_CONST;
    g.z_const = 1.0f * g.octave_scale; // :122
    wnhost::run_grid(imageSize, outputFile, [&](float *out) {
        wnhost::check(wn_perlin_grid(perlin.perm(), &g, out, nullptr), "wn_perlin_grid");
    });
    std::cout << "Generated Perlin 3D Sliced Octave " << octave << " noise: " << outputFile << std::endl;
}
