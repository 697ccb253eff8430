// gridgen.cpp -- the role of the reference's experient/main.cpp (:131-168): build the 2-D and 3-D
// wavelet tiles (tile 128, seed 12345) and the Perlin table, then write the five 256x256 raw grids
// for octaves 3, 4, 5 into <outdir>/ -- 15 batched GPU launches instead of ~1M scalar calls.
//
//   gridgen [outdir=result_raw] [--fast] [--size N]
//
// By default all 15 files are byte-identical to the reference's committed
// experient/result_raw/*.raw.  --fast routes the 3-D sliced wavelet grids through the separable
// brick kernel (within 1e-5 of the reference); --exact is accepted and is the default.
#include <sys/stat.h>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "noise_grid.h"

int main(int argc, char **argv)
{
    std::string outdir = "result_raw";
    int flags = WN_GRID_EXACT, image = 256;
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--exact")) flags = WN_GRID_EXACT;
        else if (!std::strcmp(argv[i], "--fast")) flags = WN_GRID_DEFAULT;
        else if (!std::strcmp(argv[i], "--size") && i + 1 < argc) image = std::atoi(argv[++i]);
        else outdir = argv[i];
    }
    std::cout << "=== Wavelet & Perlin Noise Comparison Generation (MI355X) ===" << std::endl;
    mkdir(outdir.c_str(), 0755);

    const int TILE_SIZE = 128;          // experient/main.cpp:137
    const unsigned int SEED = 12345;    // :138
    try {
        WaveletNoise noise2D(TILE_SIZE, SEED);
        noise2D.generateNoiseTile2D();
        WaveletNoise noise3D(TILE_SIZE, SEED);
        noise3D.generateNoiseTile3D();
        PerlinNoise perlin(SEED);
        for (int octave : {3, 4, 5}) {
            const std::string o = std::to_string(octave);
            generate2DOctaveBandNoise(image, octave, outdir + "/wavelet_noise_2D_octave_" + o + ".raw", noise2D);
            generate3DSlicedOctaveBandNoise(image, octave, outdir + "/wavelet_noise_3Dsliced_octave_" + o + ".raw", noise3D, flags);
            generate3DProjectedOctaveBandNoise(image, octave, outdir + "/wavelet_noise_3Dprojected_octave_" + o + ".raw", noise3D);
            generatePerlinNoise2D(image, octave, outdir + "/perlin_noise_2D_octave_" + o + ".raw", perlin);
            generatePerlinNoise3DSliced(image, octave, outdir + "/perlin_noise_3Dsliced_octave_" + o + ".raw", perlin);
        }
    } catch (const std::exception &e) {
        std::cerr << "gridgen: " << e.what() << std::endl;
        return 1;
    }
    std::cout << "=== Generation Complete ===" << std::endl;
    return 0;
}
