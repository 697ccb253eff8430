// WaveletNoise.h -- the reference's class WaveletNoise (WaveletNoise.h:20-59) over the MI355X
// C ABI.  Same constructor, member names and signatures, so callers written against the
// reference (texture.h, experient/main.cpp) compile unchanged; the work is done by HIP kernels.
//
//   generateNoiseTile2D/3D : Gaussian field drawn from the member mt19937 / normal_distribution
//                            exactly as the reference draws it (libstdc++ stream), filter passes
//                            on the GPU (wn_tile_generate_from_field); coefficients stay
//                            resident in HBM and are mirrored for getNoiseCoefficients().
//   evaluate2D/3D/3DProjected(p) : one sample, on the host from the mirrored coefficients (scalar_eval.h,
//                            bit-identical to the reference and to the kernels; WN_SCALAR_ON_DEVICE=1: one
//                            request to the resident scalar kernel, ~2.6 us); the batched overloads below
//                            are the GPU path.
#ifndef WAVELET_NOISE_H
#define WAVELET_NOISE_H

#include <cstddef>
#include <iostream>
#include <limits>
#include <random>
#include <string>
#include <vector>

struct wn_tile; // include/wnoise.h

// Statistical analysis structure (WaveletNoise.h:11-18)
struct DataStats {
    float avg = 0.0f;
    float var = 0.0f;
    float min_val = std::numeric_limits<float>::max();
    float max_val = std::numeric_limits<float>::lowest();
    long long count_nan_inf = 0;
    float energy = 0.0f; // Sum of squares
};

class WaveletNoise {
  public:
    WaveletNoise(int tileSize, unsigned int seed = 0); // WaveletNoise.cpp:20-26
    ~WaveletNoise();
    WaveletNoise(const WaveletNoise &) = delete;
    WaveletNoise &operator=(const WaveletNoise &) = delete;

    void generateNoiseTile2D(); // WaveletNoise.cpp:69-108
    void generateNoiseTile3D(); // WaveletNoise.cpp:142-183

    float evaluate2D(const float p[2]) const;                                  // :111-140
    float evaluate3D(const float p[3]) const;                                  // :185-215
    float evaluate3DProjected(const float p[3], const float normal[3]) const;  // :218-265

    DataStats calculateStats(const std::vector<float> &data, const std::string &name) const; // :268-288
    const std::vector<float> &getNoiseCoefficients() const;
    int getTileSize() const;

    // ---- additive: batched forms of the same members (host pointers; n points) ----------------
    void evaluate2D(const float *xy, size_t n, float *out) const;
    void evaluate3D(const float *xyz, size_t n, float *out) const;
    void evaluate3DProjected(const float *xyz, const float *normals, size_t n, float *out) const;
    // Cook & DeRose Appendix 2 WMultibandNoise (normal == NULL branch); absent from the reference.
    float WMultibandNoise(const float p[3], float s, int firstBand, int nbands, const float *w,
                          float variance = 0.18402f) const;
    void WMultibandNoise(const float *xyz, size_t n, float s, int firstBand, int nbands,
                         const float *w, float variance, float *out) const;
    // The paper's full signature: normal != nullptr makes every band WProjectedNoise (evaluate3DProjected) and
    // normalises with 0.296; normal == nullptr is the overload above with the paper's 0.210 replaced by `variance`.
    float WMultibandNoise(const float p[3], float s, const float *normal, int firstBand, int nbands,
                          const float *w, float variance = 0.296f) const;
    // The device-resident tile (an empty tile before generate*); for the C-ABI grid entry points.
    const wn_tile *tile(int dims) const;

  private:
    int tileSizeN;
    std::vector<float> noiseCoefficients;
    unsigned int randomSeed;
    std::mt19937 rng;
    std::normal_distribution<float> gaussianDist;
    mutable wn_tile *tile_;
    int tileDims = 0; // dimension of the generated tile (0: none yet)
    void generate(int dims);
};

#endif
