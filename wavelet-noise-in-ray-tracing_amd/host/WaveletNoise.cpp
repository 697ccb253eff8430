// WaveletNoise.cpp -- host side of class WaveletNoise over libwnoise_hip.so: tile filtering and every evaluation of
// more than one sample are HIP kernels behind include/wnoise.h; the scalar members (one sample, needed at once) are
// evaluated where the caller is (scalar_eval.h), or by the resident scalar kernel with WN_SCALAR_ON_DEVICE=1.
#include "WaveletNoise.h"

#include <algorithm>
#include <cmath>

#include "scalar_eval.h"
#include "wn_host.hpp"

using wnhost::check;

WaveletNoise::WaveletNoise(int tileSize, unsigned int seed)
    : tileSizeN(wn_tile_even_size(tileSize)), randomSeed(seed), rng(seed), gaussianDist(0.0f, 1.0f),
      tile_(nullptr)
{
    if (tileSizeN != tileSize) // WaveletNoise.cpp:22-25
        std::cerr << "Warning: Tile size adjusted to " << tileSizeN << " (must be even)" << std::endl;
}

WaveletNoise::~WaveletNoise() { wn_tile_destroy(tile_); }

void WaveletNoise::generate(int dims)
{
    const size_t n = static_cast<size_t>(tileSizeN);
    const size_t count = dims == 2 ? n * n : n * n * n;
    std::vector<float> field(count);
    for (size_t i = 0; i < count; ++i) field[i] = gaussianDist(rng); // WaveletNoise.cpp:74-77,146-147
    wn_tile *t = nullptr;
    check(wn_tile_generate_from_field(tileSizeN, dims, field.data(), &t), "wn_tile_generate_from_field");
    wn_tile_destroy(tile_);
    tile_ = t;
    noiseCoefficients.resize(count);
    check(wn_tile_download(tile_, noiseCoefficients.data()), "wn_tile_download");
    tileDims = dims;
}

void WaveletNoise::generateNoiseTile2D() { generate(2); }
void WaveletNoise::generateNoiseTile3D() { generate(3); }

const wn_tile *WaveletNoise::tile(int dims) const
{
    if (!tile_) check(wn_tile_create(0, dims, nullptr, &tile_), "wn_tile_create"); // empty: evaluates to 0
    return tile_;
}

// ---- scalar members: evaluated on the host from the mirrored coefficients (scalar_eval.h; bit-identical to the kernels), or one
// request each to the resident scalar kernel (wn_scalar_*, include/wnoise.h) with WN_SCALAR_ON_DEVICE=1.  A tile of the other
// dimension goes to the C ABI either way, which reports it.
float WaveletNoise::evaluate2D(const float p[2]) const
{
    if (!wnhost_scalar_on_device() && tileDims != 3)
        return wnhost_eval2d(noiseCoefficients.empty() ? nullptr : noiseCoefficients.data(), tileSizeN, p);
    float v = 0.0f;
    check(wn_scalar_eval2d(tile(2), p, &v), "wn_scalar_eval2d");
    return v;
}

float WaveletNoise::evaluate3D(const float p[3]) const
{
    if (!wnhost_scalar_on_device() && tileDims != 2)
        return wnhost_eval3d(noiseCoefficients.empty() ? nullptr : noiseCoefficients.data(), tileSizeN, p);
    float v = 0.0f;
    check(wn_scalar_eval3d(tile(3), p, &v), "wn_scalar_eval3d");
    return v;
}

float WaveletNoise::evaluate3DProjected(const float p[3], const float normal[3]) const
{
    if (!wnhost_scalar_on_device() && tileDims != 2)
        return wnhost_eval3d_projected(noiseCoefficients.empty() ? nullptr : noiseCoefficients.data(), tileSizeN, p, normal);
    float v = 0.0f;
    check(wn_scalar_eval3d_projected(tile(3), p, normal, &v), "wn_scalar_eval3d_projected");
    return v;
}

float WaveletNoise::WMultibandNoise(const float p[3], float sarg, int firstBand, int nbands,
                                    const float *w, float variance) const
{
    auto &s = wnhost::Scratch::get();
    std::copy(p, p + 3, s.in_host());
    check(wn_multiband3d_points(tile(3), static_cast<const float *>(s.in_dev()), 1, sarg, firstBand,
                                nbands, w, variance, static_cast<float *>(s.out_dev()), nullptr),
          "wn_multiband3d_points");
    check(wn_stream_sync(nullptr), "wn_stream_sync");
    return s.out_host()[0];
}

float WaveletNoise::WMultibandNoise(const float p[3], float sarg, const float *normal, int firstBand,
                                    int nbands, const float *w, float variance) const
{
    if (!normal) return WMultibandNoise(p, sarg, firstBand, nbands, w, variance);
    auto &s = wnhost::Scratch::get();
    std::copy(p, p + 3, s.in_host());
    std::copy(normal, normal + 3, s.in_host() + 4);
    const float *in = static_cast<const float *>(s.in_dev());
    check(wn_multiband3d_projected_points(tile(3), in, in + 4, 1, 1, sarg, firstBand, nbands, w, variance,
                                          static_cast<float *>(s.out_dev()), nullptr),
          "wn_multiband3d_projected_points");
    check(wn_stream_sync(nullptr), "wn_stream_sync");
    return s.out_host()[0];
}

// ---- batched members ----------------------------------------------------------------------------------
void WaveletNoise::evaluate2D(const float *xy, size_t n, float *out) const
{
    if (!n) return;
    wnhost::DeviceBuffer in(2 * n * sizeof(float)), res(n * sizeof(float));
    in.upload(xy);
    check(wn_eval2d_points(tile(2), in.as<float>(), n, res.as<float>(), nullptr), "wn_eval2d_points");
    res.download(out);
}

void WaveletNoise::evaluate3D(const float *xyz, size_t n, float *out) const
{
    if (!n) return;
    wnhost::DeviceBuffer in(3 * n * sizeof(float)), res(n * sizeof(float));
    in.upload(xyz);
    check(wn_eval3d_points(tile(3), in.as<float>(), n, res.as<float>(), nullptr), "wn_eval3d_points");
    res.download(out);
}

void WaveletNoise::evaluate3DProjected(const float *xyz, const float *normals, size_t n, float *out) const
{
    if (!n) return;
    wnhost::DeviceBuffer in(3 * n * sizeof(float)), nr(3 * n * sizeof(float)), res(n * sizeof(float));
    in.upload(xyz);
    nr.upload(normals);
    check(wn_eval3d_projected_points(tile(3), in.as<float>(), nr.as<float>(), n, res.as<float>(), nullptr),
          "wn_eval3d_projected_points");
    res.download(out);
}

void WaveletNoise::WMultibandNoise(const float *xyz, size_t n, float sarg, int firstBand, int nbands,
                                   const float *w, float variance, float *out) const
{
    if (!n) return;
    wnhost::DeviceBuffer in(3 * n * sizeof(float)), res(n * sizeof(float));
    in.upload(xyz);
    check(wn_multiband3d_points(tile(3), in.as<float>(), n, sarg, firstBand, nbands, w, variance,
                                res.as<float>(), nullptr), "wn_multiband3d_points");
    res.download(out);
}

// ---- debug helper (WaveletNoise.cpp:268-288; no caller in the reference; host-side statistics) --------
DataStats WaveletNoise::calculateStats(const std::vector<float> &data, const std::string &name) const
{
    DataStats stats;
    if (data.empty()) return stats;
    double total = 0.0, total_sq = 0.0;
    for (float v : data) {
        total += v;
        total_sq += static_cast<double>(v) * v;
        stats.min_val = std::min(stats.min_val, v);
        stats.max_val = std::max(stats.max_val, v);
    }
    stats.avg = static_cast<float>(total / data.size());
    stats.var = static_cast<float>(total_sq / data.size() - static_cast<double>(stats.avg) * stats.avg);
    std::cout << name << " stats: avg=" << stats.avg << ", var=" << stats.var
              << ", stddev=" << std::sqrt(stats.var) << std::endl;
    return stats;
}

const std::vector<float> &WaveletNoise::getNoiseCoefficients() const { return noiseCoefficients; }
int WaveletNoise::getTileSize() const { return tileSizeN; }
