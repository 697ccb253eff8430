// texture.h -- the reference's texture adaptor (texture.h:14-115) over the MI355X C ABI.
//
// Same class names, constructors and `value(u, v, p)` signature, so material.h / main.cpp of the
// reference compile against this header unchanged.  value() is one sample, evaluated on the host (scalar_eval.h;
// WN_SCALAR_ON_DEVICE=1: one request to the resident scalar kernel, ~2.6 us) and returns the reference's colour bit for
// bit; values() is the batched GPU form a renderer should use (hit points in, grey levels out, optional per-hit `active`
// bytes compacted with wavefront ballots on the device).
#ifndef TEXTURE_H
#define TEXTURE_H

#include <cstddef>
#include <cstdint>
#include <memory>

#include "WaveletNoise.h"
#include "perlin.h"
#include "scalar_eval.h"
#include "vec3.h"
#include "wn_host.hpp"

using color = vec3;
using point3 = vec3;

class texture {
  public:
    virtual ~texture() = default;
    virtual color value(double u, double v, const point3 &p) const = 0;
};

class solid_color : public texture {
  public:
    solid_color(const color &albedo) : albedo(albedo) {}
    solid_color(double red, double green, double blue) : solid_color(color(red, green, blue)) {}
    color value(double, double, const point3 &) const override { return albedo; }

  private:
    color albedo;
};

namespace wnhost {
// grey levels for n host points through a device round trip
template <typename Launch> inline void texture_batch(const float *xyz, const uint8_t *active, size_t n,
                                                     float *grey, Launch launch)
{
    if (!n) return;
    DeviceBuffer in(3 * n * sizeof(float)), out(n * sizeof(float));
    in.upload(xyz);
    out.upload(grey); // inactive points keep the caller's value
    if (active) {
        DeviceBuffer act(n);
        act.upload(active);
        launch(in.as<float>(), act.as<uint8_t>(), out.as<float>());
        out.download(grey);
    } else {
        launch(in.as<float>(), static_cast<const uint8_t *>(nullptr), out.as<float>());
        out.download(grey);
    }
}
} // namespace wnhost

class noise_texture : public texture {
  public:
    noise_texture(double scale, int octave = 4) : scale(scale), octave_level(octave) {}

    color value(double, double, const point3 &p) const override // texture.h:37-43
    {
        const float xyz[3] = {p.x(), p.y(), p.z()};
        float g = 0.0f;
        if (!wnhost_scalar_on_device()) { // one sample: on the host (scalar_eval.h)
            g = wnhost_noise_texture_value(noise.table().data(), scale, octave_level, xyz);
            return color(g, g, g);
        }
        // WN_SCALAR_ON_DEVICE=1: one request to the resident scalar kernel (include/wnoise.h, wn_scalar_*)
        wnhost::check(wn_scalar_noise_texture(noise.perm(), scale, octave_level, xyz, &g), "wn_scalar_noise_texture");
        return color(g, g, g);
    }
    // additive: batched grey levels (host pointers); active == nullptr means every point
    void values(const float *xyz, const uint8_t *active, size_t n, float *grey) const
    {
        wnhost::texture_batch(xyz, active, n, grey, [&](const float *in, const uint8_t *act, float *out) {
            wnhost::check(wn_noise_texture_points(noise.perm(), scale, octave_level, in, act, n, out, nullptr),
                          "wn_noise_texture_points");
        });
    }
    // additive: device pointers, enqueue only (default stream)
    void values_device(const float *xyz_dev, size_t n, float *grey_dev) const
    {
        wnhost::check(wn_noise_texture_points(noise.perm(), scale, octave_level, xyz_dev, nullptr, n, grey_dev, nullptr),
                      "wn_noise_texture_points");
    }

  private:
    perlin noise; // default seed, texture.h:46
    double scale;
    int octave_level;
};

class wavelet_texture : public texture {
  public:
    wavelet_texture(double scale = 1.0, int octave = 4, bool use_3d = true)
        : scale(scale), octave_level(octave), use_3d_noise(use_3d)
    {
        const int TILE_SIZE = 128;        // texture.h:55
        const unsigned int SEED = 12345;  // texture.h:56
        noise_2d = std::make_unique<WaveletNoise>(TILE_SIZE, SEED);
        noise_2d->generateNoiseTile2D();
        if (use_3d_noise) {
            noise_3d = std::make_unique<WaveletNoise>(TILE_SIZE, SEED);
            noise_3d->generateNoiseTile3D();
        }
    }

    color value(double, double, const point3 &p) const override // texture.h:67-107
    {
        const float xyz[3] = {p.x(), p.y(), p.z()};
        const bool three = use_3d_noise && noise_3d;
        float g = 0.0f;
        if (!wnhost_scalar_on_device()) { // one sample: on the host (scalar_eval.h)
            const WaveletNoise *src = three ? noise_3d.get() : noise_2d.get();
            const std::vector<float> *c = src ? &src->getNoiseCoefficients() : nullptr;
            g = wnhost_wavelet_texture_value(c && !c->empty() ? c->data() : nullptr, src ? src->getTileSize() : 0, three ? 1 : 0,
                                             scale, octave_level, xyz);
            return color(g, g, g);
        }
        wnhost::check(wn_scalar_wavelet_texture(source(three), three ? 1 : 0, scale, octave_level, xyz, &g),
                      "wn_scalar_wavelet_texture");
        return color(g, g, g);
    }
    void values(const float *xyz, const uint8_t *active, size_t n, float *grey) const
    {
        const bool three = use_3d_noise && noise_3d;
        wnhost::texture_batch(xyz, active, n, grey, [&](const float *in, const uint8_t *act, float *out) {
            wnhost::check(wn_wavelet_texture_points(source(three), three ? 1 : 0, scale, octave_level, in,
                                                    act, n, out, nullptr), "wn_wavelet_texture_points");
        });
    }
    // additive: device pointers, enqueue only (default stream)
    void values_device(const float *xyz_dev, size_t n, float *grey_dev) const
    {
        const bool three = use_3d_noise && noise_3d;
        wnhost::check(wn_wavelet_texture_points(source(three), three ? 1 : 0, scale, octave_level, xyz_dev, nullptr, n,
                                                grey_dev, nullptr), "wn_wavelet_texture_points");
    }

  private:
    const wn_tile *source(bool three) const { return three ? noise_3d->tile(3) : noise_2d->tile(2); }
    std::unique_ptr<WaveletNoise> noise_2d;
    std::unique_ptr<WaveletNoise> noise_3d;
    double scale;
    int octave_level;
    bool use_3d_noise;
};

#endif
