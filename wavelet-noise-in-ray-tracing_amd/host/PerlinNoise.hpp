// PerlinNoise.hpp -- the reference's experient/PerlinNoise.hpp:9-61 (the same improved-noise
// algorithm as perlin.h with an explicit seed and no vec3 overloads) over the MI355X C ABI.
#ifndef PERLINNOISE_HPP
#define PERLINNOISE_HPP

#include <random>
#include <vector>

#include "scalar_eval.h"
#include "wn_host.hpp"

class PerlinNoise {
  private:
    std::vector<int> p;
    wn_perm *perm_ = nullptr;

  public:
    explicit PerlinNoise(unsigned int seed = std::mt19937::default_seed) : p(512)
    {
        wnhost::check(wn_perm_create_seeded(seed, &perm_), "wn_perm_create_seeded");
        wnhost::check(wn_perm_download(perm_, p.data()), "wn_perm_download");
    }
    ~PerlinNoise() { wn_perm_destroy(perm_); }
    PerlinNoise(const PerlinNoise &) = delete;
    PerlinNoise &operator=(const PerlinNoise &) = delete;

    double noise(double x, double y, double z) const // PerlinNoise.hpp:36-56
    {
        if (!wnhost_scalar_on_device()) return wnhost_perlin(p.data(), x, y, z); // one sample: on the host (scalar_eval.h)
        double v = 0.0;
        wnhost::check(wn_scalar_perlin(perm_, x, y, z, &v), "wn_scalar_perlin");
        return v;
    }
    double noise(double x, double y) const { return noise(x, y, 0.0); } // PerlinNoise.hpp:58-60

    const std::vector<int> &table() const { return p; }
    const wn_perm *perm() const { return perm_; }
};

#endif
