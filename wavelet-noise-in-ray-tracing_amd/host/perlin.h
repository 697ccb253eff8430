// perlin.h -- the reference's class perlin (perlin.h:14-91) over the MI355X C ABI.
//
// Same constructor and members; the permutation table is built on the host by the same libstdc++
// calls as the reference (iota + std::shuffle(mt19937(seed)), perlin.h:34-39, via
// wn_perm_create_seeded) and lives on the device as 512 bytes.  noise / fractal_noise run as HIP
// kernels in fp64 with the reference's operation order: results are bit-identical.
// Additive: turb() (RTOW; absent from the reference) and batched overloads.
#ifndef PERLIN_H
#define PERLIN_H

#include <cstddef>
#include <random>
#include <vector>

#include "vec3.h"
#include "wn_host.hpp"

using point3 = vec3;

class perlin {
  private:
    std::vector<int> p; // host mirror of the table (perlin.h:16)
    wn_perm *perm_ = nullptr;

    template <typename Launch> double scalar(const float xyz[3], Launch launch) const
    {
        auto &s = wnhost::Scratch::get();
        for (int i = 0; i < 3; ++i) s.in_host()[i] = xyz[i];
        launch(static_cast<const float *>(s.in_dev()), static_cast<double *>(s.out_dev()));
        wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
        return s.out_host64()[0];
    }

  public:
    explicit perlin(unsigned int seed = std::mt19937::default_seed) : p(512)
    {
        wnhost::check(wn_perm_create_seeded(seed, &perm_), "wn_perm_create_seeded");
        wnhost::check(wn_perm_download(perm_, p.data()), "wn_perm_download");
    }
    ~perlin() { wn_perm_destroy(perm_); }
    perlin(const perlin &o) : p(o.p) { wnhost::check(wn_perm_create(p.data(), &perm_), "wn_perm_create"); }
    perlin &operator=(const perlin &) = delete;

    // perlin.h:42-62
    double noise(double x, double y, double z) const noexcept(false)
    {
        auto &s = wnhost::Scratch::get();
        s.in_host64()[0] = x;
        s.in_host64()[1] = y;
        s.in_host64()[2] = z;
        wnhost::check(wn_perlin_points(perm_, static_cast<const double *>(s.in_dev()), 1,
                                       static_cast<double *>(s.out_dev()), nullptr), "wn_perlin_points");
        wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
        return s.out_host64()[0];
    }
    double noise(double x, double y) const { return noise(x, y, 0.0); }              // perlin.h:65-67
    double noise(const point3 &q) const { return noise(q.x(), q.y(), q.z()); }        // perlin.h:70-72

    double fractal_noise(const point3 &q) const                                       // perlin.h:75-90
    {
        const float xyz[3] = {q.x(), q.y(), q.z()};
        return scalar(xyz, [&](const float *in, double *out) {
            wnhost::check(wn_perlin_fractal_points(perm_, in, 1, out, nullptr), "wn_perlin_fractal_points");
        });
    }
    // RTOW "The Next Week" turb(p, depth); absent from the reference.
    double turb(const point3 &q, int depth = 7) const
    {
        const float xyz[3] = {q.x(), q.y(), q.z()};
        return scalar(xyz, [&](const float *in, double *out) {
            wnhost::check(wn_perlin_turb_points(perm_, in, 1, depth, out, nullptr), "wn_perlin_turb_points");
        });
    }

    // ---- additive: batched forms (host pointers) and the device-resident table -------------------
    void noise(const double *xyz, size_t n, double *out) const
    {
        if (!n) return;
        wnhost::DeviceBuffer in(3 * n * sizeof(double)), res(n * sizeof(double));
        in.upload(xyz);
        wnhost::check(wn_perlin_points(perm_, in.as<double>(), n, res.as<double>(), nullptr), "wn_perlin_points");
        res.download(out);
    }
    const std::vector<int> &table() const { return p; }
    const wn_perm *perm() const { return perm_; }
};

#endif
