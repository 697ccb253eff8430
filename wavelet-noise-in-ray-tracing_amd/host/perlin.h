// perlin.h -- the reference's class perlin (perlin.h:14-91) over the MI355X C ABI.
//
// Same constructor and members; the permutation table is built on the host by the same libstdc++
// calls as the reference (iota + std::shuffle(mt19937(seed)), perlin.h:34-39, via
// wn_perm_create_seeded) and lives on the device as 512 bytes.  The batched forms run as HIP kernels in fp64 with the
// reference's operation order; the scalar members are evaluated on the host from the mirrored table (scalar_eval.h;
// WN_SCALAR_ON_DEVICE=1: by the resident scalar kernel).  Bit-identical either way.
// Additive: turb() (RTOW; absent from the reference) and batched overloads.
#ifndef PERLIN_H
#define PERLIN_H

#include <cstddef>
#include <random>
#include <vector>

#include "scalar_eval.h"
#include "vec3.h"
#include "wn_host.hpp"

using point3 = vec3;

class perlin {
  private:
    std::vector<int> p; // host mirror of the table (perlin.h:16)
    wn_perm *perm_ = nullptr;

    double scalar(const point3 &q, int kind, int depth) const // kind 1: turb, 2: fractal_noise
    {
        const float xyz[3] = {q.x(), q.y(), q.z()};
        if (!wnhost_scalar_on_device()) return kind == 1 ? wnhost_perlin_turb(p.data(), xyz, depth) : wnhost_perlin_fractal(p.data(), xyz);
        double v = 0.0;
        wnhost::check(wn_scalar_perlin_vec3(perm_, xyz, kind, depth, &v), "wn_scalar_perlin_vec3");
        return v;
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
        if (!wnhost_scalar_on_device()) return wnhost_perlin(p.data(), x, y, z);
        double v = 0.0;
        wnhost::check(wn_scalar_perlin(perm_, x, y, z, &v), "wn_scalar_perlin");
        return v;
    }
    double noise(double x, double y) const { return noise(x, y, 0.0); }              // perlin.h:65-67
    double noise(const point3 &q) const { return noise(q.x(), q.y(), q.z()); }        // perlin.h:70-72

    double fractal_noise(const point3 &q) const                                       // perlin.h:75-90
    {
        return scalar(q, 2, 0);
    }
    // RTOW "The Next Week" turb(p, depth); absent from the reference.
    double turb(const point3 &q, int depth = 7) const
    {
        return scalar(q, 1, depth);
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
