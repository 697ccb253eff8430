// scalar_api_check.cpp -- drives the SCALAR members of the host classes (the reference's own API:
// evaluate2D/3D/3DProjected(p), noise(x,y,z), fractal_noise(p), texture::value(u,v,p)) on probe
// points read from stdin and prints the results with 9/17 significant digits; tests/ compare
// them with the reference's vectors.  Every call is a batch of one on the GPU.
//
// stdin: N, then N lines "x y z";  stdout: one line per point:
//   e2d e3d e3dp(normal 0,0,1) perlin12345(x,y,z as double) fractal(vec3) tex_wavelet3d(1.0,4) tex_perlin(1.0,4)
#include <cstdint>
#include <cstdio>
#include <iostream>
#include <memory>
#include <vector>

#include "texture.h"

int main()
{
    try {
        size_t n = 0;
        if (!(std::cin >> n)) return 2;
        std::vector<float> pts(3 * n);
        for (auto &v : pts) std::cin >> v;
        WaveletNoise n2(128, 12345), n3(128, 12345), empty(128, 1);
        n2.generateNoiseTile2D();
        n3.generateNoiseTile3D();
        perlin per(12345);
        std::shared_ptr<texture> wt = std::make_shared<wavelet_texture>(1.0, 4, true);
        std::shared_ptr<texture> pt = std::make_shared<noise_texture>(1.0, 4);
        const float normal[3] = {0.0f, 0.0f, 1.0f};
        for (size_t i = 0; i < n; ++i) {
            const float *p = &pts[3 * i];
            const point3 q(p[0], p[1], p[2]);
            std::printf("%.9g %.9g %.9g %.17g %.17g %.9g %.9g\n", n2.evaluate2D(p), n3.evaluate3D(p),
                        n3.evaluate3DProjected(p, normal), per.noise((double)p[0], (double)p[1], (double)p[2]),
                        per.fractal_noise(q), wt->value(0, 0, q).x(), pt->value(0, 0, q).x());
        }
        // batched overloads must return exactly what the scalar members return
        {
            std::vector<float> b3(n), b2(n), xy(2 * n), grey(n, -7.0f), greyp(n, -7.0f);
            std::vector<uint8_t> active(n);
            for (size_t i = 0; i < n; ++i) {
                xy[2 * i] = pts[3 * i];
                xy[2 * i + 1] = pts[3 * i + 1];
                active[i] = (uint8_t)(i % 3 != 0);
            }
            n3.evaluate3D(pts.data(), n, b3.data());
            n2.evaluate2D(xy.data(), n, b2.data());
            auto *wtex = dynamic_cast<wavelet_texture *>(wt.get());
            auto *ptex = dynamic_cast<noise_texture *>(pt.get());
            wtex->values(pts.data(), active.data(), n, grey.data());
            ptex->values(pts.data(), active.data(), n, greyp.data());
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) {
                const float *p = &pts[3 * i];
                const point3 q(p[0], p[1], p[2]);
                bad += b3[i] != n3.evaluate3D(p);
                bad += b2[i] != n2.evaluate2D(p);
                bad += grey[i] != (active[i] ? wt->value(0, 0, q).x() : -7.0f);
                bad += greyp[i] != (active[i] ? pt->value(0, 0, q).x() : -7.0f);
            }
            std::printf("batch_vs_scalar_mismatches %zu\n", bad);
        }
        // a renderer-sized masked batch through the host classes: 200,000 points, ~59 % active (the reference scene's share
        // of primary rays that reach a noise-textured surface): the wave-ballot compaction path of the texture kernels as a
        // C++ caller reaches it; active lanes must equal the unmasked batch, inactive ones keep what was there
        {
            const size_t m = 200000;
            std::vector<float> big(3 * m), full(m), masked(m, -7.0f), fullp(m), maskedp(m, -7.0f);
            std::vector<uint8_t> active(m);
            uint32_t st = 12345u;
            auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (st >> 8) * (1.0f / 16777216.0f); };
            size_t on = 0;
            for (size_t i = 0; i < m; ++i) {
                big[3 * i] = rnd() * 20.0f - 10.0f;
                big[3 * i + 1] = (i % 7 == 0) ? rnd() - 0.5f : -0.5f;
                big[3 * i + 2] = rnd() * 20.0f - 10.0f;
                active[i] = rnd() < 0.593f;
                on += active[i];
            }
            auto *wtex = dynamic_cast<wavelet_texture *>(wt.get());
            auto *ptex = dynamic_cast<noise_texture *>(pt.get());
            wtex->values(big.data(), nullptr, m, full.data());
            wtex->values(big.data(), active.data(), m, masked.data());
            ptex->values(big.data(), nullptr, m, fullp.data());
            ptex->values(big.data(), active.data(), m, maskedp.data());
            size_t bad = 0;
            for (size_t i = 0; i < m; ++i) {
                bad += masked[i] != (active[i] ? full[i] : -7.0f);
                bad += maskedp[i] != (active[i] ? fullp[i] : -7.0f);
            }
            std::printf("masked_batch_mismatches %zu active %zu of %zu\n", bad, on, m);
        }
        // conventions: an un-generated object evaluates to 0 (WaveletNoise.cpp:112,186,219)
        std::printf("empty %.9g %.9g\n", empty.evaluate3D(&pts[0]), empty.evaluate2D(&pts[0]));
        std::printf("tile %d coeffs %zu\n", n3.getTileSize(), n3.getNoiseCoefficients().size());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "scalar_api_check: %s\n", e.what());
        return 1;
    }
    return 0;
}
