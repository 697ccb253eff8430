// Host check of csrc/wn_perlin_run.hpp (the gradient bookkeeping of perlin_grid_run_kernel):
//  (1) for every hash nibble, corner offset and a set of coordinates that includes both zeros,
//      run_gradient(run_k_entry(h, dy, dz), dx) has the bits of grad(h, dx, dy, dz) (perlin.h:26-31);
//  (2) a scalar emulation of the kernel's per-row algorithm (per-axis tables, one hash per cell of a
//      run, table look-up of {K, mm, t}) reproduces noise(x, y, z) (perlin.h:42-62) bit for bit on
//      lattice rows, negative coordinates and steps from 1/128 to 1.5 included.
// Test infrastructure: built by tests/test_perlin_run_host.py with g++ -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "wn_perlin_run.hpp"

static uint64_t bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

// literal restatement of perlin.h:18-31, 42-62
static double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
static double lerp(double t, double a, double b) { return a + t * (b - a); }
static double grad(int hash, double x, double y, double z)
{
    int h = hash & 15;
    double u = h < 8 ? x : y;
    double v = h < 4 ? y : (h == 12 || h == 14 ? x : z);
    return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}
static double noise(const int *p, double x, double y, double z)
{
    int X = (int)floor(x) & 255, Y = (int)floor(y) & 255, Z = (int)floor(z) & 255;
    x -= floor(x); y -= floor(y); z -= floor(z);
    double u = fade(x), v = fade(y), w = fade(z);
    int A = p[X] + Y, AA = p[A] + Z, AB = p[A + 1] + Z, B = p[X + 1] + Y, BA = p[B] + Z, BB = p[B + 1] + Z;
    return lerp(w, lerp(v, lerp(u, grad(p[AA], x, y, z), grad(p[BA], x - 1, y, z)),
                        lerp(u, grad(p[AB], x, y - 1, z), grad(p[BB], x - 1, y - 1, z))),
                lerp(v, lerp(u, grad(p[AA + 1], x, y, z - 1), grad(p[BA + 1], x - 1, y, z - 1)),
                     lerp(u, grad(p[AB + 1], x, y - 1, z - 1), grad(p[BB + 1], x - 1, y - 1, z - 1))));
}

// the kernel's row algorithm, one lane at a time
static double run_sample(const int *p, double cx, double cy, double cz)
{
    const double fx = floor(cx), fy = floor(cy), fz = floor(cz);
    const int X = (int)fx & 255, Y = (int)fy & 255, Z = (int)fz & 255;
    const double xf = cx - fx, yf = cy - fy, zf = cz - fz;
    const double u = fade(xf), v = fade(yf), w = fade(zf);
    wn::RunKEntry tab[64];
    for (int l = 0; l < 64; ++l) {
        const int kh = l & 15, kcy = (l >> 4) & 1, kcz = l >> 5;
        tab[l] = wn::run_k_entry(kh, kcy ? yf - 1.0 : yf, kcz ? zf - 1.0 : zf);
    }
    const int A = p[X] + Y, AA = p[A] + Z, AB = p[A + 1] + Z, B = p[X + 1] + Y, BA = p[B] + Z, BB = p[B + 1] + Z;
    const int h[8] = {p[AA], p[BA], p[AB], p[BB], p[AA + 1], p[BA + 1], p[AB + 1], p[BB + 1]};
    const double xm1 = xf - 1.0;
    double gr[8];
    for (int c = 0; c < 8; ++c) {
        const wn::RunKEntry e = tab[(c >> 1) * 16 + (h[c] & 15)];
        gr[c] = wn::run_gradient(e.K, e.mm, e.t, bits((c & 1) ? xm1 : xf));
    }
    const double x00 = lerp(u, gr[0], gr[1]), x10 = lerp(u, gr[2], gr[3]);
    const double x01 = lerp(u, gr[4], gr[5]), x11 = lerp(u, gr[6], gr[7]);
    return lerp(w, lerp(v, x00, x10), lerp(v, x01, x11));
}

int main()
{
    long bad = 0, checked = 0;
    const double vals[] = {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 0.25, 0.999999, -1e-300, 1e-300, 0.3, -0.7, 0.123456789, -0.987654321};
    const int nv = sizeof(vals) / sizeof(vals[0]);
    for (int h = 0; h < 16; ++h)
        for (int a = 0; a < nv; ++a)
            for (int b = 0; b < nv; ++b)
                for (int c = 0; c < nv; ++c) {
                    const wn::RunKEntry e = wn::run_k_entry(h, vals[b], vals[c]);
                    const double got = wn::run_gradient(e.K, e.mm, e.t, bits(vals[a]));
                    const double want = grad(h, vals[a], vals[b], vals[c]);
                    ++checked;
                    if (bits(got) != bits(want)) {
                        if (bad < 10) printf("grad mismatch h=%d dx=%g dy=%g dz=%g got=%a want=%a\n", h, vals[a], vals[b], vals[c], got, want);
                        ++bad;
                    }
                }
    std::mt19937 rng(7);
    std::vector<int> perm(512);
    for (int i = 0; i < 256; ++i) perm[i] = i;
    for (int i = 255; i > 0; --i) std::swap(perm[i], perm[rng() % (i + 1)]);
    for (int i = 0; i < 256; ++i) perm[256 + i] = perm[i];
    const float steps[] = {1.0f / 128, 1.0f / 8, 0.25f, 0.5f, 1.0f, 1.5f, 0.3f, -0.125f};
    for (float st : steps)
        for (int row = 0; row < 6; ++row) {
            const float y = (float)row * st * 3.0f - 2.0f, z = (float)row * st - 1.0f;
            for (int i = 0; i < 2100; ++i) {
                const float x = (float)(i - 300) * st;
                const double got = run_sample(perm.data(), (double)x, (double)y, (double)z);
                const double want = noise(perm.data(), (double)x, (double)y, (double)z);
                ++checked;
                if (bits(got) != bits(want)) {
                    if (bad < 20) printf("noise mismatch step=%g x=%g y=%g z=%g got=%a want=%a\n", st, x, y, z, got, want);
                    ++bad;
                }
            }
        }
    printf("checked %ld, mismatches %ld\n", checked, bad);
    return bad ? 1 : 0;
}
