// render.cpp -- batched renderer front-end: the reference's ray-trace
// loop (main.cpp:38-59, 119-204) with the noise evaluation taken out of the recursion.
//
// The reference evaluates tex->value(p) inside trace() at every lambertian bounce (material.h:72),
// one scalar call at a time.  The noise value only multiplies into the path's attenuation and
// never steers control flow or the rand() stream (main.cpp:54-55), so this front-end
//   1. traces the scene on the host exactly as the reference does (same float arithmetic, same
//      rand() draws, same fixed scene, main.cpp:126-163) and RECORDS each bounce's hit point,
//   2. evaluates all recorded points of a band of scanlines with ONE batched texture call on the
//      GPU (wavelet_texture::values / noise_texture::values -> wn_*_texture_points),
//   3. re-applies the attenuation products deepest-first (`attenuation * trace(...)` unwinds from
//      the last bounce) and accumulates the samples in the reference's order.
// Result: pixel-for-pixel the reference's image (tests compare with the decoded golden PNGs).
//
// Geometry here is plumbing for the noise path (scene intersection is out of scope for the GPU);
// it is restated only as far as the fixed scene needs it, with the reference's float/double
// promotions kept (vec3.h, ray.h, sphere.h:44-91, quad.h:18-66, hittable_list.h:26-40,
// material.h:23-29,63-74, rtweekend.h:37-40).  Build with -ffp-contract=off.
//
//   render [--width W] [--height H] [--spp S] [--noise 0|1] [--octave O] [--band-lines L]
//          [--out file.ppm] [--rgb file.rgb] [--dry-run]
//   --dry-run: no GPU; prints the count and FNV-1a64 of the hit-point stream (oracle/_ref check)
//   --dump-first N file: the first N hit points and their grey values as raw float32 records (x, y, z, grey)
//   --time-kernel: also times the texture kernel ALONE on every band's hit-point stream (points resident in
//                  HBM, HIP events, best of 3 launches): the device rate on the renderer's real stream
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "texture.h"

namespace {

// ---- float 3-vector arithmetic exactly as the reference's vec3.h --------------------------------
struct V {
    float x, y, z;
};
inline V add(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V scale(float t, V v) { return {t * v.x, t * v.y, t * v.z}; }      // operator*(float, vec3)
inline V divide(V v, float t) { return {v.x / t, v.y / t, v.z / t}; }     // operator/(vec3, float)
inline V mulv(V a, V b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }     // operator*(vec3, vec3)
inline float dotp(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float sqlen(V a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline float length(V a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V unit(V v)                                                        // vec3.h:140-143
{
    const double l = length(v);
    return {(float)(v.x / l), (float)(v.y / l), (float)(v.z / l)};
}
inline V cross_abs(V a, V b)                                              // vec3.h:90-95 (abs of each term)
{
    return {std::fabs(a.y * b.z - a.z * b.y), std::fabs(a.z * b.x - a.x * b.z), std::fabs(a.x * b.y - a.y * b.x)};
}
inline bool near_zero(V v) { return std::fabs(v.x) < 1e-8 && std::fabs(v.y) < 1e-8 && std::fabs(v.z) < 1e-8; }

struct Ray {
    V o, d;
    V at(float t) const { return add(o, scale(t, d)); }                   // ray.h:16-18
};

inline double random_double() { return std::rand() / (RAND_MAX + 1.0); } // rtweekend.h:37-40

// material.h:23-29.  g++ evaluates the three constructor arguments right to left, so the FIRST
// draw lands in z (checked against the reference's hit-point stream, oracle/_ref/raytrace_record).
inline V random_unit_vector()
{
    for (;;) {
        const float pz = (float)(random_double() * 2 - 1);
        const float py = (float)(random_double() * 2 - 1);
        const float px = (float)(random_double() * 2 - 1);
        const V p{px, py, pz};
        if (sqlen(p) >= 1) continue;
        return unit(p);
    }
}

enum Surface { kNone = 0, kNoiseLambertian = 1, kLight = 2 };

struct Hit {
    V p, normal;
    float t;
    int surface;
};

inline void face_normal(const Ray &r, V outward, Hit &h)                  // hittable.h:26-29
{
    const bool front = dotp(r.d, outward) < 0;
    h.normal = front ? outward : V{-outward.x, -outward.y, -outward.z};
}

struct Sphere {
    V c;
    float radius;
    int surface;
    bool hit(const Ray &r, float tmin, float tmax, Hit &h) const          // sphere.h:44-91
    {
        const V oc = sub(c, r.o);
        const float a = sqlen(r.d);
        const float hh = dotp(r.d, oc);
        const float c_term = sqlen(oc) - radius * radius;
        const float disc = hh * hh - a * c_term;
        if (disc < 0.0f) return false;
        const float sq = std::sqrt(disc);
        float root = (hh - sq) / a;
        if (root < tmin || root > tmax) {
            root = (hh + sq) / a;
            if (root < tmin || root > tmax) return false;
        }
        h.t = root;
        h.p = r.at(h.t);
        h.surface = surface;
        face_normal(r, divide(sub(h.p, c), radius), h);
        return true;
    }
};

struct Quad {
    V Q, u, v, w, normal;
    double D;
    int surface;
    Quad(V q, V uu, V vv, int s) : Q(q), u(uu), v(vv), surface(s)         // quad.h:18-27
    {
        const V n = cross_abs(u, v);
        normal = unit(n);
        D = dotp(normal, Q);
        w = divide(n, dotp(n, n));
    }
    bool hit(const Ray &r, float tmin, float tmax, Hit &h) const          // quad.h:38-66
    {
        const float denom = dotp(normal, r.d);
        if (std::fabs(denom) < 1e-8) return false;
        const double t = (D - dotp(normal, r.o)) / denom;
        if (t < tmin || t > tmax) return false;
        const V hitp = r.at((float)t);
        const V planar = sub(hitp, Q);
        const double alpha = dotp(w, cross_abs(planar, v));
        const double beta = dotp(w, cross_abs(u, planar));
        if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false; // quad.h:68-79
        h.t = (float)t;
        h.p = hitp;
        h.surface = surface;
        face_normal(r, normal, h);
        return true;
    }
};

// main.cpp:148-163: ground quad (noise), light sphere, noise sphere, tested in this order
struct Scene {
    Quad ground{V{-10, -0.5f, -10}, V{20, 0, 0}, V{0, 0, 20}, kNoiseLambertian};
    Sphere light{V{-5, 5, 0}, 0.8f, kLight};
    Sphere ball{V{1, 0, -1.75f}, 0.5f, kNoiseLambertian};
    bool hit(const Ray &r, float tmin, float tmax, Hit &out) const        // hittable_list.h:26-40
    {
        Hit tmp;
        bool any = false;
        float closest = tmax;
        if (ground.hit(r, tmin, closest, tmp)) { any = true; closest = tmp.t; out = tmp; }
        if (light.hit(r, tmin, closest, tmp)) { any = true; closest = tmp.t; out = tmp; }
        if (ball.hit(r, tmin, closest, tmp)) { any = true; closest = tmp.t; out = tmp; }
        return any;
    }
};

constexpr int kMaxDepth = 10; // main.cpp:30

struct Fnv {
    uint64_t h = 0xcbf29ce484222325ull, count = 0;
    void add(V p)
    {
        unsigned char b[12];
        const float v[3] = {p.x, p.y, p.z};
        std::memcpy(b, v, 12);
        for (unsigned char c : b) {
            h ^= c;
            h *= 0x100000001b3ull;
        }
        ++count;
    }
};

struct Options {
    int width = 1000, height = 500, spp = 100, noise = 1, octave = 4, band_lines = 25;
    std::string out_ppm, out_rgb;
    bool dry_run = false, time_kernel = false;
    size_t dump_n = 0;      // --dump-first N file: the first N hit points (xyz) and their texture values, raw float32
    std::string dump_file;
};

} // namespace

int main(int argc, char **argv)
{
    Options opt;
    for (int i = 1; i < argc; ++i) {
        auto next = [&]() { return (i + 1 < argc) ? argv[++i] : (char *)"0"; };
        if (!std::strcmp(argv[i], "--width")) opt.width = std::atoi(next());
        else if (!std::strcmp(argv[i], "--height")) opt.height = std::atoi(next());
        else if (!std::strcmp(argv[i], "--spp")) opt.spp = std::atoi(next());
        else if (!std::strcmp(argv[i], "--noise")) opt.noise = std::atoi(next());
        else if (!std::strcmp(argv[i], "--octave")) opt.octave = std::atoi(next());
        else if (!std::strcmp(argv[i], "--band-lines")) opt.band_lines = std::atoi(next());
        else if (!std::strcmp(argv[i], "--out")) opt.out_ppm = next();
        else if (!std::strcmp(argv[i], "--rgb")) opt.out_rgb = next();
        else if (!std::strcmp(argv[i], "--dry-run")) opt.dry_run = true;
        else if (!std::strcmp(argv[i], "--time-kernel")) opt.time_kernel = true;
        else if (!std::strcmp(argv[i], "--dump-first")) { opt.dump_n = (size_t)std::atoll(next()); opt.dump_file = next(); }
        else { std::fprintf(stderr, "render: unknown option %s\n", argv[i]); return 2; }
    }
    const int width = opt.width, height = opt.height, spp = opt.spp;
    const Scene scene;
    const V lower_left{-2, -1, -1}, origin{0, 0, 1}, horizontal{4, 0, 0}, vertical{0, 2, 0}; // main.cpp:126-130

    try {
        std::unique_ptr<wavelet_texture> wavelet;
        std::unique_ptr<noise_texture> perlin_tex;
        if (!opt.dry_run) {
            if (opt.noise == 1) wavelet = std::make_unique<wavelet_texture>(1.0, opt.octave, true); // main.cpp:61-70,144
            else perlin_tex = std::make_unique<noise_texture>(1.0, opt.octave);
        }
        std::vector<unsigned char> image((size_t)width * height * 3);
        Fnv stream;
        double t_trace = 0, t_noise = 0, t_kernel = 0;
        size_t total_points = 0, dumped = 0;

        // per band of scanlines: recorded bounces and per-sample bookkeeping
        std::vector<float> pts;        // xyz of every recorded bounce
        std::vector<float> grey;       // texture value per bounce (filled by the GPU)
        std::vector<uint32_t> first;   // first bounce index of every sample (+ one past the end)
        std::vector<V> terminal;       // what the path ended on: sky / emitted / black

        for (int j_top = height - 1; j_top >= 0; j_top -= opt.band_lines) {
            const int j_bot = std::max(0, j_top - opt.band_lines + 1);
            pts.clear();
            first.clear();
            terminal.clear();
            auto t0 = std::chrono::steady_clock::now();
            for (int j = j_top; j >= j_bot; --j) {
                for (int i = 0; i < width; ++i) {
                    for (int s = 0; s < spp; ++s) {
                        const float rand_u = float(std::rand()) / RAND_MAX - 0.5f;          // main.cpp:184-187
                        const float rand_v = float(std::rand()) / RAND_MAX - 0.5f;
                        const float u = float(i + 0.5f + rand_u) / width;
                        const float v = float(j + 0.5f + rand_v) / height;
                        Ray r{origin, unit(sub(add(add(lower_left, scale(u, horizontal)), scale(v, vertical)), origin))};
                        first.push_back((uint32_t)(pts.size() / 3));
                        V term{0, 0, 0};
                        for (int step = 0;; ++step) {                                       // main.cpp:38-59
                            if (step > kMaxDepth) { term = V{0, 0, 0}; break; }
                            Hit h;
                            if (!scene.hit(r, 0.001f, FLT_MAX, h)) {
                                const V ud = unit(r.d);
                                const float t = 0.5f * (ud.y + 1.0f);
                                term = add(scale(1.0f - t, V{1, 1, 1}), scale(t, V{0.40f, 0.50f, 1.00f}));
                                break;
                            }
                            if (h.surface == kLight) { term = V{4.0f, 4.0f, 4.0f}; break; } // diffuse_light::emitted
                            V dir = add(h.normal, random_unit_vector());                     // material.h:63-74
                            if (near_zero(dir)) dir = h.normal;
                            r = Ray{h.p, dir};
                            pts.push_back(h.p.x);
                            pts.push_back(h.p.y);
                            pts.push_back(h.p.z);
                            if (opt.dry_run) stream.add(h.p);
                        }
                        terminal.push_back(term);
                    }
                }
            }
            first.push_back((uint32_t)(pts.size() / 3));
            auto t1 = std::chrono::steady_clock::now();
            t_trace += std::chrono::duration<double>(t1 - t0).count();
            const size_t npts = pts.size() / 3;
            total_points += npts;
            if (opt.dry_run) continue;

            // ---- one batched texture evaluation for the whole band -----------------------------------
            grey.assign(npts, 0.0f);
            if (wavelet) wavelet->values(pts.data(), nullptr, npts, grey.data());
            else perlin_tex->values(pts.data(), nullptr, npts, grey.data());
            auto t2 = std::chrono::steady_clock::now();
            t_noise += std::chrono::duration<double>(t2 - t1).count();
            if (opt.dump_n && !opt.dump_file.empty() && dumped < opt.dump_n) { // for the parity test of the full-size stream
                const size_t take = std::min(opt.dump_n - dumped, npts);
                std::ofstream f(opt.dump_file, dumped ? std::ios::binary | std::ios::app : std::ios::binary);
                for (size_t k = 0; k < take; ++k) {
                    f.write(reinterpret_cast<const char *>(&pts[3 * k]), 3 * sizeof(float));
                    f.write(reinterpret_cast<const char *>(&grey[k]), sizeof(float));
                }
                dumped += take;
            }
            if (opt.time_kernel && npts) {
                wnhost::DeviceBuffer in(3 * npts * sizeof(float)), out(npts * sizeof(float));
                in.upload(pts.data());
                wn_timer *tm = nullptr;
                wnhost::check(wn_timer_create(&tm), "wn_timer_create");
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) { // the first launch warms the caches
                    wnhost::check(wn_timer_start(tm, nullptr), "wn_timer_start");
                    if (wavelet) wavelet->values_device(in.as<float>(), npts, out.as<float>());
                    else perlin_tex->values_device(in.as<float>(), npts, out.as<float>());
                    wnhost::check(wn_timer_stop(tm, nullptr), "wn_timer_stop");
                    float ms = 0;
                    wnhost::check(wn_timer_elapsed_ms(tm, &ms), "wn_timer_elapsed_ms");
                    if (rep) best = std::min(best, ms);
                }
                wn_timer_destroy(tm);
                t_kernel += best * 1e-3;
            }

            // ---- unwind the attenuation products deepest-first, accumulate in sample order ------------
            size_t sample = 0;
            for (int j = j_top; j >= j_bot; --j) {
                for (int i = 0; i < width; ++i) {
                    V sum{0, 0, 0};
                    for (int s = 0; s < spp; ++s, ++sample) {
                        V c = terminal[sample];
                        for (uint32_t k = first[sample + 1]; k-- > first[sample];) {
                            const float g = grey[k];
                            c = mulv(V{g, g, g}, c);                                         // attenuation * trace(...)
                        }
                        sum = add(sum, c);                                                   // color_sum += ...
                    }
                    const V c = divide(sum, float(spp));                                     // main.cpp:193-196
                    auto clamp01 = [](float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); };
                    const int rr = static_cast<int>(255.99 * clamp01(c.x));
                    const int gg = static_cast<int>(255.99 * clamp01(c.y));
                    const int bb = static_cast<int>(255.99 * clamp01(c.z));
                    const size_t idx = ((size_t)(height - 1 - j) * width + i) * 3;
                    image[idx + 0] = (unsigned char)rr;
                    image[idx + 1] = (unsigned char)gg;
                    image[idx + 2] = (unsigned char)bb;
                }
            }
        }

        if (opt.dry_run) {
            std::printf("WN_RECORD count=%llu fnv1a64=%016llx trace_s=%.2f\n", (unsigned long long)stream.count,
                        (unsigned long long)stream.h, t_trace);
            return 0;
        }
        if (!opt.out_ppm.empty()) {                                                          // main.cpp:171-172,197
            std::ofstream f(opt.out_ppm);
            f << "P3\n" << width << " " << height << "\n255\n";
            for (size_t p = 0; p < (size_t)width * height; ++p)
                f << (int)image[3 * p] << " " << (int)image[3 * p + 1] << " " << (int)image[3 * p + 2] << "\n";
        }
        if (!opt.out_rgb.empty()) {
            std::ofstream f(opt.out_rgb, std::ios::binary);
            f.write(reinterpret_cast<const char *>(image.data()), (std::streamsize)image.size());
        }
        std::printf("render: %dx%d spp %d noise %d octave %d: %zu noise evaluations (%.3f per primary ray), "
                    "host trace %.2f s, batched noise (incl. PCIe) %.3f s\n",
                    width, height, spp, opt.noise, opt.octave, total_points,
                    (double)total_points / ((double)width * height * spp), t_trace, t_noise);
        if (opt.time_kernel)
            std::printf("render: texture kernel alone on the renderer's hit-point stream: %.3f ms for %zu points = %.1f G points/s\n",
                        t_kernel * 1e3, total_points, (double)total_points / t_kernel / 1e9);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "render: %s\n", e.what());
        return 1;
    }
    return 0;
}
