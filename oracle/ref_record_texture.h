// ref_record_texture.h -- TEST INFRASTRUCTURE.  A stand-in for texture.h used ONLY to build an
// instrumented copy of the reference's ray tracer (oracle/Makefile target _ref/raytrace_record):
// the reference's unmodified main.cpp is compiled with this header in texture.h's place, so every
// tex->value(u,v,p) call the renderer makes is counted and its point folded into an FNV-1a hash
// instead of being evaluated (noise values never steer the reference's control flow or its rand()
// stream: material.h:72, main.cpp:54-55).  The count/hash pin the hit-point stream that
// tools/render.cpp must reproduce before a single noise value is involved.
#ifndef TEXTURE_H
#define TEXTURE_H
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include "vec3.h"

using color = vec3;
using point3 = vec3;

struct wn_point_recorder {
    uint64_t count = 0, hash = 0xcbf29ce484222325ull;
    void add(const point3 &p)
    {
        const float v[3] = {p.x(), p.y(), p.z()};
        unsigned char b[12];
        std::memcpy(b, v, 12);
        for (unsigned char c : b) {
            hash ^= c;
            hash *= 0x100000001b3ull;
        }
        ++count;
    }
    ~wn_point_recorder() { std::fprintf(stderr, "WN_RECORD count=%llu fnv1a64=%016llx\n", (unsigned long long)count, (unsigned long long)hash); }
};
inline wn_point_recorder &wn_recorder()
{
    static wn_point_recorder r;
    return r;
}

class texture {
  public:
    virtual ~texture() = default;
    virtual color value(double u, double v, const point3 &p) const = 0;
};
class solid_color : public texture {
  public:
    solid_color(const color &albedo) : albedo(albedo) {}
    solid_color(double r, double g, double b) : solid_color(color(r, g, b)) {}
    color value(double, double, const point3 &) const override { return albedo; }
  private:
    color albedo;
};
class noise_texture : public texture {
  public:
    noise_texture(double, int = 4) {}
    color value(double, double, const point3 &p) const override { wn_recorder().add(p); return color(0.5f, 0.5f, 0.5f); }
};
class wavelet_texture : public texture {
  public:
    wavelet_texture(double = 1.0, int = 4, bool = true) {}
    color value(double, double, const point3 &p) const override { wn_recorder().add(p); return color(0.5f, 0.5f, 0.5f); }
};
#endif
