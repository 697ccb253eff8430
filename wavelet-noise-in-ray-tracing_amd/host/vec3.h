// vec3.h -- minimal float 3-vector so the host headers stand alone.
//
// The reference's vec3 (vec3.h:8-143) is a FLOAT vector (float e[3]) and that float-ness is part
// of the hot path's numeric contract: texture.h multiplies points by scale factors through
// vec3*float.  Only what the noise/texture headers need is provided here; when these headers are
// dropped into the reference tree its own vec3.h is the one that is found.
#ifndef VEC3H
#define VEC3H

#include <cmath>

class vec3 {
  public:
    vec3() : e{0.0f, 0.0f, 0.0f} {}
    vec3(float e0, float e1, float e2) : e{e0, e1, e2} {}
    float x() const { return e[0]; }
    float y() const { return e[1]; }
    float z() const { return e[2]; }
    float operator[](int i) const { return e[i]; }
    float &operator[](int i) { return e[i]; }
    vec3 &operator*=(const float t)
    {
        e[0] *= t;
        e[1] *= t;
        e[2] *= t;
        return *this;
    }
    float e[3];
};

inline vec3 operator*(const vec3 &v, float t) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator*(float t, const vec3 &v) { return vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }

#endif
