// wn_perlin_run.hpp -- the gradient bookkeeping of perlin_grid_run_kernel (wn_perlin.hip), kept
// host-compilable so that tests/test_perlin_run_host.py can check it against a literal
// restatement of grad() (perlin.h:26-31) on the CPU, for every hash and both zero signs.
#pragma once

#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define WN_HD __host__ __device__ __forceinline__
#else
#define WN_HD inline
#endif

namespace wn {

// grad(h, dx, dy, dz) = ((h&1)==0 ? u : -u) + ((h&2)==0 ? v : -v), u = h<8 ? x : y,
// v = h<4 ? y : (h==12||h==14 ? x : z).  Along a run of x samples (dy, dz fixed) this is
//     (x term | nothing) + K,   K = P + Q  with P in {+-dy, -0.0}, Q in {+-dz, -0.0}:
// -0.0 is the identity of IEEE addition for every operand, both zeros included, so the one
// rounded addition the reference performs is reproduced and zeros keep their signs.
struct RunKEntry {
    double K;
    uint32_t mm; // all ones when the gradient has an x term
    uint32_t t;  // xor for the high dword: sign of the x term; the sign bit of -0.0 when there is none
};

WN_HD RunKEntry run_k_entry(int h, double dy, double dz)
{
    const bool has_x = h < 8 || h == 12 || h == 14;
    const bool x_negative = h < 8 ? (h & 1) != 0 : (h & 2) != 0;
    double P, Q;
    if (h < 4) { P = (h & 2) ? -dy : dy; Q = -0.0; }            // u = x, v = y
    else if (h < 8) { P = -0.0; Q = (h & 2) ? -dz : dz; }       // u = x, v = z
    else if (h == 12 || h == 14) { P = dy; Q = -0.0; }          // u = y (h & 1 == 0), v = x
    else { P = (h & 1) ? -dy : dy; Q = (h & 2) ? -dz : dz; }    // u = y, v = z
    RunKEntry e;
    e.K = P + Q;
    e.mm = has_x ? 0xffffffffu : 0u;
    e.t = has_x ? (x_negative ? 0x80000000u : 0u) : 0x80000000u;
    return e;
}

WN_HD double run_gradient(double K, uint32_t mm, uint32_t t, uint64_t dx_bits)
{
    const uint32_t lo = (uint32_t)dx_bits & mm;
#if defined(__HIP_DEVICE_COMPILE__)
    // (hi & mm) ^ t as ONE v_bitop3_b32 (truth table index = S0<<2 | S1<<1 | S2 -> 0x6a); written as plain
    // C the optimiser splits the xor into sign and magnitude halves: four instructions instead of one.
    // The halves are joined with __hiloint2double: a 64-bit shift-and-or leaves a byte-wise re-assembly
    // (v_and 0xffffff00 + v_or_b32_sdwa) of the high dword behind.
    const uint32_t hi = __builtin_amdgcn_bitop3_b32((uint32_t)(dx_bits >> 32), mm, t, 0x6a);
    return __hiloint2double((int)hi, (int)lo) + K;
#else
    const uint32_t hi = ((uint32_t)(dx_bits >> 32) & mm) ^ t;
    const uint64_t b = ((uint64_t)hi << 32) | lo;
    double a;
    memcpy(&a, &b, sizeof(a));
    return a + K;
#endif
}

} // namespace wn
