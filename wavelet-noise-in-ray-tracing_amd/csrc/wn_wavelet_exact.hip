// wn_wavelet_exact.hip -- WN_GRID_EXACT dense 3-D grids with the coefficient box staged in LDS.
//
// grid3d_direct_kernel (wn_wavelet_grid.hip) evaluates WaveletNoise::evaluate3D (WaveletNoise.cpp:185-215)
// per sample with 9 three-tap gathers from the tile in global memory: 512^3 in 840-880 us, bound by the L1 / L2
// look-ups of the gathers.  A dense lattice is axis-aligned, so a brick of samples touches a small contiguous
// box of coefficients (periodic wrap resolved when the box is filled): this kernel stages that box in LDS once per
// brick and runs the SAME 27-tap loop on it -- same mids and weights, same products ((wx*wy)*wz), same
// accumulation order f2 -> f1 -> f0, unfused -- so every value has the bits of the reference.  Only where the
// coefficient comes from changes.
//
// A workgroup (4 waves) owns 256 x 8 x 8 samples; a lane owns 4 consecutive x samples (one float4 store per row),
// a wave 16 of the brick's 64 rows.  Lattices whose box does not fit (coarse steps) stay with the gather kernel.
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"

#include <algorithm>
#include <cmath>

namespace {

using wn::GridArgs;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kEX = 256, kEY = 8, kEZ = 8; // samples per brick
constexpr int kWaves = 4;
constexpr int kMaxBoxFloats = 12 * 1024;   // 48 KB of LDS: three workgroups per CU

struct ExactArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    int nbx, nby, nbz;
    int vec4_ok;
    int box_cap; // floats of dynamic LDS the launch was given
};

__global__ __launch_bounds__(64 * kWaves) void grid3d_exact_lds_kernel(const ExactArgs a)
{
    extern __shared__ float box[];
    __shared__ int s_geo[8]; // ix0, jy0, kz0, ex, ey, ez
    const GridArgs &g = a.g;
    const float den = (float)g.den;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    const int x_first = bx * kEX, y_first = by * kEY, z_first = bz * kEZ;
    auto coord = [&](int i) { return wn::lattice_coord(i, den, g.base_range, g.octave_scale, g.post_scale); };
    auto zcoord = [&](int zi) { return g.z_const_mode ? g.z_const : coord(g.z0 + zi); };

    // ---- the brick's box: coordinates are monotone in the index (either direction), so the mids of the first and
    // last sample of each axis bound all of them; one column / row / plane of support on either side
    if (tid < 3) {
        const int lo_i = tid == 0 ? x_first : (tid == 1 ? y_first : z_first);
        const int n_i = tid == 0 ? g.nx : (tid == 1 ? g.ny : g.nz);
        const int hi_i = min(lo_i + (tid == 0 ? kEX : (tid == 1 ? kEY : kEZ)), n_i) - 1;
        const float c_lo = tid == 2 ? zcoord(lo_i) : coord(lo_i), c_hi = tid == 2 ? zcoord(hi_i) : coord(hi_i);
        int m_lo, m_hi;
        float w0, w1, w2;
        wn::bspline(c_lo, m_lo, w0, w1, w2);
        wn::bspline(c_hi, m_hi, w0, w1, w2);
        s_geo[tid] = min(m_lo, m_hi) - 1;
        s_geo[3 + tid] = abs(m_hi - m_lo) + 3;
    }
    __syncthreads();
    const int ix0 = s_geo[0], jy0 = s_geo[1], kz0 = s_geo[2];
    const int ex = s_geo[3], ey = s_geo[4], ez = s_geo[5];
    if ((long long)ex * ey * ez > a.box_cap) return; // never: the host bounds the box (memory safety); uniform

    // ---- fill: box[k][j][i] = coef[Mod(kz0+k)][Mod(jy0+j)][Mod(ix0+i)] (WaveletNoise.cpp:31-34, 209); a wave
    // takes whole (k, j) rows, its lanes consecutive columns
    const int nrows = ey * ez;
    for (int r = wave; r < nrows; r += kWaves) {
        const int k = r / ey, j = r - k * ey;
        const float *row = a.coef + ((size_t)wn::dmod(kz0 + k, a.n, a.nmask) * a.n + wn::dmod(jy0 + j, a.n, a.nmask)) * a.n;
        for (int i = lane; i < ex; i += 64) box[r * ex + i] = row[wn::dmod(ix0 + i, a.n, a.nmask)];
    }

    // ---- x: this lane's 4 samples (WaveletNoise.cpp:194-200)
    const int x0 = x_first + lane * 4;
    int cx[4];
    float wx[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int m;
        wn::bspline(coord(min(x0 + q, g.nx - 1)), m, wx[q][0], wx[q][1], wx[q][2]);
        cx[q] = m - 1 - ix0; // box column of tap f0 = 0
    }
    __syncthreads();

    const int rows_y = min(kEY, g.ny - y_first), rows_z = min(kEZ, g.nz - z_first);
    for (int r = wave; r < rows_y * rows_z; r += kWaves) {
        const int yi = r % rows_y, zi = r / rows_y;
        int my, mz;
        float wy[3], wz[3];
        wn::bspline(coord(y_first + yi), my, wy[0], wy[1], wy[2]);
        wn::bspline(zcoord(z_first + zi), mz, wz[0], wz[1], wz[2]);
        const int jrow = my - 1 - jy0, krow = mz - 1 - kz0;
        float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // WaveletNoise.cpp:202-213: f2 outer, f1, f0 inner; weight = w[0][f0]*w[1][f1]*w[2][f2]; result += weight*coef
#pragma unroll
        for (int fz = 0; fz < 3; ++fz)
#pragma unroll
            for (int fy = 0; fy < 3; ++fy) {
                const float *rowp = box + ((krow + fz) * ey + (jrow + fy)) * ex;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float *c = rowp + cx[q];
#pragma unroll
                    for (int fx = 0; fx < 3; ++fx) {
                        const float weight = wx[q][fx] * wy[fy] * wz[fz];
                        res[q] += weight * c[fx];
                    }
                }
            }
        float *dst = a.out + ((size_t)(z_first + zi) * g.ny + (y_first + yi)) * g.nx + x0;
        if (a.vec4_ok && x0 + 3 < g.nx) {
            *reinterpret_cast<v4f *>(dst) = v4f{res[0] * g.out_scale, res[1] * g.out_scale, res[2] * g.out_scale, res[3] * g.out_scale};
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (x0 + q < g.nx) dst[q] = res[q] * g.out_scale;
        }
    }
}

} // namespace

namespace wn {

// Launches the LDS-staged exact kernel when every brick's coefficient box fits; *launched tells the caller.
int exact_lds_try(const wn_tile *tile, const GridArgs &g, float *out_dev, hipStream_t stream, bool *launched)
{
    *launched = false;
    if (tile->n == 0 || g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return WN_OK;
    if (g.nx < 64) return WN_OK; // narrow grids: most of a 256-sample brick would idle
    const double step = fabs((double)g.base_range * (double)g.octave_scale * (double)g.post_scale / g.den);
    if (!std::isfinite(step)) return WN_OK;
    const double imax = std::max(std::max((double)g.nx, (double)g.ny), fabs((double)g.z0) + g.nz);
    const double pmax = step * imax + fabs((double)g.z_const) + 1.0;
    if (pmax > 1.0e6) return WN_OK; // keep mids far inside int / float-exact range
    const double slack = pmax * 4.8e-7 + 1.0; // rounding of a coordinate, and one cell of margin
    auto extent = [&](int samples) { return (long long)floor((samples - 1) * step + slack) + 1 + 3; };
    const long long ez = g.z_const_mode ? 3 : extent(kEZ);
    const long long box_floats = extent(kEX) * extent(kEY) * ez;
    if (box_floats > kMaxBoxFloats) return WN_OK; // coarse lattice: the gather kernel
    ExactArgs a{};
    a.coef = tile->dev;
    a.out = out_dev;
    a.n = tile->n;
    a.nmask = (tile->n > 0 && (tile->n & (tile->n - 1)) == 0) ? tile->n - 1 : -1;
    a.g = g;
    a.nbx = (g.nx + kEX - 1) / kEX;
    a.nby = (g.ny + kEY - 1) / kEY;
    a.nbz = (g.nz + kEZ - 1) / kEZ;
    if (a.nby > 65535 || a.nbz > 65535) return WN_OK;
    a.vec4_ok = (g.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    a.box_cap = (int)box_floats;
    const size_t lds = (size_t)box_floats * sizeof(float); // <= 48 KB: no opt-in needed
    hipLaunchKernelGGL(grid3d_exact_lds_kernel, dim3(a.nbx, a.nby, a.nbz), dim3(64 * kWaves), lds, stream, a);
    WN_LAUNCH_CHECK("grid3d_exact_lds_kernel");
    *launched = true;
    return WN_OK;
}

} // namespace wn
