// wn_wavelet_grid.hip -- dense-grid wavelet noise for gfx950 (K1/K3/K3p/K5 of SURVEY.md 8).
//
// These kernels stand under wn_eval3d_grid / wn_multiband3d_grid (tried in this order; wn_eval3d_grid first offers
// the lattice to the strip-march kernel of wn_wavelet_strip.hip):
//
//  * grid3d_sep_kernel<NB, XW>  (default).  A dense lattice is axis-aligned, so the 27-tap sum of
//    WaveletNoise::evaluate3D (WaveletNoise.cpp:202-213) factors per axis:
//        out[x,y,z] = sum_i Wx[x,i] * ( sum_j Wy[y,j] * ( sum_k Wz[z,k] * C[i,j,k] ) ).
//    Persistent workgroups (4*XW waves) walk bricks of 256*XW x 8 x BZ samples.  Per brick the
//    coefficient box (periodic wrap resolved) is staged through LDS, y and z are collapsed for
//    every sample row into LDS rows R[row][i] (9 FMAs per coefficient column), then each lane
//    produces 4 consecutive x samples per row from a 4-wide window of R with its 16 window
//    weights held in registers, and stores one float4: every wave store is 1 KiB contiguous.
//    Per-axis weights/mids are computed exactly as the reference does; only the order of the
//    final sums differs (measured <= 1.5e-6 abs; tolerance 1e-5).  NB > 1 accumulates NB bands
//    in-kernel (Cook & DeRose WMultibandNoise) with one store.
//    Bound: HBM write stream, 4 B/sample (+ the 8 MiB tile, read ~twice, L2/MALL resident).
//
//  * WN_GRID_EXACT, or lattices the brick scheme does not cover (step > 1/3 cell per sample, negative steps):
//    grid3d_exact_lds_kernel (wn_wavelet_exact.hip: the reference's 27-tap loop on an LDS-staged box) when a
//    brick's box fits LDS, else grid3d_direct_kernel: one sample per lane, gathers from the tile.  Both keep
//    the reference's loop order and unfused arithmetic: bit-identical to evaluate3D.
//
// 2-D and projected grids use direct kernels (bit-identical to evaluate2D / evaluate3DProjected).
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

using wn::GridArgs;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kMaxBands = 8;
constexpr int kBrickY = 8;    // sample rows per brick along y
constexpr int kMaxBZ = 16;    // ... and along z (power of two: 16 for fine single-band lattices, 8, fewer for thin slabs)
constexpr int kSlotZ = 8, kSlotX = kSlotZ + kMaxBZ; // table slots: 0..7 y samples, 8..23 z samples, 24/25 first/last x sample
constexpr int kBoxY = 6;      // coefficient box: at most 6 rows in y and in z (8 samples, step <= 1/3)
constexpr int kBoxZ = 6;
constexpr int kColStride = kBoxY * kBoxZ + 1; // LDS floats per box column (odd: conflict-free fills)

struct BandArgs {
    float oscale; // octave_scale of this band
    float factor; // everything that scales this band's contribution: w[b] * out_scale / out_div
    int box_off;  // float offset of this band's coefficient box in dynamic LDS
    int r_off;    // float offset of this band's collapsed rows
    int ex_cap;   // columns the LDS regions were sized for (the kernel never exceeds it)
};

struct SepArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    int bz_log2;
    int box_buf_stride; // floats between the two copies of the box region (double buffering)
    int r_buf_stride;   // ... and of the R region (single band only, else 0)
    int nbx, nby, nbz;
    int nbands;
    float inv_den; // 1/den when den is a power of two (exact), else 0
    int vec4_ok;
    int even_share_q10;
    int xw; // wave columns per brick (1 or 2)
    BandArgs band[kMaxBands];
};

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{}).
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// Bands of one multiband call are consecutive octaves: the band d below the top one has a step <= (1/3) / 2^d, so
// its coefficient box has at most box_rows_bound(d) (k, j) rows and spans at most box_col_groups(d, brick_x) groups
// of 64 columns (the host checks both).  A single band is d = 0.
__host__ __device__ constexpr int box_rows_bound(int d) { return d == 0 ? 36 : (d == 1 ? 25 : 16); }
__host__ __device__ constexpr int box_col_groups(int d, int brick_x)
{
    return ((brick_x / 3 >> (d > 4 ? 4 : d)) + 8 + 63) / 64;
}

// lattice_coord with the division replaced by an exact multiply when den is a power of two.
__device__ __forceinline__ float lattice_coord_fast(int i, float den, float inv_den, float range,
                                                    float oscale, float post)
{
    const float fi = (float)i;
    float c = ((inv_den != 0.0f) ? fi * inv_den : fi / den) * range;
    c = c * oscale;
    c = c * post;
    return c;
}

// Persistent workgroups: each of the gridDim.x workgroups owns a contiguous range of bricks
// (brick id = bx * (nby*nbz) + by + nby*bz, so a range mostly keeps bx, and with it the lane's
// x-window weights, fixed) and runs a two-barrier pipeline per brick:
//     [write prefetched coefficient box to LDS | tables of the next brick]   barrier
//     [phase 1: collapse y,z -> R rows]                                      barrier
//     [issue the next brick's coefficient loads | phase C: x-window -> float4 stores]
// LDS images per band:
//     box[i][k][j]  column-major coefficient box, i = x column, fixed strides (kColStride, kBoxY)
//                   so that a row's 9 (k,j) taps are immediate offsets from one address;
//     R[row][i]     collapsed rows, odd row stride.
// XW = wave columns per brick: the brick is XW*256 samples wide and the workgroup has 4*XW waves;
// wave (xw, wr) = (wave / 4, wave % 4) stores rows wr, wr+4, ... of the xw-th 256-sample column, so
// with XW = 2 the two halves of a 2-KiB output row are written by sibling waves at the same time.
template <int NB, int XW>
__global__ __launch_bounds__(256 * XW) void grid3d_sep_kernel(const SepArgs a)
{
    extern __shared__ float lds[];
    // per-band sample tables, triple buffered: slots 0..7 y samples, kSlotZ.. z samples,
    // kSlotX / kSlotX+1 first/last x sample of the brick
    __shared__ int s_mid[3][NB][32];
    __shared__ float s_w[3][NB][kSlotX][3];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kWaves = 4 * XW, kBrickX = 256 * XW;
    const int xw = wave >> 2, wr = wave & 3;
    const int BZ = 1 << a.bz_log2, rows = kBrickY * BZ;
    const GridArgs &g = a.g;
    const float den = (float)g.den;
    const int nyz = a.nby * a.nbz;

    const long long total_items = (long long)a.nbx * nyz;
    // Workgroups take their bricks in dispatch order.  (Hardware deals consecutive ids round-robin over the 8 XCDs; round 1 gave
    // each XCD one contiguous eighth of the volume instead -- one L2 per part of the tile.  Measured in round 2: no gain, between
    // 2 % slower and 1 % faster over six lattices on two boxes.  The tile reads it saves come out of the Infinity Cache beside the
    // store stream and cost no time, as in the strip kernel: profiles/r02_strip_xcd_renumbering.txt.)
    const int wg = (int)blockIdx.x;
    // Workgroups 2k, 2k+1 (an even and an odd XCD) share their bricks unevenly: even_share_q10 / 1024 goes to the even one.
    // In a store-bound launch every odd workgroup ends ~10 % later than its even neighbour with equal shares
    // (profiles/r02_strip_pair_timestamps.txt); 53 : 47 for single-band lattices measured 1.3-2.2 % faster on two boxes
    // (1024^3 812 -> 794 us, 2048 x 2048 x 256 721 -> 711, 768^3 386 -> 381); five bands (not store-bound) keep halves.
    auto first_item = [&](int w) { return (int)((total_items * ((long long)(w & ~1) * 512 + ((w & 1) ? a.even_share_q10 : 0))) / ((long long)gridDim.x * 512)); };
    int item = (gridDim.x & 1) ? (int)(total_items * wg / gridDim.x) : first_item(wg);
    const int item_end = (gridDim.x & 1) ? (int)(total_items * (wg + 1) / gridDim.x) : (wg + 1 == (int)gridDim.x ? (int)total_items : first_item(wg + 1));
    if (item >= item_end) return;
    int bx = item / nyz;
    int bz = (item - bx * nyz) / a.nby;
    int by = item - bx * nyz - bz * a.nby;

    auto fill_tables = [&](int buf, int tbx, int tby, int tbz) {
        const int b = tid >> 5, slot = tid & 31;
        if (b < NB && slot < kSlotX + 2) {
            int idx;
            bool is_const = false;
            if (slot < 8) idx = min(tby * kBrickY + slot, g.ny - 1);
            else if (slot < kSlotX) {
                idx = g.z0 + min(tbz * BZ + min(slot - kSlotZ, BZ - 1), g.nz - 1);
                is_const = g.z_const_mode != 0;
            } else idx = (slot == kSlotX) ? tbx * kBrickX : min(tbx * kBrickX + kBrickX - 1, g.nx - 1);
            // band scale by selects: a per-lane index into the kernel arguments would become a
            // vector-memory load, and its vmcnt(0) wait would queue behind this wave's stores
            float oscale = a.band[0].oscale;
#pragma unroll
            for (int bb = 1; bb < NB; ++bb) oscale = (b == bb) ? a.band[bb].oscale : oscale;
            const float c = is_const ? g.z_const
                                     : lattice_coord_fast(idx, den, a.inv_den, g.base_range, oscale,
                                                          g.post_scale);
            int m;
            float w0, w1, w2;
            wn::bspline(c, m, w0, w1, w2);
            s_mid[buf][b][slot] = m;
            if (slot < kSlotX) {
                s_w[buf][b][slot][0] = w0;
                s_w[buf][b][slot][1] = w1;
                s_w[buf][b][slot][2] = w2;
            }
        }
    };

    // coefficient-box geometry of a brick, read back from its tables into SGPRs
    struct Box { int ix0, jy0, kz0, ex, ey, nrows; };
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto box_of = [&](int buf, int b) {
        Box o;
        const int mx0 = sgpr(s_mid[buf][b][kSlotX]), mx1 = sgpr(s_mid[buf][b][kSlotX + 1]);
        const int my0 = sgpr(s_mid[buf][b][0]), my1 = sgpr(s_mid[buf][b][kBrickY - 1]);
        const int mz0 = sgpr(s_mid[buf][b][kSlotZ]), mz1 = sgpr(s_mid[buf][b][kSlotZ + BZ - 1]);
        o.ix0 = mx0 - 1;
        o.ex = min(mx1 - mx0 + 4, a.band[b].ex_cap); // +3 support, +1 pad column (zero-weight tap)
        o.jy0 = my0 - 1;
        o.ey = min(my1 - my0 + 3, kBoxY); // the clamps never bind (plan_sep bounds the extents): memory safety
        o.kz0 = mz0 - 1;
        o.nrows = min(mz1 - mz0 + 3, kBoxZ) * o.ey;
        return o;
    };

    // Box rows are (k, j) pairs, r = k*ey + j; wave w takes rows w, w+kWaves, ...: the row part of every
    // address is scalar, a lane contributes only its column (256-B coalesced reads).
    // Round 2: the bookkeeping of this phase was 40-50 % of the multiband kernel's instructions (~35 scalar and
    // vector instructions per load: per-row while loops, 64-bit address arithmetic, a branch per load).  Now lane r
    // computes row r's source offset and LDS slot ONCE per box (all rows in one instruction stream), a wave picks its
    // rows' values with v_readlane; loads are unconditional with clamped indices (a few redundant, never stored);
    // and rows / column groups per band are compile-time bounds (the lower bands' boxes are small).
    constexpr int kRowsPerWave = (kBoxY * kBoxZ + kWaves - 1) / kWaves;
    constexpr int kColGroups = (kBrickX / 3 + 8 + 63) / 64; // 64-column groups a box can span (step <= 1/3)
    struct RowMap { int src, at; }; // per lane r: float offset of box row r in the tile, its slot k*kBoxY + j
    auto row_map = [&](const Box &o) -> RowMap {
        const int r = min(lane, o.nrows - 1);
        int k = 0;
#pragma unroll
        for (int m = 1; m < kBoxZ; ++m) k += (r >= m * o.ey) ? 1 : 0;
        const int j = r - k * o.ey;
        RowMap rm;
        rm.src = (((o.kz0 + k) & a.nmask) * a.n + ((o.jy0 + j) & a.nmask)) * a.n;
        rm.at = k * kBoxY + j;
        return rm;
    };
    // D = the band's depth below the top band (compile time): rows per wave and column groups
    auto issue_rows = [&](const Box &o, const RowMap &rm, float (*pfr)[kColGroups], auto dc) {
        constexpr int D = decltype(dc)::value;
        constexpr int RW = (box_rows_bound(D) + kWaves - 1) / kWaves, CG = box_col_groups(D, kBrickX);
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = min(wave + kWaves * t, o.nrows - 1); // past the end: the last row again (not stored)
            const float *row = a.coef + __builtin_amdgcn_readlane(rm.src, r);
#pragma unroll
            for (int c = 0; c < CG; ++c) pfr[t][c] = row[(o.ix0 + min(64 * c + lane, o.ex - 1)) & a.nmask];
        }
    };
    auto commit_rows = [&](const Box &o, const RowMap &rm, int b, int box_buf, const float (*pfr)[kColGroups], auto dc) {
        constexpr int D = decltype(dc)::value;
        constexpr int RW = (box_rows_bound(D) + kWaves - 1) / kWaves, CG = box_col_groups(D, kBrickX);
        float *col = lds + a.band[b].box_off + box_buf * a.box_buf_stride + lane * kColStride;
#pragma unroll
        for (int t = 0; t < RW; ++t) {
            const int r = wave + kWaves * t;
            if (r < o.nrows) {
                const int at = __builtin_amdgcn_readlane(rm.at, r);
#pragma unroll
                for (int c = 0; c < CG; ++c)
                    if (64 * c + lane < o.ex) col[64 * c * kColStride + at] = pfr[t][c];
            }
        }
    };
    // single band (NB == 1): one box in flight across phase 1
    float pf[kRowsPerWave][kColGroups];
    RowMap pf_map;
    auto issue_box = [&](int buf, int b) -> Box {
        const Box o = box_of(buf, b);
        pf_map = row_map(o);
        issue_rows(o, pf_map, pf, std::integral_constant<int, 0>{});
        return o; // geometry stays in SGPRs for commit_box
    };
    auto commit_box = [&](const Box &o, int b, int box_buf) {
        commit_rows(o, pf_map, b, box_buf, pf, std::integral_constant<int, 0>{});
    };

    // this lane's 4 x samples: window start (as a column of R) and 16 window weights per band;
    // recomputed only when bx moves
    float ww[NB][4][4];
    int wbase[NB];
    int x0 = 0;
    auto x_weights = [&](int buf) {
        x0 = bx * kBrickX + xw * 256 + lane * 4;
        float xbase[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float xi = (float)min(x0 + q, g.nx - 1);
            xbase[q] = ((a.inv_den != 0.0f) ? xi * a.inv_den : xi / den) * g.base_range;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int m[4];
            float w[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c = xbase[q] * a.band[b].oscale;
                c = c * g.post_scale;
                wn::bspline(c, m[q], w[q][0], w[q][1], w[q][2]);
            }
            wbase[b] = m[0] - s_mid[buf][b][kSlotX]; // (m0 - 1) - ix0
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool up = m[q] != m[0]; // mid is m[0] or m[0]+1 (host guarantees step <= 1/3)
                ww[b][q][0] = up ? 0.0f : w[q][0];
                ww[b][q][1] = up ? w[q][0] : w[q][1];
                ww[b][q][2] = up ? w[q][1] : w[q][2];
                ww[b][q][3] = up ? w[q][2] : 0.0f;
            }
        }
    };

    // ---- phases ------------------------------------------------------------------------------------
    int rs_[NB]; // R row strides of the brick in flight (odd: conflict-free column writes)

    // phase 1: collapse y and z.  Lane = sample row (its 9 yz-weights in registers), the waves
    // split the coefficient columns:
    //     R[row][i] = out_scale * sum_k sum_j (wz[k]*wy[j]) * C[kz+k][jy+j][i]
    auto phase1 = [&](int tb, int box_buf, int r_buf) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            rs_[b] = min(sgpr(s_mid[tb][b][kSlotX + 1]) - sgpr(s_mid[tb][b][kSlotX]) + 4, a.band[b].ex_cap) | 1;
        for (int row = lane; row < rows; row += 64) { // 64 rows per pass (16-plane bricks: two passes)
            const int yi = row & (kBrickY - 1), zi = row >> 3;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int ex = min(sgpr(s_mid[tb][b][kSlotX + 1]) - sgpr(s_mid[tb][b][kSlotX]) + 4, a.band[b].ex_cap);
                const int kz = s_mid[tb][b][kSlotZ + zi] - s_mid[tb][b][kSlotZ];  // row's first box row in z
                const int jy = s_mid[tb][b][yi] - s_mid[tb][b][0];                // ... and in y
                float w9[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int j = 0; j < 3; ++j) w9[k][j] = s_w[tb][b][kSlotZ + zi][k] * s_w[tb][b][yi][j];
                const int chunk = (ex + kWaves - 1) / kWaves;
                const int i_begin = wave * chunk, i_end = min(ex, i_begin + chunk);
                const float *c = lds + a.band[b].box_off + box_buf * a.box_buf_stride + kz * kBoxY + jy +
                                 i_begin * kColStride;
                float *R = lds + a.band[b].r_off + r_buf * a.r_buf_stride + row * rs_[b] + i_begin;
                // several columns per trip: their LDS reads are in flight together (at two waves per SIMD a column
                // per trip waited out its own read latency)
#pragma unroll 4
                for (int i = i_begin; i < i_end; ++i, c += kColStride, ++R) {
                    float acc = w9[0][0] * c[0];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int j = 0; j < 3; ++j)
                            if (k | j) acc = __builtin_fmaf(w9[k][j], c[k * kBoxY + j], acc);
                    *R = acc * a.band[b].factor;
                }
            }
        }
    };

    // phase C: x from a 4-wide window of R, one float4 per lane per row
    auto phaseC = [&](int r_buf) {
        const int y_base = by * kBrickY, z_base = bz * BZ;
        const bool full = (y_base + kBrickY <= g.ny) && (z_base + BZ <= g.nz) &&
                          (bx * kBrickX + kBrickX <= g.nx) && a.vec4_ok;
        const float *Rl[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            Rl[b] = lds + a.band[b].r_off + r_buf * a.r_buf_stride + wbase[b] + wr * rs_[b];
        auto row_values = [&](float acc[4]) {
            acc[0] = acc[1] = acc[2] = acc[3] = 0.0f;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float v0 = Rl[b][0], v1 = Rl[b][1], v2 = Rl[b][2], v3 = Rl[b][3];
                Rl[b] += 4 * rs_[b];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // the band factor (weight, 1/sqrt(variance), out_scale) is already in R: one FMA
                    // chain over taps and bands, no per-sample scaling or division
                    float t = (b == 0) ? ww[b][q][0] * v0 : __builtin_fmaf(ww[b][q][0], v0, acc[q]);
                    t = __builtin_fmaf(ww[b][q][1], v1, t);
                    t = __builtin_fmaf(ww[b][q][2], v2, t);
                    t = __builtin_fmaf(ww[b][q][3], v3, t);
                    acc[q] = t;
                }
            }
        };
        // A wave owns rows wr, wr+4, ...: per z plane the y rows (wr) and (wr + 4).
        const size_t plane = (size_t)g.ny * g.nx;
        float *zrow = a.out + ((size_t)z_base * g.ny + y_base + wr) * g.nx + (size_t)bx * kBrickX + xw * 256;
        if (full) {
            // interior brick: every wave store is one 1-KiB global_store_dwordx4
            auto two_rows = [&]() {
                float acc[4];
                row_values(acc);
                reinterpret_cast<v4f *>(zrow)[lane] = v4f{acc[0], acc[1], acc[2], acc[3]};
                row_values(acc);
                reinterpret_cast<v4f *>(zrow + 4 * (size_t)g.nx)[lane] = v4f{acc[0], acc[1], acc[2], acc[3]};
                zrow += plane;
            };
            if (a.bz_log2 == 3) { // the common cases, straight-line: LDS reads of later rows are hoisted over earlier FMAs
#pragma unroll
                for (int zi = 0; zi < 8; ++zi) two_rows();
            } else if (a.bz_log2 == 4) {
#pragma unroll
                for (int zi = 0; zi < 16; ++zi) two_rows();
            } else {
                for (int zi = 0; zi < BZ; ++zi) two_rows();
            }
        } else {
            for (int zi = 0; zi < BZ; ++zi, zrow += plane) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int y = y_base + wr + 4 * h, z = z_base + zi;
                    float acc[4];
                    row_values(acc);
                    if (y < g.ny && z < g.nz) {
                        float *dst = zrow + (size_t)(4 * h) * g.nx + lane * 4;
                        if (a.vec4_ok && x0 + 3 < g.nx) {
                            *reinterpret_cast<v4f *>(dst) = v4f{acc[0], acc[1], acc[2], acc[3]};
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (x0 + q < g.nx) dst[q] = acc[q];
                        }
                    }
                }
            }
        }
    };

    auto next_brick = [&](int &nx_, int &ny_, int &nz_) {
        if (++ny_ == a.nby) { ny_ = 0; ++nz_; }
        if (nz_ == a.nbz) { nz_ = 0; ++nx_; }
    };

    // ---- pipeline ---------------------------------------------------------------------------------
    // Tables are triple buffered (brick t in slot t%3).
    // NB == 1: R and the coefficient box are double buffered and a brick costs ONE barrier:
    //     [issue loads of box(t+1)] [phase 1(t) -> R[t&1]] [commit box(t+1)] [tables(t+2)]
    //     barrier
    //     [phase C(t): R[t&1] -> stores]
    //   The loads of brick t+1 are in flight during phase 1 and retired BEFORE brick t's stores are
    //   issued, so the vmcnt wait that retires them only has brick t-1's stores ahead of it; waves
    //   that finish their stores early run ahead into phase 1 of the next brick.
    // NB > 1: single R and box (LDS and registers are the scarce resources there), boxes loaded
    //   band by band at the top of the brick: two barriers per brick.
    int tb = 0, par = 0;
    fill_tables(0, bx, by, bz);
    int n1x = bx, n1y = by, n1z = bz; // brick t+1
    next_brick(n1x, n1y, n1z);
    if (item + 1 < item_end) fill_tables(1, n1x, n1y, n1z);
    __syncthreads();
    if (NB == 1) {
        const Box o0 = issue_box(0, 0);
        commit_box(o0, 0, 0);
    }
    x_weights(0);
    int weights_bx = bx;
    if (NB == 1) __syncthreads();

    for (;;) {
        const bool has_next = item + 1 < item_end;
        const int tb1 = (tb == 2) ? 0 : tb + 1, tb2 = (tb1 == 2) ? 0 : tb1 + 1;
        int n2x = n1x, n2y = n1y, n2z = n1z; // brick t+2
        next_brick(n2x, n2y, n2z);

        if (NB == 1) {
            Box onext{};
            if (has_next) onext = issue_box(tb1, 0);
            phase1(tb, par, par);
            if (has_next) commit_box(onext, 0, par ^ 1);
            if (item + 2 < item_end) fill_tables(tb2, n2x, n2y, n2z);
            __syncthreads();
            phaseC(par);
        } else {
            // several bands: all bands' loads in flight together (29 registers for 5 bands of a 512-wide brick), then
            // all commits.  (Two copies of the boxes with brick t+1's loads in flight during phase 1 of brick t were
            // tried twice -- with the old and with this lean box code: 285 and 245-253 us against 261 and 225.)
            float pfb[NB][kRowsPerWave][kColGroups];
            Box ob[NB];
            RowMap mapb[NB];
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                ob[b] = box_of(tb, b);
                mapb[b] = row_map(ob[b]);
                issue_rows(ob[b], mapb[b], pfb[b], std::integral_constant<int, NB - 1 - b>{});
            });
            static_for<NB>([&](auto bc) {
                constexpr int b = decltype(bc)::value;
                commit_rows(ob[b], mapb[b], b, 0, pfb[b], std::integral_constant<int, NB - 1 - b>{});
            });
            __syncthreads(); // boxes complete; previous phase C done (R free)
            phase1(tb, 0, 0);
            if (item + 2 < item_end) fill_tables(tb2, n2x, n2y, n2z);
            __syncthreads(); // R complete; the boxes may be overwritten
            phaseC(0);
        }

        if (!has_next) break;
        ++item;
        bx = n1x; by = n1y; bz = n1z;
        n1x = n2x; n1y = n2y; n1z = n2z;
        tb = tb1;
        par ^= 1;
        if (bx != weights_bx) { // tables of the new brick became visible at least one barrier ago
            x_weights(tb);
            weights_bx = bx;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct kernels: the reference's loops, one sample per lane, unfused -> bit-identical.
// ------------------------------------------------------------------------------------------------
struct DirectArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    // multiband (nbands == 0: plain evaluate3D)
    int nbands;
    float band_scale[kMaxBands]; // 2^(first_band+b)
    float band_w[kMaxBands];
    float out_div;
    int apply_div;
};

template <bool PADDED>
__global__ __launch_bounds__(256) void grid3d_direct_kernel(const DirectArgs a)
{
    const GridArgs &g = a.g;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    const float den = (float)g.den;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.nx);
        const size_t r = e / g.nx;
        const int y = (int)(r % g.ny), z = (int)(r / g.ny);
        float v;
        if (a.nbands == 0) {
            const float px = wn::lattice_coord(x, den, g.base_range, g.octave_scale, g.post_scale);
            const float py = wn::lattice_coord(y, den, g.base_range, g.octave_scale, g.post_scale);
            const float pz = g.z_const_mode ? g.z_const
                                            : wn::lattice_coord(g.z0 + z, den, g.base_range,
                                                                g.octave_scale, g.post_scale);
            v = wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, px, py, pz);
        } else {
            const float px = wn::lattice_coord(x, den, g.base_range, g.octave_scale, g.post_scale);
            const float py = wn::lattice_coord(y, den, g.base_range, g.octave_scale, g.post_scale);
            const float pz = g.z_const_mode ? g.z_const
                                            : wn::lattice_coord(g.z0 + z, den, g.base_range,
                                                                g.octave_scale, g.post_scale);
            v = 0.0f;
            for (int b = 0; b < a.nbands; ++b) {
                const float s = a.band_scale[b];
                v += a.band_w[b] * wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, 2.0f * px * s,
                                                2.0f * py * s, 2.0f * pz * s);
            }
            if (a.apply_div) v /= a.out_div;
        }
        a.out[e] = v * g.out_scale;
    }
}

__global__ __launch_bounds__(256) void grid2d_direct_kernel(const DirectArgs a)
{
    const GridArgs &g = a.g;
    const size_t total = (size_t)g.nx * g.ny;
    const float den = (float)g.den;
    const int n = a.n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.nx), y = (int)(e / g.nx);
        const float px = wn::lattice_coord(x, den, g.base_range, g.octave_scale, g.post_scale);
        const float py = wn::lattice_coord(y, den, g.base_range, g.octave_scale, g.post_scale);
        const float result = wn::eval2d_exact(a.coef, n, a.nmask, px, py);
        a.out[e] = result * g.out_scale;
    }
}

} // namespace

namespace {

struct ProjGridArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    float normal[3];
};

__global__ __launch_bounds__(256) void grid3d_projected_kernel(const ProjGridArgs a)
{
    const GridArgs &g = a.g;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    const float den = (float)g.den;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % g.nx);
        const size_t r = e / g.nx;
        const int y = (int)(r % g.ny), z = (int)(r / g.ny);
        float p[3];
        p[0] = wn::lattice_coord(x, den, g.base_range, g.octave_scale, g.post_scale);
        p[1] = wn::lattice_coord(y, den, g.base_range, g.octave_scale, g.post_scale);
        p[2] = g.z_const_mode ? g.z_const
                              : wn::lattice_coord(g.z0 + z, den, g.base_range, g.octave_scale,
                                                  g.post_scale);
        a.out[e] = wn::projected_exact(a.coef, a.n, a.nmask, p, a.normal) * g.out_scale;
    }
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

inline int grid_blocks(size_t total)
{
    size_t b = (total + 255) / 256;
    const size_t cap = 256u * 8u * 4u;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

inline int ceil_pow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Plan the separable kernel; returns false when the lattice is outside its regime.
bool plan_sep_bz(const wn_tile *tile, const GridArgs &g, int nbands, const float *oscale,
                 const float *weights, float out_div, SepArgs *a, size_t *lds_bytes, int bz_cap)
{
    // two wave columns (512-sample bricks: whole 2-KiB rows written together) when the lattice is wide
    int xw = g.nx > 256 ? 2 : 1;
    // 512-wide bricks write whole 2-KiB rows (~6 % faster than 256-wide ones at 1024^3), unless their last column
    // would be mostly idle lanes (768 = 512 + 256: a quarter of all lanes): then 256-wide bricks, two workgroups per CU
    // (768^3: 376 vs 434 us, profiles/r02_sep_knob_sweep.txt)
    if (xw == 2) {
        const double padded2 = (double)((g.nx + 511) / 512) * 512, padded1 = (double)((g.nx + 255) / 256) * 256;
        if (padded2 > 1.1 * padded1) xw = 1;
    }
#ifdef WN_TUNE_ENV
    if (const char *e = getenv("WN_SEP_XW")) xw = atoi(e);
#endif
    const int kBrickX = 256 * xw;
    a->xw = xw;
    if (tile->n == 0 || nbands < 1 || nbands > kMaxBands) return false;
    if (pow2_mask(tile->n) < 0) return false; // the brick kernel wraps with a mask: power-of-two tiles
    if (g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return false;
    if (!g.z_const_mode && g.z0 < 0) return false; // negative plane indices: the exact kernel (bounds below assume indices >= 0)
    int BZ = g.nz >= bz_cap ? bz_cap : ceil_pow2(g.nz); // planes per brick
#ifdef WN_TUNE_ENV
    if (const char *e = getenv("WN_SEP_BZ")) BZ = std::min(BZ, atoi(e));
#endif
    const int rows = kBrickY * BZ;
    const double zmax = g.z_const_mode ? 0.0 : (double)g.z0 + g.nz;
    const double imax = fmax(fmax((double)g.nx, (double)g.ny), zmax);
    size_t box_total = 0, r_total = 0;
    int exs[kMaxBands];
    for (int b = 0; b < nbands; ++b) {
        const double step = (double)g.base_range * (double)oscale[b] * (double)g.post_scale / g.den;
        if (!(step >= 0.0) || !std::isfinite(step)) return false;
        const double pmax = step * imax + fabs((double)g.z_const) + 1.0;
        if (pmax > 1.0e6) return false; // keep mids far inside int / float-exact range
        const double slack = pmax * 4.8e-7; // 4 ulp of the largest coordinate
        if (3.0 * step + slack > 1.0) return false; // 4 consecutive samples span <= 2 mids
        auto extent = [&](int samples) { return (int)floor((samples - 1) * step + slack) + 1 + 3; };
        exs[b] = extent(kBrickX) + 1;
        if (extent(kBrickY) > kBoxY || (!g.z_const_mode && extent(BZ) > kBoxZ)) return false;
        {   // the kernel's compile-time bounds per band (bands of one call are consecutive octaves)
            const int d = nbands - 1 - b;
            const int ez = g.z_const_mode ? 3 : extent(BZ);
            if (extent(kBrickY) * ez > box_rows_bound(d) || exs[b] > 64 * box_col_groups(d, kBrickX)) return false;
        }
        a->band[b].oscale = oscale[b];
        a->band[b].factor = (float)((double)(weights ? weights[b] : 1.0f) * (double)g.out_scale / (double)out_div);
        a->band[b].ex_cap = exs[b];
        a->band[b].box_off = (int)box_total;
        box_total += (size_t)exs[b] * kColStride;
    }
    // single band: two copies of box and R (one-barrier pipeline); several bands: one copy each
    const int copies = nbands == 1 ? 2 : 1;
    a->box_buf_stride = nbands == 1 ? (int)box_total : 0;
    size_t off = copies * box_total; // boxes first, then the R rows
    for (int b = 0; b < nbands; ++b) {
        a->band[b].r_off = (int)(off + r_total);
        r_total += (size_t)rows * (exs[b] | 1) + 4;
    }
    a->r_buf_stride = nbands == 1 ? (int)r_total : 0;
    off += copies * r_total;
    *lds_bytes = off * sizeof(float);
    if (*lds_bytes > 120 * 1024) return false;
    a->bz_log2 = __builtin_ctz(BZ);
    a->nbx = (g.nx + kBrickX - 1) / kBrickX;
    a->nby = (g.ny + kBrickY - 1) / kBrickY;
    a->nbz = (g.nz + BZ - 1) / BZ;
    const long long blocks = (long long)a->nbx * a->nby * a->nbz;
    if (blocks > 0x7fffffffLL) return false;
    a->nbands = nbands;
    a->coef = tile->dev;
    a->n = tile->n;
    a->nmask = pow2_mask(tile->n);
    a->g = g;
    a->inv_den = ((g.den & (g.den - 1)) == 0) ? 1.0f / (float)g.den : 0.0f;
    return true;
}

// Planes per brick: 8 (fewer for thin slabs); 16 for a single band on 256-wide bricks whose 16 planes still fit the
// 6-row box and LDS (half the box loads, tables and barriers per sample).  Measured (profiles/r02_sep_knob_sweep.txt):
// 768^3 (256-wide bricks) 322 vs 378-384 us; with 512-wide bricks no gain (1024^3 800 vs 791-826, 2048 x 2048 x 256
// 743 vs 722), so those keep 8.
bool plan_sep(const wn_tile *tile, const GridArgs &g, int nbands, const float *oscale,
              const float *weights, float out_div, SepArgs *a, size_t *lds_bytes)
{
    if (nbands == 1 && !g.z_const_mode && g.nz >= 16) {
        SepArgs a16{};
        size_t lds16 = 0;
        if (plan_sep_bz(tile, g, nbands, oscale, weights, out_div, &a16, &lds16, 16) && a16.xw == 1) {
            *a = a16;
            *lds_bytes = lds16;
            return true;
        }
    }
    return plan_sep_bz(tile, g, nbands, oscale, weights, out_div, a, lds_bytes, 8);
}

// Persistent grid.  Single band: ONE 8-wave workgroup per CU (two of the 4-wave workgroups of 256-wide bricks) -- measured on MI355X (profiles/r02_sep_knob_sweep.txt):
// 2048 x 2048 x 256 slab 748 us with 1 workgroup per CU against 777-787 with 2-3, 1024^3 803 against 853-866; the
// double-buffered one-barrier pipeline already overlaps a brick's loads, collapse and stores inside one
// workgroup, and a second resident workgroup only adds contention on the store path.  Several bands: k workgroups
// per CU, k chosen (within what LDS and the 32-wave limit admit) so that the bricks divide evenly.
int persistent_grid(long long items, size_t lds_bytes, int xw, int nbands)
{
    const int cus = wn::device_compute_units(wn::current_device());
    int kmax = (int)((160 * 1024) / (lds_bytes + 2048));
    const int wave_cap = 8 / xw; // 32 waves per CU
    kmax = kmax > wave_cap ? wave_cap : (kmax < 1 ? 1 : kmax);
#ifdef WN_TUNE_ENV
    if (const char *e = getenv("WN_SEP_K")) return (int)std::min<long long>(items, (long long)cus * std::min(kmax, atoi(e)));
#endif
    if (nbands == 1) return (int)std::min<long long>(items, (long long)cus * (xw == 1 ? 2 : 1)); // 8 waves per CU either way
    int best_k = kmax;
    double best_eff = -1.0;
    for (int k = kmax; k >= (kmax > 4 ? 4 : 1); --k) {
        const long long wgs = (long long)cus * k;
        if (items <= wgs) return (int)items;
        const long long per = (items + wgs - 1) / wgs;
        const double eff = (double)items / (double)(per * wgs) * (0.9 + 0.0125 * k); // mild preference for occupancy
        if (eff > best_eff) { best_eff = eff; best_k = k; }
    }
    return (int)std::min<long long>(items, (long long)cus * best_k);
}

template <int NB, int XW>
bool launch_sep2(const SepArgs &a, size_t lds, hipStream_t s)
{
    const long long items = (long long)a.nbx * a.nby * a.nbz;
    // dynamic LDS beyond 64 KiB needs a per-(kernel, device) opt-in
    if (lds > 48 * 1024 &&
        !wn::ensure_dynamic_lds(reinterpret_cast<const void *>(&grid3d_sep_kernel<NB, XW>), wn::current_device(), 128 * 1024))
        return false;
    hipLaunchKernelGGL((grid3d_sep_kernel<NB, XW>), dim3(persistent_grid(items, lds, XW, NB)),
                       dim3(256 * XW), lds, s, a);
    return true;
}

template <int NB>
bool launch_sep(const SepArgs &a, size_t lds, hipStream_t s)
{
    return a.xw == 2 ? launch_sep2<NB, 2>(a, lds, s) : launch_sep2<NB, 1>(a, lds, s);
}

// *launched = false: the runtime refused the LDS opt-in; the caller falls back to the direct kernel.
int run_sep(const SepArgs &a, size_t lds, hipStream_t s, bool *launched)
{
    bool ok;
    switch (a.nbands) {
    case 1: ok = launch_sep<1>(a, lds, s); break;
    case 2: ok = launch_sep<2>(a, lds, s); break;
    case 3: ok = launch_sep<3>(a, lds, s); break;
    case 4: ok = launch_sep<4>(a, lds, s); break;
    case 5: ok = launch_sep<5>(a, lds, s); break;
    case 6: ok = launch_sep<6>(a, lds, s); break;
    case 7: ok = launch_sep<7>(a, lds, s); break;
    default: ok = launch_sep<8>(a, lds, s); break;
    }
    *launched = ok;
    if (!ok) return WN_OK;
    WN_LAUNCH_CHECK("grid3d_sep_kernel");
    return WN_OK;
}

} // namespace

using namespace wn;

extern "C" {

int wn_eval3d_grid(const wn_tile *tile, const wn_grid *grid, float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (!tile) return fail(WN_ERR_INVALID, "tile is NULL");
    if (tile->count && (rc = check_handle_device(tile->device, "tile")) != WN_OK) return rc;
    if (tile->count && tile->dims != 3) return fail(WN_ERR_INVALID, "wn_eval3d_grid needs a 3-D tile");
    GridArgs g;
    rc = check_grid(grid, true, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    if (total == 0) return WN_OK;
    if (!out_dev) return fail(WN_ERR_INVALID, "out_dev is NULL");

    if (!(grid->flags & WN_GRID_EXACT)) {
        bool launched = false;
        // The plane pipeline of wn_wavelet_multiband.hip with a single band (16-wave workgroups split into window / collapse /
        // store waves; the waves that store never compute or wait for a load): 512^3 91-93 us sustained against 101.9 for the
        // strip kernel, 1024^3 664 against 809 us and the 2048 x 2048 x 256 shard 646 against 710 us for the brick kernel
        // (round 3, profiles/r03_single_band_plane_pipeline.txt).  It takes EVERY lattice in its regime, thin slabs included
        // (512 x 512 x 64: 18.0 us against 17.2 for the strip kernel): a sample's bits must not depend on how the volume was cut
        // into z-slabs, so the choice of kernel must not depend on the slab's thickness (the kernels differ in the last bits:
        // each sums in its own order).  The strip and brick kernels serve what the pipeline does not cover (rows of 256 or 768
        // samples, steps above 2/7 of a cell).
        const float os1 = g.octave_scale, w1 = 1.0f;
        rc = multiband_try(tile, g, 1, &os1, &w1, 1.0f, out_dev, as_stream(stream), &launched, 0);
        if (rc || launched) return rc;
        rc = strip_try(tile, g, out_dev, as_stream(stream), &launched); // rows of k*256 samples, >= 0.18 planes per step
        if (rc || launched) return rc;
        SepArgs a{};
        size_t lds = 0;
        const float os = g.octave_scale;
        if (plan_sep(tile, g, 1, &os, nullptr, 1.0f, &a, &lds)) {
            a.out = out_dev;
            a.vec4_ok = (g.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
            a.even_share_q10 = a.nbands == 1 ? 545 : 512;
#ifdef WN_TUNE_ENV
            if (const char *e = getenv("WN_SEP_EVEN_SHARE")) a.even_share_q10 = atoi(e);
#endif
            rc = run_sep(a, lds, as_stream(stream), &launched);
            if (rc || launched) return rc;
        }
    }
    {   // bit-exact: the reference's 27-tap loop on an LDS-staged coefficient box when the bricks' boxes fit ...
        bool launched = false;
#ifdef WN_TUNE_ENV
        if (!getenv("WN_NO_EXACT_LDS"))
#endif
        rc = exact_lds_try(tile, g, out_dev, as_stream(stream), &launched);
        if (rc || launched) return rc;
    }
    // ... else on gathers from the tile
    DirectArgs d{};
    d.coef = tile->dev;
    d.out = out_dev;
    d.n = tile->n;
    d.nmask = pow2_mask(tile->n);
    d.g = g;
    d.nbands = 0;
    if (tile->dev_padded) {
        d.coef = tile->dev_padded;
        hipLaunchKernelGGL(grid3d_direct_kernel<true>, dim3(grid_blocks(total)), dim3(256), 0, as_stream(stream), d);
    } else {
        hipLaunchKernelGGL(grid3d_direct_kernel<false>, dim3(grid_blocks(total)), dim3(256), 0, as_stream(stream), d);
    }
    WN_LAUNCH_CHECK("grid3d_direct_kernel");
    return WN_OK;
}

int wn_multiband3d_grid(const wn_tile *tile, const wn_grid *grid, float s, int first_band,
                        int nbands, const float *w_host, float var_per_band, float *out_dev,
                        void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (!tile) return fail(WN_ERR_INVALID, "tile is NULL");
    if (tile->count && (rc = check_handle_device(tile->device, "tile")) != WN_OK) return rc;
    if (tile->count && tile->dims != 3) return fail(WN_ERR_INVALID, "wn_multiband3d_grid needs a 3-D tile");
    if (nbands < 0 || nbands > kMaxBands)
        return fail(WN_ERR_INVALID, "nbands must be in 0..%d (got %d)", kMaxBands, nbands);
    if (nbands && !w_host) return fail(WN_ERR_INVALID, "w_host is NULL");
    GridArgs g;
    rc = check_grid(grid, true, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    if (total == 0) return WN_OK;
    if (!out_dev) return fail(WN_ERR_INVALID, "out_dev is NULL");

    // Appendix 2: bands run while s + firstBand + b < 0; the variance sums ALL nbands.
    int active = 0;
    while (active < nbands && s + (float)first_band + (float)active < 0.0f) ++active;
    float variance = 0.0f;
    for (int b = 0; b < nbands; ++b) variance += w_host[b] * w_host[b];
    const bool apply_div = variance != 0.0f;
    const float out_div = apply_div ? sqrtf(variance * var_per_band) : 1.0f;

    float bscale[kMaxBands], wts[kMaxBands], oscale[kMaxBands];
    for (int b = 0; b < active; ++b) {
        bscale[b] = ldexpf(1.0f, first_band + b);
        wts[b] = w_host[b];
        // q = 2*p*2^(first+b): in lattice form octave_scale*2^(first+b), post 2.
        oscale[b] = g.octave_scale * bscale[b];
    }

    // z_mode == WN_Z_CONST: band b sits at 2*z_const*2^(first_band+b), which the brick kernel's per-band
    // tables do not model (they scale lattice indices, not a constant) -> the direct kernel
    if (!(grid->flags & WN_GRID_EXACT) && active >= 1 && g.post_scale == 1.0f && !g.z_const_mode) {
        SepArgs a{};
        size_t lds = 0;
        GridArgs gb = g;
        gb.post_scale = 2.0f;
        {   // the plane-pipeline kernel (wn_wavelet_multiband.hip) takes wide lattices of 2..5 bands ...
            bool launched = false;
            rc = multiband_try(tile, gb, active, oscale, wts, out_div, out_dev, as_stream(stream), &launched, 0);
            if (rc || launched) return rc;
        }
        // ... the brick kernel the rest
        if (plan_sep(tile, gb, active, oscale, wts, out_div, &a, &lds)) {
            a.out = out_dev;
            a.vec4_ok = (g.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
            a.even_share_q10 = a.nbands == 1 ? 545 : 512;
#ifdef WN_TUNE_ENV
            if (const char *e = getenv("WN_SEP_EVEN_SHARE")) a.even_share_q10 = atoi(e);
#endif
            bool launched = false;
            rc = run_sep(a, lds, as_stream(stream), &launched);
            if (rc || launched) return rc;
        }
    }
    DirectArgs d{};
    d.coef = tile->dev;
    d.out = out_dev;
    d.n = tile->n;
    d.nmask = pow2_mask(tile->n);
    d.g = g;
    d.nbands = active;
    for (int b = 0; b < active; ++b) {
        d.band_scale[b] = bscale[b];
        d.band_w[b] = wts[b];
    }
    d.out_div = out_div;
    d.apply_div = apply_div ? 1 : 0;
    if (active == 0) {
        // no band contributes: result = 0 (/ out_div) * out_scale, evaluated on the device
        d.nbands = 1;
        d.band_scale[0] = 1.0f;
        d.band_w[0] = 0.0f;
    }
    if (tile->dev_padded) {
        d.coef = tile->dev_padded;
        hipLaunchKernelGGL(grid3d_direct_kernel<true>, dim3(grid_blocks(total)), dim3(256), 0, as_stream(stream), d);
    } else {
        hipLaunchKernelGGL(grid3d_direct_kernel<false>, dim3(grid_blocks(total)), dim3(256), 0, as_stream(stream), d);
    }
    WN_LAUNCH_CHECK("grid3d_direct_kernel(multiband)");
    return WN_OK;
}

int wn_eval2d_grid(const wn_tile *tile, const wn_grid *grid, float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (!tile) return fail(WN_ERR_INVALID, "tile is NULL");
    if (tile->count && (rc = check_handle_device(tile->device, "tile")) != WN_OK) return rc;
    if (tile->count && tile->dims != 2) return fail(WN_ERR_INVALID, "wn_eval2d_grid needs a 2-D tile");
    GridArgs g;
    rc = check_grid(grid, false, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny;
    if (total == 0) return WN_OK;
    if (!out_dev) return fail(WN_ERR_INVALID, "out_dev is NULL");
    DirectArgs d{};
    d.coef = tile->dev;
    d.out = out_dev;
    d.n = tile->n;
    d.nmask = pow2_mask(tile->n);
    d.g = g;
    hipLaunchKernelGGL(grid2d_direct_kernel, dim3(grid_blocks(total)), dim3(256), 0,
                       as_stream(stream), d);
    WN_LAUNCH_CHECK("grid2d_direct_kernel");
    return WN_OK;
}

int wn_eval3d_projected_grid(const wn_tile *tile, const wn_grid *grid, const float normal[3],
                             float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (!tile || !normal) return fail(WN_ERR_INVALID, "tile/normal is NULL");
    if (tile->count && (rc = check_handle_device(tile->device, "tile")) != WN_OK) return rc;
    if (tile->count && tile->dims != 3)
        return fail(WN_ERR_INVALID, "wn_eval3d_projected_grid needs a 3-D tile");
    GridArgs g;
    rc = check_grid(grid, true, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    if (total == 0) return WN_OK;
    if (!out_dev) return fail(WN_ERR_INVALID, "out_dev is NULL");
    ProjGridArgs a{};
    a.coef = tile->dev;
    a.out = out_dev;
    a.n = tile->n;
    a.nmask = pow2_mask(tile->n);
    a.g = g;
    for (int i = 0; i < 3; ++i) a.normal[i] = normal[i];
    hipLaunchKernelGGL(grid3d_projected_kernel, dim3(grid_blocks(total)), dim3(256), 0,
                       as_stream(stream), a);
    WN_LAUNCH_CHECK("grid3d_projected_kernel");
    return WN_OK;
}

} // extern "C"
