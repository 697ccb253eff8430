// wn_perlin.hip -- Ken Perlin "improved noise" (perlin.h:42-62, experient/PerlinNoise.hpp:36-60)
// for gfx950: dense grids, point lists, turb / fractal_noise, and the noise_texture adaptor
// (K4 / K6 of SURVEY.md 8).
//
// The 512-entry permutation table sits in LDS as bytes; all arithmetic is fp64 in the
// reference's order with contraction off, and the integer hash path is plain int arithmetic, so
// results are bit-identical to the CPU classes.  Grids write one float per sample (4 B/sample
// of HBM traffic); the work per sample is ~60 fp64 operations, so this path is fp64-VALU bound,
// not HBM bound.
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"
#include "wn_perlin_run.hpp"
#include "wn_texture_eval.hpp"

#include <cmath>

namespace {

using wn::GridArgs;

enum { kNoise = 0, kTurb = 1, kFractal = 2 };

struct PerlinGridArgs {
    const uint8_t *perm;
    float *out;
    GridArgs g;
    int kind, depth;
    int vec4_ok; // rows start 16-byte aligned (nx % 4 == 0 and an aligned output pointer)
};

// Generic dense-grid kernel: one sample per lane, every sample hashes for itself.  Serves what the
// run kernel below does not (narrow grids, depth 0 or > kRunMaxDepth).
__global__ __launch_bounds__(256) void perlin_grid_generic_kernel(const PerlinGridArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    const GridArgs &g = a.g;
    const float den = (float)g.den;
    const unsigned plane = (unsigned)g.nx * (unsigned)g.ny;
    const size_t total = (size_t)plane * g.nz;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (size_t)gridDim.x * blockDim.x) {
        const unsigned z = (unsigned)(e / plane);
        const unsigned r = (unsigned)(e - (size_t)z * plane);
        const unsigned y = r / (unsigned)g.nx, x = r - y * (unsigned)g.nx;
        const float px = wn::lattice_coord((int)x, den, g.base_range, g.octave_scale, g.post_scale);
        const float py = wn::lattice_coord((int)y, den, g.base_range, g.octave_scale, g.post_scale);
        const float pz = g.z_const_mode ? g.z_const
                                        : wn::lattice_coord(g.z0 + (int)z, den, g.base_range,
                                                            g.octave_scale, g.post_scale);
        double v;
        if (a.kind == kNoise) v = wn::perlin_exact(perm, (double)px, (double)py, (double)pz);
        else if (a.kind == kTurb) v = wn::perlin_turb(perm, px, py, pz, a.depth);
        else v = wn::perlin_fractal(perm, px, py, pz);
        a.out[e] = (float)v * g.out_scale;
    }
}


// ------------------------------------------------------------------------------------------------
// perlin_grid_run_kernel -- dense grids built around what consecutive samples of an axis-aligned
// lattice SHARE (perlin.h:42-62 evaluated for a whole block):
//   * everything per axis is per axis: floor / fractional part / fade of a coordinate depend on one
//     index only.  A workgroup (8 waves) owns 512 x samples x kRunTY rows x <= kRunTZ planes and first
//     tabulates, per octave, {xf, fade(xf)} and the cell index X for its 512 x samples and the same for
//     its rows and planes (LDS; fp64, the reference's operation order);
//   * a lane walks a RUN of 8 consecutive x samples of one row.  The eight corner hashes
//     p[p[p[X]+Y]+Z] ... (perlin.h:55-61: 14 table look-ups) and everything derived from them are
//     computed once per cell the run enters (at the BASELINE lattice -- step 1/8 per sample -- once
//     per run), not once per sample;
//   * grad() (perlin.h:26-31) picks two of (x,y,z) and two signs from the low 4 hash bits.  Inside a
//     run only x moves, so a corner's gradient is  (+-dx | nothing) + K  with K = (+-dy) + (+-dz),
//     +-dy or +-dz, a per-row constant.  Per row and octave the wave builds a 64-entry LDS table
//     [cy][cz][h] -> {K, and-mask, sign-xor} with one lane per entry; a lane fetches its 8 corners'
//     entries with 8 ds_read_b128.  "Nothing" is -0.0, the identity of IEEE addition for every K
//     including both zeros, so each gradient is the single rounded addition the reference performs
//     (its two operands commute) and lattice points keep the reference's zero signs.
//   * per sample what is left is 8 x (2 and + 1 xor + 1 fp64 add) for the gradients and the 7 lerps
//     (21 fp64 ops, unfused, reference order): 30-32 fp64 + 24 integer VALU instructions per sample
//     and octave instead of ~60 + ~170.
// turb / fractal_noise loop octaves per row with the running sums of the 8 samples in registers, in
// the reference's accumulation order.  Results are bit-identical to perlin_exact / the CPU classes.
// ------------------------------------------------------------------------------------------------
constexpr int kRunMaxDepth = 8;
constexpr int kRunX = 512;  // x samples per workgroup (64 lanes x 8)
constexpr int kRunTY = 8;   // rows ...
constexpr int kRunTZ = 8;   // ... and planes per workgroup
constexpr int kRun = 8;     // samples per lane and row
// Waves per workgroup (template parameter W): they share the block's axis tables (57 KB at 7 octaves) and
// each adds 5 KB of its own, so deep turb runs 16 waves on one set of tables (4 waves per SIMD) where two
// 8-wave workgroups would not fit a CU's LDS; shallow grids use 8 (three workgroups per CU).

struct RunAxisEntry {
    double f, fade; // fractional part and its fade()
};
using wn::RunKEntry;

__host__ __device__ constexpr size_t run_lds_bytes(int depth, int kRunWaves)
{
    return 512 /* perm */ + (size_t)depth * kRunX * sizeof(RunAxisEntry) + (size_t)depth * kRunX /* x cells */ +
           (size_t)depth * (kRunTY + kRunTZ) * (sizeof(RunAxisEntry) + sizeof(int)) + kRunWaves * 64 * sizeof(RunKEntry) +
           kRunWaves * kRun * 64 * sizeof(double) /* running sums; the finished row is parked in the same slots */;
}

template <int KIND, int kRunWaves>
__global__ __launch_bounds__(64 * kRunWaves) void perlin_grid_run_kernel(const PerlinGridArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char run_lds[];
    const GridArgs &g = a.g;
    const int depth = (KIND == kNoise) ? 1 : ((KIND == kFractal) ? 6 : a.depth);
    // LDS carve-up (all 16-byte aligned)
    RunKEntry *const ktab_all = reinterpret_cast<RunKEntry *>(run_lds);                        // [waves][64]
    RunAxisEntry *const xtab = reinterpret_cast<RunAxisEntry *>(run_lds + kRunWaves * 64 * sizeof(RunKEntry)); // [depth][512]
    RunAxisEntry *const ytab = xtab + (size_t)depth * kRunX;                                      // [depth][kRunTY]
    RunAxisEntry *const ztab = ytab + (size_t)depth * kRunTY;                                     // [depth][kRunTZ]
    int *const ycell = reinterpret_cast<int *>(ztab + (size_t)depth * kRunTZ);                    // [depth][kRunTY]
    int *const zcell = ycell + depth * kRunTY;                                                    // [depth][kRunTZ]
    uint8_t *const xcell = reinterpret_cast<uint8_t *>(zcell + depth * kRunTZ);                   // [depth][512]
    uint8_t *const perm = xcell + (size_t)depth * kRunX;                                          // [512]
    double *const acc_all = reinterpret_cast<double *>(run_lds + run_lds_bytes(depth, kRunWaves) - kRunWaves * kRun * 64 * sizeof(double));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x_first = blockIdx.x * kRunX, y_first = blockIdx.y * kRunTY, z_first = blockIdx.z * kRunTZ;
    const float den = (float)g.den;

    // ---- per-axis tables of the block ---------------------------------------------------------
    for (int i = tid; i < 128; i += 64 * kRunWaves)
        reinterpret_cast<uint32_t *>(perm)[i] = reinterpret_cast<const uint32_t *>(a.perm)[i];
    // entry (axis, index): axis 0 = x (512 entries), 1 = y, 2 = z
    auto tabulate = [&](float p, RunAxisEntry *tab, int stride_entries, int slot, int *cells32, uint8_t *cells8) {
        float cur = p;         // turb: the float point doubles per octave
        double frequency = 1.0; // fractal_noise: float point times a double frequency (perlin.h:82-84)
        for (int i = 0; i < depth; ++i) {
            const double c = (KIND == kFractal) ? (double)p * frequency : (double)cur;
            const double fl = floor(c);
            const int cell = (int)fl & 255;
            const double f = c - fl;
            if (tab) tab[(size_t)i * stride_entries + slot] = RunAxisEntry{f, wn::pfade(f)};
            if (cells32) cells32[i * stride_entries + slot] = cell;
            else if (cells8) cells8[(size_t)i * stride_entries + slot] = (uint8_t)cell;
            cur *= 2.0f;
            frequency *= 2.0;
        }
    };
    // x entries are stored [octave][q][lane] (sample x = lane*8 + q): the 64 lanes of a wave read 64 adjacent
    // 16-byte entries; [lane][q] order put all lanes on the same banks (8-way conflicts, 12 % of the kernel)
    for (int xi = tid; xi < kRunX; xi += 64 * kRunWaves) {
        const int x = min(x_first + xi, g.nx - 1);
        const float px = wn::lattice_coord(x, den, g.base_range, g.octave_scale, g.post_scale);
        tabulate(px, xtab, kRunX, (xi & (kRun - 1)) * 64 + (xi >> 3), nullptr, nullptr);
        tabulate(px, nullptr, kRunX, xi, nullptr, xcell);
    }
    if (tid < kRunTY) {
        const int y = min(y_first + tid, g.ny - 1);
        tabulate(wn::lattice_coord(y, den, g.base_range, g.octave_scale, g.post_scale), ytab, kRunTY, tid, ycell, nullptr);
    } else if (tid >= 64 && tid < 64 + kRunTZ) {
        const int zi = tid - 64;
        const int z = g.z0 + min(z_first + zi, g.nz - 1);
        const float pz = g.z_const_mode ? g.z_const : wn::lattice_coord(z, den, g.base_range, g.octave_scale, g.post_scale);
        tabulate(pz, ztab, kRunTZ, zi, zcell, nullptr);
    }
    __syncthreads();

    RunKEntry *const ktab = ktab_all + wave * 64;
    const int rows_y = min(kRunTY, g.ny - y_first), rows_z = min(kRunTZ, g.nz - z_first);
    // this lane's entry of the per-row table: hash h, corner (cy, cz)
    const int kh = lane & 15, kcy = (lane >> 4) & 1, kcz = lane >> 5;
    // [q][lane]: this wave's running sums (turb / fractal); a finished sample is parked as a float in the low
    // half of its own slot, so no lane's pending sum is overwritten
    double *const acc = acc_all + wave * (kRun * 64) + lane;
    const float *const stage = reinterpret_cast<const float *>(acc_all + wave * (kRun * 64));
    for (int r = wave; r < rows_y * rows_z; r += kRunWaves) {
        const int yi = r % rows_y, zi = r / rows_y;
        double amp_sum = 0.0, weight = 1.0; // turb weight / fractal amplitude

        for (int oc = 0; oc < depth; ++oc) {
            const RunAxisEntry ye = ytab[oc * kRunTY + yi], ze = ztab[oc * kRunTZ + zi];
            const int Y = ycell[oc * kRunTY + yi], Z = zcell[oc * kRunTZ + zi];
            // ---- per-row table: entry (cy, cz, h) -> {K, mm, t} (grad(), perlin.h:26-31) -------------
            {
                const double dy = kcy ? ye.f - 1.0 : ye.f, dz = kcz ? ze.f - 1.0 : ze.f;
                const RunKEntry mine = wn::run_k_entry(kh, dy, dz);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the previous octave's reads are done
                ktab[lane] = mine;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            const double v = ye.fade, w = ze.fade;
            const RunAxisEntry *const xe = xtab + (size_t)oc * kRunX + lane; // entry q at xe[q * 64]
            // the run's 8 cell indices in one read
            const uint64_t cells = *reinterpret_cast<const uint64_t *>(xcell + (size_t)oc * kRunX + lane * kRun);

            // The corner state of the cell a lane is in: K, the and-mask and the sign-xor of its 8 corners.
            double K[8];
            uint32_t mm[8], tt[8];
            // hash the cell's 8 corners (perlin.h:55-61) and fetch their {K, mm, t}.  (Keeping the hashes of a lane's
            // previous row in a register -- a wave keeps its y row and walks z, so the cell is usually the same --
            // measured no gain: 266-273 vs 255-265 us.)
            auto hash_cell = [&](int X) {
                const int A = perm[X] + Y, AA = perm[A] + Z, AB = perm[A + 1] + Z;
                const int B = perm[X + 1] + Y, BA = perm[B] + Z, BB = perm[B + 1] + Z;
                const int h[8] = {perm[AA], perm[BA], perm[AB], perm[BB],
                                  perm[AA + 1], perm[BA + 1], perm[AB + 1], perm[BB + 1]};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const RunKEntry e = ktab[(c >> 1) * 16 + (h[c] & 15)];
                    K[c] = e.K;
                    mm[c] = e.mm;
                    tt[c] = e.t;
                }
            };
            auto sample = [&](int qs, const RunAxisEntry &x) {
                const double xf = x.f, u = x.fade, xm1 = xf - 1.0;
                const uint64_t b0 = (uint64_t)__double_as_longlong(xf), b1 = (uint64_t)__double_as_longlong(xm1);
                double gr[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) gr[c] = wn::run_gradient(K[c], mm[c], tt[c], (c & 1) ? b1 : b0);
                const double x00 = wn::plerp(u, gr[0], gr[1]), x10 = wn::plerp(u, gr[2], gr[3]);
                const double x01 = wn::plerp(u, gr[4], gr[5]), x11 = wn::plerp(u, gr[6], gr[7]);
                const double nv = wn::plerp(w, wn::plerp(v, x00, x10), wn::plerp(v, x01, x11));
                double sum;
                if (KIND == kNoise) sum = nv;
                else {
                    const double before = oc ? acc[qs * 64] : 0.0;
                    sum = (KIND == kTurb) ? before + weight * nv : before + nv * weight;
                }
                if (oc + 1 < depth) acc[qs * 64] = sum;
                else { // last octave: finish the sample and park it in its own slot
                    if (KIND == kTurb) sum = fabs(sum);
                    if (KIND == kFractal) sum = sum / (amp_sum + weight);
                    *reinterpret_cast<float *>(acc + qs * 64) = (float)sum * g.out_scale;
                }
            };
            const bool one_cell = cells == (cells & 255u) * 0x0101010101010101ull;
            if (__all(one_cell)) {
                // Every lane's run stays in one cell (the BASELINE lattice, and every octave of turb up to a step
                // of 1/8): one hash, then the 8 samples straight-line with a wave-uniform q -- LDS offsets become
                // immediates and there is no per-sample loop control.
                hash_cell((int)(cells & 255u));
#pragma unroll
                for (int qs = 0; qs < kRun; ++qs) {
                    sample(qs, xe[qs * 64]);
                    if (qs & 1) __builtin_amdgcn_sched_barrier(0); // pairs of samples: all eight interleaved need > 128 VGPRs
                }
            } else {
                // Segments of the run that stay inside one cell.  Both loops are rolled and q is a per-lane value --
                // everything indexed by q lives in LDS -- so the corner state is defined once per segment and never
                // copied between branches (an unrolled form with a conditional re-hash per sample spent a third of
                // its instructions on such copies).
                auto cell_at = [&](int qq) { return (int)((cells >> (8 * qq)) & 255u); };
                int q = 0;
                while (q < kRun) {
                    const int X = cell_at(q);
                    hash_cell(X);
                    RunAxisEntry x = xe[q * 64];
                    bool more;
                    do { // one sample per trip; two per trip (more ILP, half the loop control) measured the same
                        const int q1 = min(q + 1, kRun - 1);
                        const RunAxisEntry xnext = xe[q1 * 64]; // requested before this sample's arithmetic
                        sample(q, x);
                        ++q;
                        more = q < kRun && cell_at(q1) == X;
                        x = xnext;
                    } while (more);
                }
            }
            if (KIND == kFractal) amp_sum += weight;
            weight *= 0.5;
        }

        // the row leaves as contiguous 1-KiB wave stores
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float *const dst = a.out + ((size_t)(z_first + zi) * g.ny + (y_first + yi)) * g.nx + x_first;
        typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int xo = half * 256 + lane * 4; // sample xo + e sits in slot (q, lane') = ((xo + e) % 8, (xo + e) / 8)
            const float *const src = stage + 2 * ((((lane & 1) * 4) * 64) + (xo >> 3));
            const v4f val = v4f{src[0], src[2 * 64], src[4 * 64], src[6 * 64]};
            if (a.vec4_ok && x_first + xo + 4 <= g.nx) *reinterpret_cast<v4f *>(dst + xo) = val;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (x_first + xo + e < g.nx) dst[xo + e] = val[e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the stage is read before the next row parks into it
    }
}

struct PerlinPointsArgs {
    const uint8_t *perm;
    const double *pts64;
    const float *pts32;
    double *out;
    size_t count;
    int kind, depth;
};

__global__ __launch_bounds__(256) void perlin_points_kernel(const PerlinPointsArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        double v;
        if (a.pts64) {
            const double *p = a.pts64 + 3 * i;
            v = wn::perlin_exact(perm, p[0], p[1], p[2]);
        } else {
            const float *p = a.pts32 + 3 * i;
            if (a.kind == kNoise) v = wn::perlin_exact(perm, (double)p[0], (double)p[1], (double)p[2]);
            else if (a.kind == kTurb) v = wn::perlin_turb(perm, p[0], p[1], p[2], a.depth);
            else v = wn::perlin_fractal(perm, p[0], p[1], p[2]);
        }
        a.out[i] = v;
    }
}

// noise_texture::value (texture.h:37-43) with the same ballot compaction as the wavelet texture.
struct NoiseTexArgs {
    const uint8_t *perm;
    float fscale;       // (float)scale: vec3 * float (vec3.h:82-84)
    float octave_scale; // (float)pow(2, octave)
    const float *pts;
    const uint8_t *active;
    float *grey;
    size_t count;
    int points_per_wave;
};

__device__ __forceinline__ float noise_texture_value(const uint8_t *perm, const NoiseTexArgs &a, float px, float py,
                                                     float pz)
{
    return wn::noise_texture_value(perm, a.fscale, a.octave_scale, px, py, pz); // wn_texture_eval.hpp
}

template <bool MASKED>
__global__ __launch_bounds__(256) void noise_texture_kernel(const NoiseTexArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_perm[512];
    __shared__ float q_x[4][128], q_y[4][128], q_z[4][128];
    __shared__ unsigned q_i[4][128];
    wn::load_perm_lds(s_perm, a.perm);
    const uint8_t *perm = s_perm;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gwave = (size_t)blockIdx.x * 4 + wave;
    const size_t begin = gwave * (size_t)a.points_per_wave;
    if (begin >= a.count) return;
    const size_t end = min(a.count, begin + (size_t)a.points_per_wave);

    if (!MASKED) {
        for (size_t i = begin + lane; i < end; i += 64) {
            const float *p = a.pts + 3 * i;
            a.grey[i] = noise_texture_value(perm, a, p[0], p[1], p[2]);
        }
        return;
    }
    int queued = 0;
    for (size_t base = begin; base < end; base += 64) {
        const size_t i = base + lane;
        const bool hit = (i < end) && (a.active[i] != 0);
        const unsigned long long ballot = __ballot(hit);
        if (hit) {
            const int slot = queued + __popcll(ballot & ((1ull << lane) - 1ull));
            const float *p = a.pts + 3 * i;
            q_x[wave][slot] = p[0];
            q_y[wave][slot] = p[1];
            q_z[wave][slot] = p[2];
            q_i[wave][slot] = (unsigned)(i - begin);
        }
        queued += __popcll(ballot);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (queued >= 64) {
            const float x = q_x[wave][lane], y = q_y[wave][lane], z = q_z[wave][lane];
            const unsigned idx = q_i[wave][lane];
            const int rest = queued - 64;
            float cx = 0, cy = 0, cz = 0;
            unsigned ci = 0;
            if (lane < rest) {
                cx = q_x[wave][64 + lane];
                cy = q_y[wave][64 + lane];
                cz = q_z[wave][64 + lane];
                ci = q_i[wave][64 + lane];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane < rest) {
                q_x[wave][lane] = cx;
                q_y[wave][lane] = cy;
                q_z[wave][lane] = cz;
                q_i[wave][lane] = ci;
            }
            queued = rest;
            a.grey[begin + idx] = noise_texture_value(perm, a, x, y, z);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (lane < queued) {
        const unsigned idx = q_i[wave][lane];
        a.grey[begin + idx] =
            noise_texture_value(perm, a, q_x[wave][lane], q_y[wave][lane], q_z[wave][lane]);
    }
}

inline int blocks_for(size_t total)
{
    size_t b = (total + 255) / 256;
    const size_t cap = 256u * 8u * 8u;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int perlin_grid(const wn_perm *perm, const wn_grid *grid, int kind, int depth, float *out_dev,
                void *stream)
{
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    GridArgs g;
    rc = wn::check_grid(grid, true, &g);
    if (rc) return rc;
    const size_t total = (size_t)g.nx * g.ny * g.nz;
    if (total == 0) return WN_OK;
    if (!out_dev) return wn::fail(WN_ERR_INVALID, "out_dev is NULL");
    if ((size_t)g.nx * g.ny > 0xffffffffull) return wn::fail(WN_ERR_INVALID, "plane too large");
    PerlinGridArgs a{perm->dev, out_dev, g, kind, depth, 0};
    a.vec4_ok = (g.nx % 4 == 0) && ((reinterpret_cast<uintptr_t>(out_dev) & 15) == 0);
    const int octaves = kind == kNoise ? 1 : (kind == kFractal ? 6 : depth);
    const dim3 rgrid((g.nx + kRunX - 1) / kRunX, (g.ny + kRunTY - 1) / kRunTY, (g.nz + kRunTZ - 1) / kRunTZ);
    // the run kernel: rows of >= 128 samples (a lane owns 8 consecutive x samples), 1..8 octaves
    if (g.nx >= 128 && octaves >= 1 && octaves <= kRunMaxDepth && rgrid.y <= 65535u && rgrid.z <= 65535u) {
        const bool wide = octaves > 2; // 16 waves on one set of axis tables (see kRunWaves above)
        const size_t lds = run_lds_bytes(octaves, wide ? 16 : 8);
        const void *fn;
        if (wide)
            fn = kind == kNoise ? reinterpret_cast<const void *>(&perlin_grid_run_kernel<kNoise, 16>)
                 : kind == kTurb ? reinterpret_cast<const void *>(&perlin_grid_run_kernel<kTurb, 16>)
                                 : reinterpret_cast<const void *>(&perlin_grid_run_kernel<kFractal, 16>);
        else
            fn = kind == kNoise ? reinterpret_cast<const void *>(&perlin_grid_run_kernel<kNoise, 8>)
                 : kind == kTurb ? reinterpret_cast<const void *>(&perlin_grid_run_kernel<kTurb, 8>)
                                 : reinterpret_cast<const void *>(&perlin_grid_run_kernel<kFractal, 8>);
        if (lds <= 48 * 1024 || wn::ensure_dynamic_lds(fn, wn::current_device(), run_lds_bytes(kRunMaxDepth, wide ? 16 : 8))) {
            void *params[] = {&a};
            const hipError_t e = hipLaunchKernel(fn, rgrid, dim3(wide ? 1024 : 512), params, lds, wn::as_stream(stream));
            if (e != hipSuccess) return wn::hip_fail(e, "perlin_grid_run_kernel");
            WN_LAUNCH_CHECK("perlin_grid_run_kernel");
            return WN_OK;
        }
    }
    hipLaunchKernelGGL(perlin_grid_generic_kernel, dim3(blocks_for(total)), dim3(256), 0,
                       wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("perlin_grid_generic_kernel");
    return WN_OK;
}

int perlin_points(const wn_perm *perm, const double *p64, const float *p32, size_t n, int kind,
                  int depth, double *out_dev, void *stream)
{
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    if (n == 0) return WN_OK;
    if ((!p64 && !p32) || !out_dev) return wn::fail(WN_ERR_INVALID, "points/out pointer is NULL");
    PerlinPointsArgs a{perm->dev, p64, p32, out_dev, n, kind, depth};
    hipLaunchKernelGGL(perlin_points_kernel, dim3(blocks_for(n)), dim3(256), 0,
                       wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("perlin_points_kernel");
    return WN_OK;
}

} // namespace

extern "C" {

int wn_perlin_grid(const wn_perm *perm, const wn_grid *g, float *out_dev, void *stream)
{
    WN_ENTRY();
    return perlin_grid(perm, g, kNoise, 0, out_dev, stream);
}
int wn_perlin_turb_grid(const wn_perm *perm, const wn_grid *g, int depth, float *out_dev,
                        void *stream)
{
    WN_ENTRY();
    if (depth < 0) return wn::fail(WN_ERR_INVALID, "depth must be >= 0");
    return perlin_grid(perm, g, kTurb, depth, out_dev, stream);
}
int wn_perlin_fractal_grid(const wn_perm *perm, const wn_grid *g, float *out_dev, void *stream)
{
    WN_ENTRY();
    return perlin_grid(perm, g, kFractal, 0, out_dev, stream);
}
int wn_perlin_points(const wn_perm *perm, const double *xyz_dev, size_t n, double *out_dev,
                     void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, xyz_dev, nullptr, n, kNoise, 0, out_dev, stream);
}
int wn_perlin_points_vec3(const wn_perm *perm, const float *xyz_dev, size_t n, double *out_dev,
                          void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, nullptr, xyz_dev, n, kNoise, 0, out_dev, stream);
}
int wn_perlin_turb_points(const wn_perm *perm, const float *xyz_dev, size_t n, int depth,
                          double *out_dev, void *stream)
{
    WN_ENTRY();
    if (depth < 0) return wn::fail(WN_ERR_INVALID, "depth must be >= 0");
    return perlin_points(perm, nullptr, xyz_dev, n, kTurb, depth, out_dev, stream);
}
int wn_perlin_fractal_points(const wn_perm *perm, const float *xyz_dev, size_t n, double *out_dev,
                             void *stream)
{
    WN_ENTRY();
    return perlin_points(perm, nullptr, xyz_dev, n, kFractal, 0, out_dev, stream);
}

int wn_noise_texture_points(const wn_perm *perm, double scale, int octave, const float *xyz_dev,
                            const uint8_t *active_dev, size_t n, float *grey_dev, void *stream)
{
    WN_ENTRY();
    int rc = wn::require_device();
    if (rc) return rc;
    if (!perm) return wn::fail(WN_ERR_INVALID, "perm is NULL");
    if ((rc = wn::check_handle_device(perm->device, "perm")) != WN_OK) return rc;
    if (n == 0) return WN_OK;
    if (!xyz_dev || !grey_dev) return wn::fail(WN_ERR_INVALID, "points/grey pointer is NULL");
    NoiseTexArgs a{};
    a.perm = perm->dev;
    a.fscale = (float)scale;
    a.octave_scale = (float)std::pow(2.0, (double)octave); // texture.h:38
    a.pts = xyz_dev;
    a.active = active_dev;
    a.grey = grey_dev;
    a.count = n;
    a.points_per_wave = 1024;
    const size_t waves = (n + a.points_per_wave - 1) / a.points_per_wave;
    const size_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffull) return wn::fail(WN_ERR_INVALID, "too many points");
    if (active_dev)
        hipLaunchKernelGGL(noise_texture_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                           wn::as_stream(stream), a);
    else
        hipLaunchKernelGGL(noise_texture_kernel<false>, dim3((unsigned)blocks), dim3(256), 0,
                           wn::as_stream(stream), a);
    WN_LAUNCH_CHECK("noise_texture_kernel");
    return WN_OK;
}

} // extern "C"
