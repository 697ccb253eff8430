// wn_wavelet_points.hip -- wavelet noise on arbitrary point lists (K2 / K3-points / K3p of
// SURVEY.md 8) and the wavelet_texture adaptor over ray hit points.
//
// One point per lane; the 27 (or 9) coefficient gathers come straight from the tile in HBM
// (8 MiB: L2 / Infinity-Cache resident), arithmetic in the reference's order and unfused, so
// every result is bit-identical to the scalar CPU member it batches.  The texture kernel takes
// an optional per-point `active` byte: each wave compacts its active hits with __ballot +
// prefix popcount into a per-wave LDS queue and only runs the gather loop on full 64-lane
// batches, so rays that missed the noise-textured surfaces cost no gather slots.
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"
#include "wn_texture_eval.hpp"

#include <cmath>
#include <cstdlib>

namespace {

constexpr int kMaxBands = 8;

struct PointsArgs {
    const float *coef;
    int n, nmask;
    const float *pts;     // xyz (or xy) interleaved
    const float *normals; // projected only
    float *out;
    size_t count;
    // multiband
    int nbands;
    float band_scale[kMaxBands], band_w[kMaxBands];
    float out_div;
    int apply_div;
    int one_normal; // multiband projected: `normals` holds ONE normal for all points
};

template <bool PADDED>
__global__ __launch_bounds__(256) void eval3d_points_kernel(const PointsArgs a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float *p = a.pts + 3 * i;
        a.out[i] = wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, p[0], p[1], p[2]);
    }
}

__global__ __launch_bounds__(256) void eval2d_points_kernel(const PointsArgs a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float *p = a.pts + 2 * i;
        a.out[i] = wn::eval2d_exact(a.coef, a.n, a.nmask, p[0], p[1]);
    }
}

__global__ __launch_bounds__(256) void eval3d_projected_points_kernel(const PointsArgs a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float p[3] = {a.pts[3 * i], a.pts[3 * i + 1], a.pts[3 * i + 2]};
        const float nr[3] = {a.normals[3 * i], a.normals[3 * i + 1], a.normals[3 * i + 2]};
        a.out[i] = wn::projected_exact(a.coef, a.n, a.nmask, p, nr);
    }
}

// WMultibandNoise (paper Appendix 2, normal == NULL) per point.
template <bool PADDED>
__global__ __launch_bounds__(256) void multiband3d_points_kernel(const PointsArgs a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float *p = a.pts + 3 * i;
        float v = 0.0f;
        for (int b = 0; b < a.nbands; ++b) {
            const float s = a.band_scale[b];
            v += a.band_w[b] * wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, 2.0f * p[0] * s,
                                                2.0f * p[1] * s, 2.0f * p[2] * s);
        }
        if (a.apply_div) v /= a.out_div;
        a.out[i] = v;
    }
}

// WMultibandNoise, normal != NULL branch: every band is evaluate3DProjected (WaveletNoise.cpp:218-265).
__global__ __launch_bounds__(256) void multiband3d_projected_points_kernel(const PointsArgs a)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count;
         i += (size_t)gridDim.x * blockDim.x) {
        const float *p = a.pts + 3 * i;
        const float *nrp = a.normals + (a.one_normal ? 0 : 3 * i);
        const float nr[3] = {nrp[0], nrp[1], nrp[2]};
        float v = 0.0f;
        for (int b = 0; b < a.nbands; ++b) {
            const float s = a.band_scale[b];
            const float q[3] = {2.0f * p[0] * s, 2.0f * p[1] * s, 2.0f * p[2] * s};
            v += a.band_w[b] * wn::projected_exact(a.coef, a.n, a.nmask, q, nr);
        }
        if (a.apply_div) v /= a.out_div;
        a.out[i] = v;
    }
}

// ---- wavelet_texture::value (texture.h:67-107) --------------------------------------------------
struct TexArgs {
    const float *coef;
    int n, nmask;
    int mode; // 3: evaluate3D branch, 2: evaluate2D branch, 0: no tile (texture.h:100-102)
    double scale;
    float octave_mul; // octave_scale * 2.0f   (texture.h:77-80)
    float inv_stddev; // 1/sqrt(0.18402f) or 1/sqrt(0.19686f)
    const float *pts;
    const uint8_t *active;
    float *grey;
    size_t count;
    int points_per_wave;
};

using wn::wavelet_texture_value; // wn_texture_eval.hpp

template <bool MASKED, bool PADDED>
__global__ __launch_bounds__(256) void wavelet_texture_kernel(const TexArgs a)
{
    // per-wave compaction queue: up to 63 carried + 64 new hits
    __shared__ float q_x[4][128], q_y[4][128], q_z[4][128];
    __shared__ unsigned q_i[4][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t gwave = (size_t)blockIdx.x * 4 + wave;
    const size_t begin = gwave * (size_t)a.points_per_wave;
    if (begin >= a.count) return;
    const size_t end = min(a.count, begin + (size_t)a.points_per_wave);

    if (!MASKED) {
        for (size_t i = begin + lane; i < end; i += 64) {
            const float *p = a.pts + 3 * i;
            a.grey[i] = wavelet_texture_value<PADDED>(a, p[0], p[1], p[2]);
        }
        return;
    }

    int queued = 0; // wave-uniform
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
        if (queued >= 64) { // a full batch: every lane gathers
            const float x = q_x[wave][lane], y = q_y[wave][lane], z = q_z[wave][lane];
            const unsigned idx = q_i[wave][lane];
            // carry the remainder down before the (long) evaluation
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
            a.grey[begin + idx] = wavelet_texture_value<PADDED>(a, x, y, z);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
    if (lane < queued) { // tail batch
        const unsigned idx = q_i[wave][lane];
        a.grey[begin + idx] =
            wavelet_texture_value<PADDED>(a, q_x[wave][lane], q_y[wave][lane], q_z[wave][lane]);
    }
}

// Long point lists (wavelet_texture::value, evaluate3D, WMultibandNoise): chunks of points taken in z-plane order.
// Ray hits are far from uniformly random in 3-D: most lie on planar surfaces (the reference's ground quad: y = const), so points that share the coefficient
// plane mz of their z tap also share its few (y, z) rows -- 512-byte rows that 64 random x positions cover whole.  In
// stream order a wave's 64 points touch ~576 lines; taken in mz order they share most of them (the gathers are bound by
// L1 misses: profiles/r02_point_kernel_tile_size_probe.txt).  A workgroup counting-sorts the indices of its chunk by mz
// in LDS (one LDS atomic per point), then evaluates the points in that order with the same per-point function as the
// unsorted kernel: every result is the same float, only the order of evaluation changes.
// The key's low bits are the 64-byte line of the x tap, so that neighbouring lanes also gather from the same lines.
// The values return to stream order in LDS and leave as full lines.
// A chunk whose stream order is already coherent (the renderer's primary hits: consecutive samples of one pixel) is
// evaluated in stream order without the sorting passes: the workgroup looks at its first 256 points and sorts only when
// most neighbours of the stream lie in different planes.  Measured (80 M uniformly random quad / sphere hits): 2.65 ms
// in stream order; 2.05 ms sorted with the values stored straight to memory (scattered 4-byte stores: WRITE_SIZE 10x);
// 1.27 ms with the values staged; 1.19 ms with the x bits in the key.  The renderer's real stream: 74 -> 83 G points/s
// (always sorting cost it 16 %; 16 K-point chunks with a two-pass sort were slower on both).  Finer x keys (4 / 5 bits, 8 / 16 KB
// of histogram) sort the stand-in 5 / 6.5 % faster but their LDS costs the stream-order chunks a workgroup per CU: the
// renderer's stream 83 -> 81 / 74 G points/s.  (With histogram and values sharing their LDS all three run alike: the sorted
// path likes four workgroups per CU -- their rows fit the L1 together -- the stream-order path more.)
constexpr int kSortChunk = 4096, kSortPlanes = 128, kSortXBits = 3, kSortPerThread = kSortChunk / 256;
constexpr int kSortBins = kSortPlanes << kSortXBits;
constexpr size_t kSortMinPoints = 16 * (size_t)kSortChunk; // shorter lists: the plain kernels

// The kernel is generic over what a point is (`Ops`): count; active(i); mid_z(i) / mid_x(i) = the coefficient index of the
// z / x tap's middle (of the finest band, where there are several); eval(i); store(i, v).
// A chunk left to row_slab_points_kernel is marked in the output itself: this value (a NaN no evaluation produces; if one
// ever did, the chunk would be evaluated twice to the same floats) in the chunk's first element.
constexpr unsigned kDeferredBits = 0xffc0de42u;
constexpr int kDeferMarks = 16, kDeferMarkStride = 64; // row_slab_points_kernel: 16 waves, wave w stores element 64 w first

template <typename Ops, bool DEFER = false>
__global__ __launch_bounds__(256) void plane_sorted_points_kernel(const Ops ops)
{
    constexpr int kPer = kSortBins / 256;
    __shared__ unsigned hist[kSortBins];
    __shared__ unsigned s_changes, wave_total[4];
    __shared__ unsigned short order[kSortChunk];
    __shared__ float value[kSortChunk];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t begin = (size_t)blockIdx.x * kSortChunk;
    const int count = (int)min((size_t)kSortChunk, ops.count - begin);
    if constexpr (DEFER) {
        // Is this chunk one for the row-slab kernel that follows on the stream -- hits scattered over an axis-aligned surface?
        // The first wave looks at 32 pairs of neighbours spread over the chunk before anything else happens (one memory round
        // trip, one barrier: 16,600 of the stand-in's 19,531 workgroups end here): at least half of the pairs on different
        // planes, hardly any in neighbouring cells, three quarters of the 64 points on one middle y row.  Every other chunk
        // stays: coherent ones, and the renderer's bounce-interleaved hits, which sort well (this kernel's 20 waves per CU beat
        // the slab kernel's 16 on them).  Whole chunks only: every wave of the slab kernel has its mark.
        __shared__ int s_defer;
        if (count == kSortChunk) {
            if (wave == 0) {
                const size_t si = begin + (size_t)(lane >> 1) * (kSortChunk / 32) + (lane & 1);
                const int kz = ops.mid_z(si), kx = ops.mid_x(si), ky = ops.mid_y(si);
                const int pz = __shfl_up(kz, 1, 64), px = __shfl_up(kx, 1, 64);
                const bool second = lane & 1;
                const int changes = __popcll(__ballot(second && pz != kz));
                const int near = __popcll(__ballot(second && abs(pz - kz) <= 1 && abs(px - kx) <= 1));
                unsigned long long rest = ~0ull;
                int top = 0;
                for (int c = 0; c < 3 && rest; ++c) { // the most frequent row is among the first three distinct ones, or none is frequent
                    const int r = __shfl(ky, __ffsll((long long)rest) - 1, 64);
                    const unsigned long long same = __ballot(ky == r);
                    top = max(top, (int)__popcll(same));
                    rest &= ~same;
                }
                if (lane == 0) s_defer = changes >= 16 && near <= 3 && top >= 48;
            }
            __syncthreads();
            if (s_defer) {
                // (one mark per wave of that kernel, where the wave's own first store goes: its waves run through such chunks
                // without a barrier, and a mark one wave has overwritten must not tell another that the chunk is done)
                if (tid < kDeferMarks) ops.store(begin + kDeferMarkStride * tid, __uint_as_float(kDeferredBits));
                return;
            }
        }
    }
    for (int b = tid; b < kSortBins; b += 256) hist[b] = 0;
    if (tid == 0) s_changes = 0;
    __syncthreads();
    // sample (the chunk's first 256 points): do neighbours of the stream change plane?  (A second test, "... and share
    // rows", would spare lists scattered in all three dimensions the 5 % the sorting passes cost them -- 25.4 -> 24.1 G
    // points/s -- but it also turns away curved surfaces, which gain: the stand-in's sphere hits, 1.19 -> 1.34 ms.)
    {
        const int k = ops.mid_z(begin + min(tid, count - 1));
        const int prev = __shfl_up(k, 1, 64);
        const unsigned long long diff = __ballot(lane != 0 && prev != k);
        if (lane == 0) atomicAdd(&s_changes, (unsigned)__popcll(diff));
    }
    __syncthreads();
    if (s_changes < 128) { // coherent already: stream order, no sorting passes
        for (int i = tid; i < count; i += 256)
            if (ops.active(begin + i)) ops.store(begin + i, ops.eval(begin + i));
        return;
    }
    // pass 1: bin and rank of every point (one LDS atomic each)
    unsigned short key[kSortPerThread], rank[kSortPerThread];
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k) {
        const int i = tid + 256 * k;
        key[k] = 0xffff;
        rank[k] = 0;
        if (i < count && ops.active(begin + i)) {
            const int bin = ((ops.mid_z(begin + i) & (kSortPlanes - 1)) << kSortXBits) | ((ops.mid_x(begin + i) & 127) >> (7 - kSortXBits));
            key[k] = (unsigned short)bin;
            rank[k] = (unsigned short)atomicAdd(&hist[bin], 1u);
        }
    }
    __syncthreads();
    { // exclusive prefix sum of the bins: kPer bins per thread, wave scan, wave totals
        unsigned mine[kPer], sum = 0;
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            mine[j] = hist[tid * kPer + j];
            sum += mine[j];
        }
        unsigned inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_total[wave] = inc;
        __syncthreads();
        unsigned run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wave_total[w];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            hist[tid * kPer + j] = run;
            run += mine[j];
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kSortPerThread; ++k)
        if (key[k] != 0xffff) order[hist[key[k]] + rank[k]] = (unsigned short)(tid + 256 * k);
    __syncthreads();
    // pass 2: evaluate in bin order; the values go back to stream order in LDS, so that the chunk leaves as full
    // lines (scattered 4-byte stores straight to memory: 10x the write traffic at the fabric, WRITE_SIZE 3.1 GB)
    const int n_active = (int)(wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3]);
    for (int s2 = tid; s2 < n_active; s2 += 256) {
        const int i = order[s2];
        value[i] = ops.eval(begin + i);
    }
    __syncthreads();
    for (int i = tid; i < count; i += 256)
        if (ops.active(begin + i)) ops.store(begin + i, value[i]);
}

__device__ __forceinline__ int mid_of(float p) { return (int)ceilf(p - 0.5f); } // WaveletNoise.cpp:194-196

// wavelet_texture::value (texture.h:67-107) on a 3-D tile
template <bool MASKED, bool PADDED>
struct TextureOps {
    TexArgs a;
    size_t count;
    __device__ bool active(size_t i) const { return !MASKED || a.active[i] != 0; }
    __device__ int mid(float c) const // texture.h:71-80
    {
        float p = (float)((double)c * a.scale);
        p *= a.octave_mul;
        return mid_of(p);
    }
    __device__ int mid_z(size_t i) const { return mid(a.pts[3 * i + 2]); }
    __device__ int mid_y(size_t i) const { return mid(a.pts[3 * i + 1]); }
    __device__ int mid_x(size_t i) const { return mid(a.pts[3 * i]); }
    __device__ float eval(size_t i) const
    {
        const float *p = a.pts + 3 * i;
        return wavelet_texture_value<PADDED>(a, p[0], p[1], p[2]);
    }
    __device__ float eval_rowslab(size_t i, const float *slab, int ry) const // padded 3-D tile only
    {
        const float *p = a.pts + 3 * i;
        return wn::wavelet_texture_value_rowslab(a, p[0], p[1], p[2], slab, ry, slab, 0); // (no third row: the plane-ordered path)
    }
    __device__ const float *point(size_t i) const { return a.pts + 3 * i; }
    __device__ float eval_rowslab_at(float x, float y, float z, const float *slab, int ry, const float *third, int third_planes) const
    {
        return wn::wavelet_texture_value_rowslab(a, x, y, z, slab, ry, third, third_planes);
    }
    __device__ const float *padded_tile() const { return a.coef; }
    __device__ int tile_n() const { return a.n; }
#ifdef WN_TUNE_ENV
    int tune_share = 3;
#endif
    __device__ void store(size_t i, float v) const { a.grey[i] = v; }
    __device__ float stored(size_t i) const { return a.grey[i]; }
};

// WaveletNoise::evaluate3D per point; nbands > 0: WMultibandNoise (keyed by its finest band, the last one)
template <bool PADDED, bool MULTIBAND>
struct Eval3dOps {
    PointsArgs a;
    size_t count;
    __device__ bool active(size_t) const { return true; }
    __device__ float key_scale() const { return a.band_scale[a.nbands - 1]; }
    __device__ int mid_z(size_t i) const { return MULTIBAND ? mid_of(2.0f * a.pts[3 * i + 2] * key_scale()) : mid_of(a.pts[3 * i + 2]); }
    __device__ int mid_x(size_t i) const { return MULTIBAND ? mid_of(2.0f * a.pts[3 * i] * key_scale()) : mid_of(a.pts[3 * i]); }
    __device__ int mid_y(size_t i) const { return mid_of(a.pts[3 * i + 1]); } // (single band: the row-slab kernel)
    __device__ float eval_rowslab(size_t i, const float *slab, int ry) const  // single band, padded tile only
    {
        const float *p = a.pts + 3 * i;
        return wn::eval3d_exact_rowslab(a.coef, a.n, a.nmask, p[0], p[1], p[2], slab, ry, slab, 0); // (no third row: the plane-ordered path)
    }
    __device__ const float *point(size_t i) const { return a.pts + 3 * i; }
    __device__ float eval_rowslab_at(float x, float y, float z, const float *slab, int ry, const float *third, int third_planes) const
    {
        return wn::eval3d_exact_rowslab(a.coef, a.n, a.nmask, x, y, z, slab, ry, third, third_planes);
    }
    __device__ const float *padded_tile() const { return a.coef; }
    __device__ int tile_n() const { return a.n; }
#ifdef WN_TUNE_ENV
    int tune_share = 3;
#endif
    __device__ float eval(size_t i) const
    {
        const float *p = a.pts + 3 * i;
        if (!MULTIBAND) return wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, p[0], p[1], p[2]);
        float v = 0.0f;
        for (int b = 0; b < a.nbands; ++b) {
            const float s = a.band_scale[b];
            v += a.band_w[b] * wn::eval3d_exact<PADDED>(a.coef, a.n, a.nmask, 2.0f * p[0] * s, 2.0f * p[1] * s, 2.0f * p[2] * s);
        }
        if (a.apply_div) v /= a.out_div;
        return v;
    }
    __device__ void store(size_t i, float v) const { a.out[i] = v; }
    __device__ float stored(size_t i) const { return a.out[i]; }
};

// Long lists on a padded 128^3 tile: the plane-ordered chunks above, evaluated by persistent 1024-thread workgroups (one per
// CU) that keep TWO y rows of every z plane of the tile in LDS (130 KiB: rows ry - 1 and ry, [z][2][n + 2]).
// plane_sorted_points_kernel is bound by the texture addresser / L1 at one lane-access per clock and CU: nine divergent
// 12-byte gathers per point (profiles/r03a_texture_points_ta_counters.json), 2.7x its VALU time.  Ray hits lie on surfaces;
// on an axis-aligned one (the reference scene's ground quad y = -0.5: 85 % of its hits) every point has the same three y
// rows.  With two of them in LDS such a point gathers 3 times from memory instead of 9 and reads the rest from LDS
// (eval3d_exact_rowslab: same values, same order of products and sums -> the same bits); any other point gathers all nine
// as before.  Each chunk counts its points' middle y rows beside the sort keys; when another row than the resident one holds
// a quarter of the chunk, the workgroup loads that row pair instead (130 KiB from the L2-resident tile, ~2 us), so a stream
// that moves from one surface to the next is followed.  No dominant row: the kernel is the plane-ordered one at the same
// occupancy (16 waves per CU).
constexpr int kSlabThreads = 1024, kSlabPerThread = kSortChunk / kSlabThreads, kSlabTile = 128, kSlabTrust = 15;
constexpr size_t kSlabFloats = (size_t)kSlabTile * 2 * (kSlabTile + 2);
constexpr size_t kSlabScratchBytes = kSortBins * 4 + kSortChunk * 4 + kSortChunk * 2;
constexpr int kSlabThirdPlanes = (int)(kSlabScratchBytes / ((kSlabTile + 2) * 4));
constexpr size_t kSlabLdsBytes = kSlabFloats * sizeof(float) + kSlabScratchBytes;
// The pair of launches costs ~7 us more than plane_sorted_points_kernel alone when that kernel ends up keeping every chunk (the
// renderer's stream: its marks are looked for in vain); from 16.8 M points that is < 5 % of the call.
constexpr size_t kSlabMinPoints = 16 * 256 * (size_t)kSortChunk;

template <typename Ops>
__global__ __launch_bounds__(kSlabThreads) void row_slab_points_kernel(const Ops ops, const int nchunks, const int deferred_only)
{
    static_assert(kSortBins == kSlabThreads, "one bin per thread in the prefix sum");
    static_assert(kDeferMarks * 64 == kSlabThreads && kDeferMarkStride == 64, "one mark per wave, at the wave's first element");
    extern __shared__ __attribute__((aligned(16))) float slab[]; // [z][2][n + 2], then kSlabScratchBytes of scratch
    // the scratch is the plane-ordered path's histogram, order and values -- or, while chunks are taken in stream order, the
    // slab's THIRD row (ry + 1) for as many planes as fit (55 of 128): those points gather once or not at all
    unsigned *const hist = reinterpret_cast<unsigned *>(slab + kSlabFloats);
    float *const value = reinterpret_cast<float *>(hist + kSortBins);
    unsigned short *const order = reinterpret_cast<unsigned short *>(value + kSortChunk);
    float *const third = slab + kSlabFloats;
    __shared__ unsigned rowhist[kSlabTile];
    __shared__ unsigned s_changes, s_best, wave_total[kSlabThreads / 64];
    bool third_ok = false; // the scratch holds the third row of the resident pair
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = ops.tile_n(), stride = n + 2; // n == kSlabTile (host)
    int slab_row = -1;                          // row ry of the resident pair; -1: none yet
    int trusted = 0, trusted_plain = 0;         // chunks still to be taken in stream order without looking at their rows
    // a chunk in stream order: the thread's four points are requested together, then evaluated one after the other
    auto stream_order = [&](size_t begin, int count) {
        float xyz[kSlabPerThread][3];
        bool on[kSlabPerThread];
#pragma unroll
        for (int k = 0; k < kSlabPerThread; ++k) {
            const int i = tid + kSlabThreads * k;
            on[k] = i < count && ops.active(begin + i);
            const float *p = ops.point(begin + min(i, count - 1));
            xyz[k][0] = p[0];
            xyz[k][1] = p[1];
            xyz[k][2] = p[2];
        }
#pragma unroll
        for (int k = 0; k < kSlabPerThread; ++k)
            if (on[k])
                ops.store(begin + tid + kSlabThreads * k,
                          ops.eval_rowslab_at(xyz[k][0], xyz[k][1], xyz[k][2], slab, slab_row, third, third_ok ? kSlabThirdPlanes : 0));
    };
    // The workgroup's chunks (blockIdx.x, + gridDim.x, ...) in groups of 64.  After plane_sorted_points_kernel<Ops, true> on
    // the same stream only the chunks it left: a wave reads the marks of a group's chunks at once -- one load per lane, each
    // wave the marks only it overwrites, all the same -- instead of paying a memory round trip per chunk to learn it is done.
    for (int g0 = blockIdx.x; g0 < nchunks; g0 += 64 * (int)gridDim.x) {
    unsigned long long todo = ~0ull;
    if (deferred_only) {
        const long long c = (long long)g0 + (long long)lane * gridDim.x;
        bool left = false;
        if (c < nchunks && ((size_t)c + 1) * kSortChunk <= ops.count) // (only whole chunks are ever left)
            left = __float_as_uint(ops.stored((size_t)c * kSortChunk + kDeferMarkStride * wave)) == kDeferredBits;
        todo = __ballot(left);
    }
    for (int j = 0; j < 64; ++j) {
        const long long chunk = (long long)g0 + (long long)j * gridDim.x;
        if (chunk >= nchunks) break;
        if (!((todo >> j) & 1ull)) continue;
        const size_t begin = (size_t)chunk * kSortChunk;
        const int count = (int)min((size_t)kSortChunk, ops.count - begin);
        // After a chunk that mostly read from the slab, the next kSlabTrust chunks are evaluated in stream order unseen: no
        // histogram, no barrier, nothing written to LDS -- the waves run free.  (A point off the slab's rows gathers all nine
        // triples, as ever: what a wrong guess costs is speed.)
        if (trusted > 0) {
            --trusted;
            stream_order(begin, count);
            continue;
        }
        if (trusted_plain > 0) { // after a coherent chunk: the same for its successors, unseen
            --trusted_plain;
            for (int i = tid; i < count; i += kSlabThreads)
                if (ops.active(begin + i)) ops.store(begin + i, ops.eval(begin + i));
            continue;
        }
        __syncthreads(); // free-running chunks end here: the histograms are rewritten, the slab may be replaced
        if (tid < kSlabTile) rowhist[tid] = 0;
        if (tid == 0) {
            s_changes = 0;
            s_best = 0;
        }
        __syncthreads();
        if (tid < 256) { // is the stream already coherent?  (as in plane_sorted_points_kernel)
            const int k = ops.mid_z(begin + min(tid, count - 1));
            const int prev = __shfl_up(k, 1, 64);
            const unsigned long long diff = __ballot(lane != 0 && prev != k);
            if (lane == 0) atomicAdd(&s_changes, (unsigned)__popcll(diff));
        }
        // the middle y rows of a sample of the chunk: every fourth point
        int sampled = 0;
        {
            const int i = tid * kSlabPerThread;
            if (i < count && ops.active(begin + i)) {
                atomicAdd(&rowhist[ops.mid_y(begin + i) & (kSlabTile - 1)], 1u);
                sampled = 1;
            }
        }
        sampled = __syncthreads_count(sampled);
        if (tid < kSlabTile) { // the row most points share: (count << 8 | row), largest wins
            unsigned best = (rowhist[tid] << 8) | (unsigned)tid;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) best = max(best, (unsigned)__shfl_xor((int)best, d, 64));
            if (lane == 0) atomicMax(&s_best, best);
        }
        __syncthreads();
        const int best_row = (int)(s_best & 255u), best_count = (int)(s_best >> 8);
        const bool reload = best_count * 4 >= sampled && sampled > 0 && best_row != slab_row; // workgroup-uniform
        if (reload) { // rows best_row - 1 and best_row of every plane; the last chunk's readers passed the barrier at its end
            const int r0 = (best_row + n - 1) & (kSlabTile - 1);
            const float *tile = ops.padded_tile();
            for (int i = tid; i < (int)kSlabFloats; i += kSlabThreads) {
                const int row = i / stride, col = i - row * stride; // row = z * 2 + f
                const int z = row >> 1, y = (row & 1) ? best_row : r0;
                slab[i] = tile[((size_t)z * n + y) * stride + col];
            }
            slab_row = best_row;
            third_ok = false;
            __syncthreads();
        }
        // stream order when the stream is coherent already, or when most of the chunk reads from the slab (what is left to
        // gather then is one row of every plane, 66 KB: the plane order buys little)
        int slab_share_min = 3;
#ifdef WN_TUNE_ENV
        slab_share_min = ops.tune_share;
#endif
        if (s_changes < 128) { // coherent already: neighbours share their lines, plain gathers in stream order are the fastest form
            for (int i = tid; i < count; i += kSlabThreads)
                if (ops.active(begin + i)) ops.store(begin + i, ops.eval(begin + i));
            trusted_plain = kSlabTrust;
            continue;
        }
        if (best_row == slab_row && best_count * 4 >= sampled * slab_share_min) {
            if (!third_ok && slab_row >= 0) { // (the scratch's last users passed the barrier at the loop's top)
                const int r2 = (slab_row + 1) & (kSlabTile - 1);
                const float *tile = ops.padded_tile();
                for (int i = tid; i < kSlabThirdPlanes * stride; i += kSlabThreads) {
                    const int z = i / stride, col = i - z * stride;
                    third[i] = tile[((size_t)z * n + r2) * stride + col];
                }
                third_ok = true;
                __syncthreads();
            }
            stream_order(begin, count);
            if (best_row == slab_row && best_count * 4 >= sampled * slab_share_min) trusted = kSlabTrust;
            continue;
        }
        // pass 1: bin and rank of every point (the scratch becomes the histogram: zeroed here)
        third_ok = false;
        hist[tid] = 0;
        __syncthreads();
        unsigned short key[kSlabPerThread], rank[kSlabPerThread];
#pragma unroll
        for (int k = 0; k < kSlabPerThread; ++k) {
            const int i = tid + kSlabThreads * k;
            key[k] = 0xffff;
            rank[k] = 0;
            if (i < count && ops.active(begin + i)) {
                const int bin = ((ops.mid_z(begin + i) & (kSortPlanes - 1)) << kSortXBits) | ((ops.mid_x(begin + i) & 127) >> (7 - kSortXBits));
                key[k] = (unsigned short)bin;
                rank[k] = (unsigned short)atomicAdd(&hist[bin], 1u);
            }
        }
        __syncthreads();
        const unsigned mine = hist[tid]; // exclusive prefix sum of the bins: one per thread
        unsigned inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        if (lane == 63) wave_total[wave] = inc;
        __syncthreads();
        {
            unsigned run = inc - mine;
            for (int w = 0; w < wave; ++w) run += wave_total[w];
            hist[tid] = run;
        }
        int n_active = 0;
#pragma unroll
        for (int w = 0; w < kSlabThreads / 64; ++w) n_active += (int)wave_total[w];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kSlabPerThread; ++k)
            if (key[k] != 0xffff) order[hist[key[k]] + rank[k]] = (unsigned short)(tid + kSlabThreads * k);
        __syncthreads();
        // pass 2: evaluate in bin order; the values return to stream order in LDS and leave as full lines
        for (int s2 = tid; s2 < n_active; s2 += kSlabThreads) {
            const int i = order[s2];
            value[i] = ops.eval_rowslab(begin + i, slab, slab_row);
        }
        __syncthreads();
        for (int i = tid; i < count; i += kSlabThreads)
            if (ops.active(begin + i)) ops.store(begin + i, value[i]);
        __syncthreads(); // the next chunk zeroes the histograms and may replace the slab
    }
    }
}

// Long unmasked lists on a padded 128^3 tile: plane_sorted_points_kernel evaluates the chunks whose stream order is coherent
// already (many small workgroups, 20 waves per CU: the renderer's primary hits run 20 % faster there than in the 16 waves of
// the slab kernel) and leaves the others, marked, to row_slab_points_kernel, which follows on the same stream.
// false: not in this regime (the caller launches plane_sorted_points_kernel alone).
template <typename Ops>
bool launch_row_slab(Ops ops, int n, bool masked, hipStream_t stream)
{
    if (n != kSlabTile || masked || ops.count < kSlabMinPoints) return false;
#ifdef WN_TUNE_ENV
    if (getenv("WN_NO_ROW_SLAB")) return false;
    if (const char *e = getenv("WN_ROW_SLAB_SHARE")) ops.tune_share = atoi(e); // quarters of a chunk; 5: never
#endif
    const size_t chunks = (ops.count + kSortChunk - 1) / kSortChunk;
    if (chunks > 0x7fffffffull) return false;
    const void *fn = reinterpret_cast<const void *>(&row_slab_points_kernel<Ops>);
    const int dev = wn::current_device();
    if (!wn::ensure_dynamic_lds(fn, dev, kSlabLdsBytes)) return false; // the runtime refused the LDS opt-in
    const int grid = (int)std::min<size_t>(chunks, (size_t)wn::device_compute_units(dev));
    hipLaunchKernelGGL((plane_sorted_points_kernel<Ops, true>), dim3((unsigned)chunks), dim3(256), 0, stream, ops);
#ifdef WN_TUNE_ENV
    if (getenv("WN_ROW_SLAB_FIRST_ONLY")) return true; // experiment: the marks stay in the output
#endif
    hipLaunchKernelGGL((row_slab_points_kernel<Ops>), dim3((unsigned)grid), dim3(kSlabThreads), kSlabLdsBytes, stream, ops, (int)chunks, 1);
    return true;
}

template <typename Ops>
int launch_sorted(const Ops &ops, hipStream_t stream)
{
    const size_t blocks = (ops.count + kSortChunk - 1) / kSortChunk;
    if (blocks > 0x7fffffffull) return WN_ERR_INVALID;
    hipLaunchKernelGGL((plane_sorted_points_kernel<Ops, false>), dim3((unsigned)blocks), dim3(256), 0, stream, ops);
    return WN_OK;
}

inline bool sort_enabled()
{
#ifdef WN_TUNE_ENV
    if (getenv("WN_NO_POINT_SORT")) return false;
#endif
    return true;
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }
inline int point_blocks(size_t total)
{
    size_t b = (total + 255) / 256;
    const size_t cap = 256u * 8u * 8u;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int fill_common(const wn_tile *tile, int dims, const void *pts, size_t n, const void *out,
                PointsArgs *a)
{
    if (!tile) return wn::fail(WN_ERR_INVALID, "tile is NULL");
    if (tile->count) {
        const int rc = wn::check_handle_device(tile->device, "tile");
        if (rc) return rc;
    }
    if (tile->count && tile->dims != dims)
        return wn::fail(WN_ERR_INVALID, "tile is %d-D, this entry point needs %d-D", tile->dims, dims);
    if (n && (!pts || !out)) return wn::fail(WN_ERR_INVALID, "points/out pointer is NULL");
    a->coef = (dims == 3 && tile->dev_padded) ? tile->dev_padded : tile->dev;
    a->n = tile->n;
    a->nmask = pow2_mask(tile->n);
    a->count = n;
    return WN_OK;
}

} // namespace

using namespace wn;

extern "C" {

int wn_eval3d_points(const wn_tile *tile, const float *xyz_dev, size_t n, float *out_dev,
                     void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    PointsArgs a{};
    rc = fill_common(tile, 3, xyz_dev, n, out_dev, &a);
    if (rc || n == 0) return rc;
    a.pts = xyz_dev;
    a.out = out_dev;
    if (n >= kSortMinPoints && a.n > 0 && sort_enabled()) { // long lists: chunks in z-plane order (plane_sorted_points_kernel)
        if (tile->dev_padded && launch_row_slab(Eval3dOps<true, false>{a, n}, a.n, false, as_stream(stream))) {
            WN_LAUNCH_CHECK("row_slab_points_kernel(evaluate3D)");
            return WN_OK;
        }
        const int lrc = tile->dev_padded ? launch_sorted(Eval3dOps<true, false>{a, n}, as_stream(stream))
                                         : launch_sorted(Eval3dOps<false, false>{a, n}, as_stream(stream));
        if (lrc) return fail(lrc, "too many points");
        WN_LAUNCH_CHECK("plane_sorted_points_kernel(evaluate3D)");
        return WN_OK;
    }
    if (tile->dev_padded)
        hipLaunchKernelGGL(eval3d_points_kernel<true>, dim3(point_blocks(n)), dim3(256), 0, as_stream(stream), a);
    else
        hipLaunchKernelGGL(eval3d_points_kernel<false>, dim3(point_blocks(n)), dim3(256), 0, as_stream(stream), a);
    WN_LAUNCH_CHECK("eval3d_points_kernel");
    return WN_OK;
}

int wn_eval2d_points(const wn_tile *tile, const float *xy_dev, size_t n, float *out_dev,
                     void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    PointsArgs a{};
    rc = fill_common(tile, 2, xy_dev, n, out_dev, &a);
    if (rc || n == 0) return rc;
    a.pts = xy_dev;
    a.out = out_dev;
    hipLaunchKernelGGL(eval2d_points_kernel, dim3(point_blocks(n)), dim3(256), 0, as_stream(stream), a);
    WN_LAUNCH_CHECK("eval2d_points_kernel");
    return WN_OK;
}

int wn_eval3d_projected_points(const wn_tile *tile, const float *xyz_dev, const float *normals_dev,
                               size_t n, float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    PointsArgs a{};
    rc = fill_common(tile, 3, xyz_dev, n, out_dev, &a);
    if (rc || n == 0) return rc;
    if (!normals_dev) return fail(WN_ERR_INVALID, "normals_dev is NULL");
    a.coef = tile->dev; // the projected evaluator indexes the linear layout
    a.pts = xyz_dev;
    a.normals = normals_dev;
    a.out = out_dev;
    hipLaunchKernelGGL(eval3d_projected_points_kernel, dim3(point_blocks(n)), dim3(256), 0,
                       as_stream(stream), a);
    WN_LAUNCH_CHECK("eval3d_projected_points_kernel");
    return WN_OK;
}

int wn_multiband3d_points(const wn_tile *tile, const float *xyz_dev, size_t n, float s,
                          int first_band, int nbands, const float *w_host, float var_per_band,
                          float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (nbands < 0 || nbands > kMaxBands)
        return fail(WN_ERR_INVALID, "nbands must be in 0..%d (got %d)", kMaxBands, nbands);
    if (nbands && !w_host) return fail(WN_ERR_INVALID, "w_host is NULL");
    PointsArgs a{};
    rc = fill_common(tile, 3, xyz_dev, n, out_dev, &a);
    if (rc || n == 0) return rc;
    a.pts = xyz_dev;
    a.out = out_dev;
    int active = 0;
    while (active < nbands && s + (float)first_band + (float)active < 0.0f) ++active;
    float variance = 0.0f;
    for (int b = 0; b < nbands; ++b) variance += w_host[b] * w_host[b];
    a.nbands = active;
    for (int b = 0; b < active; ++b) {
        a.band_scale[b] = ldexpf(1.0f, first_band + b);
        a.band_w[b] = w_host[b];
    }
    a.apply_div = variance != 0.0f;
    a.out_div = a.apply_div ? sqrtf(variance * var_per_band) : 1.0f;
    if (n >= kSortMinPoints && a.n > 0 && a.nbands >= 1 && sort_enabled()) { // long lists: chunks in the finest band's z-plane order
        const int lrc = tile->dev_padded ? launch_sorted(Eval3dOps<true, true>{a, n}, as_stream(stream))
                                         : launch_sorted(Eval3dOps<false, true>{a, n}, as_stream(stream));
        if (lrc) return fail(lrc, "too many points");
        WN_LAUNCH_CHECK("plane_sorted_points_kernel(WMultibandNoise)");
        return WN_OK;
    }
    if (tile->dev_padded)
        hipLaunchKernelGGL(multiband3d_points_kernel<true>, dim3(point_blocks(n)), dim3(256), 0,
                           as_stream(stream), a);
    else
        hipLaunchKernelGGL(multiband3d_points_kernel<false>, dim3(point_blocks(n)), dim3(256), 0,
                           as_stream(stream), a);
    WN_LAUNCH_CHECK("multiband3d_points_kernel");
    return WN_OK;
}

int wn_multiband3d_projected_points(const wn_tile *tile, const float *xyz_dev, const float *normals_dev,
                                    int one_normal, size_t n, float s, int first_band, int nbands,
                                    const float *w_host, float var_per_band, float *out_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (nbands < 0 || nbands > kMaxBands)
        return fail(WN_ERR_INVALID, "nbands must be in 0..%d (got %d)", kMaxBands, nbands);
    if (nbands && !w_host) return fail(WN_ERR_INVALID, "w_host is NULL");
    PointsArgs a{};
    rc = fill_common(tile, 3, xyz_dev, n, out_dev, &a);
    if (rc || n == 0) return rc;
    if (!normals_dev) return fail(WN_ERR_INVALID, "normals_dev is NULL");
    a.coef = tile->dev; // the projected evaluator indexes the linear layout
    a.pts = xyz_dev;
    a.normals = normals_dev;
    a.one_normal = one_normal ? 1 : 0;
    a.out = out_dev;
    int active = 0;
    while (active < nbands && s + (float)first_band + (float)active < 0.0f) ++active;
    float variance = 0.0f;
    for (int b = 0; b < nbands; ++b) variance += w_host[b] * w_host[b];
    a.nbands = active;
    for (int b = 0; b < active; ++b) {
        a.band_scale[b] = ldexpf(1.0f, first_band + b);
        a.band_w[b] = w_host[b];
    }
    a.apply_div = variance != 0.0f;
    a.out_div = a.apply_div ? sqrtf(variance * var_per_band) : 1.0f;
    hipLaunchKernelGGL(multiband3d_projected_points_kernel, dim3(point_blocks(n)), dim3(256), 0, as_stream(stream), a);
    WN_LAUNCH_CHECK("multiband3d_projected_points_kernel");
    return WN_OK;
}

int wn_wavelet_texture_points(const wn_tile *tile, int use_3d, double scale, int octave,
                              const float *xyz_dev, const uint8_t *active_dev, size_t n,
                              float *grey_dev, void *stream)
{
    WN_ENTRY();
    int rc = require_device();
    if (rc) return rc;
    if (n == 0) return WN_OK;
    if (!xyz_dev || !grey_dev) return fail(WN_ERR_INVALID, "points/grey pointer is NULL");
    TexArgs a{};
    const bool has_tile = tile && tile->count != 0;
    if (has_tile && (rc = check_handle_device(tile->device, "tile")) != WN_OK) return rc;
    if (has_tile && tile->dims != (use_3d ? 3 : 2))
        return fail(WN_ERR_INVALID, "tile is %d-D but use_3d=%d", tile->dims, use_3d);
    a.coef = has_tile ? tile->dev : nullptr;
    a.n = has_tile ? tile->n : 0;
    a.nmask = pow2_mask(a.n);
    a.mode = has_tile ? (use_3d ? 3 : 2) : 0;
    a.scale = scale;
    const float octave_scale = (float)std::pow(2.0, (double)octave); // texture.h:77
    a.octave_mul = octave_scale * 2.0f;
    a.inv_stddev = 1.0f / std::sqrt(use_3d ? 0.18402f : 0.19686f);   // texture.h:84,98
    a.pts = xyz_dev;
    a.active = active_dev;
    a.grey = grey_dev;
    a.count = n;
    a.points_per_wave = 1024;
    const size_t waves = (n + a.points_per_wave - 1) / a.points_per_wave;
    const size_t blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffull) return fail(WN_ERR_INVALID, "too many points");
    bool padded = has_tile && use_3d && tile->dev_padded;
#ifdef WN_TUNE_ENV
    if (getenv("WN_POINTS_UNPADDED")) padded = false; // 27 dword gathers per point instead of 9 dwordx3
#endif
    if (padded) a.coef = tile->dev_padded;
    const dim3 grid((unsigned)blocks), block(256);
    // 3-D tile and enough points: chunks taken in z-plane order (see plane_sorted_points_kernel)
    bool sorted = a.mode == 3 && n >= kSortMinPoints;
#ifdef WN_TUNE_ENV
    if (getenv("WN_NO_POINT_SORT")) sorted = false;
#endif
    if (sorted) {
        int lrc;
        if (padded && launch_row_slab(TextureOps<false, true>{a, n}, a.n, active_dev != nullptr, as_stream(stream))) {
            WN_LAUNCH_CHECK("row_slab_points_kernel(texture)");
            return WN_OK;
        }
        if (active_dev) lrc = padded ? launch_sorted(TextureOps<true, true>{a, n}, as_stream(stream)) : launch_sorted(TextureOps<true, false>{a, n}, as_stream(stream));
        else lrc = padded ? launch_sorted(TextureOps<false, true>{a, n}, as_stream(stream)) : launch_sorted(TextureOps<false, false>{a, n}, as_stream(stream));
        if (lrc) return fail(lrc, "too many points");
        WN_LAUNCH_CHECK("plane_sorted_points_kernel(texture)");
        return WN_OK;
    }
    if (active_dev) {
        if (padded) hipLaunchKernelGGL((wavelet_texture_kernel<true, true>), grid, block, 0, as_stream(stream), a);
        else hipLaunchKernelGGL((wavelet_texture_kernel<true, false>), grid, block, 0, as_stream(stream), a);
    } else {
        if (padded) hipLaunchKernelGGL((wavelet_texture_kernel<false, true>), grid, block, 0, as_stream(stream), a);
        else hipLaunchKernelGGL((wavelet_texture_kernel<false, false>), grid, block, 0, as_stream(stream), a);
    }
    WN_LAUNCH_CHECK("wavelet_texture_kernel");
    return WN_OK;
}

} // extern "C"
