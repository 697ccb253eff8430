// experimental/wn_wavelet_rowgroup.hip -- NOT part of libwnoise_hip.so (the Makefile does not build it).
// Round-1 prototype kept for the next round: the "row group" variant of the strip-march kernel
// (../wn_wavelet_strip.hip).  Same march, same compute-wave / store-wave split, but
//   * a workgroup is 4 compute + 4 store waves and owns four consecutive rows y..y+3 of one 256-sample
//     column block; the four rows share 4 tile rows per coefficient plane;
//   * no per-item coefficient table: wave w streams tile row w of the plane kDist = 11 plane changes ahead
//     with ONE LDS-DMA instruction (global_load_lds_dwordx4, EXEC narrowed to 24 lanes, M0 = destination)
//     into a ring of 16 planes, retired with a counted vmcnt; every wave y-collapses its three rows on
//     the fly.  Reads under a saturated store stream take 5-10 us to return: with a ring of 8 planes
//     (3 ahead) the march stalled on them (365-425 ns per plane).
// Measured on MI355X (parity: all tests of tests/test_gpu_parity.py pass when it is hooked in before
// strip_try): 512^3 114-120 us (strip 112-118, brick 121-131); 1024^3 777 us with two workgroups per CU
// (brick 833, strip 843), 876 us with three; 2048x2048x256 848-924 us (brick 755).
// To try it again: add the file to SRCS, declare rowgroup_try in wn_internal.hpp and call it in
// wn_eval3d_grid ahead of strip_try.
#include "../wn_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

using wn::GridArgs;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kCW = 4;          // compute waves per workgroup = rows per item
constexpr int kSW = kCW;        // store waves per workgroup (one per compute wave)
constexpr int kCols = 96;       // coefficient columns an item may touch (host: 255*step + 7 <= 96)
constexpr int kRing = 16;       // coefficient planes in the LDS ring
constexpr int kDist = 11;       // a plane is fetched kDist plane changes before the change that waits for it
constexpr int kRowFloats = 96;  // ring row: one DMA instruction, 24 lanes x 16 bytes
constexpr int kMaxChunk = 1024; // planes per item (z table: 16 B per plane)
constexpr int kRRow = 100;      // one R row (96 columns + pad)
constexpr int kRingFloats = kRing * kCW * kRowFloats;
constexpr int kStageFloats = 2 * kCW * 256; // two steps x one 1-KiB output row per compute wave
constexpr int kZtabFloats = 4 * (kMaxChunk + 3);
constexpr size_t kLdsBytes = (size_t)(kRingFloats + kStageFloats + kCW * 2 * kRRow + kZtabFloats) * sizeof(float);
static_assert(3 * kLdsBytes <= 160 * 1024, "three workgroups share a CU");
static_assert(kRingFloats * 4 < 65536, "DMA destinations stay in the first 64 KiB of LDS");
static_assert(kDist + 5 == kRing, "the set-up fills the whole ring: planes m0-1 .. m0+3+kDist");

struct RowGroupArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    float inv_den;    // 1/den when den is a power of two (exact), else 0
    int segs_per_row; // nx / 256
    int total_groups; // segs_per_row * ny / 4: groups of four rows of one column block
    int chunk_len;    // planes per item
    int total_items;  // total_groups * number of z chunks
#ifdef WN_STRIP_STAMPS
    unsigned long long *stamps; // debug build: phase time stamps of a few workgroups
#endif
};

__device__ __forceinline__ float coord(int i, float den, float inv_den, float range, float oscale, float post)
{
    const float fi = (float)i;
    float c = ((inv_den != 0.0f) ? fi * inv_den : fi / den) * range;
    c = c * oscale;
    c = c * post;
    return c;
}

// LDS byte address of a pointer into the dynamic shared array
__device__ __forceinline__ unsigned lds_address(const float *p)
{
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char *)(const char *)p;
}

// One LDS-DMA load by the first 24 lanes: lane l reads 16 bytes at plane + off and the hardware writes them
// to LDS[dst + 16*l].  EXEC is narrowed to the 24 lanes and M0 (the LDS destination) set inside the statement,
// both restored before it ends.  s_nop 4: `plane` may come straight from v_readfirstlane (VALU-writes-SGPR ->
// VMEM-reads-SGPR hazard, which hipcc does not pad inside asm); it also covers the M0 and EXEC writes.
__device__ __forceinline__ void dma_row16(unsigned dst_byte, unsigned off, const float *plane)
{
    unsigned keep_m0;
    unsigned long long keep_exec;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_mov_b64 exec, 0xffffff\n\ts_nop 4\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep_m0), "=&s"(keep_exec)
                 : "v"(off), "s"(plane), "s"(dst_byte)
                 : "memory");
}

// a pointer the compiler must treat as wave-uniform (SGPR pair): the DMA uses it as scalar base
__device__ __forceinline__ const float *uniform_ptr(const float *p)
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
}

// Wait until at most N vector-memory instructions of this wave are outstanding (they retire in issue order).
template <int N>
__device__ __forceinline__ void retire_fixed()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. make a
// store wave wait for its outstanding global stores.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(64 * (kCW + kSW)) void grid3d_rowgroup_kernel(const RowGroupArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GridArgs &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS: the coefficient ring [slot][tile row 0..3][256] first (DMA destinations travel in M0), the
    // output stage [step parity][compute wave][256], per compute wave two R rows, the z table of the item
    float *const ring = lds;
    float *const stage = ring + kRingFloats;
    float *const rrows = stage + kStageFloats;
    float *const ztab = rrows + kCW * 2 * kRRow; // per plane of the item {mid_z, wz0, wz1, wz2}, + 3 pad entries
    const size_t plane_stride = (size_t)g.ny * g.nx;
    // Every wave of the workgroup passes the same barriers.  Per round (= item): one after the item's
    // set-up, one per plane of a full chunk, one closing the round.  Plane t: compute wave c parks its row
    // in stage[t&1][c]; after the barrier store wave c moves it to memory while the compute waves are on t+1.
    const int rounds = (a.total_items + gridDim.x - 1) / gridDim.x;

    if (wave >= kCW) {
        // ---- store waves: no arithmetic, no loads --------------------------------------------------------
        const int c = wave - kCW;
        for (int round = 0; round < rounds; ++round) {
            const int item = round * gridDim.x + blockIdx.x;
            const int chunk = item / a.total_groups, grp = item - chunk * a.total_groups;
            const int yg = grp / a.segs_per_row, xs = grp - yg * a.segs_per_row;
            const int zb = chunk * a.chunk_len;
            const int zn = item < a.total_items ? min(a.chunk_len, g.nz - zb) : 0;
            float *dst = a.out + ((size_t)zb * plane_stride + (size_t)(yg * kCW + c) * g.nx + xs * 256 + lane * 4);
            const float *src = stage + c * 256 + lane * 4;
            lds_barrier(); // set-up
            for (int t = 0; t < a.chunk_len; ++t) {
                lds_barrier();
                if (t < zn) {
                    *reinterpret_cast<v4f *>(dst) = *reinterpret_cast<const v4f *>(src + (t & 1) * (kCW * 256));
                    dst += plane_stride;
                }
            }
            lds_barrier(); // round closed: stage and z table may be rewritten
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------
#ifdef WN_STRIP_STAMPS
    int sidx = 0;
    auto stamp = [&]() { if (a.stamps && tid == 0 && (blockIdx.x % 65) == 0 && sidx < 16) a.stamps[(blockIdx.x / 65) * 16 + sidx] = wall_clock64(); ++sidx; };
#else
    auto stamp = [] {};
#endif
    stamp();
    float *const rb0 = rrows + wave * 2 * kRRow;
    const v4f *const zt = reinterpret_cast<const v4f *>(ztab);
    const unsigned ring_byte = __builtin_amdgcn_readfirstlane(lds_address(ring));
    const float den = (float)g.den;
    const int n = a.n, mask = a.nmask;
    for (int round = 0; round < rounds; ++round) {
        const int item = round * gridDim.x + blockIdx.x;
        if (item >= a.total_items) { // same barriers, no work
            for (int t = 0; t <= a.chunk_len + 1; ++t) lds_barrier();
            continue;
        }
        const int chunk = item / a.total_groups, grp = item - chunk * a.total_groups;
        const int yg = grp / a.segs_per_row, xs = grp - yg * a.segs_per_row;
        const int zb = chunk * a.chunk_len, zn = min(a.chunk_len, g.nz - zb);
        const int y = yg * kCW + wave;
        const int x_first = xs * 256, x0 = x_first + lane * 4;

        // ---- z table of the item (the four compute waves together); 3 pad entries repeat the last plane
        for (int i = tid; i < zn + 3; i += 64 * kCW) {
            int m;
            float w0, w1, w2;
            wn::bspline(coord(g.z0 + zb + min(i, zn - 1), den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m, w0, w1, w2);
            *reinterpret_cast<v4f *>(ztab + 4 * i) = v4f{__int_as_float(m), w0 * g.out_scale, w1 * g.out_scale, w2 * g.out_scale};
        }
        // ---- x: this lane's four samples -> 16 window weights, window start as a column index ------
        int mx_first;
        {
            float t0, t1, t2;
            wn::bspline(coord(x_first, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), mx_first, t0, t1, t2);
            mx_first = __builtin_amdgcn_readfirstlane(mx_first);
        }
        const int ix0 = (mx_first - 1) & ~3; // coefficient column of ring/R column 0, aligned for 16-byte loads
        float ww[4][4];
        int wbase;
        {
            int m[4];
            float w[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wn::bspline(coord(x0 + q, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m[q], w[q][0], w[q][1], w[q][2]);
            wbase = min(m[0] - 1 - ix0, kCols - 4); // the host guarantees m[0] + 2 - ix0 < kCols
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool up = m[q] != m[0]; // mid is m[0] or m[0]+1 (host guarantees step <= 1/3)
                ww[q][0] = up ? 0.0f : w[q][0];
                ww[q][1] = up ? w[q][0] : w[q][1];
                ww[q][2] = up ? w[q][1] : w[q][2];
                ww[q][3] = up ? w[q][2] : 0.0f;
            }
        }
        // ---- y: the item's rows y0..y0+3 have mids my_first or my_first+1 (host: 3 steps span < 1), so the
        // four tile rows my_first-1 .. my_first+2 serve all of them; this wave's three start at row d.
        int my_first, my_v, m0;
        float wy0, wy1, wy2;
        {
            float t0, t1, t2;
            wn::bspline(coord(yg * kCW, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), my_first, t0, t1, t2);
            wn::bspline(coord(g.z0 + zb, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m0, t0, t1, t2);
        }
        wn::bspline(coord(y, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), my_v, wy0, wy1, wy2);
        my_first = __builtin_amdgcn_readfirstlane(my_first);
        m0 = __builtin_amdgcn_readfirstlane(m0);
        const int d = min(max(__builtin_amdgcn_readfirstlane(my_v) - my_first, 0), 1);

        // ---- coefficient ring: this wave fetches tile row my_first-1+wave of every plane ------------------
        // lane q < 24 reads columns ix0+4q .. ix0+4q+3 (aligned: no wrap inside a quad); only the first 24
        // lanes take part in the DMA
        const unsigned dma_off = ((unsigned)((my_first - 1 + wave) & mask) * (unsigned)n + (unsigned)((ix0 + 4 * min(lane, kCols / 4 - 1)) & mask)) * 4u;
        auto issue_plane = [&](int kz) {
            const float *plane = uniform_ptr(a.coef + (size_t)(kz & mask) * n * n);
            dma_row16(ring_byte + (unsigned)((kz & (kRing - 1)) * kCW + wave) * (kRowFloats * 4), dma_off, plane);
        };
        // y-collapse of a landed plane: columns lane and 64 + (lane & 31) of this wave's three rows
        const float *const ca = ring + d * kRowFloats + lane, *const cb = ring + d * kRowFloats + 64 + (lane & 31);
        auto collapse = [&](int kz, float (&yv)[2]) {
            const int s = (kz & (kRing - 1)) * (kCW * kRowFloats);
            yv[0] = __builtin_fmaf(wy2, ca[s + 2 * kRowFloats], __builtin_fmaf(wy1, ca[s + kRowFloats], wy0 * ca[s]));
            yv[1] = __builtin_fmaf(wy2, cb[s + 2 * kRowFloats], __builtin_fmaf(wy1, cb[s + kRowFloats], wy0 * cb[s]));
        };
        stamp();
        for (int k = -1; k <= 3 + kDist; ++k) issue_plane(m0 + k); // kRing planes
        retire_fixed<kDist>();       // planes up to m0+3 landed
        stamp();
        lds_barrier(); // set-up: z table written, every wave's ring rows landed
        int cur_mid = m0;
        float Y0[2], Y1[2], Y2[2], Y3[2]; // planes cur_mid-1 .. cur_mid+2 (Y3: the prefetched next one)
        collapse(m0 - 1, Y0);
        collapse(m0, Y1);
        collapse(m0 + 1, Y2);
        collapse(m0 + 2, Y3);

        // R values (columns lane, 64 + (lane & 31)) of a plane; `e` = its table entry {mid, wz0, wz1, wz2}
        auto r_values = [&](const v4f e, float &ra, float &rb) {
            const int m = __builtin_amdgcn_readfirstlane(__float_as_int(e.x));
            if (__builtin_expect(m != cur_mid, 0)) { // entered the next coefficient plane (mids advance by exactly 1)
                cur_mid = m;
                Y0[0] = Y1[0]; Y0[1] = Y1[1];
                Y1[0] = Y2[0]; Y1[1] = Y2[1];
                Y2[0] = Y3[0]; Y2[1] = Y3[1];
                // plane m+2: every wave's row landed before an earlier barrier
                collapse(m + 2, Y3);
                // this wave's row of plane m+3 (the next change collapses it) must land before the next barrier;
                // the rows of m+4 .. m+2+kDist stay in flight; then fetch plane m+3+kDist
                retire_fixed<kDist - 1>();
                issue_plane(m + 3 + kDist);
            }
            ra = __builtin_fmaf(e.w, Y2[0], __builtin_fmaf(e.z, Y1[0], e.y * Y0[0]));
            rb = __builtin_fmaf(e.w, Y2[1], __builtin_fmaf(e.z, Y1[1], e.y * Y0[1]));
        };
        float *const r_a = rb0 + lane, *const r_b = rb0 + 64 + (lane & 31); // + kRRow for the other buffer
        auto write_r = [&](int buf, const v4f e) {
            float ra, rb;
            r_values(e, ra, rb);
            r_a[buf * kRRow] = ra;
            r_b[buf * kRRow] = rb;
        };
        // The same, and the workgroup barrier that hands over the output row parked a step ago, in one
        // statement: the counted wait leaves only these R writes outstanding (LDS operations of a wave
        // complete in issue order), so the barrier does not wait out an LDS write latency.  hipcc does not
        // count the hidden writes; unknown operations can only make its own waits stricter.
        auto write_r_handover = [&](int buf, const v4f e) {
            float ra, rb;
            r_values(e, ra, rb);
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %2, %3\n\ts_waitcnt lgkmcnt(2)\n\ts_barrier"
                         :: "v"(lds_address(r_a + buf * kRRow)), "v"(ra), "v"(lds_address(r_b + buf * kRRow)), "v"(rb)
                         : "memory");
        };

        stamp();
        write_r(0, zt[0]);
        write_r(1, zt[1]);
        float *const park = stage + wave * 256 + lane * 4;
        // One plane.  Issued first: the reads for LATER steps -- the window of R(z+1), written a step
        // ago into buffer `rd`, and the table entry of plane z+3.  Then R(z+2) from `e_use` into the
        // other buffer (with the hand-over of row z-1), then this plane's 16 window FMAs on `cur` (read a
        // step ago) and the parking of its row.  The two register sets alternate between the two halves
        // of the unrolled loop: no value is waited for in the step that requested it.
        auto step = [&](int z, const float (&cur)[4], float (&nxt)[4], const v4f &e_use, v4f &e_load, int rd) {
            const float *r = rb0 + rd * kRRow + wbase;
            nxt[0] = r[0]; nxt[1] = r[1]; nxt[2] = r[2]; nxt[3] = r[3];
            e_load = zt[z + 3];
            if (z != 0) write_r_handover(rd ^ 1, e_use);
            else write_r(rd ^ 1, e_use);
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = ww[q][0] * cur[0];
                t = __builtin_fmaf(ww[q][1], cur[1], t);
                t = __builtin_fmaf(ww[q][2], cur[2], t);
                o[q] = __builtin_fmaf(ww[q][3], cur[3], t);
            }
            *reinterpret_cast<v4f *>(park + (z & 1) * (kCW * 256)) = v4f{o[0], o[1], o[2], o[3]};
        };
        float wa[4], wb[4];
        v4f ea = zt[2], eb;
        {
            const float *r = rb0 + wbase;
            wa[0] = r[0]; wa[1] = r[1]; wa[2] = r[2]; wa[3] = r[3];
        }
        int z = 0;
        for (; z + 1 < zn; z += 2) {
            step(z, wa, wb, ea, eb, 1);
            step(z + 1, wb, wa, eb, ea, 0);
        }
        if (z < zn) { step(z, wa, wb, ea, eb, 1); ++z; }
        lds_barrier(); // hands over the last row
        stamp();
        for (; z <= a.chunk_len; ++z) lds_barrier(); // a short last chunk, and the barrier closing the round
        retire_fixed<0>(); // DMA of planes past the chunk: landed before the next item reuses the ring
    }
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

} // namespace

namespace wn {

// Launches the strip-march kernel when the lattice is in its regime; *launched tells the caller.
int rowgroup_try(const wn_tile *tile, const GridArgs &g, float *out_dev, hipStream_t stream, bool *launched)
{
    *launched = false;
    if (tile->n < 4 || pow2_mask(tile->n) < 0) return WN_OK;
    if (g.z_const_mode || g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return WN_OK;
    if (g.nx % 256 != 0 || g.ny % kCW != 0 || (reinterpret_cast<uintptr_t>(out_dev) & 15) != 0) return WN_OK;
    const double step = (double)g.base_range * (double)g.octave_scale * (double)g.post_scale / g.den;
    if (!(step >= 0.0) || !std::isfinite(step)) return WN_OK;
    const double imax = std::max(std::max((double)g.nx, (double)g.ny), (double)g.z0 + g.nz);
    const double pmax = step * imax + 1.0;
    if (pmax > 1.0e6) return WN_OK;
    const double slack = pmax * 4.8e-7; // fp32 rounding of a coordinate, in planes
    // 4 consecutive samples (x quad of a lane, y rows of an item) span <= 2 mids, and a plane change
    // advances the mid by exactly 1
    if (3.0 * step + slack > 1.0) return WN_OK;
    if (255.0 * step + slack + 7.0 > (double)kCols) return WN_OK; // columns of a block (+3 of alignment)
    const long long groups = (long long)(g.nx / 256) * (g.ny / kCW);
    if (groups > 0x3fffffffLL) return WN_OK;

    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    // planes per item: the z table must fit, and every workgroup slot of the chip (two per CU) should get an item
    int nchunks = (g.nz + kMaxChunk - 1) / kMaxChunk;
    const long long wgs = (getenv("WN_RG_WGS") ? atoi(getenv("WN_RG_WGS")) : 3) * (long long)cus;
    while (groups * nchunks < wgs && (g.nz + nchunks) / (nchunks + 1) >= 32) ++nchunks;
    RowGroupArgs a{};
    a.coef = tile->dev;
    a.out = out_dev;
    a.n = tile->n;
    a.nmask = pow2_mask(tile->n);
    a.g = g;
    a.inv_den = ((g.den & (g.den - 1)) == 0) ? 1.0f / (float)g.den : 0.0f;
    a.segs_per_row = g.nx / 256;
    a.total_groups = (int)groups;
    a.chunk_len = (g.nz + nchunks - 1) / nchunks;
    const long long items = groups * ((g.nz + a.chunk_len - 1) / a.chunk_len);
    if (items > 0x3fffffffLL) return WN_OK;
    a.total_items = (int)items;
    static int big_lds_device = -1; // dynamic LDS beyond 64 KiB needs a per-device opt-in for this kernel
    if (big_lds_device != dev) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&grid3d_rowgroup_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
        big_lds_device = dev;
    }
    const int blocks = (int)std::min<long long>(items, wgs);
#ifdef WN_STRIP_STAMPS
    static unsigned long long *dbg = nullptr;
    static int dbg_calls = 0;
    if (!dbg) { (void)hipMalloc(&dbg, 16 * 16 * 8); (void)hipMemset(dbg, 0, 16 * 16 * 8); }
    a.stamps = dbg;
#endif
    hipLaunchKernelGGL(grid3d_rowgroup_kernel, dim3(blocks), dim3(64 * (kCW + kSW)), kLdsBytes, stream, a);
    WN_LAUNCH_CHECK("grid3d_rowgroup_kernel");
#ifdef WN_STRIP_STAMPS
    if (++dbg_calls == 5) {
        unsigned long long h[256];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
        for (int b = 0; b < 8; ++b) {
            fprintf(stderr, "stamps wg %d:", b * 65);
            for (int i = 1; i < 8; ++i) fprintf(stderr, " %.2f", (double)(h[b * 16 + i] - h[b * 16]) / 100.0);
            fprintf(stderr, " us\n");
        }
    }
#endif
    *launched = true;
    return WN_OK;
}

} // namespace wn
