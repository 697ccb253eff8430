// wn_wavelet_strip.hip -- "strip march" kernel for dense 3-D wavelet lattices whose rows are a
// multiple of 256 samples: the store-stream-shaped variant of the separable evaluation
//     out[x,y,z] = sum_i Wx[x,i] * ( sum_k Wz[z,k] * ( sum_j Wy[y,j] * C[i,j,k] ) ).
//
// What shapes it (profiles/r01d_store_stream_microbench.txt, DESIGN.md "Store stream"):
//   * the fastest fp32 store stream on an MI355X (6.2-6.5 TB/s, the pattern of hipMemset's own
//     kernel) has waves that each issue ONE 1-KiB store per step, the whole chip writing a
//     contiguous stretch of rows per step and every wave returning to the same slot of the next
//     plane;
//   * a wave that computes AND stores loses the time its store instruction waits for the throttled
//     memory pipeline (in-order issue: compute time + store time), and global loads issued between
//     the stores cost far more than their bytes: they queue behind the stores, and vmcnt retires a
//     wave's loads and stores in issue order, so waiting for a load waits for every older store;
//   * waves that only store, fed through LDS by waves that only compute, with an LDS-only barrier
//     per step (__syncthreads() would drain vmcnt, i.e. wait for the stores), keep the store stream
//     at its floor while the compute side runs beside it.
// So a workgroup is 4 compute waves + 4 store waves, two workgroups per CU.
//   * An item is four consecutive rows y..y+3 of one 256-sample column block (x0..x0+255) times a
//     chunk of <= 128 planes; compute wave w owns row y+w and marches through the planes z.
//   * The four rows share 4 tile rows per coefficient plane.  The compute waves together first read
//     everything the item touches -- <= 38 planes x 4 tile rows x 96 columns, aligned 16-byte loads,
//     8 in flight per lane -- into an LDS table: the march itself issues no global load.  (Reads
//     that compete with a saturated store stream take 5-10 us to return.)  Round 2: only the first 13
//     planes are waited for; the rest is requested at the same time, held in registers and written to
//     the table at plane 8 of the march.
//   * March, one plane per step, lane = coefficient column: the y-collapsed planes of the three z
//     taps live in registers -- at a plane change the wave collapses its three table rows of the
//     plane after next with its fixed wy weights -- R[column] = sum_k wz[k]*Y[k] goes through a
//     per-wave LDS row so that each lane can read the 4-wide window of its 4 x samples, 16 window
//     FMAs, and the 1-KiB output row is parked in LDS.  Software-pipelined: R rows are written two
//     planes ahead, window and z-table reads are issued a step before their use.
//   * The store wave paired with the compute wave moves each parked row to memory.  The barrier of
//     step z sits in the middle of step z+1 and shares a statement with that step's R write, with a
//     counted lgkmcnt: no LDS latency is waited out at the barrier.
//   * The two workgroups of a CU take turns at the higher wave priority, 64 planes at a time, the younger one two
//     turns of three (s_setprio): the arbiter otherwise favours the older workgroup all the way, which then finishes
//     15-20 us early and leaves the CU half empty (worth 5-8 % at 512^3).
// Per-axis mids / weights are computed exactly as the reference does (WaveletNoise.cpp:194-200);
// only the order of the final sums differs (tolerance 1e-5, like the brick kernel).
#include "wn_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#ifdef WN_STRIP_STAMPS
#include <cstdio>
#endif

namespace {

using wn::GridArgs;

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int kCW = 4;          // compute waves per workgroup = rows per item
constexpr int kSW = kCW;        // store waves per workgroup (one per compute wave)
constexpr int kCols = 96;       // coefficient columns an item may touch (host: 255*step + 7 <= 96)
#ifndef WN_STRIP_PLANES
#define WN_STRIP_PLANES 37
#define WN_STRIP_CHUNK 128
#endif
constexpr int kPlanes = WN_STRIP_PLANES;     // coefficient planes an item may touch (host: (chunk_len-1)*step + 5 <= 37)
constexpr int kRowFloats = 96;  // table row = kCols
constexpr int kMaxChunk = WN_STRIP_CHUNK;  // planes per item (z table: 16 B per plane)
constexpr int kRRow = 100;      // one R row (96 columns + pad)
constexpr int kTableFloats = kPlanes * kCW * kRowFloats; // the item's coefficient table [plane][tile row 0..3][96]
constexpr int kStageFloats = 4 * kCW * 256; // two pairs of steps x one 1-KiB output row per compute wave and step
constexpr int kZtabFloats = 4 * (kMaxChunk + 3);
constexpr size_t kLdsBytes = (size_t)(kTableFloats + kStageFloats + kCW * 2 * kRRow + kZtabFloats) * sizeof(float);
static_assert(2 * kLdsBytes <= 160 * 1024, "two workgroups share a CU");

struct StripArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    float inv_den;    // 1/den when den is a power of two (exact), else 0
    int segs_per_row; // nx / 256
    int total_groups; // segs_per_row * ceil(ny / 4): groups of four rows of one column block
    int range_len;    // planes of one owner range: a workgroup takes (group, range) pairs ...
    int owners;       // ... total_groups * number of ranges of them
    int chunk_len;    // and walks a range in items of at most this many planes
    int split_fill;   // request the later planes of an item's table while its march starts
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. make a
// store wave wait for its outstanding global stores.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The first half of the grid (dispatched first, one workgroup per CU) and the second half share the CUs; they
// take turns at the higher wave priority, kPrioPeriod planes at a time.
constexpr int kPrioPeriod = 64;
// The later-dispatched workgroup of the pair has the turn twice out of three times: time stamps of pairs (round 2) showed the
// turns are not symmetric -- with the older workgroup at the higher priority the younger one starves (9 us per 16 planes
// against 4), with the younger one higher the older merely slows (4.2 against 3.2) -- so with equal turns the older still
// finished 4-12 us early.  2 : 1 measured 1-7 % faster than 1 : 1 on four boxes (3 : 1 the same, "always" worse again).
__device__ __forceinline__ void set_turn_priority(int turn)
{
    const bool younger = blockIdx.x >= gridDim.x / 2;
    const bool third = turn % 3 == 2;
    if (younger != third) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

__global__ __launch_bounds__(64 * (kCW + kSW)) void grid3d_strip_kernel(const StripArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const GridArgs &g = a.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS: the item's coefficient table [plane][tile row 0..3][96], the output stage
    // [step parity][compute wave][256], per compute wave two R rows, the z table of the item
    float *const table = lds;
    float *const stage = table + kTableFloats;
    float *const rrows = stage + kStageFloats;
    float *const ztab = rrows + kCW * 2 * kRRow; // per plane of the item {mid_z, wz0, wz1, wz2}, + 3 pad entries
    const size_t plane_stride = (size_t)g.ny * g.nx;
    // Work: owner ranges (group of four rows, range of planes), walked in items.
    // Every wave of the workgroup passes the same barriers.  Per item: one after its set-up, one per pair of
    // planes, one closing it.  Planes 2k, 2k+1: compute wave c parks its rows in stage[2k & 3][c], stage[(2k+1) & 3][c];
    // after the next barrier store wave c moves both to memory while the compute waves are on the next pair.
    int gt = 0; // planes this workgroup has done: the priority turns follow it

    if (wave >= kCW) {
        // ---- store waves: no arithmetic, no loads --------------------------------------------------------
        const int c = wave - kCW;
        for (int owner = blockIdx.x; owner < a.owners; owner += gridDim.x) {
            const int range = owner / a.total_groups, grp = owner - range * a.total_groups;
            const int yg = grp / a.segs_per_row, xs = grp - yg * a.segs_per_row;
            const int z_lo = range * a.range_len, z_hi = min(g.nz, z_lo + a.range_len);
            const int row = yg * kCW + c; // the last group of a lattice with ny % 4 != 0 has rows past the end: computed, not stored
            const float *src = stage + c * 256 + lane * 4;
            for (int zb = z_lo; zb < z_hi;) {
                const int zn = min(a.chunk_len, z_hi - zb);
                float *dst = a.out + ((size_t)zb * plane_stride + (size_t)row * g.nx + xs * 256 + lane * 4);
                lds_barrier(); // set-up
                for (int t = 0; t < zn; t += 2, gt += 2) { // a pair of planes per hand-over
                    if ((gt & (kPrioPeriod - 1)) <= 1) set_turn_priority(gt / kPrioPeriod);
                    lds_barrier();
                    if (row < g.ny) {
                        *reinterpret_cast<v4f *>(dst) = *reinterpret_cast<const v4f *>(src + (t & 3) * (kCW * 256));
                        if (t + 1 < zn)
                            *reinterpret_cast<v4f *>(dst + plane_stride) = *reinterpret_cast<const v4f *>(src + ((t + 1) & 3) * (kCW * 256));
                        dst += 2 * plane_stride;
                    }
                }
                if (zn & 1) --gt; // planes, not pairs
                lds_barrier(); // item closed: stage, tables may be rewritten
                zb += zn;
            }
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
    const float den = (float)g.den;
    const int n = a.n, mask = a.nmask;
    for (int owner = blockIdx.x; owner < a.owners; owner += gridDim.x) {
        const int range = owner / a.total_groups, grp = owner - range * a.total_groups;
        const int yg = grp / a.segs_per_row, xs = grp - yg * a.segs_per_row;
        const int z_lo = range * a.range_len, z_hi = min(g.nz, z_lo + a.range_len);
        const int y = yg * kCW + wave;
        const int x_first = xs * 256, x0 = x_first + lane * 4;
        // ---- x: this lane's four samples -> 16 window weights, window start as a column index ------
        int mx_first;
        {
            float t0, t1, t2;
            wn::bspline(coord(x_first, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), mx_first, t0, t1, t2);
            mx_first = __builtin_amdgcn_readfirstlane(mx_first);
        }
        const int ix0 = (mx_first - 1) & ~3; // coefficient column of table/R column 0, aligned for 16-byte loads
        float ww[4][4];
        int wbase, nquads;
        {
            int m[4];
            float w[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wn::bspline(coord(x0 + q, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m[q], w[q][0], w[q][1], w[q][2]);
            wbase = min(m[0] - 1 - ix0, kCols - 4); // the host guarantees m[0] + 2 - ix0 < kCols
            // column quads the block touches: lane 63 holds the last window
            nquads = min(__builtin_amdgcn_readlane(wbase, 63) / 4 + 2, kCols / 4);
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
        int my_first, my_v;
        float wy0, wy1, wy2;
        {
            float t0, t1, t2;
            wn::bspline(coord(yg * kCW, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), my_first, t0, t1, t2);
        }
        wn::bspline(coord(y, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), my_v, wy0, wy1, wy2);
        my_first = __builtin_amdgcn_readfirstlane(my_first);
        const int d = min(max(__builtin_amdgcn_readfirstlane(my_v) - my_first, 0), 1);

        for (int zb = z_lo; zb < z_hi;) { // the items of the range
            const int zn = min(a.chunk_len, z_hi - zb);
            // ---- z table of the item (the four compute waves together); 3 pad entries repeat the last plane
            for (int i = tid; i < zn + 3; i += 64 * kCW) {
                int m;
                float w0, w1, w2;
                wn::bspline(coord(g.z0 + zb + min(i, zn - 1), den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m, w0, w1, w2);
                *reinterpret_cast<v4f *>(ztab + 4 * i) = v4f{__int_as_float(m), w0 * g.out_scale, w1 * g.out_scale, w2 * g.out_scale};
            }
            int m0;
            {
                float t0, t1, t2;
                wn::bspline(coord(g.z0 + zb, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m0, t0, t1, t2);
                m0 = __builtin_amdgcn_readfirstlane(m0);
            }

            // ---- coefficient table of the item: planes m0-1 .. m_last+1, tile rows my_first-1 .. my_first+2,
            // columns ix0 .. ix0+95, filled by the four compute waves together (16 bytes per lane and load, aligned:
            // no wrap inside a quad)
            int m_last;
            {
                float t0, t1, t2;
                wn::bspline(coord(g.z0 + zb + zn - 1, den, a.inv_den, g.base_range, g.octave_scale, g.post_scale), m_last, t0, t1, t2);
                m_last = __builtin_amdgcn_readfirstlane(m_last);
            }
            const int planes = min(m_last - m0 + 3, kPlanes); // the host guarantees the bound
            // The march needs the first kEarlyPlanes planes of the table before plane ~30 of the item; the rest is requested
            // now, held in registers while the march starts, and written to LDS at plane kCommitAt -- the compute waves issue
            // no stores, so waiting for their own loads does not wait for any store.  The stores of the item start after a
            // third of the fill (round 2; the whole table first: 5.6 us at kernel start with no store in flight on the chip).
            // Measured, 12 alternating pairs of runs on two boxes: 110.9 against 114.2 us.
            constexpr int kEarlyPlanes = 13, kLate = 7, kCommitAt = 8;
            const int total = planes * kCW * nquads; // (plane, tile row, quad) triples
            auto triple_src = [&](int q, int &dst) {
                const int p = q / (kCW * nquads), rq = q - p * (kCW * nquads), row = rq / nquads, quad = rq - row * nquads;
                dst = ((p * kCW + row) * (kCols / 4) + quad) * 4;
                return (size_t)((m0 - 1 + p) & mask) * n * n + (size_t)((my_first - 1 + row) & mask) * n + (size_t)((ix0 + 4 * quad) & mask);
            };
            // late part: at most kLate triples per lane, and only when the item is long enough to reach the commit point
            const int late = (a.split_fill && zn >= 2 * kCommitAt + 2 && planes > kEarlyPlanes)
                                 ? min(total - kEarlyPlanes * kCW * nquads, kLate * 64 * kCW) : 0;
            const int early = total - late;
            {
                constexpr int kBatch = 8;
                for (int q0 = 0; q0 < early; q0 += kBatch * 64 * kCW) {
                    v4f c[kBatch];
                    int dst[kBatch];
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) {
                        const int q = min(q0 + k * 64 * kCW + tid, early - 1); // past the end: repeat the last triple
                        c[k] = *reinterpret_cast<const v4f *>(a.coef + triple_src(q, dst[k]));
                    }
#pragma unroll
                    for (int k = 0; k < kBatch; ++k) *reinterpret_cast<v4f *>(table + dst[k]) = c[k];
                }
            }
            v4f late_c[kLate];
            if (late > 0) {
#pragma unroll
                for (int k = 0; k < kLate; ++k) {
                    int dst;
                    late_c[k] = *reinterpret_cast<const v4f *>(a.coef + triple_src(min(early + k * 64 * kCW + tid, total - 1), dst));
                }
            }
            auto commit_late = [&]() {
#pragma unroll
                for (int k = 0; k < kLate; ++k) {
                    int dst;
                    (void)triple_src(min(early + k * 64 * kCW + tid, total - 1), dst);
                    *reinterpret_cast<v4f *>(table + dst) = late_c[k];
                }
            };
            // y-collapse of a plane: columns lane and 64 + (lane & 31) of this wave's three rows
            const float *const ca = table + d * kRowFloats + lane, *const cb = table + d * kRowFloats + 64 + (lane & 31);
            auto collapse = [&](int kz, float (&yv)[2]) {
                const int s = min(kz - (m0 - 1), planes - 1) * (kCW * kRowFloats); // past the item's planes: values no stored plane uses
                yv[0] = __builtin_fmaf(wy2, ca[s + 2 * kRowFloats], __builtin_fmaf(wy1, ca[s + kRowFloats], wy0 * ca[s]));
                yv[1] = __builtin_fmaf(wy2, cb[s + 2 * kRowFloats], __builtin_fmaf(wy1, cb[s + kRowFloats], wy0 * cb[s]));
            };
            lds_barrier(); // set-up: z table and coefficient table written
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
                    collapse(m + 2, Y3);
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
            // Two inline-asm hazards that each cost a GPU fault while this kernel was developed -- keep them out:
            //  (1) never tie a register that a still-in-flight load writes to a wait statement with "+v"
            //      (asm volatile("s_waitcnt vmcnt(N)" : "+v"(loaded))): hipcc may copy the register BEFORE the
            //      statement, i.e. before the data has landed.  Operands of the statement below are inputs only
            //      ("v"), produced by VALU, never by a pending memory operation;
            //  (2) a VMEM instruction inside asm that reads an SGPR written by v_readfirstlane just before needs
            //      its own "s_nop 4" (the compiler does not see the hazard through the asm).  This kernel has no
            //      VMEM instruction in asm; its only asm statements are LDS writes, waits and barriers.
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
            auto step = [&](int z, const float (&cur)[4], float (&nxt)[4], const v4f &e_use, v4f &e_load, int rd, bool handover) {
                const float *r = rb0 + rd * kRRow + wbase;
                nxt[0] = r[0]; nxt[1] = r[1]; nxt[2] = r[2]; nxt[3] = r[3];
                e_load = zt[z + 3];
                if (handover && z != 0) write_r_handover(rd ^ 1, e_use); // the two rows parked by the previous trip
                else write_r(rd ^ 1, e_use);
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = ww[q][0] * cur[0];
                    t = __builtin_fmaf(ww[q][1], cur[1], t);
                    t = __builtin_fmaf(ww[q][2], cur[2], t);
                    o[q] = __builtin_fmaf(ww[q][3], cur[3], t);
                }
                *reinterpret_cast<v4f *>(park + (z & 3) * (kCW * 256)) = v4f{o[0], o[1], o[2], o[3]};
            };
            float wa[4], wb[4];
            v4f ea = zt[2], eb;
            {
                const float *r = rb0 + wbase;
                wa[0] = r[0]; wa[1] = r[1]; wa[2] = r[2]; wa[3] = r[3];
            }
            int z = 0;
            for (; z + 1 < zn; z += 2) {
                if (z == kCommitAt && late > 0) commit_late(); // visible to the other waves after the next hand-over barriers
                if (((gt + z) & (kPrioPeriod - 1)) <= 1) set_turn_priority((gt + z) / kPrioPeriod); // two planes per trip
                step(z, wa, wb, ea, eb, 1, true);
                step(z + 1, wb, wa, eb, ea, 0, false);
            }
            if (z < zn) step(z, wa, wb, ea, eb, 1, true);
            lds_barrier(); // hands over the last pair of rows
            stamp();
            lds_barrier(); // item closed: stage, tables may be rewritten
            gt += zn;
            zb += zn;
        }
    }
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

} // namespace

namespace wn {

// Launches the strip-march kernel when the lattice is in its regime; *launched tells the caller.
int strip_try(const wn_tile *tile, const GridArgs &g, float *out_dev, hipStream_t stream, bool *launched)
{
    *launched = false;
#ifdef WN_TUNE_ENV
    if (getenv("WN_NO_STRIP")) return WN_OK;
#endif
    if (tile->n < 4 || pow2_mask(tile->n) < 0) return WN_OK;
    if (g.z_const_mode || g.nx <= 0 || g.ny <= 0 || g.nz <= 0) return WN_OK;
    if (g.z0 < 0) return WN_OK; // negative plane indices: the exact kernel (the bounds below assume indices >= 0)
    if (g.nx % 256 != 0 || (reinterpret_cast<uintptr_t>(out_dev) & 15) != 0) return WN_OK;
    const double step = (double)g.base_range * (double)g.octave_scale * (double)g.post_scale / g.den;
    if (!(step >= 0.0) || !std::isfinite(step)) return WN_OK;
    const double imax = std::max(std::max((double)g.nx, (double)g.ny), (double)g.z0 + g.nz);
    const double pmax = step * imax + 1.0;
    if (pmax > 1.0e6) return WN_OK;
    const double slack = pmax * 4.8e-7; // fp32 rounding of a coordinate, in planes
    // 4 consecutive samples (x quad of a lane, y rows of an item) span <= 2 mids, and a plane change
    // advances the mid by exactly 1
    if (step < 0.18) return WN_OK; // finer lattices: the brick kernel is faster
    if (3.0 * step + slack > 1.0) return WN_OK;
    if (255.0 * step + slack + 7.0 > (double)kCols) return WN_OK; // columns of a block (+3 of alignment)
    const long long groups = (long long)(g.nx / 256) * ((g.ny + kCW - 1) / kCW);
    if (groups > 0x3fffffffLL) return WN_OK;

    const int dev = current_device();
    const int cus = device_compute_units(dev);
    // items: the planes an item touches must fit its LDS table
    int chunk_max = kMaxChunk;
    if (step > 0.0) chunk_max = (int)std::min<double>(kMaxChunk, std::floor((kPlanes - 5 - slack) / step) + 1.0);
    if (chunk_max < 8) return WN_OK;
    // owner ranges: every workgroup slot of the chip (two per CU) should get one, as long as a range keeps >= 32 planes
    long long wgs = 2LL * cus;
    int nranges = 1;
    while (groups * nranges < wgs && (g.nz + 2 * nranges - 1) / (2 * nranges) >= 32) nranges *= 2;
#ifdef WN_TUNE_ENV
    if (const char *e = getenv("WN_STRIP_WGS")) wgs = (long long)atoi(e) * cus;
    if (const char *e = getenv("WN_STRIP_RANGES")) nranges = atoi(e);
#endif
    StripArgs a{};
    a.coef = tile->dev;
    a.out = out_dev;
    a.n = tile->n;
    a.nmask = pow2_mask(tile->n);
    a.g = g;
    a.inv_den = ((g.den & (g.den - 1)) == 0) ? 1.0f / (float)g.den : 0.0f;
    a.segs_per_row = g.nx / 256;
    a.total_groups = (int)groups;
    a.range_len = (g.nz + nranges - 1) / nranges;
    const long long owners = groups * ((g.nz + a.range_len - 1) / a.range_len);
    if (owners > 0x3fffffffLL) return WN_OK;
    a.owners = (int)owners;
    // a range is walked in equal items of at most chunk_max planes
    const int per_range = (a.range_len + chunk_max - 1) / chunk_max;
    a.chunk_len = (a.range_len + per_range - 1) / per_range;
    a.split_fill = 1;
#ifdef WN_TUNE_ENV
    if (getenv("WN_STRIP_NO_SPLIT")) a.split_fill = 0;
#endif
    // dynamic LDS beyond 64 KiB needs a per-(kernel, device) opt-in; refused -> the brick kernel serves the lattice
    if (!ensure_dynamic_lds(reinterpret_cast<const void *>(&grid3d_strip_kernel), dev, kLdsBytes)) return WN_OK;
    const int blocks = (int)std::min<long long>(owners, wgs);
#ifdef WN_STRIP_STAMPS
    static unsigned long long *dbg = nullptr;
    static int dbg_calls = 0;
    if (!dbg) { (void)hipMalloc(&dbg, 16 * 16 * 8); (void)hipMemset(dbg, 0, 16 * 16 * 8); }
    a.stamps = dbg;
#endif
    hipLaunchKernelGGL(grid3d_strip_kernel, dim3(blocks), dim3(64 * (kCW + kSW)), kLdsBytes, stream, a);
    WN_LAUNCH_CHECK("grid3d_strip_kernel");
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
