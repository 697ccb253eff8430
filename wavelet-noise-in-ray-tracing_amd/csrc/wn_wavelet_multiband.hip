// wn_wavelet_multiband.hip -- WMultibandNoise on a dense lattice for gfx950 (K5 of SURVEY.md 8): the plane pipeline.
//
// Cook & DeRose's WMultibandNoise (paper Appendix 2; absent from the reference, composed from its
// WaveletNoise::evaluate3D, WaveletNoise.cpp:185-215) sums NB consecutive octaves of the same tile.  On an
// axis-aligned lattice every band's 27-tap sum factors per axis (see wn_wavelet_grid.hip),
//     out[x,y,z] = sum_b f_b * sum_i Wx_b[x,i] * ( sum_k Wz_b[z,k] * ( sum_j Wy_b[y,j] * C[i,j,k] ) ),
// and per sample only the x contraction is left: 4 FMAs per band from a 4-wide window of collapsed rows R_b.
//
// Round 2's brick kernel (grid3d_sep_kernel<5, 2>) ran its phases strictly one after the other inside a CU -- box
// loads, y/z collapse of 64 rows, x windows + stores, two full barriers per brick, 2 waves per SIMD -- and took the
// sum of VALU + LDS + store time per brick (13.7 us for 128 KiB of output; the store floor is 5.9).  This kernel
// keeps the brick (512 x 8 x 8 samples, persistent workgroups, one per CU) and splits a 16-wave workgroup by ROLE
// around ONE barrier per z-plane; each role runs its own loop, so each role's registers are allocated on their own:
//
//  * 4 collapse waves.  Each owns two "passes" = 64 adjacent coefficient columns of one band's box.  A lane holds its
//    column of the brick in REGISTERS, already collapsed in y for the brick's 8 rows of samples (8 x K values, K = 4 or
//    5 box rows in z; once per brick, weights zero-padded to the box and wave-uniform: v_readlane -> SGPR operands):
//    no LDS reads in the per-plane collapse.  Per plane a pass collapses z (8 rows x K FMAs, K uniform weights) and
//    writes 8 row pieces of R for the NEXT plane (R is double-buffered per plane, 16 KiB a slot -- not 68 KiB a brick).
//  * 8 window waves.  Wave (x half, row pair) contracts x for two 2-KiB output rows of the CURRENT plane (per band
//    and row 2 ds_read2_b32 + 8 v_pk_fma_f32, its 80 window weights in registers) and parks them in a 4-plane output
//    ring in LDS.  They never touch memory.
//  * 4 store waves move the ring to memory two planes behind: 4 x (ds_read_b128, global_store_dwordx4) per plane and
//    wave, every wave store 1 KiB contiguous, whole 2-KiB rows.  They also request the NEXT brick's coefficient boxes:
//    tile -> LDS by LDS-DMA (global_load_lds_dwordx4, one per (k, j) box row, the row address scalar), a few rows per
//    plane over the brick's first four planes, landed before its last.  Every vector-memory instruction of the CU is
//    theirs: the waves that compute never wait in the CU's memory-instruction queue behind the stores.
//    The DMA's destination is a separate static __shared__ array so that the compiler's alias analysis does not put a
//    vmcnt(0) in front of every other LDS access; the barrier is s_waitcnt lgkmcnt(0) + s_barrier (no vmcnt drain).
//
// Numerics: per-axis mids / weights exactly as the reference computes them; the order of the sums differs
// (y, then z, then x; FMAs) -> within 1e-5 abs of the composition of evaluate3D calls (measured ~1e-6).
#include "wn_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace {

using wn::GridArgs;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int kMaxNB = 5;                  // bands this kernel is instantiated for (more: the brick kernel)
constexpr int kWW = 8, kPW = 4, kSW = 4;   // window / collapse / store waves per workgroup
constexpr int kWaves = kWW + kPW + kSW;
constexpr int kPasses = 2 * kPW;           // a collapse wave owns two passes
constexpr int kBX = 512, kBY = 8, kBZ = 8; // samples per brick
constexpr int kRing = 4;                   // output planes parked in LDS
constexpr int kSlotZ = 8, kSlotX = 16;     // table slots: 0..7 y samples, 8..15 z samples, 16 / 17 first / last x sample
constexpr int kMaxK = 5;                   // (y, z) extent of a band's coefficient box (8 samples, step < 2/7)
constexpr int kRRow = kPasses * 64;        // R: [plane slot][row of samples][pass][64 columns]
constexpr int kRPlane = kBY * kRRow;       // floats per R plane slot (16 KiB)
constexpr int kRingFloats = kRing * kBY * kBX;
constexpr int kBoxFloats = 6144;           // LDS-DMA landing zone (24 KiB): the boxes of all bands of one brick
constexpr int kDmaPlanes = kMaxNB;         // the next brick's box rows are requested band by band over the first planes ...
constexpr int kDmaLanded = kBZ - 2;        // ... and are complete at the end of this plane's iteration

struct MbBand {
    float oscale;   // octave_scale of this band
    float factor;   // everything that scales this band's contribution: w[b] * out_scale / out_div
    int first_pass; // the band's columns of an R row start at 64 * first_pass
    int K;          // (y, z) extent of the band's coefficient box: 4 or 5 rows
    int box_off;    // float offset of the band's box image in the LDS-DMA zone: [K*K rows][rowlen]
    int rowlen;     // floats per box row image (a multiple of 4: whole 16-byte chunks)
};

struct MbArgs {
    const float *coef;
    float *out;
    int n, nmask;
    GridArgs g;
    float inv_den; // 1/den when den is a power of two (exact), else 0
    int nbx, nby, nbz;
    int all_full;  // every brick lies wholly inside the lattice: no store of a plane is ever skipped
    int debug;     // WN_TUNE_ENV builds: WN_MBP_DEBUG probes (0 in the product)
    int permute, even_permille; // WN_TUNE_ENV builds: work-distribution experiments (0 / 500 in the product)
    MbBand band[kMaxNB];
    int pass_band[kPasses]; // -1: no such pass; passes are dealt in band order, a band's passes are consecutive
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// v_readlane_b32 of a float (the builtin is typed int: pass the bits, not the value)
__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

__device__ __forceinline__ float coord_of(int i, float den, float inv_den, float range, float oscale, float post)
{
    const float fi = (float)i;
    float c = ((inv_den != 0.0f) ? fi * inv_den : fi / den) * range; // exact either way when den is a power of two
    c = c * oscale;
    c = c * post;
    return c;
}

#ifdef WN_TUNE_ENV
// in-kernel time stamps of workgroup 0 (WN_MBP_DEBUG=9): [wave slot 0..2 = a window, a collapse, a store wave][iteration][3]
__device__ long long g_mbp_stamps[3 * 256 * 3];
#define MBP_STAMP(k)                                                                                                \
    do {                                                                                                            \
        if (a.debug == 9 && blockIdx.x == 0 && lane == 0 && gp < 256 && (wave == 1 || wave == kWW + 1 || wave == kWW + kPW)) \
            g_mbp_stamps[((wave == 1 ? 0 : wave == kWW + 1 ? 1 : 2) * 256 + gp) * 3 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define MBP_STAMP(k) do { } while (0)
#endif

template <int NB>
__global__ __launch_bounds__(64 * kWaves) void grid3d_mbp_kernel(const MbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];       // R: 2 plane slots; then the output ring
    __shared__ __attribute__((aligned(16))) float s_box[kBoxFloats]; // LDS-DMA landing zone: per band [(k, j) row][column]
    __shared__ int s_mid[3][NB][32];
    __shared__ float s_w[3][NB][kSlotX][3];
    __shared__ float s_oscale[NB];
    // derived per brick and band by the store waves an iteration before the collapse waves need them:
    __shared__ __attribute__((aligned(16))) float s_wy[NB][kMaxK][kBY]; // zero-padded y weights x the band's factor: [box row j][row of samples]
    __shared__ float s_wz[NB][64];                                      // per plane: three z weights in tap order + the first tap's register slot: [plane*8 + j]
    __shared__ int s_prep[NB][4];                                       // {slot of the box's first row, fresh rows, first column & 3, -}
    __shared__ int s_bandc[NB][4]; // {K, box_off, rowlen, first_pass}: read with ds_read, not reloaded from the kernel arguments

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_window = wave < kWW, is_collapse = wave >= kWW && wave < kWW + kPW;
    const GridArgs &g = a.g;
    float *const ring = lds + 2 * kRPlane;

    // bricks: id = (bx * nby + by) * nbz + bz; each workgroup owns a contiguous range and marches along z: consecutive
    // bricks mostly share their column (bx, by), so a band's coefficient box only moves by 0..2 rows in z between them
    const int nyz = a.nby * a.nbz;
    const long long total = (long long)a.nbx * nyz;
    int item0 = (int)(total * blockIdx.x / gridDim.x), item1 = (int)(total * (blockIdx.x + 1) / gridDim.x);
#ifdef WN_TUNE_ENV
    // experiments (profiles/r03_workgroup_end_times.txt): which workgroup -- workgroup w runs on XCD w mod 8 -- takes which range
    if (a.permute >= 10 && a.permute < 18 && gridDim.x == 256) { // range rotl8(w, permute - 10)
        const int r = a.permute - 10;
        const unsigned w = ((blockIdx.x << r) | (blockIdx.x >> (8 - r))) & 255u;
        item0 = (int)(total * w / gridDim.x);
        item1 = (int)(total * (w + 1) / gridDim.x);
    }
    if (a.even_permille != 500 && (gridDim.x & 1) == 0) { // uneven shares of a pair's bricks for its even / odd workgroup
        const int pair = blockIdx.x >> 1;
        const int p0 = (int)(total * (2 * pair) / gridDim.x), p1 = (int)(total * (2 * pair + 2) / gridDim.x);
        const int mid = p0 + (int)((long long)(p1 - p0) * a.even_permille / 1000);
        item0 = (blockIdx.x & 1) ? mid : p0;
        item1 = (blockIdx.x & 1) ? p1 : mid;
    }
#endif
    const int nb = item1 - item0;
    if (nb <= 0) return;
    const int G = nb * kBZ; // planes this workgroup produces
#ifdef WN_TUNE_ENV
    if (a.debug == 12 && blockIdx.x == 0 && tid == 0) { // shader clock of this launch: cycles and 100 MHz ticks, start / end
        g_mbp_stamps[0] = __builtin_amdgcn_s_memtime();
        g_mbp_stamps[1] = __builtin_amdgcn_s_memrealtime();
    }
    if (a.debug == 13 && tid == 0 && blockIdx.x < 256) { // every workgroup: start / end in 100 MHz ticks, its XCC
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_mbp_stamps[blockIdx.x * 3] = __builtin_amdgcn_s_memrealtime();
        g_mbp_stamps[blockIdx.x * 3 + 2] = xcc & 15;
    }
#endif

    struct Brick { int bx, by, bz; };
    auto brick_of = [&](int item) {
        Brick k;
        k.bx = item / nyz;
        k.by = (item - k.bx * nyz) / a.nbz;
        k.bz = item - k.bx * nyz - k.by * a.nbz;
        return k;
    };
    auto advance = [&](Brick &k) {
        if (++k.bz == a.nbz) { k.bz = 0; ++k.by; }
        if (k.by == a.nby) { k.by = 0; ++k.bx; }
    };

    // per-band sample tables of a brick (threads of the first window waves): mids and B-spline weights of its 8 rows,
    // 8 planes and first / last x sample, computed exactly as the reference does per sample
    auto fill_tables = [&](int buf, const Brick &k) {
        int tt = tid;
        asm volatile("" : "+v"(tt)); // what derives from it is computed here, not kept in (spilled) registers across the loop
        const int b = tt >> 5, slot = tt & 31;
        if (b < NB && slot < kSlotX + 2) {
            int idx;
            if (slot < kSlotZ) idx = min(k.by * kBY + slot, g.ny - 1);
            else if (slot < kSlotX) idx = g.z0 + min(k.bz * kBZ + slot - kSlotZ, g.nz - 1);
            else idx = (slot == kSlotX) ? k.bx * kBX : min(k.bx * kBX + kBX - 1, g.nx - 1);
            // (the band's scale from LDS: indexed by a lane value the kernel argument becomes a vector-memory load, which in
            // this kernel would queue behind the store waves' stores)
            const float c = coord_of(idx, (float)g.den, a.inv_den, g.base_range, s_oscale[b], g.post_scale);
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
    auto sgpr = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    // Box rows in z that band b needs FRESH for brick B (tables tB) when brick A (tables tA) was the previous one of this
    // workgroup: in the same column the box moved up by d = 0, 1 or 2 rows and keeps its other rows (the collapse waves shift
    // their registers); anything else -- another column, a larger move, the first brick -- is a whole box, d = K.
    // Both the store waves (what to request) and the collapse waves (what to shift) derive it from the same tables.
    auto fresh_rows = [&](int b, int K, int tA, int tB, bool same_column) {
        const int d = sgpr(s_mid[tB][b][kSlotZ]) - sgpr(s_mid[tA][b][kSlotZ]);
        return (same_column && d >= 0 && d <= 2) ? d : K;
    };

    // One wave per band b (a store wave: few live registers, time to spare), for the brick with tables tB that follows the
    // brick with tables tA:
    //  * s_wz[b][plane*8 + {0,1,2 | 3}]: the K box rows in z live in K register slots of the collapse waves as a ring -- tile
    //    row kabs sits in slot kabs mod K, so the rows a brick keeps from its predecessor stay where they are; a plane gets its
    //    three z weights (tap order) and the slot of its first tap;  * s_wy[b][j][yi]: zero-padded y weights with the band's factor folded in;
    //  * s_prep[b]: the slot of the box's first row, how many of its top rows are fresh, its first column modulo 4.
    auto derive_tables = [&](int b, int tA, int tB, bool same_column, bool first) {
        const int l = lane;
        {
            const int K = s_bandc[b][0];
            const int kz0 = s_mid[tB][b][kSlotZ] - 1;
            // K is 4 or 5: the two modulos by cases (a division by a run-time K is ~30 instructions)
            const int kzm = (K == 4) ? (kz0 & 3) : ((kz0 + 20 * 65536) % 5); // slot of the box's first row
            { // lane (zi, j): j < 3 the plane's three z weights in tap order, j == 3 the register slot of its first tap
                const int zi = l >> 3, j = l & 7;
                int t0 = kzm + (s_mid[tB][b][kSlotZ + zi] - s_mid[tB][b][kSlotZ]); // < 2K
                t0 -= (t0 >= K) ? K : 0;
                const float w0 = s_w[tB][b][kSlotZ + zi][0], w1 = s_w[tB][b][kSlotZ + zi][1], w2 = s_w[tB][b][kSlotZ + zi][2];
                s_wz[b][l] = j == 0 ? w0 : j == 1 ? w1 : j == 2 ? w2 : __int_as_float(t0);
            }
            if (l < kMaxK * kBY) {
                const int j = l >> 3, yi = l & 7;
                const int dd = j - (s_mid[tB][b][yi] - s_mid[tB][b][0]);
                const float w0 = s_w[tB][b][yi][0], w1 = s_w[tB][b][yi][1], w2 = s_w[tB][b][yi][2];
                float f = a.band[0].factor;
#pragma unroll
                for (int bb = 1; bb < NB; ++bb) f = (b == bb) ? a.band[bb].factor : f;
                s_wy[b][j][yi] = ((unsigned)dd <= 2u ? (dd == 0 ? w0 : dd == 1 ? w1 : w2) : 0.0f) * f;
            }
            if (l == 0) {
                const int dz = s_mid[tB][b][kSlotZ] - s_mid[tA][b][kSlotZ];
                s_prep[b][0] = kzm;
                s_prep[b][1] = (!first && same_column && dz >= 0 && dz <= 2) ? dz : K;
                s_prep[b][2] = (s_mid[tB][b][kSlotX] - 1) & 3;
            }
        }
    };

    Brick cur = brick_of(item0), nxt = cur;
    advance(nxt);
    Brick nxt2 = nxt;
    advance(nxt2);
    if (tid < NB) {
        int K = a.band[0].K, box_off = a.band[0].box_off, rowlen = a.band[0].rowlen, fp = a.band[0].first_pass;
#pragma unroll
        for (int bb = 1; bb < NB; ++bb) {
            K = (tid == bb) ? a.band[bb].K : K;
            box_off = (tid == bb) ? a.band[bb].box_off : box_off;
            rowlen = (tid == bb) ? a.band[bb].rowlen : rowlen;
            fp = (tid == bb) ? a.band[bb].first_pass : fp;
        }
        s_bandc[tid][0] = K;
        s_bandc[tid][1] = box_off;
        s_bandc[tid][2] = rowlen;
        s_bandc[tid][3] = fp;
        float os = a.band[0].oscale;
#pragma unroll
        for (int bb = 1; bb < NB; ++bb) os = (tid == bb) ? a.band[bb].oscale : os;
        s_oscale[tid] = os;
    }
    __syncthreads();
    fill_tables(0, cur);
    if (nb > 1) fill_tables(1, nxt);
    __syncthreads();
    int tb = 0, t = 0; // table buffer and index of the current brick
    auto next_brick = [&]() {
        cur = nxt;
        nxt = nxt2;
        advance(nxt2);
        tb = (tb == 2) ? 0 : tb + 1;
        ++t;
    };

    // Iteration gp (one barrier each): collapse waves write plane gp+1 into R[(gp+1)&1]; window waves contract plane gp
    // from R[gp&1] into ring[gp&3]; store waves move plane gp-2 from ring[(gp-2)&3] to memory and, during the first
    // planes of a brick, request the next brick's boxes.  Tables are triple buffered by brick.
    // Each role runs its OWN loop (same number of barriers): in one shared loop the register allocator sees the window
    // waves' 80 weights and the collapse waves' 80 collapsed coefficients live together and spills.
    if (is_collapse) {
        // ---- collapse waves ------------------------------------------------------------------------------------
        struct Pass { // wave-uniform facts of a pass
            int id, band, col0, box_off, rowlen;
            float factor;
        };
        Pass ps[2];
        int pk[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Pass &P = ps[i];
            P.id = (wave - kWW) + kPW * i;
            P.band = sgpr(a.pass_band[P.id]);
            const int bb0 = max(P.band, 0);
            pk[i] = P.band >= 0 ? sgpr(s_bandc[bb0][0]) : 0;
            P.box_off = sgpr(s_bandc[bb0][1]);
            P.rowlen = sgpr(s_bandc[bb0][2]);
            P.col0 = 64 * (P.id - sgpr(s_bandc[bb0][3]));
            float f = a.band[0].factor;
#pragma unroll
            for (int bb = 1; bb < NB; ++bb) f = (bb0 == bb) ? a.band[bb].factor : f;
            P.factor = __int_as_float(sgpr(__float_as_int(f)));
        }
        // The box has landed: the lane's column comes out of the DMA zone and is collapsed in y ONCE per brick --
        // yc[yi][k] = sum_j wy[yi][j] * c[k][j] (the band's factor folded in), weights zero-padded to the box and uniform
        // (v_readlane -> SGPR operand), one box row in y at a time -- so that a plane costs 8 x K FMAs with K uniform z
        // weights.  (First version: z then y per plane, 8 x K readlane-fed dependent FMAs per plane: 4,000-5,600 cycles a
        // plane by the time stamps.)  Marching along z, brick t+1's box is brick t's moved up by d rows: the rows it keeps
        // stay in their registers, only the d fresh rows are collapsed (whole boxes cost ~4,400 cycles of the brick's last
        // plane and 89 DMA requests a brick; at 512^3 x 5 bands d = 2 of 5 rows for the top band, <= 1 of 4 below).
        // wzv: lanes zi*8 + 0..2 = the z weights of plane zi in tap order, lane zi*8 + 3 = the slot of its first tap.
        auto prep_brick = [&](const Pass &P, auto &yc, float &wzv, auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (K > 0) {
                const int kzm = sgpr(s_prep[P.band][0]), d = sgpr(s_prep[P.band][1]);
                wzv = s_wz[P.band][lane];
                // (lanes past the box's last column -- their R columns are never read -- stay inside the row image)
                const float *img = s_box + P.box_off + min(sgpr(s_prep[P.band][2]) + P.col0 + lane, P.rowlen - 1);
                // the box's top d rows are fresh: the DMA zone holds them as image rows (f, j), f = 0..d-1
#pragma unroll
                for (int sl = 0; sl < K; ++sl) {
                    const int tau = (sl - kzm + K) % K;
                    if (tau >= K - d) { // wave-uniform
                        const float *row = img + (tau - (K - d)) * K * P.rowlen;
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            const float cj = row[j * P.rowlen];
                            const v2f c2 = {cj, cj};
#pragma unroll
                            for (int h = 0; h < 2; ++h) { // a box row's 8 weights: two broadcast ds_read_b128 into VGPRs
                                const v4f w4 = *reinterpret_cast<const v4f *>(&s_wy[P.band][j][4 * h]);
                                const v2f wlo = {w4.x, w4.y}, whi = {w4.z, w4.w};
                                yc[2 * h][sl] = (j == 0) ? wlo * c2 : __builtin_elementwise_fma(wlo, c2, yc[2 * h][sl]);
                                yc[2 * h + 1][sl] = (j == 0) ? whi * c2 : __builtin_elementwise_fma(whi, c2, yc[2 * h + 1][sl]);
                            }
                        }
                    }
                }
            }
        };
        // collapse z for plane zi of the brick: 8 row pieces of R (64 columns each; row stride and slot are constants)
        auto p1_slice = [&](float *Rw, int zi, const auto &yc, float wzv, auto kc) {
            constexpr int K = decltype(kc)::value;
            if constexpr (K > 0) {
                // the three taps in the order of the rows (first tap first), whichever slots hold them: the sum's bits depend
                // on the tile rows and the weights only -- not on where the march began, and they repeat with the tile's period
                const float w0 = readlane_f(wzv, zi * 8), w1 = readlane_f(wzv, zi * 8 + 1), w2 = readlane_f(wzv, zi * 8 + 2);
                const int t0 = __builtin_amdgcn_readlane(__float_as_int(wzv), zi * 8 + 3);
                auto taps = [&](auto tc) {
                    constexpr int T0 = decltype(tc)::value, T1 = (T0 + 1) % K, T2 = (T0 + 2) % K;
#pragma unroll
                    for (int p2 = 0; p2 < kBY / 2; ++p2) { // rows of samples in pairs: v_pk_fma_f32, the weight a scalar splat
                        v2f v = v2f{w0, w0} * yc[p2][T0];
                        v = __builtin_elementwise_fma(v2f{w1, w1}, yc[p2][T1], v);
                        v = __builtin_elementwise_fma(v2f{w2, w2}, yc[p2][T2], v);
                        Rw[(2 * p2) * kRRow] = v.x;
                        Rw[(2 * p2 + 1) * kRRow] = v.y;
                    }
                };
                using std::integral_constant;
                if (t0 == 0) taps(integral_constant<int, 0>{});
                else if (t0 == 1) taps(integral_constant<int, 1>{});
                else if (t0 == 2) taps(integral_constant<int, 2>{});
                else if (t0 == 3 || K == 4) taps(integral_constant<int, 3>{});
                else taps(integral_constant<int, K - 1>{});
            }
        };
        // Both passes' K fixed at compile time (K = 0: no such pass): register arrays of exactly 8 x K values per pass and
        // no joins between K variants inside the loop.
        auto collapse_role = [&](auto k0c, auto k1c) {
            constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
            v2f yc0[kBY / 2][K0 > 0 ? K0 : 1], yc1[kBY / 2][K1 > 0 ? K1 : 1]; // [pair of rows of samples][slot]
            float wzv0 = 0.0f, wzv1 = 0.0f;
            float *const Rw0 = lds + 64 * ps[0].id + lane, *const Rw1 = lds + 64 * ps[1].id + lane;
            lds_barrier(); // the store waves' requests for the first brick's boxes have landed
            prep_brick(ps[0], yc0, wzv0, k0c);
            prep_brick(ps[1], yc1, wzv1, k1c);
            p1_slice(Rw0, 0, yc0, wzv0, k0c);
            p1_slice(Rw1, 0, yc1, wzv1, k1c);
            lds_barrier();
            for (int gp = 0; gp < G + 2; ++gp) {
                MBP_STAMP(0);
                const int zi = gp & (kBZ - 1);
                if (gp < G) {
                    if (gp + 1 < G) {
                        if (zi == kBZ - 1) { // the next brick's boxes landed an iteration ago
                            prep_brick(ps[0], yc0, wzv0, k0c);
                            prep_brick(ps[1], yc1, wzv1, k1c);
                        }
                        MBP_STAMP(1);
#ifdef WN_TUNE_ENV
                        if (a.debug != 10)
#endif
                        {
                            const int rs = ((gp + 1) & 1) * kRPlane;
                            p1_slice(Rw0 + rs, (gp + 1) & (kBZ - 1), yc0, wzv0, k0c);
                            p1_slice(Rw1 + rs, (gp + 1) & (kBZ - 1), yc1, wzv1, k1c);
                        }
                    }
                    if (zi == kBZ - 1) next_brick();
                }
                MBP_STAMP(2);
                lds_barrier();
            }
        };
        using std::integral_constant;
        const int k0 = pk[0], k1 = pk[1]; // passes are dealt in order: a wave's second pass exists only if its first does
        if (k0 == 4 && k1 == 4) collapse_role(integral_constant<int, 4>{}, integral_constant<int, 4>{});
        else if (k0 == 4 && k1 == 5) collapse_role(integral_constant<int, 4>{}, integral_constant<int, 5>{});
        else if (k0 == 5 && k1 == 4) collapse_role(integral_constant<int, 5>{}, integral_constant<int, 4>{});
        else if (k0 == 5 && k1 == 5) collapse_role(integral_constant<int, 5>{}, integral_constant<int, 5>{});
        else if (k0 == 4) collapse_role(integral_constant<int, 4>{}, integral_constant<int, 0>{});
        else if (k0 == 5) collapse_role(integral_constant<int, 5>{}, integral_constant<int, 0>{});
        else collapse_role(integral_constant<int, 0>{}, integral_constant<int, 0>{});
    } else if (is_window) {
        // ---- window waves: x contraction ---------------------------------------------------------------------------
        const int xw = (wave >> 2) & 1, wr = wave & 3; // x half of the brick, row pair (wr, wr + 4) of each plane
        float ww[NB][4][4];
        int wbase[NB]; // float offset of the lane's window in an R row: the band's columns + the window's first column
        auto x_weights = [&](int tbuf, int bx) {
            const float den = (float)g.den;
            const int x0 = bx * kBX + xw * 256 + lane * 4;
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
                    float cc = xbase[q] * a.band[b].oscale;
                    cc = cc * g.post_scale;
                    wn::bspline(cc, m[q], w[q][0], w[q][1], w[q][2]);
                }
                wbase[b] = 64 * a.band[b].first_pass + m[0] - s_mid[tbuf][b][kSlotX]; // (m0 - 1) - ix0
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
        auto phaseC = [&](int rslot, int ringslot) {
            // both rows of the wave per band: 4 independent packed FMA chains, one LDS round trip per band
            float acc[2][4];
            const float *Rp = lds + rslot * kRPlane + wr * kRRow;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                float v[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[h][i] = Rp[h * 4 * kRRow + wbase[b] + i];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float s = (b == 0) ? ww[b][q][0] * v[h][0] : __builtin_fmaf(ww[b][q][0], v[h][0], acc[h][q]);
                        s = __builtin_fmaf(ww[b][q][1], v[h][1], s);
                        s = __builtin_fmaf(ww[b][q][2], v[h][2], s);
                        s = __builtin_fmaf(ww[b][q][3], v[h][3], s);
                        acc[h][q] = s;
                    }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                *reinterpret_cast<v4f *>(ring + (ringslot * kBY + wr + 4 * h) * kBX + xw * 256 + lane * 4) =
                    v4f{acc[h][0], acc[h][1], acc[h][2], acc[h][3]};
        };

        x_weights(0, cur.bx);
        lds_barrier();
        lds_barrier();
        for (int gp = 0; gp < G + 2; ++gp) {
            MBP_STAMP(0);
            const int zi = gp & (kBZ - 1);
            if (gp < G) {
#ifdef WN_TUNE_ENV
                if (a.debug != 11)
#endif
                phaseC(gp & 1, gp & (kRing - 1));
                MBP_STAMP(1);
                if (zi == kBZ - 1) {
                    const int tb1 = (tb == 2) ? 0 : tb + 1, tb2 = (tb1 == 2) ? 0 : tb1 + 1;
                    if (t + 2 < nb) fill_tables(tb2, nxt2); // tables of brick t+2
                    if (t + 1 < nb && nxt.bx != cur.bx) x_weights(tb1, nxt.bx);
                    next_brick();
                }
            }
            MBP_STAMP(2);
            lds_barrier();
        }
#ifdef WN_TUNE_ENV
        if (a.debug == 12 && blockIdx.x == 0 && tid == 0) {
            g_mbp_stamps[2] = __builtin_amdgcn_s_memtime();
            g_mbp_stamps[3] = __builtin_amdgcn_s_memrealtime();
        }
        if (a.debug == 13 && tid == 0 && blockIdx.x < 256) g_mbp_stamps[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    } else {
        // ---- store waves ---------------------------------------------------------------------------------------------
        const int sw = wave - (kWW + kPW);
        // LDS-DMA of a brick's boxes: one global_load_lds_dwordx4 per (k, j) box row -- lane l moves the 16-byte chunk l of
        // the row (columns 4l..4l+3 from the box's first column rounded down to a multiple of 4, so that no chunk straddles
        // the tile's wrap-around), the row address is scalar.  One band per plane over the brick's first planes, a band's rows
        // dealt to the four store waves.  Rows past the box's extent of this brick carry zero weights (wrapped addresses are
        // always valid memory).
        // Row offsets first, one lane per box row of a band (a dozen vector instructions per band and brick); a request is
        // then a v_readlane, a 64-bit scalar add, the M0 write and the load.  (Computing each row's address with scalar
        // arithmetic at the request cost ~25 instructions = ~400 cycles of the wave per request.)
        int bK[NB], bBox[NB], bRow[NB]; // per band: box rows, image offset, image row length (launch constants, in SGPRs)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bK[b] = sgpr(s_bandc[b][0]);
            bBox[b] = sgpr(s_bandc[b][1]);
            bRow[b] = sgpr(s_bandc[b][2]);
        }
        // band b's fresh rows for the brick with tables tB (its predecessor: tA): row offsets, then the requests of this wave
        auto request_band = [&](auto bc, int tA, int tB, bool same_column, bool first) {
            constexpr int b = decltype(bc)::value;
            if constexpr (b < NB) {
                const int K = bK[b];
                const int d = first ? K : fresh_rows(b, K, tA, tB, same_column);
                const int jy0 = sgpr(s_mid[tB][b][0]) - 1, kz0 = sgpr(s_mid[tB][b][kSlotZ]) - 1 + (K - d);
                const int r = min(lane, K * K - 1);
                const int f = (K == 4) ? (r >> 2) : ((r * 13) >> 6); // r / K for r < 25
                const int j = r - f * K;
                const int rowoff = (((kz0 + f) & a.nmask) * a.n + ((jy0 + j) & a.nmask)) * a.n; // lane r: tile row of image row r = (f, j)
                const int ix0 = (sgpr(s_mid[tB][b][kSlotX]) - 1) & ~3;
                const unsigned voff = (unsigned)((ix0 + 4 * lane) & a.nmask) * 4u;
                for (int rr = sw; rr < d * K; rr += kSW) {
                    // the v_readlane with every lane active: inside the lanes' `if` the row offsets of the inactive lanes
                    // are undefined (and the compiler does sink their computation into it)
                    const char *rowp = reinterpret_cast<const char *>(a.coef + __builtin_amdgcn_readlane(rowoff, rr));
                    if (4 * lane < bRow[b])
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rowp + voff),
                                                         (__attribute__((address_space(3))) void *)&s_box[bBox[b] + rr * bRow[b]],
                                                         16, 0, 0);
                }
            }
        };
        Brick sb = cur; // brick of the plane the store waves move next
        // the wave's two rows of a plane, both halves: pointers into the output advance by a plane per plane
        const size_t plane_stride = (size_t)g.ny * g.nx;
        float *orow = nullptr;
        auto brick_pointer = [&]() {
            orow = a.out + ((size_t)(sb.bz * kBZ) * g.ny + sb.by * kBY + 2 * sw) * g.nx + sb.bx * kBX + lane * 4;
        };
        brick_pointer();
        auto store_plane = [&](int p) {
            const int zi = p & (kBZ - 1), ringslot = p & (kRing - 1);
            const float *src = ring + (ringslot * kBY + 2 * sw) * kBX + lane * 4;
            v4f v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const v4f *>(src + (q >> 1) * kBX + (q & 1) * 256);
            if (a.all_full) { // every store of every plane happens: four 1-KiB wave stores, addresses by pointer bumps
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f *>(orow + (size_t)(q >> 1) * g.nx + (q & 1) * 256) = v[q];
            } else {
                const int z = sb.bz * kBZ + zi;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 2 * sw + (q >> 1), xh = q & 1;
                    const int y = sb.by * kBY + row, x = sb.bx * kBX + xh * 256 + lane * 4;
                    if (y < g.ny && z < g.nz && x + 3 < g.nx)
                        *reinterpret_cast<v4f *>(orow + (size_t)(q >> 1) * g.nx + (q & 1) * 256) = v[q];
                }
            }
            orow += plane_stride;
            if (zi == kBZ - 1) {
                advance(sb);
                brick_pointer();
            }
        };

        using std::integral_constant;
        request_band(integral_constant<int, 0>{}, 0, 0, false, true); // the first brick's boxes: every row of every band
        request_band(integral_constant<int, 1>{}, 0, 0, false, true);
        request_band(integral_constant<int, 2>{}, 0, 0, false, true);
        request_band(integral_constant<int, 3>{}, 0, 0, false, true);
        request_band(integral_constant<int, 4>{}, 0, 0, false, true);
        for (int b = sw; b < NB; b += kSW) derive_tables(b, 0, 0, false, true);
#ifdef WN_TUNE_ENV
        if (a.debug == 20 && blockIdx.x == 0) { // dump the derived tables of the first brick
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (sw == 0) {
                float *dump = reinterpret_cast<float *>(g_mbp_stamps);
                for (int i = lane; i < NB * kMaxK * kBY; i += 64) dump[i] = (&s_wy[0][0][0])[i];
                for (int i = lane; i < NB * 64; i += 64) dump[256 + i] = (&s_wz[0][0])[i];
                for (int i = lane; i < NB * 4; i += 64) dump[640 + i] = (float)(&s_prep[0][0])[i];
                for (int i = lane; i < NB * 4; i += 64) dump[680 + i] = (float)(&s_bandc[0][0])[i];
            }
            for (int gp = 0; gp < G + 2; ++gp) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            return;
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();
        lds_barrier();
        for (int gp = 0; gp < G + 2; ++gp) {
            MBP_STAMP(0);
            const int zi = gp & (kBZ - 1);
            if (gp >= 2) store_plane(gp - 2);
            if (gp < G) {
                // the next brick's boxes: a part per plane over the first planes of this brick; complete (this wave's share)
                // before the barrier that ends plane kDmaLanded, which publishes every wave's share to the collapse waves
                if (zi < NB && t + 1 < nb) { // band zi of the next brick
                    const int tb1 = (tb == 2) ? 0 : tb + 1;
                    const bool same = nxt.bx == cur.bx && nxt.by == cur.by;
                    if (zi == 0) request_band(integral_constant<int, 0>{}, tb, tb1, same, false);
                    else if (zi == 1) request_band(integral_constant<int, 1>{}, tb, tb1, same, false);
                    else if (zi == 2) request_band(integral_constant<int, 2>{}, tb, tb1, same, false);
                    else if (zi == 3) request_band(integral_constant<int, 3>{}, tb, tb1, same, false);
                    else request_band(integral_constant<int, 4>{}, tb, tb1, same, false);
                }
                // what the collapse waves need of the next brick at this brick's last plane: one band per store wave (the first
                // wave also the fifth), in the plane before it.  (Spread over two planes it slowed both: a plane takes as long as
                // its slowest wave, and the store waves are the slowest in every plane they do anything besides storing.)
                if (zi == kBZ - 2 && t + 1 < nb)
                    for (int b = sw; b < NB; b += kSW)
                        derive_tables(b, tb, (tb == 2) ? 0 : tb + 1, nxt.bx == cur.bx && nxt.by == cur.by, false);
                // vmcnt counts this wave's memory instructions in issue order: at most 4 x (kDmaLanded - kDmaPlanes + 1)
                // stores are younger than its last box row, so this wait covers every box row and no more stores than that
                // (with bricks that stick out of the lattice some stores are skipped and the count does not hold: wait for all)
                if (zi == kDmaLanded) {
                    if (a.all_full) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (kDmaLanded - kDmaPlanes + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (zi == kBZ - 1) next_brick();
            }
            MBP_STAMP(2);
            lds_barrier();
        }
    }
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

template <int NB>
bool launch_mbp(const MbArgs &a, size_t lds, long long bricks, hipStream_t s)
{
    const void *fn = reinterpret_cast<const void *>(&grid3d_mbp_kernel<NB>);
    if (!wn::ensure_dynamic_lds(fn, wn::current_device(), lds)) return false;
    const int cus = wn::device_compute_units(wn::current_device());
    const int grid = (int)std::min<long long>(bricks, cus); // one 16-wave workgroup per CU (LDS)
    hipLaunchKernelGGL((grid3d_mbp_kernel<NB>), dim3(grid), dim3(64 * kWaves), lds, s, a);
    return true;
}

} // namespace

#ifdef WN_TUNE_ENV
extern "C" __attribute__((visibility("default"))) int wn_debug_mbp_stamps(long long *out, int count)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mbp_stamps), sizeof(long long) * (size_t)count);
}
#endif

namespace wn {

// Plans and launches the plane-pipeline kernel when the lattice is in its regime (else *launched = false and the
// caller goes on to the strip / brick kernels): 1..5 consecutive-octave bands, rows wider than 256 samples and a multiple of 4,
// power-of-two tile, every band's (y, z) box of 8 samples at most 5 rows, at most 8 passes of 64 box columns.
int multiband_try(const wn_tile *tile, const GridArgs &g, int nbands, const float *oscale, const float *weights,
                  float out_div, float *out_dev, hipStream_t stream, bool *launched, int min_bricks_per_cu)
{
    *launched = false;
#ifdef WN_TUNE_ENV
    if (getenv("WN_NO_MBP")) return WN_OK;
    if (nbands == 1 && getenv("WN_NO_MBP1")) return WN_OK;
#endif
    if (nbands < 1 || nbands > kMaxNB || tile->n == 0 || pow2_mask(tile->n) < 0) return WN_OK;
    if (g.nx <= 256 || (g.nx & 3) || g.ny <= 0 || g.nz <= 0 || g.z_const_mode || g.z0 < 0) return WN_OK;
    if (reinterpret_cast<uintptr_t>(out_dev) & 15) return WN_OK;
    MbArgs a{};
    a.even_permille = 500;
    const double imax = std::max<double>(std::max(g.nx, g.ny), (double)g.z0 + g.nz);
    int passes = 0, box_off = 0;
    for (int w = 0; w < kPasses; ++w) a.pass_band[w] = -1;
    for (int b = 0; b < nbands; ++b) {
        const double step = (double)g.base_range * (double)oscale[b] * (double)g.post_scale / g.den;
        if (!(step >= 0.0) || !std::isfinite(step)) return WN_OK;
        const double pmax = step * imax + 1.0;
        if (pmax > 1.0e6) return WN_OK; // mids stay far inside the int / float-exact range
        const double slack = pmax * 4.8e-7; // 4 ulp of the largest coordinate
        if (3.0 * step + slack > 1.0) return WN_OK; // 4 consecutive samples span <= 2 mids
        auto extent = [&](int samples) { return (int)floor((samples - 1) * step + slack) + 1 + 3; };
        const int K = std::max(4, std::max(extent(kBY), extent(kBZ)));
        if (K > kMaxK) return WN_OK;
        const int ex = extent(kBX) + 1;
        const int np = (ex + 63) / 64;
        if (passes + np > kPasses) return WN_OK;
        a.band[b].first_pass = passes;
        for (int p = 0; p < np; ++p) a.pass_band[passes++] = b;
        a.band[b].oscale = oscale[b];
        a.band[b].factor = (float)((double)(weights ? weights[b] : 1.0f) * (double)g.out_scale / (double)out_div);
        a.band[b].K = K;
        a.band[b].rowlen = (ex + 3 + 3) & ~3; // the box's columns + the <= 3 columns before its first one, whole 16-byte chunks
        if (a.band[b].rowlen > 256) return WN_OK; // one chunk per lane
        a.band[b].box_off = box_off;
        box_off += K * K * a.band[b].rowlen;
    }
    if (box_off > kBoxFloats) return WN_OK;
#ifdef WN_TUNE_ENV
    if (const char *e = getenv("WN_MBP_DEBUG")) a.debug = atoi(e);
    if (const char *e = getenv("WN_MBP_EVEN_SHARE")) a.even_permille = atoi(e);
    if (const char *e = getenv("WN_MBP_PERMUTE")) a.permute = atoi(e);
#endif
    const size_t lds = (size_t)(2 * kRPlane + kRingFloats) * sizeof(float);
    a.coef = tile->dev;
    a.out = out_dev;
    a.n = tile->n;
    a.nmask = pow2_mask(tile->n);
    a.g = g;
    a.inv_den = ((g.den & (g.den - 1)) == 0) ? 1.0f / (float)g.den : 0.0f;
    a.nbx = (g.nx + kBX - 1) / kBX;
    a.nby = (g.ny + kBY - 1) / kBY;
    a.nbz = (g.nz + kBZ - 1) / kBZ;
    a.all_full = (g.nx % kBX == 0 && g.ny % kBY == 0 && g.nz % kBZ == 0) ? 1 : 0;
    const long long bricks = (long long)a.nbx * a.nby * a.nbz;
    if (bricks > 0x7fffffffLL / kBZ) return WN_OK;
    // (min_bricks_per_cu: a caller may keep small lattices away; the product passes 0 -- the kernel of a sample must not depend
    // on the thickness of the slab it is computed in.)  Every caller keeps away lattices whose last 512-wide brick column is
    // mostly padding (768 = 512 + 256: 394 us here, 314 us on 256-wide bricks).
    if (bricks < (long long)min_bricks_per_cu * wn::device_compute_units(wn::current_device())) return WN_OK;
    if ((long long)a.nbx * kBX * 10 > (long long)g.nx * 11) return WN_OK;
    bool ok;
    switch (nbands) {
    case 1: ok = launch_mbp<1>(a, lds, bricks, stream); break;
    case 2: ok = launch_mbp<2>(a, lds, bricks, stream); break;
    case 3: ok = launch_mbp<3>(a, lds, bricks, stream); break;
    case 4: ok = launch_mbp<4>(a, lds, bricks, stream); break;
    default: ok = launch_mbp<5>(a, lds, bricks, stream); break;
    }
    if (!ok) return WN_OK; // the runtime refused the LDS opt-in: the brick kernel takes the lattice
    *launched = true;
    WN_LAUNCH_CHECK("grid3d_mbp_kernel");
    return WN_OK;
}

} // namespace wn
