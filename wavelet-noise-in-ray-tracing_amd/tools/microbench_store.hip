// microbench_store.hip -- development tool: what can one MI355X sustain for a pure fp32 store
// stream of the dense-grid shape?  (The 512^3 grid kernel is bound by its 512 MiB store.)
// Build: hipcc --offload-arch=gfx950 -O3 microbench_store.hip -o microbench_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <array>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void fill_linear(float4 *out, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 x = make_float4(v, v + 1, v + 2, v + 3);
        if (NT) __builtin_nontemporal_store(v4f{x.x, x.y, x.z, x.w}, reinterpret_cast<v4f *>(out + i)); else out[i] = x;
    }
}

// brick pattern of grid3d_sep_kernel: WG = 256 x 8 x 8 samples of a 512^3 volume; wave w writes
// rows w, w+4, ... (1 KiB per wave store).
template <bool NT>
__global__ __launch_bounds__(256) void fill_bricks(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b = blockIdx.x;
    const int nbx = N / 256, nby = N / 8;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby, bz = b / nby;
    for (int row = wave; row < 64; row += 4) {
        const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
        float4 x = make_float4(v, v + row, v + 2, v + 3);
        float4 *dst = reinterpret_cast<float4 *>(out + ((size_t)z * N + y) * N + bx * 256 + lane * 4);
        if (NT) __builtin_nontemporal_store(v4f{x.x, x.y, x.z, x.w}, reinterpret_cast<v4f *>(dst)); else *dst = x;
    }
}

// persistent variant: G workgroups, each owns a contiguous range of bricks (as grid3d_sep_kernel)
template <bool NT>
__global__ __launch_bounds__(256) void fill_bricks_persistent(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbx = N / 256, nby = N / 8, nbz = N / 8, nyz = nby * nbz;
    const long long T = (long long)nbx * nyz;
    int item = (int)(T * blockIdx.x / gridDim.x);
    const int end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    for (; item < end; ++item) {
        const int bx = item / nyz, yz = item - bx * nyz, bz = yz / nby, by = yz - bz * nby;
        for (int row = wave; row < 64; row += 4) {
            const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256) + lane;
            if (NT) __builtin_nontemporal_store(x, dst); else *dst = x;
        }
    }
}

// generic persistent brick fill: brick = (XW*256) x BY x BZ samples, 4*XW waves, rows dealt to waves
template <int XW, int BY, int BZ>
__global__ __launch_bounds__(256 * XW) void fill_shape(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int nbx = N / (256 * XW), nby = N / BY, nbz = N / BZ, nyz = nby * nbz;
    const long long T = (long long)nbx * nyz;
    int item = (int)(T * blockIdx.x / gridDim.x);
    const int end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    for (; item < end; ++item) {
        const int bx = item / nyz, yz = item - bx * nyz, bz = yz / nby, by = yz - bz * nby;
        for (int row = wr; row < BY * BZ; row += 4) {
            const int y = by * BY + (row % BY), z = bz * BZ + (row / BY);
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256 * XW + xw * 256) + lane;
            *dst = x;
        }
    }
}

// cache-policy variants of the persistent brick fill: MODE 0 plain, 1 sc1 (write-through, line dropped
// from L2), 2 sc0 sc1, 3 nt
template <int MODE>
__global__ __launch_bounds__(256) void fill_policy(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbx = N / 256, nby = N / 8, nbz = N / 8, nyz = nby * nbz;
    const long long T = (long long)nbx * nyz;
    int item = (int)(T * blockIdx.x / gridDim.x);
    const int end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    for (; item < end; ++item) {
        const int bx = item / nyz, yz = item - bx * nyz, bz = yz / nby, by = yz - bz * nby;
        for (int row = wave; row < 64; row += 4) {
            const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256) + lane;
            if (MODE == 0) *dst = x;
            else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(x) : "memory");
            else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(x) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(x) : "memory");
        }
    }
}

// store bursts separated by synthetic work: per brick `valu` dependent FMA steps per wave,
// `lds` LDS read-modify-write steps, optional barrier -- what slows a store stream down?
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void fill_with_work(float *out, int N, float v, int valu, int lds, int barrier)
{
    __shared__ float sh[64 * WAVES * 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int nbx = N / (64 * WAVES), nby = N / 8, nbz = N / 8, nyz = nby * nbz;
    const long long T = (long long)nbx * nyz;
    int item = (int)(T * blockIdx.x / gridDim.x);
    const int end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    float acc = v + lane;
    sh[threadIdx.x] = acc;
    for (; item < end; ++item) {
        for (int i = 0; i < valu; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
        for (int i = 0; i < lds; ++i) {
            sh[threadIdx.x] = acc;
            acc += sh[(threadIdx.x + 17 * (i + 1)) % (64 * WAVES)];
        }
        if (barrier) __syncthreads();
        const int bx = item / nyz, yz = item - bx * nyz, bz = yz / nby, by = yz - bz * nby;
        for (int row = wr; row < 64; row += 4) {
            const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
            v4f x = v4f{acc, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 64 * WAVES + xw * 256) + lane;
            *dst = x;
        }
    }
}

// persistent brick fill, INTERLEAVED assignment: at step t the G workgroups write bricks t*G .. t*G+G-1
// (a compact region of the volume) instead of each owning a contiguous range of bricks
template <int XW>
__global__ __launch_bounds__(256 * XW) void fill_bricks_interleaved(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int nbx = N / (256 * XW), nby = N / 8, nbz = N / 8, nyz = nby * nbz;
    const int T = nbx * nyz;
    for (int item = blockIdx.x; item < T; item += gridDim.x) {
        const int bx = item / nyz, yz = item - bx * nyz, bz = yz / nby, by = yz - bz * nby;
        for (int row = wr; row < 64; row += 4) {
            const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256 * XW + xw * 256) + lane;
            *dst = x;
        }
    }
}

// interleaved assignment with flat bricks: brick = (XW*256) x BY rows of ONE plane (BY*XW KiB contiguous);
// at step t the G workgroups write bricks t*G .. t*G+G-1 = G*BY*XW KiB contiguous
template <int XW, int BY>
__global__ __launch_bounds__(256 * XW) void fill_flat_interleaved(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int nbx = N / (256 * XW), nby = N / BY;
    const long long T = (long long)nbx * nby * N;
    for (long long item = blockIdx.x; item < T; item += gridDim.x) {
        const int bx = (int)(item % nbx);
        const long long r = item / nbx;
        const int by = (int)(r % nby), z = (int)(r / nby);
        for (int row = wr; row < BY; row += 4) {
            const int y = by * BY + row;
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256 * XW + xw * 256) + lane;
            *dst = x;
        }
    }
}

// persistent bricks, contiguous ranges, but bricks ordered z-FASTEST: a workgroup keeps its y rows (hence
// its HBM channels, if channels interleave every few KiB with a 1 MiB period) and walks through the planes
template <int XW>
__global__ __launch_bounds__(256 * XW) void fill_bricks_zfast(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int nbx = N / (256 * XW), nby = N / 8, nbz = N / 8;
    const long long T = (long long)nbx * nby * nbz;
    int item = (int)(T * blockIdx.x / gridDim.x);
    const int end = (int)(T * (blockIdx.x + 1) / gridDim.x);
    for (; item < end; ++item) {
        const int bz = item % nbz, r = item / nbz, by = r % nby, bx = r / nby;
        for (int row = wr; row < 64; row += 4) {
            const int y = by * 8 + (row & 7), z = bz * 8 + (row >> 3);
            v4f x = v4f{v, v + row, v + 2, v + 3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + bx * 256 * XW + xw * 256) + lane;
            *dst = x;
        }
    }
}

// y-strips: workgroup g owns BY rows (y = g*BY ..) of every plane and walks z in bricks of BZ planes; with
// BY rows = 4 KiB and 256 workgroups the chip writes whole planes in step, every CU always the same 4 KiB
// slot of each MiB (the pattern of the 256-workgroup linear fill)
template <int XW, int BY, int BZ>
__global__ __launch_bounds__(256 * XW) void fill_strips(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw = wave >> 2, wr = wave & 3;
    const int strips = N / BY;
    for (int strip = blockIdx.x; strip < strips; strip += gridDim.x) {
        for (int z0 = 0; z0 < N; z0 += BZ) {
            for (int row = wr; row < BY * BZ; row += 4) {
                const int y = strip * BY + (row % BY), z = z0 + row / BY;
                v4f x = v4f{v, v + row, v + 2, v + 3};
                v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + xw * 256) + lane;
                *dst = x;
            }
        }
    }
}

// 4 waves per workgroup: wave w writes half (w&1) of row (w>>1) of a 2-row strip; RPW rows-pairs per wave step
template <int PLANES>
__global__ __launch_bounds__(256) void fill_strips4(float *out, int N, float v)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1, r = wave >> 1;
    const int strips = N / 2;
    for (int strip = blockIdx.x; strip < strips; strip += gridDim.x) {
        for (int z0 = 0; z0 < N; z0 += PLANES) {
#pragma unroll
            for (int p = 0; p < PLANES; ++p) {
                const int y = strip * 2 + r, z = z0 + p;
                v4f x = v4f{v, v + p, v + 2, v + 3};
                v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + half * 256) + lane;
                *dst = x;
            }
        }
    }
}

// the strip-march pattern with per-step work: `fma` independent-ish FMAs and one LDS write->read round trip
__global__ __launch_bounds__(256) void fill_strips4_work(float *out, int N, float v, int fma, int lds_trip)
{
    __shared__ float sh[2][256 + 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = wave & 1, r = wave >> 1;
    const int strips = N / 2;
    float a0 = v + lane, a1 = v - lane, a2 = 0.5f * lane, a3 = 1.0f;
    for (int strip = blockIdx.x; strip < strips; strip += gridDim.x) {
        for (int z = 0; z < N; ++z) {
            for (int i = 0; i < fma; i += 4) {
                a0 = __builtin_fmaf(a0, 1.0001f, a1);
                a1 = __builtin_fmaf(a1, 0.9999f, a2);
                a2 = __builtin_fmaf(a2, 1.0002f, a3);
                a3 = __builtin_fmaf(a3, 0.9998f, a0);
            }
            if (lds_trip) {
                sh[z & 1][threadIdx.x] = a0;
                a1 += sh[(z & 1) ^ 1][(threadIdx.x & 192) + ((lane + 1) & 63)];
            }
            const int y = strip * 2 + r;
            v4f x = v4f{a0, a1, a2, a3};
            v4f *dst = reinterpret_cast<v4f *>(out + ((size_t)z * N + y) * N + half * 256) + lane;
            *dst = x;
        }
    }
}

// Decoupled store stream: CW compute waves do `fma` FMAs per step and park their 1-KiB row in LDS; SW store
// waves (no compute) move parked rows to memory.  One workgroup barrier per step, stage double-buffered, so
// compute of step z+1 overlaps the stores of step z.  Items as in the strip kernel: (segment, z half).
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits for every
// outstanding global store of the wave to complete
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CW, int SW, int MAP = 0, int LDS_PAD = 0>
__global__ __launch_bounds__(64 * (CW + SW)) void fill_decoupled(float *out, int N, float v, int fma)
{
    __shared__ v4f stage[2][CW][64];
    __shared__ float pad[LDS_PAD + 1]; // LDS_PAD floats of ballast: limits the workgroups per CU like the real kernel's tables
    if (LDS_PAD && threadIdx.x == 0 && fma < 0) pad[LDS_PAD] = v;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = N / 2, segs = N * 2;
    const int base = blockIdx.x * CW;
    if (wave < CW) {
        float a0 = v + lane, a1 = v - lane, a2 = 0.5f * lane, a3 = 1.0f;
        for (int z = 0; z < steps; ++z) {
            for (int i = 0; i < fma; i += 4) {
                a0 = __builtin_fmaf(a0, 1.0001f, a1);
                a1 = __builtin_fmaf(a1, 0.9999f, a2);
                a2 = __builtin_fmaf(a2, 1.0002f, a3);
                a3 = __builtin_fmaf(a3, 0.9998f, a0);
            }
            stage[z & 1][wave][lane] = v4f{a0, a1, a2, a3};
            lds_barrier();
        }
    } else {
        constexpr int PER = CW / SW;
        const int s = wave - CW;
        v4f *dst[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = base + s * PER + k, chunk = item / segs, seg = item - chunk * segs;
            int y = seg >> 1, half = seg & 1;
            if (MAP == 1) { // the strip kernel's items: four rows of one column block per workgroup
                const int grp = seg / CW, c = seg - grp * CW;
                y = (grp >> 1) * CW + c;
                half = grp & 1;
            }
            dst[k] = reinterpret_cast<v4f *>(out + ((size_t)(chunk * steps) * N + y) * N + half * 256) + lane;
        }
        const size_t plane4 = (size_t)N * N / 4;
        for (int z = 0; z < steps; ++z) {
            lds_barrier();
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                *dst[k] = stage[z & 1][s * PER + k][lane];
                dst[k] += plane4;
            }
        }
    }
}

// The same split, synchronised by per-wave progress counters in LDS instead of a barrier per step: a compute
// wave may run up to K rows ahead of its store wave.
template <int CW, int SW, int K>
__global__ __launch_bounds__(64 * (CW + SW)) void fill_decoupled_ring(float *out, int N, float v, int fma)
{
    __shared__ v4f stage[CW][K][64];
    __shared__ int produced[CW], consumed[CW];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int steps = N / 2, segs = N * 2;
    const int base = blockIdx.x * CW;
    if (threadIdx.x < CW) { produced[threadIdx.x] = 0; consumed[threadIdx.x] = 0; }
    __syncthreads();
    if (wave < CW) {
        float a0 = v + lane, a1 = v - lane, a2 = 0.5f * lane, a3 = 1.0f;
        int freed = 0; // rows the store wave is known to have taken
        for (int z = 0; z < steps; ++z) {
            for (int i = 0; i < fma; i += 4) {
                a0 = __builtin_fmaf(a0, 1.0001f, a1);
                a1 = __builtin_fmaf(a1, 0.9999f, a2);
                a2 = __builtin_fmaf(a2, 1.0002f, a3);
                a3 = __builtin_fmaf(a3, 0.9998f, a0);
            }
            while (z - freed >= K) {
                freed = __builtin_amdgcn_readfirstlane(*(volatile int *)&consumed[wave]);
                if (z - freed >= K) __builtin_amdgcn_s_sleep(2);
            }
            stage[wave][z % K][lane] = v4f{a0, a1, a2, a3};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile int *)&produced[wave] = z + 1;
        }
    } else {
        constexpr int PER = CW / SW;
        const int s = wave - CW;
        v4f *dst[PER];
        int done[PER], avail[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int item = base + s * PER + k, chunk = item / segs, seg = item - chunk * segs;
            const int y = seg >> 1, half = seg & 1;
            dst[k] = reinterpret_cast<v4f *>(out + ((size_t)(chunk * steps) * N + y) * N + half * 256) + lane;
            done[k] = 0; avail[k] = 0;
        }
        const size_t plane4 = (size_t)N * N / 4;
        int left = PER * steps;
        while (left > 0) {
            bool any = false;
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int c = s * PER + k;
                if (done[k] == avail[k] && done[k] < steps)
                    avail[k] = __builtin_amdgcn_readfirstlane(*(volatile int *)&produced[c]);
                if (done[k] < avail[k]) {
                    const v4f x = stage[c][done[k] % K][lane];
                    *dst[k] = x;
                    dst[k] += plane4;
                    ++done[k];
                    --left;
                    any = true;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) *(volatile int *)&consumed[c] = done[k];
                }
            }
            if (!any) __builtin_amdgcn_s_sleep(2);
        }
    }
}

__global__ __launch_bounds__(256) void copy_linear(const float4 *in, float4 *out, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

template <typename F>
float time_it(F launch, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b); ms.push_back(t / iters);
    }
    std::sort(ms.begin(), ms.end());
    return ms[2];
}

int main(int argc, char **)
{
    const int N = 512;
    const size_t n = (size_t)N * N * N, bytes = n * 4;
    float *out, *in;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&in, bytes)); CK(hipMemset(in, 0, bytes));
    auto report = [&](const char *name, float ms, double b) { printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, b / ms / 1e6); };
    if (argc > 2) { // two arguments: sustained-load drift of the plain store stream (per-launch HIP events)
        const int launches = 400;
        std::vector<hipEvent_t> ev(launches + 1);
        for (auto &e : ev) hipEventCreate(&e);
        hipDeviceSynchronize();
        hipEventRecord(ev[0]);
        for (int i = 0; i < launches; ++i) {
            fill_linear<false><<<256, 256>>>((float4 *)out, n / 4, 1.f);
            hipEventRecord(ev[i + 1]);
        }
        hipDeviceSynchronize();
        printf("fill_linear grid=256, %d back-to-back launches, us per launch:\n", launches);
        for (int i = 0; i < launches; ++i) {
            float t; hipEventElapsedTime(&t, ev[i], ev[i + 1]);
            if (i < 10 || i % 20 == 0 || i >= launches - 5) printf("  launch %3d %7.1f\n", i, t * 1e3);
        }
        return 0;
    }
    for (int fma : {0, 32}) {
        char nm[64];
        snprintf(nm, 64, "decoupled 8c+4s fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled<8, 4><<<256, 768>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled 8c+8s fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled<8, 8><<<256, 1024>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled 4c+4s fma=%d (2 WG per CU)", fma);
        report(nm, time_it([&] { fill_decoupled<4, 4><<<512, 512>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled 4c+4s rows4 map fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled<4, 4, 1><<<512, 512>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled 4c+4s rows4 +72KB LDS fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled<4, 4, 1, 15000><<<512, 512>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled 4c+4s +72KB LDS fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled<4, 4, 0, 15000><<<512, 512>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled ring 8c+4s K=8 fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled_ring<8, 4, 8><<<256, 768>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled ring 8c+4s K=4 fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled_ring<8, 4, 4><<<256, 768>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "decoupled ring 8c+8s K=8 fma=%d", fma);
        report(nm, time_it([&] { fill_decoupled_ring<8, 8, 8><<<256, 1024>>>(out, N, 1.f, fma); }, 20), bytes);
        snprintf(nm, 64, "strips4 + work fma=%d lds=1 (ref)", fma);
        report(nm, time_it([&] { fill_strips4_work<<<256, 256>>>(out, N, 1.f, fma, 1); }, 20), bytes);
    }
    if (argc > 1) return 0; // any argument: the decoupled section only
    for (int blocks : {2048, 8192, 32768}) {
        char nm[64];
        snprintf(nm, 64, "fill_linear plain  grid=%d", blocks);
        report(nm, time_it([&] { fill_linear<false><<<blocks, 256>>>((float4 *)out, n / 4, 1.f); }, 20), bytes);
        snprintf(nm, 64, "fill_linear nt     grid=%d", blocks);
        report(nm, time_it([&] { fill_linear<true><<<blocks, 256>>>((float4 *)out, n / 4, 1.f); }, 20), bytes);
    }
    report("fill_bricks plain", time_it([&] { fill_bricks<false><<<(N / 256) * (N / 8) * (N / 8), 256>>>(out, N, 1.f); }, 20), bytes);
    report("fill_bricks nt", time_it([&] { fill_bricks<true><<<(N / 256) * (N / 8) * (N / 8), 256>>>(out, N, 1.f); }, 20), bytes);
    for (int k : {1, 2, 4, 8}) {
        char nm[64];
        snprintf(nm, 64, "fill_bricks_persistent k=%d plain", k);
        report(nm, time_it([&] { fill_bricks_persistent<false><<<256 * k, 256>>>(out, N, 1.f); }, 20), bytes);
        snprintf(nm, 64, "fill_bricks_persistent k=%d nt", k);
        report(nm, time_it([&] { fill_bricks_persistent<true><<<256 * k, 256>>>(out, N, 1.f); }, 20), bytes);
    }
    for (int k : {1, 2}) {
        char nm[96];
        for (auto cfg : {std::array<int,3>{0,0,0}, std::array<int,3>{0,0,1}, std::array<int,3>{400,0,1}, std::array<int,3>{1600,0,1}, std::array<int,3>{0,200,1}, std::array<int,3>{400,200,1}}) {
            snprintf(nm, 96, "work 8w k=%d valu=%d lds=%d bar=%d", k, cfg[0], cfg[1], cfg[2]);
            report(nm, time_it([&] { fill_with_work<8><<<256 * k, 512>>>(out, N, 1.f, cfg[0], cfg[1], cfg[2]); }, 10), bytes);
        }
    }
    for (int blocks : {256, 512, 1024}) {
        char nm[64];
        snprintf(nm, 64, "fill_linear plain  grid=%d", blocks);
        report(nm, time_it([&] { fill_linear<false><<<blocks, 256>>>((float4 *)out, n / 4, 1.f); }, 20), bytes);
    }
    report("bricks interleaved 256w k=1", time_it([&] { fill_bricks_interleaved<1><<<256, 256>>>(out, N, 1.f); }, 20), bytes);
    report("bricks interleaved 256w k=2", time_it([&] { fill_bricks_interleaved<1><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("bricks interleaved 512w k=1", time_it([&] { fill_bricks_interleaved<2><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("bricks interleaved 512w k=2", time_it([&] { fill_bricks_interleaved<2><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x4   k=1", time_it([&] { fill_flat_interleaved<2, 4><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x8   k=1", time_it([&] { fill_flat_interleaved<2, 8><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x16  k=1", time_it([&] { fill_flat_interleaved<2, 16><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x32  k=1", time_it([&] { fill_flat_interleaved<2, 32><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x64  k=1", time_it([&] { fill_flat_interleaved<2, 64><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 512x8   k=2", time_it([&] { fill_flat_interleaved<2, 8><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 256x8   k=1", time_it([&] { fill_flat_interleaved<1, 8><<<256, 256>>>(out, N, 1.f); }, 20), bytes);
    report("flat interleaved 256x16  k=2", time_it([&] { fill_flat_interleaved<1, 16><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("bricks z-fastest 512w k=1", time_it([&] { fill_bricks_zfast<2><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("bricks z-fastest 512w k=2", time_it([&] { fill_bricks_zfast<2><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("bricks z-fastest 256w k=1", time_it([&] { fill_bricks_zfast<1><<<256, 256>>>(out, N, 1.f); }, 20), bytes);
    report("bricks z-fastest 256w k=2", time_it([&] { fill_bricks_zfast<1><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("bricks z-fastest 256w k=4", time_it([&] { fill_bricks_zfast<1><<<1024, 256>>>(out, N, 1.f); }, 20), bytes);
    report("strips 512 x2 x8", time_it([&] { fill_strips<2, 2, 8><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("strips 512 x2 x4", time_it([&] { fill_strips<2, 2, 4><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("strips 512 x2 x2", time_it([&] { fill_strips<2, 2, 2><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("strips 512 x4 x8 (128 WGs)", time_it([&] { fill_strips<2, 4, 8><<<128, 512>>>(out, N, 1.f); }, 20), bytes);
    report("strips 512 x2 x16", time_it([&] { fill_strips<2, 2, 16><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("strips4 (4 waves) planes=1", time_it([&] { fill_strips4<1><<<256, 256>>>(out, N, 1.f); }, 20), bytes);
    report("strips4 (4 waves) planes=8", time_it([&] { fill_strips4<8><<<256, 256>>>(out, N, 1.f); }, 20), bytes);
    for (auto cfg : {std::array<int,2>{0,0}, std::array<int,2>{16,0}, std::array<int,2>{16,1}, std::array<int,2>{32,1}, std::array<int,2>{64,1}}) {
        char nm[64];
        snprintf(nm, 64, "strips4 + work fma=%d lds=%d", cfg[0], cfg[1]);
        report(nm, time_it([&] { fill_strips4_work<<<256, 256>>>(out, N, 1.f, cfg[0], cfg[1]); }, 20), bytes);
    }
    report("policy plain    k=2", time_it([&] { fill_policy<0><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("policy sc1      k=2", time_it([&] { fill_policy<1><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("policy sc0 sc1  k=2", time_it([&] { fill_policy<2><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("policy nt       k=2", time_it([&] { fill_policy<3><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("shape 256x8x8   k=2", time_it([&] { fill_shape<1, 8, 8><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("shape 256x16x4  k=2", time_it([&] { fill_shape<1, 16, 4><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("shape 256x32x2  k=2", time_it([&] { fill_shape<1, 32, 2><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("shape 256x64x1  k=2", time_it([&] { fill_shape<1, 64, 1><<<512, 256>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x8x8   k=1", time_it([&] { fill_shape<2, 8, 8><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x8x8   k=2", time_it([&] { fill_shape<2, 8, 8><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x32x2  k=2", time_it([&] { fill_shape<2, 32, 2><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x64x1  k=2", time_it([&] { fill_shape<2, 64, 1><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x64x1  k=1", time_it([&] { fill_shape<2, 64, 1><<<256, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x16x1  k=2", time_it([&] { fill_shape<2, 16, 1><<<512, 512>>>(out, N, 1.f); }, 20), bytes);
    report("shape 512x16x1  k=4", time_it([&] { fill_shape<2, 16, 1><<<1024, 512>>>(out, N, 1.f); }, 20), bytes);
    report("copy_linear (read+write bytes)", time_it([&] { copy_linear<<<8192, 256>>>((const float4 *)in, (float4 *)out, n / 4); }, 20), 2.0 * bytes);
    hipMemsetAsync(out, 0, bytes);
    report("hipMemsetAsync", time_it([&] { hipMemsetAsync(out, 0, bytes); }, 20), bytes);
    return 0;
}
