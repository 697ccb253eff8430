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

int main()
{
    const int N = 512;
    const size_t n = (size_t)N * N * N, bytes = n * 4;
    float *out, *in;
    CK(hipMalloc(&out, bytes)); CK(hipMalloc(&in, bytes)); CK(hipMemset(in, 0, bytes));
    auto report = [&](const char *name, float ms, double b) { printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, b / ms / 1e6); };
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
