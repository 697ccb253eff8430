// wn_tilegen.hip -- the filter half of WaveletNoise::generateNoiseTile2D/3D on the device
// (WaveletNoise.cpp:37-66 line filters, :87-107 and :153-182 pass order).
//
// Each pass low-passes every line along one axis: 32-tap analysis filter to n/2 samples, then
// the 4-tap synthesis filter back to n.  Products and sums are kept in the reference's order,
// unfused, so the tile is bit-identical to the CPU tile.  A workgroup takes a panel of adjacent
// lines through LDS; the last pass subtracts from the Gaussian field in place of a 4th kernel.
#include "wn_internal.hpp"

#include <algorithm>

namespace {

// Appendix-1 filter as the reference holds it (WaveletNoise.cpp:11-18).
__constant__ float c_analysis[32] = {
    0.000334f, -0.001528f, 0.000410f,  0.003545f, -0.000938f, -0.008233f, 0.002172f,  0.019120f,
    -0.005040f, -0.044412f, 0.011655f, 0.103311f, -0.025936f, -0.243780f, 0.033979f,  0.655340f,
    0.655340f,  0.033979f,  -0.243780f, -0.025936f, 0.103311f, 0.011655f, -0.044412f, -0.005040f,
    0.019120f,  0.002172f,  -0.008233f, -0.000938f, 0.003546f, 0.000410f, -0.001528f, 0.000334f};

struct PassArgs {
    const float *src;
    float *dst;
    const float *field; // when non-null: dst = field - lowpass (last pass)
    int n;              // line length
    size_t line_stride; // element stride along the line
    // lines are (a, b): base = a*stride_a + b*stride_b, a in [0,count_a), b in [0,count_b)
    int count_a;
    size_t stride_a, stride_b;
    int lines_per_wg;   // consecutive `a` values per workgroup
    int a_fastest;      // 1: threads sweep a first when touching global memory (stride_a == 1)
};

__global__ __launch_bounds__(256) void lowpass_lines_kernel(PassArgs p)
{
    extern __shared__ float lds[];
    const int n = p.n, half = n / 2, L = p.lines_per_wg;
    float *in = lds;                // [L][n]
    float *mid = lds + (size_t)L * n; // [L][half]
    const int a0 = blockIdx.x * L;
    const int b = blockIdx.y;
    const int lines = min(L, p.count_a - a0);
    const size_t base0 = (size_t)a0 * p.stride_a + (size_t)b * p.stride_b;

    // panel load
    const int total = lines * n;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int l, i;
        if (p.a_fastest) { l = e % lines; i = e / lines; } else { i = e % n; l = e / n; }
        in[l * n + i] = p.src[base0 + (size_t)l * p.stride_a + (size_t)i * p.line_stride];
    }
    __syncthreads();

    // analysis: to[i] = sum_{k=-16}^{15} a[k] * from[Mod(2i+k, n)]  (WaveletNoise.cpp:40-46)
    const int total_half = lines * half;
    for (int e = threadIdx.x; e < total_half; e += blockDim.x) {
        const int l = e / half, i = e - l * half;
        const float *from = in + l * n;
        float acc = 0.0f;
#pragma unroll
        for (int k = -16; k < 16; ++k) {
            int idx = (2 * i + k) % n;
            idx = idx < 0 ? idx + n : idx;
            acc += c_analysis[16 + k] * from[idx];
        }
        mid[l * half + i] = acc;
    }
    __syncthreads();

    // synthesis (WaveletNoise.cpp:55-64): even i: .75*c[i/2] then +.25*c[i/2+1];
    //                                      odd i:  .25*c[i/2] then +.75*c[i/2+1].
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        int l, i;
        if (p.a_fastest) { l = e % lines; i = e / lines; } else { i = e % n; l = e / n; }
        const float *from = mid + l * half;
        const int k = i / 2;
        const int k1 = (k + 1 == half) ? 0 : k + 1;
        const bool odd = (i & 1) != 0;
        float acc = 0.0f;
        acc += (odd ? 0.25f : 0.75f) * from[k];
        acc += (odd ? 0.75f : 0.25f) * from[k1];
        const size_t g = base0 + (size_t)l * p.stride_a + (size_t)i * p.line_stride;
        p.dst[g] = p.field ? (p.field[g] - acc) : acc;
    }
}

int run_pass(const PassArgs &args, int count_b, hipStream_t stream)
{
    PassArgs p = args;
    int L = 8192 / p.n;
    L = L < 1 ? 1 : (L > 32 ? 32 : L);
    p.lines_per_wg = L;
    const size_t lds = (size_t)L * p.n * sizeof(float) * 3 / 2;
    dim3 grid((p.count_a + L - 1) / L, count_b);
    hipLaunchKernelGGL(lowpass_lines_kernel, grid, dim3(256), lds, stream, p);
    WN_LAUNCH_CHECK("lowpass_lines_kernel");
    return WN_OK;
}

// rows with two wrap-around columns appended: dst[(z*n + y)*(n+2) + x] = src[(z*n + y)*n + (x mod n)]
__global__ __launch_bounds__(256) void padded_copy_kernel(const float *src, float *dst, int n)
{
    const int stride = n + 2;
    const size_t total = (size_t)stride * n * n;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % stride);
        const size_t row = e / stride;
        dst[e] = src[row * n + (x >= n ? x - n : x)];
    }
}

} // namespace

namespace wn {

int tile_build_padded(wn_tile *t, hipStream_t stream)
{
    if (t->dims != 3 || t->n == 0) return WN_OK;
    const size_t total = (size_t)(t->n + 2) * t->n * t->n;
    if (!t->dev_padded) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&t->dev_padded), total * sizeof(float));
        if (e != hipSuccess) {
            hip_fail(e, "hipMalloc(padded tile)");
            return WN_ERR_ALLOC;
        }
    }
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(padded_copy_kernel, dim3(blocks), dim3(256), 0, stream, t->dev, t->dev_padded, t->n);
    WN_LAUNCH_CHECK("padded_copy_kernel");
    WN_HIP(hipStreamSynchronize(stream));
    return WN_OK;
}

int tilegen_filter(wn_tile *t, const float *field_dev, hipStream_t stream)
{
    const int n = t->n;
    if (n == 0) return WN_OK;
    if (n < 2) return fail(WN_ERR_INVALID, "tile size must be >= 2");
    const size_t N = (size_t)n;
    float *t1 = nullptr, *t2 = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t1), t->count * sizeof(float));
    if (e == hipSuccess && t->dims == 3)
        e = hipMalloc(reinterpret_cast<void **>(&t2), t->count * sizeof(float));
    if (e != hipSuccess) {
        if (t1) (void)hipFree(t1);
        hip_fail(e, "hipMalloc(tilegen scratch)");
        return WN_ERR_ALLOC;
    }
    int rc = WN_OK;
    PassArgs p{};
    p.n = n;
    if (t->dims == 2) {
        // rows: line along x, lines enumerated by y            (WaveletNoise.cpp:87-92)
        p = PassArgs{field_dev, t1, nullptr, n, 1, n, N, 0, 0, 0};
        rc = run_pass(p, 1, stream);
        // columns: line along y, lines enumerated by x          (:95-100) + subtraction (:104-107)
        if (!rc) {
            p = PassArgs{t1, t->dev, field_dev, n, N, n, 1, 0, 0, 1};
            rc = run_pass(p, 1, stream);
        }
    } else {
        // X lines, for z, y                                      (:153-159)
        p = PassArgs{field_dev, t1, nullptr, n, 1, n, N, N * N, 0, 0};
        rc = run_pass(p, n, stream);
        // Y lines, for z, x                                      (:162-168)
        if (!rc) {
            p = PassArgs{t1, t2, nullptr, n, N, n, 1, N * N, 0, 1};
            rc = run_pass(p, n, stream);
        }
        // Z lines, for y, x                                      (:171-177) + subtraction (:179-182)
        if (!rc) {
            p = PassArgs{t2, t->dev, field_dev, n, N * N, n, 1, N, 0, 1};
            rc = run_pass(p, n, stream);
        }
    }
    hipError_t se = hipStreamSynchronize(stream);
    (void)hipFree(t1);
    if (t2) (void)hipFree(t2);
    if (!rc && se != hipSuccess) rc = hip_fail(se, "tile filter passes");
    return rc;
}

} // namespace wn
