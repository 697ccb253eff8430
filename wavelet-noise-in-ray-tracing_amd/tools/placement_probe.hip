// placement_probe.hip -- which workgroups share a CU?  Launches a grid with the strip kernel's footprint
// (512 threads, 72 KiB of LDS: two workgroups per CU), every workgroup records where it runs and when it
// started, then idles ~30 us so that the whole grid is resident.  Prints the blockIdx sets per CU.
// Build: hipcc --offload-arch=gfx950 -O2 tools/placement_probe.hip -o tools/placement_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

struct Rec { unsigned block, hw_id, xcc_id; unsigned long long t0; };

__global__ __launch_bounds__(512) void probe(Rec *out)
{
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        Rec r;
        r.block = blockIdx.x;
        r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID
        r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_REG_XCC_ID
        r.t0 = wall_clock64();
        out[blockIdx.x] = r;
        lds[0] = 1.0f;
    }
    const unsigned long long t = wall_clock64();
    while (wall_clock64() - t < 3000) __builtin_amdgcn_s_sleep(8); // 100 MHz clock: 30 us
}

int main()
{
    const int blocks = 512;
    Rec *dev;
    hipMalloc(&dev, blocks * sizeof(Rec));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 73 * 1024);
    for (int rep = 0; rep < 3; ++rep) {
        probe<<<blocks, 512, 72 * 1024>>>(dev);
        std::vector<Rec> h(blocks);
        hipMemcpy(h.data(), dev, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
        std::map<unsigned, std::vector<unsigned>> per_cu;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (auto &r : h) {
            const unsigned cu = (r.xcc_id & 0xf) << 16 | ((r.hw_id >> 8) & 0xff); // xcc | se,sh,cu bits
            per_cu[cu].push_back(r.block);
            tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t0);
        }
        int pairs_half = 0, pairs_adjacent = 0, other = 0;
        std::map<size_t, int> sizes;
        for (auto &kv : per_cu) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            sizes[v.size()]++;
            if (v.size() == 2) {
                if (v[1] - v[0] == blocks / 2) ++pairs_half;
                else if (v[1] - v[0] == 1) ++pairs_adjacent;
                else ++other;
            }
        }
        printf("launch %d: %zu CUs used;", rep, per_cu.size());
        for (auto &s : sizes) printf(" %d CUs hold %zu WGs;", s.second, s.first);
        printf(" pairs (b, b+%d): %d, (b, b+1): %d, other: %d; first-to-last start %.2f us\n", blocks / 2, pairs_half,
               pairs_adjacent, other, (double)(tmax - tmin) / 100.0);
        int shown = 0;
        for (auto &kv : per_cu) {
            if (shown++ >= 6) break;
            printf("   cu %05x:", kv.first);
            for (unsigned b : kv.second) printf(" %u", b);
            printf("\n");
        }
    }
    return 0;
}
