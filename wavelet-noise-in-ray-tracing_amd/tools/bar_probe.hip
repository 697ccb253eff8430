// bar_probe.hip -- can the host write device memory directly (large BAR)?  Decides where the scalar mailbox's request line
// may live (csrc/wn_mailbox.hip): a request the GPU can poll in its own memory saves the PCIe read round trip of polling
// host memory.  Touches the candidate allocations under a SIGSEGV guard and times a host-write -> device-poll -> host-read
// ping-pong for each placement that works.
#include <hip/hip_runtime.h>

#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>

static sigjmp_buf g_jb;
static void on_segv(int) { siglongjmp(g_jb, 1); }

__global__ void pingpong(volatile uint32_t *req, volatile uint32_t *resp, int rounds)
{
    uint32_t last = 0;
    for (int r = 0; r < rounds; ++r) {
        uint32_t v;
        long long spins = 0;
        do {
            v = __hip_atomic_load((uint32_t *)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (++spins > 20000000ll) return; // never hang
        } while (v == last);
        last = v;
        __hip_atomic_store((uint32_t *)resp, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

static bool host_can_touch(volatile uint32_t *p)
{
    struct sigaction sa{}, old_segv{}, old_bus{};
    sa.sa_handler = on_segv;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool ok = false;
    if (sigsetjmp(g_jb, 1) == 0) {
        *p = 0x1234u;
        ok = (*p == 0x1234u);
        *p = 0;
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    return ok;
}

static double run(volatile uint32_t *req_host, uint32_t *req_dev, volatile uint32_t *resp_host, uint32_t *resp_dev, int rounds)
{
    *req_host = 0;
    *resp_host = 0;
    hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, 0, req_dev, resp_dev, rounds);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t r = 1; r <= (uint32_t)rounds; ++r) {
        __atomic_store_n((uint32_t *)req_host, r, __ATOMIC_RELEASE);
        long long spins = 0;
        while (__atomic_load_n((uint32_t *)resp_host, __ATOMIC_ACQUIRE) != r)
            if (++spins > 200000000ll) { std::printf("  host gave up at round %u\n", r); (void)hipDeviceSynchronize(); return -1; }
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    (void)hipDeviceSynchronize();
    return us;
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int rounds = 20000;
    uint32_t *pinned = nullptr, *pinned_dev = nullptr;
    if (hipHostMalloc((void **)&pinned, 4096, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return 1;
    (void)hipHostGetDevicePointer((void **)&pinned_dev, pinned, 0);
    std::printf("request in pinned host memory, response in pinned host memory: %.3f us per round trip\n",
                run(pinned, pinned_dev, pinned + 64, pinned_dev + 64, rounds));
    struct { const char *name; unsigned flags; } tries[] = {{"hipDeviceMallocFinegrained", hipDeviceMallocFinegrained},
                                                            {"hipDeviceMallocUncached", hipDeviceMallocUncached},
                                                            {"hipDeviceMallocDefault", hipDeviceMallocDefault}};
    for (auto &t : tries) {
        uint32_t *dev = nullptr;
        const hipError_t e = hipExtMallocWithFlags((void **)&dev, 4096, t.flags);
        if (e != hipSuccess) { std::printf("%s: allocation failed (%s)\n", t.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        (void)hipMemset(dev, 0, 4096);
        (void)hipDeviceSynchronize();
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, dev) == hipSuccess)
            std::printf("%s: type %d, hostPointer %p, devicePointer %p, isManaged %d\n", t.name, (int)at.type, at.hostPointer, at.devicePointer, (int)at.isManaged);
        std::printf("%s: touching it from the host ...\n", t.name);
        const bool ok = host_can_touch(dev);
        std::printf("%s: host %s touch it\n", t.name, ok ? "CAN" : "cannot");
        if (ok)
            std::printf("  request in device memory (host writes through the BAR), response in pinned host memory: %.3f us per round trip\n",
                        run(dev, dev, pinned + 64, pinned_dev + 64, rounds));
        (void)hipFree(dev);
    }
    return 0;
}
