// wn_mailbox.hip -- the reference's SCALAR members (evaluate2D/3D/3DProjected(p), noise(x,y,z),
// fractal_noise(p), texture::value(u,v,p): one value per call) without a kernel launch per call.
//
// A launch plus a stream synchronise costs ~22 us; the reference's renderer makes 29.6 million such
// calls (main.cpp:38-59 through material.h:72).  Here a RESIDENT one-wave kernel polls a mailbox in
// pinned, device-mapped host memory: the host writes the request (one 64-byte line, sequence number
// last), the wave picks it up, evaluates it with the same exact device functions the batched kernels
// use (wn_device_eval.hpp / wn_texture_eval.hpp -> bit-identical results), and writes value and
// sequence number back.  A call is two PCIe round trips plus the evaluation.
//
// The kernel is not immortal: after kIdleTicks (2 ms) without a request, or kLifeTicks (20 ms) after its start
// however many requests keep arriving, it marks the mailbox STOPPED and exits -- so a device-wide synchronise
// (hipDeviceSynchronize, a hipFree inside wn_dev_free / wn_tile_destroy, torch.cuda.synchronize) issued by another
// thread in the middle of a burst of scalar calls never waits longer than ~20 ms, and nothing is left spinning when
// the process ends; the next scalar call starts a fresh instance (one ordinary launch, ~15 us every 20 ms).  A request
// posted while an instance is timing out is never lost: the host re-launches when it sees STOPPED with
// its request unanswered, and an instance starts from the last ANSWERED sequence number.
#include "wn_internal.hpp"
#include "wn_device_eval.hpp"
#include "wn_texture_eval.hpp"

#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

namespace {

enum : uint32_t {
    kOpEval3d = 1, kOpEval2d, kOpProjected, kOpPerlin, kOpPerlinVec3, kOpWaveletTexture, kOpNoiseTexture
};
enum : uint32_t { kStopped = 0, kRunning = 1 };
constexpr unsigned long long kIdleTicks = 200000ull;  // wall_clock64() ticks at 100 MHz: 2 ms without a request
constexpr unsigned long long kLifeTicks = 2000000ull; // ... and 20 ms in all, however busy: see the kernel

// One 64-byte line: the host fills everything, then stores `seq` (release).  The device reads the line
// with one wave-wide load and acts only when `seq` moved.
struct alignas(64) Request {
    uint32_t op;
    int32_t n;      // tile size (0: empty tile) / turb depth
    uint64_t ptr;   // coefficient tile or permutation table (device pointer)
    union {
        float f[8];
        double d[4];
    } a;            // points, normals, scales
    int32_t aux;    // texture mode / perlin kind / padded-tile flag
    uint32_t check; // makes the xor of the 16 dwords, each ROTATED LEFT BY ITS INDEX, zero: a line that arrives torn
                    // (sequence number ahead of the arguments) does not verify and is simply polled again.  The rotation
                    // makes the test position-dependent: with a plain xor, stale dwords whose old ^ new differences cancel
                    // (evaluate2D walked along the diagonal: dwords 4 and 5 change alike) would verify under the new seq.
    uint64_t seq;
};
static_assert(sizeof(Request) == 64, "one line");

struct alignas(16) Response {
    double value;
    uint64_t seq;
};

struct TexParams { // what wn::wavelet_texture_value reads
    const float *coef;
    int n, nmask, mode;
    double scale;
    float octave_mul, inv_stddev;
};

__device__ __forceinline__ int mask_of(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

__device__ double serve(const Request &r)
{
    const int n = r.n, nmask = mask_of(n);
    switch (r.op) {
    case kOpEval3d: {
        const float *coef = reinterpret_cast<const float *>(r.ptr);
        return r.aux ? (double)wn::eval3d_exact<true>(coef, n, nmask, r.a.f[0], r.a.f[1], r.a.f[2])
                     : (double)wn::eval3d_exact<false>(coef, n, nmask, r.a.f[0], r.a.f[1], r.a.f[2]);
    }
    case kOpEval2d:
        return (double)wn::eval2d_exact(reinterpret_cast<const float *>(r.ptr), n, nmask, r.a.f[0], r.a.f[1]);
    case kOpProjected: {
        const float p[3] = {r.a.f[0], r.a.f[1], r.a.f[2]}, nr[3] = {r.a.f[3], r.a.f[4], r.a.f[5]};
        return (double)wn::projected_exact(reinterpret_cast<const float *>(r.ptr), n, nmask, p, nr);
    }
    case kOpPerlin:
        return wn::perlin_exact(reinterpret_cast<const uint8_t *>(r.ptr), r.a.d[0], r.a.d[1], r.a.d[2]);
    case kOpPerlinVec3: {
        const uint8_t *perm = reinterpret_cast<const uint8_t *>(r.ptr);
        if (r.aux == 1) return wn::perlin_turb(perm, r.a.f[0], r.a.f[1], r.a.f[2], r.n);
        if (r.aux == 2) return wn::perlin_fractal(perm, r.a.f[0], r.a.f[1], r.a.f[2]);
        return wn::perlin_exact(perm, (double)r.a.f[0], (double)r.a.f[1], (double)r.a.f[2]);
    }
    case kOpWaveletTexture: {
        TexParams t;
        t.coef = reinterpret_cast<const float *>(r.ptr);
        t.n = n;
        t.nmask = nmask;
        t.mode = r.aux & 3;
        t.scale = r.a.d[0];
        t.octave_mul = r.a.f[2];
        t.inv_stddev = r.a.f[3];
        return (r.aux & 4) ? (double)wn::wavelet_texture_value<true>(t, r.a.f[4], r.a.f[5], r.a.f[6])
                           : (double)wn::wavelet_texture_value<false>(t, r.a.f[4], r.a.f[5], r.a.f[6]);
    }
    case kOpNoiseTexture:
        return (double)wn::noise_texture_value(reinterpret_cast<const uint8_t *>(r.ptr), r.a.f[0], r.a.f[1],
                                               r.a.f[2], r.a.f[3], r.a.f[4]);
    default:
        return 0.0;
    }
}

// One wave.  `req`, `resp`, `state` are device aliases of pinned host memory (reads and writes cross PCIe).
__global__ __launch_bounds__(64) void mailbox_kernel(const uint32_t *req, Response *resp, uint32_t *state,
                                                     unsigned long long last_seq)
{
    const int lane = threadIdx.x;
    unsigned long long idle_since = wall_clock64();
    const unsigned long long born = idle_since;
    for (;;) {
        // the request line in one wave-wide load: lanes 0..15 take one dword each, bypassing the caches
        uint32_t word = 0;
        if (lane < 16) word = __hip_atomic_load(req + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long seq = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 14) |
                                       ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)word, 15) << 32);
        uint32_t fold = (word << (lane & 15)) | (word >> ((32 - (lane & 15)) & 31)); // rotl by the dword's index; lanes >= 16 hold 0
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) fold ^= (uint32_t)__shfl_xor((int)fold, off, 64);
        if (seq != last_seq && __builtin_amdgcn_readfirstlane((int)fold) == 0) {
            Request r;
            uint32_t *w = reinterpret_cast<uint32_t *>(&r);
#pragma unroll
            for (int i = 0; i < 16; ++i) w[i] = (uint32_t)__builtin_amdgcn_readlane((int)word, i);
            last_seq = seq;
            if (lane == 0) {
                // value and sequence number leave as ONE 16-byte store (one PCIe write): the host sees them together.
                // (Two stores with a release between them cost a wait for the first one's acknowledgement, ~1 us.)
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const double v = serve(r);
                const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
                __builtin_nontemporal_store(v4u{(unsigned)vb, (unsigned)(vb >> 32), (unsigned)seq, (unsigned)(seq >> 32)},
                                            reinterpret_cast<v4u *>(resp));
            }
            idle_since = wall_clock64();
            // a busy instance ends too (round-2 ADVICE: one thread's burst of scalar calls must not hold up another
            // thread's device-wide synchronise for the length of the burst); the host starts the next one
            if (idle_since - born > kLifeTicks) break;
        } else {
            if (wall_clock64() - idle_since > kIdleTicks) break; // every instance ends: nothing spins for ever
            __builtin_amdgcn_s_sleep(2);
        }
    }
    if (lane == 0) {
        __hip_atomic_store(state, (uint32_t)kStopped, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // the last thing it does
    }
}

struct Mailbox {
    Request *req = nullptr;       // host pointers
    Response *resp = nullptr;
    uint32_t *state = nullptr;
    void *req_dev = nullptr, *resp_dev = nullptr, *state_dev = nullptr;
    hipStream_t stream = nullptr;
    uint64_t seq = 0;
    unsigned long long launches = 0;
    bool failed = false; // a request timed out: later calls fail at once until the instance reports STOPPED
    std::mutex mu;       // scalar calls on one device are serialised; devices do not wait for each other
};

std::mutex g_mu; // guards the map only
std::map<int, Mailbox *> g_boxes;

int create_box(int device, Mailbox **out)
{
    Mailbox *b = new Mailbox();
    char *host = nullptr;
    hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&host), 256, hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) {
        delete b;
        wn::hip_fail(e, "hipHostMalloc(mailbox)");
        return WN_ERR_ALLOC;
    }
    std::memset(host, 0, 256);
    b->req = reinterpret_cast<Request *>(host);
    b->resp = reinterpret_cast<Response *>(host + 64);
    b->state = reinterpret_cast<uint32_t *>(host + 128);
    char *dev = nullptr;
    e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev), host, 0);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        (void)hipHostFree(host);
        delete b;
        return wn::hip_fail(e, "mailbox setup");
    }
    b->req_dev = dev;
    b->resp_dev = dev + 64;
    b->state_dev = dev + 128;
    g_boxes[device] = b;
    *out = b;
    return WN_OK;
}

int start_instance(Mailbox *b)
{
    WN_ENTRY(); // the launch below is where the HIP runtime draws from rand(): park the caller's stream here, not per call
    __atomic_store_n(b->state, (uint32_t)kRunning, __ATOMIC_RELEASE);
    const unsigned long long answered = __atomic_load_n(&b->resp->seq, __ATOMIC_ACQUIRE);
    hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, b->stream, static_cast<const uint32_t *>(b->req_dev),
                       static_cast<Response *>(b->resp_dev), static_cast<uint32_t *>(b->state_dev), answered);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        __atomic_store_n(b->state, (uint32_t)kStopped, __ATOMIC_RELEASE);
        return wn::hip_fail(e, "mailbox_kernel");
    }
    ++b->launches;
    return WN_OK;
}

// Post one request and wait for its answer.  Scalar calls of all threads on one device are serialised here.
int call(int device, Request &r, double *value)
{
    int rc = wn::require_device();
    if (rc) return rc;
    if (wn::current_device() != device) return wn::fail(WN_ERR_INVALID, "handle lives on device %d, current device is %d", device, wn::current_device());
    Mailbox *b = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_boxes.find(device);
        if (it != g_boxes.end()) b = it->second;
        else {
            WN_ENTRY();
            if ((rc = create_box(device, &b)) != WN_OK) return rc;
        }
    }
    std::lock_guard<std::mutex> lock(b->mu);
    if (b->failed) { // an earlier request never came back: do not spin another 10 s behind it
        if (__atomic_load_n(b->state, __ATOMIC_ACQUIRE) != kStopped)
            return wn::fail(WN_ERR_HIP, "scalar mailbox: the resident kernel did not answer an earlier request and is still busy");
        b->failed = false;
    }

    const uint64_t seq = ++b->seq;
    r.seq = seq;
    r.check = 0;
    uint32_t fold = 0;
    for (int i = 0; i < 16; ++i) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(&r)[i];
        fold ^= (w << i) | (w >> ((32 - i) & 31));
    }
    constexpr int kCheckIndex = offsetof(Request, check) / 4;
    r.check = (fold >> kCheckIndex) | (fold << ((32 - kCheckIndex) & 31)); // rotr: the rotated xor of all 16 dwords is now 0
    std::memcpy(b->req, &r, offsetof(Request, seq));           // everything but the sequence number ...
    __atomic_store_n(&b->req->seq, seq, __ATOMIC_RELEASE);     // ... which goes last
    if (__atomic_load_n(b->state, __ATOMIC_ACQUIRE) != kRunning && (rc = start_instance(b)) != WN_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&b->resp->seq, __ATOMIC_ACQUIRE) != seq) {
        if (__atomic_load_n(b->state, __ATOMIC_ACQUIRE) == kStopped) {
            // the instance timed out around our store; did it answer first?
            if (__atomic_load_n(&b->resp->seq, __ATOMIC_ACQUIRE) == seq) break;
            if ((rc = start_instance(b)) != WN_OK) return rc;
        }
        __builtin_ia32_pause();
        if ((++spins & 0xfffff) == 0 &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
            b->failed = true;
            return wn::fail(WN_ERR_HIP, "scalar mailbox: no answer from the device within 10 s");
        }
    }
    *value = b->resp->value;
    return WN_OK;
}

inline int pow2_mask(int n) { return (n > 0 && (n & (n - 1)) == 0) ? n - 1 : -1; }

} // namespace

using namespace wn;

// The scalar entry points do not park the caller's rand() state per call (wn_internal.hpp: RandStateGuard): a served request
// launches nothing, so the HIP runtime draws nothing; the guard is taken where an instance is started or a mailbox created.
extern "C" {

int wn_scalar_eval3d(const wn_tile *tile, const float p[3], float *out)
{
    if (!tile || !p || !out) return fail(WN_ERR_INVALID, "wn_scalar_eval3d: NULL argument");
    if (tile->count && tile->dims != 3) return fail(WN_ERR_INVALID, "wn_scalar_eval3d needs a 3-D tile");
    if (!tile->count) { // empty tile: 0.0f (WaveletNoise.cpp:186-188); still requires a device like every entry point
        int rc = require_device();
        *out = 0.0f;
        return rc;
    }
    Request r{};
    r.op = kOpEval3d;
    r.n = tile->n;
    r.aux = tile->dev_padded ? 1 : 0;
    r.ptr = reinterpret_cast<uint64_t>(tile->dev_padded ? tile->dev_padded : tile->dev);
    r.a.f[0] = p[0], r.a.f[1] = p[1], r.a.f[2] = p[2];
    double v = 0;
    int rc = call(tile->device, r, &v);
    *out = (float)v;
    return rc;
}

int wn_scalar_eval2d(const wn_tile *tile, const float p[2], float *out)
{
    if (!tile || !p || !out) return fail(WN_ERR_INVALID, "wn_scalar_eval2d: NULL argument");
    if (tile->count && tile->dims != 2) return fail(WN_ERR_INVALID, "wn_scalar_eval2d needs a 2-D tile");
    if (!tile->count) {
        int rc = require_device();
        *out = 0.0f;
        return rc;
    }
    Request r{};
    r.op = kOpEval2d;
    r.n = tile->n;
    r.ptr = reinterpret_cast<uint64_t>(tile->dev);
    r.a.f[0] = p[0], r.a.f[1] = p[1];
    double v = 0;
    int rc = call(tile->device, r, &v);
    *out = (float)v;
    return rc;
}

int wn_scalar_eval3d_projected(const wn_tile *tile, const float p[3], const float normal[3], float *out)
{
    if (!tile || !p || !normal || !out) return fail(WN_ERR_INVALID, "wn_scalar_eval3d_projected: NULL argument");
    if (tile->count && tile->dims != 3) return fail(WN_ERR_INVALID, "wn_scalar_eval3d_projected needs a 3-D tile");
    if (!tile->count) {
        int rc = require_device();
        *out = 0.0f;
        return rc;
    }
    Request r{};
    r.op = kOpProjected;
    r.n = tile->n;
    r.ptr = reinterpret_cast<uint64_t>(tile->dev);
    for (int i = 0; i < 3; ++i) r.a.f[i] = p[i], r.a.f[3 + i] = normal[i];
    double v = 0;
    int rc = call(tile->device, r, &v);
    *out = (float)v;
    return rc;
}

int wn_scalar_perlin(const wn_perm *perm, double x, double y, double z, double *out)
{
    if (!perm || !out) return fail(WN_ERR_INVALID, "wn_scalar_perlin: NULL argument");
    Request r{};
    r.op = kOpPerlin;
    r.ptr = reinterpret_cast<uint64_t>(perm->dev);
    r.a.d[0] = x, r.a.d[1] = y, r.a.d[2] = z;
    return call(perm->device, r, out);
}

int wn_scalar_perlin_vec3(const wn_perm *perm, const float p[3], int kind, int depth, double *out)
{
    if (!perm || !p || !out) return fail(WN_ERR_INVALID, "wn_scalar_perlin_vec3: NULL argument");
    if (kind < 0 || kind > 2 || depth < 0 || depth > 64) // 64 octaves reach past double precision; keeps one request bounded
        return fail(WN_ERR_INVALID, "wn_scalar_perlin_vec3: kind must be 0..2 and depth 0..64");
    Request r{};
    r.op = kOpPerlinVec3;
    r.aux = kind;
    r.n = depth;
    r.ptr = reinterpret_cast<uint64_t>(perm->dev);
    r.a.f[0] = p[0], r.a.f[1] = p[1], r.a.f[2] = p[2];
    return call(perm->device, r, out);
}

int wn_scalar_wavelet_texture(const wn_tile *tile, int use_3d, double scale, int octave, const float p[3],
                              float *grey)
{
    if (!p || !grey) return fail(WN_ERR_INVALID, "wn_scalar_wavelet_texture: NULL argument");
    const bool has_tile = tile && tile->count != 0;
    if (has_tile && tile->dims != (use_3d ? 3 : 2))
        return fail(WN_ERR_INVALID, "tile is %d-D but use_3d=%d", tile->dims, use_3d);
    if (!has_tile) { // texture.h:100-104: no noise object -> 0.5 * (1 + clamp(0)) = 0.5
        int rc = require_device();
        *grey = 0.5f;
        return rc;
    }
    const bool padded = use_3d && tile->dev_padded;
    Request r{};
    r.op = kOpWaveletTexture;
    r.n = tile->n;
    r.aux = (use_3d ? 3 : 2) | (padded ? 4 : 0);
    r.ptr = reinterpret_cast<uint64_t>(padded ? tile->dev_padded : tile->dev);
    r.a.d[0] = scale;
    r.a.f[2] = (float)std::pow(2.0, (double)octave) * 2.0f;            // texture.h:77-80
    r.a.f[3] = 1.0f / std::sqrt(use_3d ? 0.18402f : 0.19686f);        // texture.h:84,98
    r.a.f[4] = p[0], r.a.f[5] = p[1], r.a.f[6] = p[2];
    double v = 0;
    int rc = call(tile->device, r, &v);
    *grey = (float)v;
    return rc;
}

int wn_scalar_noise_texture(const wn_perm *perm, double scale, int octave, const float p[3], float *grey)
{
    if (!perm || !p || !grey) return fail(WN_ERR_INVALID, "wn_scalar_noise_texture: NULL argument");
    Request r{};
    r.op = kOpNoiseTexture;
    r.ptr = reinterpret_cast<uint64_t>(perm->dev);
    r.a.f[0] = (float)scale;
    r.a.f[1] = (float)std::pow(2.0, (double)octave); // texture.h:38
    r.a.f[2] = p[0], r.a.f[3] = p[1], r.a.f[4] = p[2];
    double v = 0;
    int rc = call(perm->device, r, &v);
    *grey = (float)v;
    return rc;
}

int wn_scalar_shutdown(void)
{
    // Ends every resident instance and releases the mailboxes (pinned memory, streams).  Instances end by themselves
    // within kIdleTicks of the last request; this waits for that.  Later scalar calls start over.
    std::lock_guard<std::mutex> lock(g_mu);
    for (auto &kv : g_boxes) {
        Mailbox *b = kv.second;
        std::lock_guard<std::mutex> box_lock(b->mu);
        (void)hipStreamSynchronize(b->stream);
        (void)hipStreamDestroy(b->stream);
        (void)hipHostFree(b->req);
    }
    for (auto &kv : g_boxes) delete kv.second;
    g_boxes.clear();
    return WN_OK;
}

int wn_scalar_stats(unsigned long long *calls, unsigned long long *launches)
{
    std::lock_guard<std::mutex> lock(g_mu);
    unsigned long long c = 0, l = 0;
    for (auto &kv : g_boxes) {
        c += kv.second->seq;
        l += kv.second->launches;
    }
    if (calls) *calls = c;
    if (launches) *launches = l;
    return WN_OK;
}

} // extern "C"
