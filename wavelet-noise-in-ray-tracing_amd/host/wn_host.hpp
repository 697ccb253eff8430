// wn_host.hpp -- glue shared by the host classes: error translation, a per-thread pinned scratch
// (batch-of-one calls for the few members without a wn_scalar_* entry point) and a device buffer.
//
// Everything that takes more than one sample -- point lists, textures' values(), dense grids, tile generation -- runs on
// the GPU through libwnoise_hip.so (include/wnoise.h); a failing ABI call (most commonly: no HIP device) throws
// std::runtime_error with wn_last_error(), nothing falls back to the CPU.  Only the reference's scalar members (one sample
// per call) are evaluated on the host (scalar_eval.h), from the mirrors the classes already keep.
#pragma once

#include <cstddef>
#include <stdexcept>
#include <string>

#include "wnoise.h"

namespace wnhost {

inline void check(int rc, const char *what)
{
    if (rc != WN_OK) throw std::runtime_error(std::string(what) + ": " + wn_last_error());
}

// Pinned, device-mapped staging for batch-of-one calls (WMultibandNoise): the kernel reads the point from and
// writes the result to host memory, so such a call is launch + sync, with no memcpy calls.  The reference's own
// scalar members (evaluate*, noise, value) are evaluated on the host (scalar_eval.h) or, with WN_SCALAR_ON_DEVICE=1, by the
// resident scalar kernel (wn_scalar_*, include/wnoise.h).
class Scratch {
  public:
    static Scratch &get()
    {
        thread_local Scratch s;
        return s;
    }
    float *in_host() { return reinterpret_cast<float *>(host_); }
    double *in_host64() { return reinterpret_cast<double *>(host_); }
    float *out_host() { return reinterpret_cast<float *>(host_ + kIn); }
    double *out_host64() { return reinterpret_cast<double *>(host_ + kIn); }
    void *in_dev() { return dev_; }
    void *out_dev() { return dev_ + kIn; }
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;

  private:
    static constexpr size_t kIn = 64, kBytes = 128;
    Scratch()
    {
        void *h = nullptr, *d = nullptr;
        check(wn_host_alloc_mapped(&h, &d, kBytes), "wn_host_alloc_mapped");
        host_ = static_cast<char *>(h);
        dev_ = static_cast<char *>(d);
    }
    ~Scratch() { wn_host_free_mapped(host_); }
    char *host_ = nullptr, *dev_ = nullptr;
};

// Device buffer for the batched host-pointer overloads.
class DeviceBuffer {
  public:
    explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(wn_dev_alloc(&p_, bytes), "wn_dev_alloc"); }
    ~DeviceBuffer() { wn_dev_free(p_); }
    void *get() const { return p_; }
    template <typename T> T *as() const { return static_cast<T *>(p_); }
    void upload(const void *src) { check(wn_copy_h2d(p_, src, bytes_, nullptr), "wn_copy_h2d"); }
    void download(void *dst) const
    {
        check(wn_copy_d2h(dst, p_, bytes_, nullptr), "wn_copy_d2h");
        check(wn_stream_sync(nullptr), "wn_stream_sync");
    }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;

  private:
    void *p_ = nullptr;
    size_t bytes_;
};

} // namespace wnhost
