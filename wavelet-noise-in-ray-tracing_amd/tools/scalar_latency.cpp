// scalar_latency.cpp -- what one scalar call of the reference's API costs on this library when it is sent to the device
// (WN_SCALAR_ON_DEVICE=1, set by this tool): the resident scalar kernel (wn_scalar_*, csrc/wn_mailbox.hip) against the
// launch-plus-synchronise form it replaces (a batch of one through wn_eval3d_points), and the restart after an idle gap --
// and, beside it, the host evaluator the classes use by default (host/scalar_eval.h).
//
//   scalar_latency [calls=20000]
// stdout: one JSON line {"mailbox_us_per_call":…, "launch_sync_us_per_call":…, "after_idle_us":…,
//                        "texture_us_per_call":…, "perlin_us_per_call":…, "host_evaluator_us_per_call":…, "mismatches":0, …}
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "texture.h"

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const int calls = argc > 1 ? std::atoi(argv[1]) : 20000;
    setenv("WN_SCALAR_ON_DEVICE", "1", 1); // read once, at the first scalar call
    try {
        WaveletNoise n3(128, 12345);
        n3.generateNoiseTile3D();
        perlin per(12345);
        wavelet_texture wt(1.0, 4, true);
        std::vector<float> pts(3 * (size_t)calls);
        unsigned s = 1u;
        for (auto &v : pts) {
            s = s * 1664525u + 1013904223u;
            v = ((s >> 8) / 16777216.0f) * 40.0f - 20.0f;
        }
        std::vector<float> batched(calls), scalar(calls);
        n3.evaluate3D(pts.data(), (size_t)calls, batched.data());
        for (int i = 0; i < 100; ++i) scalar[i] = n3.evaluate3D(&pts[3 * i]); // warm: first instance starts
        double t0 = now_us();
        for (int i = 0; i < calls; ++i) scalar[i] = n3.evaluate3D(&pts[3 * i]);
        const double mailbox = (now_us() - t0) / calls;
        size_t bad = 0;
        for (int i = 0; i < calls; ++i) bad += scalar[i] != batched[i];

        // the form this replaces: a batch of one = launch + stream synchronise
        wnhost::Scratch &sc = wnhost::Scratch::get();
        const int lcalls = calls < 2000 ? calls : 2000;
        t0 = now_us();
        for (int i = 0; i < lcalls; ++i) {
            for (int k = 0; k < 3; ++k) sc.in_host()[k] = pts[3 * i + k];
            wnhost::check(wn_eval3d_points(n3.tile(3), static_cast<const float *>(sc.in_dev()), 1,
                                           static_cast<float *>(sc.out_dev()), nullptr), "wn_eval3d_points");
            wnhost::check(wn_stream_sync(nullptr), "wn_stream_sync");
            bad += sc.out_host()[0] != batched[i];
        }
        const double launch_sync = (now_us() - t0) / lcalls;

        // after an idle gap the resident kernel has ended: the next call starts a new instance
        double idle_sum = 0;
        for (int r = 0; r < 10; ++r) {
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
            t0 = now_us();
            (void)n3.evaluate3D(&pts[0]);
            idle_sum += now_us() - t0;
        }
        t0 = now_us();
        float acc = 0;
        for (int i = 0; i < calls; ++i) acc += wt.value(0, 0, point3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2])).x();
        const double tex = (now_us() - t0) / calls;
        t0 = now_us();
        double dacc = 0;
        for (int i = 0; i < calls; ++i) dacc += per.noise((double)pts[3 * i], (double)pts[3 * i + 1], (double)pts[3 * i + 2]);
        const double pn = (now_us() - t0) / calls;
        // the default of the classes: the same sample on the host (bit-identical)
        t0 = now_us();
        const std::vector<float> &coef = n3.getNoiseCoefficients();
        for (int i = 0; i < calls; ++i) bad += wnhost_eval3d(coef.data(), n3.getTileSize(), &pts[3 * i]) != batched[i];
        const double host_eval = (now_us() - t0) / calls;
        unsigned long long served = 0, launches = 0;
        wn_scalar_stats(&served, &launches);
        std::printf("{\"calls\": %d, \"mailbox_us_per_call\": %.3f, \"launch_sync_us_per_call\": %.3f, \"after_idle_us\": %.1f, "
                    "\"texture_us_per_call\": %.3f, \"perlin_us_per_call\": %.3f, \"host_evaluator_us_per_call\": %.4f, \"mismatches\": %zu, "
                    "\"scalar_calls_served\": %llu, \"resident_kernel_instances\": %llu, \"checksum\": %.6g}\n",
                    calls, mailbox, launch_sync, idle_sum / 10, tex, pn, host_eval, bad, served, launches, (double)acc + dacc);
        return bad ? 1 : 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "scalar_latency: %s\n", e.what());
        return 2;
    }
}
