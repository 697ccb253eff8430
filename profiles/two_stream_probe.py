#!/usr/bin/env python3
"""Probe: the 512^3 headline step launched back-to-back on one stream vs alternating on two streams with two output
buffers (the tail of one launch overlapping the head of the next).  python profiles/two_stream_probe.py"""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
N, K = 512, 100
noise = wn.WaveletNoise(128, 12345)
noise.generateNoiseTile3D()
outs = [torch.empty(N * N * N, dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
res = {}
for mode in ("one_stream", "two_streams", "one_stream_again"):
    def run(k):
        for i in range(k):
            if mode == "two_streams":
                with torch.cuda.stream(streams[i & 1]):
                    wn.wavelet_volume(noise, N, N, N, 0, N, 4, out=outs[i & 1])
            else:
                wn.wavelet_volume(noise, N, N, N, 0, N, 4, out=outs[0])
    run(10)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize()
    res[mode] = round((time.perf_counter() - t0) / K * 1e6, 2)
ref = wn.wavelet_volume(noise, N, N, N, 0, N, 4, exact=True)
torch.cuda.synchronize()
res["max_abs_err_buffers_vs_exact"] = [float((o.view(N, N, N) - ref).abs().max()) for o in outs]
print(json.dumps({"us_per_step": res}))
