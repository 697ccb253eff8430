"""the plane pipeline at 512^3 by number of bands (top band = octave 4): launch time over 100 back-to-back launches after 100 warm ones"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
out = torch.empty(512 ** 3, dtype=torch.float32, device="cuda")
for nb in (1, 2, 3, 4, 5):
    f = lambda: wn.multiband_volume(noise, 512, 512, 512, 0, 512, -16.0, 5 - nb, nb, [1.0] * nb, out=out)
    for _ in range(100): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100): f()
    b.record(); b.synchronize()
    print(f"{nb} band(s): {a.elapsed_time(b) * 10:.1f} us per launch")
