"""development probe: per-launch time series of a dense lattice -- back-to-back, and with a synchronise between launches
usage: launch_series.py NX NY NZ [count]"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
nx, ny, nz = (int(v) for v in sys.argv[1:4])
count = int(sys.argv[4]) if len(sys.argv) > 4 else 400
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
out = torch.empty(nx * ny * nz, dtype=torch.float32, device="cuda")
launch = wn.wavelet_volume_launcher(noise, nx, nx, ny, 0, nz, 4, out)
def series(sync_between, n):
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    if sync_between:
        us = []
        for i in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); launch(); b.record(); b.synchronize(); us.append(a.elapsed_time(b) * 1e3)
        return us
    evs[0].record()
    for i in range(n):
        launch(); evs[i + 1].record()
        if i % 16 == 15: evs[i - 8].synchronize()   # keep the queue short but never empty
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(n)]
def show(name, us):
    r = lambda v: [round(x) for x in v]
    k = len(us)
    print(name, "launches 0-9", r(us[:10]), "| 20-29", r(us[20:30]), "| 50-59", r(us[50:60]), f"| {k//2}..", r(us[k//2:k//2+10]), "| last", r(us[-10:]), "| mean of last half", round(sum(us[k//2:]) / (k - k//2), 1))
show("back-to-back     ", series(False, count))
show("sync between     ", series(True, count))
show("back-to-back again", series(False, count))
