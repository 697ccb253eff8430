"""WN_TUNE_ENV build only (WN_HIP_LIBRARY=<...>/build/tune/libwnoise_hip.so): launch time and shader clock of every launch of a
dense lattice through the plane pipeline -- workgroup 0 stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its
start and end (WN_MBP_DEBUG=12).   usage: launch_clock.py NX NY NZ [count]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WN_MBP_DEBUG"] = "12"
import numpy as np, torch
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
lib = ctypes.CDLL(os.environ["WN_HIP_LIBRARY"])
nx, ny, nz = (int(v) for v in sys.argv[1:4])
count = int(sys.argv[4]) if len(sys.argv) > 4 else 300
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
out = torch.empty(nx * ny * nz, dtype=torch.float32, device="cuda")
launch = wn.wavelet_volume_launcher(noise, nx, nx, ny, 0, nz, 4, out)
buf = (ctypes.c_longlong * 4)()
rows = []
for i in range(count):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); b.record(); b.synchronize()
    assert lib.wn_debug_mbp_stamps(buf, 4) == 0
    cyc, ticks = buf[2] - buf[0], buf[3] - buf[1]
    rows.append((a.elapsed_time(b) * 1e3, cyc, ticks))
for i in list(range(0, 12)) + list(range(12, count, max(1, count // 40))):
    us, cyc, ticks = rows[i]
    print(f"launch {i:4d}: {us:7.1f} us   workgroup 0: {cyc:8d} cycles in {ticks / 100:7.1f} us  -> {cyc / ticks * 100:6.0f} MHz")
