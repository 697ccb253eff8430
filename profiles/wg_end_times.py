"""WN_TUNE_ENV build only: start / end time of every workgroup of one launch of the plane pipeline (WN_MBP_DEBUG=13), by XCC.
usage: wg_end_times.py NX NY NZ [warm launches]"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WN_MBP_DEBUG"] = "13"
import numpy as np, torch
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
lib = ctypes.CDLL(os.environ["WN_HIP_LIBRARY"])
nx, ny, nz = (int(v) for v in sys.argv[1:4])
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 200
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
out = torch.empty(nx * ny * nz, dtype=torch.float32, device="cuda")
launch = wn.wavelet_volume_launcher(noise, nx, nx, ny, 0, nz, 4, out)
for rep in range(3):
    for _ in range(warm): launch()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); launch(); b.record(); b.synchronize()
    buf = (ctypes.c_longlong * 768)()
    assert lib.wn_debug_mbp_stamps(buf, 768) == 0
    st = np.array(buf[:], dtype=np.int64).reshape(256, 3)
    t0 = st[:, 0].min()
    start, end, xcc = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2]
    print(f"launch {a.elapsed_time(b) * 1e3:.1f} us; workgroup ends: min {end.min():.1f} median {np.median(end):.1f} max {end.max():.1f} us; starts within {start.max():.1f} us")
    for x in range(8):
        m = xcc == x
        print(f"  XCC {x}: {int(m.sum())} workgroups, end mean {end[m].mean():.1f} min {end[m].min():.1f} max {end[m].max():.1f}; blocks {np.nonzero(m)[0][:6].tolist()}")
    order = np.argsort(end)
    print("  earliest blocks", order[:8].tolist(), "latest blocks", order[-8:].tolist())
    q = end.reshape(4, 64).mean(axis=1)
    print("  mean end by quarter of the grid (= brick column bx at 2048 wide):", [round(float(v), 1) for v in q])
