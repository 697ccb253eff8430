"""WN_TUNE_ENV build only: in-kernel time stamps of the plane-pipeline multiband kernel (workgroup 0; a window wave, a collapse
wave, a store wave): stamp 0 = iteration start, 1 = window: after the x contraction / collapse: before the z collapse, 2 = before
the barrier."""
import ctypes, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["WN_MBP_DEBUG"] = "9"
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
out = torch.empty(512 ** 3, dtype=torch.float32, device="cuda")
for _ in range(3):
    wn.multiband_volume(noise, 512, 512, 512, 0, 512, out=out)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.getcwd(), "wavelet-noise-in-ray-tracing_amd", "libwnoise_hip.so"))
n = 3 * 256 * 3
buf = (ctypes.c_longlong * n)()
assert lib.wn_debug_mbp_stamps(buf, n) == 0
st = np.array(buf[:], dtype=np.int64).reshape(3, 256, 3)
t0 = st[0, 0, 0]
its = 16 * 8 + 2
for w, name in ((0, "window wave 1"), (1, "collapse wave 9"), (2, "store wave 12")):
    print(name, "(cycles since the window wave's first stamp; first 18 iterations)")
    for gp in range(18):
        print("  gp", gp, (st[w, gp] - t0).tolist())
d = np.diff(st[0, :its, 0])
print("iteration lengths (cycles):", d[:48].tolist())
print("mean iteration", float(d.mean()), "cycles; by plane index (gp & 7):", [round(float(d[k::8].mean())) for k in range(8)])
K = its - 2
wc = st[0, :K, 1] - st[0, :K, 0]; wb = st[0, 1:K + 1, 0] - st[0, :K, 2]
print("window wave: x contraction", float(wc.mean()), " barrier wait", float(wb.mean()), " by plane:", [round(float(wc[k::8].mean())) for k in range(8)])
cp = st[1, :K - 1, 1] - st[1, :K - 1, 0]; cz = st[1, :K - 1, 2] - st[1, :K - 1, 1]; cb = st[1, 1:K, 0] - st[1, :K - 1, 2]
print("collapse wave: issue / prep", [round(float(cp[k::8].mean())) for k in range(8)], " z collapse", float(cz.mean()), " barrier wait", float(cb.mean()))
s = st[2, 2:its, 2] - st[2, 2:its, 0]; sbw = st[2, 3:its, 0] - st[2, 2:its - 1, 2]
print("store wave: store_plane", float(s.mean()), " barrier wait", float(sbw.mean()))
