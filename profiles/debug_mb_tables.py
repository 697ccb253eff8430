import ctypes, importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["WN_MBP_DEBUG"] = "20"
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
noise = wn.WaveletNoise(128, 12345); noise.generateNoiseTile3D()
wn.multiband_volume(noise, 512, 512, 8, 16, 20)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.getcwd(), "wavelet-noise-in-ray-tracing_amd", "libwnoise_hip.so"))
buf = (ctypes.c_longlong * 512)()
assert lib.wn_debug_mbp_stamps(buf, 512) == 0
f = np.frombuffer(bytes(buf), dtype=np.float32)
np.set_printoptions(precision=4, suppress=True, linewidth=200)
for b in range(5):
    print("band", b, "prep", f[640 + 4 * b: 644 + 4 * b], "bandc", f[680 + 4 * b: 684 + 4 * b])
    print(" wy[j][yi]:\n", f[40 * b: 40 * b + 40].reshape(5, 8))
    print(" wz[zi][slot]:\n", f[256 + 64 * b: 256 + 64 * b + 64].reshape(8, 8))
