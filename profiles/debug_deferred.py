"""-DWN_TUNE_ENV build, WN_ROW_SLAB_FIRST_ONLY=1: which chunks of bench.py's texture_points stand-in does the plane-ordered pass leave
to the row-slab kernel?  (the marks stay in the output)"""
import sys, os, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WN_ROW_SLAB_FIRST_ONLY"] = "1"
import numpy as np, torch
wn = importlib.import_module("wavelet-noise-in-ray-tracing_amd")
m = 80_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
pts = torch.empty((m, 3), dtype=torch.float32, device="cuda")
pts[:, 0].uniform_(-10, 10, generator=g); pts[:, 1] = -0.5; pts[:, 2].uniform_(-10, 10, generator=g)
k = int(0.15 * m)
d = torch.randn((k, 3), device="cuda", generator=g)
pts[:k] = torch.tensor([1.0, 0.0, -1.75], device="cuda") + 0.5 * d / d.norm(dim=1, keepdim=True)
tex = wn.wavelet_texture(1.0, 4, True)
out = tex.grey(pts)
torch.cuda.synchronize()
first = out[: (m // 4096) * 4096].view(-1, 4096)[:, 0]
marked = torch.isnan(first)
print("chunks", first.numel(), "left to the slab kernel", int(marked.sum()), "kept", int((~marked).sum()))
kept = torch.nonzero(~marked).flatten()
print("kept chunk ids (first 20)", kept[:20].tolist(), "last", kept[-5:].tolist())
