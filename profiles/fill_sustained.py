"""development probe: does the store ceiling (torch zero_ of the output buffer) hold over seconds?  segments of 10 fills"""
import sys, torch
nbytes = int(float(sys.argv[1]) * 2**20) if len(sys.argv) > 1 else 4096 * 2**20
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
buf = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
evs = [torch.cuda.Event(enable_timing=True)]
evs[0].record()
import time
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(10):
        buf.zero_()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    if len(evs) % 8 == 0:
        evs[-1].synchronize()
torch.cuda.synchronize()
us = [evs[i].elapsed_time(evs[i + 1]) * 100 for i in range(len(evs) - 1)]
gb = [nbytes / u / 1e3 for u in us]
print(f"{nbytes/2**20:.0f} MiB fill, {len(us)} segments of 10: first 5 GB/s", [round(x) for x in gb[:5]], "last 5", [round(x) for x in gb[-5:]],
      "min", round(min(gb)), "max", round(max(gb)), "mean", round(sum(gb) / len(gb)))
