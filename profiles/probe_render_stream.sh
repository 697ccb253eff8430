#!/bin/bash
# tools/render --time-kernel: the texture kernel alone on the renderer's real hit-point stream, product build vs the
# plane-ordered kernel (-DWN_TUNE_ENV build with WN_NO_ROW_SLAB=1)
R=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/tools/render
T=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/tune
cd /tmp
for spp in 16 64; do
  echo "== spp $spp, product build"; $R --width 1920 --height 1080 --spp $spp --noise 1 --time-kernel --band-lines 135 --out /tmp/r.png 2>&1 | grep "render:"
  echo "== spp $spp, plane-ordered kernel only (tune build, WN_NO_ROW_SLAB=1)"; LD_LIBRARY_PATH=$T WN_NO_ROW_SLAB=1 $R --width 1920 --height 1080 --spp $spp --noise 1 --time-kernel --band-lines 135 --out /tmp/r2.png 2>&1 | grep "render:"
done
