#!/bin/bash
# the reference's UNMODIFIED main.cpp linked against host/ (make linkcheck): 1000x500, spp 100, wavelet 3D, octave 4
# usage: run_linked_raytracer.sh <outdir>        (WN_SCALAR_ON_DEVICE=1 in the environment: scalar calls through the mailbox)
out=$1; mkdir -p $out/rt && cd $out/rt && mkdir -p result_raytracing
exe=$GRAFT_REPO_ROOT/wavelet-noise-in-ray-tracing_amd/build/linkcheck/raytrace_main
s=$(date +%s%N)
printf '1\n4\n' | $exe > run.log 2>&1; rc=$?
e=$(date +%s%N)
echo "rc=$rc milliseconds=$(( (e - s) / 1000000 )) scalar_on_device=${WN_SCALAR_ON_DEVICE:-0}"
sha256sum result_raytracing/*.png
