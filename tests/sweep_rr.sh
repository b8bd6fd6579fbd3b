#!/bin/bash
# Sweep of the round-robin / contiguous work distribution knobs over the frame scenes.  Run on the GPU box from the repo root.
for s in cornell helmet spheres2; do
  for cfg in "0 0 0" "1 0 0" "0 1 0" "1 1 1" "1 0 1"; do
    set -- $cfg
    echo -n "$s extend=$1 shade=$2 shadow=$3: "
    RT_AMD_RR_EXTEND=$1 RT_AMD_RR_SHADE=$2 RT_AMD_RR_SHADOW=$3 timeout -k 10 120 python tests/prof_frame.py $s 800 16 16 0 3 | grep -o "Msamples/s.*" || exit 1
  done
done
