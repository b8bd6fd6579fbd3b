#!/bin/bash
# Times one frame with every kernel library under ab_libs/ and with the in-tree one; usage: ab_libs_probe.sh SCENE WIDTH SX  (GPU box, repo root)
R=$PWD
for d in "" $(ls -d $R/ab_libs/*/ 2>/dev/null); do
  echo "== ${d:-in-tree}"
  RT_AMD_LIB_DIR=${d%/} timeout -k 10 120 python -u tests/knob_sweep.py $1 $2 $3 RT_AMD_NOP 0,0 || exit 1
done
