#!/bin/bash
# A/B timing of one frame under different knob settings.  usage: ab_frame.sh SCENE WIDTH SX SY "ENV1=.. ENV2=.." "ENV..." ...
# (run on the GPU box from the repo root; an empty string "" is the default configuration)
scene=$1; w=$2; sx=$3; sy=$4; shift 4
for cfg in "$@"; do
  echo "== [$cfg]"
  env $cfg timeout -k 10 200 python3 tests/prof_frame.py $scene $w $sx $sy 0 3 || exit 1
done
