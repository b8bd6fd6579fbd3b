#!/bin/bash
# Knob sweep on the sample shares that one of 1 / 2 / 4 / 8 ranks renders (tests/scale_probe.py).  usage: sweep_share.sh "ENV=.. ENV=.." ...
for cfg in "$@"; do
  echo "== [$cfg]"
  env $cfg timeout -k 10 200 python3 tests/scale_probe.py || exit 1
done
