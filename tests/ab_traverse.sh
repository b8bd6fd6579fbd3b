#!/bin/bash
# A/B of library builds on the soup traversal (first batch only).  usage: ab_traverse.sh DIR1 DIR2 ...   ("" = the in-tree library)
for d in "$@"; do
  echo "== [$d]"
  if [ -n "$d" ]; then export RT_AMD_LIB_DIR=$d; else unset RT_AMD_LIB_DIR; fi
  timeout -k 10 200 python3 tests/prof_traverse.py 10000000 16777216 5 0 | python3 -c "import sys,json; r=json.load(sys.stdin); print(r['launch_ms'], 'ms', r['achieved'], 'GB/s', r['frac'])" || exit 1
done
