#!/bin/bash
# The two bench lines of the round, taken AFTER the counter summaries of this build are in profiles/ (bench.py quotes them by hash):
# plain, and under a kernel trace (whose per-kernel average must agree with the line's own HIP-event figure).  GPU box, repo root.
set -e
R=$PWD; OUT=$R/gpurun_out/r03b; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 500 python3 bench.py > $OUT/r03_bench_line.json 2> $OUT/bench_line.err
for s in helmet spheres2; do timeout -k 10 300 python3 bench.py --scene $s > $OUT/r03_bench_line_$s.json 2> $OUT/bench_line_$s.err; done
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_trace -- python3 $R/bench.py > $OUT/r03_bench_under_rocprof.json 2> $OUT/bench_trace.err
cp $(ls -t $OUT/bench_trace/*/*kernel_stats.csv | head -1) $OUT/r03_bench_trace_kernel_stats.csv
rm -rf $OUT/bench_trace
echo lines-done
