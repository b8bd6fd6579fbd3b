#!/bin/bash
# SQ counters of a Cornell frame slice (800x800, 256 spp) with the shadow rays NOT traced inside the shading kernel
# (RT_AMD_FUSE_SHADOW=0; FUSE=1 keeps them fused): separates the shading code from the any-hit walk.  Three passes: VALU busy / lane
# utilisation (condensed by pmc_frame_json.py), LDS, memory instructions (raw sums per kernel by pmc_raw.py).
# Run on the GPU box from the repo root; results in gpurun_out/unfused/.
R=$PWD; OUT=$R/gpurun_out/unfused; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RT_AMD_FUSE_SHADOW=${FUSE:-0}
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- python3 $R/tests/prof_frame.py cornell 800 16 16 > $OUT/run1.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 $R/tests/prof_frame.py cornell 800 16 16 > $OUT/run2.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/pmc3 -- python3 $R/tests/prof_frame.py cornell 800 16 16 > $OUT/run3.log 2>&1 || exit 1
cd $R && python3 tests/pmc_frame_json.py $OUT/pmc1 cornell 800 800 16 16 > $OUT/r02_frame_pmc_cornell_unfused.json
python3 tests/pmc_raw.py $OUT/pmc2 > $OUT/r02_frame_pmc_cornell_unfused_lds.txt; python3 tests/pmc_raw.py $OUT/pmc3 > $OUT/r02_frame_pmc_cornell_unfused_vmem.txt
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
