#!/bin/bash
# rocprofv3 evidence for one precision of the B=32 forward (run on the GPU box; round 3: the PIPELINED entry, forward_async):
# kernel stats of bench.py + separate
# counter passes (FETCH_SIZE, WRITE_SIZE, MFMA busy) over tools/prof_target.py.  Usage: tools/profile_forward.sh f32|bf16 TAG
set -u
P=$1; TAG=$2
export TMPDIR=/tmp
OUT=$(pwd)/gpurun_out/prof_${TAG}_${P}
mkdir -p $OUT
DT=""; [ "$P" = "bf16" ] && DT="--dtype bf16"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -- python bench.py $DT --steps 5 --warmup 2 --legs none > $OUT/stats_bench.json 2> $OUT/stats.log
python tools/prof_summary.py stats $OUT/stats "rocprofv3 --kernel-trace --stats -- python bench.py $DT --steps 5 --warmup 2 --legs none" > gpurun_out/${TAG}_${P}_bench_kernel_stats.txt 2>> $OUT/stats.log
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $C -f csv -d $OUT/pmc_$N -- python tools/prof_target.py ${P}p 32 4 > $OUT/pmc_$N.log 2>&1
  python tools/prof_summary.py counters $OUT/pmc_$N 4 "rocprofv3 --kernel-trace --pmc $C -- python tools/prof_target.py ${P}p 32 4" > gpurun_out/${TAG}_${P}_pmc_$N.json 2>> $OUT/pmc_$N.log
done
ls -la gpurun_out/${TAG}_${P}_*
