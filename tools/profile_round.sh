#!/bin/bash
# usage: tools/profile_round.sh <round tag>   (run on the GPU box through gpurun)
# Produces gpurun_out/prof_<tag>/: kernel-trace stats of the default bench command and the two
# separate PMC passes (FETCH_SIZE, WRITE_SIZE) the HBM-traffic figure needs.
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
rm -rf $O && mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --no-loop > $O/trace_bench.json 2> $O/trace.log || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_short -- python3 $R/bench.py --no-cpu-baseline --no-loop --no-rows --steps 20 --warmup 5 > $O/trace_short_bench.json 2> $O/trace_short.log || exit 2
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-loop --steps 5 --warmup 2 > $O/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-loop --steps 5 --warmup 2 > $O/pmc_write.log 2>&1 || exit 4
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --no-cpu-baseline --no-loop --steps 5 --warmup 2 > $O/pmc_sq.log 2>&1 || exit 5
cd $R && python3 tools/summarize_prof.py $O > $O/summary.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cp $(find $O/trace_short -name "*kernel_stats.csv" | head -1) $O/kernel_stats_short.csv
cat $O/trace_bench.json
