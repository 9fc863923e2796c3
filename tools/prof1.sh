#!/bin/bash
# usage: tools/prof1.sh <tag> [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $O/trace.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/pmc_sq1.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES_EQ_64 SQ_THREAD_CYCLES_VALU --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/pmc_sq2.log 2>&1 || echo "sq2 failed"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/pmc_fetch.log 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 "$@" > $O/pmc_write.log 2>&1 || exit 4
find $O -name "*.csv" | head -30
