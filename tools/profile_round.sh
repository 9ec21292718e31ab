#!/bin/bash
# Round profile on the MI355X box: rocprofv3 kernel stats of the default bench command + the three PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA counters: each its own run, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), summarised into
# gpurun_out/<tag>/ for profiles/<tag>_*.   usage: tools/profile_round.sh r02_a [extra bench.py args]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py --no-other-configs "$@" > $OUT/bench_line.json 2> $OUT/bench_line.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $OUT/bench_line_profiled.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs "$@" > /dev/null 2> $OUT/mfma.err
mkdir -p $OUT/summary
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_bench_kernel_stats.csv
cp $OUT/bench_line.json $OUT/summary/${TAG}_bench_line.json
cp $OUT/bench_line_profiled.json $OUT/summary/${TAG}_bench_line_profiled.json
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1)); M=$(dirname $(find $OUT/mfma -name "*counter_collection.csv" | head -1))
python3 tools/pmc_summarize.py $F $W $OUT/summary/$TAG > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma_summarize.py $M $OUT/summary/$TAG > $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma      # raw traces stay on the box (gpurun_out/ merges back <= 64 MiB)
ls -la $OUT/summary; head -12 $OUT/summary/${TAG}_bench_kernel_stats.csv | cut -c1-150
