#!/bin/bash
# Round profile on the MI355X box: rocprofv3 kernel stats of a bench command + the three PMC passes (FETCH_SIZE, WRITE_SIZE,
# MFMA counters: each its own run, --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), summarised into
# gpurun_out/<tag>/summary/ for profiles/<tag>_*.
#   usage: tools/profile_round.sh r04_a                     (the default = headline workload; the full default run is the first leg)
#          tools/profile_round.sh r04_a_config5 --config 5  (another BASELINE configuration: its summaries carry its workload key)
# The program sits directly after `--` (python3 bench.py ...): no env / bash -c hop under rocprofv3.
# Round 6: the profiled passes run --streams 1 -- the pass bench.py itself takes its `roofline` from (under the two-stream batch rotation that `value` is quoted on, a
# kernel's duration includes the other stream's work and is not a roofline input).
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT/summary
KEY=$(python3 bench.py --print-workload-key "$@")
if [ $# -eq 0 ]; then      # headline: the complete default run (compact lines of configs 2 / 3 / 5, then the headline line) as the driver sees it
  VS_BENCH_DETAILS=$OUT/summary/${TAG}_bench_details.json python3 bench.py > $OUT/bench_stdout.txt 2> $OUT/bench_line.err
else
  VS_BENCH_DETAILS=$OUT/summary/${TAG}_bench_details.json python3 bench.py --no-other-configs "$@" > $OUT/bench_stdout.txt 2> $OUT/bench_line.err
fi
cp $OUT/bench_stdout.txt $OUT/summary/${TAG}_bench_stdout.txt
tail -n 1 $OUT/bench_stdout.txt > $OUT/summary/${TAG}_bench_line.json
export VS_BENCH_DETAILS=/tmp/bench_details_scratch.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 "$@" > $OUT/bench_line_profiled.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --streams 1 "$@" > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --streams 1 "$@" > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --streams 1 "$@" > /dev/null 2> $OUT/mfma.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/summary/${TAG}_bench_kernel_stats.csv
tail -n 1 $OUT/bench_line_profiled.json > $OUT/summary/${TAG}_bench_line_profiled.json
F=$(dirname $(find $OUT/fetch -name "*counter_collection.csv" | head -1)); W=$(dirname $(find $OUT/write -name "*counter_collection.csv" | head -1)); M=$(dirname $(find $OUT/mfma -name "*counter_collection.csv" | head -1))
# steps the PMC passes ran (--warmup 1 --steps 1; the training configuration counts its launches on one more step)
NSTEPS=2; case " $* " in *" --config 3 "*) NSTEPS=3;; esac
python3 tools/pmc_summarize.py $F $W $OUT/summary/$TAG $KEY $NSTEPS > $OUT/pmc_traffic.txt 2>&1
python3 tools/pmc_mfma_summarize.py $M $OUT/summary/$TAG $KEY > $OUT/pmc_mfma.txt 2>&1
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma      # raw traces stay on the box (gpurun_out/ merges back <= 64 MiB)
python3 -c "import json,sys; sys.path.insert(0,'.'); from visinger_amd.csrc import build; json.dump({'tag':'$TAG','vs_source_hash':build.source_hash(),'args':'$*','kernel_stats':'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --streams 1 $*'}, open('$OUT/summary/${TAG}_meta.json','w'), indent=1)"
ls -la $OUT/summary; head -12 $OUT/summary/${TAG}_bench_kernel_stats.csv | cut -c1-150
