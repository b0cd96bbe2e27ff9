#!/bin/bash
# Profiles of the default bench.py command on the GPU box: kernel stats + PMC counter passes.
#   tools/profile_bench.sh <tag> <kernel name substring> [extra bench.py flags]      (run from the repo root, via gpurun)
#   e.g. tools/profile_bench.sh r04 k_sweep_      tools/profile_bench.sh r04_config3 k_joint --config 3
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the summaries the judge reads under
# gpurun_out/profiles_<tag>/ (copy them into profiles/ and commit).
set -o pipefail
TAG=$1; shift
KERN=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$(pwd)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 10 --warmup 2 --sub-configs none "$@" > $OUT/bench_stats.json 2> $OUT/bench_stats.err || exit 1
STEPS=3
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps $STEPS --warmup 0 --cpu-sample 0 --one-shot-calls 0 --sub-configs none "$@" > $OUT/pmc_fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps $STEPS --warmup 0 --cpu-sample 0 --one-shot-calls 0 --sub-configs none "$@" > $OUT/pmc_write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps $STEPS --warmup 0 --cpu-sample 0 --one-shot-calls 0 --sub-configs none "$@" > $OUT/pmc_sq.log 2>&1 || exit 4
python3 tools/pmc_summary.py $TAG $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq --steps=$STEPS --kernel=$KERN "--flags=$*" > $OUT/summary.json || exit 5
mkdir -p gpurun_out/profiles_$TAG
cp profiles/${TAG}_* gpurun_out/profiles_$TAG/
cp $OUT/bench_stats.json gpurun_out/profiles_$TAG/${TAG}_bench_n1.json
# raw trees are large: keep the summaries only
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq
