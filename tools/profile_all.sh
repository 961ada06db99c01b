#!/bin/bash
# Collects every profile the summaries under profiles/ are built from, on the GPU box:
#   gpurun --timeout 900 -- 'bash tools/profile_all.sh r01e'   then   python tools/summarise_profiles.py r01e
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), each under its own time limit.
set -e
tag=$1
export TMPDIR=/tmp
O=gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats -o s --output-format csv -- python3 bench.py --steps 5 --warmup 1 --cpu-rows 0 > $O/${tag}_stats.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-rows 0 > $O/${tag}_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_write -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --cpu-rows 0 > $O/${tag}_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY -d $O/${tag}_sq -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --cpu-rows 0 > $O/${tag}_sq.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${tag}_xstats -o s --output-format csv -- python3 tools/bench_xcorr.py > $O/${tag}_xstats.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/${tag}_n16 -o s --output-format csv -- python3 tools/bench_ncc16.py 4096 2 > $O/${tag}_n16.log 2>&1
echo done
