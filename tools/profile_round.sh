#!/bin/bash
# Collects the rocprofv3 runs the summaries under profiles/ are built from, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'   then   python tools/summarise_round.py r02 r02
# Every run is `bench.py --cpu-rows 0` (no CPU legs), i.e. the dense 8-bit NCC step, the triangulation,
# the FFT tile correlation batch and the 16-bit band, so all product kernels appear in every summary.
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), each under its own
# time limit; the chain stops at the first failure.
tag=$1
export TMPDIR=/tmp
O=gpurun_out
B="python3 bench.py --cpu-rows 0"
P="$B --second-peak off"      # counter passes: exactly one plain step, so per-step sums are sums over the run
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/${tag}_stats -o s --output-format csv -- $B --steps 5 --warmup 1 > $O/${tag}_stats.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/${tag}_fetch -o p --output-format csv -- $P --steps 1 --warmup 0 > $O/${tag}_fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/${tag}_write -o p --output-format csv -- $P --steps 1 --warmup 0 > $O/${tag}_write.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY -d $O/${tag}_sq -o p --output-format csv -- $P --steps 1 --warmup 1 > $O/${tag}_sq.log 2>&1 &&
(timeout -k 5 120 ./tools/ubench/valu_rates 8 > $O/${tag}_ubench.txt 2>&1; timeout -k 5 120 ./tools/ubench/valu_rates64 8 >> $O/${tag}_ubench.txt 2>&1; true) &&
echo done
