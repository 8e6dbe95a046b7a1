#!/bin/bash
# Measurement set of a round, run on the GPU box: tools/evidence.sh <tag>   (writes gpurun_out/<tag>_*)
#   1. rocprofv3 --kernel-trace --stats over one timed step of the default bench command
#   2. HBM traffic of the dominant kernel (FETCH_SIZE / WRITE_SIZE passes, tools/pmc_bench.sh)
#   3. one SQ counter pass (MFMA busy, LDS conflicts) over a 2-step schedule of the same command
#   4. the 64x64 / batch 64 and the makeup-transfer workloads, one step each
set -u
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
echo "== kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_default_under_rocprof.jsonl 2> $O/${TAG}_prof.err || { echo "stats pass failed"; tail -3 $O/${TAG}_prof.err; exit 1; }
cp $(find $O/${TAG}_prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_default_kernel_stats.csv
rm -rf $O/${TAG}_prof          # the per-dispatch trace is hundreds of MB; the summary is what is kept
echo "== traffic"
bash $R/tools/pmc_bench.sh $O/${TAG}_pmc_bench > $O/${TAG}_pmc_halo_traffic.json || exit 1
rm -rf $O/${TAG}_pmc_bench
echo "== sq counters"
CMD="python3 $R/bench.py --num-of-step 2 --no-cpu-baseline"
timeout -s KILL 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/${TAG}_pmc_sq -- $CMD > $O/${TAG}_pmc_sq.log 2>&1 || { echo "sq pass failed"; tail -3 $O/${TAG}_pmc_sq.log; exit 1; }
python3 $R/tools/pmc_sq_summary.py $O/${TAG}_pmc_sq "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA -- python3 bench.py --num-of-step 2 --no-cpu-baseline" > $O/${TAG}_pmc_sq_bench.json
rm -rf $O/${TAG}_pmc_sq
cd $R
echo "== 64x64 B=64"
timeout -k 10 300 python3 bench.py --config configs/model_64_64.yaml --batch 64 --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_64x64_B64.jsonl 2> $O/${TAG}_bench_64.err || { echo "64x64 bench failed"; tail -3 $O/${TAG}_bench_64.err; exit 1; }
echo "== transfer"
timeout -k 10 300 python3 bench.py --workload transfer --steps 1 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_transfer_S180.jsonl 2> $O/${TAG}_bench_transfer.err || { echo "transfer bench failed"; tail -3 $O/${TAG}_bench_transfer.err; exit 1; }
echo ok
