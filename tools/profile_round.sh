#!/bin/bash
# The commands behind profiles/rNN_*: run on the GPU box from the repo root (e.g. `gpurun -- bash tools/profile_round.sh r06`),
# everything lands in gpurun_out/; copy what is to be kept into profiles/.
#   rNN_gputest_final.txt      python -m pytest tests -m gpu
#   rNN_bench_kernel_stats.txt rocprofv3 --kernel-trace --stats over a short bench.py run (per-kernel average durations)
#   pmc_dominant.json          separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_head.py -> HBM bytes of the dominant kernel
#   rNN_pmc_mfma_busy.txt      --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over the same command -> matrix-pipe busy fraction
#   rNN_bench_line.json        the driver's command, python bench.py --steps 20 --warmup 5
# (PMC passes carry --kernel-trace only: gpurun refuses counter collection combined with the other trace domains)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest_final.txt 2>&1; tail -2 gpurun_out/${TAG}_gputest_final.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -1 gpurun_out/${TAG}_smoke.txt
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-configs > $R/gpurun_out/${TAG}_prof_bench.log 2>&1
cd $R; DB=$(find gpurun_out/prof_$TAG -name "*_results.db" | head -1); python tools/rocpd_summary.py $DB > gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o p --output-format csv -- python $R/tools/pmc_head.py > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o p --output-format csv -- python $R/tools/pmc_head.py > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_busy -o p --output-format csv -- python $R/tools/pmc_head.py > $R/gpurun_out/${TAG}_pmc_busy.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --commit "$(cat tools/_run/HEAD 2>/dev/null || echo $TAG)" > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
cp profiles/pmc_dominant.json gpurun_out/pmc_dominant.json
python tools/pmc_busy.py gpurun_out/pmc_busy > gpurun_out/${TAG}_pmc_mfma_busy.txt 2>&1
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_busy
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
tail -c 200 gpurun_out/${TAG}_bench_line.err
