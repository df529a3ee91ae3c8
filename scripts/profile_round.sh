#!/bin/bash
# usage (on the GPU box): scripts/profile_round.sh <tag>   e.g. r01
# 1. default bench line (bf16 headline + fp32 parity leg + cpu baseline)       -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command (no cpu baseline)     -> gpurun_out/<tag>_kt/
# 3. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) on the bf16 leg          -> gpurun_out/<tag>_pmc_{f,w}/
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
python bench.py > gpurun_out/${tag}_bench.log 2>&1 || { tail -5 gpurun_out/${tag}_bench.log; exit 1; }
tail -1 gpurun_out/${tag}_bench.log > gpurun_out/${tag}_bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_kt -o kt -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $root/gpurun_out/${tag}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_f -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg > $root/gpurun_out/${tag}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_pmc_w -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-fp32-leg > $root/gpurun_out/${tag}_pmc_w.log 2>&1
cd $root
cat gpurun_out/${tag}_bench.json
python scripts/pmc_summary.py $(find gpurun_out/${tag}_pmc_f -name "*counter_collection.csv" | head -1) fwd_ce_bf16 dw_bf16 head_step
python scripts/pmc_summary.py $(find gpurun_out/${tag}_pmc_w -name "*counter_collection.csv" | head -1) fwd_ce_bf16 dw_bf16 head_step
