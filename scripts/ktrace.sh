#!/bin/bash
# usage (on the GPU box): scripts/ktrace.sh <tag> [bench args...]   -> gpurun_out/<tag>/ + kernel table on stdout
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -o kt -- python3 $root/bench.py --no-cpu-baseline --no-fp32-leg --steps 100 --warmup 10 "$@" > $root/gpurun_out/$tag.log 2>&1 || { tail -5 $root/gpurun_out/$tag.log; exit 1; }
cd $root && python scripts/kstats.py $(find gpurun_out/$tag -name "*kernel_stats.csv" | head -1) fwd_ce dw_bf16 dw_ head_step feistel iota gemm reduce finalize
