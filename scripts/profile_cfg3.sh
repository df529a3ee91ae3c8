#!/bin/bash
# usage (GPU box): scripts/profile_cfg3.sh <tag>  -- kernel traces of the cfg3 two-layer head (bf16 and fp32) -> gpurun_out/<tag>_cfg3_*_kernel_stats.md
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
o=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for prec in bf16 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_kt_cfg3_$prec -o kt -- python3 $root/scripts/bench_configs.py cfg3-$prec > $o/${tag}_kt_cfg3_$prec.log 2>&1
  (cd $root && python scripts/summarize_rocprof.py $(dirname $(find $o/${tag}_kt_cfg3_$prec -name "kt_kernel_stats.csv" | head -1)) kt $o/${tag}_cfg3_${prec}_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 scripts/bench_configs.py cfg3-$prec (d_v = 1024 -> d_t = 3200, C = 1000, 4096 + 4096 rows/step, 2-layer head, AdamW; 20 warm-up + timed steps; MI355X)" > /dev/null)
  grep -h "us_per_step" $o/${tag}_kt_cfg3_$prec.log | tail -1
done
(cd $root && python scripts/bench_configs.py cfg3-bf16 2>/dev/null | tail -1; python scripts/bench_configs.py cfg3-fp32 2>/dev/null | tail -1)
