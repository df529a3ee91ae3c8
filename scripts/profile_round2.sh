#!/bin/bash
# usage (on the GPU box): scripts/profile_round2.sh <tag>   e.g. r02  -- everything the round's profiles/ files come from
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
o=gpurun_out
python bench.py > $o/${tag}_bench.log 2>&1 || { tail -5 $o/${tag}_bench.log; exit 1; }
tail -1 $o/${tag}_bench.log > $o/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/${tag}_bench_driver_args.log 2>&1; tail -1 $o/${tag}_bench_driver_args.log > $o/${tag}_bench_driver_args.json
python bench.py --force-dp-path --no-cpu-baseline --no-fp32-leg > $o/${tag}_force_dp.log 2>&1; tail -1 $o/${tag}_force_dp.log > $o/${tag}_bench_force_dp_c_loop.json
python bench.py --force-dp-path --dp-host-loop --no-cpu-baseline --no-fp32-leg > $o/${tag}_force_dp_host.log 2>&1; tail -1 $o/${tag}_force_dp_host.log > $o/${tag}_bench_force_dp_host_loop.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/${tag}_kt -o kt -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $root/$o/${tag}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/$o/${tag}_pmc_f -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/$o/${tag}_pmc_w -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $root/$o/${tag}_pmc_sq -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_sq.log 2>&1
cd $root
python scripts/summarize_rocprof.py $(dirname $(find $o/${tag}_kt -name "kt_kernel_stats.csv" | head -1)) kt $o/${tag}_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline (bf16 headline + fp32 parity leg; cfg2; MI355X; scripts/profile_round2.sh)" > /dev/null
python scripts/make_sq_json.py $(find $o/${tag}_pmc_sq -name "*counter_collection.csv" | head -1) $o/${tag}_bf16_sq.json > /dev/null
python scripts/make_pmc_json.py $(find $o/${tag}_pmc_f -name "*counter_collection.csv" | head -1) $(find $o/${tag}_pmc_w -name "*counter_collection.csv" | head -1) $o/${tag}_bf16_pmc.json > /dev/null
# micro step: per-phase stamps, step time, sweeps
python scripts/micro_stamps.py 512 100 32 4096 > $o/${tag}_micro_stamps.txt 2>&1
python scripts/small_step_timing.py 32 fp32 4 > $o/${tag}_micro_step_timing.txt 2>&1
UMLH_MICRO=0 python scripts/small_step_timing.py 32 fp32 4 >> $o/${tag}_micro_step_timing.txt 2>&1
python scripts/small_step_timing.py 32 bf16 4 >> $o/${tag}_micro_step_timing.txt 2>&1
UMLH_MICRO=0 python scripts/small_step_timing.py 32 bf16 4 >> $o/${tag}_micro_step_timing.txt 2>&1
python scripts/bench_farm.py --iters 1000 --workers 1 --grouped > $o/${tag}_farm.txt 2>&1
python scripts/bench_farm.py --iters 4000 --workers 1 --grouped >> $o/${tag}_farm.txt 2>&1
python scripts/bench_farm.py --iters 4000 --workers 1 --grouped --precision bf16 >> $o/${tag}_farm.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/${tag}_kt_micro -o kt -- python3 $root/scripts/small_step_timing.py 32 fp32 4 > $root/$o/${tag}_kt_micro.log 2>&1
cd $root
python scripts/summarize_rocprof.py $(dirname $(find $o/${tag}_kt_micro -name "kt_kernel_stats.csv" | head -1)) kt $o/${tag}_micro_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats -- python3 scripts/small_step_timing.py 32 fp32 4 (cfg1 shape d=512 C=100, batch 32+32, 4 x 100 steps through umlh_train_steps = 4 persistent launches of 100 steps each)" > /dev/null
# MultiBench alternation step: step time (z = 40 / 300, HIP encoder vs torch.nn ops) and the kernel trace at z = 40
python scripts/bench_multibench.py 2>/dev/null > $o/${tag}_multibench_step.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/${tag}_kt_mb -o kt -- python3 $root/scripts/bench_multibench.py 40 100 > $root/$o/${tag}_kt_mb.log 2>&1
cd $root
python scripts/summarize_rocprof.py $(dirname $(find $o/${tag}_kt_mb -name "kt_kernel_stats.csv" | head -1)) kt $o/${tag}_multibench_kernel_stats_after.md "Command: rocprofv3 --kernel-trace --stats -- python3 scripts/bench_multibench.py 40 100 (z = 40, T = 50, B = 32, train mode; >= 1 s warm-up + 100 timed steps; 2 multi_opt_kernel launches per step)" > /dev/null
# wide heads on the micro path (6 and 8 chunks: optimizer moments in memory), the opt-in 2-D forward, write-through stores A/B
: > $o/${tag}_micro_wide_heads.txt
for a in "32 bf16 3 768 100" "32 fp32 3 768 100" "32 bf16 3 768 1000" "32 bf16 3 1024 100" "32 fp32 3 1024 100" "32 bf16 3 512 100"; do
  echo "== scripts/small_step_timing.py $a" >> $o/${tag}_micro_wide_heads.txt
  python scripts/small_step_timing.py $a 2>&1 | grep "rep " >> $o/${tag}_micro_wide_heads.txt
done
echo "== UMLH_BF16_FWD2D=1 (fwd_ce_bf16_q)" > $o/${tag}_fwd2d_stamps.txt
UMLH_BF16_FWD2D=1 python scripts/fwd_stamps.py >> $o/${tag}_fwd2d_stamps.txt 2>&1
echo "== default (fwd_ce_bf16)" >> $o/${tag}_fwd2d_stamps.txt
UMLH_BF16_FWD2D=0 python scripts/fwd_stamps.py >> $o/${tag}_fwd2d_stamps.txt 2>&1
UMLH_BF16_FWD2D=1 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 > $o/${tag}_bench_fwd2d_on.json
UMLH_WT=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $o/${tag}_bench_plain_stores.json
UMLH_BF16_FUSE=0 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 > $o/${tag}_bench_two_launches.json
UMLH_BF16_FUSE=1 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | tail -1 > $o/${tag}_bench_fwd_dw_one_launch.json
python scripts/step_timeline.py 2>/dev/null > $o/${tag}_step_timeline.txt
tail -2 $o/${tag}_bench.json | cut -c1-600
