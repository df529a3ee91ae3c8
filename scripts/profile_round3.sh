#!/bin/bash
# usage (on the GPU box): scripts/profile_round3.sh <tag>   e.g. r03  -- everything the round's profiles/ files come from
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
o=gpurun_out
jl() { grep '^{' "$1" | tail -1; }
python bench.py > $o/${tag}_bench.log 2>&1 || { tail -5 $o/${tag}_bench.log; exit 1; }
jl $o/${tag}_bench.log > $o/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $o/${tag}_bench_driver_args.log 2>&1; jl $o/${tag}_bench_driver_args.log > $o/${tag}_bench_driver_args.json
python bench.py --force-dp-path --no-cpu-baseline --no-fp32-leg > $o/${tag}_force_dp.log 2>&1; jl $o/${tag}_force_dp.log > $o/${tag}_bench_force_dp_c_loop.json
UMLH_BF16_FUSE=0 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_bench_three_launches.json
UMLH_BF16_FUSE=0 UMLH_BF16_DW=0 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_bench_three_launches_old_dw_tile.json
UMLH_STEP_XCD=0 python bench.py --no-cpu-baseline --no-fp32-leg 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_bench_identity_home_tiles.json
python bench.py --no-cpu-baseline --no-fp32-leg --cfg3 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_bench_with_cfg3_key.json
UMLH_F32_FWD=1 python bench.py --precision fp32 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_bench_fp32_lds_staged_fwd.json
echo "bench legs done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$o/${tag}_kt -o kt -- python3 $root/bench.py --steps 200 --warmup 20 --no-cpu-baseline > $root/$o/${tag}_kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $root/$o/${tag}_pmc_f -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $root/$o/${tag}_pmc_w -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $root/$o/${tag}_pmc_sq -o pmc -- python3 $root/bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg > $root/$o/${tag}_pmc_sq.log 2>&1
cd $root
python scripts/summarize_rocprof.py $(dirname $(find $o/${tag}_kt -name "kt_kernel_stats.csv" | head -1)) kt $o/${tag}_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline (bf16 headline + fp32 parity leg; cfg2; MI355X; scripts/profile_round3.sh)" > /dev/null
python scripts/make_sq_json.py $(find $o/${tag}_pmc_sq -name "*counter_collection.csv" | head -1) $o/${tag}_bf16_sq.json > /dev/null
python scripts/make_pmc_json.py $(find $o/${tag}_pmc_f -name "*counter_collection.csv" | head -1) $(find $o/${tag}_pmc_w -name "*counter_collection.csv" | head -1) $o/${tag}_bf16_pmc.json > /dev/null
echo "rocprof legs done"
bash scripts/profile_cfg3.sh ${tag} > $o/${tag}_cfg3.log 2>&1 || true
python scripts/step_timeline.py 2>/dev/null > $o/${tag}_step_timeline.txt
python scripts/fwd32_stamps.py 2>/dev/null > $o/${tag}_fwd32_stamps.txt
python scripts/fwd_stamps.py 2>/dev/null > $o/${tag}_fwd_stamps.txt || true
echo "== LDS-DMA tile (stamps 3,4,5 = after chunks 0,1,2 of 8)" > $o/${tag}_dw_stamps.txt
python scripts/dw_stamps.py 2>/dev/null >> $o/${tag}_dw_stamps.txt
echo "== register-staged tile of rounds 1-2 (UMLH_BF16_DW=0)" >> $o/${tag}_dw_stamps.txt
UMLH_BF16_DW=0 python scripts/dw_stamps.py 2>/dev/null >> $o/${tag}_dw_stamps.txt
python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_n2_rehearsal.json
UMLH_DP_P2P=1 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg 2>/dev/null | grep '^{' | tail -1 > $o/${tag}_n2_rehearsal_p2p.json
echo "stamps + rehearsals done"
python scripts/small_step_timing.py 32 fp32 4 > $o/${tag}_micro_step_timing.txt 2>&1 || true
python scripts/small_step_timing.py 32 bf16 4 >> $o/${tag}_micro_step_timing.txt 2>&1 || true
python scripts/bench_multibench.py 2>/dev/null > $o/${tag}_multibench_step.txt || true
python scripts/bench_configs.py 2>/dev/null > $o/${tag}_other_configs.txt || true
python scripts/bench_farm.py --iters 1000 --workers 1 --grouped 2>/dev/null > $o/${tag}_farm.txt || true
python scripts/bench_farm.py --iters 4000 --workers 1 --grouped 2>/dev/null >> $o/${tag}_farm.txt || true
jl $o/${tag}_bench.json | cut -c1-400
