#!/usr/bin/env python3
"""profiles/<tag>_bf16_sq.json from one rocprofv3 --pmc pass of SQ counters over the bf16 bench leg (scripts/profile_round2.sh):
MFMA-busy share of SIMD cycles, LDS bank-conflict share, instruction-wait share per step kernel.
usage: make_sq_json.py <counter_collection.csv> <out.json>"""
import csv, json, sys
from collections import defaultdict

KERNELS = {"step_bf16": "step_bf16", "fwd_dw_bf16": "fwd_dw_bf16", "fwd_ce_bf16": "fwd_ce_bf16", "dw_bf16": "dw_bf16", "head_step_kernel": "head_step"}


def _is(pat, name):
    """kernel-name match on a whole identifier (dw_bf16 must not match fwd_dw_bf16)"""
    import re
    return re.search(r"(^|[^_A-Za-z0-9])" + re.escape(pat), name) is not None
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    for k, pat in KERNELS.items():
        if _is(pat, r["Kernel_Name"]):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"_note": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT "
                "SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -- python3 bench.py --steps 20 --warmup 3 --repeats 3 --prime 20 --no-cpu-baseline --no-fp32-leg "
                "(own pass, no other trace domain); averages per dispatch.  SQ_VALU_MFMA_BUSY_CYCLES counts 32 per v_mfma_f32_32x32x16_bf16 "
                "(262144 MFMAs per GEMM launch = 8.39e6); mfma_busy_fraction = that / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); dispatches run "
                "slower under counter collection, so the fractions are lower bounds.", "kernels": {}}
for k, c in acc.items():
    a = {n: sum(v) / len(v) for n, v in c.items()}
    d = {}
    if a.get("GRBM_GUI_ACTIVE"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        d["active_cycles_per_xcd"] = round(cyc, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            d["mfma_busy_fraction_of_simd_cycles"] = round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
    if a.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction"] = round(a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"], 4)
    if a.get("SQ_WAVE_CYCLES"):
        d["wait_inst_fraction_of_wave_cycles"] = round(a.get("SQ_WAIT_INST_ANY", 0.0) / a["SQ_WAVE_CYCLES"], 4)
    out["kernels"][k] = {**{n: round(v, 1) for n, v in a.items()}, "dispatches": len(next(iter(c.values()))), "derived": d}
json.dump(out, open(sys.argv[2], "w"), indent=2)
print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))
