#!/usr/bin/env python3
"""profiles/<tag>_bf16_pmc.json from the two rocprofv3 --pmc passes of scripts/profile_round.sh (FETCH_SIZE, WRITE_SIZE):
HBM bytes per launch of the three step kernels, corrected as MI355X_MICROARCH.md (HBM) prescribes for gfx950
(FETCH_SIZE counts 1/2 of a wide coalesced read; WRITE_SIZE is exact for 16-B streaming stores), next to the
algorithmic bytes of SURVEY 8(d).  usage: make_pmc_json.py <fetch.csv> <write.csv> <out.json>"""
import csv, json, sys
from collections import defaultdict

KERNELS = {"step_bf16": "step_bf16", "fwd_dw_bf16": "fwd_dw_bf16", "fwd_ce_bf16": "fwd_ce_bf16", "dw_bf16": "dw_bf16", "head_step_kernel": "head_step"}


def _is(pat, name):
    """kernel-name match on a whole identifier (dw_bf16 must not match fwd_dw_bf16)"""
    import re
    return re.search(r"(^|[^_A-Za-z0-9])" + re.escape(pat), name) is not None
ALG = {  # cfg2, per launch (design bytes of each kernel; the SURVEY 8(d) per-STEP figure is 20.7 MB: rows 8.4 MB + AdamW 12.3 MB)
    "step_bf16": (104_700_000, "the whole step in one launch: forward + dW + update rows below added"),
    "fwd_dw_bf16": (75_000_000, "forward + dW in one launch: the two rows below added (dZ^T written by the forward blocks and re-read by the dW blocks)"),
    "fwd_ce_bf16": (33_800_000, "8192 gathered bf16 rows x 1 KiB + W shadow 1 MiB x 8 XCD L2s + dZ^T 8192 x 1000 x 2 B written"),
    "dw_bf16": (41_200_000, "dZ^T 16.4 MB + 8192 feature rows 8.4 MB read once + 8 fp32 slabs x 2.05 MB written"),
    "head_step_kernel": (29_700_000, "8 slabs 16.4 MB + W,m,v 6.1 MB read; W,m,v 6.1 MB + bf16 shadow 1 MB written"),
}


def avg(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            for k, pat in KERNELS.items():
                if _is(pat, r["Kernel_Name"]):
                    acc[k].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


f, w = avg(sys.argv[1], "FETCH_SIZE"), avg(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2), python3 bench.py --steps 20 --warmup 3 "
                "--no-cpu-baseline --no-fp32-leg, MI355X, cfg2 bf16.  Counter unit = KiB per dispatch (average over the dispatches listed). "
                "Per MI355X_MICROARCH.md (HBM): FETCH_SIZE reads exactly 1/2 of the bytes of a wide coalesced streaming read on gfx950 -> "
                "doubled; WRITE_SIZE is exact for 16-B streaming stores.  hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  "
                "algorithmic_bytes are the kernel's DESIGN bytes (what this decomposition must move); SURVEY 8(d)'s algorithmic figure for the "
                "whole step is 20.7 MB, so the step moves step_traffic_over_algorithmic x that.",
       "kernels": {}}
tot = 0.0
for k in KERNELS:
    if k in f and k in w:
        b = (2 * f[k][0] + w[k][0]) * 1024
        tot += b
        out["kernels"][k] = {"FETCH_SIZE_KiB": round(f[k][0], 1), "WRITE_SIZE_KiB": round(w[k][0], 1), "dispatches": f[k][1],
                             "hbm_bytes_corrected": int(b), "algorithmic_bytes": ALG[k][0], "algorithmic_note": ALG[k][1]}
out["step_hbm_bytes_corrected"] = int(tot)
out["step_algorithmic_bytes_survey_8d"] = 20_700_000
out["step_traffic_over_algorithmic"] = round(tot / 20.7e6, 2)
json.dump(out, open(sys.argv[3], "w"), indent=2)
print(json.dumps(out, indent=1)[:1500])
