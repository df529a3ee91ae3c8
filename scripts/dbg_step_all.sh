#!/bin/bash
# runs the small-shape debug cases one after another; stops at the first failure (never continue on a faulted GPU)
set -o pipefail
for cfg in "$@"; do
  echo "== $cfg"
  AMD_LOG_LEVEL=1 timeout -k 5 120 python scripts/dbg_step.py $cfg 2>&1 | grep -v "amdgpu.ids" | tail -6
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "FAILED rc=$rc: $cfg"; exit 1; fi
done
echo ALL-OK
