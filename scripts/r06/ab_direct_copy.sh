#!/bin/bash
# A/B: host-pointer batches through the pinned staging ring (1 / 2 stagers per device) against the runtime's own pageable-memory copies
cd $GRAFT_REPO_ROOT
for cfg in "TA_MULTI_STAGERS=1" "TA_MULTI_STAGERS=2" "TA_MULTI_STAGERS=1 TA_MULTI_DIRECT_COPY=1" "TA_MULTI_STAGERS=2 TA_MULTI_DIRECT_COPY=1"; do
  echo "$cfg: $(env TA_TUNING=1 $cfg python bench.py --single-process --gpus 1 --scaling strong --steps 10 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("e2e ms", round(d["end_to_end_ms"],2))')"
done
