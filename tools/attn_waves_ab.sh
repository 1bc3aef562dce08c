#!/bin/bash
# A/B of 8 vs 7 waves per attention workgroup (FITCLIP_ATTN_7WAVES=1): bf16 attention inside bench.py, fp32 attention
# backward inside tools/train_bench.py.  Run on the GPU box from the repo root.
set -e
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "value", d["value"], "attention", d["time_split"]["attention"])'
python bench.py --precision bf16 --no-cpu-baseline --no-train-leg | python -c "$pick" "bf16 8 waves"
FITCLIP_ATTN_7WAVES=1 python bench.py --precision bf16 --no-cpu-baseline --no-train-leg | python -c "$pick" "bf16 7 waves"
pick2='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["phases_ms"])'
python tools/train_bench.py --lr 3e-7 | python -c "$pick2" "train, backward 8 waves"
FITCLIP_ATTN_7WAVES=1 python tools/train_bench.py --lr 3e-7 | python -c "$pick2" "train, backward 7 waves"
