#!/bin/bash
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ddp.py -m gpu -q --timeout=400 --no-header 2>&1 | grep -E "passed|failed|FAILED|^E  " | cut -c1-300 | head -40
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
line() { python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$1', d['value'], d['ms_per_step'])
except Exception:
    print('$1 FAILED:', t[-400:])"; }
{
for rep in 1 2; do
timeout 200 python bench.py --batch 128 $BA 2>&1 | tail -n 1 | line "B=128 single process"
timeout 200 python bench.py --batch 128 --force-ddp $BA 2>&1 | tail -n 1 | line "B=128 --force-ddp (torch transport, plan replay)"
timeout 200 python bench.py --batch 128 --force-ddp --transport rccl $BA 2>&1 | tail -n 1 | line "B=128 --force-ddp --transport rccl (plan replay)"
timeout 200 python bench.py --batch 128 --force-ddp --replay eager $BA 2>&1 | tail -n 1 | line "B=128 --force-ddp eager"
done
timeout 200 python bench.py --force-ddp $BA 2>&1 | tail -n 1 | line "B=1024 --force-ddp"
timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "B=1024 single process"
} | tee gpurun_out/r04_v21_ddp.txt
