#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
TAG=${TAG:-r04_v39}
timeout 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_timed_config.py tests/test_gpu_bench_sizes.py tests/test_gpu_fused_core.py -m gpu -q --timeout=300 --no-header -x 2>&1 | tail -n 6 | cut -c1-300 | tee gpurun_out/${TAG}_pytest.txt
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
line() { python -c "
import sys, json
t = sys.stdin.read()
try:
    d = json.loads(t); print('$1', d['value'], d['ms_per_step'])
except Exception:
    print('$1 FAILED:', t[-300:])"; }
{
for rep in 1 2 3; do
  DVAE_DEBUG=1 DVAE_LATE_JOIN=0 timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "rep$rep B=1024 join after the decoder forward"
  timeout 200 python bench.py $BA 2>&1 | tail -n 1 | line "rep$rep B=1024 late join"
done
DVAE_DEBUG=1 DVAE_LATE_JOIN=0 timeout 200 python bench.py --batch 128 $BA 2>&1 | tail -n 1 | line "B=128 join after the decoder forward"
timeout 200 python bench.py --batch 128 $BA 2>&1 | tail -n 1 | line "B=128 late join"
DVAE_DEBUG=1 DVAE_LATE_JOIN=0 timeout 200 python bench.py --config btcvae_dsprites $BA 2>&1 | tail -n 1 | line "btcvae_dsprites join after the decoder forward"
timeout 200 python bench.py --config btcvae_dsprites $BA 2>&1 | tail -n 1 | line "btcvae_dsprites late join"
} | tee gpurun_out/${TAG}_late_join_ab.txt
