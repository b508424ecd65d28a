#!/bin/bash
# round 2, GPU visit 11: 3-layer FC chains (dvae_mlp3_fwd / dvae_mlp3_dgrad): parity + A/B (DVAE_MLP3=0|1)
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
echo "== pytest (mlp3 kernel, steps, bench sizes)"
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_step.py tests/test_gpu_bench_sizes.py -m gpu -q --timeout=900 --no-header -x -k "mlp3 or step" > gpurun_out/pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | head -20
grep -E "^E  " gpurun_out/pytest.log | cut -c1-300 | head -20
bench() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for v in 1 0 1 0; do echo -n "B=1024 DVAE_MLP3=$v: "; DVAE_MLP3=$v bench --steps 60 --warmup 15; done | tee gpurun_out/mlp3_ab.txt
for v in 1 0; do
  for b in 128 256; do echo -n "btcvae 3ch B=$b DVAE_MLP3=$v: "; DVAE_MLP3=$v bench --batch $b --steps 200 --warmup 30; done
  echo -n "btcvae_dsprites DVAE_MLP3=$v: "; DVAE_MLP3=$v bench --config btcvae_dsprites --steps 200 --warmup 30
  echo -n "factor_dsprites DVAE_MLP3=$v: "; DVAE_MLP3=$v bench --config factor_dsprites --steps 200 --warmup 30
done | tee -a gpurun_out/mlp3_ab.txt
echo "== deferred conv reductions at small batch (with mlp3)"
for b in 128 256; do for d in 0 1; do echo -n "B=$b DVAE_DEFER_REDUCE=$d: "; DVAE_DEFER_REDUCE=$d bench --batch $b --steps 200 --warmup 30; done; done | tee -a gpurun_out/mlp3_ab.txt
