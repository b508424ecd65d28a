#!/bin/bash
# round 2, GPU visit 10: one stream vs two at small batches (DVAE_STREAMS=1|2), eager vs plan
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
bench() { timeout 300 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for st in 2 1; do
  for b in 64 128 256 512; do echo -n "btcvae 3ch B=$b streams=$st: "; DVAE_STREAMS=$st bench --batch $b; done
  echo -n "btcvae_dsprites (B=256 1ch) streams=$st: "; DVAE_STREAMS=$st bench --config btcvae_dsprites
  echo -n "factor_dsprites streams=$st: "; DVAE_STREAMS=$st bench --config factor_dsprites
  echo -n "btcvae_celeba B=1024 streams=$st: "; DVAE_STREAMS=$st bench --steps 60 --warmup 15
done 2>&1 | tee gpurun_out/streams_ab.txt
echo "== replay modes at B=128, one stream"
for m in eager plan graph; do echo -n "B=128 streams=1 replay=$m: "; DVAE_STREAMS=1 bench --batch 128 --replay $m; done | tee -a gpurun_out/streams_ab.txt
echo "== parity with one stream (step tests)"
DVAE_STREAMS=1 timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -q --timeout=600 --no-header -x > gpurun_out/pytest_1stream.log 2>&1; tail -n 2 gpurun_out/pytest_1stream.log
