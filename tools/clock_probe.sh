#!/bin/bash
# Shader clock while (a) the MFMA-only micro-benchmark, (b) the large GEMM, (c) the training step run:
# rocm-smi samples taken in the middle of ~3 s of back-to-back launches.
probe() { # label, command...
  local label=$1; shift
  "$@" > /dev/null 2>&1 &
  local pid=$!
  sleep ${PROBE_DELAY:-7}
  for i in 1 2 3; do
    echo "$label: $(rocm-smi --showclocks 2>/dev/null | grep -E 'sclk' | sed 's/.*sclk clock level: //') | $(rocm-smi --showpower 2>/dev/null | grep -iE 'power' | head -1 | sed 's/.*: //')"
    sleep 0.4
  done
  wait $pid
}
probe "idle" sleep 1
probe "gemm 2048x1000x1000 x100000" python tools/gemm_one.py 2048 1000 1000 100000
probe "btcvae step x6000" python bench.py --steps 6000 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-check
