#!/bin/bash
# All clock domains rocm-smi reports (sclk, mclk, fclk, socclk) + power, sampled while ~6 s of training steps run, and the step time
# of that very run: one data point per box for the slow-box question of DESIGN.md section 5.1.
export TMPDIR=/tmp
python bench.py --steps 12000 --warmup 30 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in > /tmp/probe_bench.log 2>&1 &
pid=$!
sleep ${PROBE_DELAY:-8}
for i in 1 2 3 4; do
  echo "sample $i: $(rocm-smi --showclocks 2>/dev/null | grep -E 'clk' | sed 's/.*GPU\[0\][^:]*: //' | tr '\n' ';') $(rocm-smi --showpower 2>/dev/null | grep -iE 'power' | head -1 | sed 's/.*: //')"
  sleep 0.5
done
wait $pid
tail -n 1 /tmp/probe_bench.log | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('step', d['ms_per_step'], 'ms', d['value'], 'img/s', d['hip_event_ms_per_step']['segments'])"
