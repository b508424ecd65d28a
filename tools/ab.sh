# A/B of an environment switch on the default bench:  bash tools/ab.sh VAR [extra bench args]
VAR=$1; shift
for v in 0 1 0 1; do
  env $VAR=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['value'], d['ms_per_step'])"
done
