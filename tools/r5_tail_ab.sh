# A/B of the encoder backward's tail balance (DVAE_TAIL_MAIN, DVAE_DEBUG=1): which encoder weight gradients the main stream
# computes after conv1's.  Same box, alternating.   gpurun -- 'bash tools/r5_tail_ab.sh'
set -u
export TMPDIR=/tmp DVAE_DEBUG=1
mkdir -p gpurun_out
OUT=gpurun_out/tail_ab.txt; : > $OUT
one() { # label, variant, bench args...
  local label=$1 v=$2; shift 2
  env DVAE_TAIL_MAIN=$v timeout 120 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in "$@" 2>&1 | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label tail_main=$v', d['value'], d['ms_per_step'])" | tee -a $OUT
}
for rep in 1 2; do
  for v in conv3,conv_64 conv_64 conv3 none; do
    one "rep$rep B=128" $v --batch 128
    one "rep$rep btcvae_dsprites" $v --config btcvae_dsprites
  done
done
for v in conv3,conv_64 conv_64 none conv3,conv_64 conv_64; do
  one "B=256 3ch" $v --batch 256
  one "B=512 3ch" $v --batch 512
  one "factor_dsprites" $v --config factor_dsprites
done
