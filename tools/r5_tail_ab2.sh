# second A/B of the encoder backward's tail: conv2's weight gradient on the main stream as well (the side stream keeps the FC
# layers' grouped launch only)
set -u
export TMPDIR=/tmp DVAE_DEBUG=1
mkdir -p gpurun_out
OUT=gpurun_out/tail_ab2.txt; : > $OUT
one() { local label=$1 v=$2; shift 2
  env DVAE_TAIL_MAIN=$v timeout 120 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in "$@" 2>&1 | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label tail_main=$v', d['value'], d['ms_per_step'])" | tee -a $OUT
}
for rep in 1 2; do
  for v in conv3,conv_64 conv2,conv3,conv_64; do
    one "rep$rep B=128" $v --batch 128
    one "rep$rep btcvae_dsprites" $v --config btcvae_dsprites
    one "rep$rep B=512 3ch" $v --batch 512
    one "rep$rep B=1024" $v
    one "rep$rep factor_dsprites" $v --config factor_dsprites
  done
done
