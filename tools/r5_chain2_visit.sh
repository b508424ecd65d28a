# FactorVAE: the discriminator's second input-gradient chain on the engine's third stream (DVAE_DISC_CHAIN2_AUX, DVAE_DEBUG=1): A/B on
# the two factor workloads, same box, alternating; then the factor tests on the shipped setting
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
OUT=gpurun_out/disc_chain2_ab.txt; : > $OUT
one() { local label=$1 v=$2; shift 2
  env DVAE_DEBUG=1 DVAE_DISC_CHAIN2_AUX=$v timeout 120 python bench.py --steps 150 --warmup 20 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in "$@" 2>&1 | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label second chain on its own stream=$v', d['value'], d['ms_per_step'])" | tee -a $OUT
}
for rep in 1 2; do
  for v in 1 0; do
    one "rep$rep factor_dsprites" $v --config factor_dsprites
    one "rep$rep factor_celeba" $v --config factor_celeba
    one "rep$rep factor 64x64x3 tensor 512" $v --config factor_celeba --batch 512
  done
done
timeout 500 python -m pytest tests/test_gpu_step.py tests/test_gpu_bench_sizes.py tests/test_gpu_timed_config.py tests/test_gpu_wide_latent.py tests/test_gpu_discriminator.py -q --timeout=200 --no-header -k "factor or discriminator" > gpurun_out/chain2_pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/chain2_pytest.log
tail -n 6 gpurun_out/chain2_pytest.log | cut -c1-300
