# sharded-path tests + the shard legs of the bench line (one rank of eight, 128 images per GPU)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_ddp.py -q --timeout=200 --no-header > gpurun_out/ddp_pytest.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/ddp_pytest.log
tail -n 6 gpurun_out/ddp_pytest.log | cut -c1-300
for r in 1 2; do
timeout 200 python bench.py --shard-legs --steps 200 --warmup 10 2>&1 | tail -2 | cut -c1-1500 | tee -a gpurun_out/shard_legs.txt
done
