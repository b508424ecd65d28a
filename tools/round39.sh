python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear" 2>&1 | tail -3
python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "factor" 2>&1 | tail -3
python tools/kbench.py 2048 2>&1 | grep -A3 "1000x1000"
DVAE_GEMM_BIG=0 python tools/kbench.py 2048 2>&1 | grep -A3 "1000x1000"
bash tools/ab.sh DVAE_GEMM_BIG --loss factor
