timeout 900 python -m pytest tests/test_gpu_fused_core.py -m gpu -q --no-header -x -k "fc_chain" 2>&1 | tail -5
python tools/ab_chain_ends.py 128 256 1024 2048
