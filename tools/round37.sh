python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "nchw or conv_fwd or convT_fwd" 2>&1 | tail -3
python -m pytest tests/test_gpu_step.py -x -q -m gpu 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -1
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200; done
python bench.py --steps 60 --warmup 10 --loss factor --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 60 --warmup 10 --batch 256 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 60 --warmup 10 --batch 64 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
