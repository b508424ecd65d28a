export TMPDIR=/tmp
R=$PWD
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k linear 2>&1 | tail -3
python -m pytest tests/test_gpu_step.py -x -q -m gpu -k "fused_step or factor_step or golden" 2>&1 | tail -3
python tools/kbench.py 1024 2>&1 | grep -i "linear wgrad"
for i in 1 2; do python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200; done
python bench.py --steps 60 --warmup 10 --loss factor --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
python bench.py --steps 60 --warmup 10 --batch 256 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-200
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof.log 2>&1)
cd tools && python timeline.py ../gpurun_out/prof/prof_results.db > ../gpurun_out/timeline.txt; tail -2 ../gpurun_out/timeline.txt
