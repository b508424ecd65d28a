export TMPDIR=/tmp
R=$PWD
for m in eager plan; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$m -o prof -- python $R/bench.py --loss factor --steps 10 --warmup 3 --replay $m --no-cpu-baseline --no-roofline --no-parity-check > $R/gpurun_out/prof_$m.log 2>&1)
  grep metric gpurun_out/prof_$m.log | cut -c1-200
done
cd tools && python prof_compare.py ../gpurun_out/prof_eager/prof_results.db ../gpurun_out/prof_plan/prof_results.db | head -30
