L=disentangling-vae_amd/lib
DVAE_HIP_LIB=$L/libdvae_hip_upnonop.so timeout 900 python -m pytest tests/test_gpu_mask_bits.py tests/test_gpu_kernels.py -m gpu -q --no-header -x 2>&1 | tail -3
for rep in 1 2; do
  python tools/ab_kernels.py 1024 up16,up16m
  DVAE_HIP_LIB=$L/libdvae_hip_upnonop.so python tools/ab_kernels.py 1024 up16
done
BA="--steps 150 --warmup 25 --no-cpu-baseline --no-roofline --no-parity-check --no-extra-configs --no-drop-in"
for rep in 1 2 3; do
 for v in "" upnonop utm1; do
  if [ -z "$v" ]; then unset DVAE_HIP_LIB; else export DVAE_HIP_LIB=$L/libdvae_hip_$v.so; fi
  for b in 1024 128; do python bench.py --batch $b $BA 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('step B=$b lib=${v:-default}', d['ms_per_step'])"; done
 done
done
unset DVAE_HIP_LIB
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-parity-check --no-extra-configs --no-drop-in 2>/dev/null | tail -n 1 > gpurun_out/r06_v9_bench_roofline.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_v9_bench_roofline.json"))
print(d["ms_per_step"]); r=d["roofline"]; print(r["kernel"], r["us_per_launch"], r["frac"], r.get("in_step_us"))
for r in d["roofline_kernels"]: print("  ", r["kernel"], r["us_per_launch"], r["frac"])
PY
