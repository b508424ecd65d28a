#!/bin/bash
# round 2, GPU visit 12: which encoder weight gradients the main stream takes at the tail (DVAE_TAIL_MAIN); Adam over one flat tensor
set -u
export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"; mkdir -p gpurun_out
bench() { timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-roofline --no-parity-check "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['hip_event_ms_per_step']['median'])"; }
for t in "" conv3 conv_64 "conv3,conv_64" conv2 "" conv3; do echo -n "DVAE_TAIL_MAIN='$t': "; DVAE_TAIL_MAIN="$t" bench; done | tee gpurun_out/tail_ab.txt
for t in "" conv3; do echo -n "factor_celeba DVAE_TAIL_MAIN='$t': "; DVAE_TAIL_MAIN="$t" bench --config factor_celeba; done | tee -a gpurun_out/tail_ab.txt
for c in 8192 1000000; do echo -n "DVAE_FLAT_CHUNK=$c: "; DVAE_FLAT_CHUNK=$c bench; done | tee -a gpurun_out/tail_ab.txt
echo "== timeline (default)"
rm -rf gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$REPO/gpurun_out/prof" -o prof -- python "$REPO/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check --no-roofline > "$REPO/gpurun_out/prof.log" 2>&1)
python tools/timeline.py gpurun_out/prof/prof_results.db > gpurun_out/timeline.txt 2>&1; tail -n 16 gpurun_out/timeline.txt
