#!/bin/bash
set -u
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
mkdir -p gpurun_out
T=gpurun_out/${TAG:-r04_v12}
DVAE_PARITY_STATS=${T}_parity_stats.json timeout 1500 python -m pytest tests -m gpu -q --timeout=600 --no-header -x > ${T}_pytest.log 2>&1
echo "pytest exit: $?"; tail -n 15 ${T}_pytest.log
