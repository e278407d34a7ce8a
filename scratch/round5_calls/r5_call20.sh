#!/bin/bash
# round 5, call 20: the hybrid's chunk graphs when one cannot be instantiated (plain launches take over), the other hybrid forms, smoke
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "hybrid or hub or stress" > gpurun_out/r5c20_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r5c20_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
