#!/bin/bash
# round 5, call 18: the last changes (text units as views, index loader) on hardware
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fastq_units.py tests/test_gpu_cli.py -x -q > gpurun_out/r5c18_pytest.log 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r5c18_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
