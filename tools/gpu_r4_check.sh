#!/bin/bash
# HEAD check: the whole -m gpu suite and smoke()
set -u
out=gpurun_out/r4check; mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $out/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
