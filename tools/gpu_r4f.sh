#!/bin/bash
set -u
out=gpurun_out/r4f; mkdir -p $out
python tools/bench_row_kernels.py > $out/row_kernels.txt 2>&1; grep -E "^ecdsa|^copy_rows|^keccak" $out/row_kernels.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -4 $out/pytest.log
