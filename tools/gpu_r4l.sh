#!/bin/bash
set -u
mkdir -p gpurun_out/r4l
ZK_HIP_LIB=$PWD/tools/micro/libzkevm_hip_diag.so timeout 600 python tools/p1_ranges.py 2>&1 | tail -20
timeout 600 python -m pytest tests/test_evm_gpu.py -m gpu -x -q -k "warm_gadget" 2>&1 | tail -3
