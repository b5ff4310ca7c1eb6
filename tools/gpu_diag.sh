#!/bin/bash
export PYTHONPATH=$PWD
echo "== current"; python tools/bench_row_kernels.py 2>&1 | grep "^copy_rows\|^bytecode \|^tx_sign"
echo "== original probe"; ZK_HIP_LIB=$PWD/zkevm_specs_amd/libzkevm_hip_orig.so python tools/bench_row_kernels.py 2>&1 | grep "^copy_rows\|^bytecode \|^tx_sign"
