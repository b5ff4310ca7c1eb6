#!/bin/bash
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_row_circuits.py tests/test_bytecode_assign.py tests/test_super_circuit.py tests/test_dropin_gpu.py tests/test_cpu_backend.py -m gpu -x -q 2>&1 | tail -3
python tools/bench_row_kernels.py 2>&1 | grep "^bytecode \|^exp \|^tx_sign"
timeout 900 python -m pytest tests/test_bench_multi_gpu_dryrun.py -m gpu -x -q -k "super" 2>&1 | tail -2
