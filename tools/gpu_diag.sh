#!/bin/bash
PROFILE_CMD="python $PWD/tools/bench_row_kernels.py" tools/profile_bench.sh row_kernels > gpurun_out/prof_rows.log 2>&1
tail -5 gpurun_out/prof_rows.log | cut -c1-200
