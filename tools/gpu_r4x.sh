#!/bin/bash
# differential fuzz of the HIP path against the oracle at HEAD of round 4
set -u
out=gpurun_out/r4x; mkdir -p $out
( echo "== state fuzz"; timeout 900 python tests/gpu_fuzz_state.py 1500 31 2>&1 | tail -2
  echo "== state fuzz, lane-quad form"; ZK_STATE_DMA=0 timeout 600 python tests/gpu_fuzz_state.py 100 9 2>&1 | tail -1
  echo "== evm pair fuzz"; timeout 900 python tests/gpu_fuzz_evm.py 40 29 2>&1 | tail -2
  echo "== evm trace fuzz (sorted / unsorted / side stream / one-shot)"; timeout 900 python tests/gpu_fuzz_evm_trace.py 2>&1 | tail -5 ) > $out/fuzz.txt 2>&1
cat $out/fuzz.txt
