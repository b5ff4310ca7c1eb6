#!/bin/bash
# round profiles: kernel trace + FETCH_SIZE / WRITE_SIZE / SQ passes for the four bench configurations
tools/profile_bench.sh evm_2p18 --workload evm --steps 50 --warmup 5 --no-other-configs > gpurun_out/prof_evm.log 2>&1
tools/profile_bench.sh state_2p16 --workload state --steps 50 --warmup 5 > gpurun_out/prof_state.log 2>&1
tools/profile_bench.sh tx_2p14 --workload tx --steps 6 --warmup 2 > gpurun_out/prof_tx.log 2>&1
tools/profile_bench.sh super_2p20 --workload super --steps 10 --warmup 3 > gpurun_out/prof_super.log 2>&1
tail -3 gpurun_out/prof_*.log
