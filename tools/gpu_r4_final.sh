#!/bin/bash
# round-4 evidence run: the -m gpu suite, smoke(), the driver's bench command, the reference's own suite through the HIP library,
# rocprofv3 passes of the five bench configurations, the row-kernel bench
set -u
out=gpurun_out/r4final; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -4 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_default.json 2> $out/bench.err; echo "bench rc=$?"
timeout 900 python tools/run_reference_suite.py --backend hip --ref-root oracle/_ref/reference --out $out/reference_suite_hip.json > $out/reference_suite_hip.log 2>&1; tail -1 $out/reference_suite_hip.log
python tools/bench_row_kernels.py > $out/row_kernels.txt 2>&1; tail -1 $out/row_kernels.txt > $out/row_kernels.json
tools/profile_bench.sh evm_oneshot_2p18 --no-session-leg --no-batch-leg --no-other-configs --steps 20 --warmup 5 > $out/prof_evm_oneshot.log 2>&1
tools/profile_bench.sh evm_2p18 --session-pass --no-other-configs --steps 50 --warmup 5 > $out/prof_evm.log 2>&1
tools/profile_bench.sh state_2p16 --workload state --steps 50 --warmup 5 > $out/prof_state.log 2>&1
tools/profile_bench.sh tx_2p14 --workload tx --steps 6 --warmup 2 > $out/prof_tx.log 2>&1
tools/profile_bench.sh super_2p20 --workload super --steps 10 --warmup 3 > $out/prof_super.log 2>&1
tail -2 $out/prof_*.log | cut -c1-300
