#!/bin/bash
# ONE parameterised lease script (run on the GPU box through gpurun): `tools/gpu_run.sh <tag> <stage>...`, stages in the order given.
#   tests      the -m gpu suite (PYTEST_ARGS narrows it)          smoke     __graft_entry__.smoke()
#   bench      the driver's command: python bench.py --gpus 1 --steps 20 --warmup 5; checks that the LAST stdout line parses
#   prof-evm   rocprofv3 passes of the one-shot headline           prof-session / prof-state / prof-tx / prof-super: the other configurations
#   rows       tools/bench_row_kernels.py                          refsuite  the reference's own tests through the HIP library
#   fuzz       differential fuzz (State, EVM pairs, EVM traces) against the oracle
#   ab:<A=x,B=y> A/B of environment switches on the one-shot headline (default, each switch alone, all together; twice)
#   cmd:<...>  any shell command (quote it)
# Everything lands in gpurun_out/<tag>/ (merged back by gpurun); copy what is to be judged into profiles/.
set -u
tag=$1; shift
out=gpurun_out/$tag; mkdir -p "$out"
for stage in "$@"; do
    t0=$(date +%s)
    case "$stage" in
    tests)    eval "timeout 2400 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-}" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log ;;
    smoke)    python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log ;;
    bench)    timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_stdout.txt 2> $out/bench.err; echo "bench rc=$?"
              cp -f bench_full.json $out/bench_full.json 2>/dev/null
              python - "$out" <<'PY'
import json, sys
lines = open(sys.argv[1] + "/bench_stdout.txt").read().splitlines()
d = json.loads(lines[-1])
print("last stdout line:", len(lines[-1]), "bytes; value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"],
      "kernel_ms", d["roofline"]["kernel_ms"], "cpu_baseline", d.get("cpu_baseline", {}).get("value"))
print(lines[-1])
PY
              ;;
    prof-evm)     tools/profile_bench.sh evm_oneshot_2p18 --no-session-leg --no-batch-leg --no-other-configs --no-live-pmc --steps 20 --warmup 5 > $out/prof_evm_oneshot.log 2>&1; tail -3 $out/prof_evm_oneshot.log | cut -c1-400 ;;
    prof-session) tools/profile_bench.sh evm_2p18 --session-pass --no-other-configs --steps 50 --warmup 5 > $out/prof_evm.log 2>&1; tail -2 $out/prof_evm.log | cut -c1-300 ;;
    prof-state)   tools/profile_bench.sh state_2p16 --workload state --steps 50 --warmup 5 > $out/prof_state.log 2>&1; tail -2 $out/prof_state.log | cut -c1-300 ;;
    prof-tx)      tools/profile_bench.sh tx_2p14 --workload tx --steps 6 --warmup 2 > $out/prof_tx.log 2>&1; tail -2 $out/prof_tx.log | cut -c1-300 ;;
    prof-super)   tools/profile_bench.sh super_2p20 --workload super --steps 10 --warmup 3 > $out/prof_super.log 2>&1; tail -2 $out/prof_super.log | cut -c1-300 ;;
    prof-super-fused) tools/profile_bench.sh super_fused_2p20 --workload super --state-fused --steps 10 --warmup 3 > $out/prof_super_fused.log 2>&1; tail -2 $out/prof_super_fused.log | cut -c1-300 ;;
    prof-rows)    PROFILE_CMD="python $PWD/tools/bench_row_kernels.py" LOGN=${LOGN:-20} tools/profile_bench.sh row_kernels > $out/prof_rows.log 2>&1; tail -2 $out/prof_rows.log | cut -c1-300 ;;
    prof-rekey)   PROFILE_CMD="python $PWD/tools/bench_rekey.py 18 8" tools/profile_bench.sh rekey_2p18 > $out/prof_rekey.log 2>&1; tail -2 $out/prof_rekey.log | cut -c1-300 ;;
    rows)     python tools/bench_row_kernels.py > $out/row_kernels.txt 2>&1; tail -1 $out/row_kernels.txt > $out/row_kernels.json; cut -c1-600 $out/row_kernels.json ;;
    fuzz)     # differential fuzz of the HIP path against the oracle (per-row / per-pair status words bit for bit)
              { echo "== python tests/gpu_fuzz_state.py ${FUZZ_STATE:-600} 31"; timeout 900 python tests/gpu_fuzz_state.py ${FUZZ_STATE:-600} 31 2>&1 | tail -2
                echo "== ZK_STATE_DMA=0 python tests/gpu_fuzz_state.py 60 9"; ZK_STATE_DMA=0 timeout 600 python tests/gpu_fuzz_state.py 60 9 2>&1 | tail -2
                echo "== python tests/gpu_fuzz_copy.py ${FUZZ_COPY:-150} 7"; timeout 900 python tests/gpu_fuzz_copy.py ${FUZZ_COPY:-150} 7 2>&1 | tail -3
                echo "== python tests/gpu_fuzz_evm.py 40 29"; timeout 900 python tests/gpu_fuzz_evm.py 40 29 2>&1 | tail -2
                echo "== python tests/gpu_fuzz_evm_trace.py"; timeout 900 python tests/gpu_fuzz_evm_trace.py 2>&1 | tail -3; } > $out/fuzz.txt 2>&1; grep -v amdgpu.ids $out/fuzz.txt | cut -c1-300 ;;
    ab:*)     # A/B of environment switches on the one-shot headline: `ab:ZK_P1_TAIL=0,ZK_PRECLEAN=0` runs the default, each switch, and all of them
              sw="${stage#ab:}"; IFS=',' read -ra S <<< "$sw"
              q="--no-other-configs --no-cpu-baseline --no-live-pmc --no-session-leg --no-batch-leg --no-cold-leg --no-fresh-leg --steps 50 --warmup 5"
              for rep in 1 2; do
                for e in "" "${S[@]}" "${sw//,/ }"; do
                  printf "%-40s " "[${e:-default}]"; env $e python bench.py $q 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f span %.4f open %.4f pass %.4f' % (d['ms_per_step'], r['kernel_ms'], r['open_ms'], r['pass_kernel_ms']))"
                done
              done | tee $out/ab_$(date +%s).txt ;;
    cmd:*)    bash -c "${stage#cmd:}" > $out/cmd_$(date +%s).log 2>&1; echo "cmd rc=$?"; tail -5 $out/cmd_*.log | cut -c1-400 ;;
    *)        echo "unknown stage $stage" ;;
    esac
    echo "[$stage: $(( $(date +%s) - t0 )) s]"
done
