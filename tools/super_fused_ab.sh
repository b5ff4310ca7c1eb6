run() { printf "%-70s " "[$1]"; env $1 python bench.py --workload super --state-fused --steps 30 --warmup 5 --no-other-configs --no-cpu-baseline --no-cold-leg --no-fresh-leg --no-oneshot-leg 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['roofline'].get('per_circuit_kernel_ms'))"; }
for rep in 1 2; do
run "X=1"
run "ZK_SUPER_ORDER=state,evm,exp,tx,copy,bytecode"
run "ZK_SUPER_ORDER=evm,state,exp,tx,copy,bytecode"
run "ZK_SUPER_PRIO=0"
run "ZK_SUPER_PRIO_SET=exp,tx,copy,bytecode,state"
run "ZK_SUPER_PRIO_SET=exp,tx,copy,bytecode,evm"
done
