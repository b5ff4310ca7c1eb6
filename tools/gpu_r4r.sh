#!/bin/bash
# A/B: zk_collect polls the stream before it blocks (ZK_SPIN_WAIT_US) / HSA_ENABLE_INTERRUPT=0
set -u
out=gpurun_out/r4r; mkdir -p $out
run() { name=$1; shift
env "$@" timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-session-leg --steps 30 --warmup 5 > $out/evm_$name.json 2>/dev/null
python - $name <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4r/evm_{sys.argv[1]}.json')); r=d['roofline']
print(f"{sys.argv[1]:10s} oneshot ms {d['ms_per_step']:.4f} span {r['kernel_ms']:.4f} batch {r['batch_ms_per_witness']:.4f} host us open {r['host_us_in_open']:.1f} launch {r['host_us_in_launch']:.1f} collect {r['host_us_in_collect']:.1f} close {r['host_us_in_close']:.1f}")
PY
}
for rep in 1 2; do
run block ZK_SPIN_WAIT_US=0
run spin ZK_SPIN_WAIT_US=2000
run hsa_poll ZK_SPIN_WAIT_US=0 HSA_ENABLE_INTERRUPT=0
run both ZK_SPIN_WAIT_US=2000 HSA_ENABLE_INTERRUPT=0
done
for wl in state tx super; do
for sp in 0 2000; do
ZK_SPIN_WAIT_US=$sp timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl spin=$sp', round(d['ms_per_step'],4))"
done; done
