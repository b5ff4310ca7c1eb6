#!/bin/bash
# Per-gadget device time of the hot EVM kernel (tuning aid, run on the GPU box):
# rocprofv3 kernel trace of tools/evm_state_costs.py for single-opcode traces at 2^LOGN steps.
R=$(cd "$(dirname "$0")/.." && pwd); cd /tmp; export TMPDIR=/tmp
for K in ${@:-POP ADDSUB MEMORY SSTORE MULMOD}; do
  LOGN=${LOGN:-17} rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/t -- python $R/tools/evm_state_costs.py $K > /dev/null 2>&1
  python - <<PY
import csv,glob
for f in glob.glob("$R/gpurun_out/t/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "evm_steps_kernel<-1" in r["Name"]: print("$K", r["Name"][:44], round(float(r["AverageNs"])/1000,1), "us")
PY
  rm -rf $R/gpurun_out/t
done
