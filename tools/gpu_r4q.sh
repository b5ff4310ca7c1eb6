#!/bin/bash
# A/B: hot EVM build split into heavy (2 waves/SIMD) + light (3 waves/SIMD)
#   base  = one hot build, 238 VGPRs (tools/micro/libzkevm_hip_base.so)
#   all3  = one hot build with every group at 3 waves/SIMD (168 VGPRs, 800 B scratch), ZK_EVM_SPLIT=0
#   split1 = heavy then light on the session's stream;  split2 = light on the side stream
set -u
out=gpurun_out/r4q; mkdir -p $out
timeout 900 python -m pytest tests/test_evm_gpu.py -m gpu -x -q > $out/tests_split1.log 2>&1; echo "tests split1 rc=$?"; tail -2 $out/tests_split1.log
run() { # name, env...
name=$1; shift
env "$@" timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-batch-leg --steps 30 --warmup 5 > $out/evm_$name.json 2>/dev/null
python - $name <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r4q/evm_{sys.argv[1]}.json')); r=d['roofline']
print(f"{sys.argv[1]:8s} oneshot ms {d['ms_per_step']:.4f} span {r['kernel_ms']:.4f} open {r['open_ms']:.4f} pass {r['pass_kernel_ms']:.4f} | resident pass {r['resident_ms_per_pass']:.4f} hot {r['resident_hot_kernel_ms']:.4f}")
PY
}
for rep in 1 2; do
run base ZK_HIP_LIB=$PWD/tools/micro/libzkevm_hip_base.so
run all3 ZK_HIP_LIB=$PWD/tools/micro/libzkevm_hip_all3.so ZK_EVM_SPLIT=0
run split1 ZK_EVM_SPLIT=1
run split2 ZK_EVM_SPLIT=2
done
