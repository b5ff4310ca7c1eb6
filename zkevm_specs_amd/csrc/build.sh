#!/bin/bash
# Build libzkevm_hip.so for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
    -Rpass-analysis=kernel-resource-usage $ZK_EXTRA_FLAGS \
    -o ../libzkevm_hip.so zkevm_hip.hip 2> build.log || { cat build.log; exit 1; }
grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" build.log | paste - - - - - - | sed 's/remark: [^ ]* //g' > resource_usage.txt || true
echo "built $(ls -la ../libzkevm_hip.so)"
