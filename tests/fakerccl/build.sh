#!/bin/bash
# builds the stand-in collective libraries of tests/fakerccl (test infrastructure; see fake_rccl.cpp)
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -shared -fPIC -Wall -o libfakerccl_host.so fake_rccl.cpp
if [ -x /opt/rocm/bin/hipcc ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 -shared -fPIC -DFAKE_RCCL_HIP -o libfakerccl_hip.so fake_rccl.cpp
fi
