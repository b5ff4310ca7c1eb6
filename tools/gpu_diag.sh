#!/bin/bash
export PYTHONPATH=$PWD
ZK_HIP_LIB=$PWD/zkevm_specs_amd/libzkevm_hip_st.so python tools/evm_warm_timeline.py 2>&1 | tail -6
