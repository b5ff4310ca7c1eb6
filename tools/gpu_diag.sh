#!/bin/bash
export PYTHONPATH=$PWD
timeout 1500 python -m pytest tests/test_state_gpu.py -m gpu -x -q 2>&1 | tail -4
