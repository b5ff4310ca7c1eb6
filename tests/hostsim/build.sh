#!/bin/bash
# Build the CPU logic harness (test infrastructure only).
set -e
cd "$(dirname "$0")"
g++ -O2 -std=c++17 -DZK_HOSTSIM -shared -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-array-bounds -o libhostsim.so hostsim.cpp
