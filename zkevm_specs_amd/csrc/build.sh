#!/bin/bash
# Build libzkevm_hip.so for gfx950 (cross-compiles without a GPU).  The translation units are independent (each k_*.hip
# defines its kernels + a host launcher, kernels.hpp), so they compile in parallel; no relocatable device code.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${ZK_BUILD_DIR:-build}
mkdir -p "$OUT"
UNITS="zkevm_hip k_state k_evm_hot k_evm_cold k_evm_warm k_evm_slow k_rows k_assign k_ecdsa k_rekey k_state_fused"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage $ZK_EXTRA_FLAGS"
pids=()
for u in $UNITS; do
    # rebuild a unit only when one of the sources is newer than its object (headers are shared: any header change rebuilds all)
    if [ ! -f "$OUT/$u.o" ] || [ -n "$(find . ../../include -maxdepth 1 \( -name '*.hpp' -o -name '*.h' -o -name "$u.hip" -o -name build.sh \) -newer "$OUT/$u.o" | head -1)" ] || [ -n "$ZK_EXTRA_FLAGS" ] || [ -f "$OUT/.extra_flags" ]; then
        ( $HIPCC $FLAGS -c -o "$OUT/$u.o" "$u.hip" 2> "$OUT/$u.log" || { cat "$OUT/$u.log"; exit 1; } ) &
        pids+=($!)
    fi
done
fail=0
for p in "${pids[@]}"; do wait "$p" || fail=1; done
if [ $fail -ne 0 ]; then echo "build failed (see $OUT/*.log)"; exit 1; fi
if [ -n "$ZK_EXTRA_FLAGS" ]; then touch "$OUT/.extra_flags"; else rm -f "$OUT/.extra_flags"; fi
OBJS=""
for u in $UNITS; do OBJS="$OBJS $OUT/$u.o"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libzkevm_hip.so $OBJS
# the CPU backend behind the same C ABI (cpu_backend.cpp: the same headers compiled for the host, OpenMP over the rows)
if [ ! -f ../libzkevm_cpu.so ] || [ -n "$(find . ../../include -maxdepth 1 \( -name '*.hpp' -o -name '*.h' -o -name cpu_backend.cpp -o -name build.sh \) -newer ../libzkevm_cpu.so | head -1)" ]; then
    g++ -O2 -std=c++17 -fopenmp -DZK_HOSTSIM -shared -fPIC -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-array-bounds -o ../libzkevm_cpu.so cpu_backend.cpp -ldl
fi
# the marshalling helper of zkevm_specs_amd/flatten.py (CPython extension, host side only)
if [ ! -f ../_flatten_ext.so ] || [ flatten_ext.c -nt ../_flatten_ext.so ]; then
    gcc -O2 -shared -fPIC -Wall -I"$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')" -o ../_flatten_ext.so flatten_ext.c
fi
cat $OUT/*.log | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size" | paste - - - - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' > "$OUT/resource_usage.txt" || true
echo "built $(ls -la ../libzkevm_hip.so)"
