set -e
mkdir -p gpurun_out/blk
export ZK_BLOCK_NATIVE=1
timeout 300 python tools/bench_block_oneshot.py > gpurun_out/blk/native.txt 2>&1 || true
ZK_BLOCK_STATE_ROWS=1 timeout 300 python tools/bench_block_oneshot.py > gpurun_out/blk/native_rows.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/blk/trace -- python $GRAFT_REPO_ROOT/tools/bench_block_oneshot.py > /dev/null 2>&1 || true
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/blk/trace -name '*kernel_trace.csv' | head -1)
python tools/block_timeline.py $f > gpurun_out/blk/timeline.txt
rm -rf gpurun_out/blk/trace
