#!/bin/bash
# round-4 GPU check: the -m gpu suite, smoke(), the default bench line (one-shot headline), rocprofv3 passes of the one-shot command
set -u
out=gpurun_out/r4a; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
tail -c 1500 $out/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4a/bench.json'))
print("evm value", d['value'], "ms/step", d['ms_per_step'])
print({k:v for k,v in d['roofline'].items() if not isinstance(v,(dict,list))})
print({k:v for k,v in d['config'].items() if not isinstance(v,(dict,list))})
if 'fresh_witness' in d: print("fresh", d['fresh_witness']['open_plus_pass_ms'], d['fresh_witness']['split'], d['fresh_witness']['one_shot_c_entry_ms'])
print({k:(round(v['value']),v.get('cores')) for k,v in d.get('cpu_baseline',{}).get('legs',{}).items()})
PY
timeout 900 tools/profile_bench.sh evm_oneshot_2p18 --no-session-leg --no-other-configs --steps 20 --warmup 5 > $out/profile.log 2>&1; tail -40 $out/profile.log
