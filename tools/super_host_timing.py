"""Host-side cost of submitting one Super circuit pass: perf_counter around every session's zk_launch over back-to-back passes
(no collect in between), next to the wall time per pass.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import _lib  # noqa: E402

_lib.load()
import torch  # noqa: E402

from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block  # noqa: E402


def dev(x):
    import numpy as np
    return torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()


parts = synth_super_block(20, seed=5)
sc = SuperCircuit(parts, device=0, to_device=dev)
order = [k for k in sc.LAUNCH_ORDER if k in sc.sessions]
for _ in range(3):
    sc.launch()
sc.collect()
torch.cuda.synchronize()
K = 20
per = {k: [] for k in order}
t0 = time.perf_counter()
for _ in range(K):
    for k in order:
        a = time.perf_counter()
        sc.sessions[k].launch()
        per[k].append(time.perf_counter() - a)
t_submit = time.perf_counter() - t0
sc.collect()
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"wall per pass {t_all / K * 1e6:.1f} us; host submission per pass {t_submit / K * 1e6:.1f} us")
for k in order:
    v = sorted(per[k])
    print(f"  {k:9s} zk_launch median {v[len(v) // 2] * 1e6:6.1f} us  max {v[-1] * 1e6:6.1f} us")
sc.close()
