"""Which block range of evm_open_phase1_kernel bounds the launch?  Opens the bench's 2^18-step witness (ZK_OPT_SINGLE_PASS, as
zk_evm_verify does) with ranges of phase 1 switched off and reads the open's device span (zk_session_timing).  Needs the tuning
build of the library (-DZK_DIAG_P1, ZK_HIP_LIB=tools/micro/libzkevm_hip_diag.so): with a range off the session's results are
INVALID, so no pass is launched here.  Between opens a 2 GiB read-only sweep evicts the witness from MALL / L2."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zkevm_specs_amd import _lib, engine  # noqa: E402

_lib.load()
import torch  # noqa: E402

from zkevm_specs_amd.synth_evm import synth_evm_trace  # noqa: E402

lib = _lib.init(0)
w = synth_evm_trace(1 << 18, seed=3)
w.pop("meta")
dev = {k: torch.from_numpy(v.view("int64") if v.dtype.name == "uint64" else v.view("int32") if v.dtype.name == "uint32" else v).cuda() for k, v in w.items()}
flush = torch.empty(1 << 28, dtype=torch.int64, device="cuda").fill_(1)
NAMES = {1: "small tables", 2: "directory rows", 4: "histogram", 8: "RW pack"}
out = {}
for mask in (0, 16, 15, 1, 2, 4, 8, 7, 14, 13, 11, 3, 12):  # 16 = nothing off, but phase 2 not launched (as in every masked run)
    os.environ["ZK_DIAG_P1_SKIP"] = str(mask)
    ts = []
    for rep in range(6):
        flush.sum().item()
        t, opts, arrs, n_pairs = engine._evm_tables(dev, False, False)
        h = ctypes.c_void_p()
        engine.check(lib.zk_evm_open(ctypes.byref(t), opts | _lib.OPT_SINGLE_PASS, ctypes.byref(h)), "zk_evm_open", lib)
        torch.cuda.synchronize()
        a, b = ctypes.c_double(), ctypes.c_double()
        lib.zk_session_timing(h, ctypes.byref(a), ctypes.byref(b))
        lib.zk_close(h)
        if rep:
            ts.append(a.value * 1e3)
    on = [n for m, n in NAMES.items() if not (mask & m)]
    out[str(mask)] = {"ranges_on": on, "open_us_median": sorted(ts)[len(ts) // 2], "open_us_min": min(ts)}
    print(mask, on, out[str(mask)]["open_us_median"], out[str(mask)]["open_us_min"], flush=True)
os.makedirs("gpurun_out/r4l", exist_ok=True)
json.dump(out, open("gpurun_out/r4l/p1_ranges.json", "w"), indent=1)
