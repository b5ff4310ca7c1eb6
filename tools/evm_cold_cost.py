#!/usr/bin/env python3
"""How long does the cold EVM launch take per kind of cold step?  (tuning aid: block traces with ONE of the SHA3 / CODECOPY / EXP kinds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from zkevm_specs_amd import engine, synth_block
from zkevm_specs_amd.super_circuit import SuperCircuit, synth_super_block

dev = lambda x: torch.from_numpy(x.view(np.int64) if x.dtype == np.uint64 else x.view(np.int32) if x.dtype == np.uint32 else x).cuda()
full = list(synth_block._BLOCK_MIX)
for mix in ([[], [full[0]], [full[1]], [full[2]], full] if not os.environ.get('ONLY') else [[full[int(k)] for k in os.environ['ONLY'].split(',') if k != '']]):
    synth_block._BLOCK_MIX[:] = mix if mix else [(1e-9, "EXP")]
    p = synth_super_block(int(os.environ.get("LOGT", "18")), seed=5)
    with SuperCircuit(p, to_device=dev) as sc:
        ev = sc.sessions["evm"]
        ev.run()
        for _ in range(10):
            ev.launch()
        r = ev.collect()
        states = p["evm"]["steps"][:, 0, 0]
        print([k for _, k in mix], "steps", p["rows"]["evm"], "copy rows", p["rows"]["copy"], "exp rows", p["rows"]["exp"], "evm kernel_ms (hot start -> cold end)", round(r.kernel_ms, 4), r.ok)
