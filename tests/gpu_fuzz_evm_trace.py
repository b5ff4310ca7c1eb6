import os, sys, random
sys.path.insert(0, os.getcwd())
import numpy as np
from tests.evm_cases import fuzz_wire, oracle_status
from zkevm_specs_amd import engine
from zkevm_specs_amd.synth_evm import synth_evm_trace
P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
bad = 0
for seed in (41, 42, 43):
    rng = random.Random(seed)
    w = synth_evm_trace(1 << 13, seed=seed); w.pop("meta")
    for _ in range(300):
        i, c = rng.randrange(w["steps"].shape[0]), rng.randrange(1, 13)
        old = int.from_bytes(w["steps"][i, c].tobytes(), "little")
        new = rng.choice([old + 1, old - 1, 0, 2**64 - 1, 2**64, 2**64 + old, 2**128, P - 1, old ^ (1 << rng.randrange(66)), rng.randrange(2**64)]) % P
        w["steps"][i, c] = np.frombuffer(new.to_bytes(32, "little"), dtype="<u8")
    for _ in range(100):
        w = fuzz_wire(w, rng)
    exp = oracle_status(w)
    for sort in (True, False):
        with engine.open_evm(w, state_sort=sort) as s:
            s.run(); got = s.read_status().tolist()
        if got != exp:
            bad += 1
            d = [(j, g, e) for j, (g, e) in enumerate(zip(got, exp)) if g != e][:5]
            print("MISMATCH seed", seed, sort, d)
    # round 4: the side-stream form of the pass and the one-shot C entry (single-pass open: staging from the step rows) on the same witness
    import torch
    with engine.open_evm(w, side_stream=True) as s:
        s.run(); s.run(); got = s.read_status().tolist()
    if got != exp:
        bad += 1; print("MISMATCH side stream, seed", seed)
    dev = {k: torch.from_numpy(v.view(np.int64) if v.dtype == np.uint64 else v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in w.items()}
    buf = torch.full((len(exp),), 0x7fffffff, dtype=torch.int32, device="cuda")
    res = engine.evm_verify(dev, status_dev=buf)
    got = buf.cpu().numpy().view(np.uint32).tolist()
    if got != exp or res.fail_count != sum(1 for e in exp if e):
        bad += 1; print("MISMATCH one-shot, seed", seed)
    print(seed, "failing", sum(1 for e in exp if e))
print("bad", bad)
