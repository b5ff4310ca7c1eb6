import sys, os, json, random, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from zkevm_specs_amd import engine
rng = random.Random(3)
r = rng.randrange(1 << 250)
out = {}
for name, n, ln in (("24KiB", 4096, 24576), ("8KiB_x16", 16, 8192), ("24KiB_x16", 16, 24576)):
    msgs = [bytes(rng.getrandbits(8) for _ in range(256)) * (ln // 256) for _ in range(n)]
    data, offsets = engine.pack_messages(msgs)
    d, o = torch.from_numpy(data).cuda(), torch.from_numpy(offsets.view(np.int64)).cuda()
    rows = torch.empty((n, 5, 4), dtype=torch.int64, device="cuda")
    with engine.open_keccak(d, o, r, engine.KECCAK_MODE_CIRCUIT, rows_dev=rows) as s:
        for _ in range(3): s.run()
        ms = sorted(s.run().kernel_ms for _ in range(8))
    out[name] = ms[len(ms)//2]
print(json.dumps(out))
