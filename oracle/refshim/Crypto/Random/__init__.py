"""RNG helpers. Seedable through ZKEVM_SHIM_SEED so golden generation is reproducible
(the reference's tests are unseeded: tests/common.py:93-110)."""
import os
import random as _random

_rng = _random.Random(int(os.environ.get("ZKEVM_SHIM_SEED", "0")) or None)


def get_random_bytes(n):
    return bytes(_rng.getrandbits(8) for _ in range(n))


from . import random  # noqa: E402,F401
