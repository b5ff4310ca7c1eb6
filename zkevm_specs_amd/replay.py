"""Failure replay, an opt-in diagnostic: re-run ONE step pair on the reference's own Python path.

The device decides every pair of a trace — since round 4 also the pairs whose word cells are >= 2^128, where the reference
computes with unbounded Python integers (csrc/bigz.hpp) — and for the first failing pair it reports the class of the
exception the reference raises and the checkpoint.  What it cannot give is the reference's exception *message*.  A caller who
has the reference installed and hands this mirror the reference's own `Tables` / `StepState` objects can ask for it
(`ZK_REPLAY=always`): the mirror then evaluates exactly that pair with `zkevm_specs.evm_circuit.main.verify_step` and lets it
raise; the device's kind is the cross-check.  `ZK_REPLAY=unsupported` replays only pairs the device reports as
`UnsupportedOnDevice` (a `StepState.aux_data` of a Python type the wire does not carry).  The default is `never`: the product
path does not execute the reference.

Nothing here is imported unless a failure has to be replayed.
"""
import os


def reference_available():
    try:
        import zkevm_specs.evm_circuit.main  # noqa: F401
    except Exception:  # noqa: BLE001  (ImportError, or a missing third-party dependency of the reference)
        return False
    return True


def is_reference_tables(tables):
    """the reference's own Tables (evm_circuit/table.py:578): the replay needs its lookup methods"""
    return type(tables).__module__.startswith("zkevm_specs.") and hasattr(tables, "rw_lookup") and hasattr(tables, "fixed_lookup")


def replay_mode():
    """ZK_REPLAY = never (default) | unsupported (only where the device has no verdict) | always (every failure: the reference's
    own message)"""
    return os.environ.get("ZK_REPLAY", "never")


def replay_step(tables, steps, idx, begin_with_first_step=False, end_with_last_step=False):
    """verify_step of pair `idx` exactly as the reference's driver builds it (evm_circuit/main.py:25-38).  Raises what the
    reference raises; returns None when the reference accepts the pair."""
    from zkevm_specs.evm_circuit.instruction import Instruction
    from zkevm_specs.evm_circuit.main import verify_step

    verify_step(Instruction(tables=tables, curr=steps[idx], next=steps[idx + 1],
                            is_first_step=bool(begin_with_first_step) and idx == 0,
                            is_last_step=bool(end_with_last_step) and idx == len(steps) - 2))
