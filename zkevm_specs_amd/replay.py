"""Failure replay: re-run ONE step pair on the reference's own Python path.

The device decides every pair of a trace; for the first failing pair it reports the class of the exception the reference
raises and the checkpoint.  Two things it cannot give are (a) the reference's exception *message* and (b) a verdict for the
few witness shapes outside the wire domain — word cells >= 2^128 where the reference computes with unbounded Python
integers (execution/mul_div_mod.py:23-41, shl_shr.py:103-127, sar.py, sdiv_smod.py:85-99, addmod.py, mulmod.py), reported
as `UnsupportedOnDevice`.  A caller that switches over from the reference has the reference installed and hands this mirror
the reference's own `Tables` / `StepState` objects, so the mirror can do what §8b recommends: evaluate exactly that pair with
`zkevm_specs.evm_circuit.main.verify_step` and let it raise.  One pair costs the reference a few linear scans of the tables
(seconds on a large block) — only ever on a failing witness.

Nothing here is imported unless a failure has to be replayed; without the reference the mirror keeps raising the mapped
exception (or `UnsupportedOnDevice`).
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
    """ZK_REPLAY = unsupported (default: only where the device has no verdict) | always (every failure: the reference's own
    message) | never"""
    return os.environ.get("ZK_REPLAY", "unsupported")


def replay_step(tables, steps, idx, begin_with_first_step=False, end_with_last_step=False):
    """verify_step of pair `idx` exactly as the reference's driver builds it (evm_circuit/main.py:25-38).  Raises what the
    reference raises; returns None when the reference accepts the pair."""
    from zkevm_specs.evm_circuit.instruction import Instruction
    from zkevm_specs.evm_circuit.main import verify_step

    verify_step(Instruction(tables=tables, curr=steps[idx], next=steps[idx + 1],
                            is_first_step=bool(begin_with_first_step) and idx == 0,
                            is_last_step=bool(end_with_last_step) and idx == len(steps) - 2))
