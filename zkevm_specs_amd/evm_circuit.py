"""Host-side mirror of the reference's EVM-circuit driver, evaluated on the MI355X.

Reference seam: `zkevm_specs.evm_circuit.verify_steps(tables, steps, begin_with_first_step=False,
end_with_last_step=False, success=True)` (evm_circuit/main.py:14-44), which loops
`verify_step(Instruction(tables, curr, next, ...))` over consecutive step pairs.  Here the loop
is one device pass (`zk_evm_verify`); arguments, the dummy EndBlock step and the error behaviour are the same:
the first failing pair decides — an AssertionError there is swallowed and re-raised iff `success`
(main.py:36-44), any other exception class propagates as is.
"""
from . import oneshot, replay
from .errors import KIND_UNSUPPORTED, exception_for_code
from .evm_tables import ExecutionState
from .flatten import flatten_evm
from .objects import Word  # noqa: F401  (re-exported: callers build code hashes with it)


class StepState:
    """Same constructor as the reference's StepState (evm_circuit/step.py:47-75)."""

    def __init__(self, execution_state, rw_counter, call_id=0, is_root=False, is_create=False, code_hash=None,
                 program_counter=0, stack_pointer=1024, gas_left=0, memory_word_size=0,
                 reversible_write_counter=0, log_id=0, aux_data=None):
        self.execution_state = execution_state
        self.rw_counter = rw_counter
        self.call_id = call_id
        self.is_root = is_root
        self.is_create = is_create
        self.code_hash = code_hash if code_hash is not None else Word(0)
        self.program_counter = program_counter
        self.stack_pointer = stack_pointer
        self.gas_left = gas_left
        self.memory_word_size = memory_word_size
        self.reversible_write_counter = reversible_write_counter
        self.log_id = log_id
        self.aux_data = aux_data


def _dummy_step():
    return StepState(ExecutionState.EndBlock, rw_counter=-1)  # main.py:11


def verify_steps(tables, steps, begin_with_first_step=False, end_with_last_step=False, success=True):
    if end_with_last_step:
        steps.append(_dummy_step())  # the reference mutates the caller's list too (main.py:21-22)
    if len(steps) < 2:
        assert success  # no pair, no exception (main.py:40-44)
        return None
    res, status = oneshot.evm_verify(flatten_evm(tables, steps), begin_with_first_step, end_with_last_step)
    exception = None
    if not res.ok:
        # The first failing pair decides (the reference's loop stops there).  Failure replay (replay.py) is an opt-in
        # diagnostic (ZK_REPLAY, default never): on request the failing pair is evaluated by the reference's own verify_step on
        # the caller's own objects, for its message; with ZK_REPLAY=unsupported only a pair without a device verdict
        # (UnsupportedOnDevice: an aux_data shape the wire does not carry) is, and a pair the reference accepts does not
        # decide, the next failing pair does.
        mode = replay.replay_mode()
        can_replay = mode != "never" and replay.is_reference_tables(tables) and replay.reference_available()
        for row in (int(j) for j in status.nonzero()[0]):
            code = int(status[row])
            exc = exception_for_code(code, f"EVM circuit step {row}")
            if can_replay and (mode == "always" or (code >> 24) == KIND_UNSUPPORTED):
                try:
                    replay.replay_step(tables, steps, row, begin_with_first_step, end_with_last_step)
                except AssertionError as e:  # main.py:36-38: swallowed, decides through `success`
                    exc = e
                else:
                    if (code >> 24) == KIND_UNSUPPORTED:
                        continue  # the reference accepts this pair
            if not isinstance(exc, AssertionError):
                raise exc
            exception = exc
            break
    if success:
        if exception:
            raise exception
    else:
        assert exception is not None
    return res


def verify_steps_status(tables, steps, begin_with_first_step=False, end_with_last_step=False):
    """Diagnostic form: per-pair status codes of every step pair (no early stop)."""
    if end_with_last_step:
        steps = list(steps) + [_dummy_step()]
    return oneshot.evm_verify(flatten_evm(tables, steps), begin_with_first_step, end_with_last_step)
