"""The drop-in boundary on the MI355X: (1) every one-shot C entry INTEGRATION.md's reference-side stub binds
(`zk_evm_verify`, `zk_state_verify`, `zk_bytecode_verify`, `zk_exp_verify`, `zk_copy_verify`, `zk_sign_verify`,
`zk_keccak_table`, `zk_state_assign`, `zk_bytecode_assign`, `zk_ecdsa_verify`) on every golden case vs the oracle;
(2) the Python mirrors of the names the reference's tests import (`verify_steps`, `check_state_row`,
`verify_copy_table`, `verify_exp_circuit`, Tx / Sig `verify_circuit`, ...) driven with witness objects, against the
outcomes of the reference's own drivers.  Bodies: tests/dropin_cases.py."""
import pytest

from tests import dropin_cases as D

pytestmark = pytest.mark.gpu


def test_oneshot_state_verify():
    D.oneshot_state()


def test_oneshot_evm_verify():
    D.oneshot_evm()


def test_oneshot_bytecode_exp_copy_sign_verify():
    D.oneshot_bytecode_exp_copy_sign()


def test_oneshot_keccak_table_assign_ecdsa():
    D.oneshot_keccak_assign_ecdsa()


def test_mirror_verify_steps_success_and_failure():
    D.mirror_evm_verify_steps()


def test_mirror_state_circuit():
    D.mirror_state()


def test_mirror_bytecode_circuit():
    D.mirror_bytecode()


def test_mirror_copy_and_exp_circuits():
    D.mirror_copy_exp()


def test_mirror_tx_and_sig_verify_circuit():
    D.mirror_tx_sig()


def test_mirror_pi_verify_circuit():
    D.mirror_pi_verify_circuit()


def test_sessions_are_independent_contexts():
    """two sessions on two streams driven from two threads; zk_read_status refuses a pass whose statuses went elsewhere"""
    import threading

    import numpy as np
    import torch

    from oracle import state_oracle, wire
    from zkevm_specs_amd import _lib, engine
    from zkevm_specs_amd.synth import synth_state_witness

    outs, errs = {}, []

    def work(seed):
        try:
            cols, flags, mpt = synth_state_witness(3000 + seed, seed=seed)
            cols[1, 100 + seed, 0] = np.uint64(2)
            _lib.init(0)
            st = torch.cuda.Stream()
            _lib.check(_lib.load().zk_set_stream(st.cuda_stream), "zk_set_stream")
            with engine.open_state(cols, flags, mpt) as s:
                for _ in range(20):
                    s.launch()
                res = s.collect()
                outs[seed] = (res, s.read_status().tolist(),
                              state_oracle.verify_rows(wire.colmajor_to_rows(cols), flags, wire.rowmajor_to_rows(mpt)))
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in (1, 2, 3, 4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    for seed, (res, got, exp) in outs.items():
        assert got == exp and res.fail_count == sum(1 for e in exp if e) >= 1
    cols, flags, mpt = synth_state_witness(2048, seed=9)
    with engine.open_state(cols, flags, mpt) as s:
        assert not s.read_status().any()  # zeroed at open
        ext = torch.zeros(2048, dtype=torch.int32, device="cuda")
        s.launch(ext)
        s.collect()
        with pytest.raises(_lib.EngineError):
            s.read_status()
        s.set_stream(torch.cuda.Stream())
        assert s.run().ok and not s.read_status().any()


def test_tally_exchange_through_the_c_abi():
    """zk_dist_* (include/zkevm_hip.h): the engine's own RCCL communicator.  One GPU here, so world = 1 — communicator creation,
    the all-gather on the communicator's stream and the host-side SUM / MIN are exercised; the result must be the local tally
    with the shard's row offset applied, and equal to what the torch.distributed mirror (reduce_tally) returns."""
    import numpy as np

    from zkevm_specs_amd import distributed, engine
    from zkevm_specs_amd.synth import synth_state_witness

    cols, flags, mpt = synth_state_witness(4096, seed=6)
    cols[1, 1234, 0] = np.uint64(2)
    cols[50, 3000, 0] ^= np.uint64(1)
    with engine.open_state(cols, flags, mpt) as s:
        res = s.run()
    assert res.fail_count >= 2 and res.first_fail_row == 1234
    with distributed.RcclTally(0, 1, device=0) as t:
        for off in (0, 1 << 20, (1 << 33) + 5):  # global rows beyond 32 bits
            assert t.reduce(res, row_offset=off) == (res.fail_count, 1234 + off, res.first_fail_code)
            assert t.reduce(res, row_offset=off) == distributed.reduce_tally(res.fail_count, res.first_fail_row, res.first_fail_code, off)

        class Clean:
            fail_count, first_fail_row, first_fail_code = 0, None, 0

        assert t.reduce(Clean, row_offset=7) == (0, None, 0)
