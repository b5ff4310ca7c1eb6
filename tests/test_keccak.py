"""Keccak-256 / CREATE-address derivations: the Python restatement (oracle/keccak.py) against known
answers, and the device header (csrc/keccak.hpp, host build) against the restatement."""
import ctypes
import random

import numpy as np

from oracle import keccak as K


def _cell(v):
    return np.frombuffer(int(v).to_bytes(32, "little"), dtype="<u8").copy()


def test_known_answers():
    assert K.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert K.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # the first contract of the zero-nonce deployer 0x6ac7ea33f8831ea9dcc53393aaa88b25a785dbf0 (well-known RLP example)
    assert K.create_address(0x6AC7EA33F8831EA9DCC53393AAA88B25A785DBF0, 0) == 0xCD234A471B72BA2F1CCF0A70FCABA648A5EECD8D


def test_device_header_matches_restatement(hostsim):
    rng = random.Random(7)
    out = (ctypes.c_uint8 * 32)()
    for n in list(range(0, 136)) + [85, 56]:
        msg = bytes(rng.getrandbits(8) for _ in range(n))
        hostsim.sim_keccak256(msg, ctypes.c_int(n), out)
        assert bytes(out) == K.keccak256(msg), n
    vp = lambda x: ctypes.c_void_p(x.ctypes.data)  # noqa: E731
    res = np.zeros(4, dtype="<u8")
    for nonce in [0, 1, 0x7F, 0x80, 255, 256, 2**64 - 1, 2**64, 2**200 + 5, 21888242871839275222246405745257275088548364400416034343698204186575808495616]:
        a = rng.getrandbits(160)
        ca, cn = _cell(a), _cell(nonce)  # keep the buffers alive across the call
        hostsim.sim_create_address(vp(ca), vp(cn), vp(res))
        assert int.from_bytes(res.tobytes(), "little") == K.create_address(a, nonce), nonce
    for _ in range(50):
        a, salt, ch = rng.getrandbits(160), rng.getrandbits(256), rng.getrandbits(256)
        ca, cs, cc = _cell(a), _cell(salt), _cell(ch)
        hostsim.sim_create2_address(vp(ca), vp(cs), vp(cc), vp(res))
        assert int.from_bytes(res.tobytes(), "little") == K.create2_address(a, salt, ch)
