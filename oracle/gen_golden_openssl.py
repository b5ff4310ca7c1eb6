#!/usr/bin/env python3
"""secp256k1 ECDSA vectors made AND judged by OpenSSL (build container: OpenSSL 3.0.2 with the secp256k1 curve).

The reference's signature check is a call into third-party eth-keys 0.4.0 (util/ec.py:109-117, tx_circuit.py:147-158),
which is not installed; oracle/ecdsa_oracle.py and oracle/refshim/eth_keys restate it.  These vectors pin the restatement
(and, through it, the device kernel) to an implementation nobody in this repository wrote: keys and signatures come from
`openssl ecparam -genkey` / `openssl pkeyutl -sign`, and the verdict of EVERY case — the valid ones and the tampered
ones (wrong digest, r / s off by one, s -> N - s, another key) — is `openssl pkeyutl -verify`'s own answer on the
DER-re-encoded (r, s).

    python3 oracle/gen_golden_openssl.py    ->  tests/golden/ecdsa_openssl.npz
        sigs  uint8[n, 5, 32]  pk_x LE, pk_y LE, digest BE, r LE, s LE   (the packed layout of zk_ecdsa_verify)
        verdict uint8[n]       0 = OpenSSL verified it, 1 = OpenSSL rejected it
        kind  U16[n]           how the case was derived
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
N_KEYS = 40


def run(*args, data=None):
    return subprocess.run(args, input=data, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def der_int(v):
    b = v.to_bytes((v.bit_length() + 7) // 8 or 1, "big")
    if b[0] & 0x80:
        b = b"\x00" + b
    return b"\x02" + bytes([len(b)]) + b


def der_sig(r, s):
    body = der_int(r) + der_int(s)
    assert len(body) < 128
    return b"\x30" + bytes([len(body)]) + body


def parse_der_sig(der):
    assert der[0] == 0x30
    i = 2
    out = []
    for _ in range(2):
        assert der[i] == 0x02
        ln = der[i + 1]
        out.append(int.from_bytes(der[i + 2:i + 2 + ln], "big"))
        i += 2 + ln
    return out


def main():
    tmp = tempfile.mkdtemp(prefix="zk_openssl_")
    rng = np.random.default_rng(20240807)  # digests only; keys and nonces are OpenSSL's
    keys = []
    for k in range(N_KEYS):
        pem = os.path.join(tmp, f"k{k}.pem")
        assert run("openssl", "ecparam", "-name", "secp256k1", "-genkey", "-noout", "-out", pem).returncode == 0
        pub = run("openssl", "ec", "-in", pem, "-pubout", "-outform", "DER").stdout
        assert pub[-65] == 4
        keys.append((pem, int.from_bytes(pub[-64:-32], "big"), int.from_bytes(pub[-32:], "big")))

    def openssl_verdict(pem, digest, r, s):
        d, sg = os.path.join(tmp, "d.bin"), os.path.join(tmp, "s.der")
        open(d, "wb").write(digest)
        open(sg, "wb").write(der_sig(r, s))
        p = run("openssl", "pkeyutl", "-verify", "-inkey", pem, "-in", d, "-sigfile", sg)
        ok = b"Signature Verified Successfully" in p.stdout
        assert ok or b"Signature Verification Failure" in p.stdout + p.stderr, (p.stdout, p.stderr)
        return 0 if ok else 1

    cases, verdicts, kinds = [], [], []

    def emit(kind, key, digest, r, s):
        pem, x, y = key
        verdicts.append(openssl_verdict(pem, digest, r, s))
        kinds.append(kind)
        cases.append(x.to_bytes(32, "little") + y.to_bytes(32, "little") + digest + r.to_bytes(32, "little") + s.to_bytes(32, "little"))

    for k, key in enumerate(keys):
        for j in range(3):
            digest = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
            if k == 0 and j == 0:
                digest = bytes(32)  # z = 0
            if k == 1 and j == 0:
                digest = b"\xff" * 32  # z > N
            d = os.path.join(tmp, "d.bin")
            open(d, "wb").write(digest)
            p = run("openssl", "pkeyutl", "-sign", "-inkey", key[0], "-in", d)
            assert p.returncode == 0, p.stderr
            r, s = parse_der_sig(p.stdout)
            emit("valid", key, digest, r, s)
            if j == 0:
                emit("high_or_low_s", key, digest, r, N - s)  # the other s of the same signature: ECDSA accepts both
                bad = bytearray(digest)
                bad[int(rng.integers(0, 32))] ^= 1 << int(rng.integers(0, 8))
                emit("wrong_digest", key, bytes(bad), r, s)
            elif j == 1:
                emit("r_plus_1", key, digest, (r + 1) % N or 1, s)
                emit("s_minus_1", key, digest, r, (s - 1) % N or 1)
            else:
                emit("wrong_key", keys[(k + 1) % N_KEYS], digest, r, s)
    sigs = np.frombuffer(b"".join(cases), dtype=np.uint8).reshape(-1, 5, 32).copy()
    verdict = np.array(verdicts, dtype=np.uint8)
    kind = np.array(kinds)
    for kd in ("valid", "high_or_low_s"):
        assert not verdict[kind == kd].any(), kd
    for kd in ("wrong_digest", "r_plus_1", "s_minus_1", "wrong_key"):
        assert verdict[kind == kd].all(), kd
    out = os.path.join(ROOT, "tests", "golden", "ecdsa_openssl.npz")
    np.savez_compressed(out, sigs=sigs, verdict=verdict, kind=kind,
                        openssl=np.array(run("openssl", "version").stdout.decode().strip()))
    print(f"{len(verdict)} cases ({int((verdict == 0).sum())} verified, {int((verdict == 1).sum())} rejected) -> {out}")


if __name__ == "__main__":
    sys.exit(main())
