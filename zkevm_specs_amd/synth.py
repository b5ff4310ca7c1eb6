"""Synthetic witness generators for the BASELINE.json configurations (numpy, seeded).

Witnesses are produced directly in the C-ABI wire format (see wire.py) because building
2^16..2^20 rows out of Python `FQ`/`Word` objects is itself Python-bound (SURVEY.md §7
"hard parts").  Every generator yields a *valid* witness (all constraints satisfied); tests
tamper cells afterwards to exercise failures.
"""
import numpy as np

# ---- State circuit (config 2) ------------------------------------------
# Tag numbering: reference state_circuit.py:42-60.
T_START, T_MEMORY, T_STACK, T_STORAGE, T_CALLCTX, T_ACCOUNT, T_REFUND, T_AL_ACC, T_AL_STOR, T_LOG, T_RECEIPT = range(1, 12)

_STATE_MIX = [  # (tag, share of rows, mean rows per key group)
    (T_MEMORY, 0.35, 3.0),
    (T_STACK, 0.30, 2.5),
    (T_CALLCTX, 0.12, 1.5),
    (T_STORAGE, 0.06, 2.0),
    (T_ACCOUNT, 0.05, 2.0),
    (T_AL_ACC, 0.02, 2.0),
    (T_AL_STOR, 0.02, 2.0),
    (T_REFUND, 0.03, 4.0),
    (T_LOG, 0.03, 1.0),
    (T_RECEIPT, 0.02, 1.0),
]


def _u64s(rng, shape):
    return rng.integers(0, 2**63, size=shape, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=shape, dtype=np.uint64)


def synth_state_witness(n, seed=2):
    """Return (rows uint64[57, n, 4], flags uint32[n], mpt uint64[m, 12, 4]) for an n-row State
    circuit: row 0 is the Start row (lexicographic selector 0), the rest follows _STATE_MIX in
    strict lexicographic key order with unique rw_counters 1..n-1."""
    assert n >= 64
    rng = np.random.default_rng(seed)
    m = n - 1
    # --- decide per-tag row counts and key groups -------------------------------------
    shares = np.array([s for _, s, _ in _STATE_MIX])
    counts = np.floor(shares / shares.sum() * m).astype(np.int64)
    counts[-1] = min(counts[-1], 3 * 2048)  # TxReceipt: tx_id must stay within [1, 2^11] (11.3)
    counts[-2] = min(counts[-2], 1 << 20)
    counts[0] += m - counts.sum()
    tag = np.empty(m, dtype=np.int64)
    gid = np.empty(m, dtype=np.int64)     # global group id (unique key tuple)
    pos = np.empty(m, dtype=np.int64)     # position inside the group
    g_tag, g_size = [], []
    off = 0
    for (t, _, mean), c in zip(_STATE_MIX, counts):
        if c == 0:
            continue
        sizes = []
        left = int(c)
        while left > 0:
            s = 1 if mean <= 1.0 else int(min(left, 1 + rng.poisson(mean - 1.0)))
            if t == T_RECEIPT:
                s = min(left, 1)
            sizes.append(s)
            left -= s
        sizes = np.array(sizes, dtype=np.int64)
        ng = len(sizes)
        base = len(g_tag)
        g_tag += [t] * ng
        g_size += sizes.tolist()
        idx = np.repeat(np.arange(ng), sizes)
        tag[off : off + c] = t
        gid[off : off + c] = base + idx
        starts = np.concatenate([[0], np.cumsum(sizes)[:-1]])
        pos[off : off + c] = np.arange(c) - np.repeat(starts, sizes)
        off += c
    g_tag = np.array(g_tag)
    g_size = np.array(g_size)
    G = len(g_tag)

    # --- per-group keys (id, address[3 limbs], field_tag, storage_key[4 limbs]) -----------
    g_id = np.zeros(G, dtype=np.uint64)
    g_addr = np.zeros((G, 3), dtype=np.uint64)  # 160-bit: limbs 0,1 full, limb 2 = 32 bits
    g_ft = np.zeros(G, dtype=np.uint64)
    g_key = np.zeros((G, 4), dtype=np.uint64)

    def sel(t):
        return np.nonzero(g_tag == t)[0]

    def rand_addr160(k):
        a = np.zeros((k, 3), dtype=np.uint64)
        a[:, 0] = _u64s(rng, k)
        a[:, 1] = _u64s(rng, k)
        a[:, 2] = rng.integers(0, 2**32, size=k, dtype=np.uint64)
        return a

    # unique composite keys per tag are obtained by drawing from spaces where collisions are
    # resolved by construction (dense counters) or are astronomically unlikely (160/256-bit).
    s_ = sel(T_MEMORY)
    if len(s_):
        g_id[s_] = 1 + (np.arange(len(s_)) % 64).astype(np.uint64)
        g_addr[s_, 0] = (np.arange(len(s_)) // 64).astype(np.uint64)
    s_ = sel(T_STACK)
    if len(s_):  # per call: contiguous stack slots 1023 downwards (diff in {0,1} when sorted)
        per_call = 64
        g_id[s_] = 1 + (np.arange(len(s_)) // per_call).astype(np.uint64)
        g_addr[s_, 0] = (1023 - (per_call - 1) + (np.arange(len(s_)) % per_call)).astype(np.uint64)
    s_ = sel(T_CALLCTX)
    if len(s_):
        g_id[s_] = 1 + (np.arange(len(s_)) // 24).astype(np.uint64)
        g_ft[s_] = 1 + (np.arange(len(s_)) % 24).astype(np.uint64)
    s_ = sel(T_STORAGE)
    if len(s_):
        g_id[s_] = 1 + rng.integers(0, 16, size=len(s_)).astype(np.uint64)
        g_addr[s_] = rand_addr160(len(s_))
        g_key[s_] = _u64s(rng, (len(s_), 4))
    s_ = sel(T_ACCOUNT)
    if len(s_):
        a = rand_addr160((len(s_) + 3) // 4)
        g_addr[s_] = np.repeat(a, 4, axis=0)[: len(s_)]
        g_ft[s_] = 1 + (np.arange(len(s_)) % 4).astype(np.uint64)
    s_ = sel(T_REFUND)
    if len(s_):
        g_id[s_] = 1 + np.arange(len(s_)).astype(np.uint64)
    s_ = sel(T_AL_ACC)
    if len(s_):
        g_id[s_] = 1 + rng.integers(0, 16, size=len(s_)).astype(np.uint64)
        g_addr[s_] = rand_addr160(len(s_))
    s_ = sel(T_AL_STOR)
    if len(s_):
        g_id[s_] = 1 + rng.integers(0, 16, size=len(s_)).astype(np.uint64)
        g_addr[s_] = rand_addr160(len(s_))
        g_key[s_] = _u64s(rng, (len(s_), 4))
    s_ = sel(T_LOG)
    if len(s_):  # id=tx_id, address=log_id, field_tag in {1 Address, 2 Topic, 3 Data}, key=index
        k = np.arange(len(s_))
        g_id[s_] = 1 + (k // 96).astype(np.uint64)
        g_addr[s_, 0] = ((k // 12) % 8).astype(np.uint64)
        g_ft[s_] = 1 + ((k // 4) % 3).astype(np.uint64)
        g_key[s_, 0] = (k % 4).astype(np.uint64)
    s_ = sel(T_RECEIPT)
    if len(s_):  # tx ids 1.. with field tags 1,2,3 in order
        k = np.arange(len(s_))
        g_id[s_] = 1 + (k // 3).astype(np.uint64)
        g_ft[s_] = 1 + (k % 3).astype(np.uint64)

    # --- order groups the way the reference's gadget really compares them --------------------
    # keys_rwc_to_limbs_in_order (state_circuit.py:552-565) packs
    #   v = ((((tag*2^28 + id)*2^160 + address)*2^16 + field_tag)*2^32 + storage_key)*2^32 + rwc
    # i.e. the 256-bit storage key is shifted by only 32 bits and *adds into* the field_tag and
    # address limbs.  A valid witness must therefore be sorted by S = address*2^48 +
    # field_tag*2^32 + storage_key inside one (tag, id); keys stay < 2^200 and addresses
    # < 2^159 here so S never carries into id/tag.
    big = np.nonzero(g_key.any(axis=1))[0]
    g_key[big, 3] &= np.uint64(0xFF)
    g_addr[big, 2] &= np.uint64(0x7FFFFFFF)
    S = np.zeros((G, 4), dtype=np.uint64)
    S[:, 0] = (g_addr[:, 0] << np.uint64(48)) | (g_ft << np.uint64(32))
    S[:, 1] = (g_addr[:, 0] >> np.uint64(16)) | (g_addr[:, 1] << np.uint64(48))
    S[:, 2] = (g_addr[:, 1] >> np.uint64(16)) | (g_addr[:, 2] << np.uint64(48))
    S[:, 3] = g_addr[:, 2] >> np.uint64(16)
    for g in big.tolist():
        v = sum(int(S[g, k]) << (64 * k) for k in range(4)) + sum(int(g_key[g, k]) << (64 * k) for k in range(4))
        for k in range(4):
            S[g, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    order = np.lexsort((S[:, 0], S[:, 1], S[:, 2], S[:, 3], g_id, g_tag))
    rank = np.empty(G, dtype=np.int64)
    rank[order] = np.arange(G)
    row_order = np.lexsort((pos, rank[gid]))
    tag, gid, pos = tag[row_order], gid[row_order], pos[row_order]
    first = pos == 0
    last = np.concatenate([gid[1:] != gid[:-1], [True]])

    # rw_counters: unique 1..m, ascending inside each group
    rwc = (1 + rng.permutation(m)).astype(np.uint64)
    grp_sorted = np.lexsort((rwc, rank[gid]))
    rwc = rwc[grp_sorted]  # rows are already grouped contiguously in rank order

    # --- is_write / values ---------------------------------------------------------------
    is_write = rng.integers(0, 2, size=m).astype(np.uint64)
    is_write[first & (tag == T_STACK)] = 1
    is_write[tag == T_LOG] = 1
    is_write[(tag == T_RECEIPT)] = 0
    cand = np.zeros((m, 4), dtype=np.uint64)  # value lo[0:2], hi[2:4] as 64-bit limbs
    wide = np.isin(tag, [T_STACK, T_STORAGE, T_REFUND])
    cand[wide] = _u64s(rng, (int(wide.sum()), 4))
    mem = tag == T_MEMORY
    cand[mem, 0] = rng.integers(0, 256, size=int(mem.sum()), dtype=np.uint64)
    cc = tag == T_CALLCTX
    cand[cc, 0] = _u64s(rng, int(cc.sum()))
    acc = tag == T_ACCOUNT
    ft_row = g_ft[gid]
    cand[acc, 0] = _u64s(rng, int(acc.sum()))
    acc_wide = acc & (ft_row != 1)
    cand[acc_wide] = _u64s(rng, (int(acc_wide.sum()), 4))
    al = np.isin(tag, [T_AL_ACC, T_AL_STOR])
    cand[al, 0] = rng.integers(0, 2, size=int(al.sum()), dtype=np.uint64)
    lg = tag == T_LOG
    cand[lg, 0] = _u64s(rng, int(lg.sum()))
    lg_topic = lg & (ft_row == 2)
    cand[lg_topic] = _u64s(rng, (int(lg_topic.sum()), 4))
    rc = tag == T_RECEIPT
    cand[rc & (ft_row == 1), 0] = 1
    cand[rc & (ft_row == 2), 0] = (g_id[gid][rc & (ft_row == 2)] * np.uint64(21000))
    cand[rc & (ft_row == 3), 0] = 0
    # first access that is a read must see 0 (2.1, 5.2, 7.3, 8.2, 9.2); storage/account reads
    # see the committed value.
    first_read = first & (is_write == 0)
    zero_first = first_read & np.isin(tag, [T_MEMORY, T_CALLCTX, T_REFUND, T_AL_ACC, T_AL_STOR])
    cand[zero_first] = 0
    # committed (initial) value per group for Storage/Account
    g_init = np.zeros((G, 4), dtype=np.uint64)
    sa = np.isin(g_tag, [T_STORAGE, T_ACCOUNT])
    g_init[sa] = _u64s(rng, (int(sa.sum()), 4))
    nonce_g = (g_tag == T_ACCOUNT) & (g_ft == 1)
    g_init[nonce_g, 1:] = 0
    init = g_init[gid]
    sa_row = np.isin(tag, [T_STORAGE, T_ACCOUNT])
    fr_sa = first_read & sa_row
    cand[fr_sa] = init[fr_sa]
    # value of a read = value of the latest defining row (write, or first row) of its group
    defining = (is_write == 1) | first
    didx = np.maximum.accumulate(np.where(defining, np.arange(m), 0))
    value = cand[didx]

    # --- roots: +5 at every last access of a Storage/Account group (mock MPT, :903-933) ---
    upd = last & sa_row
    root = (3 + 5 * np.cumsum(upd)).astype(np.uint64)
    root_prev = np.concatenate([[3], root[:-1]]).astype(np.uint64)

    # --- assemble wire cells ------------------------------------------------------------
    cols = np.zeros((57, n, 4), dtype=np.uint64)
    R = slice(1, n)
    cols[0, R, 0] = rwc
    cols[1, R, 0] = is_write
    cols[2, 0, 0] = T_START
    cols[2, R, 0] = tag.astype(np.uint64)
    cols[3, R, 0] = g_id[gid]
    addr = g_addr[gid]
    cols[4, R, 0:3] = addr
    cols[5, R, 0] = ft_row
    key = g_key[gid]
    cols[6, R, 0:2] = key[:, 0:2]
    cols[7, R, 0:2] = key[:, 2:4]
    for k in range(10):  # 16-bit address limbs
        cols[8 + k, R, 0] = (addr[:, k // 4] >> np.uint64(16 * (k % 4))) & np.uint64(0xFFFF)
    for b in range(32):  # storage-key bytes
        cols[18 + b, R, 0] = (key[:, b // 8] >> np.uint64(8 * (b % 8))) & np.uint64(0xFF)
    cols[50, R, 0:2] = value[:, 0:2]
    cols[51, R, 0:2] = value[:, 2:4]
    cols[52, R, 0:2] = init[:, 0:2]
    cols[53, R, 0:2] = init[:, 2:4]
    cols[54, 0, 0] = 3
    cols[54, R, 0] = root
    cols[56, R, 0] = 1
    flags = np.zeros(n, dtype=np.uint32)
    word_valued = np.isin(tag, [T_STACK, T_STORAGE, T_REFUND]) | acc_wide | lg_topic
    flags[1:] = np.where(word_valued, 1, 0) | np.where(sa_row & word_valued, 2, 0)

    # --- MPT table rows for the updating rows -------------------------------------------
    ui = np.nonzero(upd)[0]
    mpt = np.zeros((len(ui), 12, 4), dtype=np.uint64)
    mpt[:, 0, 0:3] = addr[ui]
    u_tag, u_ft = tag[ui], ft_row[ui]
    vz = (value[ui] == 0).all(axis=1) & (init[ui] == 0).all(axis=1)
    proof = np.where(u_tag == T_STORAGE, np.where(vz, 4, 6), np.where(vz & (u_ft == 3), 4, u_ft))
    mpt[:, 1, 0] = proof.astype(np.uint64)
    mpt[:, 2, 0:2] = key[ui][:, 0:2]
    mpt[:, 3, 0:2] = key[ui][:, 2:4]
    mpt[:, 4, 0] = root[ui]
    mpt[:, 6, 0] = root_prev[ui]
    mpt[:, 8, 0:2] = value[ui][:, 0:2]
    mpt[:, 9, 0:2] = value[ui][:, 2:4]
    mpt[:, 10, 0:2] = init[ui][:, 0:2]
    mpt[:, 11, 0:2] = init[ui][:, 2:4]
    return cols, flags, mpt


def synth_state_ops(n, seed=2):
    """The ops behind an n-row State witness, in the wire form `zk_state_assign_open` takes
    (ops uint64[12, n, 4] column-major, flags uint32[n]), plus the rows / flags / MPT table the reference's
    `assign_state_circuit` / `mpt_table_from_ops` produce for them.  Storage and Account groups are read-only here
    (value == committed value): the reference's mock MPT update takes the FIRST op's value (state_circuit.py:921-929)
    while the circuit looks up the LAST access, so only such traces satisfy the circuit with the mock table."""
    cols, flags, mpt = synth_state_witness(n, seed)
    tag = cols[2, :, 0]
    sa = (tag == T_STORAGE) | (tag == T_ACCOUNT)
    cols[1, sa, 0] = 0
    cols[50, sa] = cols[52, sa]
    cols[51, sa] = cols[53, sa]
    mpt[:, 8] = mpt[:, 10]
    mpt[:, 9] = mpt[:, 11]
    ops = np.zeros((12, n, 4), dtype=np.uint64)
    ops[0:6] = cols[0:6]
    ops[6, :, 0:2] = cols[6, :, 0:2]
    ops[6, :, 2:4] = cols[7, :, 0:2]
    ops[7:11] = cols[50:54]
    ops[11] = cols[56]
    op_flags = (flags | np.where(tag == T_ACCOUNT, 4, 0)).astype(np.uint32)
    return ops, op_flags, cols, flags, mpt


# ---- Bytecode circuit (config 1) ---------------------------------------------------------------
EMPTY_HASH = 0xC5D2460186F7233C927E7DB2DCC703C0E500B653CA82273B7BFAD8045D85A470  # keccak256("")
_FR_P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def synth_bytecode_witness(codes, k, r, digest=None):
    """Witness of the Bytecode circuit for the byte strings `codes` padded to 2^k rows, following
    the row semantics of assign_bytecode_circuit (bytecode_circuit.py:104-167): one Header row
    + one Byte row per byte, value_rlc' = value_rlc * r + byte, push-data tracking, EMPTY_HASH
    padding headers.  Returns (rows uint64[12, 2^k, 4], keccak uint64[m, 5, 4]).

    `digest(code) -> int` supplies the code hash; the circuit only checks that (rlc, length, hash)
    is a keccak-table row, so the default is a synthetic 256-bit digest (no keccak needed here)."""
    import hashlib

    from .wire import rows_to_colmajor, rows_to_rowmajor

    if digest is None:
        digest = lambda c: int.from_bytes(hashlib.blake2b(c, digest_size=32).digest(), "big")  # noqa: E731
    n = 1 << k
    rows, keccak = [], set()
    for code in codes:
        h = digest(code)
        lo, hi = h & ((1 << 128) - 1), h >> 128
        if len(rows) < n:
            rows.append([0, 0, lo, hi, 1, 0, len(code), 0, 0, 0, len(code), 0])
        rlc, left = 0, 0
        for idx, b in enumerate(code):
            is_code = left == 0
            size = b - 0x5F if 0x60 <= b <= 0x7F else 0
            rlc = (rlc * r + b) % _FR_P
            if len(rows) < n:
                rows.append([0, 0, lo, hi, 2, idx, b, int(is_code), left, rlc, len(code), size])
            left = size if is_code else left - 1
        keccak.add((2, rlc, len(code), lo, hi))
    e_lo, e_hi = EMPTY_HASH & ((1 << 128) - 1), EMPTY_HASH >> 128
    while len(rows) < n:
        rows.append([0, 0, e_lo, e_hi, 1, 0, 0, 0, 0, 0, 0, 0])
    rows[0][0] = 1
    rows[n - 1][1] = 1
    return rows_to_colmajor(rows, 12), rows_to_rowmajor([list(x) for x in sorted(keccak)], 5)


# ---- Exp circuit ------------------------------------------------------------------------------------
def synth_exp_witness(n_rows, seed=5):
    """Square-and-multiply traces (ExpCircuit.add_event semantics, evm_circuit/typing.py:880-939)
    for random (base, exponent) events, padded with dummy rows (:941-962) to n_rows."""
    import random

    from .wire import rows_to_colmajor

    rng = random.Random(seed)
    M = (1 << 256) - 1
    lo_hi = lambda v: [v & ((1 << 128) - 1), v >> 128]  # noqa: E731
    rows = []
    ident = 1
    while True:
        base = rng.getrandbits(rng.choice([8, 64, 160, 256]))
        exponent = rng.getrandbits(rng.choice([2, 5, 16, 64, 256])) + 2
        steps = []

        def rec(e):
            if e == 0:
                return 1
            if e == 1:
                return base
            e1 = rec(e // 2)
            e2 = (e1 * e1) & M
            steps.append((e1, e1, e2))
            if e % 2 == 0:
                return e2
            ex = (base * e2) & M
            steps.append((e2, base, ex))
            return ex

        rec(exponent)
        steps.reverse()
        if len(rows) + len(steps) > n_rows:
            break
        e = exponent
        for i, (a, b, d) in enumerate(steps):
            q, odd = divmod(e, 2)
            rows.append([1, 1, ident, int(i == len(steps) - 1)] + lo_hi(base) + lo_hi(e) + lo_hi(d) + lo_hi(a) + lo_hi(b)
                        + [0, 0] + lo_hi(d) + lo_hi(q) + [odd])
            e = e // 2 if odd == 0 else e - 1
        ident += rng.randrange(1, 50)
    while len(rows) < n_rows:
        rows.append([1, 0, 0, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1])
    return rows_to_colmajor(rows, 21)


# ---- Tx circuit (config 4) ------------------------------------------------------------------------
_SECP_P = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEFFFFFC2F
_SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
_SECP_G = (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
           0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8)


def _secp_add(p1, p2):
    if p1 is None or p2 is None:
        return p2 if p1 is None else p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % _SECP_P == 0:
            return None
        m = 3 * x1 * x1 * pow(2 * y1, -1, _SECP_P) % _SECP_P
    else:
        m = (y2 - y1) * pow(x2 - x1, -1, _SECP_P) % _SECP_P
    x3 = (m * m - x1 - x2) % _SECP_P
    return x3, (m * (x1 - x3) - y1) % _SECP_P


def _secp_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = _secp_add(acc, pt)
        pt, k = _secp_add(pt, pt), k >> 1
    return acc


def synth_signatures(n, seed):
    """n valid secp256k1 (pk_x, pk_y, z, r, s) tuples for synthetic witnesses.  Private keys d_i = d_0 + i and nonces
    k_i = k_0 + i, so that every further key pair and signature costs two affine additions of G instead of two scalar
    multiplications (2^14 of them in seconds of pure Python).  Test data only."""
    import random

    rng = random.Random(seed)
    d, k = rng.randrange(1, _SECP_N - n), rng.randrange(1, _SECP_N - n)
    Q, R = _secp_mul(_SECP_G, d), _secp_mul(_SECP_G, k)
    out = []
    for _ in range(n):
        z = rng.getrandbits(256)
        r = R[0] % _SECP_N
        s = pow(k, -1, _SECP_N) * (z + r * d) % _SECP_N
        out.append((Q[0], Q[1], z, r, s))
        d, k, Q, R = d + 1, k + 1, _secp_add(Q, _SECP_G), _secp_add(R, _SECP_G)
    return out


def device_keccak_digests(r):
    """digests_of for synth_tx_witness: keccak-256 of the public keys through the device table builder (zk_keccak_table,
    mode 1 = KeccakTable.add rows: output Word(digest bytes), lo = bytes 0..15 little-endian)"""
    def digests(messages):
        from . import engine

        rows = engine.keccak_table(messages, r, engine.KECCAK_MODE_TABLE)
        return [rows[i, 3].tobytes()[:16] + rows[i, 4].tobytes()[:16] for i in range(len(messages))]

    return digests


from .flatten import ECDSA_STATUS_PENDING  # noqa: E402,F401  meta[:, 0] placeholder: the Tx kernel fails every unit until the ECDSA pass filled it


def synth_tx_witness(n_txs, r, seed=4, padding=0, signed=False, digests_of=None):
    """n_txs transaction slots (+ `padding` zero slots) for the Tx circuit's SignVerify path: 64-byte public
    keys, a synthetic 32-byte digest as pub_key_hash (the circuit checks keccak-table membership of
    (RLC(pk), 64, hash), not the hash function), address = low 20 bytes, message hashes.
    signed=False: random bytes as keys, ecdsa_status = 0 (the secp256k1 verdict as a pre-computed input column).
    signed=True: real key pairs and valid signatures (byte rows 7, 8 = r, s); ecdsa_status is left PENDING for the
    device ECDSA pass (engine.open_ecdsa(..., layout=ECDSA_LAYOUT_TX_UNITS, out_dev=meta, out_stride=4)).
    digests_of (signed=True only): callable(list of 64-byte public keys, big-endian x || y) -> list of 32-byte digests,
    e.g. keccak-256 through the device table builder (device_keccak_digests below); default: a synthetic digest.
    Returns the wire dict."""
    import hashlib
    import random

    from .wire import rows_to_colmajor, rows_to_rowmajor

    rng = random.Random(seed)
    bts, cells, meta, rows, flags = [], [], [], [], []
    keccak = {(0, 0, 0, 0, 0)}
    sigs = synth_signatures(n_txs + 1, seed) if signed else None  # the last one: the padding slots' dummy (tx_circuit.py:463-475)
    real = None
    if digests_of is not None:
        assert signed, "digests_of needs the key pairs up front (signed=True)"
        real = digests_of([sigs[i][0].to_bytes(32, "big") + sigs[i][1].to_bytes(32, "big") for i in range(n_txs)])
    for i in range(n_txs + padding):
        sig_r = sig_s = bytes(32)
        if i < n_txs:
            if signed:
                qx, qy, z, sr, ss = sigs[i]
                pk_x, pk_y = qx.to_bytes(32, "little"), qy.to_bytes(32, "little")
                msg = z.to_bytes(32, "little")  # the Tx chip's msg_hash_bytes are little-endian (tx_circuit.py:131)
                sig_r, sig_s = sr.to_bytes(32, "little"), ss.to_bytes(32, "little")
            else:
                pk_x = bytes(rng.getrandbits(8) for _ in range(32))  # little-endian limbs, as in the chip
                pk_y = bytes(rng.getrandbits(8) for _ in range(32))
                msg = bytes(rng.getrandbits(8) for _ in range(32))
            h = real[i] if real is not None else hashlib.blake2b(pk_x + pk_y, digest_size=32).digest()
            acc = 0
            for b in reversed(pk_y + pk_x):
                acc = (acc * r + b) % _FR_P
            h_lo, h_hi = int.from_bytes(h[:16], "little"), int.from_bytes(h[16:], "little")
            keccak.add((1, acc, 64, h_lo, h_hi))
            addr = int.from_bytes(h[-20:], "big")
            m_lo, m_hi = int.from_bytes(msg[:16], "little"), int.from_bytes(msg[16:], "little")
        else:
            pk_x = pk_y = h = msg = bytes(32)
            addr = m_lo = m_hi = 0
            if signed:
                qx, qy, z, sr, ss = sigs[n_txs]
                pk_x, pk_y, msg = qx.to_bytes(32, "little"), qy.to_bytes(32, "little"), z.to_bytes(32, "little")
                sig_r, sig_s = sr.to_bytes(32, "little"), ss.to_bytes(32, "little")
        bts.append([list(pk_x), list(pk_y), list(pk_x), list(pk_y), list(msg), list(msg), list(h), list(sig_r), list(sig_s)])
        cells.append([addr, m_lo, m_hi, 0, 0, 0, 0, 0])
        meta.append([ECDSA_STATUS_PENDING if signed else 0, 1, 0, 0])
        for tag in range(1, 13):  # Nonce .. TxSignHash (TxContextFieldTag, table.py:147-166)
            lo, hi, w = 0, 0, 0
            if tag == 4:
                lo = addr
            elif tag == 12:
                lo, hi, w = m_lo, m_hi, 1
            elif tag in (3, 7):
                w = 1
            rows.append([i + 1, tag, 0, lo, hi])
            flags.append(w)
    return {
        "bytes": np.array(bts, dtype=np.uint8), "cells": rows_to_colmajor(cells, 8),
        "meta": np.array(meta, dtype=np.uint32), "keccak": rows_to_rowmajor([list(k) for k in sorted(keccak)], 5),
        "tx_rows": rows_to_rowmajor(rows, 5), "tx_flags": np.array(flags, dtype=np.uint32),
    }


def synth_sig_witness(n, r, seed=4, digests_of=None):
    """The Sig circuit's rows (sig_circuit.py:7-62) for the SAME n signatures synth_tx_witness(signed=True, seed) signs: the
    chip's byte strings (public key little-endian, msg_hash_bytes = to_be_bytes() of the hash, util/ec.py:86-88; (r, s)
    little-endian), the table cells (recovered_addr = low 20 bytes of keccak(pk), msg_hash = Word(msg_hash_bytes), sig_v,
    sig_r / sig_s as lo / hi) and meta = (ecdsa_status PENDING for the device ECDSA pass, is_valid = 1, 0, the chip's v).
    v = 0 throughout: ECDSAVerifyChip.verify (util/ec.py:109-117) hands it to KeyAPI.Signature, which only checks v in {0, 1}.
    digests_of: as in synth_tx_witness (default: a synthetic digest — the circuit checks keccak-table membership)."""
    import hashlib

    from .wire import rows_to_colmajor, rows_to_rowmajor

    sigs = synth_signatures(n + 1, seed)[:n]
    real = digests_of([q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big") for q in sigs]) if digests_of is not None else None
    bts, cells, meta = [], [], []
    keccak = {(0, 0, 0, 0, 0)}
    m128 = (1 << 128) - 1
    for i, (qx, qy, z, sr, ss) in enumerate(sigs):
        pk_x, pk_y = qx.to_bytes(32, "little"), qy.to_bytes(32, "little")
        msg = z.to_bytes(32, "big")
        h = real[i] if real is not None else hashlib.blake2b(pk_x + pk_y, digest_size=32).digest()
        acc = 0
        for b in reversed(pk_y + pk_x):
            acc = (acc * r + b) % _FR_P
        keccak.add((1, acc, 64, int.from_bytes(h[:16], "little"), int.from_bytes(h[16:], "little")))
        bts.append([list(pk_x), list(pk_y), list(pk_x), list(pk_y), list(msg), list(msg), list(h), list(sr.to_bytes(32, "little")),
                    list(ss.to_bytes(32, "little"))])
        cells.append([int.from_bytes(h[-20:], "big"), int.from_bytes(msg[:16], "little"), int.from_bytes(msg[16:], "little"), 0,
                      sr & m128, sr >> 128, ss & m128, ss >> 128])
        meta.append([ECDSA_STATUS_PENDING, 1, 0, 0])
    return {
        "bytes": np.array(bts, dtype=np.uint8), "cells": rows_to_colmajor(cells, 8), "meta": np.array(meta, dtype=np.uint32),
        "keccak": rows_to_rowmajor([list(k) for k in sorted(keccak)], 5),
        "tx_rows": np.zeros((0, 5, 4), dtype=np.uint64), "tx_flags": np.zeros(0, dtype=np.uint32),
    }


# ---- Copy circuit: synthetic copy events (the inputs of zk_copy_assign) ------------------------------------------------
def synth_copy_events(target_rows, seed=6, max_len=192):
    """Random copy events of every source / destination kind the reference's gadgets produce — CODECOPY / EXTCODECOPY
    (Bytecode -> Memory), CALLDATACOPY in a root call (TxCalldata -> Memory), CALLDATACOPY / RETURNDATACOPY in an internal
    call (Memory -> Memory), LOG (Memory -> TxLog), SHA3 (Memory -> RlcAcc), RETURN of a deployment (Memory -> Bytecode) —
    about `target_rows` circuit rows in total, with reads running past `src_addr_end` (padding) in a third of the events.
    Returns dict(events uint64[n, 12, 4], flags, data uint16[], offsets, r, bytecode uint64[m, 6, 4], tx uint64[k, 5, 4],
    tx_flags): the bytecode / tx tables hold exactly the rows the copy rows look up; the RW rows come out of the assignment."""
    rng = np.random.default_rng(seed)
    P = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    r = int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) * int(rng.integers(1, 1 << 62)) % P
    n_codes, n_txs = 6, 5
    codes = []
    for c in range(n_codes):
        ln = int(rng.integers(200, 900))
        codes.append((int(rng.integers(1, 1 << 62)) | (c << 64), int(rng.integers(1, 1 << 62)), rng.integers(0, 256, ln, dtype=np.uint16),
                      rng.integers(0, 2, ln, dtype=np.uint16)))
    calldata = [rng.integers(0, 256, int(rng.integers(64, 600)), dtype=np.uint16) for _ in range(n_txs)]
    events, flags, data, offsets = [], [], [], [0]
    deployed = []  # bytecode rows of the codes that Memory -> Bytecode events write (RETURN of a deployment)
    rwc, rows = 1, 0
    while rows < target_rows:
        kind = int(rng.integers(0, 6))
        length = int(rng.integers(1, max_len))
        call_id, dst_call = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        dst_addr = int(rng.integers(0, 1 << 16))
        log_id = 0
        fl = 0
        if kind == 0:    # Bytecode -> Memory
            c = codes[int(rng.integers(0, n_codes))]
            avail = len(c[2])
            src_addr = int(rng.integers(0, avail))
            src_end = avail
            src = (c[0], c[1], 1)
            dst = (dst_call, 0, 2)
            fl = 1
            src_bytes = lambda a, c=c: int(c[2][a]) | (int(c[3][a]) << 8)  # noqa: E731
        elif kind == 1:  # TxCalldata -> Memory
            t = int(rng.integers(0, n_txs))
            avail = len(calldata[t])
            src_addr, src_end = int(rng.integers(0, avail)), avail
            src, dst = (t + 1, 0, 3), (dst_call, 0, 2)
            src_bytes = lambda a, t=t: int(calldata[t][a])  # noqa: E731
        else:            # Memory -> Memory / TxLog / RlcAcc / Bytecode
            src_addr = int(rng.integers(0, 1 << 16))
            src_end = src_addr + (length if rng.random() < 0.66 else int(rng.integers(0, length + 1)))
            mem = rng.integers(0, 256, max(src_end - src_addr, 0) + 1, dtype=np.uint16)
            src = (call_id, 0, 2)
            if kind == 2:
                dst = (dst_call, 0, 2)
            elif kind == 3:
                dst, log_id, dst_addr = (int(rng.integers(1, n_txs + 1)), 0, 4), int(rng.integers(0, 8)), int(rng.integers(0, 1 << 12))
            elif kind == 4:
                dst, dst_addr = (call_id, 0, 5), 0
            else:
                dst, dst_addr = (int(rng.integers(1, 1 << 62)), int(rng.integers(1, 1 << 62)), 1), 0
                fl = 2
            is_code = rng.integers(0, 2, len(mem), dtype=np.uint16) if kind == 5 else None
            src_bytes = lambda a, mem=mem, base=src_addr, ic=is_code: int(mem[a - base]) | ((int(ic[a - base]) << 8) if ic is not None else 0)  # noqa: E731
        if rng.random() < 0.33 and kind in (0, 1):
            length = max(length, src_end - src_addr + int(rng.integers(1, 16)))  # read past the end: padding rows
        n_real = max(0, min(length, src_end - src_addr))
        if kind == 5:
            deployed.append([dst[0], dst[1], 1, 0, 0, length])
            for i in range(length):
                b = src_bytes(src_addr + i) if i < n_real else 0
                deployed.append([dst[0], dst[1], 2, dst_addr + i, b >> 8, b & 0xFF])
        data.extend(src_bytes(src_addr + i) for i in range(n_real))
        offsets.append(len(data))
        events.append([src[0], src[1], src[2], dst[0], dst[1], dst[2], src_addr, src_end, dst_addr, length, log_id, rwc])
        flags.append(fl)
        rwc += (n_real if src[2] == 2 else 0) + (length if dst[2] in (2, 4) else 0)
        rows += 2 * length
    from .wire import rows_to_rowmajor

    bc_rows = []
    for c in codes:
        bc_rows.append([c[0], c[1], 1, 0, 0, len(c[2])])
        bc_rows.extend([c[0], c[1], 2, i, int(c[3][i]), int(c[2][i])] for i in range(len(c[2])))
    bc_rows.extend(deployed)
    tx_rows = [[t + 1, 13, i, int(calldata[t][i]), 0] for t in range(n_txs) for i in range(len(calldata[t]))]
    return {"events": rows_to_rowmajor(events, 12), "flags": np.array(flags, dtype=np.uint32), "data": np.array(data, dtype=np.uint16),
            "offsets": np.array(offsets, dtype=np.uint64), "r": r, "bytecode": rows_to_rowmajor(sorted(bc_rows), 6),
            "tx": rows_to_rowmajor(tx_rows, 5), "tx_flags": np.zeros(len(tx_rows), dtype=np.uint32), "n_rows": rows}
