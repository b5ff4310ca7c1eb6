// CPU logic harness (TEST INFRASTRUCTURE): compiles the very same per-row device functions
// the HIP kernels call (csrc/*.hpp) with g++ -DZK_HOSTSIM and runs them in a plain loop, so
// the kernels' constraint logic can be checked against the oracle in the GPU-less build
// container.  It is NOT a CPU backend: the package never loads this library.
#include <vector>
#include "../../zkevm_specs_amd/csrc/state_circuit.hpp"

static void build_index(std::vector<u32>& slots, u32& mask, u32 n, u64 (*hash_of)(const ZkTable&, u32),
                        ZkTable& t) {
    u32 cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    mask = cap - 1;
    slots.assign(cap, ZK_EMPTY_SLOT);
    t.slots = slots.data();
    t.mask = mask;
    for (u32 r = 0; r < n; r++) {
        u32 s = (u32)hash_of(t, r) & mask;
        while (slots[s] != ZK_EMPTY_SLOT) s = (s + 1) & mask;
        slots[s] = r;
    }
}

extern "C" int sim_state_verify_range(const u64* cells, const u32* flags, u64 n, const u64* mpt, u64 n_mpt,
                                      u64 lo, u64 hi, u32* status);
extern "C" int sim_state_verify(const u64* cells, const u32* flags, u64 n, const u64* mpt, u64 n_mpt,
                                u32* status) {
    return sim_state_verify_range(cells, flags, n, mpt, n_mpt, 0, n, status);
}
extern "C" int sim_state_verify_range(const u64* cells, const u32* flags, u64 n, const u64* mpt, u64 n_mpt,
                                      u64 lo, u64 hi, u32* status) {
    StateArgs a = {};
    a.rows.cells = cells;
    a.rows.flags = flags;
    a.rows.n = n;
    a.mpt.cells = mpt;
    a.mpt.flags = nullptr;
    a.mpt.n = (u32)n_mpt;
    a.mpt.ncells = MPT_NCELLS;
    std::vector<u32> slots;
    u32 mask = 0;
    build_index(slots, mask, (u32)n_mpt, state_mpt_key_hash, a.mpt);
    for (u32& sv : slots)  // MPT slots carry a hash fingerprint (state_mpt_slot_value)
        if (sv != ZK_EMPTY_SLOT) sv = state_mpt_slot_value(a.mpt, sv, state_mpt_key_hash(a.mpt, sv));
    a.eval_lo = lo;
    a.eval_hi = hi;
    for (u64 i = lo; i < hi; i++) status[i] = state_check_row(a, i);
    return 0;
}

// Fr unit-test hooks
extern "C" void sim_fr_op(int op, const u64* a, const u64* b, u64* out, u64 n) {
    for (u64 i = 0; i < n; i++) {
        Fr x = fr_load(a + 4 * i), y = fr_load(b + 4 * i), r;
        switch (op) {
        case 0: r = fr_add(x, y); break;
        case 1: r = fr_sub(x, y); break;
        case 2: r = fr_mul(x, y); break;
        case 3: r = fr_mont(x, y); break;
        case 4: r = fr_neg(x); break;
        case 5: r = fr_inv(x); break;
        case 6: r = fr_div(x, y); break;
        default: r = fr_zero();
        }
        for (int k = 0; k < 4; k++) out[4 * i + k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
    }
}

// ---- EVM circuit ---------------------------------------------------------------------------
#include "../../zkevm_specs_amd/csrc/evm_circuit.hpp"
#include "../../zkevm_specs_amd/csrc/host_index.hpp"

struct HostTable {
    ZkTable t;
    std::vector<u32> slots;
};
static void host_table(HostTable& h, const u64* cells, const u32* flags, u64 n, u32 ncells,
                       u64 (*hash_of)(const ZkTable&, u32)) {
    static const u64 zero_row[64] = {0};
    static const u32 zero_flag[1] = {0};
    h.t.cells = n ? cells : zero_row;  // empty table: one readable zero row (as in the HIP library)
    h.t.flags = n ? flags : zero_flag;
    h.t.n = (u32)n;
    h.t.ncells = ncells;
    u32 mask = 0;
    build_index(h.slots, mask, (u32)n, hash_of, h.t);
}

extern "C" int sim_evm_verify(const u64* steps, u64 n_steps, const u64* rw, const u32* rw_flags, u64 n_rw,
                              const u64* bytecode, u64 n_bc, const u64* tx, const u32* tx_flags, u64 n_tx,
                              const u64* block, const u32* block_flags, u64 n_blk, const u64* copy, u64 n_copy,
                              const u64* keccak, u64 n_keccak, const u64* exp, u64 n_exp, const u64* aux, const u32* aux_kind,
                              const u64* wds, u64 n_wds, const u64* sig, u64 n_sig, const u64* ecc, u64 n_ecc, u32 aux_cells,
                              u32 opts, u32* status) {
    EvmArgs a = {};
    a.dyn = nullptr;
    a.step_recs = nullptr;
    a.defer_list = nullptr;
    a.defer_count = nullptr;
    a.steps = steps;
    a.n_steps = n_steps;
    HostTable trw, tbc, ttx, tblk, tcopy, tkeccak, texp;
    host_table(tcopy, copy, nullptr, n_copy, COPY_T_NCELLS, copy_key_hash);
    host_table(tkeccak, keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    host_table(texp, exp, nullptr, n_exp, EXP_T_NCELLS, expt_key_hash);
    a.aux = aux;
    a.aux_kind = aux_kind;
    a.aux_cells = aux_cells ? aux_cells : 2u;
    HostTable tsig, tecc;
    host_table(tsig, sig, nullptr, n_sig, SIG_T_NCELLS, sig_key_hash);
    host_table(tecc, ecc, nullptr, n_ecc, ECC_T_NCELLS, ecc_key_hash);
    a.sig = tsig.t;
    a.ecc = tecc.t;
    HostTable twd;
    host_table(twd, wds, nullptr, n_wds, 4, blk_key_hash);  // iterated in order: the index is unused
    a.withdrawals = twd.t;
    {
        const HostEvmAgg g = evm_aggregates_host(tx, tx_flags, n_tx, wds, n_wds);
        a.agg_max_txs = g.max_txs; a.agg_total_txs = g.total_txs; a.agg_invalid_txs = g.invalid_txs;
        a.agg_bad_invalid_rows = g.bad_invalid_rows; a.agg_total_wds = g.total_wds;
    }
    a.copy = tcopy.t;
    a.keccak = tkeccak.t;
    a.exp = texp.t;
    host_table(trw, rw, rw_flags, n_rw, RW_NCELLS, rw_key_hash);
    host_table(tbc, bytecode, nullptr, n_bc, BYTECODE_NCELLS, bc_key_hash);
    host_table(ttx, tx, tx_flags, n_tx, TX_NCELLS, tx_key_hash);
    host_table(tblk, block, block_flags, n_blk, BLOCK_NCELLS, blk_key_hash);
    a.rw = trw.t;
    a.bytecode = tbc.t;
    a.tx = ttx.t;
    a.block = tblk.t;
    a.perm = nullptr;
    a.prof = nullptr;
    a.n_pairs = (u32)(n_steps - 1);
    a.opts = opts & 3u;
    // bit 2 of opts: generic open-addressing indices only (no dense RW index / code directory)
    ZkRwMeta meta = rw_dense_meta_host(rw, n_rw);
    HostCodeDir dir;
    a.rw_dense = 0;
    a.rw_base = 0;
    a.rw_keys = nullptr;
    std::vector<u64> rw_keys;
    a.codes.n = 0;
    a.codes.packed = nullptr;
    if (!(opts & 4u)) {
        a.rw_dense = meta.dense;
        a.rw_base = meta.base;
        if (meta.dense) {  // packed key records, as rw_pack_kernel builds them on the device
            rw_keys.resize((size_t)n_rw * 4);
            for (u64 r = 0; r < n_rw; r++) { RwKey k = rw_pack_row(a.rw, (u32)r); for (int j = 0; j < 4; j++) rw_keys[4 * r + j] = k.w[j]; }
            a.rw_keys = rw_keys.data();
        }
        build_code_dir(bytecode, n_bc, dir);
        a.codes.entries = dir.entries.data();
        a.codes.slots = dir.slots.data();
        a.codes.mask = dir.mask;
        a.codes.n = (u32)dir.entries.size();
        a.codes.packed = dir.packed.data();
    }
    for (u64 i = 0; i + 1 < n_steps; i++) {  // as on the device: the hot and the cold instantiation split the states
        u32 c = evm_check_step<EVM_GROUP_ALL>(a, i);
        if (c == ZK_NOT_MINE) c = evm_check_step<EVM_GROUP_WARM>(a, i);
        if (c == ZK_NOT_MINE) c = evm_check_step<EVM_GROUP_COLD>(a, i);
        status[i] = c;
    }
    return 0;
}

// keccak KAT hooks: digest of a short message; CREATE / CREATE2 addresses
extern "C" void sim_keccak256(const uint8_t* msg, int len, uint8_t* out) { keccak256_block(msg, len, out); }
extern "C" void sim_create_address(const u64* address, const u64* nonce, u64* out) {
    Fr r = keccak_create_address(fr_load(address), fr_load(nonce));
    for (int k = 0; k < 4; k++) out[k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
}
extern "C" void sim_create2_address(const u64* address, const u64* salt, const u64* code_hash, u64* out) {
    Fr r = keccak_create2_address(fr_load(address), fr_load(salt), fr_load(code_hash));
    for (int k = 0; k < 4; k++) out[k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
}

// keccak table generation (csrc/keccak_table.hpp): rows [n][5][4], status [n]
#include "../../zkevm_specs_amd/csrc/keccak_table.hpp"
extern "C" int sim_keccak_table(const uint8_t* data, const u64* offsets, u64 n, const u64* r, u32 mode, u64* rows, u32* status) {
    std::vector<u64> rpow(KT_RPOW_ROWS * 4);
    kt_fill_rpow(fr_load(r), rpow.data());
    KeccakGenArgs g;
    g.data = data;
    g.offsets = offsets;
    g.n = n;
    g.rpow = rpow.data();
    g.rows = rows;
    g.mode = mode;
    g.long_list = nullptr;
    g.long_count = nullptr;
    for (u64 i = 0; i < n; i++) status[i] = keccak_table_row(g, i);
    return 0;
}

// 512/256 and 256/256 division KAT hooks: n (16 or 8 u32 limbs as u64 pairs), d -> q, r
extern "C" void sim_divmod(int wide, const u64* n, const u64* d, u64* q, u64* r, u64 count) {
    for (u64 i = 0; i < count; i++) {
        U256 dd = fr_load(d + 4 * i), rr;
        if (wide) {
            U512 nn, qq;
            for (int k = 0; k < 8; k++) { nn.v[2 * k] = (u32)n[8 * i + k]; nn.v[2 * k + 1] = (u32)(n[8 * i + k] >> 32); }
            u512_divmod(nn, dd, qq, rr, 512);
            for (int k = 0; k < 8; k++) q[8 * i + k] = (u64)qq.v[2 * k] | ((u64)qq.v[2 * k + 1] << 32);
        } else {
            U256 nn = fr_load(n + 4 * i), qq;
            u256_divmod(nn, dd, qq, rr);
            for (int k = 0; k < 4; k++) q[4 * i + k] = (u64)qq.v[2 * k] | ((u64)qq.v[2 * k + 1] << 32);
        }
        for (int k = 0; k < 4; k++) r[4 * i + k] = (u64)rr.v[2 * k] | ((u64)rr.v[2 * k + 1] << 32);
    }
}

// ---- Bytecode / Exp circuits -------------------------------------------------------------------
#include "../../zkevm_specs_amd/csrc/row_circuits.hpp"

extern "C" int sim_bytecode_verify(const u64* cells, u64 n, const u64* keccak, u64 n_keccak, const u64* r, u32* status) {
    BytecodeArgs a = {};
    a.r_mont = nullptr;
    a.rows.cells = cells;
    a.rows.flags = nullptr;
    a.rows.n = n;
    HostTable kt;
    host_table(kt, keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    a.keccak = kt.t;
    a.r = fr_load(r);
    for (u64 i = 0; i < n; i++) status[i] = bytecode_check_row(a, i);
    return 0;
}
extern "C" int sim_exp_verify(const u64* cells, u64 n, u32* status) {
    ExpArgs a = {};
    a.rows.cells = cells;
    a.rows.flags = nullptr;
    a.rows.n = n;
    for (u64 i = 0; i < n; i++) status[i] = exp_check_row(a, i);
    return 0;
}

// ---- Copy circuit ------------------------------------------------------------------------------
#include "../../zkevm_specs_amd/csrc/copy_circuit.hpp"

extern "C" int sim_copy_verify(const u64* cells, const u32* flags, u64 n, const u64* r, const u64* rw, const u32* rw_flags,
                               u64 n_rw, const u64* bytecode, u64 n_bc, const u64* tx, const u32* tx_flags, u64 n_tx,
                               u32 generic_index, u32* status) {
    CopyArgs a = {};
    a.rows.cells = cells;
    a.rows.flags = flags;
    a.rows.n = n;
    HostTable trw, tbc, ttx;
    host_table(trw, rw, rw_flags, n_rw, RW_NCELLS, rw_key_hash);
    host_table(tbc, bytecode, nullptr, n_bc, BYTECODE_NCELLS, bc_key_hash);
    host_table(ttx, tx, tx_flags, n_tx, TX_NCELLS, tx_key_hash);
    a.rw = trw.t;
    a.bytecode = tbc.t;
    a.tx = ttx.t;
    ZkRwMeta meta = rw_dense_meta_host(rw, n_rw);
    a.rw_meta = generic_index ? nullptr : &meta;
    a.r = fr_load(r);
    for (u64 i = 0; i < n; i++) status[i] = copy_check_row(a, i);
    return 0;
}

// ---- Tx / Sig circuits -----------------------------------------------------------------------
#include "../../zkevm_specs_amd/csrc/sign_circuit.hpp"

extern "C" int sim_sign_verify(const uint8_t* bytes, const u64* cells, const u32* meta, u64 n, const u64* keccak, u64 n_keccak,
                               const u64* tx_rows, const u32* tx_flags, u64 n_tx_rows, const u64* r, u32 is_sig, u32* status) {
    SignArgs a = {};
    a.bytes = bytes;
    a.cells.cells = cells;
    a.cells.flags = nullptr;
    a.cells.n = n;
    a.meta = meta;
    HostTable kt, tt;
    host_table(kt, keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    host_table(tt, tx_rows, tx_flags, n_tx_rows, TX_NCELLS, tx_key_hash);
    a.keccak = kt.t;
    a.tx_rows = tt.t;
    a.tx_rows.n = (u32)n_tx_rows;
    a.r = fr_load(r);
    std::vector<u64> rpow(64 * 4);
    sign_fill_rpow(a.r, rpow.data());
    a.rpow = rpow.data();
    a.is_sig = is_sig;
    for (u64 i = 0; i < n; i++) status[i] = sign_check_unit(a, i);
    return 0;
}

// ---- State witness assignment: the per-op device functions of csrc/state_assign.hpp in a plain loop
#include "../../zkevm_specs_amd/csrc/state_assign.hpp"
extern "C" int sim_state_assign(const u64* ops, const u32* op_flags, u64 n, u64* rows, u32* row_flags, u64* mpt,
                                u64* n_mpt, u32* status) {
    AssignArgs a = {};
    a.ops = ops; a.op_flags = op_flags; a.n = n; a.rows = rows; a.row_flags = row_flags; a.mpt = mpt;
    u32 cap = 16;
    while (cap < 2 * n + 2) cap <<= 1;
    std::vector<u32> slots(cap, ZK_EMPTY_SLOT), first(n, ASG_NONE), rank(n, 0);
    a.slots = slots.data(); a.mask = cap - 1; a.first = first.data(); a.rank = rank.data();
    a.nb = 0; a.blk_cnt = nullptr; a.blk_next = nullptr;
    // ops are inserted in REVERSE order so that the "smallest index wins" rule is what makes the result right
    for (u64 i = n; i-- > 0;)
        if (asg_has_key(asg_slot(a, ASG_TAG, i))) asg_insert(a, (u32)i);
    u32 r = 0;
    for (u64 i = 0; i < n; i++) {
        if (!asg_has_key(asg_slot(a, ASG_TAG, i))) continue;
        first[i] = asg_find_first(a, (u32)i);
        if (first[i] == (u32)i) { rank[i] = r; asg_write_mpt(a, i, r); r++; }
    }
    *n_mpt = r;
    u32 nxt = ASG_NONE;
    for (u64 i = n; i-- > 0;) {
        const u64 root = 3ull + 5ull * (nxt == ASG_NONE ? r : rank[first[nxt]]);
        status[i] = asg_write_row(a, i, root, first[i] == (u32)i);
        if (first[i] != ASG_NONE) nxt = (u32)i;
    }
    return 0;
}

// ---- secp256k1 ECDSA: the per-signature device function of csrc/secp256k1.hpp in a plain loop
#include "../../zkevm_specs_amd/csrc/secp256k1.hpp"
extern "C" int sim_ecdsa_verify(const uint8_t* bytes, u32 layout, const u32* v, u32 v_stride, u64 n, u32* status) {
    static const u32 OFF[2][5] = {{0, 32, 64, 96, 128}, {64, 96, 160, 224, 256}};
    EcdsaArgs a = {};
    a.bytes = bytes; a.stride = layout ? 288 : 160; a.v = v; a.v_stride = v_stride; a.n = n; a.first = 0; a.out = nullptr; a.out_stride = 0;
    ecdsa_single_batch(a);
    a.msg_be = layout != 1u;
    for (int k = 0; k < 5; k++) a.off[k] = OFF[layout ? 1 : 0][k];
    std::vector<u32> tab(15 * 24 * 2);
    a.qtab = tab.data(); a.qtab_lanes = 1; a.lanes_per_sig = 1;
    for (u64 i = 0; i < n; i++) status[i] = ecdsa_verify_one(a, i, tab.data(), 1);
    return 0;
}
// the lane-pair form of the kernel (k_ecdsa.hip, L = 2) in a plain loop: two partial sums, added at the end
extern "C" int sim_ecdsa_verify_pairs(const uint8_t* bytes, u32 layout, const u32* v, u32 v_stride, u64 n, u32* status) {
    static const u32 OFF[2][5] = {{0, 32, 64, 96, 128}, {64, 96, 160, 224, 256}};
    EcdsaArgs a = {};
    a.bytes = bytes; a.stride = layout ? 288 : 160; a.v = v; a.v_stride = v_stride; a.n = n; a.first = 0; a.out = nullptr; a.out_stride = 0;
    ecdsa_single_batch(a);
    a.msg_be = layout != 1u;
    for (int k = 0; k < 5; k++) a.off[k] = OFF[layout ? 1 : 0][k];
    std::vector<u32> tab(15 * 24 * 2);
    a.qtab = tab.data(); a.qtab_lanes = 2; a.lanes_per_sig = 2;
    for (u64 i = 0; i < n; i++) {
        EcdsaPrep pr0, pr1;
        const u32 st0 = ecdsa_prepare(a, i, pr0, true), st1 = ecdsa_prepare(a, i, pr1, false);
        if (st0 != ECDSA_PENDING) { status[i] = st0; continue; }
        if (st1 != ECDSA_PENDING) return -1;
        SpPoint p0 = ecdsa_partial(pr0, 0, 0, tab.data(), 2);
        const SpPoint p1 = ecdsa_partial(pr1, 1, 1, tab.data() + 1, 2);
        sp_add_ip(p0, p1);
        status[i] = ecdsa_verdict(pr0, p0);
    }
    return 0;
}

// ---- Bytecode witness assignment: the per-piece device functions of csrc/bytecode_assign.hpp in plain loops
#include "../../zkevm_specs_amd/csrc/bytecode_assign.hpp"
extern "C" int sim_bytecode_assign(const u64* in_rows, u64 n_rows, const u64* offsets, const u64* lengths, u64 n_codes, u32 k,
                                   const u64* r4, u64* rows_out) {
    BcaArgs a = {};
    a.in_rows = in_rows; a.offsets = offsets; a.lengths = lengths; a.n_in = n_rows; a.n_codes = n_codes; a.n_out = 1ull << k;
    std::vector<BcaChunk> chunks;
    std::vector<u32> c0(n_codes + 1, 0);
    for (u64 j = 0; j < n_codes; j++) {
        c0[j] = (u32)chunks.size();
        for (u64 g = offsets[j]; g < offsets[j + 1]; g += BCA_CHUNK) {
            BcaChunk c;
            c.code = (u32)j; c.start = (u32)g;
            c.count = (u32)((offsets[j + 1] - g < BCA_CHUNK) ? offsets[j + 1] - g : BCA_CHUNK);
            c.first = g == offsets[j] ? 1u : 0u;
            chunks.push_back(c);
        }
    }
    c0[n_codes] = (u32)chunks.size();
    Fr r;
    for (int q = 0; q < 4; q++) { r.v[2 * q] = (u32)r4[q]; r.v[2 * q + 1] = (u32)(r4[q] >> 32); }
    std::vector<u64> rpow(BCA_RPOW_ROWS * 4), acc(chunks.size() * 4 + 4), cin(chunks.size() * 4 + 4), rlc(n_rows * 4 + 4);
    std::vector<u32> cm(chunks.size() + 1), rc(n_rows + 1), cstate(chunks.size() + 1);
    std::vector<uint8_t> track(2 * n_rows + 2), cmap((chunks.size() + 1) * BCA_MAP_STRIDE);
    bca_fill_rpow(r, rpow.data());
    a.rpow = rpow.data(); a.chunks = chunks.data(); a.code_chunk0 = c0.data(); a.n_chunks = chunks.size();
    a.track = track.data(); a.chunk_acc = acc.data(); a.chunk_m = cm.data(); a.chunk_in = cin.data(); a.rlc = rlc.data();
    a.row_code = rc.data(); a.rows = rows_out; a.chunk_map = cmap.data(); a.chunk_state = cstate.data();
    for (u64 c = 0; c < chunks.size(); c++) bca_chunk(a, c);
    for (u64 j = 0; j < n_codes; j++) bca_prefix_code(a, j);
    for (u64 c = 0; c < chunks.size(); c++) bca_rlc_chunk(a, c);
    for (u64 i = 0; i < a.n_out; i++) bca_write_row(a, i);
    return 0;
}
// secp256k1 field products (unit-test hook): which = 0 base field (plain residues), 1 scalar field (Montgomery product)
extern "C" void sim_secp_mul(int which, const u64* a, const u64* b, u64* out, u64 n) {
    for (u64 i = 0; i < n; i++) {
        Fr x, y;
        for (int q = 0; q < 4; q++) {
            x.v[2 * q] = (u32)a[4 * i + q]; x.v[2 * q + 1] = (u32)(a[4 * i + q] >> 32);
            y.v[2 * q] = (u32)b[4 * i + q]; y.v[2 * q + 1] = (u32)(b[4 * i + q] >> 32);
        }
        const Fr r = which == 0 ? sp_mont<SecpP>(x, y) : (which == 2 ? sp_sqr_p(x) : (which == 3 ? sp_inv_n_binary(x) : (which == 4 ? sp_inv_n_safegcd(x) : sp_mont<SecpN>(x, y))));  // 3 / 4: x^-1 mod N, binary / division steps
        for (int q = 0; q < 4; q++) out[4 * i + q] = (u64)r.v[2 * q] | ((u64)r.v[2 * q + 1] << 32);
    }
}

// ---- Copy-circuit witness assignment: the per-piece device functions of csrc/copy_assign.hpp in plain loops
#include "../../zkevm_specs_amd/csrc/copy_assign.hpp"
extern "C" int sim_copy_assign(const u64* events, const u32* flags, u64 n, const uint16_t* data, const u64* offsets, const u64* r4,
                               u64* rows, u32* row_flags, u64* table, u64* rw, u32* rw_flags, u64 n_rows) {
    std::vector<CpaEvent> ev(n);
    std::vector<u64> row0(n + 1, 0);
    std::vector<CpaChunk> chunks;
    u64 nr = 0, nt = 0, nw = 0, nl = 0;
    for (u64 e = 0; e < n; e++) {
        CpaEvent& x = ev[e];
        const u64* c = events + e * CPA_EV_NCELLS * 4;
        x.src_tag = (u32)c[2 * 4]; x.dst_tag = (u32)c[5 * 4]; x.src_addr = c[6 * 4]; x.src_end = c[7 * 4]; x.dst_addr = c[8 * 4];
        x.length = c[9 * 4]; x.log_id = c[10 * 4]; x.rwc = c[11 * 4]; x.flags = flags[e];
        const u64 n_real = cpa_n_real(x);
        x.row0 = nr; x.rw0 = nw; x.data0 = offsets[e]; x.table_idx = x.length ? (u32)nt++ : CPA_NONE;
        x.rlc0 = 0; x.chunk0 = (u32)chunks.size(); x.n_chunks = 0;
        if (x.dst_tag == CPA_RLC_ACC) {
            x.rlc0 = nl; nl += x.length;
            for (u64 g = 0; g < x.length; g += CPA_CHUNK) {
                CpaChunk ch; ch.event = (u32)e; ch.start = (u32)g; ch.count = (u32)(x.length - g < CPA_CHUNK ? x.length - g : CPA_CHUNK); ch.pad = 0;
                chunks.push_back(ch); x.n_chunks++;
            }
        }
        row0[e] = nr;
        nr += 2 * x.length;
        nw += (x.src_tag == CPA_MEMORY ? n_real : 0) + ((x.dst_tag == CPA_MEMORY || x.dst_tag == CPA_TX_LOG) ? x.length : 0);
    }
    row0[n] = nr;
    if (nr != n_rows) return -1;
    std::vector<u64> rpow(CPA_RPOW_ROWS * 4), chunk_acc(chunks.size() * 4 + 4), chunk_in(chunks.size() * 4 + 4), ev_rlc(n * 4 + 4), rlc(nl * 4 + 4);
    Fr r;
    for (int q = 0; q < 4; q++) { r.v[2 * q] = (u32)r4[q]; r.v[2 * q + 1] = (u32)(r4[q] >> 32); }
    cpa_fill_rpow(r, rpow.data());
    CpaArgs a = {};
    a.events = events; a.ev = ev.data(); a.row0 = row0.data(); a.n_events = n; a.n_rows = nr; a.data = data; a.rpow = rpow.data();
    a.chunks = chunks.data(); a.n_chunks = chunks.size(); a.chunk_acc = chunk_acc.data(); a.chunk_in = chunk_in.data(); a.ev_rlc = ev_rlc.data();
    a.rlc = rlc.data(); a.rows = rows; a.row_flags = row_flags; a.table = table; a.rw = rw; a.rw_flags = rw_flags;
    for (u64 c = 0; c < chunks.size(); c++) cpa_chunk(a, c);
    for (u64 e = 0; e < n; e++) cpa_prefix_event(a, e);
    for (u64 j = 0; j < nr; j++) cpa_write_row(a, j);
    return 0;
}

// ---- Public-inputs circuit
#include "../../zkevm_specs_amd/csrc/pi_circuit.hpp"
extern "C" int sim_pi_verify(const u64* rows, u64 n, const u64* keccak, u64 n_keccak, const u64* gas, u64 n_gas, u64 circuit_len,
                             const u64* keccak_rand, const u64* byte_pow_base, u32* status) {
    PiArgs a = {};
    a.rows.cells = rows; a.rows.flags = nullptr; a.rows.n = n;
    HostTable tk, tg;
    host_table(tk, keccak, nullptr, n_keccak, KECCAK_NCELLS, keccak_key_hash);
    host_table(tg, gas, nullptr, n_gas, PI_GAS_NCELLS, pi_gas_key_hash);
    a.keccak = tk.t; a.gas = tg.t;
    Fr kr, bp;
    for (int k = 0; k < 4; k++) {
        kr.v[2 * k] = (u32)keccak_rand[k]; kr.v[2 * k + 1] = (u32)(keccak_rand[k] >> 32);
        bp.v[2 * k] = (u32)byte_pow_base[k]; bp.v[2 * k + 1] = (u32)(byte_pow_base[k] >> 32);
    }
    a.keccak_rand_m = fr_to_mont(kr); a.byte_pow_base_m = fr_to_mont(bp);
    a.circuit_len = fr_from_u64(circuit_len);
    for (u64 i = 0; i < n; i++) status[i] = pi_check_row(a, i);
    return 0;
}

// wide_witness (csrc/bigz.hpp): the unbounded-integer witness stages, one call per operand tuple.  x: 8 cells (x0.lo, x0.hi, ...,
// x3.hi) per tuple; out: 4 x 4 u64 values, 4 flags, b0, b1 per tuple.
extern "C" void sim_wide_witness(u32 op, const u64* x, u64* out, u32* flags, u64 count) {
    for (u64 i = 0; i < count; i++) {
        Fr c[8];
        for (int k = 0; k < 8; k++) c[k] = fr_load(x + (8 * i + k) * 4);
        const WideRes R = wide_witness(op, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
        for (int o = 0; o < 4; o++)
            for (int k = 0; k < 4; k++) out[(4 * i + o) * 4 + k] = (u64)R.o[o].v[2 * k] | ((u64)R.o[o].v[2 * k + 1] << 32);
        for (int o = 0; o < 4; o++) flags[6 * i + o] = R.fl[o];
        flags[6 * i + 4] = R.b0;
        flags[6 * i + 5] = R.b1;
    }
}

// PI circuit copy constraints (csrc/pi_circuit.hpp pi_copy_check)
extern "C" int sim_pi_copy_verify(const u64* cells, const uint8_t* bytes, const u32* lens, u64 n, u32* status) {
    PiCopyArgs a = {};
    a.cells = cells; a.bytes = bytes; a.lens = lens; a.n = n;
    for (u64 i = 0; i < n; i++) status[i] = pi_copy_check(a, i);
    return 0;
}
