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

extern "C" int sim_state_verify(const u64* cells, const u32* flags, u64 n, const u64* mpt, u64 n_mpt,
                                u32* status) {
    StateArgs a;
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
    for (u64 i = 0; i < n; i++) status[i] = state_check_row(a, i);
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
        default: r = fr_zero();
        }
        for (int k = 0; k < 4; k++) out[4 * i + k] = (u64)r.v[2 * k] | ((u64)r.v[2 * k + 1] << 32);
    }
}

// ---- EVM circuit ---------------------------------------------------------------------------
#include "../../zkevm_specs_amd/csrc/evm_circuit.hpp"

struct HostTable {
    ZkTable t;
    std::vector<u32> slots;
};
static void host_table(HostTable& h, const u64* cells, const u32* flags, u64 n, u32 ncells,
                       u64 (*hash_of)(const ZkTable&, u32)) {
    h.t.cells = cells;
    h.t.flags = flags;
    h.t.n = (u32)n;
    h.t.ncells = ncells;
    u32 mask = 0;
    build_index(h.slots, mask, (u32)n, hash_of, h.t);
}

extern "C" int sim_evm_verify(const u64* steps, u64 n_steps, const u64* rw, const u32* rw_flags, u64 n_rw,
                              const u64* bytecode, u64 n_bc, const u64* tx, const u32* tx_flags, u64 n_tx,
                              const u64* block, const u32* block_flags, u64 n_blk, u32 opts, u32* status) {
    EvmArgs a;
    a.steps.cells = steps;
    a.steps.flags = nullptr;
    a.steps.n = n_steps;
    HostTable trw, tbc, ttx, tblk;
    host_table(trw, rw, rw_flags, n_rw, RW_NCELLS, rw_key_hash);
    host_table(tbc, bytecode, nullptr, n_bc, BYTECODE_NCELLS, bc_key_hash);
    host_table(ttx, tx, tx_flags, n_tx, TX_NCELLS, tx_key_hash);
    host_table(tblk, block, block_flags, n_blk, BLOCK_NCELLS, blk_key_hash);
    a.rw = trw.t;
    a.bytecode = tbc.t;
    a.tx = ttx.t;
    a.block = tblk.t;
    a.perm = nullptr;
    a.n_pairs = (u32)(n_steps - 1);
    a.opts = opts;
    for (u64 i = 0; i + 1 < n_steps; i++) status[i] = evm_check_step(a, i);
    return 0;
}
