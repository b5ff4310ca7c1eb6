// Host-side construction of the bytecode directory (ZkCodeDir) from a row-major bytecode table.
// Shared by libzkevm_hip.so (session open) and the CPU logic harness.
#pragma once
#include <stdint.h>
#include <string.h>
#include <map>
#include <array>
#include <vector>
#include "common.hpp"

struct HostCodeDir {
    std::vector<uint16_t> packed;  // per bytecode-table row, see ZkCodeDir::packed
    std::vector<ZkCodeEntry> entries;
    std::vector<u32> slots;
    u32 mask = 0;
};

static inline u64 host_code_hash_key(const u64* lo, const u64* hi) {
    Fr a, b;
    for (int k = 0; k < 4; k++) {
        a.v[2 * k] = (u32)lo[k]; a.v[2 * k + 1] = (u32)(lo[k] >> 32);
        b.v[2 * k] = (u32)hi[k]; b.v[2 * k + 1] = (u32)(hi[k] >> 32);
    }
    // same mixing as zk_code_hash_key (device) — restated on u64 limbs for host code
    auto mix = [](u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; };
    auto cell = [&](u64 h, const Fr& c) {
        u64 p = ((u64)c.v[0] | ((u64)c.v[1] << 32)) ^ (((u64)c.v[2] | ((u64)c.v[3] << 32)) * 0x9e3779b97f4a7c15ull);
        u64 q = ((u64)c.v[4] | ((u64)c.v[5] << 32)) ^ (((u64)c.v[6] | ((u64)c.v[7] << 32)) * 0xc2b2ae3d27d4eb4full);
        return mix(h ^ p ^ (q << 1) ^ (q >> 63)) + 0x632be59bd9b4e019ull;
    };
    return cell(cell(0xc0de5u, a), b);
}

// rows: [n][6][4] (hash lo, hi, field_tag, index, is_code, value).  A code is "regular" when its
// rows are exactly one Header row (tag 1, index 0) plus Byte rows (tag 2) whose indices are
// 0..k-1 stored contiguously in increasing order — what Bytecode.table_assignments produces
// (evm_circuit/typing.py:390-405) after the flattener's sort.
static inline void build_code_dir(const u64* rows, u64 n, HostCodeDir& out) {
    struct Info { std::vector<u32> idx; };
    std::map<std::array<u64, 8>, Info> groups;
    for (u64 r = 0; r < n; r++) {
        std::array<u64, 8> key;
        memcpy(key.data(), rows + r * 24, 64);
        groups[key].idx.push_back((u32)r);
    }
    auto small = [&](u64 r, int c, u64& v) {  // cell value if it fits 64 bits
        const u64* p = rows + (r * 6 + c) * 4;
        v = p[0];
        return (p[1] | p[2] | p[3]) == 0;
    };
    out.packed.assign(n, 0);
    for (u64 r = 0; r < n; r++) {
        u64 is_code, value;
        if (small(r, 4, is_code) && small(r, 5, value) && is_code < 2 && value < 256)
            out.packed[r] = (uint16_t)(0x8000u | (is_code << 8) | value);
    }
    for (auto& kv : groups) {
        ZkCodeEntry e;
        memcpy(e.hash, kv.first.data(), 64);
        e.header_row = e.byte_base = e.n_bytes = 0;
        e.regular = 0;
        e.header_value = 0;
        e.header_ok = 0;
        e.pad = 0;
        const std::vector<u32>& ix = kv.second.idx;
        int headers = 0;
        bool ok = true;
        u32 first_byte = 0xffffffffu, n_bytes = 0;
        for (u32 r : ix) {
            u64 tag, index;
            if (!small(r, 2, tag) || !small(r, 3, index)) { ok = false; break; }
            if (tag == 1) {
                if (index != 0) { ok = false; break; }
                headers++;
                e.header_row = r;
            } else if (tag == 2) {
                if (first_byte == 0xffffffffu) first_byte = r;
                if (r != first_byte + n_bytes || index != n_bytes) { ok = false; break; }
                n_bytes++;
            } else {
                ok = false;
                break;
            }
        }
        if (ok && headers == 1) {
            u64 hv, hc;
            if (small(e.header_row, 5, hv) && small(e.header_row, 4, hc) && hc == 0) { e.header_value = hv; e.header_ok = 1; }
            e.regular = 1;
            e.byte_base = first_byte == 0xffffffffu ? 0 : first_byte;
            e.n_bytes = n_bytes;
        }
        out.entries.push_back(e);
    }
    u32 cap = 16;
    while (cap < 2 * out.entries.size() + 2) cap <<= 1;
    out.mask = cap - 1;
    out.slots.assign(cap, ZK_EMPTY_SLOT);
    for (u32 k = 0; k < out.entries.size(); k++) {
        u32 s = (u32)host_code_hash_key(out.entries[k].hash, out.entries[k].hash + 4) & out.mask;
        while (out.slots[s] != ZK_EMPTY_SLOT) s = (s + 1) & out.mask;
        out.slots[s] = k;
    }
}

// rw rows [n][14][4]: dense iff rw_counter of row i == rw_counter of row 0 + i (all < 2^64)
static inline ZkRwMeta rw_dense_meta_host(const u64* rows, u64 n) {
    ZkRwMeta m;
    m.dense = 0;
    m.pad = 0;
    m.base = 0;
    if (n == 0) return m;
    const u64 base = rows[0];
    bool ok = (rows[1] | rows[2] | rows[3]) == 0 && base + n >= base;
    for (u64 r = 0; ok && r < n; r++) {
        const u64* p = rows + r * 14 * 4;
        ok = p[0] == base + r && (p[1] | p[2] | p[3]) == 0;
    }
    m.dense = ok ? 1u : 0u;
    m.base = base;
    return m;
}

// Whole-table aggregates of EndBlock's last step (end_block.py:55-91) from the wire tables:
// tx rows [n][5][4] (tx_id, tag, index, value lo, hi) + flags (bit0 value.is_word); withdrawals [m][4][4].
struct HostEvmAgg {
    u32 max_txs = 0, total_txs = 0, invalid_txs = 0, bad_invalid_rows = 0, total_wds = 0;
};
static inline HostEvmAgg evm_aggregates_host(const u64* tx, const u32* tx_flags, u64 n_tx, const u64* wds, u64 n_wds) {
    HostEvmAgg g;
    auto is_small = [](const u64* c, u64 v) { return c[0] == v && (c[1] | c[2] | c[3]) == 0; };
    auto is_zero = [](const u64* c) { return (c[0] | c[1] | c[2] | c[3]) == 0; };
    for (u64 r = 0; r < n_tx; r++) {
        const u64* row = tx + r * 5 * 4;
        if (is_small(row + 4, 4)) {  // TxContextFieldTag.CallerAddress
            g.max_txs++;
            if (!(is_zero(row + 12) && is_zero(row + 16))) g.total_txs++;
        } else if (is_small(row + 4, 10)) {  // TxContextFieldTag.TxInvalid
            if (tx_flags && (tx_flags[r] & 1u)) g.bad_invalid_rows = 1;  // .value.value() asserts on a word
            if (is_small(row + 12, 1)) g.invalid_txs++;
        }
    }
    for (u64 r = 0; r < n_wds; r++)
        if (!is_zero(wds + (r * 4 + 3) * 4)) g.total_wds++;
    return g;
}
