// Device-side construction of the bytecode directory (ZkCodeDir, common.hpp) from the row-major bytecode table: the
// directory host_index.hpp's build_code_dir builds on the CPU (which the CPU logic harness keeps using), so that opening
// an EVM session over device-resident tables moves no table data to the host and nothing is read back.
//
// A code (= the rows of one bytecode hash) is served by the directory ("regular") when its rows are ONE contiguous run:
// a Header row (tag 1, index 0) followed by its Byte rows (tag 2) with indices 0..k-1 in increasing order — what
// Bytecode.table_assignments produces (evm_circuit/typing.py:390-405) after the flattener's sort by (hash, tag, index).
// Regular codes are addressed directly (header_row / byte_base + index); anything else — a hash whose rows come in several
// runs, a header elsewhere, gaps — goes through the generic open-addressing index, which gives the same answers
// (tests: generic_index=True parity).  Entry order is whatever the atomics produce; lookups reach an entry by hash.
//
// Round 3: two device functions, run as block ranges of the session-open launches (zkevm_hip.hip):
//   dirb_events_row    every row writes its packed u16 record and compares itself with the row before it; only the rows
//                      where something happens touch shared state — the first row of a run (claims / finds its hash's
//                      slot, counts the run, records the run's first row and the previous run's last row), the last table
//                      row, and rows that break the pattern (mark their hash irregular).  43,913 rows of 16 contracts are
//                      ~50 atomics instead of 43,913 CAS / atomicMin on 16 words (65 us in round 2).
//   dirb_finalize_entry one lane per directory entry: decides "regular", fills the entry, inserts it into the directory's
//                      slot table; the entry count and slot mask go to the session's EvmDyn block (no host read-back).
// The directory has a fixed capacity (DIRB_MAX_ENTRIES codes): a table with more distinct hashes gets no directory and
// every bytecode lookup goes through the generic index — correct, only slower.
#pragma once
#include "common.hpp"

#define DIRB_MAX_ENTRIES 4096u
#define DIRB_SMALL_SLOTS 16384u  // capacity of the directory's own slot table (>= 2 * DIRB_MAX_ENTRIES + 2, power of two)

struct DirBuild {
    const u64* rows;   // [n][6][4]: hash lo, hi, field_tag, index, is_code, value
    u32 n;
    // open addressing over the distinct hashes, keyed by the hash cells of a representative row; per slot:
    u32* rep;          // representative row (ZK_EMPTY_SLOT = free)           [pre-filled 0xFF]
    u32* first;        // smallest first-row of the hash's runs               [pre-filled 0xFF]
    u32* last;         // largest last-row of the hash's runs                 [pre-filled 0]
    u32* runs;         // number of runs                                      [pre-filled 0]
    u32* bad;          // a row broke the header-then-bytes pattern           [pre-filled 0]
    u32 big_mask;
    u32* list;         // [DIRB_MAX_ENTRIES]: slot of entry k
    ZkCodeEntry* entries;  // [DIRB_MAX_ENTRIES]
    EvmDyn* dyn;       // dir_entries (counter), codes_n / codes_mask (result)
    u32* small_slots;  // the directory's own slot table (ZkCodeDir::slots), DIRB_SMALL_SLOTS entries [pre-filled 0xFF]
    uint16_t* packed;  // ZkCodeDir::packed
};

__device__ __forceinline__ bool dirb_small(const u64* rows, u32 r, int c, u64& v) {
    const u64* p = rows + ((u64)r * 6 + c) * 4;
    v = p[0];
    return (p[1] | p[2] | p[3]) == 0ull;
}
__device__ __forceinline__ bool dirb_same_hash(const u64* rows, u32 a, u32 b) {
    const u64* p = rows + (u64)a * 24;
    const u64* q = rows + (u64)b * 24;
    bool eq = true;
#pragma unroll
    for (int k = 0; k < 8; k++) eq = eq && p[k] == q[k];
    return eq;
}
__device__ __forceinline__ u32 dirb_hash_of_row(const u64* rows, u32 r) {
    return (u32)zk_code_hash_key(fr_load(rows + (u64)r * 24), fr_load(rows + (u64)r * 24 + 4));
}
// slot of row r's hash: found, or claimed (the claimer numbers the entry)
__device__ __forceinline__ u32 dirb_find_or_insert(const DirBuild& d, u32 r) {
    u32 s = dirb_hash_of_row(d.rows, r) & d.big_mask;
    while (true) {
        u32 cur = d.rep[s];
        if (cur == ZK_EMPTY_SLOT) {
            cur = atomicCAS(&d.rep[s], ZK_EMPTY_SLOT, r);
            if (cur == ZK_EMPTY_SLOT) {
                const u32 k = atomicAdd(&d.dyn->dir_entries, 1u);
                if (k < DIRB_MAX_ENTRIES) d.list[k] = s;
                return s;
            }
        }
        if (dirb_same_hash(d.rows, cur, r)) return s;
        s = (s + 1) & d.big_mask;
    }
}
__device__ __forceinline__ u32 dirb_mask_for(u32 n_entries) {
    u32 cap = 16;
    while (cap < 2 * n_entries + 2) cap <<= 1;
    return cap - 1;
}
__device__ __forceinline__ void dirb_events_row(const DirBuild& d, u32 r) {
    if (r >= d.n) return;
    u64 tag, index, is_code, value;
    const bool small = dirb_small(d.rows, r, 2, tag) & dirb_small(d.rows, r, 3, index);
    uint16_t p = 0;
    if (dirb_small(d.rows, r, 4, is_code) && dirb_small(d.rows, r, 5, value) && is_code < 2 && value < 256)
        p = (uint16_t)(0x8000u | (u32)(is_code << 8) | (u32)value);
    d.packed[r] = p;
    const bool leader = r == 0 || !dirb_same_hash(d.rows, r - 1, r);
    bool ok;
    if (leader) {
        ok = small && tag == 1 && index == 0;
    } else {
        u64 ptag, pindex;
        const bool psmall = dirb_small(d.rows, r - 1, 2, ptag) & dirb_small(d.rows, r - 1, 3, pindex);
        ok = small && psmall && tag == 2 && ((ptag == 1 && index == 0) || (ptag == 2 && index == pindex + 1));
    }
    if (leader) {
        const u32 s = dirb_find_or_insert(d, r);
        atomicAdd(&d.runs[s], 1u);
        atomicMin(&d.first[s], r);
        if (!ok) d.bad[s] = 1;
        if (r != 0) atomicMax(&d.last[dirb_find_or_insert(d, r - 1)], r - 1);  // the previous run ends in front of this row
    } else if (!ok) {
        d.bad[dirb_find_or_insert(d, r)] = 1;
    }
    if (r == d.n - 1) atomicMax(&d.last[dirb_find_or_insert(d, r)], r);
}
// One lane per possible entry; lane 0 publishes the directory (entry count, slot mask) in the session's EvmDyn block.
__device__ __forceinline__ void dirb_finalize_entry(const DirBuild& d, u32 k) {
    const u32 n_entries = d.dyn->dir_entries;
    if (n_entries > DIRB_MAX_ENTRIES) return;  // too many codes: codes_n stays 0 (generic index only)
    const u32 mask = dirb_mask_for(n_entries);
    if (k == 0) {
        d.dyn->codes_n = n_entries;
        d.dyn->codes_mask = mask;
    }
    if (k >= n_entries) return;
    const u32 s = d.list[k];
    const u32 rep = d.rep[s], first = d.first[s], last = d.last[s];
    ZkCodeEntry e;
#pragma unroll
    for (int q = 0; q < 8; q++) e.hash[q] = d.rows[(u64)rep * 24 + q];
    e.header_row = e.byte_base = e.n_bytes = e.regular = 0;
    e.header_value = 0;
    e.header_ok = e.pad = 0;
    if (d.runs[s] == 1u && !d.bad[s] && last >= first) {
        u64 hv, hc;
        e.regular = 1;
        e.header_row = first;
        e.n_bytes = last - first;
        e.byte_base = e.n_bytes ? first + 1u : 0u;
        if (dirb_small(d.rows, first, 5, hv) & dirb_small(d.rows, first, 4, hc) && hc == 0) { e.header_value = hv; e.header_ok = 1; }
    }
    d.entries[k] = e;
    u32 t = (u32)zk_code_hash_key(fr_load(e.hash), fr_load(e.hash + 4)) & mask;
    while (atomicCAS(&d.small_slots[t], ZK_EMPTY_SLOT, k) != ZK_EMPTY_SLOT) t = (t + 1) & mask;
}
