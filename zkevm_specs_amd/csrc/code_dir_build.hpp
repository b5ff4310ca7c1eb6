// Device-side construction of the bytecode directory (ZkCodeDir, common.hpp) from the row-major bytecode table: the
// same directory host_index.hpp's build_code_dir builds on the CPU (which the CPU logic harness keeps using), as a
// handful of one-lane-per-row kernels, so that opening an EVM session over device-resident tables moves no table data
// to the host and the open-time work is all device work that can be timed (bench.py `fresh_witness`).
//
// A code (= the rows of one bytecode hash) is "regular" when it has exactly one Header row (tag 1, index 0) and its Byte
// rows (tag 2) are stored contiguously with indices 0..k-1 in increasing order — what Bytecode.table_assignments produces
// (evm_circuit/typing.py:390-405).  Regular codes are addressed directly (header_row / byte_base + index); anything else
// goes through the generic open-addressing index.  The order of the directory entries is whatever the atomics produce;
// lookups only ever reach an entry through the slot table, by hash.
//
// Round 3: nothing is read back by the host.  The number of codes is not known when the buffers are taken, so the
// directory has a fixed capacity (DIRB_MAX_ENTRIES codes; a table with more distinct hashes gets no directory and every
// bytecode lookup goes through the generic index — correct, only slower); the entry count and the slot mask go to the
// session's EvmDyn block, which the evaluation kernels read at entry.  Only the first row of every run of equal hashes
// inserts into the row hash table (43,913 rows of 16 contracts used to be 43,913 CAS / atomicMin operations on 16 words:
// 65 us; now 16).
#pragma once
#include "common.hpp"

#define DIRB_MAX_ENTRIES 4096u
#define DIRB_SMALL_SLOTS 16384u  // capacity of the directory's own slot table (>= 2 * DIRB_MAX_ENTRIES + 2, power of two)

struct DirBuild {
    const u64* rows;   // [n][6][4]: hash lo, hi, field_tag, index, is_code, value
    u32 n;
    u32* big_slots;    // open addressing over the ROWS keyed by the hash cells: smallest row index of each group
    u32* slot_entry;   // directory entry of the group that owns a big slot
    u32 big_mask;
    ZkCodeEntry* entries;  // [DIRB_MAX_ENTRIES]
    EvmDyn* dyn;       // dir_entries (counter), codes_n / codes_mask (result)
    u32* e_headers;    // per entry: number of Header rows
    u32* e_first;      // per entry: smallest Byte row
    u32* e_last;       // per entry: largest Byte row
    u32* e_bad;        // per entry: a row that rules out "regular"
    u32* small_slots;  // the directory's own slot table (ZkCodeDir::slots), DIRB_SMALL_SLOTS entries, pre-filled with ZK_EMPTY_SLOT
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
// slot of row r's group (the group must have been inserted)
__device__ __forceinline__ u32 dirb_find(const DirBuild& d, u32 r) {
    u32 s = dirb_hash_of_row(d.rows, r) & d.big_mask;
    while (true) {
        const u32 rep = d.big_slots[s];
        if (rep != ZK_EMPTY_SLOT && dirb_same_hash(d.rows, rep, r)) return s;
        s = (s + 1) & d.big_mask;
    }
}
__device__ __forceinline__ u32 dirb_mask_for(u32 n_entries) {
    u32 cap = 16;
    while (cap < 2 * n_entries + 2) cap <<= 1;
    return cap - 1;
}
// Run leaders (rows whose hash differs from the previous row's) insert their group; the slot keeps the smallest row
// index of the group whatever the launch order and however many runs a hash has.
__global__ void dirb_insert_kernel(DirBuild d) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= d.n) return;
    if (r != 0 && dirb_same_hash(d.rows, r - 1, r)) return;
    u32 s = dirb_hash_of_row(d.rows, r) & d.big_mask;
    while (true) {
        u32 cur = d.big_slots[s];
        if (cur == ZK_EMPTY_SLOT) {
            cur = atomicCAS(&d.big_slots[s], ZK_EMPTY_SLOT, r);
            if (cur == ZK_EMPTY_SLOT) return;
        }
        if (dirb_same_hash(d.rows, cur, r)) {  // same group: the slot keeps the smallest row index
            atomicMin(&d.big_slots[s], r);
            return;
        }
        s = (s + 1) & d.big_mask;
    }
}
// The group's smallest row numbers its entry and initialises it (entries past the capacity are counted, not stored).
__global__ void dirb_leaders_kernel(DirBuild d) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= d.n) return;
    if (r != 0 && dirb_same_hash(d.rows, r - 1, r)) return;  // not even a run leader
    const u32 s = dirb_find(d, r);
    if (d.big_slots[s] != r) return;
    const u32 k = atomicAdd(&d.dyn->dir_entries, 1u);
    d.slot_entry[s] = k;
    if (k >= DIRB_MAX_ENTRIES) return;
    ZkCodeEntry e;
#pragma unroll
    for (int q = 0; q < 8; q++) e.hash[q] = d.rows[(u64)r * 24 + q];
    e.header_row = e.byte_base = e.n_bytes = e.regular = 0;
    e.header_value = 0;
    e.header_ok = e.pad = 0;
    d.entries[k] = e;
    d.e_headers[k] = 0;
    d.e_first[k] = 0xffffffffu;
    d.e_last[k] = 0;
    d.e_bad[k] = 0;
}
// Rows of one code sit next to each other (the flattener sorts by hash), so a wavefront usually works on ONE directory
// entry: it then reduces its 64 rows with ballots and issues at most five atomics instead of 64 x 3 on the same words
// (43,913 rows of 16 contracts: 525 us with per-lane atomics).
__global__ void dirb_accumulate_kernel(DirBuild d) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = r < d.n;
    const bool overflow = d.dyn->dir_entries > DIRB_MAX_ENTRIES;  // uniform: no directory will be published
    u32 k = 0xffffffffu;
    u32 cat = 3;  // 0 rules out "regular", 1 Header row, 2 Byte row, 3 lane past the table
    if (in) {
        u64 tag, index, is_code, value;
        const bool small = dirb_small(d.rows, r, 2, tag) & dirb_small(d.rows, r, 3, index);
        cat = (!small || (tag != 1 && tag != 2) || (tag == 1 && index != 0)) ? 0u : (u32)tag;
        uint16_t p = 0;
        if (dirb_small(d.rows, r, 4, is_code) && dirb_small(d.rows, r, 5, value) && is_code < 2 && value < 256)
            p = (uint16_t)(0x8000u | (u32)(is_code << 8) | (u32)value);
        d.packed[r] = p;
        if (!overflow) k = d.slot_entry[dirb_find(d, r)];
    }
    if (overflow) return;
    const unsigned long long active = __ballot(in);
    if (active == 0ull) return;
    const u32 lane = threadIdx.x & 63u;
    const u32 k0 = __shfl(k, __ffsll((long long)active) - 1);
    if (__ballot(in && k != k0) == 0ull) {  // one entry for the whole wavefront
        const unsigned long long bad = __ballot(cat == 0), hdr = __ballot(cat == 1), byt = __ballot(cat == 2);
        const u32 wave_row0 = r - lane;
        if (lane == (u32)__ffsll((long long)active) - 1u) {
            if (bad) d.e_bad[k0] = 1;
            if (hdr) {
                atomicAdd(&d.e_headers[k0], (u32)__popcll(hdr));
                d.entries[k0].header_row = wave_row0 + (u32)__ffsll((long long)hdr) - 1u;
            }
            if (byt) {
                atomicMin(&d.e_first[k0], wave_row0 + (u32)__ffsll((long long)byt) - 1u);
                atomicMax(&d.e_last[k0], wave_row0 + 63u - (u32)__clzll((long long)byt));
                atomicAdd(&d.entries[k0].n_bytes, (u32)__popcll(byt));
            }
        }
        return;
    }
    if (!in) return;
    if (cat == 0) {
        d.e_bad[k] = 1;
    } else if (cat == 1) {
        atomicAdd(&d.e_headers[k], 1u);
        d.entries[k].header_row = r;  // exactly one writer when the code turns out regular
    } else {
        atomicMin(&d.e_first[k], r);
        atomicMax(&d.e_last[k], r);
        atomicAdd(&d.entries[k].n_bytes, 1u);
    }
}
// Byte rows must sit at first_byte + index
__global__ void dirb_check_kernel(DirBuild d) {
    const u32 r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= d.n || d.dyn->dir_entries > DIRB_MAX_ENTRIES) return;
    u64 tag, index;
    if (!(dirb_small(d.rows, r, 2, tag) & dirb_small(d.rows, r, 3, index)) || tag != 2) return;
    const u32 k = d.slot_entry[dirb_find(d, r)];
    if (index != (u64)(r - d.e_first[k])) d.e_bad[k] = 1;
}
// One lane per possible entry; publishes the directory (entry count, slot mask) in the session's EvmDyn block.
__global__ void dirb_finalize_kernel(DirBuild d) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    const u32 n_entries = d.dyn->dir_entries;
    if (n_entries > DIRB_MAX_ENTRIES) return;  // too many codes: codes_n stays 0 (generic index only)
    const u32 mask = dirb_mask_for(n_entries);
    if (k == 0) {
        d.dyn->codes_n = n_entries;
        d.dyn->codes_mask = mask;
    }
    if (k >= n_entries) return;
    ZkCodeEntry& e = d.entries[k];
    const u32 nb = e.n_bytes;
    const bool regular = !d.e_bad[k] && d.e_headers[k] == 1u && (nb == 0 || d.e_last[k] - d.e_first[k] + 1u == nb);
    if (regular) {
        u64 hv, hc;
        e.byte_base = nb ? d.e_first[k] : 0u;
        if (dirb_small(d.rows, e.header_row, 5, hv) & dirb_small(d.rows, e.header_row, 4, hc) && hc == 0) { e.header_value = hv; e.header_ok = 1; }
        e.regular = 1;
    } else {
        e.header_row = e.byte_base = e.n_bytes = 0;
    }
    u32 s = (u32)zk_code_hash_key(fr_load(e.hash), fr_load(e.hash + 4)) & mask;
    while (atomicCAS(&d.small_slots[s], ZK_EMPTY_SLOT, k) != ZK_EMPTY_SLOT) s = (s + 1) & mask;
}
