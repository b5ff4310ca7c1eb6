// Shared definitions for the constraint kernels: status codes, the pass/fail tally, the
// open-addressing row index used by every lookup table, and cell accessors.
#pragma once
#include "fr.hpp"

#if defined(ZK_HOSTSIM)
#include <string.h>
#endif

// ---------------------------------------------------------------------------------------
// Status code of one evaluated row / step:  0 = all constraints satisfied, otherwise
// (kind << 24) | site.  `kind` is the Python exception class the reference raises at the
// first failing check of that row; `site` identifies the check in
// source order (kernel-specific numbering, documented next to each kernel).
// ---------------------------------------------------------------------------------------
enum ZkKind : u32 {
    ZK_OK = 0,
    ZK_ASSERT = 1,            // AssertionError (constrain_*, plain assert)
    ZK_CONSTRAINT = 2,        // ConstraintUnsatFailure *raised* (range_check, word_to_fq)
    ZK_LOOKUP_UNSAT = 3,      // LookupUnsatFailure (table.py:688,880)
    ZK_LOOKUP_AMBIGUOUS = 4,  // LookupAmbiguousFailure (table.py:882)
    ZK_WRONG_QUERY_KEY = 5,   // WrongQueryKey (table.py:387) - unreachable from fixed call sites
    ZK_NOT_IMPLEMENTED = 6,   // NotImplementedError (main.py:63)
    ZK_TYPE_ERROR = 7,        // TypeError
    ZK_OVERFLOW_ERROR = 8,    // OverflowError (int.to_bytes on an over-wide / negative value)
    ZK_VALUE_ERROR = 9,       // ValueError (enum ctor, bytes() out of range)
    ZK_ZERO_DIVISION = 10,    // ZeroDivisionError
    ZK_INDEX_ERROR = 12,      // IndexError (tx rows shorter than MAX_TXS * 12)
    ZK_ATTRIBUTE_ERROR = 13,  // AttributeError (error_oog_precompile.py:33 calls .expr() on a plain int gas cost)
    ZK_NAME_ERROR = 11,       // NameError / UnboundLocalError (block_ctx.py:24 with a foreign opcode)
    ZK_UNSUPPORTED = 15,      // engine limitation: state/gadget not implemented on device
};
#define ZK_CODE(kind, site) ((((u32)(kind)) << 24) | ((u32)(site) & 0xffffffu))

struct ZkTally {
    unsigned long long fail_count;       // rows with status != 0
    unsigned long long first_fail;       // min over failing rows of (row << 32 | code); ~0 = none
};

#define ZK_EMPTY_SLOT 0xffffffffu

// 64-bit mixer (splitmix64 finaliser) used to spread keys over the open-addressing index.
ZK_HD u64 zk_mix64(u64 x) {
    x ^= x >> 30;
    x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27;
    x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}
ZK_HD u64 zk_hash_cell(u64 h, const Fr& c) {
    u64 a = ((u64)c.v[0] | ((u64)c.v[1] << 32)) ^ (((u64)c.v[2] | ((u64)c.v[3] << 32)) * 0x9e3779b97f4a7c15ull);
    u64 b = ((u64)c.v[4] | ((u64)c.v[5] << 32)) ^ (((u64)c.v[6] | ((u64)c.v[7] << 32)) * 0xc2b2ae3d27d4eb4full);
    return zk_mix64(h ^ a ^ (b << 1) ^ (b >> 63)) + 0x632be59bd9b4e019ull;
}

// Row-major lookup table resident in HBM: n rows of `ncells` canonical cells (32 B each),
// one u32 of per-row type bits (is_word flags of WordOrValue columns), and an
// open-addressing index (slot -> row id) keyed on a fixed subset of the cells.
struct ZkTable {
    const u64* cells;   // [n][ncells][4]
    const u32* flags;   // [n] or nullptr
    const u32* slots;   // [mask + 1], ZK_EMPTY_SLOT = empty
    u32 n;
    u32 ncells;
    u32 mask;
};
ZK_HD Fr zk_table_cell(const ZkTable& t, u32 row, u32 c) {
    return fr_load(t.cells + ((u64)row * t.ncells + c) * 4);
}

// Does row r equal the query on the cells `mask` selects?  The compiler turns a cell-by-cell compare into one load + wait per
// 16 bytes when registers are short (every probed row then costs two dependent memory latencies PER CELL); here up to eight
// cells = sixteen 16-byte loads share one round trip (then four, two, one: a 14-cell row is three trips, a mismatch in the
// first eight cells — where every table's hashed key cells sit — is one), and a group the mask does not touch is not read.
#ifndef ZK_HOSTSIM
typedef u32 zk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void zk_load2x16(const u64* p, zk_u32x4 (&x)[2]) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)" : "=&v"(x[0]), "=&v"(x[1]) : "v"(p) : "memory");
}
__device__ __forceinline__ void zk_load4x16(const u64* p, zk_u32x4 (&x)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
        "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48\n\ts_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void zk_load8x16(const u64* p, zk_u32x4 (&x)[8]) {  // 128 contiguous bytes, one wait
    asm volatile(
        "global_load_dwordx4 %0, %8, off\n\tglobal_load_dwordx4 %1, %8, off offset:16\n\t"
        "global_load_dwordx4 %2, %8, off offset:32\n\tglobal_load_dwordx4 %3, %8, off offset:48\n\t"
        "global_load_dwordx4 %4, %8, off offset:64\n\tglobal_load_dwordx4 %5, %8, off offset:80\n\t"
        "global_load_dwordx4 %6, %8, off offset:96\n\tglobal_load_dwordx4 %7, %8, off offset:112\n\ts_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void zk_load16x16(const u64* p, zk_u32x4 (&x)[16]) {  // 256 contiguous bytes, one wait
    asm volatile(
        "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:16\n\t"
        "global_load_dwordx4 %2, %16, off offset:32\n\tglobal_load_dwordx4 %3, %16, off offset:48\n\t"
        "global_load_dwordx4 %4, %16, off offset:64\n\tglobal_load_dwordx4 %5, %16, off offset:80\n\t"
        "global_load_dwordx4 %6, %16, off offset:96\n\tglobal_load_dwordx4 %7, %16, off offset:112\n\t"
        "global_load_dwordx4 %8, %16, off offset:128\n\tglobal_load_dwordx4 %9, %16, off offset:144\n\t"
        "global_load_dwordx4 %10, %16, off offset:160\n\tglobal_load_dwordx4 %11, %16, off offset:176\n\t"
        "global_load_dwordx4 %12, %16, off offset:192\n\tglobal_load_dwordx4 %13, %16, off offset:208\n\t"
        "global_load_dwordx4 %14, %16, off offset:224\n\tglobal_load_dwordx4 %15, %16, off offset:240\n\ts_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
          "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void zk_load10x16(const u64* p, zk_u32x4 (&x)[10]) {  // 160 contiguous bytes, one wait
    asm volatile(
        "global_load_dwordx4 %0, %10, off\n\t"
        "global_load_dwordx4 %1, %10, off offset:16\n\t"
        "global_load_dwordx4 %2, %10, off offset:32\n\t"
        "global_load_dwordx4 %3, %10, off offset:48\n\t"
        "global_load_dwordx4 %4, %10, off offset:64\n\t"
        "global_load_dwordx4 %5, %10, off offset:80\n\t"
        "global_load_dwordx4 %6, %10, off offset:96\n\t"
        "global_load_dwordx4 %7, %10, off offset:112\n\t"
        "global_load_dwordx4 %8, %10, off offset:128\n\t"
        "global_load_dwordx4 %9, %10, off offset:144\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void zk_load22x16(const u64* p, zk_u32x4 (&x)[22]) {  // 352 contiguous bytes, one wait
    asm volatile(
        "global_load_dwordx4 %0, %22, off\n\t"
        "global_load_dwordx4 %1, %22, off offset:16\n\t"
        "global_load_dwordx4 %2, %22, off offset:32\n\t"
        "global_load_dwordx4 %3, %22, off offset:48\n\t"
        "global_load_dwordx4 %4, %22, off offset:64\n\t"
        "global_load_dwordx4 %5, %22, off offset:80\n\t"
        "global_load_dwordx4 %6, %22, off offset:96\n\t"
        "global_load_dwordx4 %7, %22, off offset:112\n\t"
        "global_load_dwordx4 %8, %22, off offset:128\n\t"
        "global_load_dwordx4 %9, %22, off offset:144\n\t"
        "global_load_dwordx4 %10, %22, off offset:160\n\t"
        "global_load_dwordx4 %11, %22, off offset:176\n\t"
        "global_load_dwordx4 %12, %22, off offset:192\n\t"
        "global_load_dwordx4 %13, %22, off offset:208\n\t"
        "global_load_dwordx4 %14, %22, off offset:224\n\t"
        "global_load_dwordx4 %15, %22, off offset:240\n\t"
        "global_load_dwordx4 %16, %22, off offset:256\n\t"
        "global_load_dwordx4 %17, %22, off offset:272\n\t"
        "global_load_dwordx4 %18, %22, off offset:288\n\t"
        "global_load_dwordx4 %19, %22, off offset:304\n\t"
        "global_load_dwordx4 %20, %22, off offset:320\n\t"
        "global_load_dwordx4 %21, %22, off offset:336\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17]), "=&v"(x[18]), "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21])
        : "v"(p)
        : "memory");
}
__device__ __forceinline__ void zk_load28x16(const u64* p, zk_u32x4 (&x)[28]) {  // 448 contiguous bytes, one wait
    asm volatile(
        "global_load_dwordx4 %0, %28, off\n\t"
        "global_load_dwordx4 %1, %28, off offset:16\n\t"
        "global_load_dwordx4 %2, %28, off offset:32\n\t"
        "global_load_dwordx4 %3, %28, off offset:48\n\t"
        "global_load_dwordx4 %4, %28, off offset:64\n\t"
        "global_load_dwordx4 %5, %28, off offset:80\n\t"
        "global_load_dwordx4 %6, %28, off offset:96\n\t"
        "global_load_dwordx4 %7, %28, off offset:112\n\t"
        "global_load_dwordx4 %8, %28, off offset:128\n\t"
        "global_load_dwordx4 %9, %28, off offset:144\n\t"
        "global_load_dwordx4 %10, %28, off offset:160\n\t"
        "global_load_dwordx4 %11, %28, off offset:176\n\t"
        "global_load_dwordx4 %12, %28, off offset:192\n\t"
        "global_load_dwordx4 %13, %28, off offset:208\n\t"
        "global_load_dwordx4 %14, %28, off offset:224\n\t"
        "global_load_dwordx4 %15, %28, off offset:240\n\t"
        "global_load_dwordx4 %16, %28, off offset:256\n\t"
        "global_load_dwordx4 %17, %28, off offset:272\n\t"
        "global_load_dwordx4 %18, %28, off offset:288\n\t"
        "global_load_dwordx4 %19, %28, off offset:304\n\t"
        "global_load_dwordx4 %20, %28, off offset:320\n\t"
        "global_load_dwordx4 %21, %28, off offset:336\n\t"
        "global_load_dwordx4 %22, %28, off offset:352\n\t"
        "global_load_dwordx4 %23, %28, off offset:368\n\t"
        "global_load_dwordx4 %24, %28, off offset:384\n\t"
        "global_load_dwordx4 %25, %28, off offset:400\n\t"
        "global_load_dwordx4 %26, %28, off offset:416\n\t"
        "global_load_dwordx4 %27, %28, off offset:432\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]), "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15]), "=&v"(x[16]), "=&v"(x[17]), "=&v"(x[18]), "=&v"(x[19]), "=&v"(x[20]), "=&v"(x[21]), "=&v"(x[22]), "=&v"(x[23]), "=&v"(x[24]), "=&v"(x[25]), "=&v"(x[26]), "=&v"(x[27])
        : "v"(p)
        : "memory");
}
// whole row of an NCELLS-cell table in one round trip (the loaders above exist for the cell counts that need it)
template <int NCHUNKS>
__device__ __forceinline__ void zk_load_row(const u64* p, zk_u32x4 (&x)[NCHUNKS]) {
    if constexpr (NCHUNKS == 10) zk_load10x16(p, x);
    else if constexpr (NCHUNKS == 22) zk_load22x16(p, x);
    else if constexpr (NCHUNKS == 28) zk_load28x16(p, x);
    else {
#pragma unroll
        for (int k = 0; k < NCHUNKS; k++) x[k] = reinterpret_cast<const zk_u32x4*>(p)[k];
    }
}
template <int N>
__device__ __forceinline__ u32 zk_cells_diff(const zk_u32x4 (&x)[N], u32 c, const Fr* q, u32 mask) {  // N / 2 cells from cell c on
    u32 diff = 0;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const u32 cell = c + (u32)(k >> 1);
        if ((mask >> cell) & 1u) {
            const u32* qq = q[cell].v + (k & 1) * 4;
            diff |= (x[k].x ^ qq[0]) | (x[k].y ^ qq[1]) | (x[k].z ^ qq[2]) | (x[k].w ^ qq[3]);
        }
    }
    return diff;
}
#endif
ZK_HD bool zk_row_matches(const ZkTable& t, u32 r, const Fr* q, u32 mask) {
    u32 diff = 0, c = 0;
#ifndef ZK_HOSTSIM
    const u64* p = t.cells + (u64)r * t.ncells * 4;
    for (; c + 8 <= t.ncells && diff == 0; c += 8)
        if ((mask >> c) & 0xffu) {
            zk_u32x4 x[16];
            zk_load16x16(p + (u64)c * 4, x);
            diff = zk_cells_diff<16>(x, c, q, mask);
        }
    if (c + 4 <= t.ncells && diff == 0) {
        if ((mask >> c) & 0xfu) {
            zk_u32x4 x[8];
            zk_load8x16(p + (u64)c * 4, x);
            diff = zk_cells_diff<8>(x, c, q, mask);
        }
        c += 4;
    }
    if (c + 2 <= t.ncells && diff == 0) {
        if ((mask >> c) & 0x3u) {
            zk_u32x4 x[4];
            zk_load4x16(p + (u64)c * 4, x);
            diff = zk_cells_diff<4>(x, c, q, mask);
        }
        c += 2;
    }
    if (c < t.ncells && diff == 0) {
        if ((mask >> c) & 1u) {
            zk_u32x4 x[2];
            zk_load2x16(p + (u64)c * 4, x);
            diff = zk_cells_diff<2>(x, c, q, mask);
        }
        c++;
    }
    return diff == 0;
#else
    for (; c < t.ncells && diff == 0; c++)
        if ((mask >> c) & 1u) {
            const Fr x = zk_table_cell(t, r, c);
            for (int k = 0; k < 8; k++) diff |= x.v[k] ^ q[c].v[k];
        }
    return diff == 0;
#endif
}

// Direct-index metadata built when a session is opened (the analogue of the reference building
// its `Tables` sets, evm_circuit/table.py:592-625, before verify_steps runs):
//  * RW table: when the rows are sorted with consecutive rw_counters (rw[i].rw_counter == base + i,
//    what RWDictionary produces, evm_circuit/typing.py:813-845) a lookup by rw_counter is the row
//    `rw_counter - base`; otherwise the generic open-addressing index is used.
struct ZkRwMeta {
    u32 dense;
    u32 pad;
    u64 base;
};
//  * Bytecode table: one directory entry per distinct code hash whose rows are "regular" (one
//    Header row + Byte rows with indices 0..n-1, contiguous, no duplicates): a lookup by
//    (hash, tag, index) is then header_row / byte_base + index.
struct ZkCodeEntry {
    u64 hash[8];  // lo cell (4 x u64) then hi cell
    u32 header_row;
    u32 byte_base;
    u32 n_bytes;
    u32 regular;
    u64 header_value;  // the Header row's value cell (code length) when header_ok
    u32 header_ok;     // 1: that cell fits 64 bits and the row's is_code cell is 0
    u32 pad;
};
struct ZkCodeDir {
    const ZkCodeEntry* entries;
    const u32* slots;  // open addressing over entries, keyed on the hash cells
    u32 mask;
    u32 n;
    // one u16 per bytecode-table row: bit 15 = "fits" (value < 256, is_code < 2), bit 8 = is_code,
    // bits 0-7 = value; 0 = compare the row's cells instead.  A lookup of a regular code reads 2 B
    // instead of two 32-byte cells; nullptr when no directory was built.
    const uint16_t* packed;
};
// keccak table (KeccakTableRow, table.py:511-515: state_tag, input_rlc, input_len, output lo/hi), keyed on (rlc, len)
enum { KECCAK_NCELLS = 5 };
ZK_HD u64 keccak_key_hash_cells(const Fr& rlc, const Fr& len) { return zk_hash_cell(zk_hash_cell(0x6b656363u, rlc), len); }
ZK_HD u64 keccak_key_hash(const ZkTable& t, u32 r) { return keccak_key_hash_cells(zk_table_cell(t, r, 1), zk_table_cell(t, r, 2)); }
ZK_HD u64 zk_code_hash_key(const Fr& lo, const Fr& hi) { return zk_hash_cell(zk_hash_cell(0xc0de5u, lo), hi); }

// Column-major witness: cell c of row i at cells[(c * n + i) * 4].
ZK_HD bool rows_identical(const ZkTable& t, u32 r0, u32 r1) {
    bool same = true;
    for (u32 c = 0; c < t.ncells; c++) same = same && fr_eq(zk_table_cell(t, r0, c), zk_table_cell(t, r1, c));
    return same;
}

// Inline "exactly one distinct matching row" probe over an open-addressing index, query cells and mask compile-time: `kind` = 0 / ZK_LOOKUP_UNSAT / ZK_LOOKUP_AMBIGUOUS, returns the row found (0 if
// none).  Used by the EVM warm gadgets (table_lookup_inline), the Copy circuit's lookups and the keccak-table membership tests.
template <int NCELLS, u32 MASK>
ZK_HD u32 table_probe_inline(const ZkTable& t, u64 h, const Fr (&q)[NCELLS], u32& kind, Fr* out0 = nullptr, int out0_cell = 0, Fr* out1 = nullptr,
                             int out1_cell = 0) {
    u32 found = ZK_EMPTY_SLOT;
    bool ambiguous = false;
    if (out0) *out0 = fr_zero();
    if (out1) *out1 = fr_zero();
    if (t.n != 0) {
        u32 slot = (u32)h & t.mask;
        // two dependent round trips per probe: the slot and its successor together (an empty successor ends the probe sequence
        // without a trip of its own), then the candidate's whole row in one batch — compared in registers, and the cells the
        // caller wants back (out0 / out1) taken from the same batch
        u32 r = t.slots[slot], r_next = t.slots[(slot + 1) & t.mask];
        for (u32 probes = 0; probes <= t.mask; probes++) {
            if (r == ZK_EMPTY_SLOT) break;
            const u64* p = t.cells + (u64)r * NCELLS * 4;
            u32 diff = 0;
#ifndef ZK_HOSTSIM
            zk_u32x4 x[2 * NCELLS];
            zk_load_row<2 * NCELLS>(p, x);
#pragma unroll
            for (int k = 0; k < 2 * NCELLS; k++)
                if ((MASK >> (k >> 1)) & 1u) {
                    const u32* qq = q[k >> 1].v + (k & 1) * 4;
                    diff |= (x[k].x ^ qq[0]) | (x[k].y ^ qq[1]) | (x[k].z ^ qq[2]) | (x[k].w ^ qq[3]);
                }
#else
            for (int c = 0; c < NCELLS; c++)
                if ((MASK >> c) & 1u) {
                    const Fr cell = fr_load(p + 4 * c);
                    for (int k = 0; k < 8; k++) diff |= cell.v[k] ^ q[c].v[k];
                }
#endif
            if (diff == 0u) {
                if (found == ZK_EMPTY_SLOT) {
                    found = r;
#ifndef ZK_HOSTSIM
#pragma unroll
                    for (int c = 0; c < NCELLS; c++) {  // (out*_cell are compile-time constants at every call site)
                        if (out0 && c == out0_cell) { for (int k = 0; k < 4; k++) { out0->v[k] = x[2 * c][k]; out0->v[4 + k] = x[2 * c + 1][k]; } }
                        if (out1 && c == out1_cell) { for (int k = 0; k < 4; k++) { out1->v[k] = x[2 * c][k]; out1->v[4 + k] = x[2 * c + 1][k]; } }
                    }
#else
                    if (out0) *out0 = fr_load(p + 4 * out0_cell);
                    if (out1) *out1 = fr_load(p + 4 * out1_cell);
#endif
                } else if (!rows_identical(t, found, r)) {
                    ambiguous = true;
                }
            }
            slot = (slot + 1) & t.mask;
            r = r_next;
            if (r != ZK_EMPTY_SLOT) r_next = t.slots[(slot + 1) & t.mask];
        }
    }
    kind = found == ZK_EMPTY_SLOT ? (u32)ZK_LOOKUP_UNSAT : (ambiguous ? (u32)ZK_LOOKUP_AMBIGUOUS : 0u);
    return found == ZK_EMPTY_SLOT ? 0u : found;
}


struct ZkCols {
    const u64* cells;
    const u32* flags;
    u64 n;
    u32 skip;  // State rows only (ZK_OPT_STATE_COMPACT): 42 = the limb / byte columns 8..49 are absent, columns 50.. follow column 7; else 0
    u32 pad_;
};
ZK_HD Fr zk_col(const ZkCols& w, u32 c, u64 i) { return fr_load(w.cells + ((u64)c * w.n + i) * 4); }

// (pos - sub) / 2^128 in the field, for a value that goes straight into range_check(., 9).  With sub < 2^200 the field
// quotient is below 2^72 exactly when pos - sub is a non-negative integer multiple of 2^128 with a quotient below 2^72
// (r * 2^128 < 2^200 < p pins (pos - sub) mod p = r * 2^128; a negative difference would need sub > p - 2^200): the
// integer shift gives the same value then, and any other numerator only has to fail the range check like the field
// quotient does.  No Montgomery product on the way.
ZK_HD Fr div_2p128_for_range9(const Fr& pos, const Fr& sub) {
    if ((sub.v[7] | (sub.v[6] >> 8)) != 0u) return fr_mulc(fr_sub(pos, sub), frm_inv_2p128());
    Fr n;
    const u32 bw = u256_sub(n, pos, sub);
    const bool exact = !bw && (n.v[0] | n.v[1] | n.v[2] | n.v[3]) == 0u;
    Fr r = fr_zero();
    r.v[0] = n.v[4]; r.v[1] = n.v[5]; r.v[2] = n.v[6]; r.v[3] = n.v[7];
    if (!exact) r.v[7] = 0x20000000u;
    return r;
}
