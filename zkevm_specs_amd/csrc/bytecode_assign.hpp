// Bytecode-circuit witness assignment on the device.
//
// Replaces the reference's `assign_bytecode_circuit(k, bytecodes, keccak_randomness)`
// (src/zkevm_specs/bytecode_circuit.py:104-167): per bytecode the push-data tracking (`push_data_left`,
// `push_data_size`; get_push_size, evm_circuit/opcode.py:427-433) and the running `value_rlc = value_rlc * r + value`
// over the byte rows, then q_first / q_last, `length`, truncation at 2^k rows and the EMPTY_HASH padding rows.
// Input: the unrolled BytecodeTableRows of all bytecodes back to back (row-major, 6 cells: hash lo/hi, tag, index,
// is_code, value) + row offsets and byte lengths per bytecode; output: the 12-cell circuit rows column-major, exactly
// what zk_bytecode_open takes.
//
// The two per-bytecode recurrences are sequential in the reference.  Here both are cut into 64-row chunks:
//   bca_chunk        one lane per chunk: Horner over the chunk from 0 -> (acc, number of byte rows m), AND the chunk's push-data
//                    transition map: push_data_left after the chunk as a function of push_data_left before it (33 states; once
//                    the counter reaches 0 inside the chunk the rest does not depend on how it was entered, so the map is a
//                    backward pass over "state after the chunk when position p is entered with 0")
//   bca_prefix_code  one lane per bytecode: incoming value_rlc (running = running * r^m + acc) and incoming push_data_left (one
//                    table lookup) of every chunk
//   bca_rlc_chunk    one lane per chunk: value_rlc and (push_data_left, push_data_size) of each of its rows from the incoming values
//   bca_write_row    one lane per OUTPUT row (coalesced 32 B/lane stores of the 12 cells), padding rows included
// so a 24,576-byte contract costs ~2 x 64 + 384 dependent steps per lane instead of 24,576 (round 3 walked the push-data
// counter with one lane per bytecode: 2.7 ms for the block's largest contract).
#pragma once
#include "common.hpp"

enum { BCA_IN_NCELLS = 6, BCA_OUT_NCELLS = 12, BCA_CHUNK = 64, BCA_RPOW_ROWS = BCA_CHUNK + 1, BCA_MAP_STRIDE = 36 /* 33 states, padded */ };

static_assert(BCA_CHUNK == 64, "a chunk is one wavefront (k_assign.hip) and bca_dot's ten-limb accumulator holds at most 64 byte terms");

struct BcaChunk {
    u32 code;   // bytecode index
    u32 start;  // first global input row of the chunk
    u32 count;  // rows in the chunk (<= BCA_CHUNK)
    u32 first;  // 1: the chunk starts at the bytecode's first row (idx 0, the Header row: it does not feed value_rlc)
};

struct BcaArgs {
    const u64* in_rows;     // [n_in][6][4]
    const u64* offsets;     // [n_codes + 1]
    const u64* lengths;     // [n_codes]
    u64 n_in, n_codes, n_out;  // n_out = 2^k
    const u64* rpow;        // [65][4]: r^m in Montgomery form, m = 0..64 (row 1 = the randomness itself)
    const BcaChunk* chunks; // [n_chunks]
    const u32* code_chunk0; // [n_codes + 1]: first chunk of every bytecode
    u64 n_chunks;
    uint8_t* track;         // [n_in][2]: push_data_left, push_data_size
    uint8_t* chunk_map;     // [n_chunks][BCA_MAP_STRIDE]: push_data_left after the chunk for every push_data_left (0..32) before it
    u32* chunk_state;       // [n_chunks] push_data_left entering the chunk
    u64* chunk_acc;         // [n_chunks][4] Horner of the chunk from 0
    u32* chunk_m;           // [n_chunks] byte rows in the chunk
    u64* chunk_in;          // [n_chunks][4] incoming value_rlc
    u64* rlc;               // [n_in][4] value_rlc per input row
    u32* row_code;          // [n_in] bytecode of every input row (written by bca_rlc_chunk)
    u32* row_chunk;         // [n_in] chunk of every input row (device kernels: k_assign.hip bca_chunk_wave_kernel)
    u64* rows;              // out [12][n_out][4]
};

ZK_HD Fr bca_in_cell(const BcaArgs& a, u64 row, int c) { return fr_load(a.in_rows + (row * BCA_IN_NCELLS + c) * 4); }
ZK_HD void bca_store(u64* out, const Fr& x) {
    for (int j = 0; j < 4; j++) out[j] = (u64)x.v[2 * j] | ((u64)x.v[2 * j + 1] << 32);
}
// get_push_size(row.value): PUSH1..PUSH32 -> 1..32, anything else (incl. values >= 256) -> 0
ZK_HD u32 bca_push_size(const Fr& v) { return (fr_fits32(v) && v.v[0] >= 0x60u && v.v[0] <= 0x7fu) ? v.v[0] - 0x5fu : 0u; }

ZK_HD void bca_fill_rpow(const Fr& r, u64* out) {  // single lane, once per session
    const Fr rM = fr_to_mont(r);
    Fr acc = frm_one();
    for (int m = 0; m < BCA_RPOW_ROWS; m++) {
        bca_store(out + 4 * m, acc);
        acc = fr_mont(acc, rM);
    }
}
// Horner value of the byte rows [j0, j1) of a chunk (indices among the chunk's BYTE rows; the Header row of a bytecode's first chunk
// is not one), from 0: sum of value_j * r^(j1 - 1 - j).  Values below 2^32 — every well-formed row: a byte — accumulate against the
// Montgomery powers of r as plain integers (ten 32-bit limbs, 8 multiply-adds per row, independent loads), one reduction and one
// product to leave Montgomery form; a wider value (malformed witnesses only) sends the whole range down the reference's own
// recurrence `acc = acc * r + value`.  Round 4 ran that recurrence always: 64 dependent load + Montgomery-product round trips per
// chunk lane, 0.32 ms for a block's 131,072 rows whatever the occupancy.
ZK_HD Fr bca_dot(const BcaArgs& a, const BcaChunk& ch, u32 j1, uint8_t* sizes = nullptr) {  // sizes: optional, push size of every byte row [skip + j]
    const u32 skip = ch.first ? 1u : 0u;  // chunk row t = skip + j
    u32 acc[10];
#pragma unroll
    for (int k = 0; k < 10; k++) acc[k] = 0;
    u32 wide = 0;
#pragma unroll 8
    for (u32 j = 0; j < j1; j++) {
        const Fr v = bca_in_cell(a, (u64)ch.start + skip + j, 5);
        wide |= v.v[1] | v.v[2] | v.v[3] | v.v[4] | v.v[5] | v.v[6] | v.v[7];
        if (sizes) sizes[skip + j] = (uint8_t)bca_push_size(v);  // from the same batch of loads (a loop of its own was 64 serial round trips)
        const Fr pw = fr_load(a.rpow + 4 * (u64)(j1 - 1u - j));
        u64 c = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            c += (u64)acc[k] + (u64)pw.v[k] * v.v[0];  // <= (2^32 - 1)^2 + 2 (2^32 - 1) = 2^64 - 1
            acc[k] = (u32)c;
            c >>= 32;
        }
        c += acc[8];
        acc[8] = (u32)c;
        acc[9] += (u32)(c >> 32);
    }
    if (wide) {  // the exact recurrence over the same rows
        const Fr rM = fr_load(a.rpow + 4);
        Fr x = fr_zero();
        for (u32 j = 0; j < j1; j++) x = fr_add(fr_mont(x, rM), bca_in_cell(a, (u64)ch.start + skip + j, 5));
        return x;
    }
    Fr lo;
#pragma unroll
    for (int k = 0; k < 8; k++) lo.v[k] = acc[k];
    const Fr p = fr_modulus();
#pragma unroll
    for (int it = 0; it < 5; it++) {  // lo < 2^256 < 6 p
        Fr t;
        const u32 bw = u256_sub(t, lo, p);
        lo = bw ? lo : t;
    }
    // + (acc[8] + acc[9] 2^32) * 2^256 (mod p): 2^256 mod p is the Montgomery one; the sum is the Montgomery form of the value
    const Fr xM = fr_add(lo, fr_mul(fr_from_u64((u64)acc[8] | ((u64)acc[9] << 32)), frm_one()));
    return fr_mont(xM, fr_from_u64(1));
}
// The push-data counter (bytecode_circuit.py:117-130): entering a byte row with `left`, the next row is entered with
// (left == 0 ? get_push_size(value) : left - 1); the Header row (the bytecode's first row) leaves it alone.
ZK_HD void bca_chunk(const BcaArgs& a, u64 c) {
    const BcaChunk ch = a.chunks[c];
    const u32 m = ch.count - (ch.first ? 1u : 0u);
    uint8_t size[BCA_CHUNK];
    size[0] = 0;  // the Header row of a bytecode's first chunk; overwritten otherwise
    bca_store(a.chunk_acc + 4 * c, bca_dot(a, ch, m, size));
    a.chunk_m[c] = m;
    // zero_out[p]: the counter after the chunk when position p is entered with 0
    uint8_t zero_out[BCA_CHUNK + 1];
    zero_out[ch.count] = 0;
    for (int p = (int)ch.count - 1; p >= 0; p--) {
        const u32 sz = size[p];
        if (sz == 0) zero_out[p] = zero_out[p + 1];
        else if ((u32)p + sz + 1u <= ch.count) zero_out[p] = zero_out[p + sz + 1];
        else zero_out[p] = (uint8_t)(sz - (ch.count - 1u - (u32)p));  // the push data runs past the chunk
    }
    uint8_t* map = a.chunk_map + c * BCA_MAP_STRIDE;
    for (u32 left = 0; left <= 32u; left++) map[left] = left <= ch.count ? zero_out[left] : (uint8_t)(left - ch.count);
}
ZK_HD void bca_prefix_code(const BcaArgs& a, u64 j) {
    Fr running = fr_zero();
    u32 left = 0;
    for (u32 c = a.code_chunk0[j]; c < a.code_chunk0[j + 1]; c++) {
        bca_store(a.chunk_in + 4 * (u64)c, running);
        a.chunk_state[c] = left;
        running = fr_add(fr_mont(running, fr_load(a.rpow + 4 * a.chunk_m[c])), fr_load(a.chunk_acc + 4 * (u64)c));
        left = a.chunk_map[(u64)c * BCA_MAP_STRIDE + left];
    }
}
// Row t of chunk c: value_rlc after the row = (value entering the chunk) * r^(byte rows up to and including t) + their Horner value;
// push_data_left entering the row and the row's push_data_size by walking the counter from the chunk's entry state.  One output row
// per call, independent of the chunk's other rows (the device runs one lane per row: a chunk is a 64-lane block).
ZK_HD void bca_rlc_row(const BcaArgs& a, u64 c, u32 t) {
    const BcaChunk ch = a.chunks[c];
    if (t >= ch.count) return;
    const u64 g = (u64)ch.start + t;
    const u32 skip = ch.first ? 1u : 0u;
    u32 left = a.chunk_state[c];
    for (u32 q = skip; q < t; q++) {  // rows before this one: only their push sizes matter
        const u32 sz = bca_push_size(bca_in_cell(a, (u64)ch.start + q, 5));
        left = left == 0u ? sz : left - 1u;
    }
    Fr rlc = fr_load(a.chunk_in + 4 * c);
    u32 size = 0;
    if (t >= skip) {
        const u32 k = t + 1u - skip;  // byte rows 0..k-1 of the chunk end at this row
        rlc = fr_add(fr_mont(rlc, fr_load(a.rpow + 4 * (u64)k)), bca_dot(a, ch, k));
        size = bca_push_size(bca_in_cell(a, g, 5));
    }
    bca_store(a.rlc + 4 * g, rlc);
    a.row_code[g] = ch.code;
    a.track[2 * g] = (uint8_t)left;
    a.track[2 * g + 1] = (uint8_t)size;
}
ZK_HD void bca_rlc_chunk(const BcaArgs& a, u64 c) {  // host builds: every row of the chunk
    for (u32 t = 0; t < BCA_CHUNK; t++) bca_rlc_row(a, c, t);
}
// Output row i (bytecode_circuit.Row: q_first, q_last, hash lo, hi, tag, index, value, is_code, push_data_left, value_rlc,
// length, push_data_size)
ZK_HD void bca_write_row(const BcaArgs& a, u64 i) {
    const u64 n = a.n_out;
#define BCA_OUT(c) (a.rows + ((u64)(c) * n + i) * 4)
    bca_store(BCA_OUT(0), fr_from_u64(i == 0 ? 1 : 0));
    bca_store(BCA_OUT(1), fr_from_u64(i == n - 1 ? 1 : 0));
    if (i < a.n_in) {
        bca_store(BCA_OUT(2), bca_in_cell(a, i, 0));
        bca_store(BCA_OUT(3), bca_in_cell(a, i, 1));
        bca_store(BCA_OUT(4), bca_in_cell(a, i, 2));
        bca_store(BCA_OUT(5), bca_in_cell(a, i, 3));
        bca_store(BCA_OUT(6), bca_in_cell(a, i, 5));
        bca_store(BCA_OUT(7), bca_in_cell(a, i, 4));
        bca_store(BCA_OUT(8), fr_from_u64(a.track[2 * i]));
        bca_store(BCA_OUT(9), fr_load(a.rlc + 4 * i));
        bca_store(BCA_OUT(10), fr_from_u64(a.lengths[a.row_code[i]]));
        bca_store(BCA_OUT(11), fr_from_u64(a.track[2 * i + 1]));
    } else {  // padding (:150-165): Header rows of the empty bytecode
        bca_store(BCA_OUT(2), fr_from_u128(0x7bfad8045d85a470ull, 0xe500b653ca82273bull));
        bca_store(BCA_OUT(3), fr_from_u128(0x927e7db2dcc703c0ull, 0xc5d2460186f7233cull));
        bca_store(BCA_OUT(4), fr_from_u64(1));
        for (int c = 5; c < BCA_OUT_NCELLS; c++) bca_store(BCA_OUT(c), fr_zero());
    }
#undef BCA_OUT
}
