// Host-side index plumbing of the Copy-circuit witness assignment (zk_copy_assign*, csrc/copy_assign.hpp), shared by the HIP library
// and the CPU backend: offsets of every event's rows / RW rows / table row / Horner chunks, from the event cells alone (integers
// only, no witness data).  Included after the translation unit has defined ARG_TRY / g_err.
#pragma once
#include <vector>
#include "copy_assign.hpp"

struct CpaPlan {
    std::vector<CpaEvent> ev;
    std::vector<u64> row0;
    std::vector<CpaChunk> chunks;
    u64 n_rows = 0, n_table = 0, n_rw = 0, n_rlc = 0, n_data = 0;
};
// index plumbing over the event cells (integers only): offsets of every event's rows / RW rows / table row / Horner chunks
static int cpa_plan(const u64* ev_cells, const u32* flags, const u64* data_offsets, u64 n, CpaPlan& pl) {
    pl.ev.resize(n);
    pl.row0.assign(n + 1, 0);
    auto small = [&](u64 e, int c, u64& v) {
        const u64* p = ev_cells + (e * CPA_EV_NCELLS + c) * 4;
        v = p[0];
        return (p[1] | p[2] | p[3]) == 0 && v < (1ull << 62);
    };
    for (u64 e = 0; e < n; e++) {
        CpaEvent& x = pl.ev[e];
        u64 st, dt;
        ARG_TRY(small(e, 2, st) && small(e, 5, dt) && small(e, 6, x.src_addr) && small(e, 7, x.src_end) && small(e, 8, x.dst_addr) &&
                small(e, 9, x.length) && small(e, 10, x.log_id) && small(e, 11, x.rwc),
                "zk_copy_assign: an event field is outside the wire's domain (addresses, lengths, counters below 2^62)");
        ARG_TRY(st >= 1 && st <= 5 && dt >= 1 && dt <= 5 && st != CPA_TX_LOG && x.log_id < (1ull << 14) && x.length < (1ull << 31),
                "zk_copy_assign: bad copy data type tag / log id / length");
        x.src_tag = (u32)st; x.dst_tag = (u32)dt; x.flags = flags ? flags[e] : 0u;
        const u64 room = x.src_end > x.src_addr ? x.src_end - x.src_addr : 0;  // cpa_n_real on the host
        const u64 n_real = room < x.length ? room : x.length;
        x.row0 = pl.n_rows; x.rw0 = pl.n_rw; x.data0 = data_offsets[e];
        ARG_TRY(data_offsets[e + 1] >= data_offsets[e] && data_offsets[e + 1] - data_offsets[e] >= n_real, "zk_copy_assign: too few source bytes for an event");
        x.table_idx = x.length ? (u32)pl.n_table++ : CPA_NONE;
        x.rlc0 = 0; x.chunk0 = (u32)pl.chunks.size(); x.n_chunks = 0;
        if (x.dst_tag == CPA_RLC_ACC) {
            x.rlc0 = pl.n_rlc;
            pl.n_rlc += x.length;
            for (u64 g = 0; g < x.length; g += CPA_CHUNK) {
                CpaChunk c;
                c.event = (u32)e; c.start = (u32)g; c.count = (u32)(x.length - g < CPA_CHUNK ? x.length - g : CPA_CHUNK); c.pad = 0;
                pl.chunks.push_back(c);
                x.n_chunks++;
            }
        }
        pl.row0[e] = pl.n_rows;
        pl.n_rows += 2 * x.length;
        pl.n_rw += (x.src_tag == CPA_MEMORY ? n_real : 0) + ((x.dst_tag == CPA_MEMORY || x.dst_tag == CPA_TX_LOG) ? x.length : 0);
        ARG_TRY(pl.n_rows < (1ull << 32) && pl.n_rw < (1ull << 32), "zk_copy_assign: too many rows");
    }
    pl.row0[n] = pl.n_rows;
    pl.n_data = n ? data_offsets[n] : 0;
    return 0;
}
