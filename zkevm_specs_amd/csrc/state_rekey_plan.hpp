// Host side of the RW -> State re-keying (state_rekey.hpp), shared by the HIP library and the CPU backend: the compact-key plan
// from the per-class field masks the scan produced (integers only).
#pragma once
#include <string.h>
#include <vector>
#include "state_rekey.hpp"

#define RWK_MASK_WORDS_H (RWK_NCLASSES * RWK_NFIELDS * 8)
#ifndef RWK_RANK_MAX_ROWS
#define RWK_RANK_MAX_ROWS 16384u  // a class this small has its wide fields ranked (all-pairs counting: count^2 comparisons)
#endif
#ifndef RWK_RANK_MIN_BITS
#define RWK_RANK_MIN_BITS 48u     // ... when the field varies in more bits than this
#endif

struct RwkHostPlan {
    RwkPlan plan;
    std::vector<RwkRankJob> jobs;
    bool rank_field[RWK_NFIELDS] = {false, false, false, false, false};
    u64 n_kept = 0;  // rows of classes 1..11
};
static inline u32 rwk_bitlen(u32 x) { u32 n = 0; while (x) { n++; x >>= 1; } return n; }

// masks: u32[2][16][5][8] (OR, OR of complements) + u32[16] counts, as rwk_scan_kernel / the CPU scan leave them.
static inline void rwk_build_plan(const u32* masks, bool allow_ranks, RwkHostPlan& hp) {
    memset(&hp.plan, 0, sizeof(hp.plan));
    hp.jobs.clear();
    hp.n_kept = 0;
    const u32* cnt = masks + 2 * RWK_MASK_WORDS_H;
    u32 max_width = 0, job_base = 0;
    for (u32 c = 1; c < RWK_CLASS_DROPPED; c++) {
        if (!cnt[c]) continue;
        hp.n_kept += cnt[c];
        RwkClassPlan& cp = hp.plan.cls[c];
        for (u32 f = 0; f < RWK_NFIELDS; f++) {
            RwkRun runs[8];
            u32 n_runs = 0, width = 0;
            for (int w = 7; w >= 0; w--) {
                const u32 slot = (c * RWK_NFIELDS + f) * 8 + (u32)w;
                const u32 vary = masks[slot] & masks[RWK_MASK_WORDS_H + slot];  // 1 somewhere and 0 somewhere
                if (!vary) continue;
                const u32 lo = (u32)__builtin_ctz(vary), hi = 32u - (u32)__builtin_clz(vary);
                runs[n_runs].field = (uint8_t)f; runs[n_runs].word = (uint8_t)w; runs[n_runs].shift = (uint8_t)lo; runs[n_runs].nbits = (uint8_t)(hi - lo);
                n_runs++;
                width += hi - lo;
            }
            const u32 rank_bits = rwk_bitlen(cnt[c] - 1u) ? rwk_bitlen(cnt[c] - 1u) : 1u;
            if (allow_ranks && width > RWK_RANK_MIN_BITS && cnt[c] <= RWK_RANK_MAX_ROWS && hp.jobs.size() < RWK_MAX_JOBS && rank_bits < width) {
                RwkRankJob j;
                j.cls = c; j.field = f; j.base = job_base; j.count = cnt[c];
                job_base += cnt[c];
                hp.jobs.push_back(j);
                hp.rank_field[f] = true;
                RwkRun& r = cp.runs[cp.n_runs++];
                r.field = (uint8_t)(RWK_F_RANK0 + f); r.word = 0; r.shift = 0; r.nbits = (uint8_t)rank_bits;
                cp.width += rank_bits;
            } else {
                for (u32 q = 0; q < n_runs; q++) cp.runs[cp.n_runs++] = runs[q];
                cp.width += width;
            }
        }
        if (cp.width > max_width) max_width = cp.width;
    }
    hp.plan.key_bits = 4u + max_width;
    hp.plan.key_words = (hp.plan.key_bits + 31u) / 32u;
    hp.plan.n_passes = (hp.plan.key_bits + 7u) / 8u;
}
