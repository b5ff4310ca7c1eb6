/* zkevm_hip.h — C ABI of the MI355X constraint-evaluation engine (libzkevm_hip.so).
 *
 * The reference (privacy-scaling-explorations/zkevm-specs, pure Python) has no FFI: its seam
 * is the set of module-level callables its tests import by name.  Each entry
 * point below replaces the per-row Python loop of one of them; INTEGRATION.md shows the
 * ctypes stub a maintainer would add on the reference side.
 *
 * Wire format: one field cell = 4 x uint64 little-endian canonical (== `FQ.n`,
 * src/zkevm_specs/util/arithmetic.py:41-63).  Witness rows are column-major
 * uint64[n_cells][n_rows][4]; lookup tables are row-major uint64[n_rows][n_cells][4];
 * per-row type bits (is_word of WordOrValue cells, util/arithmetic.py:171-195) are uint32[n_rows].
 *
 * Ownership: the caller owns every buffer for the duration of the call (or of the session
 * for zk_*_open); nothing is retained after zk_*_close / after a one-shot call returns.
 * Immutability: the tables a session was opened over are IMMUTABLE for the session's lifetime, also with
 * ZK_OPT_DEVICE_PTRS (where the session reads the caller's buffers in place).  zk_*_open derives state from them that
 * later passes rely on — lookup indices, packed key / step / bytecode records, the dense-RW verdict, EndBlock's whole-table
 * aggregates, and (EVM) which of the warm / cold lane ranges are empty, after which those kernels are no longer launched.
 * A witness that changed is a new witness: close the session and open another one (an EVM open at 2^18 steps is ~0.1 ms
 * of device time).  Editing a table between passes is undefined: stale records would be evaluated, silently.
 * Errors: every function returns 0 on success and a negative code on infrastructure errors
 * (bad arguments, HIP failures — text via zk_last_error()); constraint failures are NOT
 * errors, they are reported in zk_result.  Nothing throws across this boundary.
 * Threading / re-entrancy: a session (zk_session*) is the context.  It captures the device and the stream that are
 * current for the calling thread when it is opened, owns its buffers, events and tally, and every later call on it
 * (zk_launch / zk_collect / zk_read_status / zk_close) works on that device and stream whatever other sessions or
 * threads do in between.  Different sessions may be driven from different threads concurrently; one session is driven
 * by one thread at a time.  zk_init / zk_set_stream select the *calling thread's* current device / stream,
 * zk_last_error returns the calling thread's last error text.  The only process-wide state is one lazily created
 * stream per device.
 */
#ifndef ZKEVM_HIP_H
#define ZKEVM_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status code of one row/step: 0 = satisfied, else (kind << 24) | site.  `kind` is the Python
 * exception class the reference raises at the first failing check of that row. */
enum zk_kind {
    ZK_KIND_OK = 0,
    ZK_KIND_ASSERTION_ERROR = 1,      /* constrain_* / assert (caught by verify_steps, main.py:36) */
    ZK_KIND_CONSTRAINT_UNSAT = 2,     /* ConstraintUnsatFailure raised (instruction.py:483,534) */
    ZK_KIND_LOOKUP_UNSAT = 3,         /* LookupUnsatFailure (table.py:688,880) */
    ZK_KIND_LOOKUP_AMBIGUOUS = 4,     /* LookupAmbiguousFailure (table.py:882) */
    ZK_KIND_WRONG_QUERY_KEY = 5,
    ZK_KIND_NOT_IMPLEMENTED = 6,      /* NotImplementedError (main.py:63) */
    ZK_KIND_TYPE_ERROR = 7,
    ZK_KIND_OVERFLOW_ERROR = 8,
    ZK_KIND_VALUE_ERROR = 9,
    ZK_KIND_ZERO_DIVISION = 10,
    ZK_KIND_INDEX_ERROR = 12,
    ZK_KIND_ATTRIBUTE_ERROR = 13,
    ZK_KIND_NAME_ERROR = 11,          /* UnboundLocalError (execution/block_ctx.py:24) */
    ZK_KIND_UNSUPPORTED = 15          /* gadget not implemented by this engine */
};

typedef struct zk_result {
    uint64_t fail_count;       /* rows whose status != 0 */
    uint64_t first_fail_row;   /* smallest failing row index, UINT64_MAX if none */
    uint32_t first_fail_code;  /* status code of that row */
    uint32_t launches;         /* kernel launches covered by this result */
    uint64_t rows_evaluated;   /* rows per launch */
    double kernel_ms;          /* mean device time of the evaluation kernel per launch (HIP events) */
} zk_result;

#define ZK_OPT_DEVICE_PTRS 1u  /* every data pointer is a device (HBM) pointer */
/* State rows without the limb / byte columns (zk_state_open, zk_state_assign_open, zk_state_assign_from_rw_open): the witness is
 * uint64[15][n][4] — rw_counter, is_write, tag, id, address, field_tag, storage_key lo / hi, then value lo / hi, initial_value lo / hi,
 * root lo / hi, lexicographic_ordering_selector (columns 0..7 and 50..56 of the 57-cell row) — and the ten 16-bit address limbs and 32
 * storage-key bytes (columns 8..49; Row.key2_limbs / key45_bytes, state_circuit.py:63-96) are DERIVED from the address and storage-key
 * cells where the circuit's checks use them, as assign_state_circuit's op2row derives them (:834-842).  For witnesses assigned on the
 * device: the assignment does not write, and the circuit does not read back, 1,344 of a row's 1,824 bytes.  What it gives up: the limb /
 * byte cells are no independent inputs, so their range and recomposition checks (state_circuit.py:505-517) reduce to "address < 2^160,
 * key halves < 2^128" (sites 5 and 7).  Every other check, the ordering, the lookups and the per-tag rules are evaluated as always.
 * Not for witnesses from outside (a tampered limb cell cannot be expressed): those use the 57-cell form. */
#define ZK_OPT_STATE_COMPACT 32u
#define ZK_OPT_BLOCK_STATE_ROWS 64u /* zk_block_verify: materialise the State witness (57-cell rows) instead of the fused form */

/* Select the GPU (HIP ordinal) for the calling thread; creates that device's engine stream on first use.  Idempotent.
 * Selecting another device drops a stream set with zk_set_stream (it belongs to the previous device). */
int zk_init(int device);
void zk_shutdown(void);
/* Sessions opened by this thread from now on run on the caller's HIP stream (e.g. torch's current stream);
 * NULL = the engine's own (non-blocking) stream of the device.  NOTE: HIP's legacy default stream is handle 0 = NULL
 * here, i.e. "engine's own": to order the engine with work of another library, hand over a real (non-default) stream. */
int zk_set_stream(void* hip_stream);
const char* zk_last_error(void);
/* BN254-Fr vector ops on the device: op 0 add, 1 sub, 2 mul, 3 montmul, 4 neg, 5 inv (of a; inv(0) = 0 as py_ecc's
 * prime_field_inv), 6 div (a * inv(b)); 16 / 17: the secp256k1 BASE-field product a * b / square a^2 mod 2^256 - 2^32 - 977
 * (operands are residues below that prime; unit-test hooks of the ECDSA kernel's multiplier).
 * (reference: FQ.__add__/__sub__/__mul__/__neg__/__truediv__ via py_ecc and FQ.inv, util/arithmetic.py:41-60) */
int zk_fr_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out, uint64_t n, uint32_t opts);

/* ---- State circuit: replaces the `for row: check_state_row(row, prev, next, tables)` loop
 *      (src/zkevm_specs/state_circuit.py:492; loop tests/test_state_circuit.py:26-30).
 *      rows: uint64[57][n][4] + flags uint32[n] (bit0 value.is_word, bit1 initial.is_word);
 *      mpt: uint64[n_mpt][12][4] (MPTTableRow, evm_circuit/table.py:461-468). */
typedef struct zk_session zk_session;
int zk_state_open(const uint64_t* rows, const uint32_t* flags, uint64_t n,
                  const uint64_t* mpt, uint64_t n_mpt, uint32_t opts, zk_session** out);
/* Multi-GPU row sharding: evaluate only rows [row_lo, row_hi) of the uploaded block; the rows
 * outside the range are a read-only halo (prev/next of the boundary rows).  Default: all rows,
 * neighbours wrapping modulo n as in the reference's driver. */
int zk_state_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi);
/* The same for every row-circuit session (State, Bytecode, Copy, Exp, Tx / Sig units, PI): a rank of a row-sharded
 * run opens its session over rows [lo, hi + halo) of the global witness (halo = the rows after the range that the
 * range's last rows read: State 1 before + 1 after, Bytecode / Exp / PI 1 after, Copy 2 after, Tx / Sig none;
 * SURVEY.md §8e) and evaluates [0, hi - lo) (State: [1, 1 + hi - lo)).  zk_result.rows_evaluated reports the range.
 * EVM sessions shard by the step rows they are opened over (pairs [lo, hi) need steps [lo, hi]). */
int zk_set_range(zk_session* s, uint64_t row_lo, uint64_t row_hi);
/* One-shot convenience: open + launch + collect (+ copy per-row status to host) + close. */
int zk_state_verify(const uint64_t* rows, const uint32_t* flags, uint64_t n,
                    const uint64_t* mpt, uint64_t n_mpt, uint32_t opts,
                    uint32_t* status_out /* nullable, n entries, host unless DEVICE_PTRS */,
                    zk_result* result);

/* ---- EVM circuit: replaces the `for (curr, next) in zip(steps, steps[1:]): verify_step(...)` loop of
 *      verify_steps (src/zkevm_specs/evm_circuit/main.py:14-44): one status per step PAIR
 *      (n_steps - 1 of them).  steps: ROW-major uint64[n_steps][13][4] (StepState, step.py:16-75 —
 *      lanes visit steps in state-sorted order, so each step's cells stay contiguous; the dummy
 *      EndBlock step of main.py:21-22 is appended by the caller when end_with_last_step);
 *      rw uint64[n][14][4] + flags (bit0 value.is_word, bit1 value_prev.is_word) — RWTableRow,
 *      table.py:447-457; bytecode uint64[n][6][4] (:438-443); tx uint64[n][5][4] + flags (:421-426);
 *      block uint64[n][4][4] + flags (:413-417).  Fixed-table lookups (table.py:14-103) are
 *      evaluated in closed form on the device.  Tables are sets: callers de-duplicate rows. */
typedef struct zk_evm_tables {
    const uint64_t* steps;      uint64_t n_steps;
    const uint64_t* rw;         const uint32_t* rw_flags;    uint64_t n_rw;
    const uint64_t* bytecode;   uint64_t n_bytecode;
    const uint64_t* tx;         const uint32_t* tx_flags;    uint64_t n_tx;
    const uint64_t* block;      const uint32_t* block_flags; uint64_t n_block;
    uint32_t begin_with_first_step;
    uint32_t end_with_last_step;
    /* tables only the copy / SHA3 / EXP gadgets look up (Tables.copy_table / keccak_table / exp_table,
     * table.py:614-619); n == 0 when absent.  copy uint64[n][14][4] (CopyTableRow :494-507: is_first,
     * src_id lo/hi, src_tag, dst_id lo/hi, dst_tag, src_addr, src_addr_end, dst_addr, length, rlc_acc,
     * rw_counter, rwc_inc); keccak uint64[n][5][4] (KeccakTableRow :511-515); exp uint64[n][11][4]
     * (ExpTableRow :538-548: is_step, identifier, is_last, base_limb0..3, exponent lo/hi,
     * exponentiation lo/hi). */
    const uint64_t* copy;       uint64_t n_copy;
    const uint64_t* keccak;     uint64_t n_keccak;
    const uint64_t* exp;        uint64_t n_exp;
    /* StepState.aux_data (step.py:44), optional (NULL = absent for every step): aux uint64[n_steps][aux_cells][4] and
     * aux_kind uint32[n_steps]: 0 none, 1 Word (lo, hi), 2 int < 2^256 (lo, hi), 3 pair of ints < p (CALL into a
     * precompile: input / return length), 4 not representable (a gadget that reads it reports ZK_UNSUPPORTED),
     * 5 ecRecover [PrecompileAuxData, randomness]: msg_hash, sig_v, sig_r, sig_s as lo/hi, recovered_addr, input_rlc,
     * output_rlc, keccak_randomness (12 cells, ecrecover.py:15-44), 6 ecAdd [px, py, qx, qy, outx, outy] (10 cells),
     * 7 ecMul [px, py, s, outx, outy] (8 cells), 8 ecPairing [input_rlc, input_pairs, is_valid_input, output] (4). */
    const uint64_t* aux;        const uint32_t* aux_kind;
    /* Tables.withdrawal_table (WithdrawalTableRow, table.py:430-434: id, validator_id, address, amount), optional:
     * uint64[n][4][4], sorted by id (the order end_block.py:152 walks them in).  Only EndBlock's last step reads it. */
    const uint64_t* withdrawals; uint64_t n_withdrawals;
    /* Tables.sig_table / ecc_table (table.py:552-575), optional: sig uint64[n][9][4] (msg_hash lo/hi, sig_v, sig_r lo/hi,
     * sig_s lo/hi, recovered_addr, is_valid); ecc uint64[n][13][4] (op_type, px lo/hi, py lo/hi, qx lo/hi, qy lo/hi,
     * input_rlc, out_x, out_y, is_valid).  Only the ecRecover / ecAdd / ecMul / ecPairing precompile states read them. */
    const uint64_t* sig;        uint64_t n_sig;
    const uint64_t* ecc;        uint64_t n_ecc;
    uint32_t aux_cells;         /* cells per step in `aux`: 0 = 2 (kinds 0-3), 12 when kinds 5-8 are present */
    uint32_t reserved;
} zk_evm_tables;
#define ZK_OPT_NO_STATE_SORT 2u /* evaluate step pairs in trace order (no state-sorted lane mapping) */
#define ZK_OPT_GENERIC_INDEX 4u /* skip the dense RW index / bytecode directory; open-addressing indices only */
#define ZK_OPT_SINGLE_PASS 8u   /* EVM sessions: the session will evaluate ONE pass (what zk_evm_verify does): skip the packed step
                                 * records, a one-off streaming pass over the step table that only pays for itself from the second
                                 * evaluation pass on; results are identical either way */
#define ZK_OPT_SIDE_STREAM 16u  /* EVM sessions: every pass runs its warm / cold gadget launches on a second stream of the device
                                 * beside the hot one (fork and join by events, results final in the session's stream order as
                                 * always).  Costs two cross-queue barriers (~8 us) and saves the sum of the two launches: a gain
                                 * when the device is shared with other sessions' passes (the Super circuit), a loss alone */
int zk_evm_open(const zk_evm_tables* t, uint32_t opts, zk_session** out);
int zk_evm_verify(const zk_evm_tables* t, uint32_t opts,
                  uint32_t* status_out /* nullable, n_steps-1 entries */, zk_result* result);
/* Batch form: n independent witnesses (e.g. the blocks of a queue), each verified exactly as zk_evm_verify does — its own open
 * (indices, key records, sort), one evaluation pass, its own tally in results[i] — software-pipelined two deep on two streams of
 * the device: the HBM-bound open of witness i + 1 runs under the latency-bound evaluation of witness i.  The caller's current
 * stream is synchronised first (uploads it enqueued are complete).  Per-pair statuses are not returned. */
int zk_evm_verify_batch(const zk_evm_tables* const* tables, uint64_t n, uint32_t opts, zk_result* results);

/* ---- Bytecode circuit: replaces the `for row: check_bytecode_row(row, next, push_table, keccak_table, r)`
 *      loop (src/zkevm_specs/bytecode_circuit.py:37-100; loop tests/test_bytecode_circuit.py:26-47, next row
 *      wraps modulo n).  rows: column-major uint64[12][n][4] (Row, bytecode_circuit.py:15-26: q_first,
 *      q_last, hash lo, hi, tag, index, value, is_code, push_data_left, value_rlc, length,
 *      push_data_size); keccak: uint64[m][5][4] (KeccakTableRow, table.py:511-515); randomness: one
 *      cell (keccak_randomness).  The push table (:174-179) is evaluated in closed form. */
int zk_bytecode_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak,
                     const uint64_t* randomness, uint32_t opts, zk_session** out);
int zk_bytecode_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak,
                       const uint64_t* randomness, uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- Exp circuit: replaces the loop of verify_exp_circuit (src/zkevm_specs/exp_circuit.py:88-97,
 *      verify_step :14-85; next row wraps).  rows: column-major uint64[21][n][4] (ExpCircuitRow,
 *      table.py:519-535: q_usable, is_step, identifier, is_last, base, exponent, exponentiation,
 *      a, b, c, d, q as lo/hi pairs, r). */
int zk_exp_open(const uint64_t* rows, uint64_t n, uint32_t opts, zk_session** out);
int zk_exp_verify(const uint64_t* rows, uint64_t n, uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- Copy circuit: replaces the loop of verify_copy_table (src/zkevm_specs/copy_circuit.py:92-130:
 *      verify_row :23-59 + verify_step :62-89 on the window (i, i+1, i+2) mod n, then the RW / bytecode /
 *      tx lookups of row i).  rows: column-major uint64[20][n][4] (CopyCircuitRow, table.py:472-491: q_step,
 *      is_first, is_last, id lo, hi, tag, addr, src_addr_end, bytes_left, value, rlc_acc, is_code, is_pad,
 *      rw_counter, rwc_inc_left, is_memory, is_bytecode, is_tx_calldata, is_tx_log, is_rlc_acc) + row_flags
 *      (bit0 id.is_word); randomness: one cell; tables as for the EVM circuit. */
typedef struct zk_copy_tables {
    const uint64_t* rows;       const uint32_t* row_flags;   uint64_t n_rows;
    const uint64_t* randomness;
    const uint64_t* rw;         const uint32_t* rw_flags;    uint64_t n_rw;
    const uint64_t* bytecode;   uint64_t n_bytecode;
    const uint64_t* tx;         const uint32_t* tx_flags;    uint64_t n_tx;
} zk_copy_tables;
int zk_copy_open(const zk_copy_tables* t, uint32_t opts, zk_session** out);
int zk_copy_verify(const zk_copy_tables* t, uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- Tx and Sig circuits: replace the per-tx loop of tx_circuit.verify_circuit (src/zkevm_specs/tx_circuit.py:
 *      253-291: SignVerifyChip.verify :205-243 + the copy constraints to the tx-table rows) and the per-row
 *      loop of sig_circuit.verify_circuit (sig_circuit.py:113-122: Row.verify :64-104).  One unit = one tx
 *      slot / one signature row.  bytes: uint8[n][9][32] (pk_x, pk_y, ecdsa pk_x, pk_y, msg_hash_bytes, ecdsa
 *      msg_hash_bytes, pub_key_hash, ecdsa sig_r LE, ecdsa sig_s LE); cells: column-major uint64[8][n][4]
 *      (address, msg_hash lo, hi, sig_v, sig_r lo, hi, sig_s lo, hi); meta: uint32[n][4] (ecdsa_status: 0
 *      verified / 1 not verified / (kind<<24) exception of the third-party secp256k1 call, expected is_valid,
 *      malformed-attribute mask, the ECDSA chip's v for Sig units); keccak: uint64[m][5][4] (is_enabled, input_rlc, input_len, output lo, hi;
 *      tx_circuit.py:38-61); tx_rows: uint64[rows][5][4] + flags (Tx circuit only). */
typedef struct zk_sign_units {
    const uint8_t* bytes;       const uint64_t* cells;       const uint32_t* meta;      uint64_t n_units;
    const uint64_t* randomness;
    const uint64_t* keccak;     uint64_t n_keccak;
    const uint64_t* tx_rows;    const uint32_t* tx_flags;    uint64_t n_tx_rows;
    uint32_t is_sig;            /* 0 = Tx circuit semantics, 1 = Sig circuit semantics */
} zk_sign_units;
int zk_sign_open(const zk_sign_units* t, uint32_t opts, zk_session** out);
int zk_sign_verify(const zk_sign_units* t, uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- Keccak table generation: one table row per byte string.
 *      mode 0 = KeccakCircuit.add (evm_circuit/typing.py:854-865; assign_keccak_table,
 *               bytecode_circuit.py:182-186): row = (2, RLC(reversed(data), r), len, Word(int.from_bytes(digest, "big")));
 *      mode 1 = KeccakTable.add (util/tables.py:18-27, tx_circuit.py:48-58):
 *               row = (1, RLC(reversed(data), r, n_bytes = 64), len, Word(digest bytes)); inputs longer
 *               than 64 bytes are rejected with status ZK_KIND_VALUE_ERROR like the reference's RLC.
 *      data: the messages back to back; offsets: uint64[n_msgs + 1] byte offsets into data
 *      (message i = data[offsets[i] .. offsets[i+1])); rows: uint64[n_msgs][5][4] row-major, the
 *      layout every keccak-table argument above takes.  With ZK_OPT_DEVICE_PTRS data, offsets,
 *      randomness and rows are device pointers (rows_dev may be NULL: the session then owns the
 *      rows and zk_keccak_read_rows copies them out).  zk_launch / zk_collect / zk_read_status work
 *      as for the circuits (status = per-message code, kernel_ms = device time of the pass). */
int zk_keccak_open(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs,
                   const uint64_t* randomness, uint32_t mode, uint64_t* rows_dev, uint32_t opts, zk_session** out);
int zk_keccak_read_rows(zk_session* s, uint64_t* rows_host);
int zk_keccak_table(const uint8_t* data, uint64_t n_bytes, const uint64_t* offsets, uint64_t n_msgs,
                    const uint64_t* randomness, uint32_t mode, uint64_t* rows_out, uint32_t opts,
                    uint32_t* status_out, zk_result* result);

/* ---- State-circuit witness assignment: replaces assign_state_circuit
 *      (src/zkevm_specs/state_circuit.py:855-884: op2row :827-852 per op + the root back-fill :866-878) and
 *      mpt_table_from_ops (:887-888, _mock_mpt_updates :904-934).
 *      ops: COLUMN-major uint64[12][n][4], one slot per field of `Operation` (:616-630): rw_counter, rw, tag, id,
 *      address, field_tag, storage_key as the op's 256-bit Python ints (NOT reduced mod p: `FQ(...)` happens on the
 *      device, and `_mpt_key` compares the raw tag), then value lo/hi, initial_value lo/hi and
 *      lexicographic_ordering_selector as field cells.  op_flags uint32[n]: bit0 value.is_word, bit1
 *      initial_value.is_word, bit2 isinstance(field_tag, AccountFieldTag) (selects the account proof types, :915).
 *      Outputs: rows uint64[57][n][4] + row_flags uint32[n] — exactly what zk_state_open takes — and the mock MPT
 *      rows uint64[n_mpt][12][4] in first-occurrence order (capacity: n rows).  Per-op status: AssertionError (site 1
 *      value / 2 initial_value: Word(x.int_value()) >= 2^256 inside _mock_mpt_updates, first op of a key only),
 *      OverflowError (site 3: address wider than 160 bits in op2row); the reference raises the first site-1/2
 *      failure if there is one, else the first site-3 failure.
 *      With ZK_OPT_DEVICE_PTRS every pointer is a device pointer and rows_dev / row_flags_dev / mpt_dev (each
 *      nullable: the session then owns the buffer) receive the outputs in place. */
int zk_state_assign_open(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_dev,
                         uint32_t* row_flags_dev, uint64_t* mpt_dev, uint32_t opts, zk_session** out);
/* Copy the outputs of the last pass to HOST buffers (each nullable); n_mpt_out = number of MPT rows. */
int zk_state_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* mpt_host,
                         uint64_t mpt_capacity_rows, uint64_t* n_mpt_out);
int zk_state_assign(const uint64_t* ops, const uint32_t* op_flags, uint64_t n, uint64_t* rows_out,
                    uint32_t* row_flags_out, uint64_t* mpt_out /* capacity n rows */, uint64_t* n_mpt_out,
                    uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- RW table -> State-circuit operations (the "RW-table lexicographic sort" half): from the EVM circuit's
 *      RW table to the op list zk_state_assign_open takes, on the device.  The reference never links the two in code (SURVEY.md
 *      Appendix A.14); its State witnesses are lists of `Operation`s (src/zkevm_specs/state_circuit.py:616-825) in the order
 *      the circuit checks — (tag, id, address, field_tag, storage_key, rw_counter) strictly increasing (:552-570) — handed to
 *      assign_state_circuit (:855-884), and a block's RW rows come out of `RWDictionary` (evm_circuit/typing.py:464-845) under
 *      the EVM side's `Target` numbering (evm_circuit/table.py:184-204).  This entry re-keys every RW row (Target -> Tag; the
 *      CallContext field tag from the address cell; Account rows without an id; the TxLog address cell unpacked into log_id /
 *      field_tag / index; initial_value = aux0 for Account / AccountStorage rows), sorts the ops by that key — stable: equal
 *      keys keep their table order, like Python's sorted() — and puts a StartOp in front.
 *      rw: uint64[n][14][4] + rw_flags uint32[n] (nullable), the EVM circuit's RW table as in zk_evm_tables.
 *      Outputs: ops COLUMN-major uint64[12][n_ops][4] + op_flags uint32[n_ops], exactly what zk_state_assign_open takes;
 *      n_ops = 1 + the rows kept, known when zk_state_ops_from_rw_open returns (*n_ops_out; at most n + 1).  Left out:
 *      CallContext rows whose field tag exceeds the State circuit's MAX_FIELD_TAG (24, state_circuit.py:34,334) — not an error —
 *      and rejected rows: status (ZK_KIND_VALUE_ERROR << 24) | 1 = the target cell is not a Target, (ZK_KIND_OVERFLOW_ERROR
 *      << 24) | 2 = storage_key hi cell >= 2^128 (lo | hi << 128 does not fit the 256-bit slot).  Status is per RW ROW (n
 *      entries); the tally counts the rejected rows.
 *      With ZK_OPT_DEVICE_PTRS rw / rw_flags are device pointers and ops_dev / op_flags_dev (each nullable: the session then owns
 *      the buffer; capacity (n + 1) ops when given) receive the outputs in place, packed for n_ops.
 *      zk_state_ops_from_rw_open runs the class scan (one streaming pass over the table) and waits for it: the sort key is
 *      compacted to the bits that vary inside each tag class, which decides the number of radix passes.  zk_launch enqueues the
 *      rest (key packing, the radix passes, the op list) asynchronously; zk_collect / zk_read_status as for the circuits. */
int zk_state_ops_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_dev, uint32_t* op_flags_dev,
                              uint32_t opts, uint64_t* n_ops_out, zk_session** out);
/* Copy the outputs of the last pass to HOST buffers (each nullable; ops_host holds n_ops ops). */
int zk_state_ops_from_rw_read(zk_session* s, uint64_t* ops_host, uint32_t* op_flags_host, uint64_t* n_ops_out);
int zk_state_ops_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint64_t* ops_out /* capacity n + 1 ops */,
                         uint32_t* op_flags_out, uint64_t* n_ops_out, uint32_t opts, uint32_t* status_out, zk_result* result);

/* The two steps in one session — RW table in, State-circuit witness out: the re-keying and the sort as above, then
 * assign_state_circuit / mpt_table_from_ops (zk_state_assign_open) over ops that are read straight from the RW rows through the
 * sorted order; the op list is never written to memory.  Outputs as zk_state_assign_open's, for n_ops rows (*n_ops_out, known
 * when this returns; buffers given with ZK_OPT_DEVICE_PTRS have capacity n_rw + 1 rows and are packed for n_ops:
 * rows uint64[57][n_ops][4], row_flags uint32[n_ops], mpt uint64[n_mpt][12][4]).  zk_launch / zk_collect / zk_read_status
 * (one code per OP) / zk_state_assign_read as for zk_state_assign_open; RW rows the re-keying rejects (codes above) count in
 * the same tally, first_fail_row then being the RW row. */
int zk_state_assign_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint64_t* rows_dev,
                                 uint32_t* row_flags_dev, uint64_t* mpt_dev, uint32_t opts, uint64_t* n_ops_out, zk_session** out);

/* RW table in, State-circuit VERDICT out — `check_state_circuit(assign_state_circuit(ops, r), ...)` over the ops of an EVM-circuit RW
 * table without the witness in between (state_circuit.py:827-889 then :492-613; tests/test_state_circuit.py's `verify(ops, tables,
 * randomness)` is this composition): the re-keying, the sort and the mock MPT updates as zk_state_assign_from_rw_open, but op2row's
 * rows are evaluated in the registers they are computed in and never stored (1,824 B per row that the two-step form writes and reads
 * back).  zk_launch (status_dev: uint32[n_ops], the State circuit's code per row) / zk_collect (the State circuit's zk_result) /
 * zk_read_status as for zk_state_open; results identical to zk_state_assign_from_rw_open + zk_state_open on its outputs.  An RW row
 * the re-keying rejects or an op assign_state_circuit raises on (codes of zk_state_assign_*) means there is no witness: zk_collect
 * then returns -1 with the count, the first row and its code in zk_last_error() (libzkevm_cpu.so: the open does).
 * Resident passes: the RW table of a session does not change, so once a zk_collect has seen a pass with a witness the session keeps the
 * sorted order, the first-access links, the MPT rows and the root ranks, and every later zk_launch enqueues the evaluation kernel
 * alone (libzkevm_hip.so; 694,509 rows: 0.29 ms for the first pass, 0.13-0.15 ms for the later ones). */
int zk_state_verify_from_rw_open(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n_rw, uint32_t opts, uint64_t* n_ops_out,
                                 zk_session** out);
int zk_state_verify_from_rw(const uint64_t* rw, const uint32_t* rw_flags, uint64_t n, uint32_t opts, uint32_t* status_out /* n_ops <= n + 1 codes */,
                            uint64_t* n_ops_out, zk_result* result);

/* ---- secp256k1 ECDSA verification: computes the `ecdsa_status` column of the Tx / Sig units
 *      on the device instead of taking it from the host.  Replaces `ECDSAVerifyChip.verify`
 *      (src/zkevm_specs/tx_circuit.py:147-158, util/ec.py:109-117), i.e. eth-keys 0.4.0's
 *      `KeyAPI.Signature(vrs=...)` validation + `KeyAPI().ecdsa_verify(msg_hash, signature, public_key)`.
 *      bytes: layout 0 = packed uint8[n][5][32]: pk_x LE, pk_y LE, msg_hash BE, sig_r LE, sig_s LE;
 *             layout 1 / 2 = the byte rows uint8[n][9][32] of the Tx / Sig units of zk_sign_units (rows 2, 3, 5, 7, 8
 *             are those five; the Tx chip keeps msg_hash little-endian, tx_circuit.py:131, the Sig chip big-endian,
 *             util/ec.py:93).
 *      v: optional recovery ids, v[i * v_stride] (the Sig circuit's chip builds Signature(vrs=[v, r, s]) — for Sig
 *      units pass meta + 3 with stride 4; NULL = the Tx circuit's fixed 0).  Status per signature: 0 verified, 1 not verified, (ZK_KIND_UNSUPPORTED << 24) | 1 = eth_keys
 *      BadSignature (v outside {0, 1}, r or s outside (0, N)), (ZK_KIND_UNSUPPORTED << 24) | 2 = a public key with y == P exactly
 *      (the one value outside the engine's domain: eth-keys' `if not p[1]` sees it as non-zero; any other coordinate >= P is
 *      reduced mod P, as eth-keys' formulas do implicitly).  Public keys that are not on the curve are evaluated with eth-keys'
 *      Jacobian case analysis (a Y == 0 point is the point at infinity, inv(0) == 0), so their verdicts are reproducible too.
 *      These verdicts for keys off the curve / coordinates >= P are those of eth-keys' NATIVE backend (pure Python,
 *      eth_keys/backends/native/ecdsa.py — what the reference's pinned environment uses); an environment with `coincurve`
 *      installed makes KeyAPI use libsecp256k1, which rejects such keys up front.  Well-formed keys verify identically either way.
 *      out_dev (DEVICE pointer, optional): out_dev[i * out_stride] = status, e.g. meta + 0 with stride 4 to fill the
 *      units' meta column in place.  zk_launch / zk_collect / zk_read_status as for the circuits (the tally counts
 *      the signatures that did not verify).
 *      Device memory: the kernel keeps a 1,440-byte table of the key's multiples per lane (one lane per signature above 2^16
 *      signatures, a lane pair below, four lanes up to 2^14): at most 2^17 lanes' worth (189 MB) is allocated per session; larger batches run as
 *      consecutive launches over the same tables. */
int zk_ecdsa_open(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n,
                  uint32_t* out_dev, uint32_t out_stride, uint32_t opts, zk_session** out);
/* One or two signature arrays verified by ONE launch (statuses: batch 0's signatures first): the Tx circuit's and the Sig
 * circuit's chips of a block are independent arrays with different layouts, and one dispatch of their union places its
 * wavefronts better than two concurrent launches do (2 x 2^14 signatures: 1.4 ms against 1.9 ms). */
typedef struct zk_ecdsa_batch {
    const uint8_t* bytes; uint32_t layout; const uint32_t* v; uint32_t v_stride; uint64_t n; uint32_t* out_dev; uint32_t out_stride;
} zk_ecdsa_batch;
int zk_ecdsa_open_batches(const zk_ecdsa_batch* batches, uint32_t n_batches /* 1 or 2 */, uint32_t opts, zk_session** out);
int zk_ecdsa_verify(const uint8_t* bytes, uint32_t layout, const uint32_t* v, uint32_t v_stride, uint64_t n,
                    uint32_t opts, uint32_t* status_out, zk_result* result);

/* ---- Bytecode-circuit witness assignment: replaces assign_bytecode_circuit(k, bytecodes,
 *      keccak_randomness) (src/zkevm_specs/bytecode_circuit.py:104-167: push-data tracking, running value_rlc, length,
 *      q_first / q_last, truncation at 2^k rows, EMPTY_HASH padding).
 *      in_rows: the BytecodeTableRows of the UnrolledBytecodes (:31-33) back to back, ROW-major uint64[n_rows][6][4]
 *      (hash lo/hi, tag, index, is_code, value — the EVM circuit's bytecode-table layout, in input order);
 *      offsets uint64[n_codes + 1]: first row of every bytecode (offsets[0] = 0, offsets[n_codes] = n_rows);
 *      lengths uint64[n_codes]: len(bytecode.bytes).  Output: COLUMN-major uint64[12][2^k][4], what zk_bytecode_open
 *      takes.  With ZK_OPT_DEVICE_PTRS rows_dev (nullable: the session then owns the rows) receives them in place.
 *      The assignment itself cannot fail: the tally is always clean. */
int zk_bytecode_assign_open(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths,
                            uint64_t n_codes, uint32_t k, const uint64_t* randomness, uint64_t* rows_dev, uint32_t opts,
                            zk_session** out);
int zk_bytecode_assign_read(zk_session* s, uint64_t* rows_host /* uint64[12][2^k][4] */);
int zk_bytecode_assign(const uint64_t* in_rows, uint64_t n_rows, const uint64_t* offsets, const uint64_t* lengths,
                       uint64_t n_codes, uint32_t k, const uint64_t* randomness, uint64_t* rows_out, uint32_t opts,
                       zk_result* result);

/* ---- Public-inputs (PI) circuit: replaces the `for i: check_row(rows[i], rows[(i + 1) % n], ...)` loop of
 *      pi_circuit.verify_circuit (src/zkevm_specs/pi_circuit.py:447-459; check_row :150-322).  rows: COLUMN-major
 *      uint64[24][n][4] (pi_circuit.Row :104-134 flattened: q_bytes_last, q_tx_table, q_tx_calldata, q_tx_calldata_start,
 *      q_rpi_keccak_lookup, q_rpi_value_start, tx_id_inv, tx_value_lo_inv, tx_id_diff_inv, calldata_gas_cost, is_final,
 *      q_withdrawal_table, rpi_bytes, rpi_bytes_keccakrlc, rpi_value_lc, rpi_digest_word lo, hi, q_rpi_byte_enable,
 *      tx_table.tx_id, .tag, .index, .value.lo, withdrawal_table.id, .amount); keccak: uint64[m][5][4] (is_enabled, input_rlc,
 *      input_len, output lo, hi; KeccakTable :74-101); gas: uint64[k][3][4] (TxCallDataGasCostAccRow :64-68: tx_id, is_final,
 *      gas_cost_acc); circuit_len (Witness.circuit_len); keccak_rand / byte_pow_base: one cell each (the module constants
 *      :834-836).  The fixed u16 table (:351) is a range check.  The copy constraints of verify_circuit (:355-445) compare
 *      table cells with byte strings of the witness on the host (zkevm_specs_amd/pi_circuit.py). */
int zk_pi_open(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
               uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts, zk_session** out);
int zk_pi_verify(const uint64_t* rows, uint64_t n, const uint64_t* keccak, uint64_t n_keccak, const uint64_t* gas, uint64_t n_gas,
                 uint64_t circuit_len, const uint64_t* keccak_rand, const uint64_t* byte_pow_base, uint32_t opts,
                 uint32_t* status_out, zk_result* result);
/* PI circuit copy constraints (pi_circuit.py:355-445): constraint i asserts cells[i] == bytes_to_fq(entry_i[::-1]), entry_i =
 * bytes[i][0 .. lens[i]) (an element of `witness.copy_constrains` as popped, big-endian, left-aligned in its 32-byte slot;
 * lens[i] > 31 fails bytes_to_fq's own assert, util/arithmetic.py:227-229); lens[i] == 0xFFFFFFFF: bytes[i] is a canonical cell
 * (little-endian) compared as is (the word equality of pi_circuit.py:358).  The host side lists the constraints in the
 * reference's statement order; the tally's first failing index is the reference's first failing assert.  Status per
 * constraint: 0, or (ZK_KIND_ASSERTION_ERROR << 24) | site (1 = length assert, 2 = equality). */
int zk_pi_copy_open(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts, zk_session** out);
int zk_pi_copy_verify(const uint64_t* cells, const uint8_t* bytes, const uint32_t* lens, uint64_t n, uint32_t opts,
                      uint32_t* status_out, zk_result* result);

/* ---- Copy-circuit witness assignment: replaces `CopyCircuit.copy(r, rw_dict, src_id, src_tag, dst_id,
 *      dst_tag, src_addr, src_addr_end, dst_addr, copy_length, src_data, log_id)` (src/zkevm_specs/evm_circuit/typing.py:
 *      1010-1091, _append_row :1093-1151), the RW rows it appends to the RWDictionary (memory_read / memory_write /
 *      tx_log_write, :482-492, :532-556) and the copy-table row `Tables._convert_copy_circuit_to_table` derives
 *      (evm_circuit/table.py:627-651).  One event = one call of copy():
 *      events ROW-major uint64[n][12][4]: src_id lo, hi, src_tag, dst_id lo, hi, dst_tag (CopyDataTypeTag, table.py:308-315),
 *             src_addr, src_addr_end, dst_addr, copy_length, log_id, rw_counter (rw_dict.rw_counter when copy() is called);
 *      flags uint32[n]: bit0 src_id is a Word, bit1 dst_id is a Word (WordOrValue type bits of the rows' id);
 *      data uint16[]: per event the source bytes read below src_addr_end (i < copy_length with src_addr + i < src_addr_end),
 *             value | is_code << 8, events back to back; data_offsets uint64[n + 1].
 *      Domain: source values are bytes, addresses / lengths / counters are below 2^62, TxLog is never a source (the
 *      reference asserts that) — anything else is rejected with an error code, never guessed.
 *      Outputs: rows COLUMN-major uint64[20][n_rows][4] + row_flags uint32[n_rows] (n_rows = 2 * sum of the lengths: what
 *      zk_copy_open takes), table uint64[n_table][14][4] (what zk_evm_tables.copy takes; one row per event that copies
 *      something, in event order), rw uint64[n_rw][14][4] + rw_flags uint32[n_rw] (the RW rows, in rw_counter order per
 *      event).  zk_copy_assign_sizes computes the three counts (host arithmetic over the events).  With
 *      ZK_OPT_DEVICE_PTRS every pointer is a device pointer and the outputs land in the caller's buffers (each nullable:
 *      the session then owns it), ready for zk_copy_open / zk_evm_open without leaving the device. */
typedef struct zk_copy_events {
    const uint64_t* events;     const uint32_t* flags;       uint64_t n_events;
    const uint16_t* data;       const uint64_t* data_offsets;
    const uint64_t* randomness;
} zk_copy_events;
int zk_copy_assign_sizes(const zk_copy_events* ev, uint32_t opts, uint64_t* n_rows, uint64_t* n_table, uint64_t* n_rw);
int zk_copy_assign_open(const zk_copy_events* ev, uint64_t* rows_dev, uint32_t* row_flags_dev, uint64_t* table_dev,
                        uint64_t* rw_dev, uint32_t* rw_flags_dev, uint32_t opts, zk_session** out);
int zk_copy_assign_read(zk_session* s, uint64_t* rows_host, uint32_t* row_flags_host, uint64_t* table_host,
                        uint64_t* rw_host, uint32_t* rw_flags_host);
int zk_copy_assign(const zk_copy_events* ev, uint64_t* rows_out, uint32_t* row_flags_out, uint64_t* table_out,
                   uint64_t* rw_out, uint32_t* rw_flags_out, uint32_t opts, zk_result* result);

/* ---- A block as a ONE-SHOT (BASELINE config 5; round 6): everything the Super circuit derives on the device and one evaluation pass of
 *      its six circuits, from the block's raw device-resident inputs, in one call.  The reference has no super-circuit driver (SURVEY.md
 *      Appendix A.14); the inputs are what its own constructors take: the EVM circuit's tables and steps (`Tables`, evm_circuit/table.py:
 *      583-625), the byte strings the block hashes (`KeccakCircuit.add`, evm_circuit/typing.py:854-865: the contracts first, then the
 *      SHA3 inputs), the copy events (`CopyCircuit.copy`, :1010-1091), the Exp circuit's rows, the Tx units.  Derived inside: the keccak
 *      table (zk_keccak_*: rows [0, n_codes) are the Bytecode circuit's table, the rest the EVM circuit's), the Bytecode circuit's rows
 *      (zk_bytecode_assign_*: the unrolled bytecodes ARE evm.bytecode, cut by code_offsets / code_lengths), the Copy circuit's rows and
 *      the EVM circuit's copy table (zk_copy_assign_*; the Copy circuit looks up evm.rw / evm.bytecode / evm.tx), the State circuit's
 *      verdict from evm.rw (zk_state_verify_from_rw_open: the State rows are evaluated where they are computed; with
 *      ZK_OPT_BLOCK_STATE_ROWS or ZK_OPT_STATE_COMPACT the witness is written and read back as zk_state_assign_from_rw_open +
 *      zk_state_open do — the same results, for comparison).  evm.copy / evm.keccak are ignored.  Four host threads drive four chains on
 *      four streams of the calling thread's device: State | keccak of the contracts -> Bytecode circuit | copy assignment -> Copy
 *      circuit, keccak of the SHA3 inputs -> EVM open + pass | Bytecode assignment, Exp, Tx (the EVM circuit's keccak table holds the
 *      SHA3 rows only, so its chain never waits for the long pass over the contracts).
 *      Every pointer is a device pointer (opts must carry ZK_OPT_DEVICE_PTRS; ZK_OPT_STATE_COMPACT is honoured).  results[c]: the
 *      tally of circuit c (enum below; rows_evaluated == 0 for a circuit without rows); a failing witness ASSIGNMENT (keccak input,
 *      State op, ...) is an error return with the text in zk_last_error.  chain_ms (nullable, ten doubles; a measurement aid): host
 *      milliseconds from the call to the end of each chain [0..3], to its start [4..7], to the end of the last one [8] and to the
 *      return [9]. */
typedef struct zk_block {
    zk_evm_tables evm;
    const uint8_t* hashed_data; uint64_t hashed_bytes; const uint64_t* hashed_offsets;  /* n_hashed + 1 offsets */
    uint64_t n_codes, n_hashed;
    const uint64_t* randomness;                                                          /* one cell: the block's keccak randomness */
    const uint64_t* code_offsets; const uint64_t* code_lengths; uint64_t n_bytecodes; uint32_t k; uint32_t reserved;
    zk_copy_events copy_events;                                                          /* n_events == 0: no Copy circuit rows */
    const uint64_t* exp_rows; uint64_t n_exp_rows;                                       /* column-major uint64[21][n][4] */
    zk_sign_units tx;                                                                    /* n_units == 0: no Tx circuit */
} zk_block;
enum { ZK_BLOCK_EVM = 0, ZK_BLOCK_STATE = 1, ZK_BLOCK_BYTECODE = 2, ZK_BLOCK_TX = 3, ZK_BLOCK_COPY = 4, ZK_BLOCK_EXP = 5, ZK_BLOCK_NCIRCUITS = 6 };
int zk_block_verify(const zk_block* b, uint32_t opts, zk_result* results /* [ZK_BLOCK_NCIRCUITS] */, double* chain_ms /* [10], nullable */);

/* ---- Session protocol shared by every circuit.
 * launch: enqueue one evaluation pass (asynchronous) on the session's stream.  status_dev: optional DEVICE buffer of
 *         n uint32 receiving the per-row status codes.  A caller-provided status_dev is final in stream order: once the
 *         work enqueued by zk_launch has completed (the caller's own event / stream synchronisation on the session's stream,
 *         no zk_collect needed) every row's code is there — for EVM sessions this includes the pairs the fast kernel hands
 *         to the general build (malformed word cells, generic-index fallbacks): that kernel is enqueued behind the pass
 *         whenever status_dev is given.  Without status_dev the codes go to the session's own buffer and are final once
 *         zk_collect or zk_read_status has returned (they run the general build only if the pass left such pairs).
 *         One-shot EVM sessions (ZK_OPT_SINGLE_PASS, no status_dev, no ZK_OPT_SIDE_STREAM) launch lazily: zk_launch enqueues the
 *         build that holds BASELINE config 3's states, and the builds for the rarer state groups are enqueued by the first
 *         zk_collect / zk_read_status / zk_launch that follows — only if the open's sort, which tells the host through the session's
 *         page-locked block, found steps of theirs (ZK_LAZY_TAIL=0: always, from zk_launch).  Results are the same either way.
 * collect: wait for all enqueued passes, return the tally of the LAST pass and the mean kernel
 *         time over the passes since the previous collect. */
int zk_launch(zk_session* s, uint32_t* status_dev);
int zk_collect(zk_session* s, zk_result* result);
/* Copy the per-row status of the last pass into a HOST buffer (n entries).  Fails when that pass wrote its statuses
 * to a caller-provided status_dev (they are the caller's then); all zero before the first pass. */
int zk_read_status(zk_session* s, uint32_t* status_host);
/* Device-side time spans, measured by HIP events that ride on the kernel dispatches themselves (measurement aid of bench.py;
 * EVM sessions, -1 where not measured).  zk_session_timing, after a zk_collect: open_ms = first to last kernel of zk_evm_open
 * (index, packed-record and first-pass sort builds), span_ms = first open kernel to the end of the session's FIRST pass.
 * zk_last_timing: the same for the calling thread's last one-shot zk_evm_verify, plus its pass span (== zk_result.kernel_ms). */
int zk_session_timing(zk_session* s, double* open_ms, double* span_ms);
int zk_last_timing(double* open_ms, double* pass_ms, double* span_ms);
/* The same three spans SUMMED over the calling thread's one-shot zk_evm_verify calls since the last reset (sums_ms[3]: open, pass,
 * span; *count = calls summed; reset != 0 clears them after reading): a caller that wants the mean span of K timed calls reads once
 * behind the loop instead of once per call (bench.py's timed region: one foreign call per step instead of two). */
int zk_timing_sums(double* sums_ms, uint64_t* count, int reset);
/* Host microseconds the calling thread's last one-shot zk_evm_verify spent inside its open / launch / collect / close calls
 * (tuning aid: where the wall time beyond the device span goes). */
int zk_last_host_phases(double* us4);

/* ---- Multi-GPU tally (§2 (vii) "RCCL all-reduce of {fail_count: SUM, first_fail_row: MIN}").  Rows shard across
 *      the GPUs of a node with no data-path collective; the only exchange is the pass / fail tally.  These entries put it behind
 *      the C ABI for hosts without torch.distributed: ONE RCCL collective per call — an all-gather of three 64-bit words per rank
 *      (fail count, first failing GLOBAL row, its status code) on the communicator's own stream, read back once; SUM and the
 *      lexicographic MIN are then taken on the host, identically on every rank (the same exchange the Python mirror makes through
 *      torch.distributed, zkevm_specs_amd/distributed.py reduce_tally).  librccl.so.1 is loaded on first use (dlopen): processes
 *      that never call these entries do not depend on it.
 *        zk_dist_unique_id  rank 0: a fresh communicator id (ncclGetUniqueId); the host carries it to the other ranks
 *        zk_dist_init       every rank, after zk_init(device): joins the communicator (collective: blocks until all ranks call)
 *        zk_dist_tally      collective.  local: this rank's zk_collect result; row_offset: global index of its row 0.
 *                           global->fail_count = SUM, first_fail_row / first_fail_code = those of the smallest failing global
 *                           row (UINT64_MAX / 0 if none), rows_evaluated = SUM, launches = local's, kernel_ms = MAX over ranks
 *        zk_dist_close      leaves the communicator
 *      The CPU backend implements them for world == 1 (the identity with the offset applied). */
#define ZK_DIST_ID_BYTES 128
typedef struct zk_comm zk_comm;
int zk_dist_unique_id(uint8_t* id /* ZK_DIST_ID_BYTES */);
int zk_dist_init(const uint8_t* id /* ZK_DIST_ID_BYTES */, int rank, int world, zk_comm** out);
int zk_dist_tally(zk_comm* c, const zk_result* local, uint64_t row_offset, zk_result* global);
int zk_dist_close(zk_comm* c);
/* Re-bind a session to another stream of its device (NULL = the engine's own); waits for its enqueued passes first. */
int zk_session_set_stream(zk_session* s, void* hip_stream);
int zk_close(zk_session* s);

#ifdef __cplusplus
}
#endif
#endif
