"""ctypes binding of libzkevm_hip.so (C ABI: include/zkevm_hip.h).

There is deliberately NO fallback: if the HIP library is missing or no GPU is usable the
engine raises — the product path never routes through a CPU implementation on its own.

ZK_BACKEND=cpu (an explicit choice, read once when the library is first loaded) binds the same C ABI to libzkevm_cpu.so
instead: the kernels' own per-row device functions compiled for the host and run with OpenMP (csrc/cpu_backend.cpp) —
BASELINE configs[0]'s "pure CPU path" through the real boundary and the optimised-CPU line of bench.py.  It implements every
entry of include/zkevm_hip.h — the verify / open / launch / collect entries of every circuit, the keccak-table and ECDSA entries
and (round 4) the State / Bytecode / Copy witness assignments; ZK_OPT_DEVICE_PTRS has no meaning there (inputs are host arrays).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
BACKEND = os.environ.get("ZK_BACKEND", "hip")
if BACKEND not in ("hip", "cpu"):
    raise ImportError(f"ZK_BACKEND={BACKEND!r}: expected 'hip' (default) or 'cpu'")
CPU_LIB_PATH = os.path.join(_HERE, "libzkevm_cpu.so")
LIB_PATH = CPU_LIB_PATH if BACKEND == "cpu" else (os.environ.get("ZK_HIP_LIB") or os.path.join(_HERE, "libzkevm_hip.so"))  # ZK_HIP_LIB: tuning builds (tools/)

EXPORTED_SYMBOLS = [
    "zk_init", "zk_shutdown", "zk_set_stream", "zk_session_set_stream", "zk_last_error", "zk_fr_op",
    "zk_state_open", "zk_state_set_range", "zk_set_range", "zk_state_verify", "zk_evm_open", "zk_evm_verify", "zk_evm_verify_batch", "zk_bytecode_open", "zk_bytecode_verify", "zk_exp_open", "zk_exp_verify", "zk_copy_open", "zk_copy_verify", "zk_sign_open", "zk_sign_verify",
    "zk_keccak_open", "zk_keccak_read_rows", "zk_keccak_table", "zk_state_assign_open", "zk_state_assign_read", "zk_state_assign", "zk_state_ops_from_rw_open", "zk_state_ops_from_rw_read", "zk_state_ops_from_rw", "zk_state_assign_from_rw_open", "zk_state_verify_from_rw_open", "zk_state_verify_from_rw", "zk_block_verify", "zk_ecdsa_open", "zk_ecdsa_open_batches", "zk_ecdsa_verify", "zk_bytecode_assign_open", "zk_bytecode_assign_read", "zk_bytecode_assign", "zk_pi_open", "zk_pi_verify", "zk_pi_copy_open", "zk_pi_copy_verify", "zk_copy_assign_sizes", "zk_copy_assign_open", "zk_copy_assign_read", "zk_copy_assign", "zk_launch", "zk_collect", "zk_read_status", "zk_close", "zk_session_timing", "zk_last_timing", "zk_timing_sums", "zk_last_host_phases", "zk_dist_unique_id", "zk_dist_init", "zk_dist_tally", "zk_dist_close",
]

OPT_DEVICE_PTRS = 1


class ZkResult(ctypes.Structure):
    _fields_ = [
        ("fail_count", ctypes.c_uint64),
        ("first_fail_row", ctypes.c_uint64),
        ("first_fail_code", ctypes.c_uint32),
        ("launches", ctypes.c_uint32),
        ("rows_evaluated", ctypes.c_uint64),
        ("kernel_ms", ctypes.c_double),
    ]


class ZkEvmTables(ctypes.Structure):
    _fields_ = [
        ("steps", ctypes.c_void_p), ("n_steps", ctypes.c_uint64),
        ("rw", ctypes.c_void_p), ("rw_flags", ctypes.c_void_p), ("n_rw", ctypes.c_uint64),
        ("bytecode", ctypes.c_void_p), ("n_bytecode", ctypes.c_uint64),
        ("tx", ctypes.c_void_p), ("tx_flags", ctypes.c_void_p), ("n_tx", ctypes.c_uint64),
        ("block", ctypes.c_void_p), ("block_flags", ctypes.c_void_p), ("n_block", ctypes.c_uint64),
        ("begin_with_first_step", ctypes.c_uint32), ("end_with_last_step", ctypes.c_uint32),
        ("copy", ctypes.c_void_p), ("n_copy", ctypes.c_uint64),
        ("keccak", ctypes.c_void_p), ("n_keccak", ctypes.c_uint64),
        ("exp", ctypes.c_void_p), ("n_exp", ctypes.c_uint64),
        ("aux", ctypes.c_void_p), ("aux_kind", ctypes.c_void_p),
        ("withdrawals", ctypes.c_void_p), ("n_withdrawals", ctypes.c_uint64),
        ("sig", ctypes.c_void_p), ("n_sig", ctypes.c_uint64),
        ("ecc", ctypes.c_void_p), ("n_ecc", ctypes.c_uint64),
        ("aux_cells", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
    ]


class ZkCopyTables(ctypes.Structure):
    _fields_ = [
        ("rows", ctypes.c_void_p), ("row_flags", ctypes.c_void_p), ("n_rows", ctypes.c_uint64),
        ("randomness", ctypes.c_void_p),
        ("rw", ctypes.c_void_p), ("rw_flags", ctypes.c_void_p), ("n_rw", ctypes.c_uint64),
        ("bytecode", ctypes.c_void_p), ("n_bytecode", ctypes.c_uint64),
        ("tx", ctypes.c_void_p), ("tx_flags", ctypes.c_void_p), ("n_tx", ctypes.c_uint64),
    ]


class ZkSignUnits(ctypes.Structure):
    _fields_ = [
        ("bytes", ctypes.c_void_p), ("cells", ctypes.c_void_p), ("meta", ctypes.c_void_p), ("n_units", ctypes.c_uint64),
        ("randomness", ctypes.c_void_p),
        ("keccak", ctypes.c_void_p), ("n_keccak", ctypes.c_uint64),
        ("tx_rows", ctypes.c_void_p), ("tx_flags", ctypes.c_void_p), ("n_tx_rows", ctypes.c_uint64),
        ("is_sig", ctypes.c_uint32),
    ]


class ZkEcdsaBatch(ctypes.Structure):
    _fields_ = [("bytes", ctypes.c_void_p), ("layout", ctypes.c_uint32), ("v", ctypes.c_void_p), ("v_stride", ctypes.c_uint32),
                ("n", ctypes.c_uint64), ("out_dev", ctypes.c_void_p), ("out_stride", ctypes.c_uint32)]


class ZkCopyEvents(ctypes.Structure):
    _fields_ = [
        ("events", ctypes.c_void_p), ("flags", ctypes.c_void_p), ("n_events", ctypes.c_uint64),
        ("data", ctypes.c_void_p), ("data_offsets", ctypes.c_void_p),
        ("randomness", ctypes.c_void_p),
    ]


class ZkBlock(ctypes.Structure):
    """zk_block (include/zkevm_hip.h): the raw device-resident inputs of zk_block_verify"""
    _fields_ = [
        ("evm", ZkEvmTables),
        ("hashed_data", ctypes.c_void_p), ("hashed_bytes", ctypes.c_uint64), ("hashed_offsets", ctypes.c_void_p),
        ("n_codes", ctypes.c_uint64), ("n_hashed", ctypes.c_uint64),
        ("randomness", ctypes.c_void_p),
        ("code_offsets", ctypes.c_void_p), ("code_lengths", ctypes.c_void_p), ("n_bytecodes", ctypes.c_uint64), ("k", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
        ("copy_events", ZkCopyEvents),
        ("exp_rows", ctypes.c_void_p), ("n_exp_rows", ctypes.c_uint64),
        ("tx", ZkSignUnits),
    ]


BLOCK_CIRCUITS = ("evm", "state", "bytecode", "tx", "copy", "exp")  # order of zk_block_verify's results (ZK_BLOCK_*)
OPT_NO_STATE_SORT = 2
OPT_GENERIC_INDEX = 4
OPT_SINGLE_PASS = 8
OPT_SIDE_STREAM = 16
OPT_STATE_COMPACT = 32  # State rows without the limb / byte columns (include/zkevm_hip.h ZK_OPT_STATE_COMPACT)
OPT_BLOCK_STATE_ROWS = 64  # zk_block_verify: materialise the State witness instead of the fused form


class EngineError(RuntimeError):
    pass


_lib = None
_inited_device = None


def load():
    """Load the shared library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(zkevm_specs_amd/csrc/build.sh). There is no CPU fallback."
        )
    # torch ships its own libamdhip64.so.7; importing it first makes the dynamic loader bind
    # this library to the SAME HIP runtime instance (one runtime per process: device pointers,
    # streams and events are then interchangeable with torch's).
    if BACKEND == "hip":
        # one hardware queue per concurrent circuit session (csrc/zkevm_hip.hip zk_default_hw_queues): the HIP runtime reads the
        # variable at its first call, which torch makes lazily.  Set here — when the HIP library is actually loaded — and not
        # at package import: a process that only uses the CPU backend, or imports the package for its witness generators, keeps
        # its environment untouched.  A host that sets the variable itself keeps its own choice.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        import torch  # noqa: F401

    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


_cpu_lib = None


def load_cpu():
    """The CPU backend's library next to the default one (same C ABI; bench.py's optimised-CPU legs call it with
    `device="cpu"` on the one-shot entries).  Never a fallback: only explicit callers get it."""
    global _cpu_lib
    if _cpu_lib is None:
        if not os.path.exists(CPU_LIB_PATH):
            raise EngineError(f"{CPU_LIB_PATH} is missing: build it with zkevm_specs_amd/csrc/build.sh")
        _cpu_lib = _bind(ctypes.CDLL(CPU_LIB_PATH))
    return _cpu_lib


def set_cpu_threads(n):
    """OpenMP threads of the CPU backend's passes (libgomp's omp_set_num_threads)"""
    ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))


def _bind(lib):
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    lib.zk_init.argtypes = [ctypes.c_int]
    lib.zk_set_stream.argtypes = [vp]
    lib.zk_session_set_stream.argtypes = [vp, vp]
    lib.zk_last_error.restype = ctypes.c_char_p
    lib.zk_fr_op.argtypes = [ctypes.c_int, vp, vp, vp, u64, u32]
    lib.zk_state_open.argtypes = [vp, vp, u64, vp, u64, u32, ctypes.POINTER(vp)]
    lib.zk_state_set_range.argtypes = [vp, u64, u64]
    lib.zk_set_range.argtypes = [vp, u64, u64]
    lib.zk_state_verify.argtypes = [vp, vp, u64, vp, u64, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_evm_open.argtypes = [ctypes.POINTER(ZkEvmTables), u32, ctypes.POINTER(vp)]
    lib.zk_evm_verify.argtypes = [ctypes.POINTER(ZkEvmTables), u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_evm_verify_batch.argtypes = [ctypes.POINTER(ctypes.POINTER(ZkEvmTables)), u64, u32, ctypes.POINTER(ZkResult)]
    lib.zk_bytecode_open.argtypes = [vp, u64, vp, u64, vp, u32, ctypes.POINTER(vp)]
    lib.zk_bytecode_verify.argtypes = [vp, u64, vp, u64, vp, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_exp_open.argtypes = [vp, u64, u32, ctypes.POINTER(vp)]
    lib.zk_exp_verify.argtypes = [vp, u64, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_copy_open.argtypes = [ctypes.POINTER(ZkCopyTables), u32, ctypes.POINTER(vp)]
    lib.zk_copy_verify.argtypes = [ctypes.POINTER(ZkCopyTables), u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_sign_open.argtypes = [ctypes.POINTER(ZkSignUnits), u32, ctypes.POINTER(vp)]
    lib.zk_sign_verify.argtypes = [ctypes.POINTER(ZkSignUnits), u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_keccak_open.argtypes = [vp, u64, vp, u64, vp, u32, vp, u32, ctypes.POINTER(vp)]
    lib.zk_keccak_read_rows.argtypes = [vp, vp]
    lib.zk_keccak_table.argtypes = [vp, u64, vp, u64, vp, u32, vp, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_state_assign_open.argtypes = [vp, vp, u64, vp, vp, vp, u32, ctypes.POINTER(vp)]
    lib.zk_state_assign_read.argtypes = [vp, vp, vp, vp, u64, ctypes.POINTER(u64)]
    lib.zk_state_assign.argtypes = [vp, vp, u64, vp, vp, vp, ctypes.POINTER(u64), u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_state_ops_from_rw_open.argtypes = [vp, vp, u64, vp, vp, u32, ctypes.POINTER(u64), ctypes.POINTER(vp)]
    lib.zk_state_ops_from_rw_read.argtypes = [vp, vp, vp, ctypes.POINTER(u64)]
    lib.zk_state_assign_from_rw_open.argtypes = [vp, vp, u64, vp, vp, vp, u32, ctypes.POINTER(u64), ctypes.POINTER(vp)]
    lib.zk_state_verify_from_rw_open.argtypes = [vp, vp, u64, u32, ctypes.POINTER(u64), ctypes.POINTER(vp)]
    lib.zk_state_verify_from_rw.argtypes = [vp, vp, u64, u32, vp, ctypes.POINTER(u64), ctypes.POINTER(ZkResult)]
    lib.zk_block_verify.argtypes = [vp, u32, vp, vp]
    lib.zk_state_ops_from_rw.argtypes = [vp, vp, u64, vp, vp, ctypes.POINTER(u64), u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_ecdsa_open.argtypes = [vp, u32, vp, u32, u64, vp, u32, u32, ctypes.POINTER(vp)]
    lib.zk_ecdsa_verify.argtypes = [vp, u32, vp, u32, u64, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_ecdsa_open_batches.argtypes = [ctypes.POINTER(ZkEcdsaBatch), u32, u32, ctypes.POINTER(vp)]
    lib.zk_bytecode_assign_open.argtypes = [vp, u64, vp, vp, u64, u32, vp, vp, u32, ctypes.POINTER(vp)]
    lib.zk_bytecode_assign_read.argtypes = [vp, vp]
    lib.zk_bytecode_assign.argtypes = [vp, u64, vp, vp, u64, u32, vp, vp, u32, ctypes.POINTER(ZkResult)]
    lib.zk_pi_open.argtypes = [vp, u64, vp, u64, vp, u64, u64, vp, vp, u32, ctypes.POINTER(vp)]
    lib.zk_pi_verify.argtypes = [vp, u64, vp, u64, vp, u64, u64, vp, vp, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_pi_copy_open.argtypes = [vp, vp, vp, u64, u32, ctypes.POINTER(vp)]
    lib.zk_pi_copy_verify.argtypes = [vp, vp, vp, u64, u32, vp, ctypes.POINTER(ZkResult)]
    lib.zk_copy_assign_sizes.argtypes = [ctypes.POINTER(ZkCopyEvents), u32, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64)]
    lib.zk_copy_assign_open.argtypes = [ctypes.POINTER(ZkCopyEvents), vp, vp, vp, vp, vp, u32, ctypes.POINTER(vp)]
    lib.zk_copy_assign_read.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.zk_copy_assign.argtypes = [ctypes.POINTER(ZkCopyEvents), vp, vp, vp, vp, vp, u32, ctypes.POINTER(ZkResult)]
    lib.zk_launch.argtypes = [vp, vp]
    lib.zk_collect.argtypes = [vp, ctypes.POINTER(ZkResult)]
    lib.zk_read_status.argtypes = [vp, vp]
    lib.zk_close.argtypes = [vp]
    dp = ctypes.POINTER(ctypes.c_double)
    lib.zk_session_timing.argtypes = [vp, dp, dp]
    lib.zk_last_timing.argtypes = [dp, dp, dp]
    lib.zk_last_host_phases.argtypes = [dp]
    lib.zk_timing_sums.argtypes = [dp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    lib.zk_dist_unique_id.argtypes = [ctypes.c_void_p]
    lib.zk_dist_init.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.zk_dist_tally.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p]
    lib.zk_dist_close.argtypes = [ctypes.c_void_p]
    return lib


def check(rc, what, lib=None):
    """`lib`: the library the failing call was made on (a Session passes its own: an error of the CPU backend must not make
    this process load torch + the HIP library just to format it)."""
    if rc != 0:
        owner = lib if lib is not None else (_lib if _lib is not None else (_cpu_lib if _cpu_lib is not None else load()))
        msg = owner.zk_last_error().decode(errors="replace")
        if not msg and lib is None and _cpu_lib is not None and owner is not _cpu_lib:
            msg = _cpu_lib.zk_last_error().decode(errors="replace")
        raise EngineError(f"{what} failed (rc={rc}): {msg}")


def init(device=None):
    """Initialise the engine on a HIP device (default: the device an earlier init() of this process selected, else
    LOCAL_RANK, else 0)."""
    global _inited_device
    if device == "cpu":  # the CPU backend's library, explicitly (see load_cpu)
        return load_cpu()
    lib = load()
    if device is None:
        device = _inited_device if _inited_device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    # zk_init is idempotent and its selection is per thread: always forward it (a cached "already initialised" flag would
    # be wrong for a second thread or after another device was selected in between)
    check(lib.zk_init(int(device)), "zk_init")
    _inited_device = device
    return lib


def ptr(x):
    """void* of a numpy array (host) or a torch tensor (device / host)."""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return ctypes.c_void_p(x.data_ptr())
    return ctypes.c_void_p(x.ctypes.data)
