"""Static EVM-circuit tables (enum numberings, opcode metadata) + generator of csrc/evm_tables.h.

These are *data* the device and the oracle need: the numbering of the reference's enums
(evm_circuit/execution_state.py:14-138, table.py:128-353, opcode.py:8-152) and the opcode
metadata of OPCODE_INFO_MAP (opcode.py:213-362).  oracle/gen_golden_evm.py asserts every entry
against the imported reference, so a drift in either direction fails golden generation.

Run `python -m zkevm_specs_amd.evm_tables > zkevm_specs_amd/csrc/evm_tables.h` to regenerate.
"""
from enum import IntEnum

_STATE_NAMES = """BeginTx EndTx EndBlock STOP ADD MUL SDIV_SMOD ADDMOD MULMOD EXP SIGNEXTEND CMP SCMP ISZERO
BITWISE NOT BYTE SHL_SHR SAR SHA3 ADDRESS BALANCE ORIGIN CALLER CALLVALUE CALLDATALOAD CALLDATASIZE
CALLDATACOPY CODESIZE CODECOPY GASPRICE EXTCODESIZE EXTCODECOPY RETURNDATASIZE RETURNDATACOPY EXTCODEHASH
BLOCKHASH BlockCtx SELFBALANCE POP MEMORY SLOAD SSTORE JUMP JUMPI PC MSIZE GAS JUMPDEST PUSH DUP SWAP LOG
CREATE CALL_OP RETURN CREATE2 REVERT SELFDESTRUCT ErrorInvalidOpcode ErrorGasUintOverflow ErrorStack
ErrorWriteProtection ErrorDepth ErrorInsufficientBalance ErrorContractAddressCollision
ErrorInvalidCreationCode ErrorNonceUintOverflow ErrorMaxCodeSizeExceeded ErrorInvalidJump
ErrorReturnDataOutOfBound ErrorOutOfGasConstant ErrorOutOfGasStaticMemoryExpansion
ErrorOutOfGasDynamicMemoryExpansion ErrorOutOfGasMemoryCopy ErrorOutOfGasAccountAccess
ErrorOutOfGasCodeStore ErrorOutOfGasLOG ErrorOutOfGasEXP ErrorOutOfGasSHA3 ErrorOutOfGasSloadSstore
ErrorOutOfGasCall ErrorOutOfGasCREATE ErrorOutOfGasSELFDESTRUCT ErrorOutOfGasPrecompile ECRECOVER SHA256
RIPEMD160 DATACOPY BIGMODEXP BN254_ADD BN254_SCALAR_MUL BN254_PAIRING BLAKE2F""".split()

ExecutionState = IntEnum("ExecutionState", _STATE_NAMES, start=1)

# States the reference dispatches in EXECUTION_STATE_IMPL (execution/__init__.py:86-171); every
# other state makes verify_step raise NotImplementedError (main.py:63).
REFERENCE_UNIMPLEMENTED = [
    "PC", "JUMPDEST", "DUP", "SWAP", "REVERT", "SELFDESTRUCT", "ErrorDepth", "ErrorInsufficientBalance",
    "ErrorContractAddressCollision", "ErrorNonceUintOverflow", "ErrorOutOfGasSELFDESTRUCT", "SHA256",
    "RIPEMD160", "BIGMODEXP", "BLAKE2F",
]

# name: (value, constant_gas, has_dynamic_gas)
_G0, _G1, _GQ, _GF3, _GF5, _GM, _GS, _GE = 0, 1, 2, 3, 5, 8, 10, 20
OPCODES = {
    "STOP": (0x00, _G0, 0), "ADD": (0x01, _GF3, 0), "MUL": (0x02, _GF5, 0), "SUB": (0x03, _GF3, 0),
    "DIV": (0x04, _GF5, 0), "SDIV": (0x05, _GF5, 0), "MOD": (0x06, _GF5, 0), "SMOD": (0x07, _GF5, 0),
    "ADDMOD": (0x08, _GM, 0), "MULMOD": (0x09, _GM, 0), "EXP": (0x0A, _G0, 1), "SIGNEXTEND": (0x0B, _GF5, 0),
    "LT": (0x10, _GF3, 0), "GT": (0x11, _GF3, 0), "SLT": (0x12, _GF3, 0), "SGT": (0x13, _GF3, 0),
    "EQ": (0x14, _GF3, 0), "ISZERO": (0x15, _GF3, 0), "AND": (0x16, _GF3, 0), "OR": (0x17, _GF3, 0),
    "XOR": (0x18, _GF3, 0), "NOT": (0x19, _GF3, 0), "BYTE": (0x1A, _GF3, 0), "SHL": (0x1B, _GF3, 0),
    "SHR": (0x1C, _GF3, 0), "SAR": (0x1D, _GF3, 0), "SHA3": (0x20, 30, 1), "ADDRESS": (0x30, _GQ, 0),
    "BALANCE": (0x31, 100, 1), "ORIGIN": (0x32, _GQ, 0), "CALLER": (0x33, _GQ, 0), "CALLVALUE": (0x34, _GQ, 0),
    "CALLDATALOAD": (0x35, _GF3, 0), "CALLDATASIZE": (0x36, _GQ, 0), "CALLDATACOPY": (0x37, _GF3, 1),
    "CODESIZE": (0x38, _GQ, 0), "CODECOPY": (0x39, _GF3, 1), "GASPRICE": (0x3A, _GQ, 0),
    "EXTCODESIZE": (0x3B, 100, 1), "EXTCODECOPY": (0x3C, 100, 1), "RETURNDATASIZE": (0x3D, _GQ, 0),
    "RETURNDATACOPY": (0x3E, _GF3, 1), "EXTCODEHASH": (0x3F, 100, 1), "BLOCKHASH": (0x40, _GE, 0),
    "COINBASE": (0x41, _GQ, 0), "TIMESTAMP": (0x42, _GQ, 0), "NUMBER": (0x43, _GQ, 0),
    "PREVRANDAO": (0x44, _GQ, 0), "GASLIMIT": (0x45, _GQ, 0), "CHAINID": (0x46, _GQ, 0),
    "SELFBALANCE": (0x47, _GF5, 0), "BASEFEE": (0x48, _GQ, 0), "POP": (0x50, _GQ, 0), "MLOAD": (0x51, _GF3, 1),
    "MSTORE": (0x52, _GF3, 1), "MSTORE8": (0x53, _GF3, 1), "SLOAD": (0x54, _G0, 1), "SSTORE": (0x55, _G0, 1),
    "JUMP": (0x56, _GM, 0), "JUMPI": (0x57, _GS, 0), "PC": (0x58, _GQ, 0), "MSIZE": (0x59, _GQ, 0),
    "GAS": (0x5A, _GQ, 0), "JUMPDEST": (0x5B, _G1, 0), "PUSH0": (0x5F, _GQ, 0),
    "LOG0": (0xA0, 0, 1), "LOG1": (0xA1, 0, 1), "LOG2": (0xA2, 0, 1), "LOG3": (0xA3, 0, 1), "LOG4": (0xA4, 0, 1),
    "CREATE": (0xF0, 32000, 1), "CALL": (0xF1, 100, 1), "CALLCODE": (0xF2, 100, 1), "RETURN": (0xF3, 0, 1),
    "DELEGATECALL": (0xF4, 100, 1), "CREATE2": (0xF5, 32000, 1), "STATICCALL": (0xFA, 100, 1),
    "REVERT": (0xFD, 0, 1), "SELFDESTRUCT": (0xFF, 5000, 1),
}
for _i in range(1, 33):
    OPCODES[f"PUSH{_i}"] = (0x5F + _i, _GF3, 0)
for _i in range(1, 17):
    OPCODES[f"DUP{_i}"] = (0x7F + _i, _GF3, 0)
    OPCODES[f"SWAP{_i}"] = (0x8F + _i, _GF3, 0)

Opcode = IntEnum("Opcode", {k: v[0] for k, v in OPCODES.items()})

# (min_stack_pointer, max_stack_pointer) of OPCODE_INFO_MAP (opcode.py:213-362), grouped
_STACK_BOUNDS = {
    (-6, 1017): "CALL CALLCODE", (-6, 1018): "LOG4", (-5, 1018): "DELEGATECALL STATICCALL", (-5, 1019): "LOG3",
    (-4, 1020): "EXTCODECOPY LOG2", (-3, 1020): "CREATE2", (-3, 1021): "CALLDATACOPY CODECOPY RETURNDATACOPY LOG1",
    (-2, 1021): "ADDMOD MULMOD CREATE", (-2, 1022): "MSTORE MSTORE8 SSTORE JUMPI LOG0 RETURN REVERT",
    (-1, 1022): "ADD MUL SUB DIV SDIV MOD SMOD EXP SIGNEXTEND LT GT SLT SGT EQ AND OR XOR BYTE SHL SHR SAR SHA3",
    (-1, 1023): "POP JUMP SELFDESTRUCT",
    (0, 1023): "ISZERO NOT BALANCE CALLDATALOAD EXTCODESIZE EXTCODEHASH BLOCKHASH MLOAD SLOAD",
    (0, 1024): "STOP JUMPDEST",
    (1, 1024): "ADDRESS ORIGIN CALLER CALLVALUE CALLDATASIZE CODESIZE GASPRICE RETURNDATASIZE COINBASE TIMESTAMP "
               "NUMBER PREVRANDAO GASLIMIT CHAINID SELFBALANCE BASEFEE PC MSIZE GAS PUSH0",
}
STACK_BOUNDS = {name: b for b, names in _STACK_BOUNDS.items() for name in names.split()}
for _i in range(1, 33):
    STACK_BOUNDS[f"PUSH{_i}"] = (1, 1024)
for _i in range(1, 17):
    STACK_BOUNDS[f"DUP{_i}"] = (1, 1024 - _i)
    STACK_BOUNDS[f"SWAP{_i}"] = (0, 1023 - _i)
assert set(STACK_BOUNDS) == set(OPCODES)


# success-case state -> responsible opcodes (execution_state.py:143-362); aux is 0 for all of them
RESPONSIBLE = {
    "STOP": ["STOP"], "ADD": ["ADD", "SUB"], "MUL": ["MUL", "DIV", "MOD"], "SDIV_SMOD": ["SDIV", "SMOD"],
    "ADDMOD": ["ADDMOD"], "MULMOD": ["MULMOD"], "EXP": ["EXP"], "SIGNEXTEND": ["SIGNEXTEND"],
    "CMP": ["LT", "GT", "EQ"], "SCMP": ["SLT", "SGT"], "ISZERO": ["ISZERO"], "BITWISE": ["AND", "OR", "XOR"],
    "NOT": ["NOT"], "BYTE": ["BYTE"], "SHL_SHR": ["SHL", "SHR"], "SAR": ["SAR"], "SHA3": ["SHA3"],
    "ADDRESS": ["ADDRESS"], "BALANCE": ["BALANCE"], "ORIGIN": ["ORIGIN"], "CALLER": ["CALLER"],
    "CALLVALUE": ["CALLVALUE"], "CALLDATALOAD": ["CALLDATALOAD"], "CALLDATASIZE": ["CALLDATASIZE"],
    "CALLDATACOPY": ["CALLDATACOPY"], "CODESIZE": ["CODESIZE"], "CODECOPY": ["CODECOPY"],
    "GASPRICE": ["GASPRICE"], "EXTCODESIZE": ["EXTCODESIZE"], "EXTCODECOPY": ["EXTCODECOPY"],
    "RETURNDATASIZE": ["RETURNDATASIZE"], "RETURNDATACOPY": ["RETURNDATACOPY"], "EXTCODEHASH": ["EXTCODEHASH"],
    "BLOCKHASH": ["BLOCKHASH"],
    "BlockCtx": ["COINBASE", "TIMESTAMP", "NUMBER", "PREVRANDAO", "GASLIMIT", "BASEFEE", "CHAINID"],
    "SELFBALANCE": ["SELFBALANCE"], "POP": ["POP"], "MEMORY": ["MLOAD", "MSTORE", "MSTORE8"],
    "SLOAD": ["SLOAD"], "SSTORE": ["SSTORE"], "JUMP": ["JUMP"], "JUMPI": ["JUMPI"], "PC": ["PC"],
    "MSIZE": ["MSIZE"], "GAS": ["GAS"], "JUMPDEST": ["JUMPDEST"],
    "PUSH": ["PUSH0"] + [f"PUSH{i}" for i in range(1, 33)], "DUP": [f"DUP{i}" for i in range(1, 17)],
    "SWAP": [f"SWAP{i}" for i in range(1, 17)], "LOG": [f"LOG{i}" for i in range(5)], "CREATE": ["CREATE"],
    "CALL_OP": ["CALL", "CALLCODE", "DELEGATECALL", "STATICCALL"], "RETURN": ["RETURN"], "CREATE2": ["CREATE2"],
    "REVERT": ["REVERT"], "SELFDESTRUCT": ["SELFDESTRUCT"],
}

HALTS_IN_SUCCESS = ["STOP", "RETURN", "SELFDESTRUCT"]
HALTS_IN_EXCEPTION = [
    "ErrorInvalidOpcode", "ErrorGasUintOverflow", "ErrorStack", "ErrorWriteProtection", "ErrorDepth",
    "ErrorInsufficientBalance", "ErrorContractAddressCollision", "ErrorInvalidCreationCode",
    "ErrorMaxCodeSizeExceeded", "ErrorInvalidJump", "ErrorReturnDataOutOfBound", "ErrorOutOfGasConstant",
    "ErrorOutOfGasStaticMemoryExpansion", "ErrorOutOfGasDynamicMemoryExpansion", "ErrorOutOfGasMemoryCopy",
    "ErrorOutOfGasAccountAccess", "ErrorOutOfGasCodeStore", "ErrorOutOfGasLOG", "ErrorOutOfGasEXP",
    "ErrorOutOfGasSHA3", "ErrorOutOfGasSloadSstore", "ErrorOutOfGasCall", "ErrorOutOfGasCREATE",
    "ErrorOutOfGasSELFDESTRUCT",
]


def halts(state):
    n = ExecutionState(state).name
    return n in HALTS_IN_SUCCESS or n in HALTS_IN_EXCEPTION or n == "REVERT"


# table.py tag enums (auto() from 1 unless stated)
Target = IntEnum("Target", "Start TxAccessListAccount TxAccessListAccountStorage TxRefund Account AccountStorage CallContext Stack Memory TxLog TxReceipt", start=1)
CallContextFieldTag = IntEnum(
    "CallContextFieldTag",
    "RwCounterEndOfReversion CallerId TxId Depth CallerAddress CalleeAddress CallDataOffset CallDataLength "
    "ReturnDataOffset ReturnDataLength Value IsSuccess IsPersistent IsStatic IsRoot IsCreate CodeHash "
    "LastCalleeId LastCalleeReturnDataOffset LastCalleeReturnDataLength ProgramCounter StackPointer GasLeft "
    "MemorySize ReversibleWriteCounter", start=1)
AccountFieldTag = IntEnum("AccountFieldTag", "Nonce Balance CodeHash NonExisting", start=1)
TxContextFieldTag = IntEnum(
    "TxContextFieldTag",
    "Nonce Gas GasPrice CallerAddress CalleeAddress IsCreate Value CallDataLength CallDataGasCost TxInvalid "
    "AccessListGasCost TxSignHash CallData", start=1)
BlockContextFieldTag = IntEnum(
    "BlockContextFieldTag", "Coinbase GasLimit Number Timestamp PrevRandao BaseFee ChainId HistoryHash WithdrawalRoot", start=1)
BytecodeFieldTag = IntEnum("BytecodeFieldTag", {"Header": 1, "Byte": 2})
FixedTableTag = IntEnum(
    "FixedTableTag",
    "Range5 Range16 Range32 Range64 Range256 Range512 Range1024 Range24_576 SignByte BitwiseAnd BitwiseOr "
    "BitwiseXor ResponsibleOpcode Pow2 OpcodeConstantGas PrecompileInfo", start=1)
RW = IntEnum("RW", {"Read": 0, "Write": 1})


def gen_header():
    out = ["// GENERATED by zkevm_specs_amd/evm_tables.py - do not edit.", "#pragma once", "#include <stdint.h>", ""]
    out.append("enum ZkExecState : uint32_t {")
    for s in ExecutionState:
        out.append(f"    ES_{s.name} = {int(s)},")
    out.append(f"    ES_COUNT = {len(ExecutionState) + 1}")
    out.append("};")
    out.append("enum ZkOpcode : uint32_t {")
    for name, (v, _, _) in sorted(OPCODES.items(), key=lambda kv: kv[1][0]):
        out.append(f"    OP_{name} = 0x{v:02x},")
    out.append("};")
    valid = [0] * 256
    gas = [0] * 256
    resp = [0] * 256
    for name, (v, g, _) in OPCODES.items():
        valid[v] = 1
        gas[v] = g
    for st, ops in RESPONSIBLE.items():
        for o in ops:
            resp[OPCODES[o][0]] = int(ExecutionState[st])

    def arr(name, ctype, vals):
        out.append(f"#define {name}_INIT {{ {', '.join(str(x) for x in vals)} }}")

    arr("ZK_OPCODE_VALID", "uint8_t", valid)
    arr("ZK_OPCODE_CONST_GAS", "uint16_t", gas)
    arr("ZK_OPCODE_RESP_STATE", "uint8_t", resp)
    dyn, mn, mx = [0] * 256, [0] * 256, [0] * 256
    for name, (v, g, d) in OPCODES.items():
        dyn[v] = d
        mn[v], mx[v] = STACK_BOUNDS[name]
    arr("ZK_OPCODE_DYNAMIC_GAS", "uint8_t", dyn)
    arr("ZK_OPCODE_MIN_SP", "int16_t", mn)
    arr("ZK_OPCODE_MAX_SP", "int16_t", mx)
    impl = [0] * (len(ExecutionState) + 1)
    for s in ExecutionState:
        impl[int(s)] = 0 if s.name in REFERENCE_UNIMPLEMENTED else 1
    arr("ZK_STATE_REF_IMPLEMENTED", "uint8_t", impl)
    hl = [0] * (len(ExecutionState) + 1)
    for s in ExecutionState:
        hl[int(s)] = 1 if halts(s) else 0
    arr("ZK_STATE_HALTS", "uint8_t", hl)
    # the same tables as immediates / one packed word per opcode: no dependent table loads on the hot path
    def mask64(bits, lo):
        return sum(1 << (k - lo) for k in range(lo, lo + 64) if k < len(bits) and bits[k])
    out.append(f"#define ZK_STATE_REF_IMPL_MASK_LO 0x{mask64(impl, 0):016x}ull")
    out.append(f"#define ZK_STATE_REF_IMPL_MASK_HI 0x{mask64(impl, 64):016x}ull")
    out.append(f"#define ZK_STATE_HALTS_MASK_LO 0x{mask64(hl, 0):016x}ull")
    out.append(f"#define ZK_STATE_HALTS_MASK_HI 0x{mask64(hl, 64):016x}ull")
    assert len(impl) <= 128
    # opinfo[opcode] = responsible state | valid << 8 | constant gas << 16
    arr("ZK_OPINFO", "uint32_t", [resp[v] | (valid[v] << 8) | (gas[v] << 16) for v in range(256)])
    for enum, prefix in [(Target, "TG"), (CallContextFieldTag, "CC"), (AccountFieldTag, "ACC"),
                         (TxContextFieldTag, "TXC"), (BlockContextFieldTag, "BLK"), (FixedTableTag, "FX")]:
        out.append(f"enum Zk{enum.__name__} : uint32_t {{")
        for e in enum:
            out.append(f"    {prefix}_{e.name} = {int(e)},")
        out.append("};")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    print(gen_header(), end="")
