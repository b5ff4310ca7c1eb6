// Keccak-256 for the two address derivations the CREATE gadgets compute inside the reference
// (instruction.py:1338-1352: rlp.encode + eth_utils.keccak, third-party there; algorithm: Keccak-f[1600],
// rate 136, pre-NIST 0x01 padding).  Single-block messages only (<= 135 bytes): the RLP of
// [address, nonce] is at most 56 bytes, the CREATE2 preimage 85 bytes.
#pragma once
#include "fr.hpp"

ZK_HD u64 keccak_rol(u64 x, int n) { return n ? ((x << n) | (x >> (64 - n))) : x; }
ZK_HD void keccak_f1600(u64 a[25]) {  // a[x + 5 * y]
    const u64 RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull,
                        0x000000000000808Bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
                        0x000000000000008Aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000Aull,
                        0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull, 0x8000000000008003ull,
                        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    for (int round = 0; round < 24; round++) {
        u64 c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ keccak_rol(c[(x + 1) % 5], 1);
        for (int k = 0; k < 25; k++) a[k] ^= d[k % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rol(a[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}
// digest of msg[0..len), len <= 135
// out of line: three call sites, all on cold paths (keeps the compile time and the code size down)
ZK_NOINLINE void keccak256_block(const uint8_t* msg, int len, uint8_t out[32]) {
    u64 a[25];
    for (int k = 0; k < 25; k++) a[k] = 0;
    for (int k = 0; k < len; k++) a[k >> 3] ^= (u64)msg[k] << (8 * (k & 7));
    a[len >> 3] ^= 0x01ull << (8 * (len & 7));
    a[16] ^= 0x80ull << 56;  // last byte of the 136-byte rate
    keccak_f1600(a);
    for (int k = 0; k < 32; k++) out[k] = (uint8_t)(a[k >> 3] >> (8 * (k & 7)));
}
// int.from_bytes(digest[12:32], "big") as a field element (160 bits)
ZK_HD Fr keccak_digest_address(const uint8_t h[32]) {
    Fr r = fr_zero();
    for (int k = 0; k < 20; k++) r.v[k >> 2] |= (u32)h[31 - k] << (8 * (k & 3));
    return r;
}
// generate_contract_address (instruction.py:1338-1340): keccak(rlp([address as 20 BE bytes, nonce]))[-20:]
ZK_HD Fr keccak_create_address(const Fr& address /* < 2^160 */, const Fr& nonce /* any canonical field value */) {
    uint8_t m[64];
    int nlen = fr_byte_len(nonce);  // minimal big-endian length, 0 for nonce == 0
    int p = 0;
    const bool single = nlen == 1 && fr_byte(nonce, 0) < 0x80;
    const int item_len = single ? 1 : 1 + nlen;
    m[p++] = (uint8_t)(0xC0 + 21 + item_len);
    m[p++] = 0x94;
    for (int k = 19; k >= 0; k--) m[p++] = (uint8_t)fr_byte(address, k);
    if (!single) m[p++] = (uint8_t)(0x80 + nlen);
    for (int k = nlen - 1; k >= 0; k--) m[p++] = (uint8_t)fr_byte(nonce, k);
    uint8_t h[32];
    keccak256_block(m, p, h);
    return keccak_digest_address(h);
}
// generate_CREAET2_contract_address (:1342-1352): keccak(0xff + address BE + salt LE32 + code_hash LE32)[-20:]
ZK_HD Fr keccak_create2_address(const Fr& address, const Fr& salt /* 256-bit integer */, const Fr& code_hash /* 256-bit integer */) {
    uint8_t m[85];
    int p = 0;
    m[p++] = 0xff;
    for (int k = 19; k >= 0; k--) m[p++] = (uint8_t)fr_byte(address, k);
    for (int k = 0; k < 32; k++) m[p++] = (uint8_t)fr_byte(salt, k);
    for (int k = 0; k < 32; k++) m[p++] = (uint8_t)fr_byte(code_hash, k);
    uint8_t h[32];
    keccak256_block(m, p, h);
    return keccak_digest_address(h);
}
