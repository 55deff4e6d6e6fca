// telomeri_amd/csrc/paf_swar.h -- the arithmetic of the device tokeniser (csrc/paf_parse.hip) that does not need a GPU to be
// checked: byte classification four at a time (SWAR), the strict integer of the host tokeniser read out of a word array,
// the word-wise name hash.  Compiled for the device by hipcc and for the host by g++ (tests/test_paf_swar.py runs every
// function against a byte-at-a-time restatement on the CPU tier); on the host the funnel shift is spelled out.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define TLM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TLM_HD inline
#endif

namespace tlm_swar {

// the low 32 bits of (hi:lo) >> shift, shift in 0..31
TLM_HD uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t shift) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, shift);
#else
    return (uint32_t) ((((uint64_t) hi << 32) | lo) >> (shift & 31u));
#endif
}

// 0x80 in every byte of the result whose byte of `w` is a newline
TLM_HD uint32_t newline_bytes(uint32_t w) {
    w ^= 0x0A0A0A0Au;
    return ~(((w & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | w | 0x7F7F7F7Fu);
}
// 0x80 in every byte of the result whose byte of `w` is whitespace: blank, \t \n \v \f \r (what operator>> skips)
TLM_HD uint32_t ws_bytes(uint32_t w) {
    const uint32_t lo7 = w & 0x7F7F7F7Fu;
    const uint32_t ge9 = (lo7 + 0x77777777u) | w, ge14 = (lo7 + 0x72727272u) | w;       // bit 7: byte >= 9, byte >= 14
    const uint32_t y = w ^ 0x20202020u;
    const uint32_t blank = ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y);                      // bit 7: byte == ' '
    return ((ge9 & ~ge14) | blank) & 0x80808080u;
}
// 0x80 in every byte at which a token starts: not whitespace, behind a byte that is (`zw` = ws_bytes of the word, `carry` = 0x80 if
// the byte in front of the word is whitespace, else 0; the next word's carry is zw >> 24).  Two per word at most.
TLM_HD uint32_t token_start_bytes(uint32_t zw, uint32_t carry) { return ~zw & ((zw << 8) | carry) & 0x80808080u; }
// 0x80 in every byte that is NOT the digit value 0..9 (x = bytes ^ '0')
TLM_HD uint32_t nondigit_bytes(uint32_t x) { return (((x & 0x7F7F7F7Fu) + 0x76767676u) | x) & 0x80808080u; }
// index of the lowest flagged byte (z != 0, flags are bit 7 of a byte)
TLM_HD uint32_t first_byte(uint32_t z) { return (uint32_t) __builtin_ctz(z) >> 3; }
TLM_HD bool ws_char(uint32_t c) { return c == ' ' || c - 9u < 5u; }
// the low n (0..4 and more) bytes of a word
TLM_HD uint32_t low_bytes(uint32_t w, uint32_t n) { return n >= 4u ? w : w & ((1u << (8u * n)) - 1u); }

// the four bytes at byte position p of a word array (any alignment)
TLM_HD uint32_t window_word(const uint32_t *w, uint32_t p) { return funnel(w[(p >> 2) + 1], w[p >> 2], (p & 3u) * 8u); }

// names are hashed a word at a time (the last one masked to the name's bytes): the same value whatever the alignment, and
// whichever of the two readers (words, bytes) met the name
TLM_HD uint32_t mix_word(uint32_t h, uint32_t w) {
    w *= 0xCC9E2D51u;
    w = (w << 15) | (w >> 17);
    w *= 0x1B873593u;
    h ^= w;
    h = (h << 13) | (h >> 19);
    return h * 5u + 0xE6546B64u;
}
TLM_HD uint32_t mix_end(uint32_t h, uint32_t len) {
    h ^= len;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    return h ^ (h >> 16);
}
constexpr uint32_t NAME_SEED = 0x9747B28Cu;

// the hash of a name read byte by byte (`rd(p)` = the byte at p): the words mix_word sees are those of the word reader
template <class Read>
TLM_HD uint32_t hash_bytes(const Read &rd, uint64_t at, uint32_t len) {
    uint32_t h = NAME_SEED;
    for (uint32_t i = 0; i < len; i += 4) {
        uint32_t w = 0;
        for (uint32_t k = 0; k < 4 && i + k < len; k++) w |= (uint32_t) (unsigned char) rd(at + i + k) << (8u * k);
        h = mix_word(h, w);
    }
    return mix_end(h, len);
}
// ... and out of a word array: the name at byte position pos ends at the first whitespace byte (there is one).  Also its
// first 16 bytes as four words, zero behind its end (the dictionary compares short names a word at a time).
struct NameScan {
    uint32_t len, hash, w0, w1, w2, w3;
};
TLM_HD NameScan scan_name(const uint32_t *sw, uint32_t pos) {
    NameScan n{0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t h = NAME_SEED;
    for (uint32_t k = 0;; k++) {
        const uint32_t w = window_word(sw, pos + 4u * k);
        const uint32_t z = ws_bytes(w);
        const uint32_t rest = z ? first_byte(z) : 4u;
        const uint32_t part = low_bytes(w, rest);
        if (k == 0u) n.w0 = part;
        else if (k == 1u) n.w1 = part;
        else if (k == 2u) n.w2 = part;
        else if (k == 3u) n.w3 = part;
        if (rest) h = mix_word(h, part);
        n.len = 4u * k + rest;
        if (z) break;
    }
    n.hash = mix_end(h, n.len);
    return n;
}

// the value of `c` (0..4) digit bytes standing in the low bytes of x (first digit lowest), x = text ^ '0'
TLM_HD uint32_t digit_group(uint32_t x, uint32_t c) {
    const uint32_t v = c ? x << (8u * (4u - c)) : 0u;                  // right-aligned in the word, leading zeros below
    const uint32_t pairs = (v * 10u + (v >> 8)) & 0x00FF00FFu;         // d0 d1 | d2 d3 -> two values below 100
    return (pairs * 100u + (pairs >> 16)) & 0xFFFFu;
}
TLM_HD uint32_t pow10_upto4(uint32_t c) { return c == 0 ? 1u : c == 1 ? 10u : c == 2 ? 100u : c == 3 ? 1000u : 10000u; }

// The token at byte position `pos` of a word array as the strict integer of the host tokeniser: [+-] then 1..10 digits,
// then whitespace, inside the int32 range.  Reads the words pos / 4 .. pos / 4 + 4.
TLM_HD bool int_token(const uint32_t *sw, uint32_t pos, int32_t &value) {
    const uint32_t wi = pos >> 2, sh = (pos & 3u) * 8u;
    const uint32_t a0 = sw[wi], a1 = sw[wi + 1], a2 = sw[wi + 2], a3 = sw[wi + 3], a4 = sw[wi + 4];
    uint32_t w0 = funnel(a1, a0, sh), w1 = funnel(a2, a1, sh), w2 = funnel(a3, a2, sh);
    const uint32_t w3 = funnel(a4, a3, sh);
    const uint32_t c0 = w0 & 0xFFu;
    const bool neg = c0 == '-';
    if (neg || c0 == '+') {                 // the digits start a byte later
        w0 = funnel(w1, w0, 8u);
        w1 = funnel(w2, w1, 8u);
        w2 = funnel(w3, w2, 8u);
    }
    const uint32_t x0 = w0 ^ 0x30303030u, x1 = w1 ^ 0x30303030u, x2 = w2 ^ 0x30303030u;
    const uint32_t n0 = nondigit_bytes(x0), n1 = nondigit_bytes(x1), n2 = nondigit_bytes(x2);
    // m = number of digits = index of the first byte that is none (12: twelve digits at least -- no integer)
    const uint32_t m = n0 ? first_byte(n0) : n1 ? 4u + first_byte(n1) : n2 ? 8u + first_byte(n2) : 12u;
    const uint32_t wt = m < 4u ? w0 : m < 8u ? w1 : w2;
    const uint32_t term = (wt >> ((m & 3u) * 8u)) & 0xFFu;
    const bool shaped = m >= 1u && m <= 10u && ws_char(term);
    const uint32_t c_0 = m < 4u ? m : 4u, r_1 = m - c_0, c_1 = r_1 < 4u ? r_1 : 4u, r_2 = r_1 - c_1, c_2 = r_2 < 4u ? r_2 : 4u;
    const uint32_t g0 = digit_group(x0, c_0), g1 = digit_group(x1, c_1), g2 = digit_group(x2, c_2);
    const unsigned long long mag = (unsigned long long) (g0 * pow10_upto4(c_1) + g1) * pow10_upto4(c_2) + g2;
    value = (int32_t) (neg ? 0u - (uint32_t) mag : (uint32_t) mag);
    return shaped && mag <= (neg ? 2147483648ull : 2147483647ull);
}

}   // namespace tlm_swar
