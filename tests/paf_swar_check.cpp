// tests/paf_swar_check.cpp -- test infrastructure: the arithmetic of the device tokeniser (telomeri_amd/csrc/paf_swar.h)
// compiled for the host and compared with byte-at-a-time restatements of the same rules.  Prints "ok <checks>" or the
// first disagreement.  Built and run by tests/test_paf_swar.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "paf_swar.h"

using namespace tlm_swar;

static unsigned long long checks = 0;
#define CHECK(cond, ...)                                                          \
    do {                                                                          \
        checks++;                                                                 \
        if (!(cond)) {                                                            \
            std::printf("FAILED %s:%d: ", __FILE__, __LINE__);                    \
            std::printf(__VA_ARGS__);                                             \
            std::printf("\n");                                                    \
            std::exit(1);                                                         \
        }                                                                         \
    } while (0)

static bool is_ws(unsigned c) { return c == ' ' || c == '\t' || c == '\n' || c == '\v' || c == '\f' || c == '\r'; }

// the strict integer of the host tokeniser (telomeri_amd/host/overlap_graph.cpp): [+-], 1..10 digits, then whitespace
static bool scalar_int(const unsigned char *p, long long &out) {
    bool neg = false;
    if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
    unsigned long long acc = 0;
    int nd = 0;
    while (*p >= '0' && *p <= '9') { acc = acc * 10 + (*p - '0'); nd++; p++; if (nd > 11) break; }
    if (nd < 1 || nd > 10 || !is_ws(*p)) return false;
    const long long v = neg ? -(long long) acc : (long long) acc;
    if (v > 2147483647ll || v < -2147483648ll) return false;
    out = v;
    return true;
}

int main() {
    std::mt19937 rng(12345);
    // ---- byte classes: every value in every position, among random neighbours
    for (unsigned pos = 0; pos < 4; pos++)
        for (unsigned v = 0; v < 256; v++)
            for (int rep = 0; rep < 64; rep++) {
                uint32_t w = rng();
                w = (w & ~(0xFFu << (8 * pos))) | (v << (8 * pos));
                const uint32_t zw = ws_bytes(w), zn = newline_bytes(w), zd = nondigit_bytes(w ^ 0x30303030u);
                for (unsigned k = 0; k < 4; k++) {
                    const unsigned b = (w >> (8 * k)) & 0xFF;
                    CHECK(((zw >> (8 * k)) & 0xFF) == (is_ws(b) ? 0x80u : 0u), "ws_bytes(%08x) byte %u", w, k);
                    CHECK(((zn >> (8 * k)) & 0xFF) == (b == '\n' ? 0x80u : 0u), "newline_bytes(%08x) byte %u", w, k);
                    CHECK(((zd >> (8 * k)) & 0xFF) == ((b >= '0' && b <= '9') ? 0u : 0x80u), "nondigit_bytes(%08x) byte %u", w, k);
                    CHECK(ws_char(b) == is_ws(b), "ws_char(%u)", b);
                }
                if (zw) {
                    unsigned first = 0;
                    while (!is_ws((w >> (8 * first)) & 0xFF)) first++;
                    CHECK(first_byte(zw) == first, "first_byte(%08x)", zw);
                }
                for (unsigned n = 0; n <= 6; n++) {
                    uint32_t want = 0;
                    for (unsigned k = 0; k < 4 && k < n; k++) want |= w & (0xFFu << (8 * k));
                    CHECK(low_bytes(w, n) == want, "low_bytes(%08x, %u)", w, n);
                }
            }
    // ---- token starts over running text: word by word with the carry, against the byte-wise rule
    for (int rep = 0; rep < 2000; rep++) {
        std::vector<unsigned char> text(256);
        for (auto &b : text) {
            const unsigned r = rng() % 8;
            b = r == 0 ? '\t' : r == 1 ? ' ' : r == 2 ? '\n' : (unsigned char) (33 + rng() % 94);
        }
        std::vector<uint32_t> words(64);
        std::memcpy(words.data(), text.data(), 256);
        const bool ws_in_front = rng() & 1;
        uint32_t carry = ws_in_front ? 0x80u : 0u;
        for (unsigned k = 0; k < 64; k++) {
            const uint32_t zw = ws_bytes(words[k]), st = token_start_bytes(zw, carry);
            carry = zw >> 24;
            CHECK(__builtin_popcount(st) <= 2, "more than two token starts in a word");
            for (unsigned j = 0; j < 4; j++) {
                const unsigned p = 4 * k + j;
                const bool prev_ws = p ? is_ws(text[p - 1]) : ws_in_front;
                CHECK((((st >> (8 * j)) & 0x80u) != 0) == (!is_ws(text[p]) && prev_ws), "token start at byte %u", p);
            }
        }
    }
    // ---- digit groups: every count, every digit string
    for (unsigned c = 0; c <= 4; c++) {
        unsigned lim = 1;
        for (unsigned k = 0; k < c; k++) lim *= 10;
        for (unsigned v = 0; v < lim; v++) {
            char s[16];
            std::snprintf(s, sizeof s, "%0*u", (int) c, v);
            uint32_t x = rng();             // whatever stands behind the digits is shifted out
            for (unsigned k = 0; k < c; k++) x = (x & ~(0xFFu << (8 * k))) | ((uint32_t) (s[k] - '0') << (8 * k));
            CHECK(digit_group(x, c) == v, "digit_group(%08x, %u) != %u", x, c, v);
        }
        CHECK(pow10_upto4(c) == lim, "pow10_upto4(%u)", c);
    }
    // ---- the funnel and the window: every alignment
    {
        std::vector<unsigned char> bytes(256);
        for (auto &b : bytes) b = (unsigned char) rng();
        std::vector<uint32_t> words(64);
        std::memcpy(words.data(), bytes.data(), 256);
        for (unsigned p = 0; p < 240; p++) {
            uint32_t want;
            std::memcpy(&want, bytes.data() + p, 4);
            CHECK(window_word(words.data(), p) == want, "window_word at %u", p);
        }
    }
    // ---- integer tokens: shapes, limits, alignments, terminators
    std::vector<std::string> toks = {"0", "7", "42", "007", "+5", "-5", "+0", "-0", "2147483647", "2147483648", "-2147483648",
                                     "-2147483649", "+2147483647", "+2147483648", "9999999999", "99999999999", "0000000001",
                                     "00000000001", "123456789012", "+", "-", "", "12x", "x12", "1x2", "1+2", "--1", "+-1",
                                     "1.5", "1e5", "12345", "123456", "1234567", "12345678", "123456789", "1234567890",
                                     "4294967296", "-4294967296", "3000000000", "-3000000000", ":", "/", "0x10", "١"};
    for (int k = 0; k < 4000; k++) {        // random digit strings of every length, some with signs, some with one foreign byte
        std::string t;
        const int len = 1 + (int) (rng() % 12);
        if (rng() % 5 == 0) t += "+-"[rng() % 2];
        for (int i = 0; i < len; i++) t += (char) ('0' + rng() % 10);
        if (rng() % 7 == 0) t[rng() % t.size()] = (char) (33 + rng() % 90);
        toks.push_back(t);
    }
    const char *terms[] = {" ", "\t", "\n", "\r", "\v", "\f", "x", "5", ",", "\x80", "\x1f", "\x0e", "\x08"};
    for (const std::string &t : toks)
        for (const char *term : terms)
            for (unsigned align = 0; align < 4; align++) {
                std::vector<unsigned char> buf(64, ' ');
                // (the kernel only asks about places where a token STARTS: the byte in front is whitespace; a terminator
                // that is a digit simply makes a longer token -- the restatement reads the same bytes)
                std::memcpy(buf.data() + 8 + align, t.data(), t.size());
                buf[8 + align + t.size()] = (unsigned char) term[0];
                std::vector<uint32_t> words(16);
                std::memcpy(words.data(), buf.data(), 64);
                int32_t got = 0;
                const bool ok = int_token(words.data(), 8 + align, got);
                long long want = 0;
                const bool want_ok = scalar_int(buf.data() + 8 + align, want);
                CHECK(ok == want_ok, "int_token('%s' + 0x%02x, align %u): verdict %d, expected %d", t.c_str(), (unsigned char) term[0],
                      align, (int) ok, (int) want_ok);
                if (ok) CHECK((long long) got == want, "int_token('%s', align %u) = %d, expected %lld", t.c_str(), align, got, want);
            }
    // ---- the name hash: words and bytes agree, at every alignment and length; different names differ (mostly)
    {
        unsigned long long equal_hashes = 0, pairs = 0;
        uint32_t prev = 0;
        for (unsigned len = 1; len <= 40; len++)
            for (unsigned align = 0; align < 4; align++)
                for (int rep = 0; rep < 50; rep++) {
                    std::vector<unsigned char> buf(96, '\t');
                    for (unsigned i = 0; i < len; i++) {
                        unsigned char c;
                        do c = (unsigned char) (rng() & 0xFF); while (is_ws(c));
                        buf[8 + align + i] = c;
                    }
                    std::vector<uint32_t> words(24);
                    std::memcpy(words.data(), buf.data(), 96);
                    const NameScan ns = scan_name(words.data(), 8 + align);
                    const uint32_t got_len = ns.len, hw = ns.hash;
                    CHECK(got_len == len, "scan_name: length %u, expected %u", got_len, len);
                    const uint32_t first16[4] = {ns.w0, ns.w1, ns.w2, ns.w3};
                    for (unsigned i = 0; i < 16; i++) {
                        const unsigned want = i < len ? buf[8 + align + i] : 0u;
                        CHECK(((first16[i >> 2] >> (8 * (i & 3))) & 0xFF) == want, "scan_name: byte %u of the first 16 of a %u-byte name", i, len);
                    }
                    const uint32_t hb = hash_bytes([&](uint64_t p) { return (char) buf[p]; }, 8 + align, len);
                    CHECK(hw == hb, "hash of a %u-byte name at alignment %u: words %08x, bytes %08x", len, align, hw, hb);
                    equal_hashes += hw == prev;
                    pairs++;
                    prev = hw;
                }
        CHECK(equal_hashes * 1000 < pairs, "the hash does not spread: %llu of %llu consecutive names collide", equal_hashes, pairs);
    }
    std::printf("ok %llu\n", checks);
    return 0;
}
