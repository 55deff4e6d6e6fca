// telomeri_amd/csrc/paf_parse.hip -- PAF text -> record columns ON THE DEVICE.
//
// OverlapGraph::load reads its file as a stream of whitespace-separated tokens (reference src/OverlapGraph.cpp:106-128):
// 12 typed tokens per record, then exactly 4 more that are discarded.  Once filter, scores and adjacency run on the GPU
// the host's share of a load is tokenising 15 GB of text (5 s on 16 cores for the A. thaliana-shaped input) and uploading
// the 5.5 GB of columns it produces.  Here the TEXT goes to HBM instead (pinned slabs filled by parallel preads, H2D
// copies behind them) and is tokenised where the graph build wants its input:
//   pass A  line starts per 16 KB tile (a line starts at byte 0 and after every '\n')   -> record index of every tile
//   pass B  the tile (+ overhang) staged in LDS and tokenised by all lanes: newlines, whitespace and token starts found
//           four bytes at a time, one scan for the places of lines and tokens, then the work items (line, field) dealt to the
//           threads field-major -- a token is fetched as 13 bytes at once, its digits tested and converted four at a time --
//           every token the strict [+-]digits integer inside int range, the strand one character; the two names go through a
//           device dictionary (open addressing over the text itself: a slot holds WHERE a name occurs first, an equal name
//           is recognised by comparing bytes) and come out as slot numbers
//   finish  occupied slots (listed as they were taken) sorted by first occurrence (file, offset) = first-seen order, the
//           numbering a sequential reader assigns; slot numbers in the columns become those ids; name bytes gathered into
//           one blob for the host
// Exactly like the host's parallel tokeniser (telomeri_amd/host/overlap_graph.cpp) this is only valid on CLEAN text --
// every line 16 tokens, plain integers, one-character strand, no blank lines; then line-wise parsing IS the token
// stream.  Anything else raises `clean = 0` and the caller falls back to the exact operator>> reader on the host.
// tests/test_gpu_parity.py compares columns, ids and names with the host tokeniser on every fixture and malformed input.
#include <functional>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

#include "cpu_share.h"
#include "paf_swar.h"
#include "tlm_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

namespace {

using namespace tlm_swar;       // byte classification, the strict integer, the name hash: paf_swar.h (checked on the CPU tier)

constexpr uint32_t TILE = 16u << 10;            // bytes of text per workgroup
constexpr uint32_t TILE_THREADS = 256;
// more line starts than this in one tile: not a PAF file (a line of 16 tokens is 32 bytes at least, a tile holds 512 of those)
constexpr uint32_t TILE_MAX_LINES = 528;
constexpr int MAX_FILES = 4;
constexpr unsigned long long EMPTY = ~0ull;

// a name's first occurrence: file (2 bits) | offset in that file's text (46 bits) | length (16 bits); ordered by
// (file, offset) = the order in which a sequential reader meets the names
__host__ __device__ inline unsigned long long pack_occ(uint32_t file, uint64_t off, uint32_t len) {
    return ((unsigned long long) file << 62) | ((unsigned long long) off << 16) | len;
}
__host__ __device__ inline uint32_t occ_file(unsigned long long p) { return (uint32_t) (p >> 62); }
__host__ __device__ inline uint64_t occ_off(unsigned long long p) { return (p >> 16) & ((1ull << 46) - 1); }
__host__ __device__ inline uint32_t occ_len(unsigned long long p) { return (uint32_t) (p & 0xFFFFu); }

struct Texts {
    const char *p[MAX_FILES];
};

struct Tmp {
    void *p = nullptr;
    ~Tmp() { tlm_pool_free(p); }
};

// ---- pass A: line starts per tile.  A line starts at the text's first byte and after every newline: the tile [b, e) holds
// (b == 0 or a newline at b - 1) + the newlines of [b, e - 1).  16 bytes per lane and load (the buffer is 64 bytes longer than
// the text and 16-byte aligned).
__global__ __launch_bounds__(TILE_THREADS) void count_lines_kernel(const char *__restrict__ text, uint64_t bytes,
                                                                   uint64_t n_tiles, uint32_t *__restrict__ tile_lines) {
    __shared__ uint32_t s_cnt;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        const uint64_t b = tile * TILE, last = min(bytes, b + TILE) - 1;
        uint32_t n = threadIdx.x == 0 && (b == 0 || text[b - 1] == '\n');
        for (uint64_t p = b + threadIdx.x * 16u; p < last; p += TILE_THREADS * 16u) {
            if (p + 16 <= last) {
                const uint4 v = *reinterpret_cast<const uint4 *>(text + p);
                n += __popc(newline_bytes(v.x)) + __popc(newline_bytes(v.y)) + __popc(newline_bytes(v.z)) + __popc(newline_bytes(v.w));
            } else {
                for (uint64_t q = p; q < last; q++) n += text[q] == '\n';
            }
        }
        for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_cnt, n);
        __syncthreads();
        if (threadIdx.x == 0) tile_lines[tile] = s_cnt;
        __syncthreads();
    }
}

// ---- the dictionary
struct Dict {
    unsigned long long *slot;       // first occurrence of the slot's name, or EMPTY
    uint32_t mask;
    uint32_t *n_names;              // occupied slots; n_names[1] = flags of the parse
    uint32_t *inserted;             // the occupied slots in the order they were taken, [n_slots / 2] (a table fuller than that
                                    // is refused): tlm_paf_finish reads these instead of searching the table
};

// ---- pass B.  Round 3: the tile is tokenised by ALL lanes, field by field, instead of one thread per line walking its bytes
// (16 tokens x ~6 dependent byte reads out of LDS, a third of the lanes idle because ~330 lines meet 256 threads):
//   1. the tile (16 KB) and its overhang (1 KB: the tile's last line ends in there) are staged in LDS;
//   2. every thread classifies 17 words of it, four bytes at a time (SWAR): newlines, whitespace, token starts (a byte that
//      is not whitespace after one that is).  It writes the token starts of its stretch into slots of its own and keeps its
//      (three at most) line starts in registers; one scan over the workgroup then numbers lines and tokens: s_linetok[l] =
//      tokens in front of line l -- a line is 16 clean tokens iff the next line's count is 16 more (checked for every
//      line) -- and s_lineslot[l] = where the line's first token is looked for; token f of the line is f slots further,
//      over at most two or three threads' slots;
//   3. the work items (line, integer field) -- 10 per line: nine integers and the strand -- are dealt to the threads field-
//      major, so that a wavefront parses the same field of 64 consecutive lines and its stores are coalesced: a token is 13
//      bytes fetched at once (five aligned words), its digits are found by a SWAR test, converted four at a time and combined;
//   4. the items (line, name) likewise: length and hash a word at a time, then the dictionary.
// A last line that runs past the overhang (names of thousands of characters) is parsed by one thread out of HBM, byte by
// byte, with the round-2 code (parse_line below).
constexpr uint32_t TILE_OVERHANG = 1024;
constexpr uint32_t STAGED_MAX = TILE + TILE_OVERHANG;
constexpr uint32_t STRETCH_WORDS = STAGED_MAX / 4 / TILE_THREADS;       // words a thread classifies
static_assert(STRETCH_WORDS * 4 * TILE_THREADS == STAGED_MAX, "the staged text is dealt out in whole words");
static_assert(STRETCH_WORDS % 2 == 1, "an odd stride between the threads' words: their LDS reads fall into different banks");
constexpr uint32_t TOK_SLOTS = STRETCH_WORDS * 2;                         // a token and the blank after it: two bytes at least
static_assert(STAGED_MAX < (1u << 14) * 2 && TILE_THREADS * TOK_SLOTS < 65536 && TOK_SLOTS < 64, "offsets and token counts are packed");

// `own(i)` = byte i of the name being interned (LDS or HBM); `off` = where it stands in its file.  `own16`: the name's bytes
// as four words, zero behind its end, when it has 16 bytes or fewer (`short_name`); the first occurrence a slot shows is then
// fetched as five aligned words at once and compared a word at a time -- one memory latency per probe instead of one per byte.
template <class Own>
__device__ inline uint32_t dict_intern(const Dict &d, const Texts &tx, const Own &own, bool short_name, uint4 own16, uint32_t file,
                                       uint64_t off, uint32_t len, uint32_t hash, bool have_first = false,
                                       unsigned long long first = 0) {
    const unsigned long long mine = pack_occ(file, off, len);
    uint32_t probes = 0;
    for (uint32_t j = hash & d.mask;; j = (j + 1) & d.mask) {
        if (probes++ > d.mask) {            // every slot taken by another name: more names than the table was made for
            atomicOr(d.n_names + 1, 8u);
            return 0;
        }
        unsigned long long cur = have_first && probes == 1u ? first : d.slot[j];      // (`first`: slot hash & mask, read earlier)
        if (cur == EMPTY) {
            cur = atomicCAS(&d.slot[j], EMPTY, mine);
            if (cur == EMPTY) {
                const uint32_t k = atomicAdd(d.n_names, 1u);
                if (k <= d.mask / 2u) d.inserted[k] = j;
                return j;
            }
        }
        if (occ_len(cur) == len) {          // whatever occurrence the slot shows at the moment spells the slot's name
            const char *o = tx.p[occ_file(cur)] + occ_off(cur);
            bool same = true;
            if (short_name) {               // (the text buffers are 64 bytes longer than the text: the fifth word exists)
                const uint32_t *g = reinterpret_cast<const uint32_t *>(reinterpret_cast<uintptr_t>(o) & ~(uintptr_t) 3);
                const uint32_t sh = ((uint32_t) reinterpret_cast<uintptr_t>(o) & 3u) * 8u;
                const uint32_t g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4];
                same = low_bytes(funnel(g1, g0, sh), len) == own16.x
                       && low_bytes(funnel(g2, g1, sh), len > 4u ? len - 4u : 0u) == own16.y
                       && low_bytes(funnel(g3, g2, sh), len > 8u ? len - 8u : 0u) == own16.z
                       && low_bytes(funnel(g4, g3, sh), len > 12u ? len - 12u : 0u) == own16.w;
            } else {
                for (uint32_t i = 0; i < len && same; i++) same = o[i] == own(i);
            }
            if (same) {
                if (mine < cur) atomicMin(&d.slot[j], mine);
                return j;
            }
        }
    }
}

struct Columns {
    int32_t *q_id, *t_id, *q_len, *q_start, *q_end, *t_len, *t_start, *t_end, *matches;
    uint8_t *strand;
};

// ---- the byte-wise reader of ONE line (a line that runs past the staged text; round 2 parsed every line with it).  The
// tokens are consumed as they come: the two names are kept as (start, length); every other token is accumulated as a
// decimal integer WHILE its end is looked for -- the strict integer of the host tokeniser: an optional sign, then 1..10
// digits and nothing else, inside the int32 range.
struct Line {
    uint64_t q_name = 0, t_name = 0;
    uint32_t q_n = 0, t_n = 0;
    int32_t v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
    bool ok = true, minus = false;
    int nt = 0;
};

// `rd(p)` = the byte at position p; [p, end) holds no newline
template <class Read>
__device__ void parse_line(const Read &rd, uint64_t p, const uint64_t end, Line &L) {
    for (;;) {
        char c = 0;
        while (p < end && ws_char((unsigned char) (c = rd(p)))) p++;
        if (p >= end) break;
        const uint64_t ts = p;
        bool neg = false, bad = false, more = true;
        uint32_t nd = 0;
        unsigned long long acc = 0;
        if (c == '+' || c == '-') {
            neg = c == '-';
            p++;
            more = p < end && !ws_char((unsigned char) (c = rd(p)));
        }
        while (more) {
            const uint32_t d = (uint32_t) (unsigned char) c - (uint32_t) '0';
            bad |= d > 9u;
            acc = acc * 10u + d;            // (wraps on a long name: `nd` refuses those as integers)
            nd++;
            p++;
            more = p < end && !ws_char((unsigned char) (c = rd(p)));
        }
        const uint64_t len = p - ts;
        const long long a = neg ? -(long long) acc : (long long) acc;
        const bool int_ok = !bad && nd >= 1 && nd <= 10 && a <= 2147483647ll && a >= -2147483648ll;
        const int32_t v = (int32_t) a;
        switch (L.nt) {
            case 0: L.q_name = ts; L.q_n = (uint32_t) min(len, (uint64_t) 0x10000); break;
            case 1: L.ok = L.ok && int_ok; L.v0 = v; break;
            case 2: L.ok = L.ok && int_ok; L.v1 = v; break;
            case 3: L.ok = L.ok && int_ok; L.v2 = v; break;
            case 4: L.ok = L.ok && len == 1; L.minus = neg; break;
            case 5: L.t_name = ts; L.t_n = (uint32_t) min(len, (uint64_t) 0x10000); break;
            case 6: L.ok = L.ok && int_ok; L.v3 = v; break;
            case 7: L.ok = L.ok && int_ok; L.v4 = v; break;
            case 8: L.ok = L.ok && int_ok; L.v5 = v; break;
            case 9: L.ok = L.ok && int_ok; L.v6 = v; break;
            case 10:                        // alignment block length and mapping quality: must be integers, never used
            case 11: L.ok = L.ok && int_ok; break;
            default: break;                 // 12..15: discarded tokens; 16 and beyond: counted, refused by the caller
        }
        L.nt++;
    }
}

// flags: 1 = a line is not 16 clean tokens, 2 = too many line starts in a tile, 4 = a name of 65 535 bytes or more
__global__ __launch_bounds__(TILE_THREADS) void parse_tiles_kernel(Texts tx, uint32_t file, uint64_t bytes, uint64_t n_tiles,
                                                                   const uint64_t *__restrict__ tile_base, Dict dict,
                                                                   Columns out, uint32_t *__restrict__ flags) {
    // the staged text; the bytes behind it (to the end of its last 16-byte piece and 32 more) are newlines: a token that
    // ends with the text ends there, and no reader below needs a bound
    __shared__ __attribute__((aligned(16))) uint32_t s_words[STAGED_MAX / 4 + 12];
    // token starts: thread t writes those of its stretch to s_tok[t * TOK_SLOTS ..] as it finds them, s_cnt[t] says how many
    __shared__ uint16_t s_tok[TILE_THREADS * TOK_SLOTS];
    __shared__ uint8_t s_cnt[TILE_THREADS];
    // per line: tokens of the staged text in front of it; where its first token is looked for (thread << 8 | slot)
    __shared__ uint16_t s_linetok[TILE_MAX_LINES + 1], s_lineslot[TILE_MAX_LINES + 1];
    __shared__ uint16_t s_qhead[TILE_MAX_LINES + 1];
    __shared__ uint32_t s_qslot[TILE_MAX_LINES + 1];
    __shared__ uint32_t s_wave[TILE_THREADS / 64];
    __shared__ uint32_t s_end;          // of the newline that ends the tile's last line: tile offset << 14 | thread << 6 | its tokens in front
    __shared__ uint32_t s_end_tok;      // tokens of the staged text in front of that newline
    __shared__ uint32_t s_last_start;   // tile offset of the tile's last line
    __shared__ uint32_t s_bad;          // a stretch with more line starts than clean text can have: nothing of the tile is parsed
    const char *text = tx.p[file];
    const char *s_text = reinterpret_cast<const char *>(s_words);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t b = tile * TILE, e = min(bytes, b + TILE);
        const uint32_t staged = (uint32_t) min((uint64_t) STAGED_MAX, bytes - b);
        // ---- 1. stage: 16 bytes per lane and load (the buffer is 64 bytes longer than the text)
        for (uint32_t o = tid * 16u; o < staged; o += TILE_THREADS * 16u) {
            uint4 v = *reinterpret_cast<const uint4 *>(text + b + o);
            if (o + 16u > staged) {         // the text ends inside this piece: newlines behind it
                uint32_t *w = reinterpret_cast<uint32_t *>(&v);
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t at = o + 4u * k;
                    if (at >= staged) w[k] = 0x0A0A0A0Au;
                    else if (at + 4u > staged) w[k] = (w[k] & (0xFFFFFFFFu >> (8u * (at + 4u - staged)))) | (0x0A0A0A0Au << (8u * (staged - at)));
                }
            }
            *reinterpret_cast<uint4 *>(s_words + (o >> 2)) = v;
            if (o + 16u >= staged) {        // (the piece that holds the last byte: 32 bytes of newlines behind it)
                *reinterpret_cast<uint4 *>(s_words + (o >> 2) + 4) = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
                *reinterpret_cast<uint4 *>(s_words + (o >> 2) + 8) = make_uint4(0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au, 0x0A0A0A0Au);
            }
        }
        const char before = b ? text[b - 1] : '\n';
        if (tid == 0) {
            s_end = 0xFFFFFFFFu;
            s_bad = 0;
        }
        __syncthreads();
        // ---- 2. classify, once.  A line starts after every newline of [0, last) (and at the tile's first byte when the byte
        // before the tile is one; a start at the tile's end belongs to the next tile); a token starts at a byte that is not
        // whitespace behind one that is -- two per word at most.  A stretch of 68 bytes of clean text holds three line
        // starts at most (a line of 16 tokens has 32 bytes): kept in registers until the scan has said where they go.
        const uint32_t last = (uint32_t) (e - b) - 1u;
        const uint32_t w_first = tid * STRETCH_WORDS;
        uint16_t *my_tok = s_tok + tid * TOK_SLOTS;
        uint32_t n_tok = 0, n_line = 0, ls0 = 0, ls1 = 0, ls2 = 0;         // line starts: (tile offset << 8) | own tokens in front
        if (tid == 0 && before == '\n') n_line = 1;                         // (ls0 = 0: offset 0, no token in front)
        {
            uint32_t carry = tid ? (ws_bytes(s_words[w_first - 1]) >> 24) : (ws_char((unsigned char) before) ? 0x80u : 0u);
            for (uint32_t k = 0; k < STRETCH_WORDS; k++) {
                const uint32_t o = 4u * (w_first + k);
                if (o >= staged) break;
                const uint32_t w = s_words[w_first + k];
                const uint32_t zw = ws_bytes(w);
                const uint32_t st = token_start_bytes(zw, carry);
                carry = zw >> 24;
                for (uint32_t z = newline_bytes(w); z; z &= z - 1u) {         // (one word in twenty-five has a newline)
                    const uint32_t kk = first_byte(z), p = o + kk;
                    const uint32_t in_front = n_tok + (uint32_t) __popc(st & ((1u << (8u * kk)) - 1u));
                    if (p < last) {
                        const uint32_t rec = ((p + 1u) << 8) | in_front;
                        if (n_line == 0) ls0 = rec;
                        else if (n_line == 1) ls1 = rec;
                        else if (n_line == 2) ls2 = rec;
                        n_line++;
                    } else if (p < staged) {        // the first of these ends the tile's last line
                        atomicMin(&s_end, (p << 14) | (tid << 6) | in_front);
                    }
                }
                if (st) {
                    my_tok[n_tok] = (uint16_t) (o + first_byte(st));
                    const uint32_t st2 = st & (st - 1u);
                    if (st2) my_tok[n_tok + 1u] = (uint16_t) (o + first_byte(st2));
                    n_tok += (uint32_t) __popc(st);
                }
            }
        }
        s_cnt[tid] = (uint8_t) n_tok;
        if (n_line > 3u) {
            atomicOr(flags, 1u);
            s_bad = 1;
        }
        // places: one scan of (lines << 16 | tokens) over the workgroup
        uint32_t incl = (n_line << 16) | n_tok;
        for (uint32_t d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t) __shfl_up((int) incl, (int) d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t all = 0, mine_base = 0;
        for (uint32_t k = 0; k < TILE_THREADS / 64; k++) {
            if (k < wave) mine_base += s_wave[k];
            all += s_wave[k];
        }
        const uint32_t total = all >> 16, total_tok = all & 0xFFFFu;
        const bool countable = total <= TILE_MAX_LINES && !s_bad;
        if (total > TILE_MAX_LINES && tid == 0) atomicOr(flags, 2u);
        if (countable) {
            const uint32_t excl = mine_base + incl - ((n_line << 16) | n_tok);
            const uint32_t at_line = excl >> 16, at_tok = excl & 0xFFFFu;
            for (uint32_t q = 0; q < n_line; q++) {
                const uint32_t rec = q == 0 ? ls0 : q == 1 ? ls1 : ls2;
                s_linetok[at_line + q] = (uint16_t) (at_tok + (rec & 0xFFu));
                s_lineslot[at_line + q] = (uint16_t) ((tid << 8) | (rec & 0xFFu));
                if (at_line + q + 1u == total) s_last_start = rec >> 8;
            }
            const uint32_t endpack = s_end;
            if (endpack != 0xFFFFFFFFu && tid == ((endpack >> 6) & 0xFFu)) s_end_tok = at_tok + (endpack & 63u);
        }
        __syncthreads();
        if (countable && total) {
            const uint32_t endpack = s_end;
            // the tile's last line ends at that newline, or with the text; otherwise it runs past the staged bytes
            const bool slow_last = endpack == 0xFFFFFFFFu && b + staged != bytes;
            const uint32_t end_tok = endpack == 0xFFFFFFFFu ? total_tok : s_end_tok;
            const uint32_t n_fast = total - (slow_last ? 1u : 0u);
            const uint64_t r0 = tile_base[tile];
            // token f of line l (it exists: the callers have compared its number with total_tok): the f-th behind the place
            // the line's first token is looked for, over the threads' slots
            auto token_at = [&](uint32_t l, uint32_t f) {
                const uint32_t ref = s_lineslot[l];
                uint32_t t = ref >> 8, idx = (ref & 0xFFu) + f, c;
                while (idx >= (c = s_cnt[t]) && t < TILE_THREADS - 1u) {
                    idx -= c;
                    t++;
                }
                return (uint32_t) s_tok[t * TOK_SLOTS + min(idx, TOK_SLOTS - 1u)];      // (idx < s_cnt[t] here; the clamp keeps any other state inside the table)
            };
            // ---- every line 16 tokens
            for (uint32_t l = tid; l < n_fast; l += TILE_THREADS) {
                const uint32_t next = l + 1u < total ? (uint32_t) s_linetok[l + 1] : end_tok;
                if (next - (uint32_t) s_linetok[l] != 16u) atomicOr(flags, 1u);
            }
            // ---- 3. the names, first half.  `name_scan`: length and hash of the name at `pos` (a word at a time; it ends at the
            // newlines behind the staged text at the latest) and its first 16 bytes as words.  The dictionary slot the hash
            // points at is READ here and looked at behind the integer fields: the two dependent memory latencies of a lookup
            // (the slot, then the text of the first occurrence it names) are what the names cost, and the first of them now
            // passes while the integers are parsed.
            struct Name {
                uint32_t pos, len, hash;
                uint4 first16;
                unsigned long long slot0;
            };
            auto name_scan = [&](uint32_t pos) {
                const NameScan ns = scan_name(s_words, pos);
                Name nm;
                nm.pos = pos;
                nm.len = ns.len;
                nm.hash = ns.hash;
                nm.first16 = make_uint4(ns.w0, ns.w1, ns.w2, ns.w3);
                nm.slot0 = dict.slot[nm.hash & dict.mask];
                return nm;
            };
            auto name_slot = [&](const Name &nm) {
                const char *name = s_text + nm.pos;
                return dict_intern(dict, tx, [&](uint32_t q) { return name[q]; }, nm.len <= 16u, nm.first16, file, b + nm.pos, nm.len,
                                   nm.hash, true, nm.slot0);
            };
            // Query names: the lines of one query follow each other (the overlaps of a read are written together), so a line
            // whose query name is the previous line's takes that line's slot -- a name seen before can be no first occurrence.
            // s_qhead[l] = l + 1 for a line that looks its name up, 0 for one that repeats; a running maximum (wave 0) then
            // names every line's head.
            auto repeats_previous = [&](uint32_t l, uint32_t pos) {
                if (l == 0u || (uint32_t) s_linetok[l - 1u] >= total_tok) return false;
                const uint32_t prev = token_at(l - 1u, 0u);
                for (uint32_t k = 0;; k++) {
                    const uint32_t w = window_word(s_words, pos + 4u * k), v = window_word(s_words, prev + 4u * k);
                    const uint32_t z = ws_bytes(w);
                    if (z) {            // the name ends in this word: the same bytes in front, and the other ends there too
                        const uint32_t rest = first_byte(z);
                        return low_bytes(w, rest) == low_bytes(v, rest) && ws_char((v >> (8u * rest)) & 0xFFu);
                    }
                    if (w != v) return false;
                }
            };
            // (this thread's line of the first round: l = tid; a tile of more than 256 lines does the rest behind the integers)
            const bool t_mine = tid < n_fast && (uint32_t) s_linetok[tid] + 5u < total_tok;
            const bool q_there = tid < n_fast && (uint32_t) s_linetok[tid] < total_tok;
            Name t_name{}, q_name{};
            bool q_head = false;
            if (t_mine) t_name = name_scan(token_at(tid, 5u));
            if (q_there) {
                const uint32_t pos = token_at(tid, 0u);
                q_head = !repeats_previous(tid, pos);
                if (q_head) q_name = name_scan(pos);
            }
            // ---- 4. the integer fields and the strand, field-major
            const uint32_t inv = n_fast > 1u ? 0xFFFFFFFFu / n_fast + 1u : 0u;       // i / n_fast = umulhi(i, inv) for i < 2^16, n_fast > 1
            for (uint32_t i = tid; i < n_fast * 10u; i += TILE_THREADS) {
                const uint32_t f = n_fast == 1u ? i : __umulhi(i, inv), l = i - f * n_fast;
                const uint32_t j = (uint32_t) s_linetok[l] + f + 1u + (f >= 4u ? 1u : 0u);      // tokens 1..4, 6..11
                if (j >= total_tok) continue;           // (a short line: refused above)
                const uint32_t pos = token_at(l, f + 1u + (f >= 4u ? 1u : 0u));
                const uint64_t r = r0 + l;
                if (f == 3u) {                          // the strand: one character
                    const uint32_t w = window_word(s_words, pos);
                    if (!ws_char((w >> 8) & 0xFFu)) atomicOr(flags, 1u);
                    out.strand[r] = (w & 0xFFu) == '-';
                    continue;
                }
                int32_t v;
                if (!int_token(s_words, pos, v)) atomicOr(flags, 1u);
                int32_t *col = f == 0u ? out.q_len : f == 1u ? out.q_start : f == 2u ? out.q_end : f == 4u ? out.t_len
                               : f == 5u ? out.t_start : f == 6u ? out.t_end : f == 7u ? out.matches : nullptr;
                if (col) col[r] = v;                    // (alignment block length, mapping quality: integers, never used)
            }
            // ---- 5. the names, second half
            if (t_mine) out.t_id[r0 + tid] = (int32_t) name_slot(t_name);
            if (tid < n_fast) {
                s_qhead[tid] = q_there && !q_head ? (uint16_t) 0 : (uint16_t) (tid + 1u);
                s_qslot[tid] = q_head ? name_slot(q_name) : 0u;
            }
            for (uint32_t l = tid + TILE_THREADS; l < n_fast; l += TILE_THREADS) {
                if ((uint32_t) s_linetok[l] + 5u < total_tok) out.t_id[r0 + l] = (int32_t) name_slot(name_scan(token_at(l, 5u)));
                if ((uint32_t) s_linetok[l] >= total_tok) { s_qhead[l] = (uint16_t) (l + 1u); s_qslot[l] = 0; continue; }
                const uint32_t pos = token_at(l, 0u);
                const bool repeat = repeats_previous(l, pos);
                s_qhead[l] = repeat ? (uint16_t) 0 : (uint16_t) (l + 1u);
                if (!repeat) s_qslot[l] = name_slot(name_scan(pos));
            }
            __syncthreads();
            if (wave == 0) {
                uint32_t run = 0;
                for (uint32_t base = 0; base < n_fast; base += 64u) {
                    const uint32_t l = base + lane;
                    uint32_t m = l < n_fast ? (uint32_t) s_qhead[l] : 0u;
                    for (uint32_t d = 1; d < 64; d <<= 1) {
                        const uint32_t up = (uint32_t) __shfl_up((int) m, (int) d);
                        if (lane >= d) m = max(m, up);
                    }
                    m = max(m, run);
                    if (l < n_fast) s_qhead[l] = (uint16_t) m;
                    run = (uint32_t) __shfl((int) m, 63);
                }
            }
            __syncthreads();
            for (uint32_t l = tid; l < n_fast; l += TILE_THREADS) out.q_id[r0 + l] = (int32_t) s_qslot[(uint32_t) s_qhead[l] - 1u];
            // ---- a last line that runs past the staged bytes: one thread, byte by byte out of HBM
            if (slow_last && tid == 0) {
                const uint64_t from = b + s_last_start;
                uint64_t q = from;
                while (q < bytes && text[q] != '\n') q++;
                auto rd = [&](uint64_t o) { return text[o]; };
                Line L;
                parse_line(rd, from, q, L);
                const uint64_t r = r0 + total - 1u;
                if (!L.ok || L.nt != 16) {
                    atomicOr(flags, 1u);
                } else if (L.q_n >= 0xFFFFu || L.t_n >= 0xFFFFu) {
                    atomicOr(flags, 4u);
                } else {
                    const uint64_t qn = L.q_name, tn = L.t_name;
                    out.q_id[r] = (int32_t) dict_intern(dict, tx, [&](uint32_t k) { return text[qn + k]; }, false, make_uint4(0u, 0u, 0u, 0u), file, qn, L.q_n,
                                                        hash_bytes(rd, qn, L.q_n));
                    out.t_id[r] = (int32_t) dict_intern(dict, tx, [&](uint32_t k) { return text[tn + k]; }, false, make_uint4(0u, 0u, 0u, 0u), file, tn, L.t_n,
                                                        hash_bytes(rd, tn, L.t_n));
                    out.q_len[r] = L.v0; out.q_start[r] = L.v1; out.q_end[r] = L.v2;
                    out.t_len[r] = L.v3; out.t_start[r] = L.v4; out.t_end[r] = L.v5;
                    out.matches[r] = L.v6;
                    out.strand[r] = L.minus;
                }
            }
        }
        __syncthreads();
    }
}

// ---- finish
__global__ void collect_slots_kernel(const unsigned long long *__restrict__ slot, const uint32_t *__restrict__ inserted, uint32_t n,
                                     unsigned long long *__restrict__ keys, uint32_t *__restrict__ vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = inserted[i];
    keys[i] = slot[j];
    vals[i] = j;
}

__global__ void names_kernel(uint32_t n, const unsigned long long *__restrict__ occ, const uint32_t *__restrict__ slot_of,
                             Texts tx, uint32_t *__restrict__ id_of_slot, uint32_t *__restrict__ len_out,
                             uint8_t *__restrict__ is_ctg) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    id_of_slot[slot_of[i]] = i;
    const uint32_t len = occ_len(occ[i]);
    len_out[i] = len;
    const char *s = tx.p[occ_file(occ[i])] + occ_off(occ[i]);
    // Utils::startsWithInsensitive(name, "ctg") (Utils.hpp:17-27)
    is_ctg[i] = len >= 3 && (s[0] | 0x20) == 'c' && (s[1] | 0x20) == 't' && (s[2] | 0x20) == 'g';
}

__global__ void blob_kernel(uint32_t n, const unsigned long long *__restrict__ occ, const uint32_t *__restrict__ off,
                            Texts tx, char *__restrict__ blob) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const char *s = tx.p[occ_file(occ[i])] + occ_off(occ[i]);
    const uint32_t len = occ_len(occ[i]);
    for (uint32_t k = 0; k < len; k++) blob[off[i] + k] = s[k];
}

__global__ void remap_kernel(uint64_t n, int32_t *__restrict__ q_id, int32_t *__restrict__ t_id,
                             const uint32_t *__restrict__ id_of_slot) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        q_id[i] = (int32_t) id_of_slot[(uint32_t) q_id[i]];
        t_id[i] = (int32_t) id_of_slot[(uint32_t) t_id[i]];
    }
}

unsigned grid_of(uint64_t n, unsigned per = 256) {
    return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + per - 1) / per, 256u * 32u));
}

}   // namespace

struct tlm_paf {
    tlm_ctx *ctx = nullptr;
    int n_files = 0;
    bool poisoned = false;      // a file was refused (not clean) or failed: its names are in the dictionary, the handle is only good for destroy
    char *text[MAX_FILES] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t bytes[MAX_FILES] = {0, 0, 0, 0};
    uint64_t n[MAX_FILES] = {0, 0, 0, 0};
    int anchors[MAX_FILES] = {0, 0, 0, 0};
    Columns cols[MAX_FILES] = {};
    unsigned long long *slot = nullptr;
    uint64_t n_slots = 0;
    uint32_t *n_names_dev = nullptr;     // [0] occupied slots, [1] flags
    uint32_t *inserted = nullptr;        // Dict::inserted
    bool finished = false;
    uint32_t n_names = 0;
    uint8_t *is_ctg = nullptr;
    uint32_t *name_off = nullptr;        // [n_names + 1]
    char *blob = nullptr;
    uint64_t blob_bytes = 0;
    double upload_ms = 0, parse_ms = 0, finish_ms = 0;
    char *pending = nullptr;             // text buffer handed out by tlm_paf_text_alloc, not yet committed
    uint64_t pending_bytes = 0;
    tlm_exchange_info share{};           // what the sharded load moved between the ranks, summed over the files
};

// The piece of the sharded load and of the uploads: TLM_TEXT_PIECE.  CHECKED BUILD ONLY (tests): TLM_TEXT_PIECE_LOG2 makes the
// pieces of tlm_paf_add_file small, so that a text of a few hundred KB is brought in runs of several pieces by several ranks
// (never together with the upload helper, whose window slots are TLM_TEXT_PIECE).
static uint64_t text_piece_bytes() {
#ifdef TLM_CHECKED
    if (const char *e = getenv("TLM_TEXT_PIECE_LOG2")) {
        const int b = atoi(e);
        if (b >= 12 && b <= 30) return 1ull << b;
    }
#endif
    return TLM_TEXT_PIECE;
}

// does this handle's context load its files in shares?  (a communicator of more than one rank, and no request for the old way)
static bool sharded_load(const tlm_paf *p) {
    return p->ctx->comm && p->ctx->world > 1 && !getenv("TLM_LOAD_REPLICATED");
}
static void add_share(tlm_paf *p, const tlm_exchange_info &x) {
    p->share.bytes_sent += x.bytes_sent;
    p->share.bytes_received += x.bytes_received;
    p->share.wall_ms += x.wall_ms;
}

extern "C" int tlm_paf_create(tlm_ctx *ctx, tlm_paf **out) {
    TLM_CHECK(ctx && out, TLM_EINVAL, "tlm_paf_create: null argument");
    tlm_paf *p = new tlm_paf();
    p->ctx = ctx;
    *out = p;
    return TLM_OK;
}

extern "C" void tlm_paf_destroy(tlm_paf *p) {
    if (!p) return;
    (void) hipSetDevice(p->ctx->device);
    (void) hipStreamSynchronize(p->ctx->stream);
    for (int f = 0; f < MAX_FILES; f++) {
        tlm_pool_free(p->text[f]);
        void *c[] = {p->cols[f].q_id, p->cols[f].t_id, p->cols[f].q_len, p->cols[f].q_start, p->cols[f].q_end, p->cols[f].t_len,
                     p->cols[f].t_start, p->cols[f].t_end, p->cols[f].matches, p->cols[f].strand};
        for (void *b : c) tlm_pool_free(b);
    }
    tlm_pool_free(p->slot); tlm_pool_free(p->inserted); tlm_pool_free(p->n_names_dev); tlm_pool_free(p->is_ctg); tlm_pool_free(p->name_off);
    tlm_pool_free(p->blob);
    tlm_pool_free(p->pending);
    delete p;
}

// text already on the device (p->text[f], p->bytes[f]): tokenise it
static int tokenise_on_device(tlm_paf *p, int f, uint64_t *n_records, int *clean) {
    hipStream_t s = p->ctx->stream;
    const uint64_t bytes = p->bytes[f];
    *clean = 1;
    *n_records = 0;
    if (bytes == 0) return TLM_OK;
    const auto t0 = std::chrono::steady_clock::now();
    const uint64_t n_tiles = (bytes + TILE - 1) / TILE;
    Tmp tile_lines, tile_base, scan_tmp;
    if (int rc = tlm_pool_alloc(&tile_lines.p, n_tiles * 4, s)) return rc;
    if (int rc = tlm_pool_alloc(&tile_base.p, (n_tiles + 1) * 8, s)) return rc;
    hipLaunchKernelGGL(count_lines_kernel, dim3(grid_of(n_tiles, 1)), dim3(TILE_THREADS), 0, s, p->text[f], bytes, n_tiles,
                       (uint32_t *) tile_lines.p);
    {
        struct ToU64 {
            __host__ __device__ uint64_t operator()(uint32_t v) const { return v; }
        };
        auto in = rocprim::make_transform_iterator((const uint32_t *) tile_lines.p, ToU64());
        size_t tb = 0;
        TLM_HIP(rocprim::exclusive_scan(nullptr, tb, in, (uint64_t *) tile_base.p, (uint64_t) 0, (size_t) n_tiles,
                                        rocprim::plus<uint64_t>(), s));
        if (int rc = tlm_pool_alloc(&scan_tmp.p, tb, s)) return rc;
        TLM_HIP(rocprim::exclusive_scan(scan_tmp.p, tb, in, (uint64_t *) tile_base.p, (uint64_t) 0, (size_t) n_tiles,
                                        rocprim::plus<uint64_t>(), s));
    }
    uint64_t last_base = 0;
    uint32_t last_lines = 0;
    TLM_HIP(hipMemcpyAsync(&last_base, (uint64_t *) tile_base.p + (n_tiles - 1), 8, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipMemcpyAsync(&last_lines, (uint32_t *) tile_lines.p + (n_tiles - 1), 4, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    const uint64_t n = last_base + last_lines;
    TLM_CHECK(n < (1ull << 31), TLM_ELIMIT, "tlm_paf: %llu lines exceed the 2^31 record limit", (unsigned long long) n);
    // dictionary: created with the first file, sized from its record count (names are few compared with records)
    if (!p->slot) {
        // (at least 4 M slots = 32 MB: a small first file may be followed by a large one; a table that ends up more than
        // half full is not trusted and sends the graph to the host tokeniser)
        uint64_t want = std::max<uint64_t>(1u << 22, std::min<uint64_t>(n, 1ull << 27));
        uint64_t cap = 1;
        while (cap < want) cap <<= 1;
        p->n_slots = cap;
        if (int rc = tlm_pool_alloc((void **) &p->slot, cap * 8, s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &p->inserted, (cap / 2 + 1) * 4, s)) return rc;
        if (int rc = tlm_pool_alloc((void **) &p->n_names_dev, 16, s)) return rc;
        TLM_HIP(hipMemsetAsync(p->slot, 0xFF, cap * 8, s));
        TLM_HIP(hipMemsetAsync(p->n_names_dev, 0, 16, s));
    }
    Columns &c = p->cols[f];
    int32_t **i32[] = {&c.q_id, &c.t_id, &c.q_len, &c.q_start, &c.q_end, &c.t_len, &c.t_start, &c.t_end, &c.matches};
    for (int32_t **col : i32)
        if (int rc = tlm_pool_alloc((void **) col, (n ? n : 1) * 4, s)) return rc;
    if (int rc = tlm_pool_alloc((void **) &c.strand, n ? n : 1, s)) return rc;
    Texts tx;
    for (int k = 0; k < MAX_FILES; k++) tx.p[k] = p->text[k];
    Dict d{p->slot, (uint32_t) (p->n_slots - 1), p->n_names_dev, p->inserted};
    TLM_HIP(hipMemsetAsync(p->n_names_dev + 1, 0, 4, s));
    hipLaunchKernelGGL(parse_tiles_kernel, dim3(grid_of(n_tiles, 1)), dim3(TILE_THREADS), 0, s, tx, (uint32_t) f, bytes, n_tiles,
                       (const uint64_t *) tile_base.p, d, c, p->n_names_dev + 1);
    uint32_t h[2] = {0, 0};
    TLM_HIP(hipMemcpyAsync(h, p->n_names_dev, 8, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    TLM_HIP(hipGetLastError());
    p->parse_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (h[1] != 0 || (uint64_t) h[0] * 2 > p->n_slots) {       // not clean PAF text (or a dictionary too full to trust)
        *clean = 0;
        return TLM_OK;
    }
    p->n[f] = n;
    *n_records = n;
    return TLM_OK;
}

// The registered-mapping upload of a file into `dst` (device memory).  Straight out of the page cache: the file is
// mapped, and piece by piece (1 GiB) the pages are faulted in by all cores, pinned (hipHostRegister) and handed to ONE DMA
// each; the faults of piece k+1 run while piece k crosses PCIe.  No staging copy on the host at all (measured on the
// 15.6 GB read-read file: 0.29 s of copies at 55 GB/s, against 1.3 s of preads into pinned slabs, the fallback).
// `defer_to`: the context that unmaps the file later; null = the mapping stays until the process ends (upload helper).
struct PieceSink {           // where the 1 GiB pieces of an upload go (default: consecutive places of one buffer)
    char *base = nullptr;
    uint64_t n_slots = 0;                               // 0: piece k at base + k GiB; else at slot k % n_slots of a window
    std::function<bool(uint64_t)> before;               // slot of piece k may be written? (waits; false = give up)
    std::function<bool(uint64_t)> arrived;              // piece k is on the device
    char *at(uint64_t k, uint64_t piece) const { return base + (n_slots ? k % n_slots : k) * piece; }
};

// Pieces [k0, k1) only (the sharded load: a rank brings its run of pieces); the default is the whole file.
static bool upload_mapped(int fd, uint64_t bytes, const PieceSink &sink, hipStream_t s, const char *path, tlm_ctx *defer_to,
                          uint64_t k0 = 0, uint64_t k1 = ~0ull) {
    bool uploaded = false;
    {
        const auto tr = std::chrono::steady_clock::now();
        const uint64_t PIECE = text_piece_bytes();
        const uint64_t n_all = (bytes + PIECE - 1) / PIECE;
        k1 = std::min(k1, n_all);
        k0 = std::min(k0, k1);
        if (k0 == k1) return true;                  // nothing of this file is ours to bring
        // only the run's bytes are mapped (the mapping's offset is a multiple of the piece size, hence of the page size);
        // `m` is biased so that m[b] is byte b of the FILE for every b of the run
        const uint64_t map_lo = k0 * PIECE, map_bytes = std::min<uint64_t>(bytes, k1 * PIECE) - map_lo;
        char *m_run = (char *) ::mmap(nullptr, map_bytes, PROT_READ, MAP_SHARED, fd, (off_t) map_lo);
        if (m_run != (char *) MAP_FAILED) {
            char *m = m_run - map_lo;
            const unsigned workers = std::max(1u, std::min(tlm_cpu_share(), 32u));
            const uint64_t n_pieces = k1;           // (loops below run k0 <= k < n_pieces; arrays are indexed by k)
            auto fault_in = [&](uint64_t k) {        // touch one byte of every page of piece k, all cores
                const uint64_t b0 = k * PIECE, b1 = std::min(bytes, b0 + PIECE);
                std::atomic<uint64_t> next{b0};
                auto work = [&] {
                    unsigned sink = 0;
                    for (uint64_t a = next.fetch_add(4u << 20); a < b1; a = next.fetch_add(4u << 20))
                        for (uint64_t q = a; q < std::min(b1, a + (4u << 20)); q += 4096) sink += (unsigned char) m[q];
                    asm volatile("" ::"r"(sink));
                };
                std::vector<std::thread> th;
                for (unsigned t = 1; t < workers; t++) th.emplace_back(work);
                work();
                for (auto &x : th) x.join();
            };
            bool ok = true;
            double t_fault = 0, t_reg = 0;
            std::vector<hipEvent_t> done(n_pieces, nullptr);
            uint64_t registered = k0;               // pieces [k0, registered) are registered with the device
            std::thread ahead;                      // faults of the next piece
            auto t0f = std::chrono::steady_clock::now();
            fault_in(k0);
            t_fault += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0f).count();
            uint64_t told = k0;                     // pieces [k0, told) have been reported as arrived
            auto report = [&](uint64_t upto, bool wait) {       // pieces < upto whose copies are complete
                while (ok && told < upto && done[told]) {
                    if (wait) { if (hipEventSynchronize(done[told]) != hipSuccess) ok = false; }
                    else if (hipEventQuery(done[told]) != hipSuccess) { (void) hipGetLastError(); break; }
                    if (ok && sink.arrived && !sink.arrived(told)) ok = false;
                    told++;
                }
            };
            for (uint64_t k = k0; ok && k < n_pieces; k++) {
                const uint64_t b0 = k * PIECE, len = std::min(bytes, b0 + PIECE) - b0;
                if (k + 1 < n_pieces) ahead = std::thread([&, k] { fault_in(k + 1); });
                const auto t0r = std::chrono::steady_clock::now();
                ok = hipHostRegister(m + b0, len, hipHostRegisterDefault) == hipSuccess;
                t_reg += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0r).count();
                if (ok && sink.n_slots && k - k0 >= sink.n_slots) {     // the slot's previous piece must have been taken over
                    report(k - sink.n_slots + 1, true);
                    if (ok && sink.before && !sink.before(k)) ok = false;
                }
                if (ok) {
                    registered = k + 1;
                    ok = hipMemcpyAsync(sink.at(k, PIECE), m + b0, len, hipMemcpyHostToDevice, s) == hipSuccess
                         && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess
                         && hipEventRecord(done[k], s) == hipSuccess;
                }
                report(k, false);
                if (ahead.joinable()) {
                    const auto t0w = std::chrono::steady_clock::now();
                    ahead.join();
                    t_fault += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0w).count();
                }
            }
            if (!ok) (void) hipGetLastError();
            const auto t0c = std::chrono::steady_clock::now();
            (void) hipStreamSynchronize(s);
            report(n_pieces, true);
            const double t_copy = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0c).count();
            for (hipEvent_t e : done)
                if (e) (void) hipEventDestroy(e);
            // unmapping 15 GB of populated pages takes longer than the copies did (0.65 s) and holds up every allocation
            // of the process meanwhile: deferred to the context, which does it when the device has just been given a long
            // kernel (the walks), or when it goes
            for (uint64_t k = k0; k < registered; k++) (void) hipHostUnregister(m + k * PIECE);   // (instant while the device idles)
            if (defer_to) tlm_ctx_defer(defer_to, [m_run, map_bytes] {
                const auto t1 = std::chrono::steady_clock::now();
                // 0.55 s for the 3.8 M page-table entries of a 15.6 GB file, one core, whatever the thread count.  Two
                // things were tried against it (gpurun_out/r02_child): dropping the entries 64 MB at a time
                // (MADV_DONTNEED) so that the address-space lock is free in between -- every call makes the driver stop
                // and restart the process's device queues (the range had been registered with it), the walk kernels took
                // 1.6-2.5 s instead of 0.78 s; and not mapping at all (all cores copy the file into a ring of pinned
                // slabs, DMA per slab) -- the cores of a one-GPU box copy at 11 GB/s, 1.4 s against the 0.31 s of the
                // registered mapping.  One munmap it stays; in about one run of four it stops the queues once, for as
                // long as it takes (walk kernels 1.33-1.37 s), and the wall-clock is the same: the host waits either way.
                ::munmap(m_run, map_bytes);
                if (getenv("TLM_HOST_TIMING"))
                    fprintf(stderr, "[host timing] behind the scenes: unmapping %.0f ms\n",
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
            });
            uploaded = ok;
            if (getenv("TLM_HOST_TIMING"))
                fprintf(stderr, "[host timing] text upload of %s (%s): %u workers, waiting for page faults %.0f ms, pinning %.0f ms, "
                                "waiting for the copies %.0f ms, all %.0f ms\n", path, ok ? "mapped + pinned" : "FAILED, falling back",
                        workers, t_fault, t_reg, t_copy,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tr).count());
        }
    }
    return uploaded;
}

extern "C" int tlm_paf_add_file(tlm_paf *p, const char *path, int anchors, uint64_t *n_records, int *clean) {
    TLM_CHECK(p && path && n_records && clean, TLM_EINVAL, "tlm_paf_add_file: null argument");
    TLM_CHECK(!p->poisoned, TLM_EINVAL, "tlm_paf_add_file: an earlier file of this handle was refused; destroy the handle and start over");
    TLM_CHECK(!p->finished && p->n_files < MAX_FILES, TLM_EINVAL, "tlm_paf_add_file: at most %d files, before tlm_paf_finish", MAX_FILES);
    TLM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const int fd = ::open(path, O_RDONLY);
    TLM_CHECK(fd >= 0, TLM_EINVAL, "Cannot open file: %s", path);
    struct stat st;
    ::fstat(fd, &st);
    const uint64_t bytes = (uint64_t) st.st_size;
    TLM_CHECK(bytes < (1ull << 46), TLM_ELIMIT, "tlm_paf_add_file: file too large");
    const int f = p->n_files;
    const auto t0 = std::chrono::steady_clock::now();
    int rc = tlm_pool_alloc((void **) &p->text[f], bytes + 64, s);
    // the sharded load (ABI 5): this rank brings its run of pieces only, the ranks swap their runs below
    const bool sharded = sharded_load(p);
    const uint64_t piece = text_piece_bytes(), n_pieces_all = (bytes + piece - 1) / piece;
    uint64_t k0 = 0, k1 = n_pieces_all;
    if (sharded) {          // (tlm_dist_text_pieces, with this build's piece)
        k0 = n_pieces_all * (uint64_t) p->ctx->rank / (uint64_t) p->ctx->world;
        k1 = n_pieces_all * ((uint64_t) p->ctx->rank + 1) / (uint64_t) p->ctx->world;
    }
    const uint64_t run_lo = std::min<uint64_t>(bytes, k0 * piece), run_hi = std::min<uint64_t>(bytes, k1 * piece);
    bool uploaded = false;
    if (rc == TLM_OK && bytes && !getenv("TLM_UPLOAD_SLABS")) {
        PieceSink sink;
        sink.base = p->text[f];
        uploaded = upload_mapped(fd, bytes, sink, s, path, p->ctx, k0, k1);
    }
    if (rc == TLM_OK && run_hi > run_lo && !uploaded) {
        // pinned slabs filled by parallel preads (the page cache copies at memory speed), H2D copies behind them:
        // the readers of slab k+1 work while slab k crosses PCIe
        constexpr size_t PIECE = 4u << 20;
        constexpr int N_SLABS = 3;
        const size_t SLAB = (size_t) std::min<uint64_t>(256u << 20, std::max<uint64_t>(PIECE, (bytes + 2) / 3));
        const unsigned readers = std::max(1u, std::min(tlm_cpu_share(), 32u));
        char *slab[N_SLABS] = {nullptr, nullptr, nullptr};
        hipEvent_t done[N_SLABS] = {nullptr, nullptr, nullptr};
        bool ok = true;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
        double t_pin = 0, t_wait = 0, t_read = 0;
        auto tp = now();
        for (int k = 0; k < N_SLABS && ok; k++)
            ok = hipHostMalloc((void **) &slab[k], SLAB, hipHostMallocDefault) == hipSuccess
                 && hipEventCreateWithFlags(&done[k], hipEventDisableTiming) == hipSuccess;
        t_pin = ms_since(tp);
        uint64_t at = run_lo;
        for (int k = 0; ok && at < run_hi; k = (k + 1) % N_SLABS) {
            const size_t len = (size_t) std::min<uint64_t>(SLAB, run_hi - at);
            tp = now();
            ok = hipEventSynchronize(done[k]) == hipSuccess;        // the slab's previous copy has left it
            t_wait += ms_since(tp);
            tp = now();
            std::atomic<size_t> next{0};
            std::atomic<bool> read_ok{true};
            auto work = [&] {
                for (size_t a = next.fetch_add(PIECE); a < len; a = next.fetch_add(PIECE)) {
                    size_t b = std::min(len, a + PIECE);
                    while (a < b) {
                        const ssize_t got = ::pread(fd, slab[k] + a, b - a, (off_t) (at + a));
                        if (got <= 0) { read_ok = false; return; }
                        a += (size_t) got;
                    }
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < readers; t++) th.emplace_back(work);
            work();
            for (auto &x : th) x.join();
            t_read += ms_since(tp);
            ok = ok && read_ok.load()
                 && hipMemcpyAsync(p->text[f] + at, slab[k], len, hipMemcpyHostToDevice, s) == hipSuccess
                 && hipEventRecord(done[k], s) == hipSuccess;
            at += len;
        }
        tp = now();
        ok = ok && hipStreamSynchronize(s) == hipSuccess;
        t_wait += ms_since(tp);
        tp = now();
        for (int k = 0; k < N_SLABS; k++) {
            if (done[k]) (void) hipEventDestroy(done[k]);
            if (slab[k]) (void) hipHostFree(slab[k]);
        }
        t_pin += ms_since(tp);
        if (getenv("TLM_HOST_TIMING"))
            fprintf(stderr, "[host timing] text upload of %s: %u readers, pinned slabs %.0f ms, reads %.0f ms, waiting for copies "
                            "%.0f ms, device allocation %.0f ms\n", path, readers, t_pin, t_read, t_wait,
                    std::chrono::duration<double, std::milli>(tp - t0).count() - t_pin - t_read - t_wait);
        if (!ok) {
            tlm_set_error("tlm_paf_add_file: reading or uploading %s failed", path);
            rc = TLM_EHIP;
        }
    }
    ::close(fd);
    if (sharded) {          // COLLECTIVE, whatever this rank's upload came to: a failure here takes every rank out together
        tlm_exchange_info x{};
        rc = tlm_text_share(p->ctx, p->text[f], bytes, piece, rc, &x);
        add_share(p, x);
        if (rc != TLM_OK) p->poisoned = true;
        if (getenv("TLM_HOST_TIMING"))
            fprintf(stderr, "[host timing] sharded load of %s: rank %d of %d brought pieces [%llu, %llu) of %llu, swap %.0f ms (%.2f GB in)\n",
                    path, p->ctx->rank, p->ctx->world, (unsigned long long) k0, (unsigned long long) k1,
                    (unsigned long long) n_pieces_all, x.wall_ms, x.bytes_received / 1e9);
    }
    if (rc != TLM_OK) return rc;
    p->upload_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    p->bytes[f] = bytes;
    p->anchors[f] = anchors;
    p->n_files = f + 1;
    rc = tokenise_on_device(p, f, n_records, clean);
    if (rc != TLM_OK || !*clean) { p->n_files = f; p->poisoned = true; }       // nothing usable was appended (the buffers go with the handle)
    return rc;
}

// ---- the text of a file brought to the device by ANOTHER process (an upload helper): the file's mapping, its
// registration with the device and the taking down of both then belong to that process -- unmapping 15.6 GB costs 0.55 s
// of one core under the address-space lock and, in the process that does it, disturbs the device queues.
// The two processes share a small WINDOW of device memory (TLM_UPLOAD_SLOTS slots of 1 GiB, exported once through HIP IPC:
// exporting the 61 GB text buffer of the chr1-shaped input itself took seconds): the helper DMAs piece k of the file
// into slot k mod slots and says so, this process copies the slot to its place in the text buffer (device to device) and
// gives the slot back.
extern "C" int tlm_upload_window_create(tlm_ctx *ctx, uint8_t *handle_out, uint64_t *bytes_out) {
    TLM_CHECK(ctx && handle_out && bytes_out, TLM_EINVAL, "tlm_upload_window_create: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= TLM_IPC_HANDLE_BYTES, "TLM_IPC_HANDLE_BYTES too small");
    TLM_HIP(hipSetDevice(ctx->device));
    if (!ctx->upload_window) TLM_HIP(hipMalloc(&ctx->upload_window, (size_t) TLM_UPLOAD_SLOTS << 30));     // (its own driver allocation)
    hipIpcMemHandle_t h;
    TLM_HIP(hipIpcGetMemHandle(&h, ctx->upload_window));
    std::memset(handle_out, 0, TLM_IPC_HANDLE_BYTES);
    std::memcpy(handle_out, &h, sizeof(h));
    *bytes_out = (uint64_t) TLM_UPLOAD_SLOTS << 30;
    return TLM_OK;
}

extern "C" int tlm_paf_text_begin(tlm_paf *p, uint64_t bytes) {
    TLM_CHECK(p, TLM_EINVAL, "tlm_paf_text_begin: null argument");
    TLM_CHECK(!p->poisoned, TLM_EINVAL, "tlm_paf_text_begin: an earlier file of this handle was refused; destroy the handle and start over");
    TLM_CHECK(!p->finished && p->n_files < MAX_FILES && !p->pending, TLM_EINVAL, "tlm_paf_text_begin: no room for a file");
    TLM_CHECK(bytes < (1ull << 46), TLM_ELIMIT, "tlm_paf_text_begin: text too large");
    TLM_HIP(hipSetDevice(p->ctx->device));
    if (int rc = tlm_pool_alloc((void **) &p->pending, bytes + 64, p->ctx->stream)) return rc;
    p->pending_bytes = bytes;
    return TLM_OK;
}

extern "C" int tlm_paf_text_piece(tlm_paf *p, uint64_t piece) {
    TLM_CHECK(p && p->pending, TLM_EINVAL, "tlm_paf_text_piece: no text was begun");
    const uint64_t b0 = piece << 30;
    TLM_CHECK(p->ctx->upload_window, TLM_EINVAL, "tlm_paf_text_piece: tlm_upload_window_create has not been called");
    TLM_CHECK(b0 < p->pending_bytes, TLM_EINVAL, "tlm_paf_text_piece: piece %llu lies beyond the text", (unsigned long long) piece);
    TLM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const uint64_t len = std::min<uint64_t>(p->pending_bytes, b0 + (1ull << 30)) - b0;
    TLM_HIP(hipMemcpyAsync(p->pending + b0, (char *) p->ctx->upload_window + ((piece % TLM_UPLOAD_SLOTS) << 30), len,
                           hipMemcpyDeviceToDevice, s));
    TLM_HIP(hipStreamSynchronize(s));           // the slot may be written again
    return TLM_OK;
}

// COLLECTIVE (see the header): the pending text holds this rank's pieces; afterwards all of them
extern "C" int tlm_paf_text_share(tlm_paf *p, int failed, tlm_exchange_info *info) {
    TLM_CHECK(p && p->pending, TLM_EINVAL, "tlm_paf_text_share: no text was begun");
    tlm_exchange_info x{};
    if (failed) tlm_set_error("tlm_paf_text_share: this rank could not bring its pieces of the text to the device");
    const int rc = tlm_text_share(p->ctx, p->pending, p->pending_bytes, TLM_TEXT_PIECE, failed ? TLM_EHIP : TLM_OK, &x);
    add_share(p, x);
    if (info) *info = x;
    if (rc != TLM_OK) p->poisoned = true;
    return rc;
}

extern "C" int tlm_paf_share_info(const tlm_paf *p, tlm_exchange_info *info) {
    TLM_CHECK(p && info, TLM_EINVAL, "tlm_paf_share_info: null argument");
    *info = p->share;
    return TLM_OK;
}

extern "C" void tlm_paf_text_abort(tlm_paf *p) {
    if (!p || !p->pending) return;
    tlm_pool_free(p->pending);
    p->pending = nullptr;
}

extern "C" int tlm_paf_text_commit(tlm_paf *p, int anchors, uint64_t *n_records, int *clean) {
    TLM_CHECK(p && n_records && clean && p->pending, TLM_EINVAL, "tlm_paf_text_commit: no text was begun");
    TLM_HIP(hipSetDevice(p->ctx->device));
    const int f = p->n_files;
    p->text[f] = p->pending;
    p->pending = nullptr;
    p->bytes[f] = p->pending_bytes;
    p->anchors[f] = anchors;
    p->n_files = f + 1;
    const int rc = tokenise_on_device(p, f, n_records, clean);
    if (rc != TLM_OK || !*clean) { p->n_files = f; p->poisoned = true; }
    return rc;
}

// the helper's side: the window of the other process, and a file through it
extern "C" int tlm_upload_window_open(tlm_ctx *ctx, const uint8_t *handle) {
    TLM_CHECK(ctx && handle, TLM_EINVAL, "tlm_upload_window_open: null argument");
    TLM_HIP(hipSetDevice(ctx->device));
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    TLM_HIP(hipIpcOpenMemHandle(&ctx->peer_window, h, hipIpcMemLazyEnablePeerAccess));
    return TLM_OK;
}

extern "C" int tlm_upload_file_window(tlm_ctx *ctx, const char *path, uint64_t bytes, tlm_upload_piece_fn arrived,
                                      tlm_upload_piece_fn slot_free, void *user) {
    return tlm_upload_file_window_range(ctx, path, bytes, 0, ~0ull, arrived, slot_free, user);
}

extern "C" int tlm_upload_file_window_range(tlm_ctx *ctx, const char *path, uint64_t bytes, uint64_t piece_begin, uint64_t piece_end,
                                            tlm_upload_piece_fn arrived, tlm_upload_piece_fn slot_free, void *user) {
    TLM_CHECK(ctx && path && arrived && slot_free, TLM_EINVAL, "tlm_upload_file_window: null argument");
    TLM_CHECK(ctx->peer_window, TLM_EINVAL, "tlm_upload_file_window: tlm_upload_window_open has not been called");
    TLM_HIP(hipSetDevice(ctx->device));
    const int fd = ::open(path, O_RDONLY);
    struct stat st;
    bool ok = fd >= 0 && ::fstat(fd, &st) == 0 && (uint64_t) st.st_size == bytes;
    if (ok && bytes) {
        PieceSink sink;
        sink.base = (char *) ctx->peer_window;
        sink.n_slots = TLM_UPLOAD_SLOTS;
        sink.before = [&](uint64_t k) { return slot_free(k, user) == 0; };
        sink.arrived = [&](uint64_t k) { return arrived(k, user) == 0; };
        ok = upload_mapped(fd, bytes, sink, ctx->stream, path, nullptr, piece_begin, piece_end);
    }
    if (fd >= 0) ::close(fd);
    if (!ok) {
        tlm_set_error("tlm_upload_file_window: reading or uploading %s failed", path);
        return TLM_EHIP;
    }
    return TLM_OK;
}

extern "C" int tlm_paf_add_text(tlm_paf *p, const char *text_host, uint64_t bytes, int anchors, uint64_t *n_records, int *clean) {
    TLM_CHECK(p && (text_host || bytes == 0) && n_records && clean, TLM_EINVAL, "tlm_paf_add_text: null argument");
    TLM_CHECK(!p->poisoned, TLM_EINVAL, "tlm_paf_add_text: an earlier file of this handle was refused; destroy the handle and start over");
    TLM_CHECK(!p->finished && p->n_files < MAX_FILES, TLM_EINVAL, "tlm_paf_add_text: at most %d files, before tlm_paf_finish", MAX_FILES);
    TLM_CHECK(bytes < (1ull << 46), TLM_ELIMIT, "tlm_paf_add_text: text too large");
    TLM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const int f = p->n_files;
    if (int rc = tlm_pool_alloc((void **) &p->text[f], bytes + 64, s)) return rc;
    if (bytes) TLM_HIP(hipMemcpyAsync(p->text[f], text_host, bytes, hipMemcpyHostToDevice, s));
    TLM_HIP(hipStreamSynchronize(s));
    p->bytes[f] = bytes;
    p->anchors[f] = anchors;
    p->n_files = f + 1;
    const int rc = tokenise_on_device(p, f, n_records, clean);
    if (rc != TLM_OK || !*clean) { p->n_files = f; p->poisoned = true; }
    return rc;
}

extern "C" int tlm_paf_finish(tlm_paf *p, uint32_t *n_names_out, uint64_t *blob_bytes_out) {
    TLM_CHECK(p && n_names_out, TLM_EINVAL, "tlm_paf_finish: null argument");
    TLM_CHECK(!p->finished, TLM_EINVAL, "tlm_paf_finish: called twice");
    TLM_CHECK(!p->poisoned, TLM_EINVAL, "tlm_paf_finish: a file of this handle was refused; its names would be numbered with the others");
    TLM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t n = 0;
    if (p->slot) {
        TLM_HIP(hipMemcpyAsync(&n, p->n_names_dev, 4, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
    }
    TLM_CHECK((uint64_t) n * 2 <= p->n_slots || !p->slot, TLM_EINVAL, "tlm_paf_finish: the dictionary of a refused parse");
    p->n_names = n;
    if (int rc = tlm_pool_alloc((void **) &p->is_ctg, n ? n : 1, s)) return rc;
    if (int rc = tlm_pool_alloc((void **) &p->name_off, ((size_t) n + 1) * 4, s)) return rc;
    TLM_HIP(hipMemsetAsync(p->name_off, 0, ((size_t) n + 1) * 4, s));
    if (n) {
        Tmp keys, vals, keys2, vals2, sort_tmp, id_of_slot, lens, scan_tmp;
        if (int rc = tlm_pool_alloc(&keys.p, (size_t) n * 8, s)) return rc;
        if (int rc = tlm_pool_alloc(&vals.p, (size_t) n * 4, s)) return rc;
        if (int rc = tlm_pool_alloc(&keys2.p, (size_t) n * 8, s)) return rc;
        if (int rc = tlm_pool_alloc(&vals2.p, (size_t) n * 4, s)) return rc;
        if (int rc = tlm_pool_alloc(&id_of_slot.p, p->n_slots * 4, s)) return rc;
        if (int rc = tlm_pool_alloc(&lens.p, ((size_t) n + 1) * 4, s)) return rc;
        hipLaunchKernelGGL(collect_slots_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p->slot, (const uint32_t *) p->inserted, n,
                           (unsigned long long *) keys.p, (uint32_t *) vals.p);
        size_t tb = 0;
        TLM_HIP(rocprim::radix_sort_pairs(nullptr, tb, (unsigned long long *) keys.p, (unsigned long long *) keys2.p,
                                          (uint32_t *) vals.p, (uint32_t *) vals2.p, (size_t) n, 0, 64, s));
        if (int rc = tlm_pool_alloc(&sort_tmp.p, tb, s)) return rc;
        TLM_HIP(rocprim::radix_sort_pairs(sort_tmp.p, tb, (unsigned long long *) keys.p, (unsigned long long *) keys2.p,
                                          (uint32_t *) vals.p, (uint32_t *) vals2.p, (size_t) n, 0, 64, s));
        Texts tx;
        for (int k = 0; k < MAX_FILES; k++) tx.p[k] = p->text[k];
        TLM_HIP(hipMemsetAsync(lens.p, 0, ((size_t) n + 1) * 4, s));
        hipLaunchKernelGGL(names_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, (const unsigned long long *) keys2.p,
                           (const uint32_t *) vals2.p, tx, (uint32_t *) id_of_slot.p, (uint32_t *) lens.p, p->is_ctg);
        size_t tb2 = 0;
        TLM_HIP(rocprim::exclusive_scan(nullptr, tb2, (const uint32_t *) lens.p, p->name_off, 0u, (size_t) n + 1,
                                        rocprim::plus<uint32_t>(), s));
        if (int rc = tlm_pool_alloc(&scan_tmp.p, tb2, s)) return rc;
        TLM_HIP(rocprim::exclusive_scan(scan_tmp.p, tb2, (const uint32_t *) lens.p, p->name_off, 0u, (size_t) n + 1,
                                        rocprim::plus<uint32_t>(), s));
        uint32_t total = 0;
        TLM_HIP(hipMemcpyAsync(&total, p->name_off + n, 4, hipMemcpyDeviceToHost, s));
        TLM_HIP(hipStreamSynchronize(s));
        p->blob_bytes = total;
        if (int rc = tlm_pool_alloc((void **) &p->blob, total ? total : 1, s)) return rc;
        hipLaunchKernelGGL(blob_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, (const unsigned long long *) keys2.p,
                           (const uint32_t *) p->name_off, tx, p->blob);
        for (int f = 0; f < p->n_files; f++)
            if (p->n[f])
                hipLaunchKernelGGL(remap_kernel, dim3(grid_of(p->n[f])), dim3(256), 0, s, p->n[f], p->cols[f].q_id, p->cols[f].t_id,
                                   (const uint32_t *) id_of_slot.p);
        TLM_HIP(hipStreamSynchronize(s));
        TLM_HIP(hipGetLastError());
    }
    // the text is no longer needed
    for (int f = 0; f < MAX_FILES; f++) {
        tlm_pool_free(p->text[f]);
        p->text[f] = nullptr;
    }
    tlm_pool_free(p->slot);
    tlm_pool_free(p->inserted);
    p->slot = nullptr;
    p->inserted = nullptr;
    p->finished = true;
    p->finish_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *n_names_out = n;
    if (blob_bytes_out) *blob_bytes_out = p->blob_bytes;
    return TLM_OK;
}

extern "C" int tlm_paf_names(const tlm_paf *p, uint32_t *name_off_host, char *blob_host, uint8_t *is_ctg_host) {
    TLM_CHECK(p && p->finished, TLM_EINVAL, "tlm_paf_names: call tlm_paf_finish first");
    TLM_HIP(hipSetDevice(p->ctx->device));
    hipStream_t s = p->ctx->stream;
    // (the caller's buffers are pageable: pinned for the duration of the copy, a copy of the 12 MB name blob runs at the
    // link's rate instead of being staged through the runtime's bounce buffers -- 17 ms -> a few)
    struct Pinned {
        void *p = nullptr;
        Pinned(void *ptr, size_t bytes) {
            if (ptr && bytes >= (1u << 20) && hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess) p = ptr;
            else (void) hipGetLastError();
        }
        ~Pinned() { if (p) (void) hipHostUnregister(p); }
    };
    const Pinned pin_off(name_off_host, ((size_t) p->n_names + 1) * 4), pin_blob(blob_host, p->blob_bytes);
    if (name_off_host) TLM_HIP(hipMemcpyAsync(name_off_host, p->name_off, ((size_t) p->n_names + 1) * 4, hipMemcpyDeviceToHost, s));
    if (blob_host && p->blob_bytes) TLM_HIP(hipMemcpyAsync(blob_host, p->blob, p->blob_bytes, hipMemcpyDeviceToHost, s));
    if (is_ctg_host && p->n_names) TLM_HIP(hipMemcpyAsync(is_ctg_host, p->is_ctg, p->n_names, hipMemcpyDeviceToHost, s));
    TLM_HIP(hipStreamSynchronize(s));
    return TLM_OK;
}

extern "C" int tlm_paf_records(const tlm_paf *p, int file, tlm_records *out) {
    TLM_CHECK(p && out && p->finished && file >= 0 && file < p->n_files, TLM_EINVAL, "tlm_paf_records: bad argument");
    const Columns &c = p->cols[file];
    out->n = p->n[file];
    out->q_id = c.q_id; out->t_id = c.t_id; out->q_len = c.q_len; out->q_start = c.q_start; out->q_end = c.q_end;
    out->t_len = c.t_len; out->t_start = c.t_start; out->t_end = c.t_end; out->matches = c.matches;
    out->strand = c.strand;
    out->anchors = p->anchors[file];
    return TLM_OK;
}

extern "C" int tlm_paf_timing(const tlm_paf *p, double *upload_ms, double *parse_ms, double *finish_ms) {
    TLM_CHECK(p, TLM_EINVAL, "tlm_paf_timing: null argument");
    if (upload_ms) *upload_ms = p->upload_ms;
    if (parse_ms) *parse_ms = p->parse_ms;
    if (finish_ms) *finish_ms = p->finish_ms;
    return TLM_OK;
}

extern "C" int tlm_paf_build_graph(tlm_paf *p, const tlm_filter_params *fp, tlm_graph **out) {
    TLM_CHECK(p && fp && out && p->finished && p->n_files >= 1, TLM_EINVAL, "tlm_paf_build_graph: finish the files first");
    tlm_records recs[MAX_FILES];
    for (int f = 0; f < p->n_files; f++)
        if (int rc = tlm_paf_records(p, f, &recs[f])) return rc;
    return tlm_graph_build(p->ctx, fp, recs, p->n_files, p->n_names, p->is_ctg, out);
}
