// telomeri_amd/host/overlap_graph.cpp -- OverlapGraph over the C ABI.
//
// The host keeps what is inherently host work in the reference's load (OverlapGraph.cpp:97-154): reading
// the file, tokenising with operator>> semantics, and giving names ids.  Filtering, scoring, node
// numbering and adjacency construction run on the GPU (tlm_graph_build_host); the result is downloaded
// into the public nodes_ / edges_ so that every consumer of the reference's data members keeps working.
#include <mutex>
#include <spawn.h>
#include <signal.h>
#include "telomeri_host.hpp"

extern char **environ;

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <functional>
#include <memory>
#include <thread>
#include <unordered_map>
#include <fcntl.h>
#include <sched.h>
#include <sys/resource.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/telomeri_hip.h"
#include "../csrc/cpu_share.h"

namespace {

[[noreturn]] void die(const char *what) {
    std::cerr << "telomeri_hip: " << what << ": " << tlm_last_error() << std::endl;
    std::exit(1);
}

// --- operator>> on a memory buffer --------------------------------------------------------------
struct Reader {
    const char *p, *end;
    static bool ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }
    void skip() { while (p < end && ws(*p)) p++; }
    bool token(const char *&s, size_t &n) {          // >> std::string
        skip();
        if (p >= end) return false;
        s = p;
        while (p < end && !ws(*p)) p++;
        n = (size_t) (p - s);
        return true;
    }
    bool integer(int &v) {                            // >> int: [+-]digits; overflow sets failbit
        skip();
        if (p >= end) return false;
        const char *q = p;
        bool neg = false;
        if (*q == '+' || *q == '-') { neg = *q == '-'; q++; }
        const char *d0 = q;
        long long acc = 0;
        bool ovf = false;
        while (q < end && *q >= '0' && *q <= '9') {
            if (!ovf) { acc = acc * 10 + (*q - '0'); ovf = acc > (1ll << 32); }
            q++;
        }
        if (q == d0) return false;
        p = q;
        if (neg) acc = -acc;
        if (ovf || acc > INT_MAX || acc < INT_MIN) return false;
        v = (int) acc;
        return true;
    }
    bool character(char &c) {                         // >> char
        skip();
        if (p >= end) return false;
        c = *p++;
        return true;
    }
};

}   // namespace

// One process per GPU.  A multi-GPU job is N copies of this program started with
//   TLM_WORLD=N TLM_RANK=r (or torchrun's WORLD_SIZE / RANK / LOCAL_RANK) and TLM_RENDEZVOUS=<path on a shared
//   filesystem>: rank 0 writes the 128-byte RCCL id there, the others wait for the file.
// Device: TLM_DEVICE, else LOCAL_RANK, else the rank, else 0.
static int env_int(const char *a, const char *b, int dflt) {
    const char *v = std::getenv(a);
    if (!v && b) v = std::getenv(b);
    return v ? std::atoi(v) : dflt;
}

int OverlapGraph::device_ordinal() {
    const int world = env_int("TLM_WORLD", "WORLD_SIZE", 1), rank = env_int("TLM_RANK", "RANK", 0);
    return env_int("TLM_DEVICE", "LOCAL_RANK", world > 1 ? rank : 0);
}

// ---- the upload helper (a second process; see start_upload_helper in the header) --------------------------------
// Protocol (lines): "G" -> the helper creates its context and says "READY".  "W <ipc handle, hex>" -> "OK": the window of device
// memory both use.  "U <bytes> <path>" (or "R <bytes> <first piece> <end piece> <path>": a run of pieces, the sharded load of
// a job of several ranks) -> "P <k>" for every 1 GiB piece that has arrived in its slot, answered with "A" once
// the piece has been copied to its place when the slot will be needed again, then "OK".  "F <path>" -> "OK <index file>
// <ms>": a sequence file scanned (answers collected later).  "ERR <message>" to anything that failed.
namespace {
struct UploadHelper {
    pid_t pid = -1;
    FILE *to = nullptr, *from = nullptr;
    bool ready = false;
    int pending_replies = 0;            // FASTA index requests whose answers have not been collected
    bool window = false;                // the helper has opened this process's upload window
    bool told_to_go = false;            // "G" has been sent
} g_helper;

bool helper_wait_ready();
bool helper_go();

// the window of device memory the two processes share: created here, opened by the helper (once)
bool helper_window() {
    if (g_helper.window) return true;
    uint8_t handle[TLM_IPC_HANDLE_BYTES];
    uint64_t bytes = 0;
    if (tlm_upload_window_create(OverlapGraph::context(), handle, &bytes) != TLM_OK) return false;
    std::string req = "W ";
    char b[3];
    for (int i = 0; i < TLM_IPC_HANDLE_BYTES; i++) { std::snprintf(b, sizeof(b), "%02x", handle[i]); req += b; }
    req += "\n";
    char reply[512] = "";
    if (std::fputs(req.c_str(), g_helper.to) < 0 || std::fflush(g_helper.to) != 0 || !std::fgets(reply, sizeof(reply), g_helper.from)
        || std::strncmp(reply, "OK", 2) != 0) {
        OverlapGraph::stop_upload_helper();
        return false;
    }
    g_helper.window = true;
    return true;
}

// "G": the helper may create its context now.  For large inputs not before this process has obtained its text buffer:
// while a second process was coming up on the device, the 61 GB buffer of the chr1-shaped input took 5.3 s to obtain
// instead of 11 ms (start_upload_helper sends it at once when the overlap files are small: 0.1 s less to wait for).
bool helper_go() {
    if (g_helper.pid < 0) return false;
    if (g_helper.told_to_go) return true;
    if (std::fputs("G\n", g_helper.to) < 0 || std::fflush(g_helper.to) != 0) {
        OverlapGraph::stop_upload_helper();
        return false;
    }
    g_helper.told_to_go = true;
    return true;
}

bool helper_wait_ready() {
    if (g_helper.pid < 0) return false;
    if (g_helper.ready) return true;
    if (!helper_go()) return false;
    char line[256];
    if (!std::fgets(line, sizeof(line), g_helper.from) || std::strncmp(line, "READY", 5) != 0) {
        OverlapGraph::stop_upload_helper();
        return false;
    }
    g_helper.ready = true;
    return true;
}
}   // namespace

void OverlapGraph::start_upload_helper(const char *exe, bool context_at_once) {
    if (g_helper.pid >= 0 || std::getenv("TLM_NO_UPLOAD_HELPER")) return;
    int down[2], up[2];
    if (::pipe(down) != 0) return;
    if (::pipe(up) != 0) { ::close(down[0]); ::close(down[1]); return; }
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, down[0], 0);
    posix_spawn_file_actions_adddup2(&fa, up[1], 1);
    // (not this process's stderr: whoever waits for the end of our output would wait for the helper's exit as well)
    posix_spawn_file_actions_addopen(&fa, 2, "/dev/null", O_WRONLY, 0);
    posix_spawn_file_actions_addclose(&fa, down[1]);
    posix_spawn_file_actions_addclose(&fa, up[0]);
    const std::string dev = std::to_string(device_ordinal());
    ::setenv("TLM_UPLOAD_DEVICE", dev.c_str(), 1);
    char arg0[] = "telomeri_hip", arg1[] = "--upload-helper";
    char *argv[] = {arg0, arg1, nullptr};
    pid_t pid = -1;
    const int rc = ::posix_spawn(&pid, exe, &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    ::close(down[0]);
    ::close(up[1]);
    if (rc != 0) { ::close(down[1]); ::close(up[0]); return; }
    g_helper.pid = pid;
    if (std::getenv("TLM_HOST_TIMING")) std::fprintf(stderr, "[host timing] upload helper process %ld\n", (long) pid);
    g_helper.to = ::fdopen(down[1], "w");
    g_helper.from = ::fdopen(up[0], "r");
    ::signal(SIGPIPE, SIG_IGN);         // a helper that died is a failed upload, not the end of this process
    if (context_at_once) helper_go();
}

// (the program's main thread lets the helper go while another thread may be waiting for its last answers)
static std::mutex g_helper_mu;

static void stop_upload_helper_locked() {
    if (g_helper.pid < 0) return;
    if (g_helper.to) std::fclose(g_helper.to);          // end of input: the helper leaves (and takes its mappings down) on its own
    g_helper.to = nullptr;
    if (g_helper.pending_replies > 0) return;           // answers to queued requests are still to come (collect_fasta_index)
    if (g_helper.from) std::fclose(g_helper.from);
    g_helper = UploadHelper();
}

void OverlapGraph::stop_upload_helper() {
    std::lock_guard<std::mutex> lk(g_helper_mu);
    stop_upload_helper_locked();
}

bool OverlapGraph::request_fasta_index(const char *filename) {
    if (!helper_wait_ready() || std::strchr(filename, '\n')) return false;
    std::lock_guard<std::mutex> lk(g_helper_mu);
    if (!g_helper.to) return false;
    const std::string req = std::string("F ") + filename + "\n";
    if (std::fputs(req.c_str(), g_helper.to) < 0 || std::fflush(g_helper.to) != 0) return false;
    g_helper.pending_replies++;
    return true;
}

std::string OverlapGraph::collect_fasta_index() {
    FILE *from = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_helper_mu);
        if (g_helper.pid < 0 || g_helper.pending_replies <= 0 || !g_helper.from) return std::string();
        from = g_helper.from;           // (stays open while an answer is pending)
    }
    char reply[4200] = "";
    const bool got = std::fgets(reply, sizeof(reply), from) != nullptr;
    std::lock_guard<std::mutex> lk(g_helper_mu);
    g_helper.pending_replies--;
    std::string name;
    if (got && std::strncmp(reply, "OK ", 3) == 0) {
        name = reply + 3;
        while (!name.empty() && (name.back() == '\n' || name.back() == '\r')) name.pop_back();
        const size_t tab = name.find('\t');
        if (tab != std::string::npos) {
            if (std::getenv("TLM_HOST_TIMING"))
                std::fprintf(stderr, "[host timing] helper process: sequence file scanned and indexed in %s ms\n", name.c_str() + tab + 1);
            name.resize(tab);
        }
    }
    if (g_helper.pending_replies == 0 && !g_helper.to) stop_upload_helper_locked();
    return name;
}

int OverlapGraph::upload_helper_main() {
    {   // nothing of the device is touched before the other process says so (see helper_wait_ready)
        char go[16];
        if (!std::fgets(go, sizeof(go), stdin) || go[0] != 'G') std::_Exit(0);
    }
    tlm_ctx *ctx = nullptr;
    const char *dev = std::getenv("TLM_UPLOAD_DEVICE");
    if (tlm_ctx_create(dev ? std::atoi(dev) : 0, &ctx) != TLM_OK) {
        std::printf("ERR %s\n", tlm_last_error());
        std::fflush(stdout);
        return 1;
    }
    std::printf("READY\n");
    std::fflush(stdout);
    std::string line;
    for (int c; (c = std::getchar()) != EOF;) {
        if (c != '\n') { line.push_back((char) c); continue; }
        unsigned long long bytes = 0, piece0 = 0, piece1 = ~0ull;
        char hex[2 * TLM_IPC_HANDLE_BYTES + 1];
        int used = 0;
        if (std::sscanf(line.c_str(), "W %128s", hex) == 1 && std::strlen(hex) == 2 * TLM_IPC_HANDLE_BYTES) {
            uint8_t handle[TLM_IPC_HANDLE_BYTES];
            for (int i = 0; i < TLM_IPC_HANDLE_BYTES; i++) {
                unsigned v = 0;
                std::sscanf(hex + 2 * i, "%2x", &v);
                handle[i] = (uint8_t) v;
            }
            if (tlm_upload_window_open(ctx, handle) == TLM_OK) std::printf("OK\n");
            else std::printf("ERR %s\n", tlm_last_error());
        } else if ((std::sscanf(line.c_str(), "U %llu %n", &bytes, &used) == 1 && used > 0)
                   || (std::sscanf(line.c_str(), "R %llu %llu %llu %n", &bytes, &piece0, &piece1, &used) == 3 && used > 0)) {
            const char *path = line.c_str() + used;         // ("R": pieces [piece0, piece1) only -- the sharded load)
            // piece k is in its slot: say so; before a slot is written again: wait for the word that its piece was taken over
            auto arrived = [](uint64_t k, void *) -> int {
                std::printf("P %llu\n", (unsigned long long) k);
                return std::fflush(stdout) == 0 ? 0 : 1;
            };
            auto slot_free = [](uint64_t, void *) -> int {
                char ack[16];
                return std::fgets(ack, sizeof(ack), stdin) && ack[0] == 'A' ? 0 : 1;
            };
            if (std::getenv("TLM_HELPER_FAIL")) std::printf("ERR refused (TLM_HELPER_FAIL: tests of the fallback)\n");
            else if (tlm_upload_file_window_range(ctx, path, bytes, piece0, piece1, arrived, slot_free, nullptr) == TLM_OK) std::printf("OK\n");
            else std::printf("ERR %s\n", tlm_last_error());
        } else if (line.size() > 2 && line[0] == 'F' && line[1] == ' ') {
            // scan a sequence file here (the mapping and its taking down are this process's) and leave the index in a file
            static const Path no_path;
            static int serial = 0;
            // (the scan runs beside the other process's load: low priority, and fewer threads than the CPU share -- cpu_share.h)
            (void) ::setpriority(PRIO_PROCESS, 0, 19);
            Scaffolder sc(no_path);
            sc.scan_only_ = true;
            const auto t_scan = std::chrono::steady_clock::now();
            const char *dir = ::access("/dev/shm", W_OK) == 0 ? "/dev/shm" : "/tmp";
            // an unpredictable name (mkstemp reserves it; save_index creates the file itself, exclusively, so the
            // reservation is dropped first), gone again on every path that does not hand it to the other process
            std::string idx = std::string(dir) + "/tlm_index_" + std::to_string((long) ::getpid()) + "_" + std::to_string(serial++) + "_XXXXXX";
            const int reserve = ::mkstemp(&idx[0]);
            if (reserve >= 0) {
                ::close(reserve);
                ::unlink(idx.c_str());
            }
            if (reserve >= 0 && sc.load(line.c_str() + 2) && sc.save_index(idx.c_str())) {
                std::printf("OK %s\t%.0f\n", idx.c_str(),
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_scan).count());
            } else {
                if (reserve >= 0) ::unlink(idx.c_str());
                std::printf("ERR cannot index %s\n", line.c_str() + 2);
            }
        } else {
            std::printf("ERR bad request\n");
        }
        std::fflush(stdout);
        line.clear();
    }
    std::_Exit(0);      // the mappings and the context go with the process
}

tlm_ctx *OverlapGraph::context() {
    static tlm_ctx *ctx = nullptr;
    if (ctx) return ctx;
    const int world = env_int("TLM_WORLD", "WORLD_SIZE", 1), rank = env_int("TLM_RANK", "RANK", 0);
    const int device = device_ordinal();
    if (tlm_ctx_create(device, &ctx) != TLM_OK) die("no usable gfx950 device (there is no CPU fallback)");
    if (world > 1) {
        const char *rv = std::getenv("TLM_RENDEZVOUS");
        if (!rv) {
            std::cerr << "telomeri_hip: TLM_WORLD=" << world << " needs TLM_RENDEZVOUS=<file shared by the ranks>" << std::endl;
            std::exit(1);
        }
        uint8_t id[TLM_UNIQUE_ID_BYTES];
        if (rank == 0) {
            if (tlm_dist_unique_id(id) != TLM_OK) die("tlm_dist_unique_id");
            const std::string tmp = std::string(rv) + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof(id), f) != sizeof(id) || std::fclose(f) != 0 || std::rename(tmp.c_str(), rv) != 0) {
                std::perror(rv);
                std::exit(1);
            }
        } else {
            size_t got = 0;
            for (int tries = 0; tries < 6000 && got != sizeof(id); tries++) {       // up to 10 minutes
                if (FILE *f = std::fopen(rv, "rb")) {
                    got = std::fread(id, 1, sizeof(id), f);
                    std::fclose(f);
                }
                if (got != sizeof(id)) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (got != sizeof(id)) {
                std::cerr << "telomeri_hip: rank " << rank << " found no RCCL id at " << rv << std::endl;
                std::exit(1);
            }
        }
        if (tlm_dist_init(ctx, rank, world, id) != TLM_OK) die("tlm_dist_init");
    }
    return ctx;
}

// TLM_HOST_LEAN=1: a host that cannot call set_lean() / set_packed() -- the reference's own main() linked against these
// classes (INTEGRATION.md option A) -- still keeps the adjacency and the paths' node lists on the device.  Default: every
// public member is filled as the reference fills it.
static bool env_host_lean() {
    const char *e = std::getenv("TLM_HOST_LEAN");
    return e && std::atoi(e) != 0;
}
bool telomeri_host_lean_default() { return env_host_lean(); }

OverlapGraph::OverlapGraph() : lean_(env_host_lean()) {}


// ---- PAF ingestion ---------------------------------------------------------
// The reference reads the file as ONE token stream with operator>> (OverlapGraph.cpp:109-127): 12 typed tokens,
// then exactly 4 more, whatever the line structure.  Tokenising is the host's share of load and dominates its
// wall-clock once filtering/scoring/adjacency run on the GPU, so it is done on all cores:
//   * the file is mmap'ed and cut into byte ranges at line ends, one per thread;
//   * a range is "clean" when every non-empty line has exactly 16 tokens, its 9 numeric tokens are plain
//     [+-]digits inside int range and its strand token is one character.  On clean input line-wise parsing IS the
//     token stream, so the threads parse independently (names -> per-thread tables, merged in file order);
//   * anything else (ragged lines, "12abc", overflow, test_load_num_) falls back to the sequential token-stream
//     reader below, which reproduces operator>> exactly, including where the reference silently stops.
namespace {

struct Chunk {       // one thread's byte range of a file
    const char *begin = nullptr, *end = nullptr;
    size_t lines = 0;                // non-blank lines = records of a clean range (pass 1)
    size_t base = 0;                 // first record of the range in the file's columns
    size_t records = 0;              // records written (pass 2)
    std::vector<std::pair<const char *, uint32_t>> names;      // local name table, first-seen order
    std::vector<uint64_t> name_hash;                            // FNV-1a of each local name
    bool clean = true;
};

struct NameHash {
    static uint64_t fnv(const char *s, size_t n) {
        uint64_t h = 1469598103934665603ull;
        for (size_t i = 0; i < n; i++) { h ^= (unsigned char) s[i]; h *= 1099511628211ull; }
        return h;
    }
    // open addressing; an entry carries the first 16 bytes of its name, so a lookup touches one cache line and the text
    // of the file only for longer names
    struct Entry {
        uint64_t hash;
        int32_t id;         // -1 = empty
        uint32_t len;
        char head[16];
    };
    std::vector<Entry> tab;
    const std::vector<std::pair<const char *, uint32_t>> *names = nullptr;
    size_t mask = 0, used = 0;
    void init(const std::vector<std::pair<const char *, uint32_t>> *nm, size_t cap = 1 << 12) {
        names = nm;
        tab.assign(cap, Entry{0, -1, 0, {0}});
        mask = cap - 1;
        used = 0;
    }
    void grow() {
        std::vector<Entry> old;
        old.swap(tab);
        tab.assign(old.size() * 2, Entry{0, -1, 0, {0}});
        mask = tab.size() - 1;
        for (const Entry &e : old)
            if (e.id >= 0) {
                size_t j = e.hash & mask;
                while (tab[j].id >= 0) j = (j + 1) & mask;
                tab[j] = e;
            }
    }
    // returns the id of (s, n); *fresh = true if it was not present (the caller appends it to `names` first)
    int32_t find_or_add(const char *s, size_t n, int32_t next_id, bool *fresh, uint64_t *hash_out) {
        const uint64_t h = fnv(s, n);
        *hash_out = h;
        size_t j = h & mask;
        const size_t hn = n < 16 ? n : 16;
        while (tab[j].id >= 0) {
            const Entry &e = tab[j];
            if (e.hash == h && e.len == n && std::memcmp(e.head, s, hn) == 0
                && (n <= 16 || std::memcmp((*names)[(size_t) e.id].first + 16, s + 16, n - 16) == 0)) {
                *fresh = false;
                return e.id;
            }
            j = (j + 1) & mask;
        }
        Entry &e = tab[j];
        e.hash = h;
        e.id = next_id;
        e.len = (uint32_t) n;
        std::memcpy(e.head, s, hn);
        *fresh = true;
        if (++used * 2 > tab.size()) grow();
        return next_id;
    }
};

// whitespace inside a line, as operator>> sees it (the newline is handled by the line loop): every such character is
// <= ' ', so one compare settles the common case
inline bool is_ws(char c) {
    return (unsigned char) c <= ' ' && (c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f');
}

// strict integer: the whole token must be [+-]digits and fit an int
inline bool strict_int(const char *s, const char *e, int32_t &v) {
    if (s >= e) return false;
    bool neg = false;
    if (*s == '+' || *s == '-') { neg = *s == '-'; s++; }
    if (s >= e || e - s > 10) return false;
    long long acc = 0;
    for (; s < e; s++) {
        if (*s < '0' || *s > '9') return false;
        acc = acc * 10 + (*s - '0');
    }
    if (neg) acc = -acc;
    if (acc > INT_MAX || acc < INT_MIN) return false;
    v = (int32_t) acc;
    return true;
}

// pass 1: lines with at least one token
size_t count_records(const char *p, const char *end) {
    size_t n = 0;
    while (p < end) {
        const char *eol = (const char *) std::memchr(p, '\n', (size_t) (end - p));
        if (!eol) eol = end;
        while (p < eol && is_ws(*p)) p++;
        n += p < eol;
        p = eol + 1;
    }
    return n;
}

struct Columns {            // where a range writes its records (the file's columns, from the range's base on)
    int32_t *q_loc, *t_loc, *q_len, *q_start, *q_end, *t_len, *t_start, *t_end, *matches;
    uint8_t *strand;
};

// pass 2: record i of the range goes to out.*[i]; q_loc / t_loc hold LOCAL name ids until the merge
void parse_range(Chunk &c, const Columns &out) {
    NameHash nh;
    nh.init(&c.names);
    const char *p = c.begin, *end = c.end;
    const char *tok[16];
    const char *tend[16];
    size_t r = 0;
    const char *last_q = nullptr;       // files come grouped by query: the previous record's name is the usual hit
    size_t last_q_n = 0;
    int32_t last_q_id = -1;
    while (p < end) {
        const char *eol = (const char *) std::memchr(p, '\n', (size_t) (end - p));
        if (!eol) eol = end;
        int nt = 0;
        const char *q = p;
        while (q < eol) {
            while (q < eol && is_ws(*q)) q++;
            if (q >= eol) break;
            if (nt == 16) { nt = 17; break; }
            tok[nt] = q;
            while (q < eol && !is_ws(*q)) q++;
            tend[nt++] = q;
        }
        p = eol + 1;
        if (nt == 0) continue;                       // blank line: no tokens, the stream just moves on
        int32_t v[9];
        static const int num_idx[9] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
        bool ok = nt == 16 && tend[4] - tok[4] == 1 && r < c.lines;
        for (int k = 0; ok && k < 9; k++) ok = strict_int(tok[num_idx[k]], tend[num_idx[k]], v[k]);
        if (!ok) { c.clean = false; return; }
        bool fresh;
        uint64_t h;
        const size_t qn = (size_t) (tend[0] - tok[0]);
        int32_t qi;
        if (last_q && qn == last_q_n && std::memcmp(last_q, tok[0], qn) == 0) {
            qi = last_q_id;
        } else {
            qi = nh.find_or_add(tok[0], qn, (int32_t) c.names.size(), &fresh, &h);
            if (fresh) { c.names.emplace_back(tok[0], (uint32_t) qn); c.name_hash.push_back(h); }
            last_q = tok[0];
            last_q_n = qn;
            last_q_id = qi;
        }
        int32_t ti = nh.find_or_add(tok[5], (size_t) (tend[5] - tok[5]), (int32_t) c.names.size(), &fresh, &h);
        if (fresh) { c.names.emplace_back(tok[5], (uint32_t) (tend[5] - tok[5])); c.name_hash.push_back(h); }
        out.q_loc[r] = qi; out.t_loc[r] = ti;
        out.q_len[r] = v[0]; out.q_start[r] = v[1]; out.q_end[r] = v[2];
        out.t_len[r] = v[3]; out.t_start[r] = v[4]; out.t_end[r] = v[5];
        out.matches[r] = v[6];
        out.strand[r] = *tok[4] == '-';
        r++;
    }
    c.records = r;
}

unsigned host_threads() { return std::max(1u, std::min(tlm_cpu_share(), 64u)); }

}   // namespace

// The name table of a graph (all its files), sharded by hash so that the per-thread name tables of a parsed file are
// merged by all cores: shard s owns the names with hash % SHARDS == s.  Ids are FIRST-SEEN order over the files (query
// before target inside a record), exactly what one sequential reader assigns, whatever the number of threads.
struct OverlapGraph::NameIndex {
    static constexpr unsigned SHARDS = 64;
    struct Pending {            // a name first seen in the file being merged
        const char *p;
        uint32_t n;
        uint64_t first;         // (chunk << 32) | local index: where the file mentions it first
        int32_t id;             // global id, assigned once all shards are through
    };
    struct Shard {
        std::vector<uint64_t> hash;
        std::vector<int32_t> id;            // -1 empty, >= 0 global id, <= -2 pending number -(id + 2)
        size_t used = 0;
        std::vector<Pending> pending;
        Shard() : hash(1024, 0), id(1024, -1) {}
        void grow() {
            std::vector<uint64_t> oh;
            std::vector<int32_t> oi;
            oh.swap(hash);
            oi.swap(id);
            hash.assign(oh.size() * 2, 0);
            id.assign(oi.size() * 2, -1);
            const size_t mask = id.size() - 1;
            for (size_t i = 0; i < oi.size(); i++)
                if (oi[i] != -1) {
                    size_t j = (oh[i] / SHARDS) & mask;
                    while (id[j] != -1) j = (j + 1) & mask;
                    id[j] = oi[i];
                    hash[j] = oh[i];
                }
        }
        // slot of (s, n) -- or of the empty slot where it belongs
        size_t probe(const char *s, size_t n, uint64_t h, const std::vector<std::string> &names) const {
            const size_t mask = id.size() - 1;
            size_t j = (h / SHARDS) & mask;
            for (; id[j] != -1; j = (j + 1) & mask) {
                if (hash[j] != h) continue;
                if (id[j] >= 0) {
                    const std::string &nm = names[(size_t) id[j]];
                    if (nm.size() == n && std::memcmp(nm.data(), s, n) == 0) return j;
                } else {
                    const Pending &pd = pending[(size_t) (-(id[j] + 2))];
                    if (pd.n == n && std::memcmp(pd.p, s, n) == 0) return j;
                }
            }
            return j;
        }
    };
    Shard shards[SHARDS];

    // one name at a time (the sequential token-stream reader)
    int32_t intern(const char *s, size_t n, std::vector<std::string> &names) {
        const uint64_t h = NameHash::fnv(s, n);
        Shard &sh = shards[h % SHARDS];
        const size_t j = sh.probe(s, n, h, names);
        if (sh.id[j] >= 0) return sh.id[j];
        const int32_t gid = (int32_t) names.size();
        names.emplace_back(s, n);
        sh.id[j] = gid;
        sh.hash[j] = h;
        if (++sh.used * 2 > sh.id.size()) sh.grow();
        return gid;
    }

    // the local name tables of one file's chunks -> global ids (glob[t][i] for local name i of chunk t)
    void merge(const std::vector<Chunk> &chunks, std::vector<std::string> &names, std::vector<std::vector<int32_t>> &glob,
               unsigned threads) {
        glob.resize(chunks.size());
        for (size_t t = 0; t < chunks.size(); t++) glob[t].assign(chunks[t].names.size(), -1);
        // 1. every shard looks at the names that are its own, in file order
        std::atomic<unsigned> next{0};
        auto work = [&] {
            for (unsigned s = next++; s < SHARDS; s = next++) {
                Shard &sh = shards[s];
                for (size_t t = 0; t < chunks.size(); t++) {
                    const Chunk &c = chunks[t];
                    for (size_t i = 0; i < c.names.size(); i++) {
                        const uint64_t h = c.name_hash[i];
                        if (h % SHARDS != s) continue;
                        const size_t j = sh.probe(c.names[i].first, c.names[i].second, h, names);
                        if (sh.id[j] == -1) {
                            sh.pending.push_back(Pending{c.names[i].first, c.names[i].second, ((uint64_t) t << 32) | i, -1});
                            const int32_t provisional = -(int32_t) sh.pending.size() - 1;     // pending number k -> -(k + 2)
                            sh.id[j] = provisional;
                            sh.hash[j] = h;
                            if (++sh.used * 2 > sh.id.size()) sh.grow();
                            glob[t][i] = provisional;
                        } else {
                            glob[t][i] = sh.id[j];
                        }
                    }
                }
            }
        };
        run_threads(threads, work);
        // 2. new names in the order the file mentions them first
        std::vector<std::pair<uint64_t, std::pair<uint32_t, uint32_t>>> fresh;
        for (unsigned s = 0; s < SHARDS; s++)
            for (size_t k = 0; k < shards[s].pending.size(); k++) fresh.push_back({shards[s].pending[k].first, {s, (uint32_t) k}});
        std::sort(fresh.begin(), fresh.end());
        const size_t base = names.size();
        names.resize(base + fresh.size());
        for (size_t r = 0; r < fresh.size(); r++) shards[fresh[r].second.first].pending[fresh[r].second.second].id = (int32_t) (base + r);
        // 3. names_ entries, table slots and the chunks' maps get the final ids
        next = 0;
        auto finish = [&] {
            for (unsigned s = next++; s < SHARDS; s = next++) {
                Shard &sh = shards[s];
                for (const Pending &pd : sh.pending) names[(size_t) pd.id].assign(pd.p, pd.n);
                for (size_t j = 0; j < sh.id.size(); j++)
                    if (sh.id[j] <= -2) sh.id[j] = sh.pending[(size_t) (-(sh.id[j] + 2))].id;
            }
        };
        run_threads(threads, finish);
        next = 0;
        auto remap = [&] {
            for (unsigned t = next++; t < chunks.size(); t = next++) {
                const Chunk &c = chunks[t];
                for (size_t i = 0; i < glob[t].size(); i++)
                    if (glob[t][i] <= -2) glob[t][i] = shards[c.name_hash[i] % SHARDS].pending[(size_t) (-(glob[t][i] + 2))].id;
            }
        };
        run_threads(threads, remap);
        for (Shard &sh : shards) {
            sh.pending.clear();
            sh.pending.shrink_to_fit();
        }
    }

    template <class F> static void run_threads(unsigned n, F &f) {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < n; t++) th.emplace_back([&f] { f(); });
        f();
        for (auto &x : th) x.join();
    }
};

OverlapGraph::~OverlapGraph() {
    if (dev_) tlm_graph_destroy(dev_);
    if (paf_) tlm_paf_destroy(paf_);
    delete index_;
}

// Lean graphs in a begin_loads() / end_loads() bracket: the file's TEXT goes to the device and is tokenised there
// (tlm_paf_add_file; csrc/paf_parse.hip).  Returns false when the device path does not apply or the text is not clean PAF
// (then every file tokenised on the device so far is tokenised again on the host, so that all files share one name table).
bool OverlapGraph::tokenize_on_device(char *filepath, bool anchors, ulong &records) {
    if (!lean_ || !deferred_ || test_load_num_ != 0) return false;
    if (const char *e = std::getenv("TLM_DEVICE_PARSE")) {
        if (std::atoi(e) == 0) return false;
    }
    if (!files_.empty() && device_files_.empty()) return false;        // an earlier file already went through the host
    if (::access(filepath, R_OK) != 0) return false;                   // the host tokeniser reports it, as the reference does
    const auto t_begin = std::chrono::steady_clock::now();
    if (!paf_ && tlm_paf_create(context(), &paf_) != TLM_OK) die("tlm_paf_create");
    uint64_t n = 0;
    int clean = 0;
    bool through_helper = false;
    struct stat st;
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    const auto t_h0 = std::chrono::steady_clock::now();
    // (the text buffer first: obtaining 61 GB of device memory took 5 s once the window had been exported)
    bool helper_ok = g_helper.pid >= 0 && ::stat(filepath, &st) == 0 && st.st_size > 0 && !std::strchr(filepath, '\n')
                     && tlm_paf_text_begin(paf_, (uint64_t) st.st_size) == TLM_OK;
    const double ms_ready = since(t_h0);
    const auto t_h1 = std::chrono::steady_clock::now();
    if (helper_ok && !(helper_wait_ready() && helper_window())) {
        tlm_paf_text_abort(paf_);
        helper_ok = false;
    }
    if (helper_ok) {
        if (std::getenv("TLM_HOST_TIMING"))
            std::fprintf(stderr, "[host timing] upload helper: context + paf handle %.0f ms, text buffer %.0f ms, ready + window %.0f ms\n",
                         std::chrono::duration<double, std::milli>(t_h0 - t_begin).count(), ms_ready, since(t_h1));
        const auto t_up = std::chrono::steady_clock::now();
        const uint64_t bytes = (uint64_t) st.st_size;
        // The sharded load (a job of several ranks; include/telomeri_hip.h, ABI 5): the helper brings THIS rank's run of
        // pieces only, the ranks then swap their runs device to device (tlm_paf_text_share) -- OverlapGraph::load reads
        // the whole file on every rank (OverlapGraph.cpp:94-131), which is N uploads of the same text through one host.
        uint64_t k0 = 0, k1 = (bytes + TLM_TEXT_PIECE - 1) / TLM_TEXT_PIECE;
        const bool sharded = tlm_dist_world(context()) > 1 && !std::getenv("TLM_LOAD_REPLICATED");
        if (sharded) (void) tlm_dist_text_pieces(context(), bytes, &k0, &k1);
        const std::string req = (sharded ? "R " + std::to_string(bytes) + " " + std::to_string(k0) + " " + std::to_string(k1) : "U " + std::to_string(bytes))
                                + " " + filepath + "\n";
        bool ok = k0 == k1 || (std::fputs(req.c_str(), g_helper.to) >= 0 && std::fflush(g_helper.to) == 0);
        char reply[512] = "";
        while (ok && k0 < k1) {             // "P <k>": piece k is in its slot; "OK": the file (or this rank's run of it) is through
            if (!std::fgets(reply, sizeof(reply), g_helper.from)) { ok = false; break; }
            if (std::strncmp(reply, "OK", 2) == 0) break;
            unsigned long long k = 0;
            if (std::sscanf(reply, "P %llu", &k) != 1 || tlm_paf_text_piece(paf_, k) != TLM_OK) { ok = false; break; }
            if (k + TLM_UPLOAD_SLOTS < k1) ok = std::fputs("A\n", g_helper.to) >= 0 && std::fflush(g_helper.to) == 0;
        }
        if (ok && sharded) {
            if (std::getenv("TLM_HOST_TIMING"))
                std::fprintf(stderr, "[host timing] sharded load of %s: pieces [%llu, %llu) brought by the helper process in %.0f ms\n", filepath,
                             (unsigned long long) k0, (unsigned long long) k1, since(t_up));
            tlm_exchange_info x{};
            const auto t_s = std::chrono::steady_clock::now();
            if (tlm_paf_text_share(paf_, 0, &x) != TLM_OK) die("tlm_paf_text_share");      // (a peer failed, or the swap did: the job ends)
            if (std::getenv("TLM_HOST_TIMING"))
                std::fprintf(stderr, "[host timing] sharded load of %s: swap with the other ranks %.0f ms (%.2f GB in)\n", filepath, since(t_s),
                             x.bytes_received / 1e9);
        }
        if (ok) {
            if (std::getenv("TLM_HOST_TIMING"))
                std::fprintf(stderr, "[host timing] text upload of %s by the helper process: %.0f ms\n", filepath,
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_up).count());
            const auto t_c = std::chrono::steady_clock::now();
            if (tlm_paf_text_commit(paf_, anchors ? 1 : 0, &n, &clean) != TLM_OK) clean = 0;
            if (std::getenv("TLM_HOST_TIMING")) std::fprintf(stderr, "[host timing] upload helper: commit (tokenise) %.0f ms\n", since(t_c));
            through_helper = true;
        } else {
            if (std::getenv("TLM_HOST_TIMING")) std::fprintf(stderr, "[host timing] upload helper failed (%s): uploading in-process\n", reply);
            tlm_paf_text_abort(paf_);
            g_helper.pending_replies = 0;
            stop_upload_helper();
        }
    }
    if (!through_helper && tlm_paf_add_file(paf_, filepath, anchors ? 1 : 0, &n, &clean) != TLM_OK) {
        // (a file that cannot be opened is reported by the host tokeniser with the reference's message)
        clean = 0;
    }
    if (!clean) {
        tlm_paf_destroy(paf_);
        paf_ = nullptr;
        std::vector<std::pair<std::string, bool>> redo;
        redo.swap(device_files_);
        files_.clear();
        for (auto &f : redo) {
            ulong ctr = 0;
            std::vector<char> path(f.first.begin(), f.first.end());
            path.push_back(0);
            if (!tokenize(path.data(), f.second, ctr)) return false;
        }
        return false;
    }
    device_files_.emplace_back(filepath, anchors);
    files_.emplace_back();              // keeps the file count; no host columns
    files_.back().anchors = anchors;
    records = (ulong) n;
    parse_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    parsed_on_device_ = true;
    parsed_parallel_ = false;
    return true;
}

bool OverlapGraph::load(char *filepath, bool anchors) {
    ensure_names();
    ulong ctr = 0;
    if (tokenize_on_device(filepath, anchors, ctr)) {
        pending_.emplace_back(filepath, ctr);
        return true;
    }
    parsed_on_device_ = false;
    if (!tokenize(filepath, anchors, ctr)) return false;
    if (deferred_) {
        pending_.emplace_back(filepath, ctr);
        return true;
    }
    if (!rebuild()) return false;
    size_t kept = edges_.size();        // (a lean graph has no edges_: the count comes from the device)
    if (lean_) {
        tlm_graph_info gi;
        tlm_graph_get_info(dev_, &gi);
        kept = (size_t) (gi.n_adj / 2);
    }
    std::cout << "Loaded " << kept << '/' << ctr << " edges from " << filepath << std::endl;   // :150
    return true;
}

bool OverlapGraph::end_loads() {
    deferred_ = false;
    if (pending_.empty()) return true;
    if (!rebuild()) return false;
    tlm_graph_info gi;
    tlm_graph_get_info(dev_, &gi);
    // edges_.size() after each load in the reference = kept records of all files up to that one
    const size_t first = files_.size() - pending_.size();
    uint64_t kept = 0;
    for (size_t i = 0; i < files_.size(); i++) {
        kept += i < 4 ? gi.n_kept_per_file[i] : 0;      // (the ABI reports the first four files)
        if (i >= first)
            std::cout << "Loaded " << kept << '/' << pending_[i - first].second << " edges from " << pending_[i - first].first
                      << std::endl;
    }
    pending_.clear();
    return true;
}

uint64_t OverlapGraph::parse_checksum() const {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *) p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    for (const File &f : files_) {
        for (const auto *col : {&f.q_id, &f.t_id, &f.q_len, &f.q_start, &f.q_end, &f.t_len, &f.t_start, &f.t_end, &f.matches})
            mix(col->data(), col->size() * 4);
        mix(f.strand.data(), f.strand.size());
    }
    ensure_names();
    for (const std::string &n : names_) mix(n.data(), n.size() + 1);
    return h;
}

bool OverlapGraph::tokenize(char *filepath, bool anchors, ulong &ctr) {
    const auto t_begin = std::chrono::steady_clock::now();
    const int fd = ::open(filepath, O_RDONLY);
    if (fd < 0) {
        std::cerr << "Cannot open file: " << filepath << std::endl;          // OverlapGraph.cpp:100-103
        return false;
    }
    struct stat st;
    ::fstat(fd, &st);
    const size_t size = (size_t) st.st_size;
    const char *data = size ? (const char *) ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0) : "";
    if (size && data == MAP_FAILED) {
        ::close(fd);
        std::cerr << "Cannot open file: " << filepath << std::endl;
        return false;
    }

    files_.emplace_back();
    File &f = files_.back();
    f.anchors = anchors;
    ctr = 0;
    bool parsed = false;

    // ---- parallel line-wise parse of clean input
    if (test_load_num_ == 0 && size) {
        unsigned nt = host_threads();
        nt = (unsigned) std::max<size_t>(1, std::min<size_t>(nt, size / (1 << 20) + 1));
        if (const char *e = std::getenv("TLM_PARSE_THREADS")) nt = (unsigned) std::min(64, std::max(1, std::atoi(e)));
        std::vector<const char *> cut(nt + 1);
        cut[0] = data;
        cut[nt] = data + size;
        for (unsigned t = 1; t < nt; t++) {
            const char *p = data + size / nt * t;
            const char *nl = (const char *) std::memchr(p, '\n', (size_t) (data + size - p));
            cut[t] = nl ? nl + 1 : data + size;
        }
        std::vector<Chunk> chunks(nt);
        std::vector<std::thread> th;
        // pass 1: records per range -> every range knows where it writes
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                chunks[t].begin = cut[t];
                chunks[t].end = cut[t + 1];
                chunks[t].lines = count_records(cut[t], cut[t + 1]);
            });
        for (auto &x : th) x.join();
        th.clear();
        size_t total = 0;
        for (Chunk &c : chunks) { c.base = total; total += c.lines; }
        for (auto *col : {&f.q_id, &f.t_id, &f.q_len, &f.q_start, &f.q_end, &f.t_len, &f.t_start, &f.t_end, &f.matches})
            col->resize_uninitialised(total);
        f.strand.resize_uninitialised(total);
        // pass 2: parse straight into the file's columns
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                const size_t b = chunks[t].base;
                parse_range(chunks[t], Columns{f.q_id.data() + b, f.t_id.data() + b, f.q_len.data() + b, f.q_start.data() + b,
                                                f.q_end.data() + b, f.t_len.data() + b, f.t_start.data() + b, f.t_end.data() + b,
                                                f.matches.data() + b, f.strand.data() + b});
            });
        for (auto &x : th) x.join();
        th.clear();
        const auto t_parsed = std::chrono::steady_clock::now();
        bool clean = true;
        for (const Chunk &c : chunks) clean &= c.clean && c.records == c.lines;
        if (clean) {
            // name tables: merged by all cores (NameIndex), ids in first-seen order; then local -> global ids in place
            std::vector<std::vector<int32_t>> glob;
            if (!index_) index_ = new NameIndex();
            index_->merge(chunks, names_, glob, nt);
            const auto t_merged = std::chrono::steady_clock::now();
            for (unsigned t = 0; t < nt; t++)
                th.emplace_back([&, t] {
                    const Chunk &c = chunks[t];
                    int32_t *q = f.q_id.data() + c.base, *tt = f.t_id.data() + c.base;
                    for (size_t i = 0; i < c.records; i++) {
                        q[i] = glob[t][(size_t) q[i]];
                        tt[i] = glob[t][(size_t) tt[i]];
                    }
                });
            for (auto &x : th) x.join();
            ctr = total;
            parsed = true;
            if (getenv("TLM_HOST_TIMING")) {
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                std::cerr << "[host timing] tokenise " << filepath << ": " << nt << " threads, count + parse " << ms(t_begin, t_parsed)
                          << " ms, name merge " << ms(t_parsed, t_merged) << " ms, id remap "
                          << ms(t_merged, std::chrono::steady_clock::now()) << " ms" << std::endl;
            }
        } else {
            for (auto *col : {&f.q_id, &f.t_id, &f.q_len, &f.q_start, &f.q_end, &f.t_len, &f.t_start, &f.t_end, &f.matches})
                col->resize_uninitialised(0);
            f.strand.resize_uninitialised(0);
        }
    }

    // ---- sequential token stream: the reference's exact operator>> behaviour for everything else
    if (!parsed) {
        Reader r{data, data + size};
        for (;;) {                                                           // OverlapGraph.cpp:109-121
            const char *qn, *tn, *dummy;
            size_t qn_n, tn_n, dn;
            int q_len, q_start, q_end, t_len, t_start, t_end, matches, block_len, mapq;
            char strand;
            if (!(r.token(qn, qn_n) && r.integer(q_len) && r.integer(q_start) && r.integer(q_end) && r.character(strand)
                  && r.token(tn, tn_n) && r.integer(t_len) && r.integer(t_start) && r.integer(t_end)
                  && r.integer(matches) && r.integer(block_len) && r.integer(mapq)))
                break;
            for (int k = 0; k < 4; k++) r.token(dummy, dn);                  // :124-127: exactly 4 more tokens
            ++ctr;
            if (!index_) index_ = new NameIndex();
            auto intern = [&](const char *s, size_t n) -> int32_t { return index_->intern(s, n, names_); };
            f.q_id.push_back(intern(qn, qn_n));
            f.t_id.push_back(intern(tn, tn_n));
            f.q_len.push_back(q_len); f.q_start.push_back(q_start); f.q_end.push_back(q_end);
            f.t_len.push_back(t_len); f.t_start.push_back(t_start); f.t_end.push_back(t_end);
            f.matches.push_back(matches);
            f.strand.push_back(strand == '-');
            if (test_load_num_ > 0 && test_load_num_ < ctr) break;            // :145-147
        }
    }
    if (size) ::munmap((void *) data, size);
    ::close(fd);
    parse_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    parsed_parallel_ = parsed;
    return true;
}

void OverlapGraph::ensure_names() const {
    if (!names_pending_) return;
    names_pending_ = false;
    const size_t n = name_off_.empty() ? 0 : name_off_.size() - 1;
    names_.resize(n);
    for (size_t i = 0; i < n; i++) names_[i].assign(name_blob_.data() + name_off_[i], name_off_[i + 1] - name_off_[i]);
    name_off_ = {};
    name_blob_ = {};
}

bool OverlapGraph::rebuild() {
    tlm_ctx *ctx = context();
    tlm_filter_params dfp;
    dfp.mode = (int32_t) filter_params_.mode;          // same numbering as TLM_MODE_*
    dfp.min_overlap_length = filter_params_.min_overlap_length;
    dfp.min_overlap_percentage = filter_params_.min_overlap_percentage;
    dfp.max_overhang_length = filter_params_.max_overhang_length;
    dfp.max_overhang_percentage = filter_params_.max_overhang_percentage;
    if (paf_) {         // the files were tokenised on the device: names from there, the columns never leave it
        uint32_t n_names = 0;
        uint64_t blob_bytes = 0;
        const auto t_fin = std::chrono::steady_clock::now();
        auto since = [](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        };
        if (tlm_paf_finish(paf_, &n_names, &blob_bytes) != TLM_OK) die("tlm_paf_finish");
        const double fin_ms = since(t_fin);
        const auto t_nm = std::chrono::steady_clock::now();
        std::vector<uint32_t> off((size_t) n_names + 1);
        std::vector<char> blob(blob_bytes ? blob_bytes : 1);
        if (tlm_paf_names(paf_, off.data(), blob.data(), nullptr) != TLM_OK) die("tlm_paf_names");
        // names_ itself is not filled here: the nodes below take their names straight from the blob, and the table of
        // ALL names (a million and more strings, 80 ms of this thread) is only read again by the host tokeniser and the
        // tests' checksum -- ensure_names() makes it then.  (Made on a second thread beside the walks instead, its
        // allocations slowed the Monte-Carlo kernel from 785 to 910 ms: every growth of the address space goes through the
        // driver's notifier; made by all cores here, the process took 90 ms longer to leave.)
        names_.clear();
        name_off_ = std::move(off);
        name_blob_ = std::move(blob);
        names_pending_ = true;
        const double names_ms = since(t_nm);
        const auto t_bg = std::chrono::steady_clock::now();
        if (dev_) { tlm_graph_destroy(dev_); dev_ = nullptr; }
        if (tlm_paf_build_graph(paf_, &dfp, &dev_) != TLM_OK) die("tlm_paf_build_graph");
        const double bg_ms = since(t_bg);
        double up = 0, pa = 0, fi = 0;
        tlm_paf_timing(paf_, &up, &pa, &fi);
        if (getenv("TLM_HOST_TIMING"))
            std::cerr << "[host timing] device tokeniser: text upload " << up << " ms, tokenise " << pa << " ms, names " << fi
                      << " ms" << std::endl;
        const auto t_ds = std::chrono::steady_clock::now();
        tlm_paf_destroy(paf_);
        paf_ = nullptr;
        const double ds_ms = since(t_ds);
        tlm_graph_info gi;
        tlm_graph_get_info(dev_, &gi);
        build_ms_ = gi.build_ms;
        const auto t_dl = std::chrono::steady_clock::now();
        const size_t N = gi.n_nodes;
        std::vector<uint32_t> node_len(N);
        std::vector<uint8_t> anchor(N);
        std::vector<int32_t> name_id(N);
        if (tlm_graph_download(dev_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               node_len.data(), anchor.data(), name_id.data()) != TLM_OK)
            die("tlm_graph_download");
        const double dl_ms = since(t_dl);
        nodes_.clear();
        nodes_.resize(N);
        edges_.clear();
        const double rs_ms = since(t_dl) - dl_ms;
        // (one thread: filled by 8, the table took 40 ms less and the process 0.1 s longer to leave)
        for (size_t i = 0; i < N; i++) {
            Node &n = nodes_[i];
            n.anchor = anchor[i] != 0;
            n.index = (uint) i;
            n.length = node_len[i];
            const size_t id = (size_t) name_id[i];
            n.name.assign(name_blob_.data() + name_off_[id], name_off_[id + 1] - name_off_[id]);
        }
        if (getenv("TLM_HOST_TIMING"))
            std::cerr << "[host timing] node table: " << N << " nodes, download " << dl_ms << " ms, resize " << rs_ms << " ms, fill "
                      << since(t_dl) - dl_ms - rs_ms << " ms" << std::endl;
        download_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dl).count();
        if (getenv("TLM_HOST_TIMING"))
            std::cerr << "[host timing] after the text: tlm_paf_finish " << fin_ms << " ms, names to host " << names_ms
                      << " ms, tlm_paf_build_graph " << bg_ms << " ms (device build " << build_ms_ << "), tlm_paf_destroy " << ds_ms
                      << " ms, node table " << download_ms_ << " ms" << std::endl;
        return true;
    }
    ensure_names();
    std::vector<tlm_records> recs(files_.size());
    for (size_t i = 0; i < files_.size(); i++) {
        const File &f = files_[i];
        tlm_records &r = recs[i];
        r.n = f.q_id.size();
        r.q_id = f.q_id.data(); r.t_id = f.t_id.data();
        r.q_len = f.q_len.data(); r.q_start = f.q_start.data(); r.q_end = f.q_end.data();
        r.t_len = f.t_len.data(); r.t_start = f.t_start.data(); r.t_end = f.t_end.data();
        r.matches = f.matches.data();
        r.strand = f.strand.data();
        r.anchors = f.anchors ? 1 : 0;
    }
    std::vector<uint8_t> is_ctg(names_.size());
    for (size_t i = 0; i < names_.size(); i++) is_ctg[i] = Utils::startsWithInsensitive(names_[i], "ctg");   // :134
    tlm_filter_params fp;
    fp.mode = (int32_t) filter_params_.mode;          // same numbering as TLM_MODE_*
    fp.min_overlap_length = filter_params_.min_overlap_length;
    fp.min_overlap_percentage = filter_params_.min_overlap_percentage;
    fp.max_overhang_length = filter_params_.max_overhang_length;
    fp.max_overhang_percentage = filter_params_.max_overhang_percentage;
    if (dev_) { tlm_graph_destroy(dev_); dev_ = nullptr; }
    if (tlm_graph_build_host(ctx, &fp, recs.data(), (int) recs.size(), (uint32_t) names_.size(), is_ctg.data(), &dev_)
        != TLM_OK)
        die("tlm_graph_build_host");
    tlm_graph_info gi;
    tlm_graph_get_info(dev_, &gi);
    build_ms_ = gi.build_ms;

    // download and materialise the reference's data members
    const auto t_dl = std::chrono::steady_clock::now();
    const size_t N = gi.n_nodes, A = (size_t) gi.n_adj;
    if (lean_) {        // nodes_ without their edge lists; edges_ stays empty (see set_lean)
        std::vector<uint32_t> node_len(N);
        std::vector<uint8_t> anchor(N);
        std::vector<int32_t> name_id(N);
        if (tlm_graph_download(dev_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                               node_len.data(), anchor.data(), name_id.data()) != TLM_OK)
            die("tlm_graph_download");
        nodes_.clear();
        nodes_.resize(N);
        edges_.clear();
        for (size_t i = 0; i < N; i++) {
            Node &n = nodes_[i];
            n.anchor = anchor[i] != 0;
            n.index = (uint) i;
            n.length = node_len[i];
            n.name = names_[(size_t) name_id[i]];
        }
        download_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_dl).count();
        return true;
    }
    std::vector<uint64_t> row_ptr(N + 1);
    std::vector<uint32_t> nbr(A), qs(A), qe(A), ts(A), te(A), node_len(N), kept(gi.n_kept);
    std::vector<float> os(A), si(A), es(A);
    std::vector<uint8_t> strand(A), anchor(N);
    std::vector<int32_t> name_id(N);
    if (tlm_graph_download(dev_, row_ptr.data(), nbr.data(), qs.data(), qe.data(), ts.data(), te.data(), os.data(),
                           si.data(), es.data(), strand.data(), node_len.data(), anchor.data(), name_id.data()) != TLM_OK)
        die("tlm_graph_download");
    if (tlm_graph_kept_records(dev_, kept.data()) != TLM_OK) die("tlm_graph_kept_records");
    const auto t_copied = std::chrono::steady_clock::now();
    nodes_.clear();
    nodes_.resize(N);
    std::vector<uint32_t> node_of_name(names_.size(), 0xFFFFFFFFu);
    {
        unsigned nt = std::max(1u, std::min(32u, tlm_cpu_share()));
        if (N < 4096) nt = 1;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                for (size_t i = N * t / nt; i < N * (t + 1) / nt; i++) {
                    Node &n = nodes_[i];
                    n.anchor = anchor[i] != 0;
                    n.index = (uint) i;
                    n.length = node_len[i];
                    n.name = names_[(size_t) name_id[i]];
                    node_of_name[(size_t) name_id[i]] = (uint32_t) i;
                    n.edges.resize((size_t) (row_ptr[i + 1] - row_ptr[i]));
                    for (size_t k = 0; k < n.edges.size(); k++) {
                        const size_t e = (size_t) row_ptr[i] + k;
                        n.edges[k] = Edge{nbr[e], (uint) i, qs[e], qe[e], ts[e], te[e], os[e], si[e], es[e], strand[e] != 0};
                    }
                }
            });
        for (auto &x : th) x.join();
    }
    const auto t_nodes = std::chrono::steady_clock::now();
    // edges_ in record order (OverlapGraph.cpp:241): the k-th kept record's unswapped copy is the next unread
    // entry of its target's list (per-node order is record order, each record adds one entry to either list).
    // Every thread scans all kept records but advances only the list cursors of its own range of nodes, and writes the
    // edges_ slots of the records whose target lies in that range.
    const size_t K = kept.size();
    std::vector<uint32_t> qn(K), tn(K);
    std::vector<size_t> base(files_.size() + 1, 0);
    for (size_t i = 0; i < files_.size(); i++) base[i + 1] = base[i] + files_[i].q_id.size();
    edges_.clear();
    edges_.resize(K);
    {
        unsigned nt = std::max(1u, std::min(32u, tlm_cpu_share()));
        if (K < 65536) nt = 1;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t fi = 0;
                for (size_t k = K * t / nt; k < K * (t + 1) / nt; k++) {
                    const uint32_t g = kept[k];
                    while (g >= base[fi + 1]) fi++;
                    while (g < base[fi]) fi--;
                    const size_t r = g - base[fi];
                    qn[k] = node_of_name[(size_t) files_[fi].q_id[r]];
                    tn[k] = node_of_name[(size_t) files_[fi].t_id[r]];
                }
            });
        for (auto &x : th) x.join();
        th.clear();
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                const uint32_t lo = (uint32_t) (N * t / nt), hi = (uint32_t) (N * (t + 1) / nt);
                std::vector<uint32_t> cursor(hi - lo, 0);
                for (size_t k = 0; k < K; k++) {
                    const uint32_t a = tn[k], q = qn[k];
                    if (a >= lo && a < hi) edges_[k] = nodes_[a].edges[cursor[a - lo]++];
                    if (q >= lo && q < hi) cursor[q - lo]++;
                }
            });
        for (auto &x : th) x.join();
    }
    const auto t_end = std::chrono::steady_clock::now();
    download_ms_ = std::chrono::duration<double, std::milli>(t_end - t_dl).count();
    if (getenv("TLM_HOST_TIMING"))
        std::cerr << "[host timing] device->host copies " << std::chrono::duration<double, std::milli>(t_copied - t_dl).count()
                  << " ms, nodes_ " << std::chrono::duration<double, std::milli>(t_nodes - t_copied).count() << " ms, edges_ "
                  << std::chrono::duration<double, std::milli>(t_end - t_nodes).count() << " ms" << std::endl;
    return true;
}

void OverlapGraph::Node::to_stream(std::ostream &s) const {
    s << (anchor ? '*' : ' ') << 'n' << index << "  len: " << length << "  name: " << name << "  edges: " << edges.size()
      << std::endl;
}

void OverlapGraph::Edge::to_stream(std::ostream &s) const {
    s << q_index << "-" << t_index << "  OS: " << overlap_score << "  ES: " << extension_score
      << "  SI: " << sequence_identity << std::endl;
}

static std::string stats_text(ulong anchors, ulong reads, ulong total, ulong min_len, ulong max_len, ulong min_con, ulong max_con,
                              ulong n_edges, ulong min_OL, ulong max_OL, float min_OS, float max_OS, float min_ES, float max_ES,
                              float min_SI, float max_SI) {
    std::stringstream str;
    str << "Nodes" << '\n'
        << "-  anchor: " << anchors << '\n' << "-    read: " << reads << '\n' << "-   total: " << total << '\n'
        << "- min_len: " << min_len << '\n' << "- max_len: " << max_len << '\n'
        << "- min_con: " << min_con << '\n' << "- max_con: " << max_con << '\n'
        << "Edges" << '\n'
        << "-   total: " << n_edges << '\n' << "-  min_OL: " << min_OL << '\n' << "-  max_OL: " << max_OL << '\n'
        << "-  min_OS: " << min_OS << '\n' << "-  max_OS: " << max_OS << '\n'
        << "-  min_ES: " << min_ES << '\n' << "-  max_ES: " << max_ES << '\n'
        << "-  min_SI: " << min_SI << '\n' << "-  max_SI: " << max_SI << '\n' << std::endl;
    return str.str();
}

std::string OverlapGraph::stats() {     // the report of OverlapGraph.cpp:24-95, same text
    if (lean_ && dev_) {                // the same extrema, reduced on the device over the resident adjacency
        tlm_graph_stats st;
        if (tlm_graph_get_stats(dev_, &st) != TLM_OK) die("tlm_graph_get_stats");
        return stats_text(st.anchors, st.reads, nodes_.size(), st.min_len, st.max_len, st.min_con, st.max_con, st.edges,
                          st.edges ? st.min_ol : ULONG_MAX, st.max_ol, st.min_os, st.max_os, st.min_es, st.max_es, st.min_si,
                          st.max_si);
    }
    ulong min_len = (ulong) -1, max_len = 0, min_con = (ulong) -1, max_con = 0, anchors = 0, reads = 0;
    for (const Node &n : nodes_) {
        (n.anchor ? anchors : reads)++;
        min_len = std::min<ulong>(min_len, n.length);
        max_len = std::max<ulong>(max_len, n.length);
        min_con = std::min<ulong>(min_con, n.edges.size());
        max_con = std::max<ulong>(max_con, n.edges.size());
    }
    ulong min_OL = ULONG_MAX, max_OL = 0;
    float min_OS = FLT_MAX, max_OS = 0, min_ES = FLT_MAX, max_ES = 0, min_SI = FLT_MAX, max_SI = 0;
    for (const Edge &e : edges_) {
        if (min_OS > e.overlap_score) min_OS = e.overlap_score;
        if (max_OS < e.overlap_score) max_OS = e.overlap_score;
        if (min_ES > e.extension_score) min_ES = e.extension_score;
        if (max_ES < e.extension_score) max_ES = e.extension_score;
        if (min_SI > e.sequence_identity) min_SI = e.sequence_identity;
        if (max_SI < e.sequence_identity) max_SI = e.sequence_identity;
        const int ol = (int) std::max(e.q_end - e.q_start, e.t_end - e.t_start);   // unsigned max, then int (:65)
        if (min_OL > (ulong) ol) min_OL = (ulong) ol;
        if (max_OL < (ulong) ol) max_OL = (ulong) ol;
    }
    return stats_text(anchors, reads, nodes_.size(), min_len, max_len, min_con, max_con, edges_.size(), min_OL, max_OL, min_OS,
                      max_OS, min_ES, max_ES, min_SI, max_SI);
}

// --- Utils ---------------------------------------------------------------------------------------

bool Utils::startsWithInsensitive(const std::string &str, const std::string &temp) {   // Utils.hpp:17-27
    if (str.length() < temp.length()) return false;
    for (size_t i = 0; i < temp.length(); i++)
        if (std::tolower((unsigned char) str[i]) != std::tolower((unsigned char) temp[i])) return false;
    return true;
}

float Utils::getMetric(const OverlapGraph::Edge &e, const Metrics &metric) {            // Utils.hpp:47-60
    switch (metric) {
        case Metrics::EXTENSION_SCORE: return e.extension_score;
        case Metrics::EXTENSION_SCORE_SQRT: return std::sqrt(e.extension_score);
        case Metrics::OVERLAP_SCORE: return e.overlap_score;
        case Metrics::OVERLAP_SCORE_SQRT: return std::sqrt(e.overlap_score);
    }
    return 0.0f;
}

const char *Utils::getMetricName(const Metrics &metric) {                               // Utils.hpp:62-75
    switch (metric) {
        case Metrics::EXTENSION_SCORE: return "EXTENSION_SCORE";
        case Metrics::EXTENSION_SCORE_SQRT: return "EXTENSION_SCORE_SQRT";
        case Metrics::OVERLAP_SCORE: return "OVERLAP_SCORE";
        case Metrics::OVERLAP_SCORE_SQRT: return "OVERLAP_SCORE_SQRT";
    }
    return "";
}
