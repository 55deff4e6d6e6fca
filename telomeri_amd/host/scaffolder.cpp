// telomeri_amd/host/scaffolder.cpp -- Scaffolder: FASTA/FASTQ in, one scaffold record out.
// Restates reference src/Scaffolder.cpp with its observable behaviour: two-line records, name = first token minus
// its first character, inclusive substring ends, '-' strand complemented but not reversed, the whole sequence
// reversed at the end, the node-name chain printed on stdout.
#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <iterator>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <thread>
#include "../csrc/cpu_share.h"
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "telomeri_host.hpp"

// Scaffolder::load (Scaffolder.cpp:12-41).  The reference reads the file line by line and takes the lines in turns:
// a name line (first whitespace-separated token minus its first character), a sequence line, and for a file name ending
// in 'q' two more lines that are skipped.  Nothing is checked: the roles follow from the line NUMBER alone.  That makes
// the file splittable: the bytes are cut into ranges at line ends, the ranges' line counts give every range the number
// of its first line, and all cores fill names_ / sequences_ in place (the reads of the A. thaliana-shaped input are
// 10 GB of text).
namespace {

const char *next_line(const char *p, const char *end, const char *&line_end) {
    const char *nl = (const char *) std::memchr(p, '\n', (size_t) (end - p));
    line_end = nl ? nl : end;
    return nl ? nl + 1 : end;
}

// what `std::istringstream iss(line); iss >> line;` leaves in `line`, minus its first character
std::string name_of_line(const char *b, const char *e) {
    const char *p = b;
    while (p < e && std::isspace((unsigned char) *p)) p++;
    if (p == e) return e - b > 1 ? std::string(b + 1, e) : std::string();      // no token: the line as it is
    const char *q = p;
    while (q < e && !std::isspace((unsigned char) *q)) q++;
    return std::string(p + 1, q);
}

}   // namespace

bool Scaffolder::load(const char *filename) {
    const size_t len = std::strlen(filename);
    const size_t period = len && filename[len - 1] == 'q' ? 4 : 2;      // lines per record
    const int fd = ::open(filename, O_RDONLY);
    if (fd < 0) {
        std::cerr << "Cannot open file: " << filename << std::endl;
        return false;
    }
    struct stat st;
    ::fstat(fd, &st);
    const size_t size = (size_t) st.st_size;
    if (size == 0) {
        ::close(fd);
        return true;
    }
    const char *data = (const char *) ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (data == MAP_FAILED) {
        ::close(fd);
        std::cerr << "Cannot open file: " << filename << std::endl;
        return false;
    }
    // (a scan in the helper process runs beside the other process's load: it leaves a quarter of the CPU share alone)
    unsigned nt = scan_only_ ? std::max(1u, tlm_cpu_share() * 3 / 4) : tlm_cpu_share();
    nt = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(nt, 64u), size / (4u << 20) + 1));
    std::vector<const char *> cut(nt + 1);
    cut[0] = data;
    cut[nt] = data + size;
    for (unsigned t = 1; t < nt; t++) {
        const char *p = data + size / nt * t;
        const char *nl = (const char *) std::memchr(p, '\n', (size_t) (data + size - p));
        cut[t] = nl ? nl + 1 : data + size;
    }
    // lines per range -> number of each range's first line
    std::vector<size_t> first_line(nt + 1, 0);
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t n = 0;
                for (const char *p = cut[t], *e; p < cut[t + 1]; n++) p = next_line(p, cut[t + 1], e);
                first_line[t + 1] = n;
            });
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < nt; t++) first_line[t + 1] += first_line[t];
    }
    const size_t lines = first_line[nt];
    const size_t name_base = names_.size(), seq_base = sequences_.size();
    names_.resize(name_base + (lines + period - 1) / period);                   // lines 0, P, 2P, ...
    sequences_.resize(seq_base + (lines >= 2 ? (lines - 2) / period + 1 : 0));          // lines 1, P + 1, ...
    where_.resize(sequences_.size(), View{-1, 0, 0});
    const int file = (int) files_.size();
    {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&, t] {
                size_t k = first_line[t];
                for (const char *p = cut[t], *e; p < cut[t + 1]; k++) {
                    const char *b = p;
                    p = next_line(p, cut[t + 1], e);
                    if (k % period == 0) names_[name_base + k / period] = name_of_line(b, e);
                    else if (k % period == 1) where_[seq_base + k / period] = View{file, (size_t) (b - data), (size_t) (e - b)};
                }
            });
        for (auto &x : th) x.join();
    }
    // The mapping goes again, here (telomeri_hip loads on a second thread while the device walks): 2.6 million page-table
    // entries of the 10 GB A. thaliana-shaped reads cost 0.4 s to take down, wherever that happens.  What the scaffold
    // needs of the sequences is read from the descriptor.
    if (!scan_only_) ::munmap((void *) data, size);
    files_.push_back(fd);
    if (scan_only_) return true;
    for (size_t i = name_base; i < names_.size(); i++) index_.emplace(names_[i], i);     // keeps the first index of a name
    return true;
}

// index file: "TLMIDX1\n", u64 names, then per name u32 length + bytes; u64 sequence slots, then per slot u64 offset,
// u64 length, u8 present
bool Scaffolder::save_index(const char *index_file) const {
    // a NEW file of this user only: never through a link somebody left in a shared directory, never over an existing file
    const int wfd = ::open(index_file, O_CREAT | O_EXCL | O_WRONLY | O_NOFOLLOW | O_CLOEXEC, 0600);
    FILE *f = wfd >= 0 ? ::fdopen(wfd, "wb") : nullptr;
    if (!f) {
        if (wfd >= 0) ::close(wfd);
        return false;
    }
    std::fputs("TLMIDX1\n", f);
    const uint64_t nn = names_.size(), ns = where_.size();
    std::fwrite(&nn, 8, 1, f);
    for (const std::string &n : names_) {
        const uint32_t len = (uint32_t) n.size();
        std::fwrite(&len, 4, 1, f);
        std::fwrite(n.data(), 1, len, f);
    }
    std::fwrite(&ns, 8, 1, f);
    for (const View &v : where_) {
        const uint64_t rec[2] = {(uint64_t) v.off, (uint64_t) v.n};
        const uint8_t present = v.file >= 0;
        std::fwrite(rec, 8, 2, f);
        std::fwrite(&present, 1, 1, f);
    }
    const bool written = !std::ferror(f);
    if (std::fclose(f) != 0 || !written) {
        ::unlink(index_file);
        return false;
    }
    return true;
}

bool Scaffolder::load_index(const char *index_file, const char *filename) {
    const int rfd = ::open(index_file, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    FILE *f = rfd >= 0 ? ::fdopen(rfd, "rb") : nullptr;
    if (!f) {
        if (rfd >= 0) ::close(rfd);
        return false;
    }
    // the counts in the file are bounded by the file itself before anything is sized from them: a name costs at least 4
    // bytes, a sequence record 17
    struct stat st;
    const uint64_t file_bytes = ::fstat(rfd, &st) == 0 ? (uint64_t) st.st_size : 0;
    char magic[8];
    uint64_t nn = 0, ns = 0;
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "TLMIDX1\n", 8) == 0 && std::fread(&nn, 8, 1, f) == 1
              && nn <= file_bytes / 4;
    std::vector<std::string> names;
    std::vector<View> where;
    if (ok) names.resize(nn);
    for (uint64_t i = 0; ok && i < nn; i++) {
        uint32_t len = 0;
        ok = std::fread(&len, 4, 1, f) == 1;
        if (ok) {
            ok = len <= file_bytes;
            if (ok) names[i].resize(len);
            ok = ok && (len == 0 || std::fread(&names[i][0], 1, len, f) == len);
        }
    }
    ok = ok && std::fread(&ns, 8, 1, f) == 1 && ns <= file_bytes / 17;
    const int fd = ok ? ::open(filename, O_RDONLY) : -1;
    ok = ok && (fd >= 0 || ns == 0);
    const int file = (int) files_.size();
    if (ok) where.resize(ns);
    for (uint64_t i = 0; ok && i < ns; i++) {
        uint64_t rec[2];
        uint8_t present = 0;
        ok = std::fread(rec, 8, 2, f) == 2 && std::fread(&present, 1, 1, f) == 1;
        where[i] = View{present ? file : -1, (size_t) rec[0], (size_t) rec[1]};
    }
    std::fclose(f);
    if (!ok) {
        if (fd >= 0) ::close(fd);
        return false;
    }
    if (fd >= 0) files_.push_back(fd);
    const size_t name_base = names_.size();
    names_.insert(names_.end(), std::make_move_iterator(names.begin()), std::make_move_iterator(names.end()));
    sequences_.resize(sequences_.size() + where.size());
    where_.insert(where_.end(), where.begin(), where.end());
    for (size_t i = name_base; i < names_.size(); i++) index_.emplace(names_[i], i);     // keeps the first index of a name
    return true;
}

Scaffolder::~Scaffolder() {
    for (int fd : files_) ::close(fd);
}

namespace {
bool read_at(int fd, char *p, size_t n, size_t off) {
    while (n) {
        const ssize_t r = ::pread(fd, p, n, (off_t) off);
        if (r <= 0) return false;
        p += r;
        off += (size_t) r;
        n -= (size_t) r;
    }
    return true;
}
}   // namespace

void Scaffolder::materialiseAll() {
    for (size_t i = 0; i < sequences_.size(); i++)
        if (where_[i].file >= 0) {
            sequences_[i].resize(where_[i].n);
            if (where_[i].n) read_at(files_[(size_t) where_[i].file], &sequences_[i][0], where_[i].n, where_[i].off);
            where_[i].file = -1;
        }
}

std::string &Scaffolder::getSequenceFrom(const std::string &name) {     // :104-112
    auto it = index_.find(name);
    if (it == index_.end() || it->second >= sequences_.size()) {
        // the reference indexes one past the end here (undefined behaviour); we stop with a message
        std::cerr << "Scaffolder: no sequence named '" << name << "'" << std::endl;
        std::exit(1);
    }
    const size_t i = it->second;
    if (where_[i].file >= 0) {      // first use: out of the file
        sequences_[i].resize(where_[i].n);
        if (where_[i].n) read_at(files_[(size_t) where_[i].file], &sequences_[i][0], where_[i].n, where_[i].off);
        where_[i].file = -1;
    }
    return sequences_[i];
}

std::string Scaffolder::substring(std::string &str, ulong start, ulong end) {      // :114-117, both ends inclusive
    return str.substr(start, end - start + 1);          // throws std::out_of_range like the reference when start > size
}

void Scaffolder::swapBases(std::string &str) {          // :129-147
    for (char &c : str) {
        switch (c) {
            case 'A': c = 'T'; break;
            case 'T': c = 'A'; break;
            case 'G': c = 'C'; break;
            case 'C': c = 'G'; break;
            default: break;
        }
    }
}

// ---- Scaffolder::write (Scaffolder.cpp:43-102), written from its behaviour ---------------------------------
// The scaffold runs from the LAST node of the path to the first.  The reference appends the reverse of every piece to one
// string and reverses that string at the end; the same text comes out when the pieces are kept as they are cut and
// written in the opposite order, which is what happens here.  Pieces, for a path of nodes n_0..n_m and edges e_0..e_m-1
// (e_k joins n_k and n_k+1; coordinates with INCLUSIVE ends, clamped at the end of the string like std::string::substr):
//   n_0        : seq[e_0.t_end + 1 ..]
//   n_k, 0<k<m : with a = e_k.t_end, b = e_k-1.q_end:   a < b  -> seq[a + 1 .. b], complemented (NOT reversed) when
//                e_k-1 lies on the '-' strand; a == b -> nothing; a > b -> the scaffold loses the a - b characters
//                appended last (the read is skipped and does not appear in the printed name chain either)
//   n_m        : seq[0 .. e_m-1.q_end]
namespace {

struct Piece {              // a cut of a sequence that stays where it is: a loaded string (p) or a range of a file (fd, off)
    const char *p;
    int fd;
    size_t off, n;
    bool complement;        // A <-> T, G <-> C on the way out (Scaffolder::swapBases)
};

class PieceStack {          // the pieces cut so far, oldest first
    std::vector<Piece> pieces_;
    size_t total_ = 0;

public:
    void push(Piece piece) {
        total_ += piece.n;
        pieces_.push_back(piece);
    }
    // Drop the `count` characters appended last.  In the reference's reversed string those are its tail; here they are
    // the HEAD of the newest piece, then of the one before it, and so on.
    // QUIRK (unsigned arithmetic of Scaffolder.cpp:90): asking for more than there is leaves everything in place.
    void drop_last(ulong count) {
        if (count > total_) return;
        total_ -= count;
        while (count > 0) {
            Piece &newest = pieces_.back();
            if (newest.n <= count) {
                count -= newest.n;
                pieces_.pop_back();
            } else {
                if (newest.p) newest.p += count;
                newest.off += count;
                newest.n -= count;
                count = 0;
            }
        }
    }
    // The scaffold text, newest piece first: every piece has its place in one buffer (prefix sums of the lengths); the
    // cores copy / read their share of the pieces into it (214 000 pieces of a few hundred characters on the
    // A. thaliana-shaped input).
    // The scaffold's text, newest piece first, behind the `header` bytes the file already holds, and a newline behind it.  The
    // file is grown to its final size and MAPPED: the threads copy their pieces straight into the page cache (and take the
    // page faults of their share), no assembled copy and no single 77 MB write.  What cannot be mapped gets the text through a
    // buffer and the stream.
    bool write_newest_first(const char *filename, size_t header, std::ostream &out) const {
        const size_t n = pieces_.size();
        std::vector<size_t> at(n + 1, 0);
        for (size_t i = 0; i < n; i++) at[i + 1] = at[i] + pieces_[n - 1 - i].n;      // i-th piece written = pieces_[n-1-i]
        const size_t total = header + at[n] + 1;
        const int fd = ::open(filename, O_RDWR);
        char *map = (char *) MAP_FAILED;
        if (fd >= 0 && ::ftruncate(fd, (off_t) total) == 0) {
            map = (char *) ::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == (char *) MAP_FAILED && ::ftruncate(fd, (off_t) header) != 0) {       // (back to the title line)
                ::close(fd);
                return false;
            }
        }
        // (not zero-filled: the threads below write every byte, and take the page faults of their share)
        std::unique_ptr<char[]> buffer(map == (char *) MAP_FAILED ? new char[at[n] + 1] : nullptr);
        char *const text = map != (char *) MAP_FAILED ? map + header : buffer.get();
        char swap[256];
        for (int c = 0; c < 256; c++) swap[c] = (char) c;
        swap[(unsigned char) 'A'] = 'T'; swap[(unsigned char) 'T'] = 'A';
        swap[(unsigned char) 'G'] = 'C'; swap[(unsigned char) 'C'] = 'G';
        unsigned nt = tlm_cpu_share();
        nt = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(nt, 16u), n / 4096 + 1));
        std::vector<char> good(nt, 1);
        auto work = [&](unsigned t) {
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; i++) {
                const Piece &pc = pieces_[n - 1 - i];
                char *dst = text + at[i];
                if (pc.p) std::memcpy(dst, pc.p, pc.n);
                else if (pc.n && !read_at(pc.fd, dst, pc.n, pc.off)) good[t] = 0;
                if (pc.complement)
                    for (size_t k = 0; k < pc.n; k++) dst[k] = swap[(unsigned char) dst[k]];
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        text[at[n]] = '\n';
        bool ok = true;
        for (char g : good) ok = ok && g;
        if (fd >= 0) ::close(fd);
        if (map != (char *) MAP_FAILED) {
            ::munmap(map, total);
            return ok;
        }
        // (a pipe, a device, a file system without shared mappings: the text goes behind the title line through the stream)
        out.write(text, (std::streamsize) (at[n] + 1));
        out.flush();
        return ok && out.good();
    }
};

// Scaffolder::substring (Scaffolder.cpp:114-117: str.substr(start, end - start + 1), both ends inclusive) on a view of
// `size` characters: clamped at the end of the sequence, std::out_of_range when `start` lies beyond it -- as
// std::string::substr.
Piece cut(const Piece &whole, ulong start, ulong end, bool complement) {
    const size_t size = whole.n;
    if (start > size) throw std::out_of_range("basic_string::substr");
    const ulong count = end - start + 1;         // (wraps like the reference's unsigned arithmetic)
    return Piece{whole.p ? whole.p + start : nullptr, whole.fd, whole.off + start, (size_t) std::min<ulong>(count, size - start),
                 complement};
}

}   // namespace

bool Scaffolder::write(const char *filename) {
    const auto t_w = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_w).count(); };
    std::ofstream out(filename);
    if (!out) {
        std::cerr << "Cannot write file: " << filename << std::endl;
        return false;
    }
    out << ">Resulting_HERA_scaffold" << std::endl;
    const size_t n_nodes = p_.nodes_.size(), n_edges = p_.edges_.size();
    if (n_nodes == 0) {
        std::cerr << "Scaffold is empty!" << std::endl;
        return false;
    }
    auto name_of = [&](size_t k) -> const std::string & { return p_.nodes_[k]->name; };
    if (n_nodes == 1) {         // QUIRK: the sequence comes out REVERSED here, and without a newline after it
        std::cout << name_of(0) << std::endl;
        const std::string &seq = getSequenceFrom(name_of(0));
        out << std::string(seq.rbegin(), seq.rend());
        return true;
    }
    // The pieces are views: a read contributes a couple of hundred characters of its ten thousand, so the sequences stay
    // where they are (the mapped file, or sequences_[i] once somebody has asked for the string) until the final write.
    // The sequences of the nodes that contribute text are looked up by name first, by all cores (455 000 finds in a table
    // of a million names on the A. thaliana-shaped input: 86 ms of cache misses on one core); a name that is missing is
    // reported for the first node in path order, as the serial loop did.
    std::vector<size_t> wanted;                 // node positions whose sequence is cut below
    wanted.push_back(0);
    for (size_t k = 1; k < n_edges; k++)
        if ((ulong) p_.edges_[k]->t_end < (ulong) p_.edges_[k - 1]->q_end) wanted.push_back(k);
    wanted.push_back(n_nodes - 1);
    std::vector<Piece> views(wanted.size());
    {
        constexpr size_t MISSING = (size_t) -1;
        unsigned nt = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(tlm_cpu_share(), 16u), wanted.size() / 4096 + 1));
        std::vector<size_t> first_missing(nt, MISSING);
        auto work = [&](unsigned t) {
            for (size_t j = wanted.size() * t / nt; j < wanted.size() * (t + 1) / nt; j++) {
                auto it = index_.find(name_of(wanted[j]));
                if (it == index_.end() || it->second >= sequences_.size()) {
                    if (first_missing[t] == MISSING) first_missing[t] = j;
                    continue;
                }
                const size_t i = it->second;
                views[j] = where_[i].file >= 0 ? Piece{nullptr, files_[(size_t) where_[i].file], where_[i].off, where_[i].n, false}
                                               : Piece{sequences_[i].data(), -1, 0, sequences_[i].size(), false};
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (size_t j : first_missing)
            if (j != MISSING) {
                std::cerr << "Scaffolder: no sequence named '" << name_of(wanted[j]) << "'" << std::endl;
                std::exit(1);
            }
    }
    size_t next_view = 0;
    PieceStack scaffold;
    std::string chain = name_of(0);         // the node-name chain the reference prints piece by piece
    Piece seq = views[next_view++];
    scaffold.push(cut(seq, (ulong) p_.edges_[0]->t_end + 1, seq.n, false));
    for (size_t k = 1; k < n_edges; k++) {
        const OverlapGraph::Edge &arriving = *p_.edges_[k - 1], &leaving = *p_.edges_[k];
        const ulong from = leaving.t_end, to = arriving.q_end;
        if (from < to) {
            seq = views[next_view++];
            chain += '-';
            chain += name_of(k);
            scaffold.push(cut(seq, from + 1, to, arriving.relative_strand));
        } else if (from > to) {
            scaffold.drop_last(from - to);
        }
    }
    chain += '-';
    chain += name_of(n_nodes - 1);
    std::cout << chain << std::endl;
    seq = views[next_view++];
    scaffold.push(cut(seq, 0, p_.edges_[n_edges - 1]->q_end, false));
    const double pieces_ms = since();
    out.flush();                                      // (the title line is in the file: the text goes behind it)
    if (!scaffold.write_newest_first(filename, sizeof(">Resulting_HERA_scaffold\n") - 1, out)) {
        std::cerr << "Cannot read a sequence file again" << std::endl;
        return false;
    }
    if (std::getenv("TLM_HOST_TIMING"))
        std::cerr << "[host timing] scaffold write: " << n_nodes << " nodes looked up and cut in " << pieces_ms
                  << " ms, text assembled and written in " << since() - pieces_ms << " ms" << std::endl;
    return true;
}
