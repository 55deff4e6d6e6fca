// telomeri_amd/host/scaffolder.cpp -- Scaffolder: FASTA/FASTQ in, one scaffold record out.
// Restates reference src/Scaffolder.cpp with its observable behaviour: two-line records, name = first token minus
// its first character, inclusive substring ends, '-' strand complemented but not reversed, the whole sequence
// reversed at the end, the node-name chain printed on stdout.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

#include "telomeri_host.hpp"

bool Scaffolder::load(const char *filename) {           // Scaffolder.cpp:12-41
    const bool fastq = filename[std::strlen(filename) - 1] == 'q';
    std::ifstream in(filename);
    if (!in) {
        std::cerr << "Cannot open file: " << filename << std::endl;
        return false;
    }
    std::string line;
    bool reading_name = true;
    while (std::getline(in, line)) {
        if (reading_name) {
            std::istringstream iss(line);
            iss >> line;                                // first token (an empty line keeps `line` as it was)
            names_.emplace_back(line.empty() ? line : line.substr(1));
            index_.emplace(names_.back(), names_.size() - 1);       // keeps the first index of a repeated name
        } else {
            sequences_.push_back(line);
            if (fastq) {                                // skip "+" and the quality line
                std::getline(in, line);
                std::getline(in, line);
            }
        }
        reading_name = !reading_name;
    }
    return true;
}

std::string &Scaffolder::getSequenceFrom(const std::string &name) {     // :104-112
    auto it = index_.find(name);
    if (it == index_.end() || it->second >= sequences_.size()) {
        // the reference indexes one past the end here (undefined behaviour); we stop with a message
        std::cerr << "Scaffolder: no sequence named '" << name << "'" << std::endl;
        std::exit(1);
    }
    return sequences_[it->second];
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

class PieceStack {          // the pieces cut so far, oldest first
    std::vector<std::string> pieces_;
    size_t total_ = 0;

public:
    void push(std::string piece) {
        total_ += piece.size();
        pieces_.push_back(std::move(piece));
    }
    // Drop the `count` characters appended last.  In the reference's reversed string those are its tail; here they are
    // the HEAD of the newest piece, then of the one before it, and so on.
    // QUIRK (unsigned arithmetic of Scaffolder.cpp:90): asking for more than there is leaves everything in place.
    void drop_last(ulong count) {
        if (count > total_) return;
        total_ -= count;
        while (count > 0) {
            std::string &newest = pieces_.back();
            if (newest.size() <= count) {
                count -= newest.size();
                pieces_.pop_back();
            } else {
                newest.erase(0, count);
                count = 0;
            }
        }
    }
    void write_newest_first(std::ostream &out) const {
        for (auto it = pieces_.rbegin(); it != pieces_.rend(); ++it) out << *it;
    }
};

}   // namespace

bool Scaffolder::write(const char *filename) {
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
    PieceStack scaffold;
    std::cout << name_of(0);
    {
        std::string &seq = getSequenceFrom(name_of(0));
        scaffold.push(substring(seq, (ulong) p_.edges_[0]->t_end + 1, seq.size()));
    }
    for (size_t k = 1; k < n_edges; k++) {
        const OverlapGraph::Edge &arriving = *p_.edges_[k - 1], &leaving = *p_.edges_[k];
        const ulong from = leaving.t_end, to = arriving.q_end;
        if (from < to) {
            std::string piece = substring(getSequenceFrom(name_of(k)), from + 1, to);
            if (arriving.relative_strand) swapBases(piece);
            std::cout << '-' << name_of(k);
            scaffold.push(std::move(piece));
        } else if (from > to) {
            scaffold.drop_last(from - to);
        }
    }
    std::cout << '-' << name_of(n_nodes - 1) << std::endl;
    scaffold.push(substring(getSequenceFrom(name_of(n_nodes - 1)), 0, p_.edges_[n_edges - 1]->q_end));
    scaffold.write_newest_first(out);
    out << std::endl;
    return true;
}
