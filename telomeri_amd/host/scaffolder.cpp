// telomeri_amd/host/scaffolder.cpp -- Scaffolder: FASTA/FASTQ in, one scaffold record out.
// Restates reference src/Scaffolder.cpp with its observable behaviour: two-line records, name = first token minus
// its first character, inclusive substring ends, '-' strand complemented but not reversed, the whole sequence
// reversed at the end, the node-name chain printed on stdout.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

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

static std::string reversed(std::string s) {
    std::reverse(s.begin(), s.end());
    return s;
}

bool Scaffolder::write(const char *filename) {          // Scaffolder.cpp:43-102
    std::ofstream out(filename);
    if (!out) {
        std::cerr << "Cannot write file: " << filename << std::endl;
        return false;
    }
    out << ">Resulting_HERA_scaffold" << std::endl;
    std::string sequence, tmp;
    if (p_.nodes_.empty()) {
        std::cerr << "Scaffold is empty!" << std::endl;
        return false;
    }
    if (p_.nodes_.size() == 1) {                        // single node: the sequence reversed, no trailing newline
        std::cout << p_.nodes_[0]->name << std::endl;
        out << reversed(getSequenceFrom(p_.nodes_[0]->name));
        return true;
    }
    std::cout << p_.nodes_[0]->name;
    tmp = getSequenceFrom(p_.nodes_[0]->name);
    sequence += reversed(substring(tmp, (ulong) p_.edges_[0]->t_end + 1, tmp.size()));
    for (ulong i = 1; i < p_.edges_.size(); i++) {
        const OverlapGraph::Edge *e = p_.edges_[i - 1], *e_next = p_.edges_[i];
        const ulong start = e_next->t_end, end = e->q_end;
        if (start < end) {
            tmp = getSequenceFrom(p_.nodes_[i]->name);
            if (e->relative_strand) swapBases(tmp);
            std::cout << '-' << p_.nodes_[i]->name;
            sequence += reversed(substring(tmp, start + 1, end));
        } else if (start == end) {
            continue;
        } else {                                        // going backwards: drop the afflicted tail
            sequence = substring(sequence, 0, sequence.length() - 1 - (start - end));
        }
    }
    std::cout << '-' << p_.nodes_.back()->name << std::endl;
    tmp = getSequenceFrom(p_.nodes_.back()->name);
    sequence += reversed(substring(tmp, 0, p_.edges_.back()->q_end));
    std::reverse(sequence.begin(), sequence.end());
    out << sequence << std::endl;
    return true;
}
