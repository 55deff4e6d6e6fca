// telomeri_amd/host/overlap_graph.cpp -- OverlapGraph over the C ABI.
//
// The host keeps what is inherently host work in the reference's load (OverlapGraph.cpp:97-154): reading
// the file, tokenising with operator>> semantics, and giving names ids.  Filtering, scoring, node
// numbering and adjacency construction run on the GPU (tlm_graph_build_host); the result is downloaded
// into the public nodes_ / edges_ so that every consumer of the reference's data members keeps working.
#include "telomeri_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/telomeri_hip.h"

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

tlm_ctx *OverlapGraph::context() {
    static tlm_ctx *ctx = nullptr;
    if (!ctx) {
        const char *d = std::getenv("TLM_DEVICE");
        if (tlm_ctx_create(d ? std::atoi(d) : 0, &ctx) != TLM_OK) die("no usable gfx950 device (there is no CPU fallback)");
    }
    return ctx;
}

OverlapGraph::OverlapGraph() = default;

OverlapGraph::~OverlapGraph() {
    if (dev_) tlm_graph_destroy(dev_);
}

bool OverlapGraph::load(char *filepath, bool anchors) {
    std::ifstream in(filepath, std::ios::binary);
    if (!in) {
        std::cerr << "Cannot open file: " << filepath << std::endl;          // OverlapGraph.cpp:100-103
        return false;
    }
    std::string buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    in.close();

    files_.emplace_back();
    File &f = files_.back();
    f.anchors = anchors;
    Reader r{buf.data(), buf.data() + buf.size()};
    ulong ctr = 0;
    for (;;) {                                                               // OverlapGraph.cpp:109-121
        const char *qn, *tn, *dummy;
        size_t qn_n, tn_n, dn;
        int q_len, q_start, q_end, t_len, t_start, t_end, matches, block_len, mapq;
        char strand;
        if (!(r.token(qn, qn_n) && r.integer(q_len) && r.integer(q_start) && r.integer(q_end) && r.character(strand)
              && r.token(tn, tn_n) && r.integer(t_len) && r.integer(t_start) && r.integer(t_end) && r.integer(matches)
              && r.integer(block_len) && r.integer(mapq)))
            break;
        for (int k = 0; k < 4; k++) r.token(dummy, dn);                      // :124-127: exactly 4 more tokens
        ++ctr;
        auto intern = [&](const char *s, size_t n) -> int32_t {
            std::string key(s, n);
            auto it = name_id_.find(key);
            if (it != name_id_.end()) return it->second;
            const int32_t id = (int32_t) names_.size();
            names_.push_back(key);
            name_id_.emplace(std::move(key), id);
            return id;
        };
        f.q_id.push_back(intern(qn, qn_n));
        f.t_id.push_back(intern(tn, tn_n));
        f.q_len.push_back(q_len); f.q_start.push_back(q_start); f.q_end.push_back(q_end);
        f.t_len.push_back(t_len); f.t_start.push_back(t_start); f.t_end.push_back(t_end);
        f.matches.push_back(matches);
        f.strand.push_back(strand == '-');
        if (test_load_num_ > 0 && test_load_num_ < ctr) break;                // :145-147
    }
    if (!rebuild()) return false;
    std::cout << "Loaded " << edges_.size() << '/' << ctr << " edges from " << filepath << std::endl;   // :150
    return true;
}

bool OverlapGraph::rebuild() {
    tlm_ctx *ctx = context();
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
    const size_t N = gi.n_nodes, A = (size_t) gi.n_adj;
    std::vector<uint64_t> row_ptr(N + 1);
    std::vector<uint32_t> nbr(A), qs(A), qe(A), ts(A), te(A), node_len(N), kept(gi.n_kept);
    std::vector<float> os(A), si(A), es(A);
    std::vector<uint8_t> strand(A), anchor(N);
    std::vector<int32_t> name_id(N);
    if (tlm_graph_download(dev_, row_ptr.data(), nbr.data(), qs.data(), qe.data(), ts.data(), te.data(), os.data(),
                           si.data(), es.data(), strand.data(), node_len.data(), anchor.data(), name_id.data()) != TLM_OK)
        die("tlm_graph_download");
    if (tlm_graph_kept_records(dev_, kept.data()) != TLM_OK) die("tlm_graph_kept_records");
    nodes_.clear();
    nodes_.resize(N);
    std::vector<uint32_t> node_of_name(names_.size(), 0xFFFFFFFFu);
    for (size_t i = 0; i < N; i++) {
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
    // edges_ in record order (OverlapGraph.cpp:241): the k-th kept record's unswapped copy is the next unread
    // entry of its target's list (per-node order is record order, each record adds one entry to either list)
    edges_.clear();
    edges_.reserve(kept.size());
    std::vector<uint32_t> cursor(N, 0);
    std::vector<size_t> base(files_.size() + 1, 0);
    for (size_t i = 0; i < files_.size(); i++) base[i + 1] = base[i] + files_[i].q_id.size();
    for (uint32_t g : kept) {
        size_t fi = 0;
        while (g >= base[fi + 1]) fi++;
        const size_t r = g - base[fi];
        const uint32_t qn = node_of_name[(size_t) files_[fi].q_id[r]], tn = node_of_name[(size_t) files_[fi].t_id[r]];
        edges_.push_back(nodes_[tn].edges[cursor[tn]++]);
        cursor[qn]++;
    }
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

std::string OverlapGraph::stats() {     // the report of OverlapGraph.cpp:24-95, same text
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
    std::stringstream str;
    str << "Nodes" << '\n'
        << "-  anchor: " << anchors << '\n' << "-    read: " << reads << '\n' << "-   total: " << nodes_.size() << '\n'
        << "- min_len: " << min_len << '\n' << "- max_len: " << max_len << '\n'
        << "- min_con: " << min_con << '\n' << "- max_con: " << max_con << '\n'
        << "Edges" << '\n'
        << "-   total: " << edges_.size() << '\n' << "-  min_OL: " << min_OL << '\n' << "-  max_OL: " << max_OL << '\n'
        << "-  min_OS: " << min_OS << '\n' << "-  max_OS: " << max_OS << '\n'
        << "-  min_ES: " << min_ES << '\n' << "-  max_ES: " << max_ES << '\n'
        << "-  min_SI: " << min_SI << '\n' << "-  max_SI: " << max_SI << '\n' << std::endl;
    return str.str();
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
