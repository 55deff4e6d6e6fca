// telomeri_amd/host/post_tool.cpp -- `telomeri_post`: runs ONLY the host stages (grouping, consensus, scaffold path,
// Scaffolder) on a graph and a path list read from a dump in oracle/ref_dump's text format (its `n`, `a` and `U`
// sections).  No GPU is involved: this is how the restatement of SURVEY.md §8f rows 3-4 is compared, in the build
// container, with the unmodified reference (same paths in -> same scaffold FASTA out).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include <functional>

#include "telomeri_host.hpp"

int run_post_stages(PathManager &pm, std::map<PathManager::AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                    const char *reads_file, const char *contigs_file, const char *output_file,
                    std::function<Scaffolder *()> loaded, bool leave_mapped);

// ---- the same input as raw arrays (`--binary <dir>`): what a full-size comparison can afford (the text dump of the A. thaliana
// shape would be 20 GB of digits).  Little-endian files in <dir>:
//   meta.txt "N A P T" | names.txt (N lines) | node_len.u32 node_anchor.u8 [N] | row_ptr.u64 [N+1]
//   nbr q_start q_end t_start t_end .u32, os si es .f32, strand.u8 [A]                (CSR, per-node order = record order)
//   path_off.u64 [P+1] path_length.i64 [P] path_nodes.u32 path_slots.u32 [T]          (paths_ AFTER filterUnique)
template <typename T>
static bool read_array(const std::string &dir, const char *name, size_t n, std::vector<T> &out) {
    out.resize(n);
    FILE *f = std::fopen((dir + "/" + name).c_str(), "rb");
    if (!f) { std::cerr << "Cannot open file: " << dir << "/" << name << std::endl; return false; }
    const size_t got = n ? std::fread(out.data(), sizeof(T), n, f) : 0;
    const bool at_end = std::fgetc(f) == EOF;
    std::fclose(f);
    if (got != n || !at_end) { std::cerr << dir << "/" << name << ": expected " << n << " entries" << std::endl; return false; }
    return true;
}

static bool load_binary(const std::string &dir, OverlapGraph &g, std::vector<Path> &paths) {
    size_t N = 0, A = 0, P = 0, T = 0;
    {
        std::ifstream m(dir + "/meta.txt");
        if (!(m >> N >> A >> P >> T)) { std::cerr << "Cannot read " << dir << "/meta.txt" << std::endl; return false; }
    }
    std::vector<uint32_t> node_len, nbr, qs, qe, ts, te, pn, psl;
    std::vector<uint8_t> node_anchor, strand;
    std::vector<uint64_t> row_ptr, off;
    std::vector<float> os, si, es;
    std::vector<int64_t> plen;
    if (!read_array(dir, "node_len.u32", N, node_len) || !read_array(dir, "node_anchor.u8", N, node_anchor)
        || !read_array(dir, "row_ptr.u64", N + 1, row_ptr)) return false;
    g.nodes_.resize(N);
    {
        std::ifstream names(dir + "/names.txt");
        for (size_t i = 0; i < N; i++) {
            OverlapGraph::Node &nd = g.nodes_[i];
            if (!std::getline(names, nd.name)) { std::cerr << dir << "/names.txt: " << N << " names expected" << std::endl; return false; }
            nd.index = (uint) i;
            nd.anchor = node_anchor[i] != 0;
            nd.length = node_len[i];
            if (row_ptr[i + 1] < row_ptr[i] || row_ptr[i + 1] > A) { std::cerr << "row_ptr is not a prefix" << std::endl; return false; }
            nd.edges.resize(row_ptr[i + 1] - row_ptr[i]);
        }
    }
    if (!read_array(dir, "nbr.u32", A, nbr) || !read_array(dir, "q_start.u32", A, qs) || !read_array(dir, "q_end.u32", A, qe)
        || !read_array(dir, "t_start.u32", A, ts) || !read_array(dir, "t_end.u32", A, te) || !read_array(dir, "os.f32", A, os)
        || !read_array(dir, "si.f32", A, si) || !read_array(dir, "es.f32", A, es) || !read_array(dir, "strand.u8", A, strand)) return false;
    for (size_t i = 0; i < N; i++)
        for (uint64_t k = row_ptr[i], j = 0; k < row_ptr[i + 1]; k++, j++) {
            if (nbr[k] >= N) { std::cerr << "neighbour out of range" << std::endl; return false; }
            g.nodes_[i].edges[j] = OverlapGraph::Edge{nbr[k], (uint) i, qs[k], qe[k], ts[k], te[k], os[k], si[k], es[k], strand[k] != 0};
        }
    { std::vector<uint32_t>().swap(nbr); std::vector<uint32_t>().swap(qs); std::vector<uint32_t>().swap(qe); std::vector<uint32_t>().swap(ts);
      std::vector<uint32_t>().swap(te); std::vector<float>().swap(os); std::vector<float>().swap(si); std::vector<float>().swap(es); }
    if (!read_array(dir, "path_off.u64", P + 1, off) || !read_array(dir, "path_length.i64", P, plen)
        || !read_array(dir, "path_nodes.u32", T, pn) || !read_array(dir, "path_slots.u32", T, psl)) return false;
    paths.resize(P);
    for (size_t i = 0; i < P; i++) {
        if (off[i + 1] < off[i] || off[i + 1] > T) { std::cerr << "path_off is not a prefix" << std::endl; return false; }
        Path &p = paths[i];
        const size_t n = off[i + 1] - off[i];
        p.nodes_.reserve(n);
        p.edges_.reserve(n ? n - 1 : 0);
        for (size_t k = 0; k < n; k++) {
            const uint32_t v = pn[off[i] + k];
            if (v >= N) { std::cerr << "path node out of range" << std::endl; return false; }
            p.nodes_.push_back(&g.nodes_[v]);
            if (k + 1 < n) {
                const uint32_t sl = psl[off[i] + k];
                if (sl >= g.nodes_[v].edges.size()) { std::cerr << "path slot out of range" << std::endl; return false; }
                p.edges_.push_back(&g.nodes_[v].edges[sl]);
            }
        }
        p.updateLength();
        if (p.length() != plen[i]) { std::cerr << "binary dump: path length mismatch at path " << i << std::endl; return false; }
    }
    return true;
}

static float bits_to_float(unsigned u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char **argv) {
    PathManager pm;
    PathManager::Parameters gp = pm.params_;
    const char *groups_out = nullptr;
    const char *binary_dir = nullptr;
    std::vector<char *> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--len-thr" && i + 1 < argc) pm.params_.len_threshold = std::atol(argv[++i]);
        else if (a == "--w-size" && i + 1 < argc) pm.params_.window_size = std::strtoul(argv[++i], nullptr, 10);
        else if (a == "--r-thr" && i + 1 < argc) pm.params_.ratio_threshold = std::strtof(argv[++i], nullptr);
        else if (a == "--binary" && i + 1 < argc) binary_dir = argv[++i];
        else if (a == "--groups" && i + 4 < argc) {        // <out file> <len_thr> <w_size> <r_thr>: group report only
            groups_out = argv[++i];
            gp.len_threshold = std::atol(argv[++i]);
            gp.window_size = std::strtoul(argv[++i], nullptr, 10);
            gp.ratio_threshold = std::strtof(argv[++i], nullptr);
        } else pos.push_back(argv[i]);
    }
    if (!pos.empty() && std::string(pos[0]) == "--list-sequences") {      // Scaffolder::load alone: what it makes of the files
        Path none;
        Scaffolder sc(none);
        for (size_t i = 1; i < pos.size(); i++) {
            if (std::getenv("TLM_POST_VIA_INDEX")) {       // scanned by one Scaffolder, read from its index by this one
                Scaffolder scan(none);
                scan.scan_only_ = true;
                const std::string idx = std::string(pos[i]) + ".index";
                if (!scan.load(pos[i]) || !scan.save_index(idx.c_str()) || !sc.load_index(idx.c_str(), pos[i])) return 1;
                std::remove(idx.c_str());
            } else if (!sc.load(pos[i])) {
                return 1;
            }
        }
        sc.materialiseAll();
        std::printf("names %zu sequences %zu\n", sc.names_.size(), sc.sequences_.size());
        for (size_t i = 0; i < sc.names_.size(); i++) {
            std::printf("%zu\t[%s]", i, sc.names_[i].c_str());
            if (i < sc.sequences_.size()) std::printf("\t[%s]", sc.sequences_[i].c_str());
            std::printf("\n");
        }
        return 0;
    }
    if (binary_dir) pos.insert(pos.begin(), const_cast<char *>(binary_dir));       // takes the place of <dump>
    if (pos.size() != 4) {
        std::cerr << "usage: telomeri_post [--len-thr N --w-size N --r-thr F] (<dump> | --binary <dir>) <reads.fa> <contigs.fa> <out.fa>" << std::endl;
        return 2;
    }
    OverlapGraph g;
    std::vector<Path> paths;
    if (binary_dir && !load_binary(binary_dir, g, paths)) return 1;
    std::ifstream in(binary_dir ? "/dev/null" : pos[0]);
    if (!in) { std::cerr << "Cannot open file: " << pos[0] << std::endl; return 1; }
    std::string line;
    bool in_unique = false;
    while (std::getline(in, line)) {
        std::istringstream is(line);
        std::string tag;
        is >> tag;
        if (tag == "N") {
            size_t n; is >> n; g.nodes_.resize(n);
        } else if (tag == "n") {
            unsigned idx, anchor, len; size_t deg; std::string name;
            is >> idx >> anchor >> len >> deg >> name;
            OverlapGraph::Node &nd = g.nodes_[idx];
            nd.index = idx; nd.anchor = anchor != 0; nd.length = len; nd.name = name; nd.edges.resize(deg);
        } else if (tag == "a") {
            unsigned owner, nbr, qs, qe, ts, te, osb, sib, esb; size_t slot; int strand;
            is >> owner >> slot >> nbr >> qs >> qe >> ts >> te >> osb >> sib >> esb >> strand;
            g.nodes_[owner].edges[slot] = OverlapGraph::Edge{nbr, owner, qs, qe, ts, te, bits_to_float(osb), bits_to_float(sib),
                                                              bits_to_float(esb), strand != 0};
        } else if (tag == "U") {
            in_unique = true;
        } else if (tag == "K" || tag == "B") {
            in_unique = false;
        } else if (tag == "p" && in_unique) {
            long length; size_t n;
            is >> length >> n;
            Path p;
            std::vector<unsigned> nodes(n), slots(n ? n - 1 : 0);
            for (auto &x : nodes) is >> x;
            for (auto &x : slots) is >> x;
            for (size_t k = 0; k < n; k++) {
                p.nodes_.push_back(&g.nodes_[nodes[k]]);
                if (k + 1 < n) p.edges_.push_back(&g.nodes_[nodes[k]].edges[slots[k]]);
            }
            p.updateLength();
            if (p.length() != length) { std::cerr << "dump: path length mismatch" << std::endl; return 1; }
            paths.push_back(std::move(p));
        }
    }
    pm.adoptPaths(std::move(paths));
    std::cout << pm.stats() << std::endl;
    // PathManager::getPathsBetweenAnchors (PathManager.cpp:626-645) on the host: this tool has no device
    std::map<PathManager::AnchorPair, std::vector<const Path *>> pba;
    for (const Path &p : pm.paths()) pba[{p.nodes_.front(), p.nodes_.back()}].push_back(&p);
    if (groups_out) {       // the same report as `ref_dump --groups` (oracle/ref_dump.cpp)
        FILE *out = std::fopen(groups_out, "w");
        if (!out) { std::perror(groups_out); return 1; }
        std::fprintf(out, "G %ld %lu %.9g\n", gp.len_threshold, gp.window_size, (double) gp.ratio_threshold);
        for (auto &kv : pba) {
            std::vector<const Path *> v = kv.second;
            std::vector<PathGroup> pgs = PathManager::constructGroups(v, gp);
            std::fprintf(out, "q %u %u %zu\n", kv.first.first->index, kv.first.second->index, pgs.size());
            for (PathGroup &pg : pgs) {
                const size_t before = pg.pig_.size();
                pg.discardNotFrequent();
                pg.calculateConsensusPath();
                pg.calculateValidPathNumber();
                std::fprintf(out, "g %zu %zu %ld %d %ld %ld\n", before, pg.pig_.size(), pg.consensus ? pg.consensus->length() : -1L,
                             pg.valid_path_number, pg.pig_.front()->length(), pg.pig_.back()->length());
            }
        }
        std::fclose(out);
        return 0;
    }
    if (std::getenv("TLM_POST_VIA_INDEX")) {
        // the sequence files as telomeri_hip gets them from its helper process: scanned by one Scaffolder, whose index
        // (Scaffolder::save_index) another one reads (load_index)
        static const Path no_path;
        static Scaffolder tables(no_path);
        auto loaded = [&]() -> Scaffolder * {
            for (int k = 1; k <= 2; k++) {
                Scaffolder scan(no_path);
                scan.scan_only_ = true;
                const std::string idx = std::string(pos[3]) + ".index" + std::to_string(k);
                if (!scan.load(pos[k]) || !scan.save_index(idx.c_str())) return nullptr;
                const bool ok = tables.load_index(idx.c_str(), pos[k]);
                std::remove(idx.c_str());
                if (!ok) return nullptr;
            }
            return &tables;
        };
        return run_post_stages(pm, pba, pos[1], pos[2], pos[3], loaded, false);
    }
    return run_post_stages(pm, pba, pos[1], pos[2], pos[3], nullptr, false);
}
