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

static float bits_to_float(unsigned u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char **argv) {
    PathManager pm;
    PathManager::Parameters gp = pm.params_;
    const char *groups_out = nullptr;
    std::vector<char *> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--len-thr" && i + 1 < argc) pm.params_.len_threshold = std::atol(argv[++i]);
        else if (a == "--w-size" && i + 1 < argc) pm.params_.window_size = std::strtoul(argv[++i], nullptr, 10);
        else if (a == "--r-thr" && i + 1 < argc) pm.params_.ratio_threshold = std::strtof(argv[++i], nullptr);
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
    if (pos.size() != 4) {
        std::cerr << "usage: telomeri_post [--len-thr N --w-size N --r-thr F] <dump> <reads.fa> <contigs.fa> <out.fa>" << std::endl;
        return 2;
    }
    std::ifstream in(pos[0]);
    if (!in) { std::cerr << "Cannot open file: " << pos[0] << std::endl; return 1; }
    OverlapGraph g;
    std::vector<Path> paths;
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
