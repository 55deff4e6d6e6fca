// oracle/ref_dump.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Dump driver linked against the UNMODIFIED reference sources where they lie under
// /root/reference (never copied into this repo; see oracle/Makefile target `ref`).
// It drives the reference's own public API for the hot path and writes what it built as text:
//   * OverlapGraph::load(char*, bool)                     (reference include/OverlapGraph.hpp:90)
//   * PathManager::buildMonteCarlo / buildDeterministic   (reference include/PathManager.hpp:16-19)
//   * PathManager::getPathsBetweenAnchors                 (reference include/PathManager.hpp:55-57)
// `paths_` is private in the reference; every element is reachable through
// getPathsBetweenAnchors() as a `const Path*` INTO paths_, so sorting those pointers by
// address recovers the append order of paths_ (a std::vector is contiguous).
// An edge is reported as its slot inside the owner's list: the reference stores
// `const Edge*` into `g.nodes_[owner].edges`, so slot = pointer - &edges[0].
//
// Output format (one record per line, all integers decimal, floats as IEEE-754 bit patterns):
//   TLMDUMP 1
//   N <nodes>
//   n <index> <anchor> <length> <degree> <name>
//   A <adjacency entries>
//   a <owner> <slot> <nbr> <q_start> <q_end> <t_start> <t_end> <os_bits> <si_bits> <es_bits> <strand>
//   B <builder id> <accepted paths>          builder id: 0 MC/ES 1 MC/OS 2 DET/ES 3 DET/OS
//                                            (+4: the *_SQRT metric of the same builder)
//   p <length> <n_nodes> <node ids...> <edge slots...>
//   U <paths>                                paths_ of ONE PathManager after the four default builders (Main.cpp:237-243)
//                                            and filterUnique (Main.cpp:247), in paths_ order; followed by `p` lines
//   K <pairs>                                getPathsBetweenAnchors (Main.cpp:251-252) in map order
//   k <front node> <back node> <count> <indices into the U list...>
//   G <len_thr> <w_size> <r_thr>             (--groups) PathManager::constructGroups + the PathGroup members with THESE
//                                            parameters on each pair's paths (independent of the walk parameters)
//   q <front> <back> <groups>
//   g <paths> <paths after discardNotFrequent> <consensus length or -1> <valid_path_number> <first length> <last length>
#include <OverlapGraph.hpp>
#include <PathManager.hpp>
#include <Utils.hpp>

#include <algorithm>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static unsigned bits_of(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

static void dump_builder(FILE *out, const OverlapGraph &g, PathManager &pm, int builder_id) {
    auto pba = pm.getPathsBetweenAnchors();
    std::vector<const Path *> all;
    for (auto &kv : pba)
        for (const Path *p : kv.second) all.push_back(p);
    std::sort(all.begin(), all.end());  // address order == paths_ order
    std::fprintf(out, "B %d %zu\n", builder_id, all.size());
    for (const Path *p : all) {
        std::fprintf(out, "p %ld %zu", p->length(), p->nodes_.size());
        for (const OverlapGraph::Node *n : p->nodes_) std::fprintf(out, " %u", n->index);
        for (const OverlapGraph::Edge *e : p->edges_) {
            const OverlapGraph::Node &owner = g.nodes_[e->t_index];
            std::fprintf(out, " %ld", (long) (e - owner.edges.data()));
        }
        std::fprintf(out, "\n");
    }
}

int main(int argc, char **argv) {
    OverlapGraph::FilterParameters fp = {OverlapGraph::AVG, 0, 0.0f, ULONG_MAX, 1.0f};   // Main.cpp:56-62
    PathManager::Parameters pp = {2000, 60, 4700000ul, 1000ul, 0.9f};                    // Main.cpp:63-69
    bool want_graph = true, want_paths = true, want_sqrt = false, want_groups = false;
    PathManager::Parameters gp = {2000, 60, 4700000ul, 1000ul, 0.9f};
    std::vector<char *> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> char * {
            if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--filter-avg") fp.mode = OverlapGraph::AVG;
        else if (a == "--filter-min") fp.mode = OverlapGraph::MIN;
        else if (a == "--filter-max") fp.mode = OverlapGraph::MAX;
        else if (a == "--filter-sum") fp.mode = OverlapGraph::SUM;
        else if (a == "--min-oll") fp.min_overlap_length = std::strtoul(next(), nullptr, 10);
        else if (a == "--min-olp") fp.min_overlap_percentage = std::strtof(next(), nullptr);
        else if (a == "--max-ohl") fp.max_overhang_length = std::strtoul(next(), nullptr, 10);
        else if (a == "--max-ohp") fp.max_overhang_percentage = std::strtof(next(), nullptr);
        else if (a == "--rb-att") pp.rebuild_attempts = std::atoi(next());
        else if (a == "--bt-att") pp.backtrack_attempts = std::atoi(next());
        else if (a == "--len-thr") pp.len_threshold = std::atol(next());
        else if (a == "--no-graph") want_graph = false;
        else if (a == "--no-paths") want_paths = false;
        else if (a == "--sqrt") want_sqrt = true;
        else if (a == "--groups") {
            want_groups = true;
            gp.len_threshold = std::atol(next());
            gp.window_size = std::strtoul(next(), nullptr, 10);
            gp.ratio_threshold = std::strtof(next(), nullptr);
        }
        else pos.push_back(argv[i]);
    }
    if (pos.size() != 3) {
        std::fprintf(stderr, "usage: ref_dump [options] <rr.paf> <cr.paf> <out.dump>\n");
        return 2;
    }

    OverlapGraph g;
    g.filter_params_ = fp;
    if (!g.load(pos[0], false) || !g.load(pos[1], true)) return 1;   // Main.cpp:223

    FILE *out = std::fopen(pos[2], "w");
    if (!out) { std::perror(pos[2]); return 1; }
    std::fprintf(out, "TLMDUMP 1\n");
    std::fprintf(out, "N %zu\n", g.nodes_.size());
    size_t adj = 0;
    for (const OverlapGraph::Node &n : g.nodes_) {
        std::fprintf(out, "n %u %d %u %zu %s\n", n.index, n.anchor ? 1 : 0, n.length, n.edges.size(), n.name.c_str());
        adj += n.edges.size();
    }
    std::fprintf(out, "A %zu\n", adj);
    if (want_graph) {
        for (const OverlapGraph::Node &n : g.nodes_) {
            for (size_t s = 0; s < n.edges.size(); s++) {
                const OverlapGraph::Edge &e = n.edges[s];
                std::fprintf(out, "a %u %zu %u %u %u %u %u %u %u %u %d\n", e.t_index, s, e.q_index,
                             e.q_start, e.q_end, e.t_start, e.t_end, bits_of(e.overlap_score),
                             bits_of(e.sequence_identity), bits_of(e.extension_score), e.relative_strand ? 1 : 0);
            }
        }
    }
    if (want_paths) {
        struct B { int id; bool mc; Utils::Metrics m; };
        std::vector<B> bs = {{0, true, Utils::Metrics::EXTENSION_SCORE}, {1, true, Utils::Metrics::OVERLAP_SCORE},
                             {2, false, Utils::Metrics::EXTENSION_SCORE}, {3, false, Utils::Metrics::OVERLAP_SCORE}};
        if (want_sqrt) {
            bs.push_back({4, true, Utils::Metrics::EXTENSION_SCORE_SQRT});
            bs.push_back({5, true, Utils::Metrics::OVERLAP_SCORE_SQRT});
            bs.push_back({6, false, Utils::Metrics::EXTENSION_SCORE_SQRT});
            bs.push_back({7, false, Utils::Metrics::OVERLAP_SCORE_SQRT});
        }
        for (const B &b : bs) {
            // One PathManager per builder call: each call owns a fresh default-seeded mt19937
            // (PathManager.cpp:13), so this equals the appended segments of Main.cpp:237-243.
            PathManager pm;
            pm.params_ = pp;
            if (b.mc) pm.buildMonteCarlo(g, b.m);
            else pm.buildDeterministic(g, b.m);
            dump_builder(out, g, pm, b.id);
        }
    }
    if (want_paths) {
        // the pipeline of Main.cpp:230-252 on one PathManager
        PathManager pm;
        pm.params_ = pp;
        pm.buildMonteCarlo(g, Utils::Metrics::EXTENSION_SCORE);
        pm.buildMonteCarlo(g, Utils::Metrics::OVERLAP_SCORE);
        pm.buildDeterministic(g, Utils::Metrics::EXTENSION_SCORE);
        pm.buildDeterministic(g, Utils::Metrics::OVERLAP_SCORE);
        pm.filterUnique();
        auto pba = pm.getPathsBetweenAnchors();
        std::vector<const Path *> all;
        for (auto &kv : pba)
            for (const Path *p : kv.second) all.push_back(p);
        std::sort(all.begin(), all.end());
        std::fprintf(out, "U %zu\n", all.size());
        for (const Path *p : all) {
            std::fprintf(out, "p %ld %zu", p->length(), p->nodes_.size());
            for (const OverlapGraph::Node *n : p->nodes_) std::fprintf(out, " %u", n->index);
            for (const OverlapGraph::Edge *e : p->edges_) std::fprintf(out, " %ld", (long) (e - g.nodes_[e->t_index].edges.data()));
            std::fprintf(out, "\n");
        }
        std::fprintf(out, "K %zu\n", pba.size());
        for (auto &kv : pba) {
            std::fprintf(out, "k %u %u %zu", kv.first.first->index, kv.first.second->index, kv.second.size());
            for (const Path *p : kv.second)
                std::fprintf(out, " %ld", (long) (std::lower_bound(all.begin(), all.end(), p) - all.begin()));
            std::fprintf(out, "\n");
        }
        if (want_groups) {
            std::fprintf(out, "G %ld %lu %.9g\n", gp.len_threshold, gp.window_size, (double) gp.ratio_threshold);
            for (auto &kv : pba) {
                std::vector<const Path *> v = kv.second;      // constructGroups sorts its argument
                std::vector<PathGroup> pgs = PathManager::constructGroups(v, gp);
                std::fprintf(out, "q %u %u %zu\n", kv.first.first->index, kv.first.second->index, pgs.size());
                for (PathGroup &pg : pgs) {
                    const size_t before = pg.pig_.size();
                    pg.discardNotFrequent();
                    pg.calculateConsensusPath();
                    pg.calculateValidPathNumber();
                    std::fprintf(out, "g %zu %zu %ld %d %ld %ld\n", before, pg.pig_.size(),
                                 pg.consensus ? pg.consensus->length() : -1L, pg.valid_path_number,
                                 pg.pig_.front()->length(), pg.pig_.back()->length());
                }
            }
        }
    }
    std::fclose(out);
    return 0;
}
