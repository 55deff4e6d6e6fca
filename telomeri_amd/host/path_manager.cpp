// telomeri_amd/host/path_manager.cpp -- Path and PathManager over the C ABI.
#include "telomeri_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <thread>

#include "../../include/telomeri_hip.h"

// ---- Path -----------------------------------------------------------------------------------------

void Path::updateLength() {             // Path.cpp:6-16: first node's length + every edge's increment but the last
    if (nodes_.empty()) {
        length_ = 0;
        return;
    }
    length_ = nodes_[0]->length;
    for (size_t i = 1; i < edges_.size(); i++) length_ += (long) edges_[i - 1]->q_start - (long) edges_[i - 1]->t_start;
}

bool Path::operator==(const Path &o) const {
    return nodes_.size() == o.nodes_.size() && std::equal(nodes_.begin(), nodes_.end(), o.nodes_.begin());
}

std::ostream &operator<<(std::ostream &s, const Path &p) {      // "*n3-n17-...": Path.cpp:24-43
    for (size_t i = 0; i < p.nodes_.size(); i++) {
        if (i) s << "-";
        if (p.nodes_[i]->anchor) s << '*';
        s << 'n' << p.nodes_[i]->index;
    }
    return s;
}

// ---- PathManager ----------------------------------------------------------------------------------

void PathManager::build(const OverlapGraph &g, const Utils::Metrics &metric, bool mc) {
    tlm_ctx *ctx = OverlapGraph::context();
    tlm_graph *dg = g.device_graph();
    if (!dg) {
        std::cerr << "PathManager: the graph has not been loaded" << std::endl;
        std::exit(1);
    }
    const int m = (int) metric;          // Utils::Metrics and TLM_METRIC_* share their numbering
    tlm_walk_params wp{params_.rebuild_attempts, params_.backtrack_attempts, params_.len_threshold};
    tlm_pathset *ps = nullptr;
    tlm_counters c{};
    const int rc = mc ? tlm_walk_mc(ctx, dg, m, &wp, 0, tlm_mc_units(dg, &wp), &ps, &c)
                      : tlm_walk_det(ctx, dg, m, 0, tlm_det_units(dg), &ps, &c);
    if (rc != TLM_OK) {
        std::cerr << "telomeri_hip: walk failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    graph_ = &g;
    uint64_t n_paths = 0;
    tlm_pathset_sizes(ps, &n_paths, nullptr, nullptr);
    append_from(g, ps);
    dev_sets_.push_back(ps);             // stays on the device for filterUnique / getPathsBetweenAnchors
    last_counters = Counters{c.units, c.steps, c.evals, c.draws, c.accepted, c.kernel_ms};
    std::cout << "Found " << n_paths << " paths." << std::endl;      // PathManager.cpp:172,387
}

PathManager::~PathManager() {
    for (tlm_pathset *ps : dev_sets_) tlm_pathset_destroy(ps);
}

void PathManager::append_from(const OverlapGraph &g, tlm_pathset *ps) {
    uint64_t n_paths = 0, n_nodes = 0;
    tlm_pathset_sizes(ps, &n_paths, &n_nodes, nullptr);
    std::vector<uint64_t> off(n_paths + 1);
    std::vector<int64_t> length(n_paths);
    std::vector<uint32_t> nodes(n_nodes), slots(n_nodes);
    if (tlm_pathset_export(ps, 0, nullptr, length.data(), off.data(), nodes.data(), slots.data()) != TLM_OK) {
        std::cerr << "telomeri_hip: export failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    // Path objects in set order = the reference's append order (PathManager.cpp:154,226,382); the pointer chasing
    // into nodes_[..].edges is spread over the host cores
    const size_t base = paths_.size();
    paths_.resize(base + n_paths);
    unsigned nt = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if (n_paths < 256) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++)
        th.emplace_back([&, t] {
            for (uint64_t i = n_paths * t / nt; i < n_paths * (t + 1) / nt; i++) {
                Path &p = paths_[base + i];
                const uint64_t a = off[i], b = off[i + 1];
                p.nodes_.resize(b - a);
                p.edges_.resize(b - a - 1);
                for (uint64_t k = a; k < b; k++) {
                    const OverlapGraph::Node *n = &g.nodes_[nodes[k]];
                    p.nodes_[k - a] = n;
                    if (k + 1 < b) p.edges_[k - a] = &n->edges[slots[k]];
                }
                p.setLength((long) length[i]);
            }
        });
    for (auto &x : th) x.join();
}

tlm_pathset *PathManager::single_set() {
    if (dev_sets_.empty()) return nullptr;
    if (dev_sets_.size() > 1) {
        tlm_pathset *all = nullptr;
        if (tlm_pathset_concat(OverlapGraph::context(), (int) dev_sets_.size(), dev_sets_.data(), &all) != TLM_OK) {
            std::cerr << "telomeri_hip: concat failed: " << tlm_last_error() << std::endl;
            std::exit(1);
        }
        for (tlm_pathset *ps : dev_sets_) tlm_pathset_destroy(ps);
        dev_sets_.assign(1, all);
    }
    return dev_sets_[0];
}

void PathManager::buildMonteCarlo(const OverlapGraph &g, const Utils::Metrics &metric) {
    std::cout << "> Monte Carlo heuristic: " << Utils::getMetricName(metric) << std::endl;      // :16
    build(g, metric, true);
}

void PathManager::buildDeterministic(const OverlapGraph &g, const Utils::Metrics &metric) {
    std::cout << "> Deterministic heuristic: " << Utils::getMetricName(metric) << std::endl;    // :183
    build(g, metric, false);
}

void PathManager::filterUnique() {      // PathManager.cpp:432-438: ADJACENT duplicates only; on the device
    tlm_pathset *all = single_set();
    if (!all) return;
    tlm_pathset *uniq = nullptr;
    if (tlm_pathset_filter_unique(OverlapGraph::context(), all, &uniq) != TLM_OK) {
        std::cerr << "telomeri_hip: filterUnique failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    tlm_pathset_destroy(all);
    dev_sets_.assign(1, uniq);
    paths_.clear();
    append_from(*graph_, uniq);
}

std::tuple<ulong, ulong, ulong> PathManager::getMinMaxSumPathLength() {
    ulong min_len = ULONG_MAX, max_len = 0, sum_len = 0;
    for (const Path &p : paths_) {
        const ulong l = (ulong) p.length();
        sum_len += l;
        min_len = std::min(min_len, l);
        max_len = std::max(max_len, l);
    }
    return std::make_tuple(min_len, max_len, sum_len);
}

std::string PathManager::stats() {      // the report of PathManager.cpp:440-465, same text
    ulong min_len, max_len, sum_len;
    std::tie(min_len, max_len, sum_len) = getMinMaxSumPathLength();
    ulong negatives = 0;
    for (const Path &p : paths_) negatives += p.length() < 0;
    std::stringstream str;
    str << "Paths" << '\n'
        << "- total_num: " << paths_.size() << '\n' << "-   min_len: " << min_len << '\n'
        << "-   max_len: " << max_len << '\n' << "-   avg_len: " << (paths_.empty() ? 0 : sum_len / paths_.size())
        << std::endl;
    if (negatives > 0) str << "- negatives: " << negatives << std::endl;
    return str.str();
}

std::map<std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *>, std::vector<const Path *>>
PathManager::getPathsBetweenAnchors() {  // PathManager.cpp:626-645: buckets by (front, back), each in paths_ order
    std::map<std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *>, std::vector<const Path *>> out;
    tlm_pathset *all = single_set();
    if (!all) return out;
    std::vector<uint32_t> order(paths_.size());
    std::vector<uint64_t> key(paths_.size());
    if (tlm_pathset_anchor_pairs(OverlapGraph::context(), all, order.data(), key.data()) != TLM_OK) {
        std::cerr << "telomeri_hip: getPathsBetweenAnchors failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    for (size_t j = 0; j < order.size(); j++) {      // keys ascending = the map's (Node*, Node*) order
        const Path &p = paths_[order[j]];
        out[{p.nodes_.front(), p.nodes_.back()}].push_back(&p);
    }
    return out;
}
