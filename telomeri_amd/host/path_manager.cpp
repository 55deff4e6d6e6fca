// telomeri_amd/host/path_manager.cpp -- Path and PathManager over the C ABI.
#include "telomeri_host.hpp"

#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <sstream>

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
    uint64_t n_paths = 0, n_nodes = 0, n_units = 0;
    tlm_pathset_sizes(ps, &n_paths, &n_nodes, &n_units);
    std::vector<uint64_t> unit(n_paths), off(n_paths + 1);
    std::vector<int64_t> length(n_paths);
    std::vector<uint32_t> nodes(n_nodes), slots(n_nodes);
    if (tlm_pathset_export(ps, 0, unit.data(), length.data(), off.data(), nodes.data(), slots.data()) != TLM_OK) {
        std::cerr << "telomeri_hip: export failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    tlm_pathset_destroy(ps);
    // materialise Path objects in unit order = the reference's append order (PathManager.cpp:154,226,382)
    paths_.reserve(paths_.size() + n_paths);
    for (uint64_t i = 0; i < n_paths; i++) {
        Path p;
        const uint64_t a = off[i], b = off[i + 1];
        p.nodes_.reserve(b - a);
        p.edges_.reserve(b - a - 1);
        for (uint64_t k = a; k < b; k++) {
            const OverlapGraph::Node *n = &g.nodes_[nodes[k]];
            p.nodes_.push_back(n);
            if (k + 1 < b) p.edges_.push_back(&n->edges[slots[k]]);
        }
        p.setLength((long) length[i]);
        paths_.push_back(std::move(p));
    }
    last_counters = Counters{c.units, c.steps, c.evals, c.draws, c.accepted, c.kernel_ms};
    std::cout << "Found " << n_paths << " paths." << std::endl;      // PathManager.cpp:172,387
}

void PathManager::buildMonteCarlo(const OverlapGraph &g, const Utils::Metrics &metric) {
    std::cout << "> Monte Carlo heuristic: " << Utils::getMetricName(metric) << std::endl;      // :16
    build(g, metric, true);
}

void PathManager::buildDeterministic(const OverlapGraph &g, const Utils::Metrics &metric) {
    std::cout << "> Deterministic heuristic: " << Utils::getMetricName(metric) << std::endl;    // :183
    build(g, metric, false);
}

void PathManager::filterUnique() {      // PathManager.cpp:432-438: ADJACENT duplicates only
    paths_.erase(std::unique(paths_.begin(), paths_.end()), paths_.end());
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
PathManager::getPathsBetweenAnchors() {  // PathManager.cpp:626-645: buckets by (front, back) in paths_ order
    std::map<std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *>, std::vector<const Path *>> out;
    for (const Path &p : paths_) out[{p.nodes_.front(), p.nodes_.back()}].push_back(&p);
    return out;
}
