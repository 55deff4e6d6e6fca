// telomeri_amd/host/path_manager.cpp -- Path and PathManager over the C ABI.
#include "telomeri_host.hpp"

#include <chrono>
#include <cstdio>
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

bool Path::operator==(const Path &o) const {       // Path.cpp:46-49: the node sequences are equal (edges are not compared)
    if (this == &o) return true;                    // (a group's consensus is compared with every member, itself included)
    if (node_count() != o.node_count()) return false;
    // two paths of device sets: a different hash of the node sequence settles it without fetching either list
    if (owner_ && o.owner_ && hash_ != o.hash_) return false;
    if (!have_nodes_) owner_->materialise(*this);
    if (!o.have_nodes_) o.owner_->materialise(o);
    return std::equal(nodes_.begin(), nodes_.end(), o.nodes_.begin());
}

size_t Path::count_equal(const std::vector<const Path *> &members) const {
    size_t n = 0;
    std::vector<const Path *> on_device;        // same manager, same count and hash, node lists still on the device
    for (const Path *p : members) {
        if (p == this) { n++; continue; }
        if (p->node_count() != node_count()) continue;
        if (owner_ && p->owner_ && hash_ != p->hash_) continue;
        if (owner_ && p->owner_ == owner_ && !p->have_nodes_) on_device.push_back(p);
        else if (*this == *p) n++;
    }
    if (!on_device.empty()) n += owner_->count_equal_on_device(*this, on_device);
    return n;
}

std::ostream &operator<<(std::ostream &s, const Path &p) {      // "*n3-n17-...": Path.cpp:24-43
    if (!p.have_nodes_) p.owner_->materialise(p);
    for (size_t i = 0; i < p.nodes_.size(); i++) {
        if (i) s << "-";
        if (p.nodes_[i]->anchor) s << '*';
        s << 'n' << p.nodes_[i]->index;
    }
    return s;
}

// ---- PathManager ----------------------------------------------------------------------------------

static void die_tlm(const char *what) {
    std::cerr << "telomeri_hip: " << what << " failed: " << tlm_last_error() << std::endl;
    std::exit(1);
}

// The one exchange of a multi-GPU job (no-op in a one-process job): every rank's builder sets -> paths_ of the whole
// job on every rank, in the reference's order (tlm_pathset_allgather).  Call after the builders, before filterUnique.
void PathManager::gather() {
    tlm_ctx *ctx = OverlapGraph::context();
    // (TLM_FORCE_GATHER: a one-process job goes through the gathered representation too -- what tests on one GPU can reach)
    if ((tlm_dist_world(ctx) <= 1 && !std::getenv("TLM_FORCE_GATHER")) || dev_sets_.empty()) return;
    const char *mode = std::getenv("TLM_EXCHANGE");
    tlm_exchange_info xi{};
    if (mode && std::string(mode) == "full") {      // every node list to every rank (tlm_pathset_allgather)
        tlm_pathset *all = nullptr;
        if (tlm_pathset_allgather(ctx, (int) dev_sets_.size(), dev_sets_.data(), &all, &xi) != TLM_OK) die_tlm("allgather");
        for (tlm_pathset *ps : dev_sets_) tlm_pathset_destroy(ps);
        dev_sets_.assign(1, all);
        paths_.clear();
        edge_pool_.clear();
        append_from(*graph_, all);
        if (!packed_) materialiseAll();
    } else {            // summaries to every rank, node lists on demand: every rank goes on to run the same host stages
        if (tlm_pathset_gather_summaries(ctx, (int) dev_sets_.size(), dev_sets_.data(), &gathered_, &xi) != TLM_OK)
            die_tlm("gather_summaries");
        for (tlm_pathset *ps : dev_sets_) tlm_pathset_destroy(ps);
        dev_sets_.clear();
        uint64_t n = 0;
        tlm_gathered_sizes(gathered_, &n, nullptr);
        std::vector<int64_t> length(n);
        std::vector<uint32_t> front(n), back(n), count(n);
        std::vector<uint64_t> hash(n);
        gathered_dup_.assign(n, 0);
        gathered_block_first_.assign(n, 0);
        if (tlm_gathered_export(gathered_, nullptr, length.data(), front.data(), back.data(), count.data(), hash.data(), nullptr,
                                gathered_dup_.data(), gathered_block_first_.data()) != TLM_OK)
            die_tlm("gathered export");
        paths_.clear();
        edge_pool_.clear();
        paths_.resize(n);
        for (uint64_t i = 0; i < n; i++) {
            Path &p = paths_[i];
            p.owner_ = this;
            p.index_ = i;
            p.hash_ = hash[i];
            p.front_ = front[i];
            p.back_ = back[i];
            p.n_nodes_ = count[i];
            p.have_nodes_ = false;
            p.setLength((long) length[i]);
        }
        if (!packed_) materialiseAll();
    }
    std::cout << "Gathered " << paths_.size() << " paths from " << tlm_dist_world(ctx) << " ranks (" << xi.bytes_sent
              << " bytes sent, " << xi.device_ms << " ms)." << std::endl;
}

// Which units a rank of a multi-process job walks.  "sliced" (default): of EVERY anchor the rank's slice of the trials /
// first edges (tlm_dist_unit_share: balanced whatever the anchors weigh); "contiguous" (TLM_PARTITION=contiguous): round
// 3's bundles of whole anchors (tlm_dist_anchor_range).  The exchange returns paths_ order either way.
static bool partition_is_sliced() {
    const char *e = std::getenv("TLM_PARTITION");
    return !(e && std::string(e) == "contiguous");
}
static std::vector<uint64_t> unit_share(tlm_ctx *ctx, tlm_graph *dg, const tlm_walk_params &wp, bool det) {
    uint64_t n = 0;
    if (tlm_dist_unit_share(dg, &wp, tlm_dist_rank(ctx), tlm_dist_world(ctx), det ? 1 : 0, nullptr, &n) != TLM_OK) die_tlm("unit share");
    std::vector<uint64_t> u(n);
    if (tlm_dist_unit_share(dg, &wp, tlm_dist_rank(ctx), tlm_dist_world(ctx), det ? 1 : 0, u.data(), &n) != TLM_OK) die_tlm("unit share");
    return u;
}

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
    // this rank's share of the units: the anchors of its bundle (all of them in a one-process job)
    uint64_t u0 = 0, u1 = mc ? tlm_mc_units(dg, &wp) : tlm_det_units(dg);
    const bool sliced = tlm_dist_world(ctx) > 1 && partition_is_sliced();
    if (tlm_dist_world(ctx) > 1 && !sliced) {
        uint32_t a0 = 0, a1 = 0;
        if (tlm_dist_anchor_range(dg, tlm_dist_rank(ctx), tlm_dist_world(ctx), &a0, &a1) != TLM_OK) die_tlm("anchor range");
        if (mc) {
            u0 = (uint64_t) a0 * (uint64_t) std::max(params_.rebuild_attempts, 0);
            u1 = (uint64_t) a1 * (uint64_t) std::max(params_.rebuild_attempts, 0);
        } else {
            std::vector<uint64_t> det_off((size_t) g.nodes_.size() + 2);
            tlm_graph_info gi;
            tlm_graph_get_info(dg, &gi);
            det_off.resize((size_t) gi.n_anchors + 1);
            if (tlm_graph_anchor_table(dg, nullptr, det_off.data()) != TLM_OK) die_tlm("anchor table");
            u0 = det_off[a0];
            u1 = det_off[a1];
        }
    }
    int rc;
    if (sliced) {
        const std::vector<uint64_t> units = unit_share(ctx, dg, wp, !mc);
        rc = tlm_walk_list(ctx, dg, mc ? 0 : 1, m, -1, &wp, units.size(), units.data(), &ps, &c);
    } else {
        rc = mc ? tlm_walk_mc(ctx, dg, m, &wp, u0, u1, &ps, &c) : tlm_walk_det(ctx, dg, m, u0, u1, &ps, &c);
    }
    if (rc != TLM_OK) {
        std::cerr << "telomeri_hip: walk failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    graph_ = &g;
    uint64_t n_paths = 0;
    tlm_pathset_sizes(ps, &n_paths, nullptr, nullptr);
    append_from(g, ps);
    dev_sets_.push_back(ps);             // stays on the device: filterUnique, getPathsBetweenAnchors, materialise
    if (!packed_) materialiseAll();
    last_counters = Counters{c.units, c.steps, c.evals, c.draws, c.accepted, c.kernel_ms};
    std::cout << "Found " << n_paths << " paths." << std::endl;      // PathManager.cpp:172,387
}

void PathManager::buildAll(const OverlapGraph &g, const Utils::Metrics &ma, const Utils::Metrics &mb) {
    tlm_ctx *ctx = OverlapGraph::context();
    tlm_graph *dg = g.device_graph();
    if (!dg) {
        std::cerr << "PathManager: the graph has not been loaded" << std::endl;
        std::exit(1);
    }
    tlm_walk_params wp{params_.rebuild_attempts, params_.backtrack_attempts, params_.len_threshold};
    uint64_t mc0 = 0, mc1 = tlm_mc_units(dg, &wp), d0 = 0, d1 = tlm_det_units(dg);
    const bool sliced = tlm_dist_world(ctx) > 1 && partition_is_sliced();
    if (tlm_dist_world(ctx) > 1 && !sliced) {
        uint32_t a0 = 0, a1 = 0;
        if (tlm_dist_anchor_range(dg, tlm_dist_rank(ctx), tlm_dist_world(ctx), &a0, &a1) != TLM_OK) die_tlm("anchor range");
        tlm_graph_info gi;
        tlm_graph_get_info(dg, &gi);
        std::vector<uint64_t> det_off((size_t) gi.n_anchors + 1);
        if (tlm_graph_anchor_table(dg, nullptr, det_off.data()) != TLM_OK) die_tlm("anchor table");
        mc0 = (uint64_t) a0 * (uint64_t) std::max(params_.rebuild_attempts, 0);
        mc1 = (uint64_t) a1 * (uint64_t) std::max(params_.rebuild_attempts, 0);
        d0 = det_off[a0];
        d1 = det_off[a1];
    }
    tlm_pathset *sets[4] = {nullptr, nullptr, nullptr, nullptr};
    tlm_counters c[4];
    const auto t_call = std::chrono::steady_clock::now();
    if (sliced) {
        const std::vector<uint64_t> mcu = unit_share(ctx, dg, wp, false), detu = unit_share(ctx, dg, wp, true);
        if (tlm_walk_builders_list(ctx, dg, (int) ma, (int) mb, &wp, mcu.size(), mcu.data(), detu.size(), detu.data(), sets, c) != TLM_OK)
            die_tlm("walk");
    } else if (tlm_walk_builders(ctx, dg, (int) ma, (int) mb, &wp, mc0, mc1, d0, d1, sets, c) != TLM_OK) die_tlm("walk");
    const double call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    graph_ = &g;
    last_counters = Counters{0, 0, 0, 0, 0, 0.0};
    const Utils::Metrics metric_of[4] = {ma, mb, ma, mb};
    {       // room for the four sets at once (grown set by set, the Path objects of the earlier sets were moved three times)
        uint64_t all = 0;
        for (int k = 0; k < 4; k++) {
            uint64_t n = 0;
            tlm_pathset_sizes(sets[k], &n, nullptr, nullptr);
            all += n;
        }
        paths_.reserve(paths_.size() + all);
    }
    for (int k = 0; k < 4; k++) {
        std::cout << (k < 2 ? "> Monte Carlo heuristic: " : "> Deterministic heuristic: ") << Utils::getMetricName(metric_of[k])
                  << std::endl;                                                 // PathManager.cpp:16,183
        uint64_t n_paths = 0;
        tlm_pathset_sizes(sets[k], &n_paths, nullptr, nullptr);
        append_from(g, sets[k]);
        dev_sets_.push_back(sets[k]);
        std::cout << "Found " << n_paths << " paths." << std::endl;            // :172,387
        last_counters.units += c[k].units; last_counters.steps += c[k].steps; last_counters.evals += c[k].evals;
        last_counters.draws += c[k].draws; last_counters.accepted += c[k].accepted;
        last_counters.kernel_ms += k < 2 ? c[k].kernel_ms : c[k].kernel_busy_ms;
    }
    if (std::getenv("TLM_HOST_TIMING"))
        std::fprintf(stderr, "[host timing] builders: the device call %.0f ms (Monte-Carlo launch %.0f ms; deterministic launch %.0f ms from "
                             "queueing to its end, %.0f ms of it running), then %.0f ms for the paths' views\n", call_ms, c[0].kernel_ms,
                     c[2].kernel_ms, c[2].kernel_busy_ms,
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count() - call_ms);
    // the builders of main() are over (Main.cpp:237-243 is the only place that walks): what the job allocates from here on --
    // paths_ as one set, its filtered copy -- comes out of the workspaces' memory instead of on top of it
    if (tlm_ctx_release_workspaces(ctx) != TLM_OK) die_tlm("release workspaces");
    if (!packed_) materialiseAll();
}

bool telomeri_host_lean_default();      // overlap_graph.cpp: TLM_HOST_LEAN
PathManager::PathManager() : packed_(telomeri_host_lean_default()) {}

PathManager::~PathManager() {
    for (tlm_pathset *ps : dev_sets_) tlm_pathset_destroy(ps);
    if (gathered_) tlm_gathered_destroy(gathered_);
}

void PathManager::append_from(const OverlapGraph &g, tlm_pathset *ps) {
    // Path objects in set order = the reference's append order (PathManager.cpp:154,226,382).  Only the per-path
    // summary crosses PCIe here (36 bytes a path); the node and slot lists stay in the device set.
    uint64_t n_paths = 0;
    tlm_pathset_sizes(ps, &n_paths, nullptr, nullptr);
    std::vector<int64_t> length(n_paths);
    std::vector<uint32_t> front(n_paths), back(n_paths), count(n_paths);
    std::vector<uint64_t> hash(n_paths);
    if (tlm_pathset_export(ps, 0, nullptr, length.data(), nullptr, nullptr, nullptr) != TLM_OK) die_tlm("export");
    if (tlm_pathset_summary(OverlapGraph::context(), ps, front.data(), back.data(), count.data(), hash.data()) != TLM_OK)
        die_tlm("summary");
    const size_t base = paths_.size();
    paths_.resize(base + n_paths);
    for (uint64_t i = 0; i < n_paths; i++) {
        Path &p = paths_[base + i];
        p.owner_ = this;
        p.index_ = base + i;
        p.hash_ = hash[i];
        p.front_ = front[i];
        p.back_ = back[i];
        p.n_nodes_ = count[i];
        p.have_nodes_ = false;
        p.setLength((long) length[i]);
    }
    (void) g;
}

void PathManager::materialise(const Path &p) const { materialise(std::vector<const Path *>{&p}); }

// node and slot lists of the paths `v` (all of this manager, none materialised yet), concatenated in that order
void PathManager::fetch_lists(const std::vector<const Path *> &v, std::vector<uint32_t> &nodes, std::vector<uint32_t> &slots) const {
    uint64_t total = 0;
    for (const Path *p : v) total += p->n_nodes_;
    nodes.assign(total, 0);
    slots.assign(total, 0);
    if (v.empty()) return;
    if (gathered_) {        // from the owning ranks: a collective, every rank asks for the same paths at the same moment
        std::vector<uint64_t> idx(v.size());
        for (size_t j = 0; j < v.size(); j++) idx[j] = v[j]->index_;
        if (tlm_gathered_fetch(OverlapGraph::context(), gathered_, idx.size(), idx.data(), nodes.data(), slots.data()) != TLM_OK)
            die_tlm("gathered fetch");
        return;
    }
    // from the device sets (paths_ = the sets one after the other): one fetch per set, results back in request order
    std::vector<uint64_t> set_end;
    uint64_t acc = 0;
    for (tlm_pathset *ps : dev_sets_) {
        uint64_t n = 0;
        tlm_pathset_sizes(ps, &n, nullptr, nullptr);
        set_end.push_back(acc += n);
    }
    std::vector<uint64_t> at(v.size() + 1, 0);
    for (size_t j = 0; j < v.size(); j++) at[j + 1] = at[j] + v[j]->n_nodes_;
    std::vector<std::vector<size_t>> by_set(dev_sets_.size());
    for (size_t j = 0; j < v.size(); j++) {
        const size_t k = std::upper_bound(set_end.begin(), set_end.end(), v[j]->index_) - set_end.begin();
        if (k >= dev_sets_.size()) {
            std::cerr << "PathManager: path " << v[j]->index_ << " is not in a device set" << std::endl;
            std::exit(1);
        }
        by_set[k].push_back(j);
    }
    for (size_t k = 0; k < by_set.size(); k++) {
        if (by_set[k].empty()) continue;
        const uint64_t first = k ? set_end[k - 1] : 0;
        std::vector<uint64_t> idx;
        uint64_t sub = 0;
        for (size_t j : by_set[k]) {
            idx.push_back(v[j]->index_ - first);
            sub += v[j]->n_nodes_;
        }
        std::vector<uint32_t> sn(sub), ss(sub);
        if (tlm_pathset_fetch(OverlapGraph::context(), dev_sets_[k], idx.size(), idx.data(), sn.data(), ss.data()) != TLM_OK)
            die_tlm("fetch");
        uint64_t a = 0;
        for (size_t j : by_set[k]) {
            std::copy(sn.begin() + (long) a, sn.begin() + (long) (a + v[j]->n_nodes_), nodes.begin() + (long) at[j]);
            std::copy(ss.begin() + (long) a, ss.begin() + (long) (a + v[j]->n_nodes_), slots.begin() + (long) at[j]);
            a += v[j]->n_nodes_;
        }
    }
}

size_t PathManager::count_equal_on_device(const Path &ref, const std::vector<const Path *> &candidates) const {
    // one request: the reference path first (unless its nodes are here already), then the candidates
    std::vector<const Path *> v;
    if (!ref.have_nodes_) v.push_back(&ref);
    v.insert(v.end(), candidates.begin(), candidates.end());
    std::vector<uint32_t> nodes, slots;
    fetch_lists(v, nodes, slots);
    const size_t len = ref.node_count();
    std::vector<uint32_t> want(len);
    size_t at = 0;
    if (ref.have_nodes_) {
        for (size_t j = 0; j < len; j++) want[j] = (uint32_t) ref.nodes_[j]->index;
    } else {
        std::copy(nodes.begin(), nodes.begin() + (long) len, want.begin());
        at = len;
    }
    size_t n = 0;
    for (size_t k = 0; k < candidates.size(); k++, at += len)        // (every candidate has the reference's node count)
        if (std::equal(want.begin(), want.end(), nodes.begin() + (long) at)) n++;
    return n;
}

void PathManager::materialise(const std::vector<const Path *> &want) const {
    std::vector<const Path *> v;
    for (const Path *p : want)
        if (!p->have_nodes_ && p->owner_ == this && std::find(v.begin(), v.end(), p) == v.end()) v.push_back(p);
    if (v.empty()) return;
    std::vector<uint32_t> nodes, slots;
    fetch_lists(v, nodes, slots);
    const bool lean = graph_->lean();
    std::vector<tlm_edge> fetched;
    if (lean) {         // the Edge records of these paths, from the device CSR (nodes_[..].edges is empty)
        std::vector<uint32_t> en, es;
        uint64_t a = 0;
        for (const Path *p : v) {
            for (uint64_t j = 0; j + 1 < p->n_nodes_; j++) {
                en.push_back(nodes[a + j]);
                es.push_back(slots[a + j]);
            }
            a += p->n_nodes_;
        }
        fetched.resize(en.size());
        if (tlm_graph_edges(graph_->device_graph(), en.size(), en.data(), es.data(), fetched.data()) != TLM_OK)
            die_tlm("edge fetch");
    }
    uint64_t a = 0, fe = 0;
    for (const Path *cp : v) {
        Path &p = const_cast<Path &>(*cp);          // paths_ elements are ours; filling the lazy members is not a change
        const uint64_t n = p.n_nodes_;
        p.nodes_.resize(n);
        p.edges_.resize(n ? n - 1 : 0);
        for (uint64_t j = 0; j < n; j++) {
            const OverlapGraph::Node *node = &graph_->nodes_[nodes[a + j]];
            p.nodes_[j] = node;
            if (j + 1 < n) {
                if (lean) {
                    const tlm_edge &e = fetched[fe++];
                    edge_pool_.push_back(OverlapGraph::Edge{e.q_index, e.t_index, e.q_start, e.q_end, e.t_start, e.t_end,
                                                            e.overlap_score, e.sequence_identity, e.extension_score,
                                                            e.relative_strand != 0});
                    p.edges_[j] = &edge_pool_.back();
                } else {
                    p.edges_[j] = &node->edges[slots[a + j]];
                }
            }
        }
        a += n;
        p.have_nodes_ = true;
    }
}

void PathManager::materialiseAll() {
    // in slices: the staging buffers of one fetch hold the node lists of the slice
    const size_t step = 4096;
    for (size_t i = 0; i < paths_.size(); i += step) {
        std::vector<const Path *> v;
        for (size_t j = i; j < std::min(paths_.size(), i + step); j++) v.push_back(&paths_[j]);
        materialise(v);
    }
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
    if (gathered_) {
        // Inside a (rank, builder) block the owner has compared neighbours exactly.  A block's first path follows a path
        // of ANOTHER block (another rank, or another builder): when the two summaries coincide -- which needs two builders
        // or two ranks to have produced the same walk back to back -- the node lists are fetched and compared.
        const size_t n = paths_.size();
        std::vector<uint8_t> drop(gathered_dup_);
        std::vector<const Path *> suspects;
        std::vector<size_t> suspect_at;
        for (size_t i = 1; i < n; i++)
            if (gathered_block_first_[i] && paths_[i].n_nodes_ == paths_[i - 1].n_nodes_ && paths_[i].hash_ == paths_[i - 1].hash_)
                suspect_at.push_back(i);
        for (size_t i : suspect_at) {
            suspects.push_back(&paths_[i - 1]);
            suspects.push_back(&paths_[i]);
        }
        if (!suspects.empty()) {
            std::vector<uint32_t> nodes, slots;
            fetch_lists(suspects, nodes, slots);
            uint64_t a = 0;
            for (size_t i : suspect_at) {
                const uint64_t len = paths_[i].n_nodes_;
                drop[i] = std::equal(nodes.begin() + (long) a, nodes.begin() + (long) (a + len), nodes.begin() + (long) (a + len));
                a += 2 * len;
            }
        }
        std::vector<Path> kept;
        kept.reserve(n);
        for (size_t i = 0; i < n; i++)
            if (!drop[i]) kept.push_back(std::move(paths_[i]));
        paths_.swap(kept);
        return;
    }
    tlm_pathset *all = single_set();
    if (!all) return;
    tlm_pathset *uniq = nullptr;
    if (tlm_pathset_filter_unique_take(OverlapGraph::context(), all, &uniq) != TLM_OK) {      // consumes `all`
        std::cerr << "telomeri_hip: filterUnique failed: " << tlm_last_error() << std::endl;
        std::exit(1);
    }
    dev_sets_.assign(1, uniq);
    paths_.clear();
    edge_pool_.clear();
    append_from(*graph_, uniq);
    if (!packed_) materialiseAll();
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
    if (gathered_) {        // from the summaries, as the reference's loop does it
        for (const Path &p : paths_) out[{&graph_->nodes_[p.front_], &graph_->nodes_[p.back_]}].push_back(&p);
        return out;
    }
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
        out[{&graph_->nodes_[p.front_], &graph_->nodes_[p.back_]}].push_back(&p);
    }
    return out;
}
