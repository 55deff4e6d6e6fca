// telomeri_amd/host/consensus.cpp -- grouping, per-group consensus, tournament and scaffold-path chaining
// (SURVEY.md §8f row 3).  Host code, as in the reference: O(#paths) pointer/map logic.  The final scaffold depends
// on several accidents of the reference's implementation; each is restated and marked QUIRK.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <set>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "../csrc/cpu_share.h"
#include "telomeri_host.hpp"

// ---- PathWindow (src/PathWindow.cpp:6-39) -----------------------------------------------------------
PathWindow::PathWindow(ulong l, ulong u, const std::vector<const Path *> &sp) : sum_frqs(0) {
    for (const Path *p : sp) {
        const ulong len = (ulong) p->length();      // long -> ulong as in the reference's comparisons
        if (len >= u) break;
        if (len >= l) {
            piw_.push_back(p);
            frqs[len] += 1;
            sum_frqs++;
        }
    }
}

static bool by_freq(const std::pair<const ulong, int> &a, const std::pair<const ulong, int> &b) { return a.second < b.second; }

std::pair<ulong, int> PathWindow::getLowestFrequencyEntry() const {     // first minimum in key order
    return *std::min_element(frqs.begin(), frqs.end(), by_freq);
}

std::pair<ulong, int> PathWindow::getHighestFrequencyEntry() const {    // first maximum in key order
    return *std::max_element(frqs.begin(), frqs.end(), by_freq);
}

// ---- PathGroup (src/PathGroup.cpp) --------------------------------------------------------------------
PathGroup::PathGroup(std::vector<const Path *>::const_iterator begin, std::vector<const Path *>::const_iterator end)
    : pig_(begin, end), consensus(nullptr), valid_path_number(0) {
    // (groups come sorted by length: a length is the map's last key or a new last key, no search; any other order
    // takes the general way to the same map)
    for (const Path *p : pig_) {
        const ulong len = (ulong) p->length();
        if (!frqs.empty() && frqs.rbegin()->first == len) frqs.rbegin()->second += 1;
        else if (frqs.empty() || frqs.rbegin()->first < len) frqs.emplace_hint(frqs.end(), len, 1);
        else frqs[len] += 1;
    }
}

std::pair<ulong, int> PathGroup::getHighestFrequencyEntry() const {
    return *std::max_element(frqs.begin(), frqs.end(), by_freq);
}

void PathGroup::discardNotFrequent() {                  // PathGroup.cpp:40-55
    const int threshold_plf = getHighestFrequencyEntry().second / 2;
    if (threshold_plf == 0) return;
    // The reference visits every path: `if (frqs[len] < threshold) frqs.erase(len)`.  QUIRK: operator[] re-creates an
    // erased length with frequency 0, which is below the threshold and is erased again: the net effect is "every length
    // of a path of the group whose frequency is below the threshold leaves frqs" (a key that no path has stays).
    const bool sorted = std::is_sorted(pig_.begin(), pig_.end(),
                                       [](const Path *a, const Path *b) { return (ulong) a->length() < (ulong) b->length(); });
    if (sorted) {
        // groups come sorted by length (constructGroups): one walk over the paths and the map side by side, and the
        // map is made again from what stays (370 000 erasures with their rebalancing were 44 ms of the A. thaliana-
        // shaped run's consensus stage)
        std::vector<std::pair<ulong, int>> stay;
        auto pi = pig_.cbegin();
        for (const auto &kv : frqs) {
            while (pi != pig_.cend() && (ulong) (*pi)->length() < kv.first) ++pi;
            const bool of_a_path = pi != pig_.cend() && (ulong) (*pi)->length() == kv.first;
            if (!(of_a_path && kv.second < threshold_plf)) stay.push_back(kv);
        }
        frqs = std::map<ulong, int>(stay.begin(), stay.end());
    } else {
        for (const Path *p : pig_) {
            const ulong len = (ulong) p->length();
            if (frqs[len] < threshold_plf) frqs.erase(len);
        }
    }
    deletePathsBelowThreshold();
}

void PathGroup::deletePathsBelowThreshold() {           // PathGroup.cpp:113-125
    std::vector<const Path *> saved;
    saved.reserve(pig_.size());
    bool have = false, present = false;     // (equal lengths lie together in a sorted group: one lookup per run)
    ulong last = 0;
    for (const Path *p : pig_) {
        const ulong len = (ulong) p->length();
        if (!have || len != last) {
            have = true;
            last = len;
            present = frqs.count(len) != 0;
        }
        if (present) saved.push_back(p);
    }
    pig_ = saved;
}

void PathGroup::calculateConsensusPath() {              // PathGroup.cpp:58-79
    // QUIRK: long - long compared with an unsigned constant
    if ((ulong) (pig_.back()->length() - pig_.front()->length()) > CONSENSUS_THRESHOLD) {
        consensus = nullptr;
        return;
    }
    // QUIRK: std::accumulate with an `int` initial value: the running sum is truncated to int after every addition
    // (and sign-extended into the lambda's size_t parameter); the final int is converted to size_t for the division
    int acc = 0;
    for (const Path *p : pig_) acc = (int) ((size_t) acc + (size_t) p->length());
    const size_t avg = (size_t) acc / pig_.size();
    for (const Path *p : pig_)
        if ((ulong) p->length() >= avg) {
            consensus = p;
            return;
        }
    consensus = pig_.back();
}

void PathGroup::calculateValidPathNumber() {            // PathGroup.cpp:83-93
    valid_path_number = 0;
    if (!consensus) return;
    valid_path_number = (int) consensus->count_equal(pig_);     // operator== on every member, the device-resident ones in one fetch
}

// ---- PathManager: groups (src/PathManager.cpp:505-624) ------------------------------------------------
// Written from the behaviour, not from the reference's text.  The reference cuts the sorted
// lengths into windows of `window_size`, rebuilding a frequency map per window by rescanning all paths; here ONE pass
// over the sorted lengths yields the length histogram and, per non-empty window, the three numbers the rule uses.
namespace {

struct LengthRun {          // a run of equal lengths in the sorted list
    ulong length;
    int count;
};

struct WindowSummary {      // a non-empty window [lower, lower + window_size)
    int paths = 0;                          // PathWindow::getSumFreqs
    LengthRun rarest{0, 0}, commonest{0, 0};   // first minimum / first maximum of the frequencies in length order
};

// Windows are anchored at the shortest length; window k covers [min + k*w, min + (k+1)*w).  Empty windows do not exist
// for the rule below (the reference drops them, PathManager.cpp:531-533).
std::vector<WindowSummary> summarise_windows(const std::vector<LengthRun> &runs, ulong window_size) {
    std::vector<WindowSummary> out;
    ulong current = 0;
    for (const LengthRun &r : runs) {
        const ulong k = (r.length - runs.front().length) / window_size;
        if (out.empty() || k != current) {
            out.emplace_back();
            out.back().rarest = out.back().commonest = r;
            current = k;
        }
        WindowSummary &w = out.back();
        w.paths += r.count;
        if (r.count < w.rarest.count) w.rarest = r;
        if (r.count > w.commonest.count) w.commonest = r;
    }
    return out;
}

// A border is the rarest length of the emptiest of three neighbouring windows, if that frequency is "significantly"
// below the commonest frequency of the fullest of the three (valley < ratio * peak, an int against a float product).
// Ties go to the leftmost window, as std::min / std::max over an initializer_list return the first extremum.
std::vector<ulong> border_lengths(const std::vector<WindowSummary> &w, float ratio_threshold) {
    std::set<ulong> borders;
    for (size_t mid = 1; mid + 1 < w.size(); mid++) {
        size_t valley = mid - 1, peak = mid - 1;
        for (size_t j = mid; j <= mid + 1; j++) {
            if (w[j].paths < w[valley].paths) valley = j;
            if (w[j].paths > w[peak].paths) peak = j;
        }
        if (w[valley].rarest.count < ratio_threshold * w[peak].commonest.count) borders.insert(w[valley].rarest.length);
    }
    return std::vector<ulong>(borders.begin(), borders.end());
}

}   // namespace

std::vector<PathGroup> PathManager::constructGroups(std::vector<const Path *> &v, Parameters params) {
    // QUIRK: std::sort is not stable; tie order among equal lengths is libstdc++'s introsort result on this input
    // order, and the group consensus ("first path with length >= average") depends on it.  Same call, same library.
    std::sort(v.begin(), v.end(), [](const Path *a, const Path *b) { return a->length() < b->length(); });
    // lengths are compared as ulong from here on (the reference stores Path::length() into ulong variables)
    const ulong shortest = (ulong) v.front()->length(), longest = (ulong) v.back()->length();
    std::vector<PathGroup> groups;
    auto one_group = [&]() {
        groups.emplace_back(v.cbegin(), v.cend());
        return groups;
    };
    if (longest - shortest < (ulong) params.len_threshold) return one_group();
    // (a negative length next to a positive one makes `shortest` the larger number: the reference then builds no window
    // at all and reads past an empty vector; here that input simply stays one group)
    if (shortest > longest || params.window_size == 0) return one_group();
    std::vector<LengthRun> runs;
    for (const Path *p : v) {
        const ulong len = (ulong) p->length();
        if (!runs.empty() && runs.back().length == len) runs.back().count++;
        else runs.push_back(LengthRun{len, 1});
    }
    const std::vector<ulong> borders = border_lengths(summarise_windows(runs, params.window_size), params.ratio_threshold);
    if (borders.empty()) return one_group();
    // group k = [first path not yet used, first path at or beyond border k) -- but never empty: a group takes at least
    // the first unused path even when that one already lies beyond the border (PathManager.cpp:568-572)
    auto first = v.cbegin();
    for (ulong border : borders) {
        if (first == v.cend()) break;       // (the reference would scan past the end here; a border is a length of `v`)
        auto last = std::partition_point(first + 1, v.cend(), [&](const Path *p) { return (ulong) p->length() < border; });
        groups.emplace_back(first, last);
        first = last;
    }
    if (first != v.cend()) groups.emplace_back(first, v.cend());
    return groups;
}

std::pair<ulong, ulong> PathManager::getMinMaxPathLength(std::vector<const Path *> &v) {
    ulong min_len = ULONG_MAX, max_len = 0;
    for (const Path *p : v) {
        const ulong l = (ulong) p->length();
        min_len = std::min(min_len, l);
        max_len = std::max(max_len, l);
    }
    return {min_len, max_len};
}

// ---- the per-pair stage of main() (src/Main.cpp:254-342) ----------------------------------------------
// Written from its behaviour, in two parts.
//   (1) per group: drop the infrequent lengths, pick the group's consensus, count the paths of its length; a group
//       without consensus only leaves its log line.
//   (2) among the groups of one anchor pair that have a consensus: none -> no consensus for the pair; one -> it; otherwise
//       rank them by consensus length, longest first (std::sort: the order among equal lengths is whatever libstdc++'s
//       introsort makes of the input order, and the scaffold depends on it -- so the SAME sort runs here, over indices);
//       two groups -> the longer one; more -> walk down the ranking holding a champion, which is replaced whenever the next
//       group has at least twice the champion's number of valid paths.
namespace {

struct GroupResult {            // what part (2) needs of a group that has a consensus
    const Path *consensus;
    long length;
    int valid_paths;            // (int as in PathGroup: the comparison below is the reference's int arithmetic)
};

struct StageClock {             // TLM_HOST_TIMING: where the stage's time goes
    double discard_ms = 0, consensus_ms = 0, valid_ms = 0;
    std::chrono::steady_clock::time_point t;
    void start() { t = std::chrono::steady_clock::now(); }
    double lap() {
        const auto n = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(n - t).count();
        t = n;
        return ms;
    }
};

// the part of settling a group that touches nothing but lengths: any thread may do it
void settle_lengths(PathGroup &pg, StageClock &clk) {
    clk.start();
    pg.discardNotFrequent();
    clk.discard_ms += clk.lap();
    pg.calculateConsensusPath();
    clk.consensus_ms += clk.lap();
}

// ... and the part that COMPARES paths: equal hashes are settled by the node lists, which are fetched from the device --
// from the owning ranks, as a collective, when the paths were gathered: in the reference's order, by the calling thread
bool settle_group(PathGroup &pg, StageClock &clk, GroupResult &out) {
    clk.start();
    pg.calculateValidPathNumber();
    clk.valid_ms += clk.lap();
    if (!pg.consensus) {
        std::cout << "No consensus sequence for group!\n" << std::endl;
        return false;
    }
    out = GroupResult{pg.consensus, pg.consensus->length(), pg.valid_path_number};
    std::cout << "Consensus length: " << out.length << "    " << "Valid path number: " << out.valid_paths << std::endl;
    return true;
}

const Path *consensus_among(const std::vector<GroupResult> &with_consensus) {
    const size_t n = with_consensus.size();
    if (n == 0) return nullptr;
    std::vector<size_t> ranking(n);
    for (size_t i = 0; i < n; i++) ranking[i] = i;
    if (n >= 2)
        std::sort(ranking.begin(), ranking.end(),
                  [&](size_t x, size_t y) { return with_consensus[x].length > with_consensus[y].length; });
    size_t champion = ranking[0];
    if (n > 2)
        for (size_t r = 1; r < n; r++) {
            const size_t challenger = ranking[r];
            if (with_consensus[challenger].valid_paths >= 2 * with_consensus[champion].valid_paths) champion = challenger;
        }
    return with_consensus[champion].consensus;
}

void announce(const char *opening, const OverlapGraph::Node &from, const OverlapGraph::Node &to, const char *closing) {
    std::cout << opening << from.name << "' and anchor '" << to.name << closing << std::endl;
}

// The anchor pairs are independent of each other: their work is dealt to the host's threads, what they have to say is
// kept per pair and printed afterwards, pair by pair, in the order of the reference's single loop.
template <class Fn>
void for_each_pair(size_t n, const Fn &fn) {
    const unsigned nt = (unsigned) std::max<size_t>(1, std::min<size_t>(std::min(tlm_cpu_share(), 32u), n));
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto &x : th) x.join();
}

}   // namespace

long PathManager::selectConsensus(std::map<AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                                  std::map<AnchorPair, std::vector<PathGroup>> &groups_for_anchors,
                                  std::map<AnchorPair, const Path *> &consensus_for_anchors, const Parameters &params) {
    // One task per anchor pair for everything that only looks at LENGTHS -- the pair's groups, the infrequent lengths
    // dropped, the consensus of every group picked -- by one thread, so that what the first half allocates (a frequency map
    // per group) is given back by the thread that took it (two parallel loops made the threads free each other's memory:
    // 135 ms of CPU time in `discardNotFrequent` for 23 ms of work).  What compares paths (the valid-path count: node
    // lists may have to be fetched, a collective when the paths were gathered) and all reports follow on this thread, in
    // the reference's order: every pair's groups are reported before any consensus is looked for.
    const auto t_start = std::chrono::steady_clock::now();
    struct PairOutcome {
        std::vector<PathGroup> groups;
        StageClock clk;
    };
    std::vector<std::map<AnchorPair, std::vector<const Path *>>::iterator> pairs;
    for (auto it = paths_between_anchors.begin(); it != paths_between_anchors.end(); ++it) pairs.push_back(it);
    std::vector<PairOutcome> outcome(pairs.size());
    for_each_pair(pairs.size(), [&](size_t i) {
        PairOutcome &o = outcome[i];
        o.groups = constructGroups(pairs[i]->second, params);       // sorts the pair's paths in place
        for (PathGroup &pg : o.groups) settle_lengths(pg, o.clk);
    });
    for (size_t i = 0; i < pairs.size(); i++) {
        const AnchorPair &pair = pairs[i]->first;
        announce("====> Constructing groups for paths between anchor '", *pair.first, *pair.second, "'...");
        groups_for_anchors[pair] = std::move(outcome[i].groups);
        announce("<==== Finished constructing groups for paths between anchor '", *pair.first, *pair.second, "'!\n");
    }
    StageClock clk;
    long pairs_with_consensus = 0;
    std::vector<GroupResult> settled;
    for (size_t i = 0; i < pairs.size(); i++) {
        const AnchorPair &pair = pairs[i]->first;
        announce("====> Finding consensus path in each group between anchor '", *pair.first, *pair.second, "'...");
        settled.clear();
        for (PathGroup &pg : groups_for_anchors[pair]) {
            GroupResult r;
            if (settle_group(pg, clk, r)) settled.push_back(r);
        }
        std::cout << "Done finding consensus path in each group.\n";
        const Path *final_consensus = consensus_among(settled);
        consensus_for_anchors[pair] = final_consensus;
        if (final_consensus) {
            pairs_with_consensus++;
            std::cout << "Final consensus (consensus among groups) length: " << final_consensus->length() << '\n';
        } else {
            std::cout << "Final consensus not found!\n";
        }
        announce("====> Done finding consensus between anchor '", *pair.first, *pair.second, "'!");
        clk.discard_ms += outcome[i].clk.discard_ms;
        clk.consensus_ms += outcome[i].clk.consensus_ms;
    }
    const auto t_groups = t_start;
    if (std::getenv("TLM_HOST_TIMING"))
        std::cerr << "[host timing] selectConsensus: groups and consensus of all pairs "
                  << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_groups).count()
                  << " ms (summed over the threads: discardNotFrequent " << clk.discard_ms << ", calculateConsensusPath " << clk.consensus_ms
                  << ", calculateValidPathNumber " << clk.valid_ms << ")" << std::endl;
    return pairs_with_consensus;
}

// ---- PathManager::constructConsensusPath (src/PathManager.cpp:664-767) --------------------------------
// Written from the behaviour.  Candidates are the anchor pairs that have a consensus and at least `min_path_num`
// paths; the pair with the most paths seeds the scaffold, which then grows by the best-supported candidate that
// continues it on the left (X -> current left end) or on the right (current right end -> X).
namespace {

struct Candidate {
    const Path *consensus;
    ulong support;              // number of paths between the two anchors
};

typedef const OverlapGraph::Node *Anchor;

}   // namespace

Path PathManager::constructConsensusPath(const std::map<AnchorPair, std::vector<const Path *>> &paths_between_contigs,
                                         std::map<AnchorPair, const Path *> &consensus_paths, ulong min_path_num) {
    std::map<AnchorPair, Candidate> open;       // candidates not yet built into the scaffold
    std::vector<Anchor> anchors;                // every anchor of a candidate, in order of first appearance
    auto remember = [&](Anchor a) {
        if (std::find(anchors.begin(), anchors.end(), a) == anchors.end()) anchors.push_back(a);
    };
    const AnchorPair *seed = nullptr;
    ulong seed_support = 0;
    for (const auto &entry : paths_between_contigs) {
        const Path *consensus = consensus_paths.at(entry.first);
        const ulong support = entry.second.size();
        if (!consensus || support < min_path_num) continue;
        remember(entry.first.first);
        remember(entry.first.second);
        open[entry.first] = Candidate{consensus, support};
        if (support > seed_support) {           // the first of equally supported pairs, in map order
            seed_support = support;
            seed = &entry.first;
        }
    }
    if (!seed) throw std::out_of_range("map::at");      // what the reference's consensus_paths.at() throws here

    const auto t_fetch = std::chrono::steady_clock::now();
    {   // the node lists of all candidates in ONE fetch (any of them may be chained in)
        std::vector<const Path *> need;
        for (const auto &c : open) need.push_back(c.second.consensus);
        materialise(need);
    }
    const auto t_chain = std::chrono::steady_clock::now();
    const Path *seed_path = open.at(*seed).consensus;
    if (!seed_path->materialised()) materialise(*seed_path);
    Path scaffold;                              // a plain Path: node and edge lists only
    scaffold.nodes_ = seed_path->nodes_;
    scaffold.edges_ = seed_path->edges_;
    scaffold.setLength(seed_path->length());
    Anchor left_end = seed->first, right_end = seed->second;
    // QUIRK: only the seed's two anchors are ever excluded from the search; the seed's own candidate stays open
    const Anchor excluded[2] = {left_end, right_end};

    std::vector<const Path *> in_front;         // the pieces that grew the scaffold to the left, nearest first
    while (!open.empty()) {
        const AnchorPair *best = nullptr;
        AnchorPair best_key;
        ulong best_support = 0;
        bool grow_left = false;
        for (Anchor x : anchors) {
            if (x == excluded[0] || x == excluded[1]) continue;
            // QUIRK: an end moves as soon as a better candidate shows up DURING the scan, so later anchors are tested
            // against the moved end -- and an end moved for a candidate that is then beaten stays moved
            auto l = open.find(AnchorPair{x, left_end});
            if (l != open.end() && l->second.support > best_support) {
                best_support = l->second.support;
                best_key = l->first;
                best = &best_key;
                grow_left = true;
                left_end = x;
            }
            auto r = open.find(AnchorPair{right_end, x});
            if (r != open.end() && r->second.support > best_support) {
                best_support = r->second.support;
                best_key = r->first;
                best = &best_key;
                grow_left = false;
                right_end = x;
            }
        }
        if (!best) break;
        const Path *piece = open.at(*best).consensus;
        if (!piece->materialised()) materialise(*piece);
        const size_t m = piece->edges_.size();
        if (grow_left) {        // nodes 0..m-1 and all edges go in front; node m is the scaffold's current first node
            in_front.push_back(piece);      // (put there in one go below: every insertion at the front moved the whole scaffold)
        } else {                // node 0 is the scaffold's current last node
            scaffold.nodes_.insert(scaffold.nodes_.end(), piece->nodes_.begin() + 1, piece->nodes_.begin() + 1 + (long) m);
            scaffold.edges_.insert(scaffold.edges_.end(), piece->edges_.begin(), piece->edges_.end());
        }
        open.erase(*best);
    }
    if (!in_front.empty()) {
        size_t more_nodes = 0, more_edges = 0;
        for (const Path *piece : in_front) {
            more_nodes += piece->edges_.size();
            more_edges += piece->edges_.size();
        }
        std::vector<const OverlapGraph::Node *> nodes;
        std::vector<const OverlapGraph::Edge *> edges;
        nodes.reserve(more_nodes + scaffold.nodes_.size());
        edges.reserve(more_edges + scaffold.edges_.size());
        for (size_t k = in_front.size(); k-- > 0;) {        // the last one added stands first
            const Path *piece = in_front[k];
            nodes.insert(nodes.end(), piece->nodes_.begin(), piece->nodes_.begin() + (long) piece->edges_.size());
            edges.insert(edges.end(), piece->edges_.begin(), piece->edges_.end());
        }
        nodes.insert(nodes.end(), scaffold.nodes_.begin(), scaffold.nodes_.end());
        edges.insert(edges.end(), scaffold.edges_.begin(), scaffold.edges_.end());
        scaffold.nodes_.swap(nodes);
        scaffold.edges_.swap(edges);
    }
    if (std::getenv("TLM_HOST_TIMING"))
        std::cerr << "[host timing] scaffold path: node lists of the candidates "
                  << std::chrono::duration<double, std::milli>(t_chain - t_fetch).count() << " ms, chaining "
                  << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_chain).count() << " ms" << std::endl;
    return scaffold;
}
