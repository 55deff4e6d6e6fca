// telomeri_amd/host/consensus.cpp -- grouping, per-group consensus, tournament and scaffold-path chaining
// (SURVEY.md §8f row 3).  Host code, as in the reference: O(#paths) pointer/map logic.  The final scaffold depends
// on several accidents of the reference's implementation; each is restated and marked QUIRK.
#include <algorithm>
#include <iostream>
#include <numeric>
#include <set>

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
    for (const Path *p : pig_) frqs[(ulong) p->length()] += 1;
}

std::pair<ulong, int> PathGroup::getHighestFrequencyEntry() const {
    return *std::max_element(frqs.begin(), frqs.end(), by_freq);
}

void PathGroup::discardNotFrequent() {                  // PathGroup.cpp:40-55
    const int threshold_plf = getHighestFrequencyEntry().second / 2;
    if (threshold_plf == 0) return;
    for (const Path *p : pig_) {
        const ulong len = (ulong) p->length();
        // QUIRK: operator[] re-creates an erased length with frequency 0, which is below the threshold and is
        // erased again: the net effect is "erase every length whose frequency is below the threshold"
        if (frqs[len] < threshold_plf) frqs.erase(len);
    }
    deletePathsBelowThreshold();
}

void PathGroup::deletePathsBelowThreshold() {           // PathGroup.cpp:113-125
    std::vector<const Path *> saved;
    saved.reserve(pig_.size());
    for (const Path *p : pig_)
        if (frqs.count((ulong) p->length()) != 0) saved.push_back(p);
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
    for (const Path *p : pig_)
        if (*consensus == *p) valid_path_number++;
}

// ---- PathManager: groups (src/PathManager.cpp:505-624) ------------------------------------------------
static std::vector<ulong> getBorderPathLengths(const std::vector<PathWindow> &pws, float ratio_threshold) {
    std::vector<ulong> dividing;
    // (the reference's loop bound `i < n - 1` on size_t would run wild for n == 0; a window always exists)
    for (size_t i = 1, n = pws.size(); n >= 2 && i < n - 1; i++) {
        const PathWindow &l = pws[i - 1], &c = pws[i], &r = pws[i + 1];
        auto less = [](const PathWindow &a, const PathWindow &b) { return a.getSumFreqs() < b.getSumFreqs(); };
        const PathWindow &valley = std::min({l, c, r}, less);       // leftmost smallest
        const PathWindow &peak = std::max({l, c, r}, less);         // leftmost largest
        const std::pair<ulong, int> valley_pair = valley.getLowestFrequencyEntry();
        const std::pair<ulong, int> peak_pair = peak.getHighestFrequencyEntry();
        if (valley_pair.second < ratio_threshold * peak_pair.second) dividing.push_back(valley_pair.first);
    }
    std::set<ulong> s(dividing.begin(), dividing.end());
    dividing.assign(s.begin(), s.end());
    return dividing;
}

std::vector<PathGroup> PathManager::constructGroups(std::vector<const Path *> &v, Parameters params) {
    // QUIRK: std::sort is not stable; tie order among equal lengths is libstdc++'s introsort result on this input
    // order, and the group consensus ("first path with length >= average") depends on it.  Same call, same library.
    std::sort(v.begin(), v.end(), [](const Path *a, const Path *b) { return a->length() < b->length(); });
    const ulong min_len = (ulong) v.front()->length(), max_len = (ulong) v.back()->length();
    std::vector<PathGroup> pgs;
    if (max_len - min_len < (ulong) params.len_threshold) {
        pgs.emplace_back(v.begin(), v.end());
        return pgs;
    }
    std::vector<PathWindow> pws;
    for (ulong lower = min_len, upper = lower + params.window_size; lower <= max_len;
         lower = upper, upper += params.window_size) {
        pws.emplace_back(lower, upper, v);
        if (pws.back().getSumFreqs() == 0) pws.pop_back();
    }
    const std::vector<ulong> bs = getBorderPathLengths(pws, params.ratio_threshold);
    if (bs.empty()) {
        pgs.emplace_back(v.begin(), v.end());
        return pgs;
    }
    std::vector<const Path *>::const_iterator begin = v.begin(), end;
    for (ulong cur_border : bs) {
        // the reference scans without a bound; a border is the length of a path in `v`, so the scan stops
        for (end = begin + 1; end != v.end() && (ulong) (*end)->length() < cur_border; end++) {}
        pgs.emplace_back(begin, end);
        begin = end;
    }
    pgs.emplace_back(begin, v.cend());
    return pgs;
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
long PathManager::selectConsensus(std::map<AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                                  std::map<AnchorPair, std::vector<PathGroup>> &groups_for_anchors,
                                  std::map<AnchorPair, const Path *> &consensus_for_anchors, const Parameters &params) {
    for (auto &pbai : paths_between_anchors) {
        const OverlapGraph::Node &a1 = *pbai.first.first, &a2 = *pbai.first.second;
        std::cout << "====> Constructing groups for paths between anchor '" << a1.name << "' and anchor '" << a2.name
                  << "'..." << std::endl;
        groups_for_anchors[{&a1, &a2}] = constructGroups(pbai.second, params);      // sorts pbai.second in place
        std::cout << "<==== Finished constructing groups for paths between anchor '" << a1.name << "' and anchor '"
                  << a2.name << "'!\n" << std::endl;
    }
    long consensus_num = 0;
    for (auto &agp : groups_for_anchors) {
        const OverlapGraph::Node &a1 = *agp.first.first, &a2 = *agp.first.second;
        std::cout << "====> Finding consensus path in each group between anchor '" << a1.name << "' and anchor '" << a2.name
                  << "'..." << std::endl;
        std::vector<PathGroup *> pgswc;
        for (PathGroup &pg : agp.second) {
            pg.discardNotFrequent();
            pg.calculateConsensusPath();
            pg.calculateValidPathNumber();
            if (pg.consensus) {
                std::cout << "Consensus length: " << pg.consensus->length() << "    "
                          << "Valid path number: " << pg.valid_path_number << std::endl;
                pgswc.push_back(&pg);
            } else {
                std::cout << "No consensus sequence for group!\n" << std::endl;
            }
        }
        std::cout << "Done finding consensus path in each group.\n";
        const Path *chosen = nullptr;
        if (pgswc.size() == 1) {
            chosen = pgswc[0]->consensus;
        } else if (pgswc.size() >= 2) {
            std::sort(pgswc.begin(), pgswc.end(), [](const PathGroup *p1, const PathGroup *p2) {
                return p1->consensus->length() > p2->consensus->length();
            });
            const PathGroup *longer = pgswc[0];
            if (pgswc.size() > 2) {                     // Main.cpp:319-330: consecutive comparisons
                for (size_t i = 1; i < pgswc.size(); i++) {
                    const PathGroup *shorter = pgswc[i];
                    if (2 * longer->valid_path_number <= shorter->valid_path_number) longer = shorter;
                }
            }
            chosen = longer->consensus;
        }
        consensus_for_anchors[{&a1, &a2}] = chosen;
        if (chosen && ++consensus_num) std::cout << "Final consensus (consensus among groups) length: " << chosen->length() << '\n';
        else std::cout << "Final consensus not found!\n";
        std::cout << "====> Done finding consensus between anchor '" << a1.name << "' and anchor '" << a2.name << "'!"
                  << std::endl;
    }
    return consensus_num;
}

// ---- PathManager::constructConsensusPath (src/PathManager.cpp:664-767) --------------------------------
Path PathManager::constructConsensusPath(const std::map<AnchorPair, std::vector<const Path *>> &paths_between_contigs,
                                         std::map<AnchorPair, const Path *> &consensus_paths, ulong min_path_num) {
    std::map<AnchorPair, std::pair<const Path *, ulong>> filtered;
    std::vector<const OverlapGraph::Node *> nodes;
    auto contains = [](const std::vector<const OverlapGraph::Node *> &v, const OverlapGraph::Node *n) {
        return std::find(v.begin(), v.end(), n) != v.end();
    };
    ulong max_path_num = 0;
    AnchorPair max_path_key{nullptr, nullptr};
    for (const auto &pair : paths_between_contigs) {
        if (!consensus_paths.at(pair.first)) continue;
        const ulong path_num = pair.second.size();
        if (path_num >= min_path_num) {
            if (!contains(nodes, pair.first.first)) nodes.push_back(pair.first.first);
            if (!contains(nodes, pair.first.second)) nodes.push_back(pair.first.second);
            filtered[pair.first] = {consensus_paths[pair.first], path_num};
            if (max_path_num < path_num) {
                max_path_num = path_num;
                max_path_key = pair.first;
            }
        }
    }
    // the reference throws std::out_of_range here when no pair passed the filter
    Path scaffold = *consensus_paths.at(max_path_key);
    const OverlapGraph::Node *left_anchor = max_path_key.first, *right_anchor = max_path_key.second;
    std::vector<const OverlapGraph::Node *> visited{left_anchor, right_anchor};    // QUIRK: never extended below
    AnchorPair pair;
    while (!filtered.empty()) {             // QUIRK: the seed's own entry is still in `filtered`
        bool left = false, found = false;
        max_path_num = 0;
        for (const OverlapGraph::Node *n : nodes) {
            if (contains(visited, n)) continue;
            pair = {n, left_anchor};
            if (filtered.count(pair) > 0 && max_path_num < filtered[pair].second) {
                max_path_num = filtered[pair].second;
                max_path_key = pair;
                left = true;
                found = true;
                left_anchor = n;            // QUIRK: the anchors move DURING the scan over the candidates
            }
            pair = {right_anchor, n};
            if (filtered.count(pair) > 0 && max_path_num < filtered[pair].second) {
                max_path_num = filtered[pair].second;
                max_path_key = pair;
                left = false;
                found = true;
                right_anchor = n;
            }
        }
        if (!found) break;
        const Path *p = filtered[max_path_key].first;
        if (!left) {
            for (long i = 0; i < (long) p->edges_.size(); i++) {
                scaffold.nodes_.push_back(p->nodes_[(size_t) i + 1]);
                scaffold.edges_.push_back(p->edges_[(size_t) i]);
            }
        } else {
            for (long i = (long) p->edges_.size() - 1; i >= 0; i--) {
                scaffold.nodes_.insert(scaffold.nodes_.begin(), p->nodes_[(size_t) i]);
                scaffold.edges_.insert(scaffold.edges_.begin(), p->edges_[(size_t) i]);
            }
        }
        filtered.erase(max_path_key);
    }
    return scaffold;
}
