// telomeri_amd/host/include/PathManager.hpp -- the reference's PathManager (include/PathManager.hpp:12-65): the two
// builders run on the device (include/telomeri_hip.h: tlm_walk_mc / tlm_walk_det), filterUnique and
// getPathsBetweenAnchors as device post-passes, grouping / consensus on the host as in the reference.  Includes what the
// reference's header includes (include/PathManager.hpp:4-10).
#ifndef TELOMERI_HIP_PATHMANAGER_HPP
#define TELOMERI_HIP_PATHMANAGER_HPP

#include <sstream>
#include <tuple>

#include <OverlapGraph.hpp>
#include <Path.hpp>
#include <PathGroup.hpp>
#include <Utils.hpp>

class PathManager {
    std::vector<Path> paths_;
    // the same paths on the device, one set per builder call (or one filtered set): filterUnique and
    // getPathsBetweenAnchors run there
    std::vector<tlm_pathset *> dev_sets_;
    const OverlapGraph *graph_ = nullptr;

public:
    PathManager();
    ~PathManager();
    PathManager(const PathManager &) = delete;
    PathManager &operator=(const PathManager &) = delete;
    struct Parameters {                 // PathManager.hpp:27-40
        int rebuild_attempts;
        int backtrack_attempts;
        long len_threshold;
        ulong window_size;
        float ratio_threshold;
    };
    Parameters params_{2000, 60, 4700000l, 1000ul, 0.9f};   // Main.cpp:63-69

    /** PathManager.cpp:12-177 on the device.  One std::mt19937 per walk, seeded with
     *  tlm_walk_seed(metric, anchor node, trial): the stated deviation from the reference's shared stream. */
    void buildMonteCarlo(const OverlapGraph &g, const Utils::Metrics &metric);
    /** PathManager.cpp:179-386 on the device. */
    void buildDeterministic(const OverlapGraph &g, const Utils::Metrics &metric);
    /** Not in the reference: the four builder calls of main() (Main.cpp:237-243: buildMonteCarlo under `a` and `b`, then
     *  buildDeterministic under `a` and `b`) as ONE device call (tlm_walk_builders: both metrics share a launch, the
     *  deterministic launch fills the Monte-Carlo launch's tail).  Same paths_, same order, same log lines. */
    void buildAll(const OverlapGraph &g, const Utils::Metrics &a, const Utils::Metrics &b);
    /** Not in the reference: the single exchange of a one-process-per-GPU job (every rank's paths -> every rank, in the
     *  order one PathManager would hold them); a no-op without TLM_WORLD > 1.  Between the builders and filterUnique. */
    void gather();
    void filterUnique();                // PathManager.cpp:432-438
    std::tuple<ulong, ulong, ulong> getMinMaxSumPathLength();   // PathManager.cpp:467-502
    std::string stats();                // PathManager.cpp:440-465
    std::map<std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *>, std::vector<const Path *>>
    getPathsBetweenAnchors();           // PathManager.cpp:626-645

    typedef std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *> AnchorPair;
    /** PathManager.cpp:508-584 (sorts `v` by length as the reference does). */
    static std::vector<PathGroup> constructGroups(std::vector<const Path *> &v, Parameters params);
    static std::pair<ulong, ulong> getMinMaxPathLength(std::vector<const Path *> &v);       // :648-662
    /** PathManager.cpp:664-767. */
    Path constructConsensusPath(const std::map<AnchorPair, std::vector<const Path *>> &paths_between_contigs,
                                std::map<AnchorPair, const Path *> &consensus_paths, ulong min_path_num);
    /** The per-pair stage of main() (Main.cpp:254-342): groups, per-group consensus, tournament among groups.
     *  Prints the reference's log lines; returns the number of pairs with a consensus. */
    static long selectConsensus(std::map<AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                                std::map<AnchorPair, std::vector<PathGroup>> &groups_for_anchors,
                                std::map<AnchorPair, const Path *> &consensus_for_anchors, const Parameters &params);

    const std::vector<Path> &paths() const { return paths_; }   // not in the reference (paths_ is private there)
    /** Not in the reference.  Packed mode: builders leave the node / edge lists of their paths on the device;
     *  materialise() fills nodes_ / edges_ of one path (or of all) from there.  Off (the default, unless TLM_HOST_LEAN=1
     *  is set in the environment): every path is materialised as it is built, as the reference does.  telomeri_hip
     *  switches it on (host/main.cpp). */
    void set_packed(bool on) { packed_ = on; }
    void materialise(const Path &p) const;
    void materialise(const std::vector<const Path *> &ps) const;
    /** Path::count_equal for paths of this manager that are still on the device: node lists of `ref` and of the
     *  candidates in one fetch (a collective when the paths were gathered), compared here; nothing is materialised. */
    size_t count_equal_on_device(const Path &ref, const std::vector<const Path *> &candidates) const;
    void materialiseAll();
    /** Test hook (CPU-only validation of the host stages): take paths produced elsewhere. */
    void adoptPaths(std::vector<Path> &&p) { paths_ = std::move(p); }
    struct Counters { uint64_t units, steps, evals, draws, accepted; double kernel_ms; };
    Counters last_counters{};

private:
    void build(const OverlapGraph &g, const Utils::Metrics &metric, bool mc);
    void append_from(const OverlapGraph &g, tlm_pathset *ps);   // Path objects (packed views) of a device set
    tlm_pathset *single_set();                                   // concatenation of dev_sets_ (owned by dev_sets_)
    bool packed_;
    // multi-GPU jobs after gather(): the summaries of every rank's paths (paths_ = views of them, Path::index_ = position
    // in the gathered order); node lists are fetched from their owners, collectively, when somebody asks
    tlm_gathered *gathered_ = nullptr;
    std::vector<uint8_t> gathered_dup_, gathered_block_first_;
    void fetch_lists(const std::vector<const Path *> &v, std::vector<uint32_t> &nodes, std::vector<uint32_t> &slots) const;
    // paths_ is the concatenation of dev_sets_ in order (Path::index_ = position in paths_); Edge records fetched for
    // the paths of a lean graph live here (a deque: pointers into it stay valid)
    mutable std::deque<OverlapGraph::Edge> edge_pool_;
};

#endif
