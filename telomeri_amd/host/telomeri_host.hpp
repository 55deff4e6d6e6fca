// telomeri_amd/host/telomeri_host.hpp -- the reference's host-side class surface for the hot path,
// re-implemented over the C ABI of include/telomeri_hip.h (no CPU fallback: every member that
// computes goes through libtelomeri_hip.so).
//
// Same names, members, argument meaning and error behaviour as the reference, so that its main()
// (reference src/Main.cpp:212-252) compiles against this header unchanged:
//   OverlapGraph        reference include/OverlapGraph.hpp:9-130
//   Path                reference include/Path.hpp:7-21
//   PathManager         reference include/PathManager.hpp:12-65   (builders + filterUnique + stats +
//                                                                  getPathsBetweenAnchors)
//   Utils::Metrics ...  reference include/Utils.hpp:17-27,43-75
// Not here (SURVEY.md §8f "next" rows): PathGroup / PathWindow / constructGroups /
// constructConsensusPath / Scaffolder.
#ifndef TELOMERI_HOST_HPP
#define TELOMERI_HOST_HPP

#include <climits>
#include <cstdint>
#include <map>
#include <ostream>
#include <string>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

struct tlm_ctx;
struct tlm_graph;
struct tlm_pathset;

typedef unsigned int uint;
typedef unsigned long ulong;

enum class ContigPosition : char { NONE, QUERY, TARGET };   // OverlapGraph.hpp:126-130

class OverlapGraph {
public:
    struct Edge {                       // OverlapGraph.hpp:27-43; in node X's list t_index == X, q_index = neighbour
        uint q_index, t_index;
        uint q_start, q_end, t_start, t_end;
        float overlap_score, sequence_identity, extension_score;
        bool relative_strand;
        void to_stream(std::ostream &) const;
    };
    struct Node {                       // OverlapGraph.hpp:13-25
        bool anchor;
        uint index;
        std::vector<Edge> edges;
        uint length;
        std::string name;
        bool operator==(const Node &o) const { return name == o.name; }
        void to_stream(std::ostream &) const;
    };
    enum LengthCalculationMode { MIN, MAX, AVG, SUM };      // OverlapGraph.hpp:56-65
    struct FilterParameters {           // OverlapGraph.hpp:67-78
        LengthCalculationMode mode;
        ulong min_overlap_length;
        float min_overlap_percentage;
        ulong max_overhang_length;
        float max_overhang_percentage;
    };

    std::vector<Node> nodes_;
    std::vector<Edge> edges_;
    FilterParameters filter_params_{AVG, 0, 0.0f, ULONG_MAX, 1.0f};   // Main.cpp:56-62
    uint test_load_num_ = 0;

    OverlapGraph();
    ~OverlapGraph();
    OverlapGraph(const OverlapGraph &) = delete;
    OverlapGraph &operator=(const OverlapGraph &) = delete;

    /** OverlapGraph::load (OverlapGraph.cpp:97-154): false + "Cannot open file: <path>" on stderr if the file
     *  cannot be opened; malformed input silently ends the parse; prints "Loaded K/R edges from <path>".
     *  Calls accumulate into one graph (Main.cpp:223 loads rr then cr). */
    bool load(char *filepath, bool anchors);
    std::string stats();                // OverlapGraph.cpp:24-95

    // --- device side (not in the reference) ---
    tlm_graph *device_graph() const { return dev_; }
    static tlm_ctx *context();          // one context per process, device $TLM_DEVICE (default 0)
    double last_build_ms() const { return build_ms_; }      // device time of the last graph build
    double last_parse_ms() const { return parse_ms_; }      // host tokenising + interning of the last load
    double last_download_ms() const { return download_ms_; }
    bool last_parse_was_parallel() const { return parsed_parallel_; }
    /** The host half of load alone (no GPU needed): tokenise `filepath` into the pending columns. */
    bool tokenize(char *filepath, bool anchors, ulong &records);
    uint64_t parse_checksum() const;    // FNV-1a over every tokenised column and the name table (tests)
    size_t parsed_names() const { return names_.size(); }

private:
    struct File {                       // one tokenised PAF file (SoA int32 columns)
        std::vector<int32_t> q_id, t_id, q_len, q_start, q_end, t_len, t_start, t_end, matches;
        std::vector<uint8_t> strand;
        bool anchors;
    };
    std::vector<File> files_;
    std::vector<std::string> names_;    // name table shared by all files (ids in first-seen order)
    std::unordered_map<std::string, int32_t> name_id_;
    tlm_graph *dev_ = nullptr;
    double build_ms_ = 0, parse_ms_ = 0, download_ms_ = 0;
    bool parsed_parallel_ = false;
    bool rebuild();                     // device build over all files loaded so far + download into nodes_/edges_
};

class Path {                            // Path.hpp:7-21
    long length_ = 0;

public:
    std::vector<const OverlapGraph::Node *> nodes_;
    std::vector<const OverlapGraph::Edge *> edges_;
    void updateLength();                // Path.cpp:6-16
    long length() const { return length_; }
    void setLength(long l) { length_ = l; }   // the device reports Path::length(); updateLength() recomputes it
    bool operator==(const Path &o) const;     // Path.cpp:46-49: node sequence only
    friend std::ostream &operator<<(std::ostream &s, const Path &p);
};

namespace Utils {
enum class Metrics { EXTENSION_SCORE, OVERLAP_SCORE, EXTENSION_SCORE_SQRT, OVERLAP_SCORE_SQRT };   // Utils.hpp:43-45
float getMetric(const OverlapGraph::Edge &e, const Metrics &metric);
const char *getMetricName(const Metrics &metric);
bool startsWithInsensitive(const std::string &str, const std::string &temp);
}   // namespace Utils

class PathManager {
    std::vector<Path> paths_;
    // the same paths on the device, one set per builder call (or one filtered set): filterUnique and
    // getPathsBetweenAnchors run there
    std::vector<tlm_pathset *> dev_sets_;
    const OverlapGraph *graph_ = nullptr;

public:
    PathManager() = default;
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
    void filterUnique();                // PathManager.cpp:432-438
    std::tuple<ulong, ulong, ulong> getMinMaxSumPathLength();   // PathManager.cpp:467-502
    std::string stats();                // PathManager.cpp:440-465
    std::map<std::pair<const OverlapGraph::Node *, const OverlapGraph::Node *>, std::vector<const Path *>>
    getPathsBetweenAnchors();           // PathManager.cpp:626-645

    const std::vector<Path> &paths() const { return paths_; }   // not in the reference (paths_ is private there)
    struct Counters { uint64_t units, steps, evals, draws, accepted; double kernel_ms; };
    Counters last_counters{};

private:
    void build(const OverlapGraph &g, const Utils::Metrics &metric, bool mc);
    void append_from(const OverlapGraph &g, tlm_pathset *ps);   // materialise Path objects from a device set
    tlm_pathset *single_set();                                   // concatenation of dev_sets_ (owned by dev_sets_)
};

#endif
