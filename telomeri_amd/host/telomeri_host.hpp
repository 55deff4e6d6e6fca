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
//   PathGroup           reference include/PathGroup.hpp:8-56, src/PathGroup.cpp
//   PathWindow          reference include/PathWindow.hpp:8-36, src/PathWindow.cpp
//   Scaffolder          reference include/Scaffolder.hpp:11-34, src/Scaffolder.cpp
// The grouping / consensus / scaffold stages are plain host code: small pointer and
// std::map logic that the reference keeps on the host too; they are restated with every quirk that the final
// scaffold depends on (see the comments at each member).
#ifndef TELOMERI_HOST_HPP
#define TELOMERI_HOST_HPP

#include <climits>
#include <cstdint>
#include <deque>
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
struct tlm_paf;
struct tlm_gathered;

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
    /** Not in the reference.  A lean graph keeps nodes_ (name, index, length, anchor) but leaves nodes_[i].edges and
     *  edges_ EMPTY: the adjacency stays on the device, where the walks read it (233 M entries = 9 GB of Edge structs at
     *  the A. thaliana size), stats() is reduced there, and the few Edge records the Scaffolder needs are fetched per path
     *  (PathManager::materialise).  Default off: every public member is filled as in the reference. */
    void set_lean(bool on) { lean_ = on; }
    bool lean() const { return lean_; }
    /** Not in the reference.  Between begin_loads() and end_loads(), load() only tokenises its file; the device graph
     *  is built and the public members are filled ONCE, by end_loads(), which then prints the "Loaded K/R edges" line
     *  of every file.  Without the pair every load() rebuilds everything loaded so far (same final state). */
    void begin_loads() { deferred_ = true; }
    bool end_loads();

    // --- device side (not in the reference) ---
    tlm_graph *device_graph() const { return dev_; }
    static tlm_ctx *context();          // one context per process, device $TLM_DEVICE (default 0)
    /** The device the context is (or will be) created on: $TLM_DEVICE / $LOCAL_RANK / the rank of a multi-process job. */
    static int device_ordinal();
    /** Starts `exe --upload-helper`: a second process that brings the overlap files' text to the device for this one
     *  (tokenize_on_device), so that mapping, pinning and -- above all -- unmapping 15 GB of text happen in ITS address
     *  space.  To be called before this process touches the device.  Without a helper the text is uploaded in-process.
     *  `context_at_once`: the helper may come up on the device beside this process (small inputs; see helper_go). */
    static void start_upload_helper(const char *exe, bool context_at_once);
    static void stop_upload_helper();   // no more requests: the helper leaves on its own (after the ones still queued)
    /** Asks the helper to scan a FASTA / FASTQ file (Scaffolder::load there, save_index to a temporary file); the answer
     *  -- the index file's name, to be read with Scaffolder::load_index and deleted -- is collected later, in the order
     *  of the requests.  false / empty name: no helper, or it failed (the caller loads the file itself). */
    static bool request_fasta_index(const char *filename);
    static std::string collect_fasta_index();
    /** The helper's side (main() of `--upload-helper`): serves upload requests on stdin / stdout until end of input. */
    static int upload_helper_main();
    double last_build_ms() const { return build_ms_; }      // device time of the last graph build
    double last_parse_ms() const { return parse_ms_; }      // host tokenising + interning of the last load
    double last_download_ms() const { return download_ms_; }
    bool last_parse_was_parallel() const { return parsed_parallel_; }
    /** "device", "parallel" (host, all cores) or "sequential" (host, the exact token-stream reader) */
    const char *last_parse_mode() const { return parsed_on_device_ ? "device" : parsed_parallel_ ? "parallel" : "sequential"; }
    /** The host half of load alone (no GPU needed): tokenise `filepath` into the pending columns. */
    bool tokenize(char *filepath, bool anchors, ulong &records);
    uint64_t parse_checksum() const;    // FNV-1a over every tokenised column and the name table (tests)
    size_t parsed_names() const { ensure_names(); return names_.size(); }

private:
    // A column of a tokenised file: a vector whose resize does NOT zero-fill (the parser threads write every element;
    // first touch by the writers spreads the page faults of a 5 GB table over all cores).
    template <class T> class Column {
        T *p_ = nullptr;
        size_t n_ = 0, cap_ = 0;

    public:
        Column() = default;
        Column(const Column &) = delete;
        Column &operator=(const Column &) = delete;
        Column(Column &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        ~Column() { ::operator delete(p_); }
        void resize_uninitialised(size_t n) {
            if (n > cap_) {
                T *q = static_cast<T *>(::operator new(n * sizeof(T)));
                for (size_t i = 0; i < n_; i++) q[i] = p_[i];
                ::operator delete(p_);
                p_ = q;
                cap_ = n;
            }
            n_ = n;
        }
        void push_back(T v) {
            if (n_ == cap_) {
                const size_t keep = n_;
                resize_uninitialised(cap_ ? cap_ * 2 : 1024);
                n_ = keep;
            }
            p_[n_++] = v;
        }
        size_t size() const { return n_; }
        T *data() { return p_; }
        const T *data() const { return p_; }
        T &operator[](size_t i) { return p_[i]; }
        const T &operator[](size_t i) const { return p_[i]; }
    };
    struct File {                       // one tokenised PAF file (SoA int32 columns)
        Column<int32_t> q_id, t_id, q_len, q_start, q_end, t_len, t_start, t_end, matches;
        Column<uint8_t> strand;
        bool anchors = false;
    };
    std::vector<File> files_;
    mutable std::vector<std::string> names_;    // name table shared by all files (ids in first-seen order)
    // after a load of device-tokenised text the table is still the device's blob; ensure_names() turns it into names_
    mutable std::vector<uint32_t> name_off_;
    mutable std::vector<char> name_blob_;
    mutable bool names_pending_ = false;
    void ensure_names() const;
    struct NameIndex;                   // names_ by name (overlap_graph.cpp)
    NameIndex *index_ = nullptr;
    tlm_graph *dev_ = nullptr;
    double build_ms_ = 0, parse_ms_ = 0, download_ms_ = 0;
    bool parsed_parallel_ = false;
    // lean graphs between begin_loads() and end_loads(): the text is tokenised on the device (tlm_paf_*), the host keeps
    // no columns at all; a file that is not clean PAF text sends every file of the graph back to the host tokeniser
    bool parsed_on_device_ = false;
    tlm_paf *paf_ = nullptr;
    std::vector<std::pair<std::string, bool>> device_files_;       // (path, anchors) of the files tokenised on the device
    bool tokenize_on_device(char *filepath, bool anchors, ulong &records);
    bool deferred_ = false;
    bool lean_ = false;
    std::vector<std::pair<std::string, ulong>> pending_;   // (path, records read) of the loads since begin_loads()
    bool rebuild();                     // device build over all files loaded so far + download into nodes_/edges_
};

class PathManager;

class Path {                            // Path.hpp:7-21
    long length_ = 0;
    // A path that lives in a device Path set (PathManager) carries what grouping and bucketing need -- length, end
    // nodes, node count, a hash of the node sequence -- and gets its nodes_ / edges_ only when somebody asks for them
    // (PathManager::materialise): 372 k paths x 3500 nodes would be 20 GB of pointers at the A. thaliana size, of which
    // the scaffold uses a few hundred paths.
    friend class PathManager;
    const PathManager *owner_ = nullptr;
    uint64_t index_ = 0, hash_ = 0;
    uint32_t front_ = 0, back_ = 0, n_nodes_ = 0;
    mutable bool have_nodes_ = true;    // false: nodes_ / edges_ are still on the device

public:
    std::vector<const OverlapGraph::Node *> nodes_;
    std::vector<const OverlapGraph::Edge *> edges_;
    void updateLength();                // Path.cpp:6-16
    long length() const { return length_; }
    void setLength(long l) { length_ = l; }   // the device reports Path::length(); updateLength() recomputes it
    bool operator==(const Path &o) const;     // Path.cpp:46-49: node sequence only
    bool materialised() const { return have_nodes_; }
    size_t node_count() const { return have_nodes_ ? nodes_.size() : n_nodes_; }
    friend std::ostream &operator<<(std::ostream &s, const Path &p);
};

namespace Utils {
enum class Metrics { EXTENSION_SCORE, OVERLAP_SCORE, EXTENSION_SCORE_SQRT, OVERLAP_SCORE_SQRT };   // Utils.hpp:43-45
float getMetric(const OverlapGraph::Edge &e, const Metrics &metric);
const char *getMetricName(const Metrics &metric);
bool startsWithInsensitive(const std::string &str, const std::string &temp);
}   // namespace Utils

class PathWindow {                      // PathWindow.hpp:8-36
    std::vector<const Path *> piw_;
    std::map<ulong, int> frqs;
    int sum_frqs;

public:
    PathWindow(ulong l, ulong u, const std::vector<const Path *> &sp);   // paths with length in [l, u)
    std::pair<ulong, int> getLowestFrequencyEntry() const;
    std::pair<ulong, int> getHighestFrequencyEntry() const;
    int getSumFreqs() const { return sum_frqs; }
};

class PathGroup {                       // PathGroup.hpp:8-56
public:
    std::vector<const Path *> pig_;
    std::map<ulong, int> frqs;
    const Path *consensus;
    int valid_path_number;
    PathGroup(std::vector<const Path *>::const_iterator begin, std::vector<const Path *>::const_iterator end);
    void discardNotFrequent();
    void calculateConsensusPath();
    void calculateValidPathNumber();

private:
    std::pair<ulong, int> getHighestFrequencyEntry() const;
    void deletePathsBelowThreshold();
    static constexpr ulong CONSENSUS_THRESHOLD = 100000;
};

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
    /** Not in the reference.  Packed mode (default on): builders leave the node / edge lists of their paths on the
     *  device; materialise() fills nodes_ / edges_ of one path (or of all) from there.  Off: every path is materialised
     *  as it is built, as the reference does. */
    void set_packed(bool on) { packed_ = on; }
    void materialise(const Path &p) const;
    void materialise(const std::vector<const Path *> &ps) const;
    void materialiseAll();
    /** Test hook (CPU-only validation of the host stages): take paths produced elsewhere. */
    void adoptPaths(std::vector<Path> &&p) { paths_ = std::move(p); }
    struct Counters { uint64_t units, steps, evals, draws, accepted; double kernel_ms; };
    Counters last_counters{};

private:
    void build(const OverlapGraph &g, const Utils::Metrics &metric, bool mc);
    void append_from(const OverlapGraph &g, tlm_pathset *ps);   // Path objects (packed views) of a device set
    tlm_pathset *single_set();                                   // concatenation of dev_sets_ (owned by dev_sets_)
    bool packed_ = true;
    // multi-GPU jobs after gather(): the summaries of every rank's paths (paths_ = views of them, Path::index_ = position
    // in the gathered order); node lists are fetched from their owners, collectively, when somebody asks
    tlm_gathered *gathered_ = nullptr;
    std::vector<uint8_t> gathered_dup_, gathered_block_first_;
    void fetch_lists(const std::vector<const Path *> &v, std::vector<uint32_t> &nodes, std::vector<uint32_t> &slots) const;
    // paths_ is the concatenation of dev_sets_ in order (Path::index_ = position in paths_); Edge records fetched for
    // the paths of a lean graph live here (a deque: pointers into it stay valid)
    mutable std::deque<OverlapGraph::Edge> edge_pool_;
};

class Scaffolder {                      // Scaffolder.hpp:11-34
public:
    const Path &p_;
    std::vector<std::string> names_;
    /** As in the reference, one string per record -- but filled on demand: load() scans the file and notes where every
     *  sequence line lies; getSequenceFrom() reads the ones somebody asks for, write() reads the few hundred characters it
     *  cuts from each read of the scaffold (a fifth of the reads on the A. thaliana-shaped input).  materialiseAll()
     *  fills the rest (the reference's state). */
    std::vector<std::string> sequences_;
    explicit Scaffolder(const Path &p) : p_(p) {}
    /** The tables of `loaded` (a Scaffolder whose load() calls ran before the scaffold path existed -- telomeri_hip reads
     *  the FASTA files while the device walks) under the path `p`. */
    Scaffolder(const Path &p, Scaffolder &&loaded)
        : p_(p), names_(std::move(loaded.names_)), sequences_(std::move(loaded.sequences_)), index_(std::move(loaded.index_)),
          where_(std::move(loaded.where_)), files_(std::move(loaded.files_)) {
        loaded.files_.clear();
    }
    ~Scaffolder();
    Scaffolder(const Scaffolder &) = delete;
    Scaffolder &operator=(const Scaffolder &) = delete;
    void materialiseAll();
    bool load(const char *filename);                    // Scaffolder.cpp:12-41
    /** What the load() calls so far have found (names; where every sequence line lies), written to `index_file` -- and
     *  read back by ANOTHER process's Scaffolder, which then stands as if it had loaded `filename` itself
     *  (telomeri_hip's helper process scans the 10 GB of reads while the device walks; the scan maps the whole file, and
     *  taking such a mapping down again costs 0.35 s under a lock the walking process needs). */
    /** A Scaffolder that only scans for save_index(): load() then neither builds the name lookup nor takes the file's
     *  mapping down (the helper process ends right after; both cost as much as the scan itself). */
    bool scan_only_ = false;
    bool save_index(const char *index_file) const;
    bool load_index(const char *index_file, const char *filename);
    bool write(const char *filename);                   // Scaffolder.cpp:43-102
    std::string substring(std::string &str, ulong start, ulong end);   // inclusive ends (:114-117)
    std::string &getSequenceFrom(const std::string &name);              // :104-112 (first match)
    void swapBases(std::string &str);                                   // :129-147

private:
    std::unordered_map<std::string, size_t> index_;     // name -> first index (the reference scans linearly)
    struct View { int file; size_t off, n; };           // file < 0: none (the string is in sequences_, or there is no such line)
    std::vector<View> where_;                           // where the sequence line of record i lies in a loaded file
    std::vector<int> files_;                            // the loaded files (descriptors), until the Scaffolder goes
};

#endif
