// telomeri_amd/host/include/OverlapGraph.hpp -- the reference's OverlapGraph (include/OverlapGraph.hpp:9-130) over the
// C ABI of include/telomeri_hip.h: same class name, nested types, public data members and member signatures, so that
// the reference's main() (src/Main.cpp:3-5 includes <PathManager.hpp>, <Stopwatch.hpp>, <Scaffolder.hpp>) compiles
// against this directory unchanged.  No CPU fallback: load() computes through libtelomeri_hip.so.
#ifndef TELOMERI_HIP_OVERLAPGRAPH_HPP
#define TELOMERI_HIP_OVERLAPGRAPH_HPP

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

#endif
