// telomeri_amd/host/main.cpp -- `telomeri_hip`: the reference's command line for the hot path.
//
//   telomeri_hip [options] <rr.paf> <cr.paf> <reads.fasta> <contigs.fasta> <out.fasta>
//
// Options and defaults as reference src/Main.cpp:56-69,100-192 (the parser there, not its help text).
// The stages run as in Main.cpp:212-388: load rr + cr, the four builders, filterUnique, stats, then grouping,
// consensus, scaffold path and Scaffolder (pipeline.cpp).  `--dump <file>` additionally writes graph and paths in
// oracle/ref_dump's text format so the result can be compared with the reference's; `--no-scaffold` stops after the
// hot path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <memory>
#include <thread>
#include <cstring>
#include <iostream>
#include <string>

#include <unistd.h>

#include <sys/stat.h>

#include "telomeri_host.hpp"
#include "../../include/telomeri_hip.h"

int run_post_stages(PathManager &pm, std::map<PathManager::AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                    const char *reads_file, const char *contigs_file, const char *output_file,
                    std::function<Scaffolder *()> loaded, bool leave_mapped);

static bool parse_ulong(const char *s, ulong &v) {
    char *end = nullptr;
    if (!s || !*s || *s == '-') return false;
    v = std::strtoul(s, &end, 10);
    return end && *end == 0;
}

static bool parse_perc(const char *s, float &v) {
    char *end = nullptr;
    if (!s || !*s) return false;
    v = std::strtof(s, &end);
    return end && *end == 0 && v >= 0.0f && v <= 1.0f;
}

static unsigned bits_of(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

static bool g_context_in_use = false;       // (the context is created by the first load)
static void stamp(const char *what) {       // TLM_HOST_TIMING: where the wall-clock outside the stages goes (start-up, exit)
    if (!std::getenv("TLM_HOST_TIMING")) return;
    const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    std::fprintf(stderr, "[host timing] %s at %.3f\n", what, now);
    tlm_memory_info mi;
    if (g_context_in_use && tlm_ctx_memory_info(OverlapGraph::context(), &mi) == TLM_OK)
        std::fprintf(stderr, "[host timing]   device memory so far: %llu driver allocations in %.0f ms, %llu requests served from the cache, "
                             "%.1f GB live, %.1f GB cached\n", (unsigned long long) mi.driver_allocs, mi.driver_alloc_ms,
                     (unsigned long long) mi.reuses, mi.live_bytes * 1e-9, mi.cached_bytes * 1e-9);
}

int main(int argc, char **argv) {
    if (argc == 2 && std::strcmp(argv[1], "--upload-helper") == 0) return OverlapGraph::upload_helper_main();
    stamp("main entered");
    struct AtExit { ~AtExit() { stamp("main left"); } } at_exit;
    OverlapGraph graph;
    PathManager pm;
    const char *mode_string = "AVG";
    std::string dump;
    bool parse_only = false, no_scaffold = false, materialise = false;
    std::vector<char *> pos;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto need = [&](const char *what) -> char * {
            if (i + 1 >= argc) { std::cerr << "Missing value for " << what << std::endl; std::exit(1); }
            return argv[++i];
        };
        ulong u;
        float f;
        if (a == "--filter-avg") { graph.filter_params_.mode = OverlapGraph::AVG; mode_string = "AVG"; }
        else if (a == "--filter-min") { graph.filter_params_.mode = OverlapGraph::MIN; mode_string = "MIN"; }
        else if (a == "--filter-max") { graph.filter_params_.mode = OverlapGraph::MAX; mode_string = "MAX"; }
        else if (a == "--filter-sum") { graph.filter_params_.mode = OverlapGraph::SUM; mode_string = "SUM"; }
        else if (a == "--min-oll") { if (!parse_ulong(need("--min-oll"), u)) return 1; graph.filter_params_.min_overlap_length = u; }
        else if (a == "--min-olp") { if (!parse_perc(need("--min-olp"), f)) return 1; graph.filter_params_.min_overlap_percentage = f; }
        else if (a == "--max-ohl") { if (!parse_ulong(need("--max-ohl"), u)) return 1; graph.filter_params_.max_overhang_length = u; }
        else if (a == "--max-ohp") { if (!parse_perc(need("--max-ohp"), f)) return 1; graph.filter_params_.max_overhang_percentage = f; }
        else if (a == "--rb-att") { if (!parse_ulong(need("--rb-att"), u)) return 1; pm.params_.rebuild_attempts = (int) u; }
        else if (a == "--bt-att") { if (!parse_ulong(need("--bt-att"), u)) return 1; pm.params_.backtrack_attempts = (int) u; }
        else if (a == "--len-thr") { if (!parse_ulong(need("--len-thr"), u)) return 1; pm.params_.len_threshold = (long) u; }
        else if (a == "--w-size") { if (!parse_ulong(need("--w-size"), u)) return 1; pm.params_.window_size = u; }
        else if (a == "--r-thr") { if (!parse_perc(need("--r-thr"), f)) return 1; pm.params_.ratio_threshold = f; }
        else if (a == "--dump") dump = need("--dump");
        else if (a == "--parse-only") parse_only = true;
        else if (a == "--materialise") materialise = true;     // fill every nodes_[..].edges, edges_ and Path list, as the reference does
        else if (a == "--no-scaffold") no_scaffold = true;      // stop after getPathsBetweenAnchors (hot path only)
        else if (a == "--test-load-num") { if (!parse_ulong(need("--test-load-num"), u)) return 1; graph.test_load_num_ = (uint) u; }
        else pos.push_back(argv[i]);
    }
    if (parse_only) {       // host tokeniser alone: `telomeri_hip --parse-only <file.paf>...`, no GPU involved
        for (char *f : pos) {
            ulong n = 0;
            if (!graph.tokenize(f, false, n)) return 1;
            std::cout << "parsed " << n << " records from " << f << " in " << graph.last_parse_ms() << " ms ("
                      << (graph.last_parse_was_parallel() ? "parallel" : "sequential") << ")" << std::endl;
        }
        std::cout << "names " << graph.parsed_names() << " checksum " << graph.parse_checksum() << std::endl;
        return 0;
    }
    if (pos.size() != 5) {
        std::cerr << "Please provide a path to read-read and contig-read files!\n"
                  << "Usage: telomeri_hip [option...] <read-read_overlap_file>.paf <contig-read_overlap_file>.paf "
                  << "<reads_file>.fasta <contigs_file>.fasta <output_file>.fasta [--dump <paths.txt>]" << std::endl;
        return 1;
    }
    std::cout << "Filter:\n    Mode: " << mode_string << "\n    Min overlap length: " << graph.filter_params_.min_overlap_length
              << "\n    Min overlap percentage: " << graph.filter_params_.min_overlap_percentage
              << "\n    Max overhang length: " << graph.filter_params_.max_overhang_length
              << "\n    Max overhang percentage: " << graph.filter_params_.max_overhang_percentage << std::endl;
    std::cout << "Path construction:\n    Rebuild attempts: " << pm.params_.rebuild_attempts
              << "\n    Backtrack attempts: " << pm.params_.backtrack_attempts
              << "\n    Length threshold: " << pm.params_.len_threshold << "\n" << std::endl;

    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&]() {
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    };
    // the overlap text is brought to the device by a helper process (its mappings are its own to take down); started
    // before this process has touched the device
    if (!materialise && dump.empty()) {
        uint64_t text = 0;
        for (int k = 0; k < 2; k++) {
            struct stat st;
            if (::stat(pos[k], &st) == 0) text += (uint64_t) st.st_size;
        }
        OverlapGraph::start_upload_helper("/proc/self/exe", text < (24ull << 30));
    }
    stamp("loading starts");
    std::cout << "\nLoading overlaps..." << std::endl;
    double parse_ms = 0, build_ms = 0, dl_ms = 0;
    // By default the adjacency and the paths' node lists stay on the device (lean graph, packed paths): the host stages
    // need lengths, end nodes and equality of paths, and the Edge records of the scaffold's paths only.  --dump and
    // --materialise fill every public member of the reference's classes.
    if (!dump.empty()) materialise = true;
    graph.set_lean(!materialise);
    pm.set_packed(!materialise);
    graph.begin_loads();            // one device build + one materialisation for both files
    for (int k = 0; k < 2; k++) {                                                     // Main.cpp:223
        if (!graph.load(pos[k], k == 1)) return 1;
        parse_ms += graph.last_parse_ms();
    }
    // The reads and contigs are not needed before the Scaffolder runs: the helper process scans them while the device
    // walks (Main.cpp:357-360 loads them after the consensus; same tables, and a file that cannot be read ends the run at
    // the same place with the same message).  In-process this cannot be had for free: a thread that maps and scans
    // 10 GB here while the walks are launched delays them by 0.3-0.6 s.
    int fasta_requested = 0;
    if (!no_scaffold && !std::getenv("TLM_NO_FASTA_AHEAD"))
        for (int k = 2; k < 4 && OverlapGraph::request_fasta_index(pos[k]); k++) fasta_requested++;
    if (!graph.end_loads()) return 1;
    build_ms = graph.last_build_ms();
    dl_ms = graph.last_download_ms();
    std::cout << "load breakdown: tokenise+intern " << parse_ms << " ms (" << graph.last_parse_mode()
              << "), device graph build " << build_ms << " ms, download+materialise " << dl_ms << " ms" << std::endl;
    std::cout << "Done (" << lap() << "s)" << std::endl << graph.stats() << std::endl;

    static const Path no_path;
    std::unique_ptr<Scaffolder> fasta;
    std::thread fasta_thread;           // takes the helper's answers in while the device walks (a million names to look up by)
    bool fasta_ok = false;
    if (fasta_requested == 2) {
        fasta.reset(new Scaffolder(no_path));
        fasta_thread = std::thread([&] {
            fasta_ok = true;
            for (int k = 2; k < 4 && fasta_ok; k++) {       // in the order of the requests; what the helper could not do is done here
                const std::string idx = OverlapGraph::collect_fasta_index();
                const bool got = !idx.empty() && fasta->load_index(idx.c_str(), pos[k]);
                if (!idx.empty()) ::unlink(idx.c_str());
                fasta_ok = got || fasta->load(pos[k]);
            }
        });
    } else {
        for (; fasta_requested > 0; fasta_requested--) {        // (half an answer is no answer: the files are loaded here)
            const std::string idx = OverlapGraph::collect_fasta_index();
            if (!idx.empty()) ::unlink(idx.c_str());
        }
    }
    struct Join {
        std::thread &t;
        ~Join() { if (t.joinable()) t.join(); }
    } join_fasta{fasta_thread};

    g_context_in_use = true;
    stamp("graph loaded");
    std::cout << "Calculating paths..." << std::endl;
    FILE *out = nullptr;
    if (!dump.empty()) {
        out = std::fopen(dump.c_str(), "w");
        if (!out) { std::perror(dump.c_str()); return 1; }
        std::fprintf(out, "TLMDUMP 1\nN %zu\n", graph.nodes_.size());
        size_t adj = 0;
        for (const auto &n : graph.nodes_) {
            std::fprintf(out, "n %u %d %u %zu %s\n", n.index, n.anchor ? 1 : 0, n.length, n.edges.size(), n.name.c_str());
            adj += n.edges.size();
        }
        std::fprintf(out, "A %zu\n", adj);
        for (const auto &n : graph.nodes_)
            for (size_t s = 0; s < n.edges.size(); s++) {
                const auto &e = n.edges[s];
                std::fprintf(out, "a %u %zu %u %u %u %u %u %u %u %u %d\n", e.t_index, s, e.q_index, e.q_start, e.q_end,
                             e.t_start, e.t_end, bits_of(e.overlap_score), bits_of(e.sequence_identity),
                             bits_of(e.extension_score), e.relative_strand ? 1 : 0);
            }
    }
    struct B { int id; bool mc; Utils::Metrics m; };
    const B builders[4] = {{0, true, Utils::Metrics::EXTENSION_SCORE}, {1, true, Utils::Metrics::OVERLAP_SCORE},
                           {2, false, Utils::Metrics::EXTENSION_SCORE}, {3, false, Utils::Metrics::OVERLAP_SCORE}};
    uint64_t evals = 0;
    double kernel_ms = 0;
    if (!out) {     // the four builder calls of Main.cpp:237-243 as one device call (same paths_, same log lines)
        pm.buildAll(graph, Utils::Metrics::EXTENSION_SCORE, Utils::Metrics::OVERLAP_SCORE);
        evals += pm.last_counters.evals;
        kernel_ms += pm.last_counters.kernel_ms;
    }
    for (const B &b : builders) {                                                     // Main.cpp:237-243
        if (!out) break;        // (--dump wants the builders one by one, to write a section after each)
        const size_t before = pm.paths().size();
        if (b.mc) pm.buildMonteCarlo(graph, b.m);
        else pm.buildDeterministic(graph, b.m);
        evals += pm.last_counters.evals;
        kernel_ms += pm.last_counters.kernel_ms;
        if (out) {
            std::fprintf(out, "B %d %zu\n", b.id, pm.paths().size() - before);
            for (size_t i = before; i < pm.paths().size(); i++) {
                const Path &p = pm.paths()[i];
                std::fprintf(out, "p %ld %zu", p.length(), p.nodes_.size());
                for (const auto *n : p.nodes_) std::fprintf(out, " %u", n->index);
                for (const auto *e : p.edges_) std::fprintf(out, " %ld", (long) (e - graph.nodes_[e->t_index].edges.data()));
                std::fprintf(out, "\n");
            }
        }
    }
    // the helper is let go only now: a process that leaves gives its share of the device back, and the driver interrupts
    // every running kernel for that (the Monte-Carlo launch took 835 ms instead of 740 ms beside a leaving helper)
    OverlapGraph::stop_upload_helper();
    stamp("builders done");
    pm.gather();                    // multi-GPU jobs: the one exchange (nothing happens in a one-process job)
    // Every rank now knows every path (summaries; node lists come from their owners on demand, collectively): all ranks
    // run the same host stages below, rank 0 keeps the scaffold file.
    const bool keeps_output = tlm_dist_rank(OverlapGraph::context()) == 0;
    pm.filterUnique();                                                                // Main.cpp:247
    stamp("unique filtered");
    auto pba = pm.getPathsBetweenAnchors();                                           // Main.cpp:251-252
    stamp("anchor pairs done");
    if (out) {
        std::fprintf(out, "U %zu\n", pm.paths().size());
        for (const Path &p : pm.paths()) {
            std::fprintf(out, "p %ld %zu", p.length(), p.nodes_.size());
            for (const auto *n : p.nodes_) std::fprintf(out, " %u", n->index);
            for (const auto *e : p.edges_) std::fprintf(out, " %ld", (long) (e - graph.nodes_[e->t_index].edges.data()));
            std::fprintf(out, "\n");
        }
        std::fprintf(out, "K %zu\n", pba.size());
        for (const auto &kv : pba) {
            std::fprintf(out, "k %u %u %zu", kv.first.first->index, kv.first.second->index, kv.second.size());
            for (const Path *p : kv.second) std::fprintf(out, " %ld", (long) (p - pm.paths().data()));
            std::fprintf(out, "\n");
        }
        std::fclose(out);
    }
    std::cout << "Done (" << lap() << "s)" << std::endl << pm.stats() << std::endl;
    std::cout << "edge-extension evals: " << evals << " in " << kernel_ms << " ms of walk kernels ("
              << (kernel_ms > 0 ? evals / (kernel_ms * 1e-3) : 0.0) << " evals/s)" << std::endl;
    std::cout << "Anchor pairs with paths: " << pba.size() << std::endl;
    stamp("paths built");
    if (no_scaffold) return 0;
    const auto t_total = std::chrono::steady_clock::now();
    static char dev_null[] = "/dev/null";
    std::function<Scaffolder *()> loaded;
    if (fasta) loaded = [&]() -> Scaffolder * {
        if (fasta_thread.joinable()) fasta_thread.join();
        return fasta_ok ? fasta.get() : nullptr;
    };
    const bool fast_exit = tlm_dist_world(OverlapGraph::context()) == 1 && !std::getenv("TLM_CLEAN_EXIT");
    const int rc = run_post_stages(pm, pba, pos[2], pos[3], keeps_output ? pos[4] : dev_null, loaded, fast_exit);     // Main.cpp:254-388
    stamp("scaffold written");
    std::cout << "Host stages after the path build: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_total).count()
              << "s" << std::endl;
    // The scaffold is on disk.  Tearing the process down piece by piece (unmapping 10 GB of FASTA, returning 20 GB of
    // device memory block by block, destroying the context) took 0.5 s of the A. thaliana run; the kernel does the same
    // for a process that simply ends.  One-process jobs only (ranks of a multi-GPU job leave their communicator in order);
    // TLM_CLEAN_EXIT=1 keeps the destructors (leak checkers).
    if (fast_exit) {
        std::cout.flush();
        std::cerr.flush();
        std::fflush(nullptr);
        stamp("main left");
        std::_Exit(rc);
    }
    return rc;
}
