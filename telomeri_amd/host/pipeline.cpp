// telomeri_amd/host/pipeline.cpp -- the stages of the reference's main() after getPathsBetweenAnchors
// (src/Main.cpp:254-388): groups, consensus, scaffold path, FASTA.  Shared by telomeri_hip (paths from the GPU) and
// telomeri_post (paths from a dump file; CPU-only validation of these host stages against the reference).
#include <chrono>
#include <cstdlib>
#include <functional>
#include <iostream>

#include "telomeri_host.hpp"

// `loaded`: the outcome of Scaffolder::load(reads_file) && load(contigs_file) run ahead of time on a Scaffolder without a
// path (null: load here, as the reference does); it is waited for where the reference loads.  `leave_mapped`: the
// caller ends the process as soon as this returns.
int run_post_stages(PathManager &pm, std::map<PathManager::AnchorPair, std::vector<const Path *>> &paths_between_anchors,
                    const char *reads_file, const char *contigs_file, const char *output_file,
                    std::function<Scaffolder *()> loaded, bool leave_mapped) {
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&]() {
        auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return s;
    };
    std::map<PathManager::AnchorPair, std::vector<PathGroup>> groups_for_anchors;
    std::map<PathManager::AnchorPair, const Path *> consensus_for_anchors;
    const auto t_c = std::chrono::steady_clock::now();
    const long consensus_num = PathManager::selectConsensus(paths_between_anchors, groups_for_anchors, consensus_for_anchors,
                                                            pm.params_);
    const double select_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count();
    if (consensus_num == 0) {
        std::cerr << "No consensus found!" << std::endl;                    // Main.cpp:344-347
        return 1;
    }
    std::cout << "Building the scaffold..." << std::endl;
    const Path scaffold = pm.constructConsensusPath(paths_between_anchors, consensus_for_anchors, 10);   // Main.cpp:352
    if (std::getenv("TLM_HOST_TIMING"))
        std::cerr << "[host timing] consensus: groups + consensus per anchor pair " << select_ms << " ms, scaffold path "
                  << std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_c).count() - select_ms << " ms"
                  << std::endl;
    std::cout << "Done (" << lap() << "s)" << std::endl;
    std::cout << "\nLoading files for scaffolding..." << std::endl;
    Scaffolder *ahead = loaded ? loaded() : nullptr;
    if (loaded && !ahead) return 1;
    Scaffolder empty(scaffold);
    // (on the heap: a caller that ends the process right after the scaffold is written leaves the 10 GB of mapped FASTA
    // to the kernel instead of unmapping them here -- 0.5 s on the A. thaliana-shaped input)
    Scaffolder *scaff = new Scaffolder(scaffold, std::move(ahead ? *ahead : empty));
    struct Drop {
        Scaffolder *s;
        bool keep;
        ~Drop() { if (!keep) delete s; }
    } drop{scaff, false};
    if (!ahead && (!scaff->load(reads_file) || !scaff->load(contigs_file))) return 1;     // Main.cpp:357-360
    std::cout << "Done (" << lap() << "s)" << std::endl;
    std::cout << "Writing result to: " << output_file << std::endl;
    if (!scaff->write(output_file)) return 1;                               // Main.cpp:385-387
    std::cout << "Done (" << lap() << "s)" << std::endl;
    drop.keep = leave_mapped;
    return 0;
}
