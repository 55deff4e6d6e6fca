// telomeri_amd/host/include/Scaffolder.hpp -- the reference's Scaffolder (include/Scaffolder.hpp:11-34, src/Scaffolder.cpp).
#ifndef TELOMERI_HIP_SCAFFOLDER_HPP
#define TELOMERI_HIP_SCAFFOLDER_HPP

#include <Path.hpp>

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
