// telomeri_amd/host/include/Path.hpp -- the reference's Path (include/Path.hpp:7-21, src/Path.cpp:6-49).
#ifndef TELOMERI_HIP_PATH_HPP
#define TELOMERI_HIP_PATH_HPP

#include <climits>

#include <OverlapGraph.hpp>

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
    /** How many of `members` equal *this (operator== on each, itself included when it is among them) -- what
     *  PathGroup::calculateValidPathNumber counts (PathGroup.cpp:83-93).  Members of a device Path set whose node count and
     *  sequence hash equal this path's are settled on their node lists, fetched in ONE request instead of one per member
     *  (a group of the hostile config holds thousands of paths equal to its consensus). */
    size_t count_equal(const std::vector<const Path *> &members) const;
    bool materialised() const { return have_nodes_; }
    size_t node_count() const { return have_nodes_ ? nodes_.size() : n_nodes_; }
    friend std::ostream &operator<<(std::ostream &s, const Path &p);
};

#endif
