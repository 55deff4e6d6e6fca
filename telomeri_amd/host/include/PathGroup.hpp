// telomeri_amd/host/include/PathGroup.hpp -- the reference's PathGroup (include/PathGroup.hpp:8-56, src/PathGroup.cpp).
#ifndef TELOMERI_HIP_PATHGROUP_HPP
#define TELOMERI_HIP_PATHGROUP_HPP

#include <map>

#include <Path.hpp>

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

#endif
