// telomeri_amd/host/include/PathWindow.hpp -- the reference's PathWindow (include/PathWindow.hpp:8-36, src/PathWindow.cpp).
#ifndef TELOMERI_HIP_PATHWINDOW_HPP
#define TELOMERI_HIP_PATHWINDOW_HPP

#include <map>

#include <Path.hpp>

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

#endif
