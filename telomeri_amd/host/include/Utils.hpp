// telomeri_amd/host/include/Utils.hpp -- the reference's Utils namespace (include/Utils.hpp:17-75).  The reference's
// header is where its translation units get <algorithm>, <sstream>, <chrono> and <cmath> from (src/Main.cpp:312 calls
// std::sort without including <algorithm> itself), so the same standard headers are included here.
// Not provided: Utils::fileExtensionMatches (Utils.hpp:34-41; unused, and it indexes with `filepath.back() - i`).
#ifndef TELOMERI_HIP_UTILS_HPP
#define TELOMERI_HIP_UTILS_HPP

#include <algorithm>
#include <chrono>
#include <cmath>
#include <sstream>
#include <string>

#include <OverlapGraph.hpp>

namespace Utils {
enum class Metrics { EXTENSION_SCORE, OVERLAP_SCORE, EXTENSION_SCORE_SQRT, OVERLAP_SCORE_SQRT };   // Utils.hpp:43-45
float getMetric(const OverlapGraph::Edge &e, const Metrics &metric);
const char *getMetricName(const Metrics &metric);
bool startsWithInsensitive(const std::string &str, const std::string &temp);
template <typename T> static bool contains(const std::vector<T> &v, const T &e) {                // Utils.hpp:29-32
    return std::find(v.begin(), v.end(), e) != v.end();
}
}   // namespace Utils

#endif
