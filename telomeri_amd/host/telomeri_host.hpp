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
// The classes live in include/, one header per reference header and under the reference's file names
// (OverlapGraph.hpp, Path.hpp, PathGroup.hpp, PathWindow.hpp, PathManager.hpp, Scaffolder.hpp, Utils.hpp, Stopwatch.hpp):
// `-I telomeri_amd/host/include` is all the reference's main() needs (oracle/Makefile, target Telomeri_on_hip).  This
// file is the umbrella the host's own translation units include.
#ifndef TELOMERI_HOST_HPP
#define TELOMERI_HOST_HPP

#include "include/OverlapGraph.hpp"
#include "include/Path.hpp"
#include "include/Utils.hpp"
#include "include/PathWindow.hpp"
#include "include/PathGroup.hpp"
#include "include/PathManager.hpp"
#include "include/Scaffolder.hpp"
#include "include/Stopwatch.hpp"

#endif
