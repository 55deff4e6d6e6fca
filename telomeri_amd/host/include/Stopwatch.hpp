// telomeri_amd/host/include/Stopwatch.hpp -- the lap timer main() prints its "Done (x s)" lines with (reference
// include/Stopwatch.hpp:6-21: start / lap / stop, seconds as double).  Header-only here; steady_clock instead of
// system_clock (a lap must not go backwards when the wall clock is set).
#ifndef TELOMERI_HIP_STOPWATCH_HPP
#define TELOMERI_HIP_STOPWATCH_HPP

#include <chrono>

class Stopwatch {
    typedef std::chrono::steady_clock clock;
    clock::time_point start_, lap_start_;
    static double seconds(clock::time_point a, clock::time_point b) { return std::chrono::duration<double>(b - a).count(); }

public:
    /** Start the timer: total and lap both count from now. */
    void start() { start_ = lap_start_ = clock::now(); }
    /** Time since the last lap() (or start()); starts a new lap, the total runs on. */
    double lap() {
        const clock::time_point now = clock::now();
        const double s = seconds(lap_start_, now);
        lap_start_ = now;
        return s;
    }
    /** Time since start(). */
    double stop() { return seconds(start_, clock::now()); }
};

#endif
