// host_pool.hpp -- a few persistent helper threads for the per-camera host set-up (api.cpp prepare_frame): the
// reference evaluates Spherical's sin / cos per pixel inside its OpenMP loop (src/Spherical.cpp:23-25,
// main/hmap.cpp:978); here they are W + H table entries per camera, glibc calls that must stay on the host, and
// for a moving camera they are filled by several host threads while the GPU renders the previous frame.
#pragma once
#include <functional>

namespace hmrm {

// Runs fn(begin, end) over [0, n) cut into contiguous pieces, on the calling thread and up to `helpers` pool
// threads, and returns when every piece is done.  Pieces are at least `grain` items; with n <= grain (or no
// pool) the caller does it all.  Callers are serialised (one job at a time).
void parallel_ranges(int n, int grain, const std::function<void(int, int)> &fn);

// Number of helper threads the pool runs (0 = none: HMRM_HOST_THREADS=1 or a one-core host).
int host_pool_helpers();

} // namespace hmrm
