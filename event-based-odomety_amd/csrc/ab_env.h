// ab_env.h — A/B switches and test hooks of the host side.
//
// The shipped library (libebo_hip.so) reads four documented environment variables and nothing else:
//   EBO_HOST_THREADS, EBO_HOST_SPIN_US   the host LM's worker threads and their spin time (host_pool.h)
//   EBO_EDGE_CS_MB                       cap of the edge loss's direction table (ebo_api.cpp)
//   EBO_LM_NO_AVX2                       scalar band Cholesky in the host LM (host_lm.cpp)
//   EBO_SOLVE_TRACE, EBO_INGEST_TRACE    diagnostics on stderr
// Everything else -- forcing an implementation a launch would not pick at this size, block shapes, ablations,
// the switches the equivalence tests flip -- goes through ab_env() / ab_size(), which look at the environment only
// in the -DEBO_AB build (`make ab` -> libebo_hip_ab.so: what tools/ab/*, tools/sweep_impl.py and the tests' `ebo_ab`
// fixture load).  In the shipped build they are constant: one path per call, no knob string in the binary.
#pragma once

#include <cstdlib>

namespace ebo_ab
{
#ifdef EBO_AB
inline const char* ab_env(const char* name) { return std::getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif
inline size_t ab_size(const char* name, size_t dflt)
{
	const char* v = ab_env(name);
	if (!v || !*v)
	{
		return dflt;
	}
	return static_cast<size_t>(std::strtoull(v, nullptr, 10));
}
}  // namespace ebo_ab
using ebo_ab::ab_env;
using ebo_ab::ab_size;
