// lockstep.h -- the host side of the lock-step solves (EBO_SOLVE_GLOBAL; EBO_SOLVE_INDEPENDENT where the
// device-resident solver is not used): many HostLm state machines advanced together so that every round is
// ONE batched evaluation of all data terms, the per-window LM steps spread over a HostPool, and -- with
// many windows -- two (to four) groups of windows pipelined: while the backend evaluates one group the host
// takes another's results, runs its LM steps and posts its next points.
//
// Templates on the BACKEND that evaluates the data terms, so that this file is free of HIP: libebo_hip.so
// instantiates it with the device (ebo_api.cpp, CtxLockstepBackend), tests/cpp/hostlm_stress.cpp with a CPU
// objective evaluated on another thread -- the same driver code, run under ThreadSanitizer and
// AddressSanitizer + UBSan by the CPU test-suite (sanitizers on the CPU build).
//
// Backend:
//   int  eval(const double* flows, double* r, double* J /* may be null */, const unsigned char* modes /* may be null */,
//             int windowSlots = 0);
//        synchronous; modes[i]: 0 skip slot i, 1 value, 2 value + Jacobian (null: all slots, Jacobian iff J);
//        windowSlots > 0: ONE mode per aligned run of that many slots (a window's patches)
//   bool pipelined(int windows, size_t slots);  int groups();
//   int  pipeline_begin(size_t slots, int groups);  void pipeline_end(int groups);
//   int  eval_begin(const double* flows, const unsigned char* modes, int group, size_t s0, size_t s1, bool wantJac,
//                   int windowSlots);
//        asynchronous: may read flows[2 s0, 2 s1) and modes[0, slots) only until it returns; windowSlots > 0: the
//        caller guarantees ONE mode per aligned run of that many slots (a window's patches)
//   int  eval_finish(const unsigned char* modes, int group, size_t s0, size_t s1, bool wantJac, double* r, double* J,
//                    int windowSlots);
//        waits for that group's round, writes r / J of slots [s0, s1) with a non-zero mode
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host_lm.h"
#include "ab_env.h"
#include "host_pool.h"

namespace ebo
{
// EBO_SOLVE_GLOBAL: lm[w] = window w's problem (P flow slots each); flows: [Wn][P][2] scratch of the rounds.
template <class Backend>
int lockstep_global(Backend& be, int Wn, int P, std::vector<HostLm>& lm, std::vector<double>& flows, bool trace)
{
	flows.assign(static_cast<size_t>(Wn) * P * 2, 0.0);
	std::vector<double> r(static_cast<size_t>(Wn) * P), J(static_cast<size_t>(Wn) * P * 2);
	std::vector<unsigned char> modes(static_cast<size_t>(Wn) * P, 0), wmode(Wn, 0);
	// (with spinning workers a thread pays off from two windows' LM steps up: 64 windows, 17.2 -> see DESIGN 4.3)
	HostPool pool(static_cast<size_t>(Wn), 2);
	// Speculation, once a batch has thinned out and the device waits for the host more than the host for the device: a
	// window that asks for the COST at a candidate is evaluated with its Jacobian as well; if the step is accepted --
	// the solver then asks for value and Jacobian at that very point -- the request is answered from that evaluation
	// without another round trip.  The solver sees the same numbers in the same order (same kernel, same point):
	// same trajectory, same evaluation counts, fewer rounds.  Applied to rounds of at most specMax running windows
	// (EBO_SOLVE_SPECULATE, default 8; 0 = never).
	const char* specEnv = ab_env("EBO_SOLVE_SPECULATE");
	const size_t specMax = specEnv ? static_cast<size_t>(std::max(0, std::atoi(specEnv))) : 8;
	std::vector<unsigned char> spec(static_cast<size_t>(Wn), 0);
	std::vector<double> specPoint(specMax ? static_cast<size_t>(Wn) * P * 2 : 0);
	// what window w wants evaluated next, as the mode of the launch: 0 nothing (done), 1 value, 2 value + Jacobian
	auto ask = [&](size_t w, bool speculate) -> unsigned char {
		double* fw = &flows[w * P * 2];
		HostLm::Request q = lm[w].request(fw);
		if (q == HostLm::NEED_JACOBIAN && spec[w] && std::memcmp(fw, &specPoint[w * P * 2], static_cast<size_t>(P) * 2 * sizeof(double)) == 0)
		{
			lm[w].supply(&r[w * P], &J[w * P * 2]);  // the window's slots still hold that evaluation
			q = lm[w].request(fw);
		}
		spec[w] = 0;
		if (q == HostLm::DONE)
		{
			return 0;
		}
		if (q == HostLm::NEED_JACOBIAN)
		{
			return 2;
		}
		if (speculate)
		{
			spec[w] = 1;
			std::memcpy(&specPoint[w * P * 2], fw, static_cast<size_t>(P) * 2 * sizeof(double));
			return 2;
		}
		return 1;
	};
	// trace (EBO_SOLVE_TRACE=1): where a lock-step solve spends its time (stderr, one line per call)
	double tReq = 0.0, tEval = 0.0, tSup = 0.0, tLaunch = 0.0;
	const auto tStart = std::chrono::steady_clock::now();
	int rounds = 0;
	auto now = [] { return std::chrono::steady_clock::now(); };
	auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
		return std::chrono::duration<double, std::milli>(b - a).count();
	};
	// Many windows: two halves in flight.  While the device evaluates one half, the host takes the
	// other half's results, runs its LM steps and asks for its next points -- with 256 windows of
	// the reference configuration the host side (16 ms of LM steps and requests per solve) had
	// grown to half of the wall time once the kernels got faster.  Per window the sequence of
	// requests and evaluations is unchanged: same results.
	const size_t nfAll = static_cast<size_t>(Wn) * P;
	if (be.pipelined(Wn, nfAll))
	{
		constexpr int kMaxGroups = 4;
		const int G = std::min(std::max(be.groups(), 2), kMaxGroups);
		int rc = be.pipeline_begin(nfAll, G);
		if (rc)
		{
			return rc;
		}
		size_t wSplit[kMaxGroups + 1];
		for (int g = 0; g <= G; ++g)
		{
			wSplit[g] = static_cast<size_t>(Wn) * g / G;
		}
		std::vector<unsigned char> gmodes[kMaxGroups];
		for (int g = 0; g < G; ++g)
		{
			gmodes[g].assign(nfAll, 0);
		}
		bool inflight[kMaxGroups] = {false, false, false, false}, gJac[kMaxGroups] = {false, false, false, false};
		// the windows of a group that have not finished (a finished window never asks again: its mode stays 0)
		std::vector<size_t> running[kMaxGroups];
		for (int g = 0; g < G; ++g)
		{
			for (size_t w = wSplit[g]; w < wSplit[g + 1]; ++w)
			{
				running[g].push_back(w);
			}
		}
		// request: every running window of the group says what it wants next; true if any is still running
		auto request = [&](int g) {
			std::vector<size_t>& run = running[g];
			const bool speculate = specMax != 0 && run.size() <= specMax;
			// (a request is a copy of 2 P doubles and a memset: threads only from 64 windows per thread up)
			pool.parallel_for(run.size(), 64, [&](size_t b, size_t e) {
				for (size_t k = b; k < e; ++k)
				{
					const size_t w = run[k];
					wmode[w] = ask(w, speculate);
					std::memset(&gmodes[g][w * P], wmode[w], static_cast<size_t>(P));
				}
			});
			gJac[g] = false;
			size_t keep = 0;
			for (size_t k = 0; k < run.size(); ++k)
			{
				const size_t w = run[k];
				if (wmode[w] != 0)
				{
					gJac[g] = gJac[g] || wmode[w] == 2;
					run[keep++] = w;
				}
			}
			run.resize(keep);
			return keep != 0;
		};
		auto launch = [&](int g) {
			inflight[g] = true;
			++rounds;
			const auto tl = now();
			struct Acc
			{
				double& t;
				std::chrono::steady_clock::time_point t0;
				~Acc() { t += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
			} acc{tLaunch, tl};
			return be.eval_begin(flows.data(), gmodes[g].data(), g, wSplit[g] * P, wSplit[g + 1] * P, gJac[g], P);
		};
		for (int g = 0; g < G && rc == EBO_OK; ++g)
		{
			if (request(g))
			{
				rc = launch(g);
			}
		}
		auto anyInflight = [&] { for (int g = 0; g < G; ++g) { if (inflight[g]) return true; } return false; };
		std::vector<size_t> live;
		live.reserve(static_cast<size_t>(Wn));
		while (rc == EBO_OK && anyInflight())
		{
			for (int g = 0; g < G && rc == EBO_OK; ++g)
			{
				if (!inflight[g])
				{
					continue;
				}
				const auto t1 = now();
				rc = be.eval_finish(gmodes[g].data(), g, wSplit[g] * P, wSplit[g + 1] * P, gJac[g], r.data(), J.data(), P);
				inflight[g] = false;
				if (rc)
				{
					break;
				}
				const auto t2 = now();
				// the windows of the group that are still running (late in a solve: a few stragglers, which
				// then do not pay for waking the pool)
				live = running[g];
				pool.parallel_for(live.size(), 2, [&](size_t b, size_t e) {
					for (size_t k = b; k < e; ++k)
					{
						const size_t w = live[k];
						lm[w].supply(&r[w * P], wmode[w] == 2 ? &J[w * P * 2] : nullptr);
					}
				});
				const auto t3 = now();
				const bool more = request(g);
				tEval += ms(t1, t2);
				tSup += ms(t2, t3);
				const auto t4 = now();
				tReq += ms(t3, t4);
				if (trace && ab_env("EBO_SOLVE_TRACE_ROUNDS"))
				{
					std::fprintf(stderr, "[ebo]   round %4d group %d live %4zu jac %d: wait %.3f supply %.3f request %.3f ms\n", rounds, g,
								 live.size(), gJac[g] ? 1 : 0, ms(t1, t2), ms(t2, t3), ms(t3, t4));
				}
				if (more)
				{
					rc = launch(g);
				}
			}
		}
		be.pipeline_end(G);
		if (rc)
		{
			return rc;
		}
	}
	else
	for (;;)
	{
		const auto t0 = now();
		// every window says what it wants next (its own point, value or value + Jacobian);
		// finished windows drop out of the launch
		size_t runningNow = 0;
		for (int w = 0; w < Wn; ++w)
		{
			runningNow += (rounds == 0 || wmode[w] != 0) ? 1 : 0;
		}
		const bool speculate = specMax != 0 && runningNow <= specMax;
		pool.parallel_for(static_cast<size_t>(Wn), 8, [&](size_t b, size_t e) {
			for (size_t w = b; w < e; ++w)
			{
				if (rounds != 0 && wmode[w] == 0)
				{
					continue;  // finished in an earlier round (its mode table entries are 0 already)
				}
				wmode[w] = ask(w, speculate);
				std::memset(&modes[w * P], wmode[w], static_cast<size_t>(P));
			}
		});
		bool any = false, anyJac = false, uniform = true;
		for (int w = 0; w < Wn; ++w)
		{
			any = any || wmode[w] != 0;
			anyJac = anyJac || wmode[w] == 2;
			uniform = uniform && wmode[w] == wmode[0];
		}
		if (!any)
		{
			break;
		}
		// all windows in the same phase (always so for a single window): no mode table needed
		const auto t1 = now();
		int rc = be.eval(flows.data(), r.data(), anyJac ? J.data() : nullptr, uniform ? nullptr : modes.data(), P);
		if (rc)
		{
			return rc;
		}
		const auto t2 = now();
		pool.parallel_for(static_cast<size_t>(Wn), 2, [&](size_t b, size_t e) {
			for (size_t w = b; w < e; ++w)
			{
				if (wmode[w] != 0)
				{
					lm[w].supply(&r[w * P], wmode[w] == 2 ? &J[w * P * 2] : nullptr);
				}
			}
		});
		const auto t3 = now();
		tReq += ms(t0, t1);
		tEval += ms(t1, t2);
		tSup += ms(t2, t3);
		++rounds;
	}
	if (trace)
	{
		std::fprintf(stderr, "[ebo] lock-step solve: %d windows, %d rounds: request %.2f ms, evaluation (staging + kernels + sync; pipelined: waiting only) %.2f ms, supply (LM steps) %.2f ms, launches (pipelined) %.2f ms, driver total %.2f ms\n",
					 Wn, rounds, tReq, tEval, tSup, tLaunch, ms(tStart, now()));
	}
	return 0;
}

// One 2-parameter HostLm per active patch (lms[k] owns flow slot slot[k] of nf), all advanced in lock step.
template <class Backend>
int lockstep_independent(Backend& be, std::vector<HostLm>& lms, const std::vector<size_t>& slot, size_t nf,
						 std::vector<double>& flows)
{
	flows.assign(nf * 2, 0.0);
	std::vector<double> r(nf), J(nf * 2);
	std::vector<unsigned char> modes(nf, 0);
	HostPool pool(lms.size(), 256);
	for (;;)
	{
		// every patch says what it wants next; finished patches drop out of the launch
		pool.parallel_for(lms.size(), 256, [&](size_t b, size_t e) {
			for (size_t k = b; k < e; ++k)
			{
				const HostLm::Request q = lms[k].request(&flows[2 * slot[k]]);
				modes[slot[k]] = q == HostLm::DONE ? 0 : (q == HostLm::NEED_JACOBIAN ? 2 : 1);
			}
		});
		bool any = false, anyJac = false;
		for (size_t k = 0; k < lms.size(); ++k)
		{
			any = any || modes[slot[k]] != 0;
			anyJac = anyJac || modes[slot[k]] == 2;
		}
		if (!any)
		{
			break;
		}
		const int rc = be.eval(flows.data(), r.data(), anyJac ? J.data() : nullptr, modes.data());
		if (rc)
		{
			return rc;
		}
		pool.parallel_for(lms.size(), 256, [&](size_t b, size_t e) {
			for (size_t k = b; k < e; ++k)
			{
				if (modes[slot[k]] != 0)
				{
					lms[k].supply(&r[slot[k]], modes[slot[k]] == 2 ? &J[2 * slot[k]] : nullptr);
				}
			}
		});
	}
	return 0;
}
}  // namespace ebo
