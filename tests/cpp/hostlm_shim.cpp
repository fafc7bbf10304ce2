// hostlm_shim.cpp — C entry points around the PRODUCT's host solver (csrc/host_lm.cpp) so that
// the CPU test-suite can drive it without a GPU: the test supplies the data-term residuals and
// Jacobians (from the oracle) that the device would supply.  Test infrastructure.
#include "../../event-based-odomety_amd/csrc/host_lm.h"

extern "C" {
void* hlm_create(int npx, int npy, const unsigned char* active, double tvWeight, double tvHuber,
				 const ebo_solver_opts* o)
{
	std::vector<uint8_t> a(active, active + npx * npy);
	return new ebo::HostLm(npx, npy, a, tvWeight, tvHuber, *o);
}
int hlm_request(void* h, double* flows) { return static_cast<int>(static_cast<ebo::HostLm*>(h)->request(flows)); }
void hlm_supply(void* h, const double* r, const double* J) { static_cast<ebo::HostLm*>(h)->supply(r, J); }
void hlm_result(void* h, double* flows) { static_cast<ebo::HostLm*>(h)->result(flows); }
void hlm_stats(void* h, int* out4, double* out2)
{
	const ebo::HostLm::Stats& s = static_cast<ebo::HostLm*>(h)->stats();
	out4[0] = s.iterations;
	out4[1] = s.evals_cost;
	out4[2] = s.evals_jac;
	out4[3] = s.termination;
	out2[0] = s.initial_cost;
	out2[1] = s.final_cost;
}
void hlm_destroy(void* h) { delete static_cast<ebo::HostLm*>(h); }
}
