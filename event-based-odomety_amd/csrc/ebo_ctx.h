// ebo_ctx.h — private to the host side of libebo_hip.so: the context behind the opaque ebo_ctx
// of include/ebo.h and the few helpers its translation units share (ebo_api.cpp: context, windows,
// evaluation, solves, count images; ebo_windows.cpp: loading events; ebo_tracker.cpp: tracked patches; ebo_motion_field.cpp;
// ebo_io.cpp; ebo_comm.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "../../include/ebo.h"
#include "ebo_internal.h"
#include "field_tv.h"
#include "host_lm.h"

using namespace ebo;

namespace ebo_host
{
inline size_t env_size(const char* name, size_t dflt)
{
	const char* v = std::getenv(name);
	if (!v || !*v)
	{
		return dflt;
	}
	return static_cast<size_t>(std::strtoull(v, nullptr, 10));
}

struct WindowInfo
{
	int64_t t_ref;
	uint64_t n_events;
};
}  // namespace ebo_host
using namespace ebo_host;

struct ebo_ctx
{
	ebo_params prm;
	int npx = 0, npy = 0, P = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	bool capturing = false;  // between ebo_graph_begin and ebo_graph_end
	std::string err;

	size_t cap_events = 0;
	int cap_windows = 0;
	int n_windows = 0;

	uint64_t* d_events = nullptr;
	Unit* d_units = nullptr;
	int32_t* d_unit_maxdt = nullptr;  // [units] max |t_ref(window) - t| over the unit's events (count kernels' displacement bound)
	double* d_flows = nullptr;
	double* d_out = nullptr;
	double* d_partials = nullptr;
	size_t partials_cap = 0;
	int32_t* d_counts = nullptr;
	double* d_image = nullptr;
	void* d_aux = nullptr;
	size_t aux_cap = 0;
	unsigned char* d_modes = nullptr;        // per-flow-slot evaluation modes of a lock-step solve
	size_t modes_cap = 0;
	const unsigned char* modes_active = nullptr;  // non-null only inside eval_host(modes)
	LiveWindows live_active;                      // n > 0 only inside a compact round of a pipelined lock-step solve
	double2* d_opt_grid = nullptr;   // Optimizer::setGrad's interleaved gradient grid [H][W]
	bool opt_grid_valid = false;
	void* d_opt = nullptr;           // scratch of ebo_optimizer_eval / _solve
	size_t opt_cap = 0;
	unsigned long long* d_count_ovf = nullptr;  // k_count_bands' overflow list / k_csort_* sorted list
	size_t count_ovf_cap = 0;
	unsigned int* d_count_bins = nullptr;       // k_csort_*: counts, starts, cursors per (window, band)
	size_t count_bins_cap = 0;                  // in bins
	int32_t* d_stats = nullptr;
	void* d_scratch = nullptr;  // patch-integrate staging
	size_t scratch_cap = 0;
	double* d_edge_w = nullptr;      // the 49 tensor weights of the edge loss (device table)
	double edge_w_sigma = -1.0;      // sigma_st they were built for
	unsigned long long* edge_stats_dev = nullptr;  // set only while ebo_edge_work_stats runs its one evaluation
	double* d_edge_cs = nullptr;     // eigenvector directions of the edge loss's eigenvalue pass, [workgroup slot][cap_px][2]
	int n_cus = 0;                   // compute units of the device (0: not asked yet)
	// the compact path's per-launch tables in one allocation: int length (+ 3 ints of padding) | int list[items] (the
	// units whose arrays do not fit the compact layout) | int4 bbox[items] (k_edge_classify's tap bounding boxes)
	int* d_edge_defer = nullptr;
	size_t edge_defer_cap = 0;       // items
	size_t edge_cs_cap = 0;          // bytes
	void* d_edge_scratch = nullptr;  // edge-loss fallback arrays
	size_t edge_scratch_cap = 0;
	void* d_field = nullptr;         // motion field of ebo_init_motion_field (+ its staging)
	size_t field_cap = 0;
	bool field_valid = false;
	const int* d_field_fixed = nullptr;  // fixed points of that field, [field_nfixed][2]
	int field_nfixed = 0;
	void* d_tvf = nullptr;           // workspace of ebo_interpolate_motion_field
	size_t tvf_cap = 0;
	void* comm = nullptr;            // ncclComm_t of ebo_comm_init
	int comm_rank = 0, comm_size = 1;
	uint64_t* d_comm_cnt = nullptr;  // [nranks + 2] counts / flags of the exchange (allocated by ebo_comm_init)
	uint64_t* pin_comm = nullptr;    // the same, pinned host side
	void* d_comm_buf = nullptr;      // track exchange buffer; grown collectively (ebo_comm.cpp: ensure_comm_buf)
	size_t comm_buf_cap = 0;
	void* d_raw = nullptr;           // raw 24-byte (or compact 8-byte) events staged for device bucketing
	hipStream_t copy_stream = nullptr;  // uploads of ebo_set_windows / ebo_set_windows8, overlapped with the bucketing
	hipEvent_t copy_done[8] = {};
	void* d_bucket = nullptr;        // bucketing scratch
	unsigned int* d_chunk_hist = nullptr;  // per-chunk bucket histograms / first ranks of the stable scatter
	size_t chunk_hist_cap = 0;       // in entries
	size_t bucket_cap = 0;

	std::vector<Unit> units;       // [Wn][P+1], stray unit last in each window
	std::vector<int64_t> unit_tref;
	std::vector<int64_t> unit_tmin, unit_tmax;  // ebo_set_patches: earliest / latest event time per unit (ebo_count_image_shard)
	std::vector<int16_t> unit_box;              // ebo_set_patches: [unit][4] = min x, max x, min y, max y of the unit's events
	uint64_t units_gen = 0;                     // bumped by every ebo_set_*: invalidates the cached shard tables below
	void* d_shard_tbl = nullptr;                // BandUnit[units] (band image) / int32 dt_win[units] (dense shard image)
	size_t shard_tbl_cap = 0;
	uint64_t shard_tbl_gen = 0;                 // units_gen the table was built for (0 = none)
	int shard_tbl_kind = 0;                     // 1 = dt_win only, 2 = BandUnit
	std::vector<int64_t> shard_tbl_tref;        // the windows' reference times it was built for
	int* d_escaped_own = nullptr;               // ebo_count_image_band's own flag word
	std::vector<WindowInfo> windows;
	std::vector<uint64_t> h_packed;
	// pinned, device-visible staging of one evaluation round (flows in, (r, J0, J1) out, modes):
	// small rounds let the kernels read and write it directly (no copy packets at all), large
	// ones copy from/to it at DMA speed
	uint32_t* d_route_xy = nullptr;  // ebo_route_set_events: x:16 | y:16 per event of the chunk
	size_t route_cap = 0;
	size_t route_n = 0;
	void* pin_route = nullptr;       // pinned, device-visible arguments and results of ebo_route_events
	size_t pin_route_cap = 0;
	void* pin_bucket = nullptr;      // pinned mirror of the bucketing results (offsets in; units, reference times, flag out)
	size_t pin_bucket_cap = 0;
	double* pin_flows = nullptr;
	double* pin_out = nullptr;
	unsigned char* pin_modes = nullptr;
	size_t pin_cap = 0;

	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	int max_rw = 0, max_rh = 0;
	int grid_max_rw = 0, grid_max_rh = 0;
	// the REGULAR patch of the loaded units (the grid's patch size; of patches loaded by ebo_set_patches the
	// smallest rect, which for a shard of a grid is the grid's regular patch): what the launch shapes follow
	int reg_rw = 0, reg_rh = 0;
	int custom_n = 0;  // > 0: units were loaded by ebo_set_patches (arbitrary rects)

	int cur_patches() const { return custom_n ? custom_n : P; }
	size_t n_flows() const
	{
		return custom_n ? static_cast<size_t>(custom_n) : static_cast<size_t>(n_windows) * P;
	}
	size_t unit_index(int window, int patch) const
	{
		return custom_n ? static_cast<size_t>(patch) : static_cast<size_t>(window) * (P + 1) + patch;
	}

	int fail(int code, const std::string& msg)
	{
		err = msg;
		return code;
	}
	int hip(hipError_t e, const char* what)
	{
		if (e == hipSuccess)
		{
			return EBO_OK;
		}
		err = std::string(what) + ": " + hipGetErrorString(e);
		return EBO_ERR_HIP;
	}
};

// Between ebo_graph_begin and ebo_graph_end only the asynchronous *_device calls may run: anything that copies
// through pageable memory, allocates or synchronises invalidates the recording -- and on ROCm 7.2 leaves the
// process unable to use the stream again -- so every other entry point refuses up front.
static const char* const kNotWhileRecording =
	"not while recording a graph (ebo_graph_begin): only ebo_eval_device, ebo_solve_device and ebo_count_image_device can be recorded";

// helpers defined in ebo_api.cpp and shared by the other host translation units
namespace ebo_host
{
extern thread_local std::string g_create_error;  // errors without a context (ebo_last_error(NULL))
extern const size_t kLdsBudget;
extern const unsigned int kZeroCopyFlags;
ebo::EvalConsts make_consts(const ebo_ctx* c);
void rect_of(const ebo_ctx* c, int px, int py, int& x, int& y, int& w, int& h);
bool mid_timestamp(int64_t a, int64_t b, int64_t& out);
int ensure_scratch(ebo_ctx* c, size_t bytes);
int ensure_aux(ebo_ctx* c, size_t bytes);
ebo::SolveConsts make_solve_consts(const ebo_solver_opts* o);
int check_solver_opts(ebo_ctx* c, const ebo_solver_opts* o);
int shard_table(ebo_ctx* c, int n_windows, const int64_t* window_t_ref_us, const ebo::BandUnit** out, bool* uniformFlows);
bool band_ok(const ebo_ctx* c, const ebo_band* b);
}  // namespace ebo_host
