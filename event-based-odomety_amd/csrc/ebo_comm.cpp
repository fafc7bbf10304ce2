// ebo_comm.cpp — sharding helper and the RCCL exchange of include/ebo.h (librccl.so is dlopened).
#include "ebo_ctx.h"

using namespace ebo;

extern "C" {

int ebo_shard_range(int n_units, int rank, int world, int* begin, int* end)
{
	if (n_units < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end)
	{
		return EBO_ERR_ARG;
	}
	const int base = n_units / world, rem = n_units % world;
	*begin = rank * base + std::min(rank, rem);
	*end = *begin + base + (rank < rem ? 1 : 0);
	return EBO_OK;
}

// ---- RCCL exchange without any framework --------------------------------------
// librccl.so is loaded lazily with RTLD_LOCAL on the first ebo_comm_* call: a process that
// brings its own RCCL (PyTorch does) and never calls these entry points is not affected.
namespace
{
struct RcclApi
{
	void* handle = nullptr;
	int (*GetUniqueId)(void*) = nullptr;
	int (*CommInitRank)(void**, int, ebo_comm_id, int) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
	int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t) = nullptr;
	int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	int (*CommDestroy)(void*) = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
};

RcclApi* rccl_api(std::string& err)
{
	static RcclApi api;
	if (api.handle)
	{
		return &api;
	}
	const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
	void* h = nullptr;
	for (const char* n : names)
	{
		h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
		if (h)
		{
			break;
		}
	}
	if (!h)
	{
		err = std::string("cannot load librccl.so: ") + dlerror();
		return nullptr;
	}
	api.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclGetUniqueId"));
	api.CommInitRank = reinterpret_cast<int (*)(void**, int, ebo_comm_id, int)>(dlsym(h, "ncclCommInitRank"));
	api.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(h, "ncclAllGather"));
	api.AllReduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclAllReduce"));
	api.Reduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t)>(dlsym(h, "ncclReduce"));
	api.Send = reinterpret_cast<int (*)(const void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclSend"));
	api.Recv = reinterpret_cast<int (*)(void*, size_t, int, int, void*, hipStream_t)>(dlsym(h, "ncclRecv"));
	api.GroupStart = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupStart"));
	api.GroupEnd = reinterpret_cast<int (*)()>(dlsym(h, "ncclGroupEnd"));
	api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
	api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
	if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.AllReduce || !api.Reduce || !api.CommDestroy || !api.Send ||
		!api.Recv || !api.GroupStart || !api.GroupEnd)
	{
		err = "librccl.so lacks the expected nccl* symbols";
		dlclose(h);
		return nullptr;
	}
	api.handle = h;
	return &api;
}
}  // namespace

int ebo_comm_unique_id(ebo_comm_id* id)
{
	if (!id)
	{
		return EBO_ERR_ARG;
	}
	RcclApi* api = rccl_api(g_create_error);
	if (!api)
	{
		return EBO_ERR_COMM;
	}
	return api->GetUniqueId(id) == 0 ? EBO_OK : EBO_ERR_COMM;
}

int ebo_comm_init(ebo_ctx* c, const ebo_comm_id* id, int rank, int nranks)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!id || nranks <= 0 || rank < 0 || rank >= nranks)
	{
		return c->fail(EBO_ERR_ARG, "bad communicator arguments");
	}
	RcclApi* api = rccl_api(c->err);
	if (!api)
	{
		return EBO_ERR_COMM;
	}
	(void)hipSetDevice(c->prm.device);
	if (c->comm)
	{
		api->CommDestroy(c->comm);
		c->comm = nullptr;
	}
	const int rc = api->CommInitRank(&c->comm, nranks, *id, rank);
	if (rc != 0)
	{
		c->comm = nullptr;
		return c->fail(EBO_ERR_COMM, std::string("ncclCommInitRank: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"));
	}
	c->comm_rank = rank;
	c->comm_size = nranks;
	// the exchange's own small buffers, so that no collective path has to allocate before its first collective
	// (a new communicator starts with an empty exchange buffer: its growth is a function of what THIS communicator's
	// ranks have gathered, so that every rank holds the same capacity)
	hipFree(c->d_comm_buf);
	c->d_comm_buf = nullptr;
	c->comm_buf_cap = 0;
	hipFree(c->d_comm_cnt);
	c->d_comm_cnt = nullptr;
	if (c->pin_comm)
	{
		hipHostFree(c->pin_comm);
		c->pin_comm = nullptr;
	}
	const size_t words = static_cast<size_t>(nranks) + 2;
	hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_comm_cnt), words * sizeof(uint64_t));
	if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->pin_comm), words * sizeof(uint64_t), hipHostMallocDefault);
	if (e != hipSuccess)
	{
		api->CommDestroy(c->comm);
		c->comm = nullptr;
		return c->hip(e, "communicator staging");
	}
	return EBO_OK;
}

int ebo_allgather_device(ebo_ctx* c, const double* d_send, double* d_recv, size_t count_per_rank)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!c->comm)
	{
		return c->fail(EBO_ERR_STATE, "no communicator: call ebo_comm_init first");
	}
	if (!d_send || !d_recv)
	{
		return c->fail(EBO_ERR_ARG, "null device pointer");
	}
	std::string err;
	RcclApi* api = rccl_api(err);
	const int rc = api->AllGather(d_send, d_recv, count_per_rank, 8 /* ncclFloat64 */, c->comm, c->stream);
	if (rc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("ncclAllGather: ") + (api->GetErrorString ? api->GetErrorString(rc) : "error"));
	}
	return EBO_OK;
}

int ebo_comm_size(const ebo_ctx* c, int* rank, int* nranks)
{
	if (!c || (!rank && !nranks))
	{
		return EBO_ERR_ARG;
	}
	if (rank)
	{
		*rank = c->comm ? c->comm_rank : 0;
	}
	if (nranks)
	{
		*nranks = c->comm ? c->comm_size : 1;
	}
	return EBO_OK;
}

int ebo_reduce_sum_device(ebo_ctx* c, const double* d_send, double* d_recv, size_t count, int root)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!c->comm)
	{
		return c->fail(EBO_ERR_STATE, "no communicator: call ebo_comm_init first");
	}
	if (!d_send || (!d_recv && (root < 0 || root == c->comm_rank)) || root >= c->comm_size)
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_reduce_sum_device");
	}
	std::string err;
	RcclApi* api = rccl_api(err);
	const int rc = root < 0 ? api->AllReduce(d_send, d_recv, count, 8 /* ncclFloat64 */, 0 /* ncclSum */, c->comm, c->stream)
							: api->Reduce(d_send, d_recv, count, 8, 0, root, c->comm, c->stream);
	if (rc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string(root < 0 ? "ncclAllReduce: " : "ncclReduce: ") +
										 (api->GetErrorString ? api->GetErrorString(rc) : "error"));
	}
	return EBO_OK;
}

// Config 5's exchange: variable-length per-rank lists of 32-byte track records.
// Layout of the context's scratch: [send: maxN records][recv: nranks x maxN records]; the
// counts travel first through the head of the same buffer.
//
// Collective discipline: a rank must never leave between two collectives on a condition the other
// ranks do not share, or they wait in the next collective for ever.  So everything a rank decides
// BEFORE the second all-gather is decided from the gathered counts (the same on every rank); the
// conditions that are this rank's own (output buffer too small, its count echoed back changed) are
// reported AFTER the rank has taken part in the second all-gather.
namespace
{
const uint64_t kPoisonCount = ~0ull;

// the counts collective alone: cnt[nranks] on every rank.  Nothing on this path allocates (d_comm_cnt and pin_comm
// belong to the communicator, ebo_comm_init), and a rank whose staging copy fails still takes part: its slot keeps
// the poison it was preset with, every rank sees it and every rank reports EBO_ERR_COMM -- nobody is left waiting.
int gather_track_counts(ebo_ctx* c, RcclApi* api, size_t n_local, std::vector<uint64_t>& cnt)
{
	const size_t nr = static_cast<size_t>(c->comm_size);
	uint64_t* d_cnt = c->d_comm_cnt;
	cnt.assign(nr, 0);
	c->pin_comm[0] = n_local;
	hipError_t e = hipMemsetAsync(d_cnt, 0xFF, sizeof(uint64_t), c->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_cnt, c->pin_comm, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
	const int nrc = api->AllGather(d_cnt, d_cnt + 1, 1, 5 /* ncclUint64 */, c->comm, c->stream);
	if (nrc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("ncclAllGather(counts): ") + (api->GetErrorString ? api->GetErrorString(nrc) : "error"));
	}
	hipError_t e2 = hipMemcpyAsync(c->pin_comm + 1, d_cnt + 1, nr * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
	if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
	if (e != hipSuccess || e2 != hipSuccess)
	{
		return c->hip(e != hipSuccess ? e : e2, "track counts");
	}
	for (size_t q = 0; q < nr; ++q)
	{
		cnt[q] = c->pin_comm[1 + q];
		if (cnt[q] == kPoisonCount)
		{
			return c->fail(EBO_ERR_COMM, "a rank could not stage its track count");
		}
	}
	return EBO_OK;
}

// The communicator's exchange buffer, grown by the same rule on every rank (from the gathered counts), so that
// "it has to grow" is a fact all ranks share; then growing is collective too: every rank tries, the ranks agree
// (one int, ncclAllReduce min) and either all have the larger buffer or all report the failure.
int ensure_comm_buf(ebo_ctx* c, RcclApi* api, size_t bytes)
{
	if (bytes <= c->comm_buf_cap)
	{
		return EBO_OK;
	}
	void* fresh = nullptr;
	const hipError_t ea = hipMalloc(&fresh, bytes);
	if (ea != hipSuccess)
	{
		(void)hipGetLastError();
		fresh = nullptr;
	}
	int32_t* d_ok = reinterpret_cast<int32_t*>(c->d_comm_cnt);
	reinterpret_cast<int32_t*>(c->pin_comm)[0] = fresh ? 1 : 0;
	hipError_t e = hipMemsetAsync(d_ok, 0, sizeof(int32_t), c->stream);  // a failed copy leaves "not ok"
	if (e == hipSuccess) e = hipMemcpyAsync(d_ok, c->pin_comm, sizeof(int32_t), hipMemcpyHostToDevice, c->stream);
	const int nrc = api->AllReduce(d_ok, d_ok, 1, 2 /* ncclInt32 */, 3 /* ncclMin */, c->comm, c->stream);
	hipError_t e2 = hipSuccess;
	if (nrc == 0)
	{
		e2 = hipMemcpyAsync(c->pin_comm, d_ok, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream);
		if (e2 == hipSuccess) e2 = hipStreamSynchronize(c->stream);
	}
	const bool allOk = nrc == 0 && e == hipSuccess && e2 == hipSuccess && reinterpret_cast<int32_t*>(c->pin_comm)[0] == 1;
	if (!allOk)
	{
		if (fresh)
		{
			hipFree(fresh);
		}
		return nrc != 0 ? c->fail(EBO_ERR_COMM, "ncclAllReduce(buffer growth)")
						: c->fail(EBO_ERR_HIP, "a rank is out of device memory for the track exchange");
	}
	hipFree(c->d_comm_buf);
	c->d_comm_buf = fresh;
	c->comm_buf_cap = bytes;
	return EBO_OK;
}
}  // namespace

int ebo_allgather_track_counts(ebo_ctx* c, size_t n_local, size_t* n_all, size_t* counts)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!c->comm)
	{
		return c->fail(EBO_ERR_STATE, "no communicator: call ebo_comm_init first");
	}
	if (!n_all)
	{
		return c->fail(EBO_ERR_ARG, "null pointer");
	}
	std::string err;
	RcclApi* api = rccl_api(err);
	(void)hipSetDevice(c->prm.device);
	std::vector<uint64_t> cnt;
	const int rc = gather_track_counts(c, api, n_local, cnt);
	if (rc)
	{
		return rc;
	}
	size_t total = 0;
	for (size_t q = 0; q < cnt.size(); ++q)
	{
		total += cnt[q];
		if (counts)
		{
			counts[q] = cnt[q];
		}
	}
	*n_all = total;
	return EBO_OK;
}

int ebo_allgather_tracks(ebo_ctx* c, const ebo_track_point* local, size_t n_local, ebo_track_point* all,
						 size_t cap, size_t* n_all, size_t* counts)
{
	static_assert(sizeof(ebo_track_point) == 32, "track record layout");
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (!c->comm)
	{
		return c->fail(EBO_ERR_STATE, "no communicator: call ebo_comm_init first");
	}
	if ((n_local && !local) || !n_all || (cap && !all))
	{
		return c->fail(EBO_ERR_ARG, "null pointer");
	}
	std::string err;
	RcclApi* api = rccl_api(err);
	(void)hipSetDevice(c->prm.device);
	const size_t nr = static_cast<size_t>(c->comm_size);
	// 1. counts
	std::vector<uint64_t> cnt;
	int rc = gather_track_counts(c, api, n_local, cnt);
	if (rc)
	{
		return rc;
	}
	size_t total = 0, maxN = 0;
	for (size_t q = 0; q < nr; ++q)
	{
		total += cnt[q];
		maxN = std::max<size_t>(maxN, cnt[q]);
		if (counts)
		{
			counts[q] = cnt[q];
		}
	}
	*n_all = total;
	if (maxN == 0)  // the same on every rank
	{
		return EBO_OK;
	}
	// 2. one all-gather of max-padded records -- every rank takes part, whatever its own buffer holds
	const bool fits = total <= cap;
	const bool echoed = cnt[static_cast<size_t>(c->comm_rank)] == n_local;
	const size_t slot = maxN * sizeof(ebo_track_point);
	rc = ensure_comm_buf(c, api, (nr + 1) * slot);  // a collective decision: all ranks go on, or all return here
	if (rc)
	{
		return rc;
	}
	char* d_send = static_cast<char*>(c->d_comm_buf);
	char* d_recv = d_send + slot;
	hipError_t e = hipMemsetAsync(d_send, 0, slot, c->stream);
	if (e == hipSuccess && n_local)
	{
		e = hipMemcpyAsync(d_send, local, std::min<size_t>(n_local, maxN) * sizeof(ebo_track_point), hipMemcpyHostToDevice,
						   c->stream);
	}
	const int nrc = api->AllGather(d_send, d_recv, slot, 0 /* ncclInt8 */, c->comm, c->stream);
	if (e != hipSuccess)
	{
		(void)hipStreamSynchronize(c->stream);
		return c->hip(e, "H2D tracks");
	}
	if (nrc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("ncclAllGather(tracks): ") + (api->GetErrorString ? api->GetErrorString(nrc) : "error"));
	}
	if (!echoed || !fits)
	{
		(void)hipStreamSynchronize(c->stream);
		return !echoed ? c->fail(EBO_ERR_COMM, "track count of this rank came back changed")
					   : c->fail(EBO_ERR_ARG, "track output buffer too small");
	}
	// compacting copies: rank q's real records only
	size_t at = 0;
	for (size_t q = 0; q < nr && e == hipSuccess; ++q)
	{
		if (cnt[q])
		{
			e = hipMemcpyAsync(all + at, d_recv + q * slot, cnt[q] * sizeof(ebo_track_point), hipMemcpyDeviceToHost, c->stream);
			at += cnt[q];
		}
	}
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	return c->hip(e, "D2H tracks");
}

// The halo exchange of the band-limited final image (ebo_band.cpp): one grouped send / recv with the two
// neighbouring ranks, then the ranks agree on the fallback flag.  Every rank makes the same calls in the same order:
// what differs between ranks (whether it has a neighbour, how many rows) comes from the band plan, which every rank
// derives from the same row bounds.
int ebo_band_exchange_device(ebo_ctx* c, int n_windows, const ebo_band* band, const uint32_t* d_top, const uint32_t* d_bottom,
							 uint32_t* d_from_above, uint32_t* d_from_below, int32_t* d_escaped)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (n_windows <= 0 || !band_ok(c, band) || !d_escaped)
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_band_exchange_device");
	}
	if (!c->comm)
	{
		return EBO_OK;  // no communicator: nothing travels, the rank's own flag is the verdict
	}
	// (a communicator of ONE rank takes the same calls -- an empty group and the reduction of the flag -- so that the one
	// rank a single-GPU box allows runs the code N ranks run, tests/test_gpu_comm.py)
	const size_t W = static_cast<size_t>(c->prm.image_w), nw = static_cast<size_t>(n_windows);
	const size_t up = static_cast<size_t>(band->own_row0 - band->band_row0) * W * nw;
	const size_t down = static_cast<size_t>(band->band_row1 - band->own_row1) * W * nw;
	const size_t fromAbove = static_cast<size_t>(band->recv_above) * W * nw, fromBelow = static_cast<size_t>(band->recv_below) * W * nw;
	if ((up && !d_top) || (down && !d_bottom) || (fromAbove && !d_from_above) || (fromBelow && !d_from_below))
	{
		return c->fail(EBO_ERR_ARG, "null halo buffer");
	}
	std::string err;
	RcclApi* api = rccl_api(err);
	(void)hipSetDevice(c->prm.device);
	const int r = c->comm_rank;
	int nrc = api->GroupStart();
	// (a rank without rows has an empty band: it sends and receives nothing, as its neighbours' plans say)
	if (nrc == 0 && up && r > 0) nrc = api->Send(d_top, up, 3 /* ncclUint32 */, r - 1, c->comm, c->stream);
	if (nrc == 0 && down && r < c->comm_size - 1) nrc = api->Send(d_bottom, down, 3, r + 1, c->comm, c->stream);
	if (nrc == 0 && fromAbove && r > 0) nrc = api->Recv(d_from_above, fromAbove, 3, r - 1, c->comm, c->stream);
	if (nrc == 0 && fromBelow && r < c->comm_size - 1) nrc = api->Recv(d_from_below, fromBelow, 3, r + 1, c->comm, c->stream);
	const int erc = api->GroupEnd();
	if (nrc == 0) nrc = erc;
	if (nrc == 0) nrc = api->AllReduce(d_escaped, d_escaped, 1, 2 /* ncclInt32 */, 2 /* ncclMax */, c->comm, c->stream);
	if (nrc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("band exchange: ") + (api->GetErrorString ? api->GetErrorString(nrc) : "error"));
	}
	return EBO_OK;
}

int ebo_band_gather_device(ebo_ctx* c, int n_windows, const int* row_bounds, const double* d_image_own, int root, double* d_full)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	const int nr = c->comm ? c->comm_size : 1, me = c->comm ? c->comm_rank : 0;
	if (n_windows <= 0 || !row_bounds || root < 0 || root >= nr || (me == root && !d_full) || row_bounds[0] != 0 ||
		row_bounds[nr] != c->prm.image_h)
	{
		return c->fail(EBO_ERR_ARG, "bad argument to ebo_band_gather_device");
	}
	for (int q = 0; q < nr; ++q)
	{
		if (row_bounds[q + 1] < row_bounds[q])
		{
			return c->fail(EBO_ERR_ARG, "row bounds must not decrease");
		}
	}
	(void)hipSetDevice(c->prm.device);
	const size_t W = static_cast<size_t>(c->prm.image_w), H = static_cast<size_t>(c->prm.image_h);
	const size_t mine = static_cast<size_t>(row_bounds[me + 1] - row_bounds[me]) * W;
	if (mine && !d_image_own)
	{
		return c->fail(EBO_ERR_ARG, "null own image");
	}
	if (!c->comm)
	{
		return c->hip(hipMemcpyAsync(d_full, d_image_own, static_cast<size_t>(n_windows) * H * W * sizeof(double), hipMemcpyDeviceToDevice,
									 c->stream),
					  "own image -> full image");
	}
	// (a communicator of one rank sends to and receives from itself inside the group: the code path of N ranks)
	std::string err;
	RcclApi* api = rccl_api(err);
	int nrc = api->GroupStart();
	for (int w = 0; w < n_windows && nrc == 0; ++w)
	{
		if (mine)
		{
			nrc = api->Send(d_image_own + static_cast<size_t>(w) * mine, mine, 8 /* ncclFloat64 */, root, c->comm, c->stream);
		}
		for (int q = 0; me == root && q < nr && nrc == 0; ++q)
		{
			const size_t rows = static_cast<size_t>(row_bounds[q + 1] - row_bounds[q]);
			if (rows)
			{
				nrc = api->Recv(d_full + static_cast<size_t>(w) * H * W + static_cast<size_t>(row_bounds[q]) * W, rows * W, 8, q, c->comm,
								c->stream);
			}
		}
	}
	const int erc = api->GroupEnd();
	if (nrc == 0) nrc = erc;
	if (nrc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("band gather: ") + (api->GetErrorString ? api->GetErrorString(nrc) : "error"));
	}
	return EBO_OK;
}

int ebo_comm_destroy(ebo_ctx* c)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, kNotWhileRecording);
	}
	if (c->comm)
	{
		std::string err;
		RcclApi* api = rccl_api(err);
		if (api)
		{
			(void)hipStreamSynchronize(c->stream);
			api->CommDestroy(c->comm);
		}
		c->comm = nullptr;
	}
	// what ebo_comm_init and the exchanges allocated goes with the communicator (it used to wait for the next
	// ebo_comm_init: a destroyed context leaked device and pinned memory)
	if (c->d_comm_cnt || c->d_comm_buf || c->pin_comm)
	{
		(void)hipSetDevice(c->prm.device);
		(void)hipStreamSynchronize(c->stream);
		hipFree(c->d_comm_cnt);
		c->d_comm_cnt = nullptr;
		hipFree(c->d_comm_buf);
		c->d_comm_buf = nullptr;
		c->comm_buf_cap = 0;
		if (c->pin_comm)
		{
			hipHostFree(c->pin_comm);
			c->pin_comm = nullptr;
		}
	}
	c->comm_rank = 0;
	c->comm_size = 1;
	return EBO_OK;
}

}  // extern "C"
