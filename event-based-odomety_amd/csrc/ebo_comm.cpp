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
	api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
	api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
	if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.AllReduce || !api.Reduce || !api.CommDestroy)
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
// the counts collective alone: cnt[nranks] on every rank
int gather_track_counts(ebo_ctx* c, RcclApi* api, size_t n_local, std::vector<uint64_t>& cnt)
{
	const size_t nr = static_cast<size_t>(c->comm_size);
	int rc = ensure_scratch(c, (nr + 1) * sizeof(uint64_t));
	if (rc)
	{
		return rc;
	}
	uint64_t* d_cnt = static_cast<uint64_t*>(c->d_scratch);
	const uint64_t mine = n_local;
	cnt.assign(nr, 0);
	hipError_t e = hipMemcpyAsync(d_cnt, &mine, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // `mine` is a local
	if (e != hipSuccess)
	{
		return c->hip(e, "H2D track count");
	}
	const int nrc = api->AllGather(d_cnt, d_cnt + 1, 1, 5 /* ncclUint64 */, c->comm, c->stream);
	if (nrc != 0)
	{
		return c->fail(EBO_ERR_COMM, std::string("ncclAllGather(counts): ") + (api->GetErrorString ? api->GetErrorString(nrc) : "error"));
	}
	e = hipMemcpyAsync(cnt.data(), d_cnt + 1, nr * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	return c->hip(e, "track counts");
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
	rc = ensure_scratch(c, (nr + 1) * slot);
	if (rc)
	{
		return rc;  // out of device memory: nothing a rank can do for the others here
	}
	char* d_send = static_cast<char*>(c->d_scratch);
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
	return EBO_OK;
}

}  // extern "C"
