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
	api.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(h, "ncclCommDestroy"));
	api.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(h, "ncclGetErrorString"));
	if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy)
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

int ebo_comm_destroy(ebo_ctx* c)
{
	if (!c)
	{
		return EBO_ERR_ARG;
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
