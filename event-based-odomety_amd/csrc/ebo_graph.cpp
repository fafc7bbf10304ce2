// ebo_graph.cpp — HIP graphs over the asynchronous *_device entry points (include/ebo.h: ebo_graph_*).
//
// A launch-bound loop (one batched evaluation per step, a solve + its count image per window) pays, per step,
// the host side of every launch: argument set-up, geometry, the runtime's packet build.  Recorded once into a
// hipGraph the whole step replays with one call and no host work in between, so the stream never runs dry
// between steps (bench.py: ms_per_step == the kernel's own duration).
#include "ebo_ctx.h"

struct ebo_graph
{
	hipGraph_t graph = nullptr;
	hipGraphExec_t exec = nullptr;
	int device = 0;
};

extern "C" {

int ebo_graph_begin(ebo_ctx* c)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, "ebo_graph_begin: already recording");
	}
	if (c->stream == nullptr)
	{
		// the legacy default stream cannot be captured (and trying poisons it for the process)
		return c->fail(EBO_ERR_STATE, "ebo_graph_begin: the context launches on the default stream; give it a created stream "
									  "(ebo_set_stream) or keep its own");
	}
	(void)hipSetDevice(c->prm.device);
	// thread-local mode: other threads of the process (another context, PyTorch's allocator) keep working
	const int rc = c->hip(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
	c->capturing = rc == EBO_OK;
	return rc;
}

int ebo_graph_end(ebo_ctx* c, ebo_graph** out)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!c->capturing)
	{
		return c->fail(EBO_ERR_STATE, "ebo_graph_end without ebo_graph_begin");
	}
	c->capturing = false;
	hipGraph_t g = nullptr;
	// always ends the capture, also when a recorded call failed (the stream must not stay in capture mode)
	hipError_t e = hipStreamEndCapture(c->stream, &g);
	if (e != hipSuccess || !g)
	{
		(void)hipGetLastError();
		return c->hip(e != hipSuccess ? e : hipErrorUnknown, "hipStreamEndCapture (a recorded call synchronised or allocated?)");
	}
	if (!out)
	{
		hipGraphDestroy(g);
		return c->fail(EBO_ERR_ARG, "null graph pointer");
	}
	hipGraphExec_t x = nullptr;
	e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
	if (e != hipSuccess)
	{
		hipGraphDestroy(g);
		return c->hip(e, "hipGraphInstantiate");
	}
	ebo_graph* r = new ebo_graph;
	r->graph = g;
	r->exec = x;
	r->device = c->prm.device;
	*out = r;
	return EBO_OK;
}

int ebo_graph_launch(ebo_ctx* c, ebo_graph* g, int times)
{
	if (!c)
	{
		return EBO_ERR_ARG;
	}
	if (!g || !g->exec || times < 0)
	{
		return c->fail(EBO_ERR_ARG, "bad graph or repeat count");
	}
	if (c->capturing)
	{
		return c->fail(EBO_ERR_STATE, "ebo_graph_launch while recording");
	}
	(void)hipSetDevice(c->prm.device);
	for (int i = 0; i < times; ++i)
	{
		const hipError_t e = hipGraphLaunch(g->exec, c->stream);
		if (e != hipSuccess)
		{
			return c->hip(e, "hipGraphLaunch");
		}
	}
	return EBO_OK;
}

void ebo_graph_destroy(ebo_graph* g)
{
	if (!g)
	{
		return;
	}
	(void)hipSetDevice(g->device);
	if (g->exec)
	{
		hipGraphExecDestroy(g->exec);
	}
	if (g->graph)
	{
		hipGraphDestroy(g->graph);
	}
	delete g;
}

}  // extern "C"
