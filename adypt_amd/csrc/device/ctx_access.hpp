// Internal seam between tracer.hip (owner of struct adypt_ctx) and multi.hip (RCCL gather of the tile shards): the few
// fields the collective needs, without exposing the context's layout.  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <string>

struct adypt_ctx;

namespace adypt {

struct CtxInfo {
	int device;
	hipStream_t stream;            // the context's own (non-blocking) stream: collectives are enqueued behind the rendering
	int rank, nranks, width, height;
	int n_local_px;                // owned blocks x 1024
	float4 *accum;                 // compact block-major running mean (image 0) of the owned blocks
};

CtxInfo ctx_info(adypt_ctx *c);
void ctx_set_error(adypt_ctx *c, const std::string &msg);
// where multi.hip parks its per-context communicator (freed by adypt_destroy through *free_fn)
void **ctx_comm_slot(adypt_ctx *c, void (***free_fn)(void *));

}  // namespace adypt
