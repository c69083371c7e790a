// Native multi-GPU boundary (include/adypt_hip.h "native multi-GPU"): the one exchange step of the sharded frame — the
// gather of the fp32 radiance tiles on the root GPU — done with RCCL inside the library, for a single process that owns N
// devices (adypt_multi) and for one process per device (adypt_comm_*).  SURVEY.md §8(e): tiles are disjoint, so there is
// no reduction: every peer sends its compact block-major buffer straight out of its accumulation image to the root over
// its own xGMI link (7 links into the root run in parallel), the root un-tiles all of them with one kernel per rank.
//
// The reference has no distributed code at all (single GL context); the seam this stands behind is OglPathTracer's
// result read-back (src/Tracer/OglPathTracer.cpp:203-205).
//
// librccl is loaded with dlopen on first use: the library has no link-time dependency on it, and a process that already
// holds an RCCL (e.g. the copy PyTorch ships) shares that one instead of bringing a second.
#include "ctx_access.hpp"
#include "host_transport.hpp"
#include "tunables.hpp"
#include "../../../include/adypt_hip.h"
#include "../../../include/adypt_host.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

using namespace adypt;

namespace {

struct RcclApi {
	void *handle = nullptr;
	std::string path;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr; // optional
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
RcclApi g_rccl;
thread_local std::string g_multi_error;

// One RCCL per process: an already loaded copy first (RTLD_NOLOAD), then the system one.  ADYPT_RCCL_LIB overrides.
RcclApi *rccl(std::string *err)
{
	std::lock_guard<std::mutex> lock(g_rccl_mutex);
	if(g_rccl.handle) return &g_rccl;
	const Tunables tun = read_tunables();
	if(tun.comm_transport_host)
	{	// TEST HOOK (host_transport.hpp; only after adypt_enable_test_hooks): the same calls, carried through shared memory, so that N ranks can share one device
		namespace ht = adypt_host_transport;
		RcclApi a;
		a.handle = (void *)&g_rccl; a.path = "host transport (test hook)";
		a.GetUniqueId = ht::GetUniqueId; a.CommInitRank = ht::CommInitRank; a.CommInitAll = ht::CommInitAll; a.CommDestroy = ht::CommDestroy;
		a.Send = ht::Send; a.Recv = ht::Recv; a.AllReduce = ht::AllReduce; a.GroupStart = ht::GroupStart; a.GroupEnd = ht::GroupEnd;
		a.GetErrorString = ht::GetErrorString;
		if(tun.host_transport_timeout_s > 0.0) ht::default_timeout_s() = tun.host_transport_timeout_s;
		g_rccl = a;
		return &g_rccl;
	}
	std::vector<std::pair<std::string, int>> tries;
	if(!tun.rccl_lib.empty()) tries.push_back({tun.rccl_lib, RTLD_NOW | RTLD_LOCAL});
	tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
	tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD});
	tries.push_back({"librccl.so.1", RTLD_NOW | RTLD_LOCAL});
	tries.push_back({"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL});
	tries.push_back({"librccl.so", RTLD_NOW | RTLD_LOCAL});
	std::string last;
	for(const auto &t : tries)
	{
		void *h = dlopen(t.first.c_str(), t.second);
		if(!h) { if(const char *e = dlerror()) last = e; continue; }
		RcclApi a;
		a.handle = h; a.path = t.first;
#define LOAD(field, sym) a.field = (decltype(a.field))dlsym(h, sym)
		LOAD(GetUniqueId, "ncclGetUniqueId"); LOAD(CommInitRank, "ncclCommInitRank"); LOAD(CommInitAll, "ncclCommInitAll");
		LOAD(CommDestroy, "ncclCommDestroy"); LOAD(Send, "ncclSend"); LOAD(Recv, "ncclRecv"); LOAD(AllReduce, "ncclAllReduce");
		LOAD(GroupStart, "ncclGroupStart"); LOAD(GroupEnd, "ncclGroupEnd"); LOAD(GetErrorString, "ncclGetErrorString"); LOAD(CommCount, "ncclCommCount");
#undef LOAD
		if(a.GetUniqueId && a.CommInitRank && a.CommInitAll && a.CommDestroy && a.Send && a.Recv && a.AllReduce && a.GroupStart && a.GroupEnd && a.GetErrorString)
		{
			g_rccl = a;
			return &g_rccl;
		}
		last = t.first + ": not an RCCL (symbols missing)";
		dlclose(h);
	}
	*err = "RCCL is not available (" + last + ")";
	return nullptr;
}

// the shard geometry every rank computes identically: float4 elements of rank r's compact buffer
std::vector<int64_t> shard_counts(int width, int height, int nranks)
{
	std::vector<int64_t> v((size_t)nranks);
	for(int r = 0; r < nranks; ++r) v[(size_t)r] = adypt_shard_block_count(width, height, r, nranks) * 1024;
	return v;
}

// Per-context communicator + root-side buffers.  Parked in adypt_ctx::comm, freed by adypt_destroy.
struct Comm {
	RcclApi *api = nullptr;
	ncclComm_t comm = nullptr;
	int device = 0, rank = 0, nranks = 1;
	std::vector<int64_t> counts;     // float4 per rank
	int64_t stride = 0;              // max of counts: rank r's tiles land at gathered + r * stride
	float4 *gathered = nullptr;      // root only
	float *rgb = nullptr;            // root only: assembled W*H*3
	double *scratch = nullptr;       // all-reduce staging (device)
	static constexpr int kScratch = 64;
	const Tunables tun = read_tunables(); // read once, when the communicator is made (tunables.hpp)
};

void free_comm(void *p)
{
	Comm *k = (Comm *)p;
	if(!k) return;
	(void)hipSetDevice(k->device);
	if(k->comm && k->api) (void)k->api->CommDestroy(k->comm);
	if(k->gathered) (void)hipFree(k->gathered);
	if(k->rgb) (void)hipFree(k->rgb);
	if(k->scratch) (void)hipFree(k->scratch);
	delete k;
}

#define HIP_OK(ctx, expr)                                                                              \
	do {                                                                                               \
		hipError_t e_ = (expr);                                                                        \
		if(e_ != hipSuccess) { ctx_set_error((ctx), std::string(#expr) + ": " + hipGetErrorString(e_)); return ADYPT_E_HIP; } \
	} while(0)
#define NCCL_OK(ctx, api, expr)                                                                        \
	do {                                                                                               \
		ncclResult_t r_ = (expr);                                                                      \
		if(r_ != ncclSuccess) { ctx_set_error((ctx), std::string(#expr) + ": " + (api)->GetErrorString(r_)); return ADYPT_E_HIP; } \
	} while(0)

// Bounds the one exchange of the data path.  A collective whose peer never arrives does not return and cannot be cancelled (the streams are
// blocked behind it); the only safe way out of a process that has touched the GPU is to END it — never to re-exec it.  So: a thread that, if the
// gather has not finished within ADYPT_GATHER_TIMEOUT seconds (default 120, 0 = no watchdog), prints where the gather stands and the state of
// every rank's stream to stderr and exits the process with code 86.  Armed around the RCCL calls and the drains that follow them, nothing else.
class GatherWatchdog {
public:
	GatherWatchdog(double timeout_s, std::function<std::string()> state) : stage_("start")
	{
		if(!(timeout_s > 0.0)) return;
		try
		{
			thread_ = std::thread([this, timeout_s, state] {
				std::unique_lock<std::mutex> lock(m_);
				if(cv_.wait_for(lock, std::chrono::duration<double>(timeout_s), [this] { return done_; })) return;
				fprintf(stderr, "[adypt] gather watchdog: the radiance gather has not finished after %.1f s (stage: %s)\n%s[adypt] exiting the process with code 86 (a collective cannot be cancelled)\n",
						timeout_s, stage_.load(), state().c_str());
				fflush(stderr);
				_exit(86);
			});
		}
		catch(const std::exception &) {} // (no thread to be had: the gather runs unwatched — no exception crosses the C boundary)
	}
	~GatherWatchdog()
	{
		if(!thread_.joinable()) return;
		{ std::lock_guard<std::mutex> lock(m_); done_ = true; }
		cv_.notify_all();
		thread_.join();
	}
	void stage(const char *s) { stage_.store(s); }
private:
	std::mutex m_;
	std::condition_variable cv_;
	bool done_ = false;
	std::atomic<const char *> stage_;
	std::thread thread_;
};
const char *stream_state(int device, hipStream_t stream)
{
	if(hipSetDevice(device) != hipSuccess) return "device not reachable";
	const hipError_t q = hipStreamQuery(stream);
	return q == hipSuccess ? "idle" : q == hipErrorNotReady ? "busy" : hipGetErrorString(q);
}
// TEST HOOK (ADYPT_GATHER_STALL_TEST=1 after adypt_enable_test_hooks): the gather never proceeds — what a lost peer looks like to the caller
void stall_if_asked(const Tunables &tun, GatherWatchdog &wd)
{
	if(!tun.gather_stall_test) return;
	wd.stage("stalled by ADYPT_GATHER_STALL_TEST");
	for(;;) std::this_thread::sleep_for(std::chrono::seconds(1));
}

// buffers of a communicator whose ncclComm_t already exists
int finish_comm(adypt_ctx *ctx, Comm *k)
{
	const CtxInfo i = ctx_info(ctx);
	k->device = i.device; k->rank = i.rank; k->nranks = i.nranks;
	k->counts = shard_counts(i.width, i.height, i.nranks);
	k->stride = std::max<int64_t>(1024, *std::max_element(k->counts.begin(), k->counts.end()));
	HIP_OK(ctx, hipSetDevice(i.device));
	HIP_OK(ctx, hipMalloc((void **)&k->scratch, Comm::kScratch * sizeof(double)));
	if(i.rank == 0)
	{
		HIP_OK(ctx, hipMalloc((void **)&k->gathered, (size_t)k->stride * (size_t)i.nranks * sizeof(float4)));
		HIP_OK(ctx, hipMalloc((void **)&k->rgb, (size_t)i.width * i.height * 3 * sizeof(float)));
	}
	return ADYPT_OK;
}

Comm *comm_of(adypt_ctx *ctx)
{
	void (**free_fn)(void *) = nullptr;
	return (Comm *)*ctx_comm_slot(ctx, &free_fn);
}
void park_comm(adypt_ctx *ctx, Comm *k)
{
	void (**free_fn)(void *) = nullptr;
	void **slot = ctx_comm_slot(ctx, &free_fn);
	if(*slot) free_comm(*slot);
	*slot = k; *free_fn = free_comm;
}

// root side after the tiles have arrived (or for a single rank): own tiles + un-tiling, all on the root's stream
int assemble_on_root(adypt_ctx *ctx, Comm *k)
{
	const CtxInfo i = ctx_info(ctx);
	HIP_OK(ctx, hipSetDevice(i.device));
	if(i.n_local_px > 0)
		HIP_OK(ctx, hipMemcpyAsync(k->gathered, i.accum, (size_t)i.n_local_px * sizeof(float4), hipMemcpyDeviceToDevice, i.stream));
	// pixels no rank owns do not exist (every block has an owner); the image is fully overwritten
	int r = adypt_assemble_radiance(ctx, k->gathered, k->stride, k->rgb); // enqueues k_untile per rank and drains the stream
	return r;
}

}  // namespace

// -----------------------------------------------------------------------------------------------------------------------
// one process, N devices
// -----------------------------------------------------------------------------------------------------------------------
struct adypt_multi {
	std::vector<adypt_ctx *> ctx;
	std::vector<int> devices;
	std::string error;
	bool comms_ready = false;
	bool shared_device = false; // test hook (ADYPT_MULTI_SHARED_DEVICE=1): several shards on ONE device; the exchange is a device copy
	int spp = 0;
	std::vector<double> setup_s; // seconds adypt_create took per device (they run concurrently)
	Tunables tun;
};

namespace {

int mfail(adypt_multi *m, int code, const std::string &msg) { m->error = msg; return code; }
int mfail_ctx(adypt_multi *m, int code, adypt_ctx *c) { m->error = adypt_last_error(c); return code; }

int multi_comm_init(adypt_multi *m)
{
	if(m->comms_ready) return ADYPT_OK;
	std::string err;
	RcclApi *api = rccl(&err);
	if(!api) return mfail(m, ADYPT_E_HIP, err);
	const int n = (int)m->ctx.size();
	std::vector<ncclComm_t> comms((size_t)n, nullptr);
	ncclResult_t r = api->CommInitAll(comms.data(), n, m->devices.data()); // rank i lives on devices[i]
	if(r != ncclSuccess) return mfail(m, ADYPT_E_HIP, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
	// every communicator is parked in its context FIRST (the context then owns and frees it, whatever happens next): a failure of
	// finish_comm for rank i must not leave the communicators of ranks i+1.. behind, and a retry must not create a second set
	std::vector<Comm *> parked((size_t)n, nullptr);
	for(int i = 0; i < n; ++i)
	{
		Comm *k = new Comm();
		k->api = api; k->comm = comms[(size_t)i];
		park_comm(m->ctx[(size_t)i], k);
		parked[(size_t)i] = k;
	}
	for(int i = 0; i < n; ++i)
	{
		int rr = finish_comm(m->ctx[(size_t)i], parked[(size_t)i]);
		if(rr != ADYPT_OK) return mfail_ctx(m, rr, m->ctx[(size_t)i]); // a later call starts over: park_comm frees what is parked now
	}
	m->comms_ready = true;
	return ADYPT_OK;
}

}  // namespace

extern "C" {

const char *adypt_multi_last_error(const adypt_multi *m) { return m ? m->error.c_str() : g_multi_error.c_str(); }

int adypt_create_multi(adypt_multi **out, const adypt_scene_desc *desc, const int *device_ids, int n_dev)
{
	g_multi_error.clear();
	if(!out || !desc || !device_ids || n_dev < 1 || n_dev > 64) { g_multi_error = "adypt_create_multi: bad arguments"; return ADYPT_E_INVALID; }
	*out = nullptr;
	// RCCL refuses two ranks on one device.  ADYPT_MULTI_SHARED_DEVICE=1 (after adypt_enable_test_hooks) is a test hook for boxes with a single GPU: the same fan-out,
	// sharding, ordering and un-tiling, with the peer -> root transfers done by device-to-device copies instead of ncclSend / ncclRecv.
	const Tunables tun = read_tunables();
	const bool shared = tun.multi_shared_device; // (false unless adypt_enable_test_hooks was called)
	for(int i = 0; i < n_dev; ++i)
		for(int j = 0; j < i; ++j)
			if(device_ids[i] == device_ids[j] && !shared) { g_multi_error = "adypt_create_multi: device listed twice (RCCL needs distinct devices)"; return ADYPT_E_INVALID; }
	adypt_multi *m = new adypt_multi();
	m->shared_device = shared;
	m->tun = tun;
	// One host thread per device: each context is an upload of the whole scene (80 MB .. 1.9 GB) plus its own allocations, and the devices
	// do not share a PCIe link — created one after the other, an 8-GPU start-up took 8 x the 1-GPU time.  (Several shards on ONE device, the
	// test hook, are created in turn: they would only queue up behind each other on the device's legacy stream.)
	std::vector<adypt_ctx *> made((size_t)n_dev, nullptr);
	std::vector<int> rc((size_t)n_dev, ADYPT_OK);
	std::vector<std::string> why((size_t)n_dev);
	m->setup_s.assign((size_t)n_dev, 0.0);
	// the Woop matrices (OglScene::init_triangles) are the same for every device: computed once here, not once per context
	std::vector<float> woop_once;
	adypt_scene_desc shared_desc = *desc;
	if(!shared_desc.woop && n_dev > 1 && shared_desc.n_refs > 0 && shared_desc.triangles && shared_desc.tri_indices)
	{
		woop_once.resize((size_t)shared_desc.n_refs * 12);
		adypt_woop_matrices(shared_desc.triangles, shared_desc.tri_indices, shared_desc.n_refs, woop_once.data());
		shared_desc.woop = woop_once.data();
	}
	auto create_one = [&](int i) {
		const auto t0 = std::chrono::steady_clock::now();
		adypt_scene_desc d = shared_desc;
		d.device = device_ids[i]; d.tile_rank = i; d.tile_nranks = n_dev;
		rc[(size_t)i] = adypt_create(&made[(size_t)i], &d);
		if(rc[(size_t)i] != ADYPT_OK) why[(size_t)i] = adypt_last_error(nullptr); // (thread-local: read on the thread that failed)
		m->setup_s[(size_t)i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
	};
	if(shared || n_dev == 1) for(int i = 0; i < n_dev; ++i) create_one(i);
	else
	{
		std::vector<std::thread> workers;
		std::vector<char> started((size_t)n_dev, 0);
		for(int i = 0; i < n_dev; ++i)
		{
			try { workers.emplace_back(create_one, i); started[(size_t)i] = 1; }
			catch(const std::exception &) {} // (no thread to be had: that device's context is created on this thread below)
		}
		for(int i = 0; i < n_dev; ++i) if(!started[(size_t)i]) create_one(i);
		for(std::thread &t : workers) t.join();
	}
	for(int i = 0; i < n_dev; ++i)
		if(rc[(size_t)i] != ADYPT_OK)
		{
			g_multi_error = std::string("device ") + std::to_string(device_ids[i]) + ": " + why[(size_t)i];
			for(adypt_ctx *c : made) adypt_destroy(c); // (null-safe)
			delete m;
			return rc[(size_t)i];
		}
	for(int i = 0; i < n_dev; ++i) { m->ctx.push_back(made[(size_t)i]); m->devices.push_back(device_ids[i]); }
	*out = m;
	return ADYPT_OK;
}

void adypt_destroy_multi(adypt_multi *m)
{
	if(!m) return;
	for(adypt_ctx *c : m->ctx) adypt_destroy(c); // frees the communicators as well (adypt_ctx::comm_free)
	delete m;
}

int adypt_multi_device_count(const adypt_multi *m) { return m ? (int)m->ctx.size() : ADYPT_E_INVALID; }
double adypt_multi_setup_seconds(const adypt_multi *m, int i) { return (m && i >= 0 && i < (int)m->setup_s.size()) ? m->setup_s[(size_t)i] : -1.0; }
adypt_ctx *adypt_multi_context(adypt_multi *m, int i) { return (m && i >= 0 && i < (int)m->ctx.size()) ? m->ctx[(size_t)i] : nullptr; }

#define FOR_ALL(m, call)                                                \
	do {                                                                \
		if(!(m)) return ADYPT_E_INVALID;                                \
		for(adypt_ctx *c : (m)->ctx) { const int r_ = (call); if(r_ != ADYPT_OK) return mfail_ctx((m), r_, c); } \
	} while(0)

int adypt_multi_set_params(adypt_multi *m, const adypt_pt_params *p) { FOR_ALL(m, adypt_set_params(c, p)); return ADYPT_OK; }
int adypt_multi_set_camera(adypt_multi *m, const float o[3], const float ip[16], const float iv[16]) { FOR_ALL(m, adypt_set_camera(c, o, ip, iv)); return ADYPT_OK; }
int adypt_multi_set_lookahead(adypt_multi *m, int enabled) { FOR_ALL(m, adypt_set_lookahead(c, enabled)); return ADYPT_OK; }
int adypt_multi_reset(adypt_multi *m) { FOR_ALL(m, adypt_reset(c)); return ADYPT_OK; }
int adypt_multi_set_sun_visibility(adypt_multi *m, int enabled, const float dir[3]) { FOR_ALL(m, adypt_set_sun_visibility(c, enabled, dir)); return ADYPT_OK; }
int adypt_multi_set_instrumentation(adypt_multi *m, int flags) { FOR_ALL(m, adypt_set_instrumentation(c, flags)); return ADYPT_OK; }
// every context writes the pixels of its own tiles into the caller's W*H*4 image: together they are the whole window
int adypt_multi_read_display(adypt_multi *m, uint8_t *rgba8) { if(!rgba8) return ADYPT_E_INVALID; FOR_ALL(m, adypt_read_display(c, rgba8)); return ADYPT_OK; }
int adypt_multi_get_stats(adypt_multi *m, adypt_stats *out)
{
	if(!m || !out) return ADYPT_E_INVALID;
	memset(out, 0, sizeof(*out));
	for(adypt_ctx *c : m->ctx)
	{
		adypt_stats s;
		const int r = adypt_get_stats(c, &s);
		if(r != ADYPT_OK) return mfail_ctx(m, r, c);
		// counts add up over the devices; the devices run concurrently, so times are the slowest device's
		out->rays += s.rays; out->nodes_visited += s.nodes_visited; out->tris_tested += s.tris_tested; out->hits += s.hits; out->shaded += s.shaded;
		out->stack_overflows += s.stack_overflows; out->bad_materials += s.bad_materials; out->audit_errors += s.audit_errors;
		out->path_rays += s.path_rays; out->path_nodes += s.path_nodes; out->path_tris += s.path_tris; out->path_hits += s.path_hits; out->path_shaded += s.path_shaded;
		out->max_stack = std::max(out->max_stack, s.max_stack); out->trace_launches = std::max(out->trace_launches, s.trace_launches);
		out->trace_ms = std::max(out->trace_ms, s.trace_ms); out->shade_ms = std::max(out->shade_ms, s.shade_ms);
		out->path_ms = std::max(out->path_ms, s.path_ms); out->path_launches = std::max(out->path_launches, s.path_launches);
	}
	return ADYPT_OK;
}
int adypt_multi_get_spp(const adypt_multi *m) { return (m && !m->ctx.empty()) ? adypt_get_spp(m->ctx[0]) : ADYPT_E_INVALID; }

int adypt_multi_trace_primary(adypt_multi *m, int viewer_type)
{
	// the primary viewer frame is one short pass per device; the calls are synchronous per device (adypt_trace_primary)
	FOR_ALL(m, adypt_trace_primary(c, viewer_type));
	return ADYPT_OK;
}

int adypt_multi_trace_spp(adypt_multi *m, int n_spp)
{
	FOR_ALL(m, adypt_trace_spp_async(c, n_spp)); // every device has its frames enqueued before the first one is waited for
	FOR_ALL(m, adypt_wait(c));
	return ADYPT_OK;
}

int adypt_multi_comm_init(adypt_multi *m)
{
	if(!m) return ADYPT_E_INVALID;
	if(m->shared_device) return mfail(m, ADYPT_E_STATE, "adypt_multi_comm_init: no RCCL communicator in the shared-device test mode (RCCL needs distinct devices)");
	return multi_comm_init(m);
}

// ranks as the communicator itself reports them (what a scaling run prints next to n_gpus)
static int comm_ranks_of(adypt_ctx *ctx)
{
	Comm *k = comm_of(ctx);
	if(!k || !k->comm) return 0;
	int n = 0;
	if(k->api && k->api->CommCount && k->api->CommCount(k->comm, &n) == ncclSuccess) return n;
	return k->nranks; // (a transport without ncclCommCount: the size the communicator was created with)
}
int adypt_multi_comm_ranks(adypt_multi *m) { return (m && !m->ctx.empty()) ? comm_ranks_of(m->ctx[0]) : ADYPT_E_INVALID; }
int adypt_comm_ranks(adypt_ctx *ctx) { return ctx ? comm_ranks_of(ctx) : ADYPT_E_INVALID; }

int adypt_multi_gather_radiance(adypt_multi *m, void **rgb_device)
{
	if(!m || !rgb_device) return ADYPT_E_INVALID;
	*rgb_device = nullptr;
	const int n = (int)m->ctx.size();
	adypt_ctx *root = m->ctx[0];
	if((n == 1 || m->shared_device) && !m->comms_ready)
	{
		// a single device has nothing to exchange (and the shared-device test hook exchanges by device copies): root-side buffers
		// only, no communicator
		if(!comm_of(root))
		{
			Comm *k = new Comm();
			park_comm(root, k);
			int r = finish_comm(root, k);
			if(r != ADYPT_OK) return mfail_ctx(m, r, root);
		}
	}
	else
	{
		int r = multi_comm_init(m);
		if(r != ADYPT_OK) return r;
	}
	Comm *k0 = comm_of(root);
	GatherWatchdog wd(n > 1 ? m->tun.gather_timeout_s : 0.0, [m] {
		std::string s;
		for(size_t i = 0; i < m->ctx.size(); ++i)
		{
			const CtxInfo ci = ctx_info(m->ctx[i]);
			s += "[adypt]   rank " + std::to_string(i) + " (device " + std::to_string(ci.device) + ", " + std::to_string(ci.n_local_px) + " local pixels): stream " + stream_state(ci.device, ci.stream) + "\n";
		}
		return s;
	});
	if(n > 1) stall_if_asked(m->tun, wd);
	if(n > 1 && m->shared_device)
	{
		for(int r = 1; r < n; ++r)
		{
			const CtxInfo pi = ctx_info(m->ctx[(size_t)r]);
			if(pi.n_local_px == 0) continue;
			int w = adypt_wait(m->ctx[(size_t)r]); // the peer's frames are done (its stream is not the root's)
			if(w != ADYPT_OK) return mfail_ctx(m, w, m->ctx[(size_t)r]);
			const CtxInfo ri = ctx_info(root);
			if(hipMemcpyAsync(k0->gathered + (size_t)r * (size_t)k0->stride, pi.accum, (size_t)pi.n_local_px * sizeof(float4), hipMemcpyDeviceToDevice, ri.stream) != hipSuccess)
				return mfail(m, ADYPT_E_HIP, "shared-device gather: copy failed");
		}
	}
	else if(n > 1)
	{
		// the one exchange: grouped point-to-point = ncclGather with exact per-rank sizes; peer r -> root over its own link
		RcclApi *api = k0->api;
		wd.stage("ncclGroupStart .. ncclGroupEnd (grouped ncclSend / ncclRecv)");
		ncclResult_t gr = api->GroupStart();
		if(gr != ncclSuccess) return mfail(m, ADYPT_E_HIP, std::string("ncclGroupStart: ") + api->GetErrorString(gr));
		ncclResult_t bad = ncclSuccess;
		for(int r = 1; r < n && bad == ncclSuccess; ++r)
		{
			const CtxInfo pi = ctx_info(m->ctx[(size_t)r]);
			if(pi.n_local_px == 0) continue;
			Comm *kr = comm_of(m->ctx[(size_t)r]);
			(void)hipSetDevice(pi.device);
			bad = api->Send(pi.accum, (size_t)pi.n_local_px * 4, ncclFloat, 0, kr->comm, pi.stream);
			if(bad != ncclSuccess) break;
			const CtxInfo ri = ctx_info(root);
			(void)hipSetDevice(ri.device);
			bad = api->Recv(k0->gathered + (size_t)r * (size_t)k0->stride, (size_t)pi.n_local_px * 4, ncclFloat, r, k0->comm, ri.stream);
		}
		gr = api->GroupEnd();
		if(bad != ncclSuccess || gr != ncclSuccess)
			return mfail(m, ADYPT_E_HIP, std::string("RCCL gather: ") + api->GetErrorString(bad != ncclSuccess ? bad : gr));
	}
	wd.stage("un-tiling on the root (drains the root's stream: the receives)");
	int r = assemble_on_root(root, k0);
	if(r != ADYPT_OK) return mfail_ctx(m, r, root);
	wd.stage("draining the peers' streams (their sends)");
	// the peers' sends complete with the root's receives; drain their streams so their images may be overwritten again
	for(int i = 1; i < n; ++i) { int w = adypt_wait(m->ctx[(size_t)i]); if(w != ADYPT_OK) return mfail_ctx(m, w, m->ctx[(size_t)i]); }
	*rgb_device = k0->rgb;
	return ADYPT_OK;
}

int adypt_multi_read_radiance(adypt_multi *m, float *rgb)
{
	if(!m || !rgb) return ADYPT_E_INVALID;
	void *dev = nullptr;
	int r = adypt_multi_gather_radiance(m, &dev);
	if(r != ADYPT_OK) return r;
	const CtxInfo i = ctx_info(m->ctx[0]);
	if(hipSetDevice(i.device) != hipSuccess || hipMemcpy(rgb, dev, (size_t)i.width * i.height * 3 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
		return mfail(m, ADYPT_E_HIP, "adypt_multi_read_radiance: device-to-host copy failed");
	return ADYPT_OK;
}

// -----------------------------------------------------------------------------------------------------------------------
// one process per GPU
// -----------------------------------------------------------------------------------------------------------------------
int adypt_comm_unique_id(char id[ADYPT_COMM_ID_BYTES])
{
	static_assert(ADYPT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
	if(!id) return ADYPT_E_INVALID;
	std::string err;
	RcclApi *api = rccl(&err);
	if(!api) { g_multi_error = err; return ADYPT_E_HIP; }
	ncclUniqueId u;
	ncclResult_t r = api->GetUniqueId(&u);
	if(r != ncclSuccess) { g_multi_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return ADYPT_E_HIP; }
	memcpy(id, u.internal, ADYPT_COMM_ID_BYTES);
	return ADYPT_OK;
}

int adypt_comm_init(adypt_ctx *ctx, const char id[ADYPT_COMM_ID_BYTES])
{
	if(!ctx || !id) return ADYPT_E_INVALID;
	std::string err;
	RcclApi *api = rccl(&err);
	if(!api) { ctx_set_error(ctx, err); return ADYPT_E_HIP; }
	const CtxInfo i = ctx_info(ctx);
	HIP_OK(ctx, hipSetDevice(i.device));
	ncclUniqueId u;
	memcpy(u.internal, id, ADYPT_COMM_ID_BYTES);
	Comm *k = new Comm();
	k->api = api;
	ncclResult_t r = api->CommInitRank(&k->comm, i.nranks, u, i.rank);
	if(r != ncclSuccess) { delete k; ctx_set_error(ctx, std::string("ncclCommInitRank: ") + api->GetErrorString(r)); return ADYPT_E_HIP; }
	park_comm(ctx, k);
	return finish_comm(ctx, k);
}

int adypt_comm_gather_radiance(adypt_ctx *ctx, void **rgb_device)
{
	if(!ctx || !rgb_device) return ADYPT_E_INVALID;
	*rgb_device = nullptr;
	Comm *k = comm_of(ctx);
	const CtxInfo i = ctx_info(ctx);
	if(!k)
	{
		if(i.nranks != 1) { ctx_set_error(ctx, "adypt_comm_gather_radiance: call adypt_comm_init first"); return ADYPT_E_STATE; }
		k = new Comm();
		park_comm(ctx, k);
		int r = finish_comm(ctx, k);
		if(r != ADYPT_OK) return r;
	}
	HIP_OK(ctx, hipSetDevice(i.device));
	const Tunables &tun = k->tun;
	// What the watchdog times is the collective and its drain — not the frames a caller has queued in front of it with adypt_trace_spp_async, nor the skew
	// between ranks that are still rendering: this rank's own stream is drained first (rendering cannot hang on another rank: it has no collective).
	if(i.nranks > 1) { int r = adypt_wait(ctx); if(r != ADYPT_OK) return r; }
	GatherWatchdog wd(i.nranks > 1 ? tun.gather_timeout_s : 0.0, [i] {
		return "[adypt]   rank " + std::to_string(i.rank) + " of " + std::to_string(i.nranks) + " (device " + std::to_string(i.device) + ", " + std::to_string(i.n_local_px) + " local pixels): stream " +
			   stream_state(i.device, i.stream) + "\n";
	});
	if(i.nranks > 1) stall_if_asked(tun, wd);
	if(i.nranks > 1)
	{
		RcclApi *api = k->api;
		wd.stage("ncclGroupStart .. ncclGroupEnd (grouped ncclSend / ncclRecv)");
		NCCL_OK(ctx, api, api->GroupStart());
		ncclResult_t bad = ncclSuccess;
		if(i.rank == 0)
		{
			for(int r = 1; r < i.nranks && bad == ncclSuccess; ++r)
				if(k->counts[(size_t)r] > 0)
					bad = api->Recv(k->gathered + (size_t)r * (size_t)k->stride, (size_t)k->counts[(size_t)r] * 4, ncclFloat, r, k->comm, i.stream);
		}
		else if(i.n_local_px > 0) bad = api->Send(i.accum, (size_t)i.n_local_px * 4, ncclFloat, 0, k->comm, i.stream);
		ncclResult_t ge = api->GroupEnd();
		if(bad != ncclSuccess || ge != ncclSuccess)
		{
			ctx_set_error(ctx, std::string("RCCL gather: ") + api->GetErrorString(bad != ncclSuccess ? bad : ge));
			return ADYPT_E_HIP;
		}
	}
	if(i.rank == 0)
	{
		wd.stage("un-tiling on the root (drains the root's stream: the receives)");
		int r = assemble_on_root(ctx, k);
		if(r != ADYPT_OK) return r;
		*rgb_device = k->rgb;
		return ADYPT_OK;
	}
	wd.stage("draining this rank's stream (its send)");
	return adypt_wait(ctx); // the send has left the accumulation image
}

int adypt_comm_read_radiance(adypt_ctx *ctx, float *rgb)
{
	if(!ctx) return ADYPT_E_INVALID;
	void *dev = nullptr;
	int r = adypt_comm_gather_radiance(ctx, &dev);
	if(r != ADYPT_OK || !dev) return r;
	if(!rgb) return ADYPT_E_INVALID;
	const CtxInfo i = ctx_info(ctx);
	HIP_OK(ctx, hipMemcpy(rgb, dev, (size_t)i.width * i.height * 3 * sizeof(float), hipMemcpyDeviceToHost));
	return ADYPT_OK;
}

int adypt_comm_allreduce(adypt_ctx *ctx, double *values, int n, int op)
{
	if(!ctx || !values || n < 1 || n > Comm::kScratch || (op != 0 && op != 1)) return ADYPT_E_INVALID;
	const CtxInfo i = ctx_info(ctx);
	if(i.nranks == 1) return ADYPT_OK;
	Comm *k = comm_of(ctx);
	if(!k || !k->comm) { ctx_set_error(ctx, "adypt_comm_allreduce: call adypt_comm_init first"); return ADYPT_E_STATE; }
	HIP_OK(ctx, hipSetDevice(i.device));
	HIP_OK(ctx, hipMemcpyAsync(k->scratch, values, (size_t)n * sizeof(double), hipMemcpyHostToDevice, i.stream));
	NCCL_OK(ctx, k->api, k->api->AllReduce(k->scratch, k->scratch, (size_t)n, ncclDouble, op == 0 ? ncclSum : ncclMax, k->comm, i.stream));
	HIP_OK(ctx, hipMemcpyAsync(values, k->scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, i.stream));
	HIP_OK(ctx, hipStreamSynchronize(i.stream));
	return ADYPT_OK;
}

int adypt_comm_barrier(adypt_ctx *ctx)
{
	double one = 1.0;
	int r = adypt_comm_allreduce(ctx, &one, 1, 0);
	return r != ADYPT_OK ? r : adypt_device_synchronize(ctx);
}

int adypt_device_synchronize(adypt_ctx *ctx)
{
	if(!ctx) return ADYPT_E_INVALID;
	const CtxInfo i = ctx_info(ctx);
	HIP_OK(ctx, hipSetDevice(i.device));
	HIP_OK(ctx, hipDeviceSynchronize());
	return ADYPT_OK;
}

}  // extern "C"
