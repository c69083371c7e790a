/*
 * adypt_hip.h — C-ABI of the MI355X-native tracer that replaces Adypt's src/Tracer (OglScene + OglPathTracer)
 * and the three GLSL compute programs (shaders/{traversal,primaryray,pathtracer}.glsl).
 *
 * The reference has no FFI/plugin layer: the seam is the two C++ classes OglScene and OglPathTracer that
 * Instance (src/Instance.cpp:33-35,44-57) and Application (src/Application.cpp:108-109,220-234,275) call.
 * Every entry point below names the reference interface it stands in for.  Plain pointers and sizes only;
 * the caller owns all host arrays, the callee copies them to HBM at create time (mirrors the immutable
 * glNamedBufferStorage(..., flags = 0) uploads of src/Tracer/OglScene.cpp:125-134).
 *
 * Threading: one host thread drives a context; calls are synchronous unless stated; a context is bound to one
 * HIP device and owns its HIP streams.  No exceptions cross this boundary: every call returns ADYPT_OK (0) or a
 * negative ADYPT_E_* code and adypt_last_error() gives the text (the reference printf()s and returns bool).
 */
#ifndef ADYPT_HIP_H
#define ADYPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADYPT_ABI_VERSION 4

enum adypt_status {
	ADYPT_OK = 0,
	ADYPT_E_INVALID = -1,        /* bad argument / inconsistent array sizes */
	ADYPT_E_NO_DEVICE = -2,      /* no HIP device (the product has no CPU fallback) */
	ADYPT_E_HIP = -3,            /* HIP runtime error, text in adypt_last_error */
	ADYPT_E_OOM = -4,
	ADYPT_E_STACK_OVERFLOW = -5, /* traversal needed more than stackSize entries (reference: undefined behaviour, traversal.glsl:60) */
	ADYPT_E_BAD_MATERIAL = -6,   /* a hit triangle has material id outside [0, n_mats) (reference: uMaterials[-1], pathtracer.glsl:77) */
	ADYPT_E_IO = -7,
	ADYPT_E_PARSE = -8,
	ADYPT_E_STATE = -9           /* call order violated (e.g. trace before set_camera) */
};

typedef struct adypt_ctx adypt_ctx;

/* One decoded texture: RGB8, rows top to bottom exactly as stbi_load(..., 3) returns them
 * (src/Tracer/OglScene.cpp:26-34); sampled GL_LINEAR / GL_REPEAT / single level (:33-38). */
typedef struct adypt_texture {
	int32_t width, height;
	const uint8_t *rgb;
} adypt_texture;

/* Inputs of OglScene::Initialize(const Scene&, const WideBVH&) (src/Tracer/OglScene.hpp:43) + the image size
 * OglPathTracer::Initialize receives (src/Tracer/OglPathTracer.cpp:11-16), as flat arrays. */
typedef struct adypt_scene_desc {
	const void *nodes;           /* WideBVH::GetNodes(): 80-byte WideBVHNode (src/BVH/WideBVH.hpp:13-26)          SSBO 0 */
	int64_t n_nodes;
	const int32_t *tri_indices;  /* WideBVH::GetTriIndices(): reference -> scene triangle                          SSBO 1 */
	int64_t n_refs;
	const float *woop;           /* 12 floats per reference in tri_indices order, or NULL: computed on the host
	                                exactly as OglScene::init_triangles does (src/Tracer/OglScene.cpp:93-116)     SSBO 2 */
	const void *triangles;       /* Scene::GetTriangles(): 100-byte Triangle (src/Util/Shape.hpp:70-74)            SSBO 3 */
	int64_t n_tris;
	const void *materials;       /* 64-byte GPUMaterial (src/Tracer/OglScene.hpp:19-28)                            SSBO 4 */
	int64_t n_mats;
	const adypt_texture *textures; /* bindless sampler table (OglScene.cpp:84-90)                                   UBO 3 */
	int32_t n_textures;
	int32_t width, height;       /* IMG_SIZE */
	int32_t device;              /* HIP device ordinal */
	int32_t tile_rank;           /* pixel-tile shard: this context renders the 32x32 blocks whose owner is tile_rank */
	int32_t tile_nranks;         /* 1 = whole image */
} adypt_scene_desc;

/* InstanceConfig::PT (src/InstanceConfig.hpp:21-27) as consumed by update_config_args (OglPathTracer.cpp:214-225),
 * plus the explicit seed that replaces std::random_device in create_buffers (OglPathTracer.cpp:156-162). */
typedef struct adypt_pt_params {
	int32_t stack_size, max_bounce, subpixel, tmp_lifetime;
	float ray_tmin, clamp, sun[3];
	uint32_t shift_seed;
} adypt_pt_params;

/* One closest-hit record (outputs of BVHIntersection, traversal.glsl:14,253-254, + instrumentation). */
typedef struct adypt_hit {
	int32_t ref_idx;   /* index into tri_indices / woop (pre-remap), -1 = miss */
	int32_t tri_id;    /* uTriIndices[ref_idx], -1 = miss */
	float u, v, t;
	uint32_t nodes, tris, hash, max_depth; /* filled only by the instrumented kernel variant (with_stats) */
} adypt_hit;

typedef struct adypt_stats {
	uint64_t rays;            /* BVHIntersection invocations (cached primaries are not rays) */
	uint64_t nodes_visited;   /* instrumented runs only */
	uint64_t tris_tested;     /* instrumented runs only */
	uint64_t hits;            /* instrumented runs only */
	uint64_t shaded;          /* FetchInfo executions */
	uint64_t stack_overflows;
	uint64_t bad_materials;
	uint32_t max_stack;       /* instrumented runs only */
	uint32_t trace_launches;
	double trace_ms;          /* sum of HIP-event durations of the traversal kernel launches (timing enabled): k_trace and k_path */
	double shade_ms;          /* same for gen/shade kernels */
	double path_ms;           /* the part of trace_ms that is k_path launches (one launch = every bounce after the first of a batch) */
	uint32_t path_launches;   /* the part of trace_launches that is k_path launches */
	uint32_t audit_errors;    /* slot-claim audit (adypt_set_instrumentation flag 4): queue slots not written exactly once; must be 0 */
	uint64_t path_rays;       /* the part of rays / nodes_visited / tris_tested / hits / shaded that k_path launches account for */
	uint64_t path_nodes, path_tris, path_hits, path_shaded; /* (instrumented runs only, like their totals) */
} adypt_stats;

int adypt_abi_version(void);

/* TEST-ONLY.  The library has a few hooks that exist so that the multi-GPU paths can be exercised on a machine with ONE GPU (several tile shards
 * on one device, the RCCL call table served by a shared-memory transport) and so that its own detectors can be tested (a planted queue corruption,
 * a stalled gather).  They are selected through the environment (csrc/device/tunables.hpp lists them) but are IGNORED unless this function has been
 * called in the process with ADYPT_TEST_HOOKS_MAGIC: the environment alone cannot change what the shipped library does.  tests/ call it;
 * bench.py only under --rehearsal.  Returns ADYPT_E_INVALID for any other value.  There is no way to switch the hooks off again. */
#define ADYPT_TEST_HOOKS_MAGIC 0x7465737468303031ull /* "testh001" */
int adypt_enable_test_hooks(uint64_t magic);
int adypt_test_hooks_enabled(void);

/* OglScene::Initialize + OglPathTracer::Initialize (create_buffers / bind_buffers).  Fails with ADYPT_E_NO_DEVICE
 * when no GPU is present. */
int adypt_create(adypt_ctx **out, const adypt_scene_desc *desc);
void adypt_destroy(adypt_ctx *ctx);
const char *adypt_last_error(const adypt_ctx *ctx); /* ctx may be NULL: error of the last failed adypt_create */

/* OglPathTracer::update_config_args (OglPathTracer.cpp:214-225); also (re)generates the per-pixel Sobol shift
 * image from shift_seed.  Like the reference, new values take effect for path tracing at the next adypt_reset(). */
int adypt_set_params(adypt_ctx *ctx, const adypt_pt_params *params);
/* OglPathTracer::SetCamera (OglPathTracer.cpp:27-32) with the two inverses already taken (column-major, as glm). */
int adypt_set_camera(adypt_ctx *ctx, const float origin[3], const float inv_proj[16], const float inv_view[16]);

/* OglPathTracer::Trace(false): one primary-ray viewer frame (primaryray.glsl), viewer_type = ViewerTypes
 * (OglPathTracer.hpp:20: 0 diffuse, 1 specular, 2 emissive, 4 normal, 5 position).  Resets the spp counter. */
int adypt_trace_primary(adypt_ctx *ctx, int viewer_type);
/* OglPathTracer::Trace(true), n_spp times: per frame Sobol::Next, spp++, wavefront passes.  The first call after
 * adypt_reset()/adypt_trace_primary() clears the result image and restarts the Sobol sequence (OglPathTracer.cpp:39-46). */
int adypt_trace_spp(adypt_ctx *ctx, int n_spp);
/* The same, returning as soon as the frames are enqueued on the context's stream (one host thread can then keep several
 * contexts = GPUs busy); adypt_wait blocks until the context is idle and reports what adypt_trace_spp would have
 * (stack overflow, HIP errors).  Every other entry point that reads results synchronises by itself. */
int adypt_trace_spp_async(adypt_ctx *ctx, int n_spp);
int adypt_wait(adypt_ctx *ctx);
/* SURVEY.md §8 f1, off by default (= the reference as it runs): enables the occlusion query the reference has commented
 * out in Render (shaders/pathtracer.glsl:132, `if(!BVHIntersection(origin, normalize(vec3(0.6, 1, 0.2))))`): a path that
 * leaves the scene receives the sun term only if an any-hit ray (traversal.glsl:257-494) from its last position towards
 * `dir` (NULL = the reference's (0.6, 1, 0.2); normalised here) finds nothing.  Takes effect for the frames traced next.
 * The query is traced INSIDE the one-launch pipeline, like the stubbed call inside the reference's single dispatch: the escaped path keeps its slot in k_path and is
 * traced once more as a ray that ends at its first accepted triangle (measured: 0.95 of the rays/s with the option off; the queries are rays of the census). */
int adypt_set_sun_visibility(adypt_ctx *ctx, int enabled, const float dir[3]);
int adypt_reset(adypt_ctx *ctx);
int adypt_get_spp(const adypt_ctx *ctx); /* OglPathTracer::GetSPP */
/* How many consecutive frames adypt_trace_spp traces as one wavefront pass (1..128).  Frames are independent
 * samples and the running mean is applied afterwards in frame order, so results are bit-identical for every value;
 * more frames in flight keep small images / tile shards saturated.  Default: chosen from the local pixel count
 * (about 64 Mi paths per pass: 32 frames at 1920x1080, 128 for the tile shard of an 8-GPU run).  Reallocates the ray
 * queues; when that fails the previous value is kept (and the error returned), and if nothing can be allocated any more
 * the trace calls return ADYPT_E_STATE. */
int adypt_set_frames_in_flight(adypt_ctx *ctx, int n_frames);
int adypt_get_frames_in_flight(const adypt_ctx *ctx);
/* How many sub-batches a batch of frames is cut into (1..4, default 1; ADYPT_PIPELINE in the environment overrides the
 * default).  Each sub-batch is the chain camera rays -> [traversal -> shade] x maxBounce on its own HIP stream over its own
 * window of the ray queues, so one chain's traversal launch covers the drain of the other's and the time the other's shade
 * kernel spends streaming the queues through HBM — the wavefront counterpart of the reference's single dispatch that runs the
 * whole bounce loop without a barrier (shaders/pathtracer.glsl:107, src/Tracer/OglPathTracer.cpp:60).  1 = one chain on the
 * context's stream (kernel timings of adypt_get_stats are then of launches that had the GPU to themselves).  Results are
 * bit-identical for every value. */
int adypt_set_pipeline(adypt_ctx *ctx, int n_pipes);
int adypt_get_pipeline(const adypt_ctx *ctx);
/* Look-ahead for callers that ask for ONE frame per call, as Instance::Update does (src/Instance.cpp:44-57 -> Trace(true),
 * src/Tracer/OglPathTracer.cpp:34-61).  Off (default): adypt_trace_spp(ctx, n) traces exactly n frames.  On: a call that
 * needs frames not traced yet traces a whole pass of frames_in_flight frames — frame k's sample depends only on k (Sobol
 * point k, per-pixel shift, the sub-pixel offset and primary-hit cache of its tmpLifetime group), so the extra frames are
 * the very samples later calls would compute — applies the running mean for the n frames asked for, and parks the rest;
 * the following calls only apply one running-mean step per frame.  The image, adypt_get_spp and image 1 (adypt_read_hits)
 * after every call are bit-identical to frame-by-frame tracing; adypt_get_stats counts work when it is done, i.e. includes
 * frames traced ahead.  adypt_set_camera, adypt_reset, adypt_trace_primary, adypt_set_sun_visibility and
 * adypt_set_frames_in_flight drop the parked frames (they are traced again, with the new state, when asked for).
 * One frame per wavefront pass (adypt_set_frames_in_flight(ctx, 1), or any call that asks for a single frame with look-ahead off): consecutive single frames
 * run on two HIP streams — frame k + 1's bounce 0 and k_path are enqueued under the END of frame k's k_path launch (its last paths' sequential bounces, a
 * quarter of a 1080p frame's time) whenever the call itself asks for frame k + 1; with look-ahead ON also across calls: frame k + 1 is STARTED (not parked:
 * adypt_get_lookahead_frames stays 0) before the call for frame k returns, unless it would re-trace primary rays (image 1 stays what frame-by-frame tracing
 * leaves there).  A started frame is waited for and forgotten by the calls listed above and by adypt_trace_rays; images are bit-identical either way
 * (ADYPT_SINGLE_OVERLAP=0 in the environment: strictly one frame after the other).  Off above 2^22 local pixels, where a launch's end no longer matters. */
int adypt_set_lookahead(adypt_ctx *ctx, int enabled);
/* Bounces 1 .. maxBounce-1 of a batch of frames in ONE persistent launch (k_path: the reference's for(b < uMaxBounce) inside one
 * dispatch, shaders/pathtracer.glsl:107, src/Tracer/OglPathTracer.cpp:60) instead of a traversal and a shade launch per bounce.  On by
 * default (ADYPT_FUSED_BOUNCES=0 in the environment: off); used for every batch — a single frame included (a batch of one; ADYPT_SINGLE_FUSED=0
 * keeps the launch-per-bounce frame) — unless the sub-batch pipeline is on, the batch has more than 2^26 paths, or the sun-visibility query is on with 32 bounces
 * configured (the query travels as bounce index 31).
 * Images are bit-identical either way.  The getter says whether the LAST batch used it.
 * Memory: k_path looks a hit's triangle up by REFERENCE index in a second copy of the 128-byte triangle records (n_refs x 128 B: 43 MB for the
 * 249 k-triangle bench scene), allocated at adypt_create (1.3 GB for the 10.1 M references of the 10 M-triangle stand-in; measured +3 % there and +1 % on the bench scene against the remap, so no size limit
 * by default: ADYPT_REF_TRIANGLES_MAX_MB sets one); above the limit, or when the allocation fails, k_path goes through the 4-byte uTriIndices
 * remap instead — same image. */
int adypt_set_fused_bounces(adypt_ctx *ctx, int enabled);
int adypt_get_fused_bounces(const adypt_ctx *ctx);
int adypt_get_lookahead_frames(const adypt_ctx *ctx); /* frames currently parked */

/* glGetTextureImage(m_result_tex, GL_RGB, GL_FLOAT) of OglPathTracer::SaveResult (OglPathTracer.cpp:203-205):
 * W*H*3 floats, row 0 = top of the image.  Pixels of blocks this context does not own are left untouched. */
int adypt_read_radiance(adypt_ctx *ctx, float *rgb);
/* What OglPathTracer::DrawScreen shows (shaders/screen.glsl:15-21 over the result image, src/Tracer/OglPathTracer.cpp
 * DrawScreen): gamma 1/2.2 for viewer types <= 3 (diffuse, specular, emissive, path-traced radiance), normalize * 0.5 +
 * 0.5 for the normal / position viewers, as 8-bit RGBA (W*H*4 bytes, row 0 = top, alpha 255).  The viewer type is the
 * one of the frame in the result image (the last adypt_trace_primary's, 3 after adypt_trace_spp).  Pixels of blocks
 * this context does not own are left untouched. */
int adypt_read_display(adypt_ctx *ctx, uint8_t *rgba8);
/* content of image 1 (uPrimaryTmpImg, pathtracer.glsl:114-127): scene triangle id and uv of the cached primary hit */
int adypt_read_hits(adypt_ctx *ctx, int32_t *tri, float *uv);

/* Trace an arbitrary batch of rays through the same traversal kernel: rays = n x 8 floats
 * (ox, oy, oz, tmin, dx, dy, dz, unused).  with_stats selects the instrumented kernel variant. */
int adypt_trace_rays(adypt_ctx *ctx, const float *rays, int64_t n, adypt_hit *hits, int with_stats);
/* The any-hit overload `bool BVHIntersection(origin_tmin, dir)` (traversal.glsl:257-494; present in the reference but
 * never called by its shaders): same traversal, every ray stops at the FIRST accepted triangle in traversal order.
 * hits[i].tri_id != -1  <=>  the GLSL function returns true; u, v, t describe that first accepted triangle. */
int adypt_trace_rays_any(adypt_ctx *ctx, const float *rays, int64_t n, adypt_hit *hits, int with_stats);

/* instrumentation: bit 0 = per-launch HIP-event timing, bit 1 = instrumented traversal (node/triangle counts), bit 2 = slot-claim audit of the
 * ray queues (every queue a kernel appends to is poisoned before the launch and checked after it: each slot below the segment's counter written
 * exactly once, by a distinct path, nothing above it; adypt_stats::audit_errors counts violations — a debugging aid, several times slower) */
int adypt_set_instrumentation(adypt_ctx *ctx, int flags);
int adypt_get_stats(adypt_ctx *ctx, adypt_stats *out);
int adypt_reset_stats(adypt_ctx *ctx);
/* SIMD-occupancy profile of the instrumented traversal launches since the last reset (measurement only; wave-level
 * sums): out[0] loop trips, [1] lanes holding a ray summed over trips, [2] trips in which the wave ran its Woop test, [3] lanes testing a triangle
 * summed over those, [4] slab-test phases executed, [5] lanes active summed over those, [6] refill events,
 * [7] trips with no lane holding a ray. */
int adypt_get_wave_profile(adypt_ctx *ctx, uint64_t out[8]);
/* New (measurement): out[0] = shader-clock cycles (s_memtime), out[1] = ticks of the constant 100 MHz counter (s_memrealtime), both over the
 * lifetime of workgroup 0 of every traversal launch since adypt_reset_stats: out[0] / out[1] x 0.1 = the GHz the chip held under the
 * traversal kernel in THIS run (bench.py prices the vector-ALU roof with it). */
int adypt_get_shader_clock(adypt_ctx *ctx, uint64_t out[2]);

/* ---- pixel-tile sharding plumbing (multi-GPU: one context per GPU/process, one gather per output frame) ---- */
/* number of RGBA float4 elements in the compact local radiance buffer (owned blocks x 1024 pixels) */
int64_t adypt_local_pixel_count(const adypt_ctx *ctx);
/* device pointer of the compact local radiance buffer (float4 per local pixel, block-major) */
int adypt_local_radiance_device(adypt_ctx *ctx, void **dptr);
/* device-to-device copy of the compact local radiance into caller-owned device memory (e.g. a torch tensor that
 * a torch.distributed / RCCL gather then sends): copies local_pixel_count float4, zero-fills up to capacity_float4 */
int adypt_copy_local_radiance(adypt_ctx *ctx, void *dst_device, int64_t capacity_float4);
/* On the rank that received the gather: un-tile the compact buffers of all `tile_nranks` ranks (device memory, rank r at
 * gathered + r * stride_float4 float4) into one W*H*3 fp32 image in device memory (row 0 = top) — the assembled
 * equivalent of the reference's result texture (OglPathTracer.cpp:203-205 reads it back with glGetTextureImage). */
int adypt_assemble_radiance(adypt_ctx *ctx, const void *gathered_device, int64_t stride_float4, void *rgb_device);
/* blocks owned by `rank` out of `nranks` for a width x height image (same function the contexts use) */
int64_t adypt_shard_block_count(int width, int height, int rank, int nranks);
/* scatter one rank's compact buffer (host memory, block-major float4) into a W*H*3 host image */
int adypt_untile_host(int width, int height, int rank, int nranks, const float *local_rgba, float *rgb);

/* ---- native multi-GPU (SURVEY.md §8b "Inputs: device_ids[], n_dev" / §8e): RCCL inside the library -----------------------
 * The frame shards by 32x32 pixel tile with no data-path collective; the ONE exchange is the gather of the fp32 radiance
 * tiles on the root GPU per output frame — grouped ncclSend / ncclRecv with the exact per-rank sizes (the equivalent of
 * one ncclGather, rccl.h:745, without padding), each peer -> root over its own xGMI link, followed by the un-tiling kernel
 * on the root.  librccl is loaded on first use (dlopen), so the library itself has no link-time dependency on it.
 *
 * (1) One process, N devices — what a C++ host such as the reference's Instance (src/Instance.cpp:33-57) needs to use more
 * than one GPU: one adypt_multi stands for N contexts (tile rank i on device_ids[i], scene replicated), every call fans out
 * to all of them from the calling thread (the per-device work is asynchronous), adypt_multi_read_radiance gathers.
 * Test hook for single-GPU machines (only after adypt_enable_test_hooks): with ADYPT_MULTI_SHARED_DEVICE=1 in the environment a device may be
 * listed several times (RCCL refuses that); the shards then share the device and the peer -> root transfers are device-to-device copies. */
typedef struct adypt_multi adypt_multi;
int adypt_create_multi(adypt_multi **out, const adypt_scene_desc *desc /* device, tile_rank, tile_nranks ignored */,
                       const int *device_ids, int n_dev);
void adypt_destroy_multi(adypt_multi *m);
const char *adypt_multi_last_error(const adypt_multi *m); /* m may be NULL: error of the last failed adypt_create_multi */
int adypt_multi_device_count(const adypt_multi *m);
/* seconds adypt_create took for device_ids[i] (the N contexts are created concurrently, one host thread each); -1 for a bad index */
double adypt_multi_setup_seconds(const adypt_multi *m, int i);
adypt_ctx *adypt_multi_context(adypt_multi *m, int i);    /* the context of device_ids[i] (statistics, tunables); owned by m */
int adypt_multi_set_params(adypt_multi *m, const adypt_pt_params *params);
int adypt_multi_set_camera(adypt_multi *m, const float origin[3], const float inv_proj[16], const float inv_view[16]);
int adypt_multi_set_lookahead(adypt_multi *m, int enabled);
int adypt_multi_trace_primary(adypt_multi *m, int viewer_type);
int adypt_multi_trace_spp(adypt_multi *m, int n_spp);     /* all devices enqueue, then all are waited for */
int adypt_multi_reset(adypt_multi *m);
int adypt_multi_set_sun_visibility(adypt_multi *m, int enabled, const float direction[3]); /* adypt_set_sun_visibility on every device */
int adypt_multi_set_instrumentation(adypt_multi *m, int flags);
/* counts summed over the devices; trace_ms / shade_ms / trace_launches / max_stack of the slowest (largest) device — they run concurrently */
int adypt_multi_get_stats(adypt_multi *m, adypt_stats *out);
/* adypt_read_display for the whole window: every device converts and writes the pixels of its own tiles into rgba8 (W*H*4) */
int adypt_multi_read_display(adypt_multi *m, uint8_t *rgba8);
int adypt_multi_get_spp(const adypt_multi *m);
/* glGetTextureImage of OglPathTracer::SaveResult (OglPathTracer.cpp:203-205) for the whole image: the gather + un-tiling on
 * device_ids[0], then one device-to-host copy of W*H*3 floats (row 0 = top). */
int adypt_multi_read_radiance(adypt_multi *m, float *rgb);
/* the same, leaving the assembled W*H*3 fp32 image in the HBM of device_ids[0] (library-owned buffer, valid until the next
 * gather or adypt_destroy_multi) — resident like the reference's result texture */
/* Bounded: if the gather (this call and adypt_comm_gather_radiance alike) has not finished within ADYPT_GATHER_TIMEOUT seconds (environment;
 * default 120, 0 = unbounded) a watchdog thread prints where it stands and the state of every rank's stream to stderr and ENDS THE PROCESS with
 * exit code 86 — a collective whose peer never arrives cannot be cancelled, and a process that has touched the GPU must not be re-executed. */
int adypt_multi_gather_radiance(adypt_multi *m, void **rgb_device);
/* creates the RCCL communicators now (otherwise: at the first gather, and only when n_dev > 1); lets a caller — and the
 * one-GPU test — find out at start-up whether RCCL is usable */
int adypt_multi_comm_init(adypt_multi *m);
/* ranks of the communicator as RCCL reports them (ncclCommCount); 0 = no communicator exists (one device, or not initialised yet) */
int adypt_multi_comm_ranks(adypt_multi *m);

/* (2) One process per GPU (launchers that fork a rank per device): every rank creates its context with tile_rank = rank,
 * tile_nranks = world; rank 0 makes an id, the launcher's own channel (a file, MPI, a socket) carries its 128 bytes to the
 * other ranks, every rank calls adypt_comm_init.  The calls below are collective over the ranks of the shard. */
#define ADYPT_COMM_ID_BYTES 128
int adypt_comm_unique_id(char id[ADYPT_COMM_ID_BYTES]);
int adypt_comm_init(adypt_ctx *ctx, const char id[ADYPT_COMM_ID_BYTES]);
int adypt_comm_ranks(adypt_ctx *ctx); /* ncclCommCount of the context's communicator; 0 = none */
/* the one gather: on rank 0 *rgb_device = assembled W*H*3 fp32 image in HBM (library-owned), elsewhere NULL; returns after
 * the context's stream has drained */
int adypt_comm_gather_radiance(adypt_ctx *ctx, void **rgb_device);
/* gather + copy to host memory on rank 0 (rgb may be NULL on the other ranks) */
int adypt_comm_read_radiance(adypt_ctx *ctx, float *rgb);
/* control-plane helpers for launchers without a communication layer of their own (bench.py): in-place all-reduce of n
 * doubles (op 0 = sum, 1 = max), and a barrier (= all-reduce of one value + stream drain) */
int adypt_comm_allreduce(adypt_ctx *ctx, double *values, int n, int op);
int adypt_comm_barrier(adypt_ctx *ctx);
/* blocks until everything enqueued on the context's device has finished (hipDeviceSynchronize on its device) */
int adypt_device_synchronize(adypt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* ADYPT_HIP_H */
