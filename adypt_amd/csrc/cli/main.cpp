// adypt_hip — headless equivalent of the reference application for one .config file:
//   Instance::InitializeFromFile / Initialize (src/Instance.cpp:10-42,59-69) -> N x Trace(true) -> SaveResult
//   (src/Tracer/OglPathTracer.cpp:199-212).  The interactive window / ImGui front-end is out of scope.
//
//   adypt_hip scene.config [--spp N] [--out result.exr] [--fp16] [--primary TYPE] [--preview file.png] [--sun-visibility] [--seed S]
//             [--device D | --devices D0,D1,...] [--save-every K]
//   --sun-visibility: enable the occlusion query the reference has commented out (pathtracer.glsl:132)
//   --preview: what the reference shows in its window (shaders/screen.glsl), as PNG
//   --devices: pixel tiles sharded over several GPUs of the node (adypt_create_multi), radiance gathered on the first one
//   --save-every K: progressive rendering as in the reference's window — the running mean is written to --out (and --preview)
//                   every K samples; the file on disk is always a complete image of what has converged so far
#include "adypt_hip.h"
#include "adypt_host.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static double now_ms()
{
	return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
	if(argc < 2) { fprintf(stderr, "usage: %s scene.config [--spp N] [--out file.exr] [--fp16] [--primary TYPE] [--preview file.png] [--sun-visibility] [--seed S] [--device D | --devices D0,D1,...] [--save-every K]\n", argv[0]); return 2; }
	int spp = 64, fp16 = 0, primary = -1, sun_visibility = 0, save_every = 0;
	std::vector<int> devices(1, 0);
	unsigned seed = 12345;
	std::string out = "result.exr", preview;
	for(int i = 2; i < argc; ++i)
	{
		std::string a = argv[i];
		if(a == "--spp" && i + 1 < argc) spp = atoi(argv[++i]);
		else if(a == "--out" && i + 1 < argc) out = argv[++i];
		else if(a == "--preview" && i + 1 < argc) preview = argv[++i];
		else if(a == "--fp16") fp16 = 1;
		else if(a == "--sun-visibility") sun_visibility = 1;
		else if(a == "--test-hooks") (void)adypt_enable_test_hooks(ADYPT_TEST_HOOKS_MAGIC); // tests only: lets ADYPT_MULTI_SHARED_DEVICE put several shards on one device
		else if(a == "--primary" && i + 1 < argc) primary = atoi(argv[++i]);
		else if(a == "--seed" && i + 1 < argc) seed = (unsigned)strtoul(argv[++i], nullptr, 10);
		else if(a == "--device" && i + 1 < argc) devices.assign(1, atoi(argv[++i]));
		else if(a == "--devices" && i + 1 < argc)
		{
			devices.clear();
			for(const char *q = argv[++i]; *q;)
			{
				char *end = nullptr;
				const long v = strtol(q, &end, 10);
				if(end == q || v < 0) { fprintf(stderr, "bad --devices list %s\n", argv[i]); return 2; }
				devices.push_back((int)v);
				q = *end == ',' ? end + 1 : end;
				if(*end && *end != ',') { fprintf(stderr, "bad --devices list %s\n", argv[i]); return 2; }
			}
			if(devices.empty()) { fprintf(stderr, "empty --devices list\n"); return 2; }
		}
		else if(a == "--save-every" && i + 1 < argc) save_every = atoi(argv[++i]);
		else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
	}
	adypt_config cfg;
	adypt_config_default(&cfg);
	if(adypt_config_load(argv[1], &cfg) != ADYPT_OK) { fprintf(stderr, "[INSTANCE]Err: Invalid instance %s: %s\n", argv[1], adypt_host_last_error()); return 1; }
	printf("[INSTANCE]Info: Instance loaded from %s\n", argv[1]);

	adypt_scene *scene = nullptr;
	if(adypt_scene_load(cfg.obj_filename, &scene) != ADYPT_OK) { fprintf(stderr, "[INSTANCE]Err: Unable to load scene %s: %s\n", cfg.obj_filename, adypt_host_last_error()); return 1; }
	const void *tris, *mats, *tex;
	int64_t n_tris = adypt_scene_triangles(scene, &tris), n_mats = adypt_scene_materials(scene, &mats);
	int32_t n_tex = adypt_scene_textures(scene, &tex);
	printf("[SCENE]Info: %lld triangles loaded from %s\n", (long long)n_tris, cfg.obj_filename);
	if(*adypt_scene_warnings(scene)) printf("[SCENE]Warn: %s", adypt_scene_warnings(scene)); // undecodable textures: their materials render black

	adypt_bvh *bvh = nullptr;
	if(adypt_bvh_load(cfg.bvh_filename, &cfg.bvh, &bvh) != ADYPT_OK)
	{
		adypt_build_info info;
		if(adypt_bvh_build(scene, &cfg.bvh, &bvh, &info) != ADYPT_OK) { fprintf(stderr, "[INSTANCE]Err: bvh build failed: %s\n", adypt_host_last_error()); return 1; }
		printf("[SBVH]building lasted %.0f ms, %lld nodes, %lld references\n[WideBVH]built with %lld nodes (%.0f ms)\n", info.sbvh_ms,
			   (long long)info.sbvh_nodes, (long long)info.refs, (long long)info.wide_nodes, info.wide_ms);
		if(adypt_bvh_save(bvh, cfg.bvh_filename, &cfg.bvh) != ADYPT_OK) { fprintf(stderr, "[INSTANCE]Err: Unable to save bvh %s\n", cfg.bvh_filename); return 1; }
	}
	const void *nodes; const int32_t *idx;
	adypt_scene_desc d;
	memset(&d, 0, sizeof(d));
	d.n_nodes = adypt_bvh_nodes(bvh, &nodes); d.nodes = nodes;
	d.n_refs = adypt_bvh_tri_indices(bvh, &idx); d.tri_indices = idx;
	d.triangles = tris; d.n_tris = n_tris; d.materials = mats; d.n_mats = n_mats;
	d.textures = (const adypt_texture *)tex; d.n_textures = n_tex;
	d.width = cfg.width; d.height = cfg.height; d.device = devices[0]; d.tile_rank = 0; d.tile_nranks = 1;
	// one device: a plain context; several: the library's multi-device boundary (tile rank i on devices[i], one gather per saved image)
	adypt_ctx *ctx = nullptr;
	adypt_multi *multi = nullptr;
	const bool many = devices.size() > 1;
	if(many ? adypt_create_multi(&multi, &d, devices.data(), (int)devices.size()) != ADYPT_OK : adypt_create(&ctx, &d) != ADYPT_OK)
	{
		fprintf(stderr, "[TRACER]Err: %s\n", many ? adypt_multi_last_error(nullptr) : adypt_last_error(nullptr));
		return 1;
	}
	auto err = [&]() { return many ? adypt_multi_last_error(multi) : adypt_last_error(ctx); };
	adypt_pt_params p;
	p.stack_size = cfg.stack_size; p.max_bounce = cfg.max_bounce; p.subpixel = cfg.subpixel; p.tmp_lifetime = cfg.tmp_lifetime;
	p.ray_tmin = cfg.ray_tmin; p.clamp = cfg.clamp; memcpy(p.sun, cfg.sun, 12); p.shift_seed = seed;
	float ip[16], iv[16];
	adypt_camera_matrices(cfg.fov, cfg.yaw, cfg.pitch, cfg.width, cfg.height, ip, iv);
	int r = many ? adypt_multi_set_params(multi, &p) : adypt_set_params(ctx, &p);
	if(r == ADYPT_OK) r = many ? adypt_multi_set_camera(multi, cfg.position, ip, iv) : adypt_set_camera(ctx, cfg.position, ip, iv);
	if(r == ADYPT_OK && sun_visibility) r = many ? adypt_multi_set_sun_visibility(multi, 1, nullptr) : adypt_set_sun_visibility(ctx, 1, nullptr);
	if(r == ADYPT_OK) r = many ? adypt_multi_set_instrumentation(multi, 1) : adypt_set_instrumentation(ctx, 1);
	if(r != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", err()); return 1; }

	std::vector<float> rgb((size_t)cfg.width * cfg.height * 3, 0.0f);
	std::vector<uint8_t> rgba8;
	// SaveResult (OglPathTracer.cpp:199-212) + the window's picture; written to a temporary name first so that a reader of a
	// progressive render never sees a half-written file
	auto save = [&]() -> bool {
		if((many ? adypt_multi_read_radiance(multi, rgb.data()) : adypt_read_radiance(ctx, rgb.data())) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", err()); return false; }
		const std::string tmp = out + ".part";
		if(adypt_save_exr(tmp.c_str(), rgb.data(), cfg.width, cfg.height, fp16) != ADYPT_OK || rename(tmp.c_str(), out.c_str()) != 0) { fprintf(stderr, "[PT]ERR: %s\n", adypt_host_last_error()); return false; }
		if(!preview.empty())
		{
			rgba8.assign((size_t)cfg.width * cfg.height * 4, 0);
			if((many ? adypt_multi_read_display(multi, rgba8.data()) : adypt_read_display(ctx, rgba8.data())) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", err()); return false; }
			const std::string ptmp = preview + ".part";
			if(adypt_save_png(ptmp.c_str(), rgba8.data(), cfg.width, cfg.height) != ADYPT_OK || rename(ptmp.c_str(), preview.c_str()) != 0) { fprintf(stderr, "[PT]ERR: %s\n", adypt_host_last_error()); return false; }
		}
		return true;
	};

	double t0 = now_ms(), t_save = 0.0;
	if(primary >= 0) r = many ? adypt_multi_trace_primary(multi, primary) : adypt_trace_primary(ctx, primary);
	else if(save_every <= 0 || save_every >= spp) r = many ? adypt_multi_trace_spp(multi, spp) : adypt_trace_spp(ctx, spp);
	else
		for(int done = 0; done < spp && r == ADYPT_OK;)
		{
			const int n = std::min(save_every, spp - done);
			r = many ? adypt_multi_trace_spp(multi, n) : adypt_trace_spp(ctx, n);
			done += n;
			if(r == ADYPT_OK && done < spp)
			{
				const double ts = now_ms();
				if(!save()) return 1;
				t_save += now_ms() - ts;
				printf("[PT]INFO: %d spp saved to %s\n", done, out.c_str());
				fflush(stdout);
			}
		}
	double t1 = now_ms() - t_save;
	if(r != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", err()); return 1; }
	adypt_stats st;
	if((many ? adypt_multi_get_stats(multi, &st) : adypt_get_stats(ctx, &st)) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", err()); return 1; }
	printf("[PT]INFO: %d spp on %d GPU%s, %llu rays in %.1f ms wall (%.1f Mrays/s; traversal kernels %.1f ms, shade kernels %.1f ms)\n",
		   many ? adypt_multi_get_spp(multi) : adypt_get_spp(ctx), (int)devices.size(), many ? "s" : "", (unsigned long long)st.rays, t1 - t0,
		   st.rays / ((t1 - t0) * 1e3), st.trace_ms, st.shade_ms);
	if(!save()) return 1;
	printf("[PT]INFO: Saved image to %s\n", out.c_str());
	if(!preview.empty()) printf("[PT]INFO: Saved preview to %s\n", preview.c_str());
	if(many) adypt_destroy_multi(multi);
	else adypt_destroy(ctx);
	adypt_bvh_free(bvh);
	adypt_scene_free(scene);
	// like ~Instance (src/Instance.cpp:83-86): the config is written back on exit
	adypt_config_save(argv[1], &cfg);
	return 0;
}
