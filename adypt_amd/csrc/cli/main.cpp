// adypt_hip — headless equivalent of the reference application for one .config file:
//   Instance::InitializeFromFile / Initialize (src/Instance.cpp:10-42,59-69) -> N x Trace(true) -> SaveResult
//   (src/Tracer/OglPathTracer.cpp:199-212).  The interactive window / ImGui front-end is out of scope.
//
//   adypt_hip scene.config [--spp N] [--out result.exr] [--fp16] [--primary TYPE] [--preview file.png] [--sun-visibility] [--seed S] [--device D]
//   --sun-visibility: enable the occlusion query the reference has commented out (pathtracer.glsl:132)
//   --preview: what the reference shows in its window (shaders/screen.glsl), as PNG
#include "adypt_hip.h"
#include "adypt_host.h"

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
	if(argc < 2) { fprintf(stderr, "usage: %s scene.config [--spp N] [--out file.exr] [--fp16] [--primary TYPE] [--preview file.png] [--sun-visibility] [--seed S] [--device D]\n", argv[0]); return 2; }
	int spp = 64, fp16 = 0, primary = -1, device = 0, sun_visibility = 0;
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
		else if(a == "--primary" && i + 1 < argc) primary = atoi(argv[++i]);
		else if(a == "--seed" && i + 1 < argc) seed = (unsigned)strtoul(argv[++i], nullptr, 10);
		else if(a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
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
	d.width = cfg.width; d.height = cfg.height; d.device = device; d.tile_rank = 0; d.tile_nranks = 1;
	adypt_ctx *ctx = nullptr;
	if(adypt_create(&ctx, &d) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(nullptr)); return 1; }
	adypt_pt_params p;
	p.stack_size = cfg.stack_size; p.max_bounce = cfg.max_bounce; p.subpixel = cfg.subpixel; p.tmp_lifetime = cfg.tmp_lifetime;
	p.ray_tmin = cfg.ray_tmin; p.clamp = cfg.clamp; memcpy(p.sun, cfg.sun, 12); p.shift_seed = seed;
	float ip[16], iv[16];
	adypt_camera_matrices(cfg.fov, cfg.yaw, cfg.pitch, cfg.width, cfg.height, ip, iv);
	if(adypt_set_params(ctx, &p) != ADYPT_OK || adypt_set_camera(ctx, cfg.position, ip, iv) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(ctx)); return 1; }
	if(sun_visibility && adypt_set_sun_visibility(ctx, 1, nullptr) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(ctx)); return 1; }
	adypt_set_instrumentation(ctx, 1);
	double t0 = now_ms();
	int r = primary >= 0 ? adypt_trace_primary(ctx, primary) : adypt_trace_spp(ctx, spp);
	double t1 = now_ms();
	if(r != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(ctx)); return 1; }
	adypt_stats st;
	adypt_get_stats(ctx, &st);
	printf("[PT]INFO: %d spp, %llu rays in %.1f ms wall (%.1f Mrays/s; traversal kernels %.1f ms, shade kernels %.1f ms)\n", adypt_get_spp(ctx),
		   (unsigned long long)st.rays, t1 - t0, st.rays / ((t1 - t0) * 1e3), st.trace_ms, st.shade_ms);
	std::vector<float> rgb((size_t)cfg.width * cfg.height * 3, 0.0f);
	if(adypt_read_radiance(ctx, rgb.data()) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(ctx)); return 1; }
	if(adypt_save_exr(out.c_str(), rgb.data(), cfg.width, cfg.height, fp16) != ADYPT_OK) { fprintf(stderr, "[PT]ERR: %s\n", adypt_host_last_error()); return 1; }
	printf("[PT]INFO: Saved image to %s\n", out.c_str());
	if(!preview.empty())
	{
		std::vector<uint8_t> rgba8((size_t)cfg.width * cfg.height * 4, 0);
		if(adypt_read_display(ctx, rgba8.data()) != ADYPT_OK) { fprintf(stderr, "[TRACER]Err: %s\n", adypt_last_error(ctx)); return 1; }
		if(adypt_save_png(preview.c_str(), rgba8.data(), cfg.width, cfg.height) != ADYPT_OK) { fprintf(stderr, "[PT]ERR: %s\n", adypt_host_last_error()); return 1; }
		printf("[PT]INFO: Saved preview to %s\n", preview.c_str());
	}
	adypt_destroy(ctx);
	adypt_bvh_free(bvh);
	adypt_scene_free(scene);
	// like ~Instance (src/Instance.cpp:83-86): the config is written back on exit
	adypt_config_save(argv[1], &cfg);
	return 0;
}
