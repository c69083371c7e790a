/*
 * adypt_host.h — C-ABI of the host-side pieces either side of the GPU path: the .config scene interface, OBJ/MTL
 * scene loading, the CPU SBVH -> CWBVH8 builder and its .bvh cache, camera matrices, Sobol stream, EXR output.
 * They reproduce what Instance::Initialize (src/Instance.cpp:10-42) strings together in the reference so that the
 * tracer (adypt_hip.h) can run headless from an Adypt .config file.  None of these functions needs a GPU.
 */
#ifndef ADYPT_HOST_H
#define ADYPT_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- InstanceConfig (src/InstanceConfig.hpp:12-48, src/InstanceConfig.cpp:10-192) ------------------------------ */
typedef struct adypt_bvh_params { /* InstanceConfig::BVH; also the 12-byte header field of the .bvh cache */
	int32_t max_spatial_depth;
	float triangle_sah, node_sah;
} adypt_bvh_params;

typedef struct adypt_config {
	int32_t width, height;
	adypt_bvh_params bvh;
	/* InstanceConfig::PT */
	int32_t invocation_size, stack_size, max_bounce, subpixel, tmp_lifetime;
	float ray_tmin, clamp, sun[3];
	/* InstanceConfig::Cam */
	float speed, mouse_sensitive, fov, yaw, pitch, position[3];
	char obj_filename[1024], bvh_filename[1024];
} adypt_config;

void adypt_config_default(adypt_config *cfg);                       /* InstanceConfig::SetDefault */
int adypt_config_load(const char *path, adypt_config *cfg);         /* InstanceConfig::LoadFromFile (strict: all keys, typed) */
int adypt_config_parse(const char *json_text, adypt_config *cfg);
/* InstanceConfig::GetJson: writes at most cap bytes incl. NUL, returns the needed size */
size_t adypt_config_json(const adypt_config *cfg, char *buf, size_t cap);
int adypt_config_save(const char *path, const adypt_config *cfg);   /* InstanceConfig::SaveToFile */
const char *adypt_host_last_error(void);

/* Camera::Control (src/Tracer/Camera.cpp:25-59) without the window: the key / mouse state GLFW and ImGui would report is
 * passed in.  keys: ADYPT_KEY_* bits held this frame; mouse_dx/dy: cursor movement in pixels while the left button is
 * held (0 otherwise); frame_seconds: fps.GetDelta().  Same arithmetic: move = frame_seconds * speed along yaw (+90 / -90 /
 * 180 degrees for A / D / S), space / shift move along y, yaw -= dx * sensitivity (mod 360), pitch -= dy * sensitivity
 * clamped to [-90, 90]. */
enum { ADYPT_KEY_W = 1, ADYPT_KEY_A = 2, ADYPT_KEY_S = 4, ADYPT_KEY_D = 8, ADYPT_KEY_SPACE = 16, ADYPT_KEY_LEFT_SHIFT = 32 };
void adypt_camera_control(adypt_config *cfg, uint32_t keys, float mouse_dx, float mouse_dy, float frame_seconds);

/* ---- Scene (src/Util/Scene.cpp:9-136) + material/texture conversion (src/Tracer/OglScene.cpp:12-91) ------------- */
typedef struct adypt_scene adypt_scene;
int adypt_scene_load(const char *obj_path, adypt_scene **out);      /* Scene::LoadFromFile + init_materials */
void adypt_scene_free(adypt_scene *s);
int64_t adypt_scene_triangles(const adypt_scene *s, const void **tris);   /* 100-byte records, OBJ order */
int64_t adypt_scene_materials(const adypt_scene *s, const void **mats);   /* 64-byte GPUMaterial records */
int32_t adypt_scene_textures(const adypt_scene *s, const void **tex /* adypt_texture[] */);
void adypt_scene_aabb(const adypt_scene *s, float lo[3], float hi[3]);
/* "" or one line per diffuse texture that could not be decoded: like the reference after a failed stbi_load
 * (src/Tracer/OglScene.cpp:12-43) the scene still loads and the material renders with Kd = 0, but the loss is reported
 * (this loader reads PNM, PNG, BMP, TGA and JPEG with stb_image's pixels; stb_image also reads GIF, PSD, HDR, PIC and 16-bit BMP) */
const char *adypt_scene_warnings(const adypt_scene *s);
/* wrap caller-provided triangles (no OBJ): used for huge procedural scenes */
int adypt_scene_from_arrays(const void *tris, int64_t n_tris, const void *mats, int64_t n_mats, adypt_scene **out);

/* ---- BVH (src/BVH/SBVHBuilder.*, WideBVHBuilder.*, WideBVH.*) ---------------------------------------------------- */
typedef struct adypt_bvh adypt_bvh;
typedef struct adypt_build_info {
	int64_t sbvh_nodes, refs, wide_nodes;
	double sbvh_ms, wide_ms;
} adypt_build_info;
/* Worker threads of adypt_bvh_build (the reference's SBVHBuilder is single-threaded,
 * src/BVH/SBVHBuilder.cpp:49-71).  0 = automatic: $ADYPT_BUILD_THREADS, else the cores the process may use.  The
 * node and index arrays do not depend on the thread count. */
int adypt_host_set_threads(int n);
int adypt_host_get_threads(void);
/* Self-test of the multi-threaded sort the parallel build uses for its largest nodes (csrc/host/exact_sort.hpp): it
 * must return the very permutation std::sort returns (ties between the halves of a split triangle make the
 * permutation part of the node array).  pattern 0 random, 1 heavy ties, 2 sorted, 3 reversed, 4 all equal,
 * 5 quicksort killer (reaches the heap-sort fallback).  Returns 0 when both sorts agree byte for byte. */
int adypt_host_selftest_sort(int64_t n, uint32_t seed, int pattern, int threads, int64_t min_task);
/* SBVHBuilder{cfg,&sbvh,scene}.Run(); WideBVHBuilder{cfg,&wbvh,sbvh}.Run(); (src/Instance.cpp:24-26) */
int adypt_bvh_build(const adypt_scene *s, const adypt_bvh_params *p, adypt_bvh **out, adypt_build_info *info);
int adypt_bvh_load(const char *path, const adypt_bvh_params *expected, adypt_bvh **out);  /* WideBVH::LoadFromFile */
int adypt_bvh_save(const adypt_bvh *b, const char *path, const adypt_bvh_params *p);       /* WideBVH::SaveToFile */
void adypt_bvh_free(adypt_bvh *b);
int64_t adypt_bvh_nodes(const adypt_bvh *b, const void **nodes);          /* 80-byte records */
int64_t adypt_bvh_tri_indices(const adypt_bvh *b, const int32_t **idx);

/* ---- small restated host functions ---------------------------------------------------------------------------- */
/* OglScene::init_triangles (src/Tracer/OglScene.cpp:93-116): out = 12 floats per reference */
void adypt_woop_matrices(const void *tris, const int32_t *tri_indices, int64_t n_refs, float *out);
/* Camera::GetView/GetProjection + the inverses of SetCamera (src/Tracer/Camera.cpp:13-23, OglPathTracer.cpp:27-32) */
void adypt_camera_matrices(float fov, float yaw, float pitch, int width, int height, float inv_proj[16], float inv_view[16]);
/* Sobol::Next (src/Util/Sobol.cpp:16-21): points of frames [first, first+n), dim <= 64; out = n*dim floats */
int adypt_sobol_points(int dim, int first, int n, float *out);
/* RG8 shift image bytes from mt19937(seed) (src/Tracer/OglPathTracer.cpp:156-162): width*height*2 bytes */
void adypt_shift_bytes(uint32_t seed, int width, int height, uint8_t *out);
/* SaveEXR(rgb, W, H, 3, fp16, path) as OglPathTracer::SaveResult calls it (OglPathTracer.cpp:207): scanline,
 * ZIP, channels B,G,R, HALF or FLOAT */
int adypt_save_exr(const char *path, const float *rgb, int width, int height, int save_as_fp16);
/* 8-bit RGBA (adypt_read_display) -> PNG: the headless stand-in for the reference's window */
int adypt_save_png(const char *path, const uint8_t *rgba8, int width, int height);
/* minimal reader of the files adypt_save_exr writes (round-trip tests / tools) */
/* stbi_load(filename, &w, &h, &channels, 3) as OglScene::load_texture calls it (src/Tracer/OglScene.cpp:26-34): tightly packed RGB8, row 0
 * = top.  PNM, PNG (all depths, Adam7), BMP, TGA, JPEG (baseline / progressive; stb_image's own IDCT, up-sampling and colour arithmetic),
 * all pinned to the reference's stb_image by tests/golden/images.  *rgb is released with adypt_free. */
int adypt_load_image_rgb8(const char *path, uint8_t **rgb, int32_t *width, int32_t *height);
int adypt_load_exr(const char *path, float **rgb, int *width, int *height);
void adypt_free(void *p);

#ifdef __cplusplus
}
#endif
#endif /* ADYPT_HOST_H */
