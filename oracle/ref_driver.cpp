// TEST INFRASTRUCTURE — not product code.
//
// Command-line driver around the *unmodified* Adypt reference sources, compiled from where
// they lie under /root/reference by oracle/Makefile (target `_ref`).  Nothing from the
// reference is copied into this repository: this file only #includes reference headers via -I
// and calls into them.  The binary lands in oracle/_ref/ (git-ignored) and is used
//   * to generate the golden vectors committed under tests/golden/ (tests/golden/make_golden.py)
//   * to pin oracle/oracle.cpp (the CPU restatement) and the product's host code against the
//     real reference CPU code (BVH arrays, Woop matrices, Sobol stream, .config JSON, EXR files)
//   * optionally as the "reference" CPU leg of config 1 (SBVH -> CWBVH8 build time).
// The GPU half of the reference (GLSL 4.5 compute shaders) cannot be compiled or run here.
//
// Sub-commands (all outputs are raw little-endian binary unless stated):
//   build  OBJ OUT.bvh DEPTH TRI_SAH NODE_SAH   Scene::LoadFromFile + SBVHBuilder + WideBVHBuilder + SaveToFile
//   scene  OBJ OUT.tris OUT.mats                Triangle[] (100 B) and GPUMaterial[] (64 B)
//   woop   OBJ IN.bvh DEPTH TRI_SAH NODE_SAH OUT.woop   OglScene::init_triangles (3 x vec4 per reference)
//   sobol  DIM N OUT.f32                        N calls of Sobol::Next with DIM dimensions
//   sobolmat ROWS OUT.u32                       first ROWS rows of kMatrices (32 u32 each)
//   config IN.config                            LoadFromFile; prints GetJson() to stdout; exit 1 if rejected
//   camera FOV YAW PITCH W H OUT.f32            inverse(projection)[16], inverse(view)[16] (column-major)
//   exrsave IN.rgbf32 W H FP16 OUT.exr          tinyexr SaveEXR exactly as OglPathTracer::SaveResult calls it
//   exrload IN.exr OUT.rgbaf32                  tinyexr LoadEXR -> prints "W H", writes RGBA f32
//   imgload IN.(jpg|png|...) OUT.rgb8           stbi_load(file, &w, &h, &c, 3) as OglScene::load_texture calls it -> prints "W H", writes RGB8
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <string>
#include <vector>

// OglScene keeps init_triangles / init_materials private; they are pure CPU code (glm only) as
// long as no texture is referenced, so the driver reaches them directly.
#define private public
#include "Tracer/OglScene.hpp"
#undef private
#include "BVH/SBVHBuilder.hpp"
#include "BVH/WideBVHBuilder.hpp"
#include "BVH/WideBVH.hpp"
#include "Util/Scene.hpp"
#include "Util/Sobol.hpp"
#include "InstanceConfig.hpp"
#include <glm/gtc/matrix_transform.hpp>
#include <tinyexr.h>
#include <stb_image.h> // declarations only: the implementation is compiled with the reference's OglScene.cpp

namespace sobol_data {
#include "Util/Sobol.inl"
}

static bool write_file(const char *fn, const void *p, size_t n)
{
	FILE *f = fopen(fn, "wb");
	if(!f) { fprintf(stderr, "cannot write %s\n", fn); return false; }
	fwrite(p, 1, n, f);
	fclose(f);
	return true;
}
static std::vector<char> read_file(const char *fn)
{
	std::vector<char> b;
	FILE *f = fopen(fn, "rb");
	if(!f) { fprintf(stderr, "cannot read %s\n", fn); exit(2); }
	fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
	b.resize((size_t)n);
	if(n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", fn); exit(2); }
	fclose(f);
	return b;
}

static InstanceConfig::BVH bvh_cfg(char **a)
{
	InstanceConfig::BVH c;
	c.m_max_spatial_depth = atoi(a[0]);
	c.m_triangle_sah = (float)atof(a[1]);
	c.m_node_sah = (float)atof(a[2]);
	return c;
}

int main(int argc, char **argv)
{
	if(argc < 2) { fprintf(stderr, "usage: adypt_ref <cmd> ...\n"); return 2; }
	std::string cmd = argv[1];

	if(cmd == "build" && argc == 7)
	{
		Scene scene;
		if(!scene.LoadFromFile(argv[2])) return 1;
		InstanceConfig::BVH cfg = bvh_cfg(argv + 4);
		WideBVH wbvh;
		SBVH sbvh;
		auto t0 = std::chrono::steady_clock::now();
		SBVHBuilder{cfg, &sbvh, scene}.Run();
		auto t1 = std::chrono::steady_clock::now();
		WideBVHBuilder{cfg, &wbvh, sbvh}.Run();
		auto t2 = std::chrono::steady_clock::now();
		if(!wbvh.SaveToFile(argv[3], cfg)) return 1;
		fprintf(stderr, "REF_BUILD tris=%zu sbvh_nodes=%zu refs=%zu wide_nodes=%zu sbvh_ms=%.1f wide_ms=%.1f\n",
				scene.GetTriangles().size(), sbvh.GetNodes().size(), wbvh.GetTriIndices().size(), wbvh.GetNodes().size(),
				std::chrono::duration<double, std::milli>(t1 - t0).count(),
				std::chrono::duration<double, std::milli>(t2 - t1).count());
		return 0;
	}
	if(cmd == "scene" && argc == 5)
	{
		Scene scene;
		if(!scene.LoadFromFile(argv[2])) return 1;
		static_assert(sizeof(Triangle) == 100, "Triangle layout");
		write_file(argv[3], scene.GetTriangles().data(), scene.GetTriangles().size() * sizeof(Triangle));
		OglScene ogl;
		std::vector<OglScene::GPUMaterial> mats;
		std::vector<GLuint64> handles;
		for(const auto &m : scene.GetTinyobjMaterials())
			if(!m.diffuse_texname.empty()) { fprintf(stderr, "textured material: needs GL, unsupported here\n"); return 1; }
		ogl.init_materials(scene, &mats, &handles);
		static_assert(sizeof(OglScene::GPUMaterial) == 64, "GPUMaterial layout");
		write_file(argv[4], mats.data(), mats.size() * sizeof(OglScene::GPUMaterial));
		return 0;
	}
	if(cmd == "woop" && argc == 8)
	{
		Scene scene;
		if(!scene.LoadFromFile(argv[2])) return 1;
		WideBVH wbvh;
		if(!wbvh.LoadFromFile(argv[3], bvh_cfg(argv + 4))) { fprintf(stderr, "bvh rejected\n"); return 1; }
		OglScene ogl;
		std::vector<glm::vec4> woop;
		ogl.init_triangles(scene, wbvh, &woop);
		write_file(argv[7], woop.data(), woop.size() * sizeof(glm::vec4));
		return 0;
	}
	if(cmd == "sobol" && argc == 5)
	{
		unsigned dim = (unsigned)atoi(argv[2]); int n = atoi(argv[3]);
		static Sobol gen;
		gen.Reset(dim);
		std::vector<float> out((size_t)dim * n);
		for(int i = 0; i < n; ++i) gen.Next(out.data() + (size_t)i * dim);
		write_file(argv[4], out.data(), out.size() * 4);
		return 0;
	}
	if(cmd == "sobolmat" && argc == 4)
	{
		int rows = atoi(argv[2]);
		write_file(argv[3], sobol_data::kMatrices, (size_t)rows * 32 * 4);
		return 0;
	}
	if(cmd == "config" && argc == 3)
	{
		InstanceConfig cfg;
		if(!cfg.LoadFromFile(argv[2])) return 1;
		fputs(cfg.GetJson().c_str(), stdout);
		return 0;
	}
	if(cmd == "camera" && argc == 8)
	{
		// Camera::GetView / GetProjection (src/Tracer/Camera.cpp:13-23) + OglPathTracer::SetCamera
		// (src/Tracer/OglPathTracer.cpp:27-32); Camera.cpp itself needs GLFW headers and cannot be compiled here,
		// so the same three glm calls are issued directly against the reference's vendored glm.
		float fov = (float)atof(argv[2]), yaw = (float)atof(argv[3]), pitch = (float)atof(argv[4]);
		int w = atoi(argv[5]), h = atoi(argv[6]);
		glm::mat4 view = glm::rotate(glm::identity<glm::mat4>(), glm::radians(-pitch), glm::vec3(1.0f, 0.0f, 0.0f));
		view = glm::rotate(view, glm::radians(-yaw), glm::vec3(0.0f, 1.0f, 0.0f));
		glm::mat4 proj = glm::tweakedInfinitePerspective(glm::radians(fov), w / (float)h, 0.01f);
		glm::mat4 out[2] = { glm::inverse(proj), glm::inverse(view) };
		write_file(argv[7], out, sizeof(out));
		return 0;
	}
	if(cmd == "exrsave" && argc == 7)
	{
		std::vector<char> px = read_file(argv[2]);
		int w = atoi(argv[3]), h = atoi(argv[4]), fp16 = atoi(argv[5]);
		const char *err = nullptr;
		int r = SaveEXR((const float *)px.data(), w, h, 3, fp16, argv[6], &err);
		if(r < 0) { fprintf(stderr, "SaveEXR: %s\n", err ? err : "?"); return 1; }
		return 0;
	}
	if(cmd == "exrload" && argc == 4)
	{
		float *rgba = nullptr; int w = 0, h = 0; const char *err = nullptr;
		if(LoadEXR(&rgba, &w, &h, argv[2], &err) < 0) { fprintf(stderr, "LoadEXR: %s\n", err ? err : "?"); return 1; }
		printf("%d %d\n", w, h);
		write_file(argv[3], rgba, (size_t)w * h * 16);
		free(rgba);
		return 0;
	}
	if(cmd == "imgload" && argc == 4)
	{
		int w = 0, h = 0, c = 0;
		unsigned char *px = stbi_load(argv[2], &w, &h, &c, 3);
		if(!px) { fprintf(stderr, "stbi_load: %s\n", stbi_failure_reason()); return 1; }
		printf("%d %d\n", w, h);
		write_file(argv[3], px, (size_t)w * h * 3);
		stbi_image_free(px);
		return 0;
	}
	fprintf(stderr, "bad command line\n");
	return 2;
}
