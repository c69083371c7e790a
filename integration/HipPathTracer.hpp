// The reference-side binding of INTEGRATION.md as a real header: dropped into Adypt as src/Tracer/HipPathTracer.hpp (and
// linked with -ladypt_hip) it replaces OglScene + OglPathTracer behind the same method names.  It is written against the
// reference's own headers (Scene, WideBVH, InstanceConfig, tinyobj, stb_image, tinyexr, glm) and is syntax-checked against
// them by tests/test_integration_header.py whenever /root/reference is present; nothing of the reference is copied here.
//
//   Instance.hpp:   HipPathTracer m_path_tracer;                       // instead of OglScene m_oglscene; OglPathTracer m_path_tracer;
//   Instance.cpp:   m_path_tracer.Initialize(&m_config.m_pt_cfg, scene, wbvh, m_config.m_width, m_config.m_height);
//                   m_path_tracer.SetCamera(m_camera.GetProjection(), m_camera.GetView(), m_config.m_cam_cfg.m_position);
//                   m_path_tracer.Trace(enable_pt);  ...  m_path_tracer.SaveResult(name, fp16);
#ifndef ADYPT_HIPPATHTRACER_HPP
#define ADYPT_HIPPATHTRACER_HPP

#include <adypt_hip.h>

#include "../Util/Scene.hpp"
#include "../BVH/WideBVH.hpp"
#include "../InstanceConfig.hpp"

#include <glm/glm.hpp>
#include <stb_image.h>
#include <tinyexr.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

class HipPathTracer
{
public:
	enum ViewerTypes { kDiffuse = 0, kSpecular, kEmissive, kPTRadiance, kNormal, kPosition };
	ViewerTypes m_viewer_type = kDiffuse;

private:
	// the 64-byte record uuMaterials holds (shaders/pathtracer.glsl:10-19)
	struct GPUMaterial
	{
		int32_t m_dtex; float m_dr, m_dg, m_db;
		int32_t m_etex; float m_er, m_eg, m_eb;
		int32_t m_stex; float m_sr, m_sg, m_sb;
		int32_t m_illum; float m_shininess, m_dissolve, m_refraction_index;
	};
	static_assert(sizeof(GPUMaterial) == 64, "material record layout");

	adypt_multi *m_gpus = nullptr; // one tile shard per device; a single device is the n_dev = 1 case of the same calls
	const InstanceConfig::PT *m_config = nullptr;
	int m_width = 0, m_height = 0;
	std::vector<GPUMaterial> m_materials;
	std::vector<adypt_texture> m_textures;
	std::vector<unsigned char *> m_texture_pixels; // stbi_load results, freed in the destructor

	// diffuse textures are shared by name; 8-bit RGB, first row = top (Scene.cpp flips v at load)
	int load_texture(std::map<std::string, int> &by_name, const std::string &filename)
	{
		std::map<std::string, int>::const_iterator it = by_name.find(filename);
		if(it != by_name.end()) return it->second;
		int w = 0, h = 0, channels = 0;
		unsigned char *data = stbi_load(filename.c_str(), &w, &h, &channels, 3);
		if(!data) { printf("[SCENE]ERR: unable to load texture %s\n", filename.c_str()); return -1; }
		adypt_texture t;
		t.width = w; t.height = h; t.rgb = data;
		m_textures.push_back(t);
		m_texture_pixels.push_back(data);
		return by_name[filename] = (int)m_textures.size() - 1;
	}

	void init_materials(const Scene &scene)
	{
		std::map<std::string, int> by_name;
		for(const tinyobj::material_t &ml : scene.GetTinyobjMaterials())
		{
			GPUMaterial g;
			g.m_dtex = ml.diffuse_texname.empty() ? -1 : load_texture(by_name, scene.GetBasePath() + ml.diffuse_texname);
			g.m_dr = ml.diffuse[0]; g.m_dg = ml.diffuse[1]; g.m_db = ml.diffuse[2];
			g.m_etex = -1; g.m_er = ml.emission[0]; g.m_eg = ml.emission[1]; g.m_eb = ml.emission[2];
			g.m_stex = -1; g.m_sr = ml.specular[0]; g.m_sg = ml.specular[1]; g.m_sb = ml.specular[2];
			g.m_illum = ml.illum; g.m_shininess = ml.shininess; g.m_dissolve = ml.dissolve; g.m_refraction_index = ml.ior;
			m_materials.push_back(g);
		}
	}

	bool update_config_args()
	{
		adypt_pt_params p;
		p.stack_size = m_config->m_stack_size; p.max_bounce = m_config->m_max_bounce;
		p.subpixel = m_config->m_subpixel; p.tmp_lifetime = m_config->m_tmp_lifetime;
		p.ray_tmin = m_config->m_ray_tmin; p.clamp = m_config->m_clamp;
		p.sun[0] = m_config->m_sun.x; p.sun[1] = m_config->m_sun.y; p.sun[2] = m_config->m_sun.z;
#ifdef ADYPT_BINDING_FIXED_SEED
		p.shift_seed = ADYPT_BINDING_FIXED_SEED; // reproducible runs (tests); the reference seeds its shift image from std::random_device
#else
		p.shift_seed = std::random_device{}();
#endif
		return adypt_multi_set_params(m_gpus, &p) == ADYPT_OK;
	}

public:
	HipPathTracer() = default;
	HipPathTracer(const HipPathTracer &) = delete;
	HipPathTracer &operator=(const HipPathTracer &) = delete;
	~HipPathTracer()
	{
		adypt_destroy_multi(m_gpus);
		for(unsigned char *p : m_texture_pixels) stbi_image_free(p);
	}

	// OglScene::Initialize(scene, bvh) + OglPathTracer::Initialize(config, oglscene, width, height).
	// devices: HIP device ordinals; with more than one the frame is sharded by 32x32 pixel tile over them (scene replicated,
	// no communication while rendering) and SaveResult gathers the radiance on devices[0] over RCCL.
	// lookahead_frames: Instance::Update calls Trace(true) once per window frame.  0 (default, the windowed application): every call
	// traces exactly its one frame — even frame pacing, the smallest ray queues (1 frame in flight: ~0.4 GB per device at 1080p).
	// N > 0 (headless rendering, e.g. the CLI loop that calls Trace(true) spp times and then SaveResult): a call that needs an untraced
	// frame traces a whole wavefront pass of N frames and the following N - 1 calls only apply their running-mean step — 1.4x the
	// rays per second (DESIGN.md §4 "one frame per call"), but bursty: one call in N takes N frames' time, and the queues grow to
	// N frames in flight (8 x 16 B x N x pixels per device).  Images are bit-identical either way.
	// 1 (a good default for a window): still one frame per wavefront pass and the smallest queues, but the frame after the one asked for is STARTED on a
	// second HIP stream before the call returns (under the end of the current frame's launch: +15 % at 1080p, even pacing); it is dropped when the camera
	// moves, a viewer frame is traced or the sample counter is reset.
	bool Initialize(const InstanceConfig::PT *config, const Scene &scene, const WideBVH &bvh, int width, int height,
	                const std::vector<int> &devices = std::vector<int>(1, 0), int lookahead_frames = 0)
	{
		m_config = config; m_width = width; m_height = height;
		init_materials(scene);
		adypt_scene_desc d;
		memset(&d, 0, sizeof(d));
		d.nodes = bvh.GetNodes().data();            d.n_nodes = (int64_t)bvh.GetNodes().size();
		d.tri_indices = bvh.GetTriIndices().data(); d.n_refs = (int64_t)bvh.GetTriIndices().size();
		d.woop = nullptr;                           // computed like OglScene::init_triangles
		d.triangles = scene.GetTriangles().data();  d.n_tris = (int64_t)scene.GetTriangles().size();
		d.materials = m_materials.data();           d.n_mats = (int64_t)m_materials.size();
		d.textures = m_textures.data();             d.n_textures = (int32_t)m_textures.size();
		d.width = width; d.height = height;
		if(adypt_create_multi(&m_gpus, &d, devices.data(), (int)devices.size()) != ADYPT_OK) { printf("[PT]ERR: %s\n", adypt_multi_last_error(nullptr)); return false; }
		const int fif = lookahead_frames > 0 ? (lookahead_frames < 128 ? lookahead_frames : 128) : 1;
		for(int i = 0; i < adypt_multi_device_count(m_gpus); ++i)
			if(adypt_set_frames_in_flight(adypt_multi_context(m_gpus, i), fif) != ADYPT_OK) { printf("[PT]ERR: %s\n", adypt_last_error(adypt_multi_context(m_gpus, i))); return false; }
		adypt_multi_set_lookahead(m_gpus, lookahead_frames > 0 ? 1 : 0);
		return update_config_args();
	}

	void SetCamera(const glm::mat4 &projection, const glm::mat4 &view, const glm::vec3 &position)
	{
		const glm::mat4 inv_projection = glm::inverse(projection), inv_view = glm::inverse(view);
		adypt_multi_set_camera(m_gpus, &position.x, &inv_projection[0][0], &inv_view[0][0]);
	}

	// Trace(true): one more sample per pixel; Trace(false): one primary-ray viewer frame and the sample counter restarts
	void Trace(bool enable_pt)
	{
		if(enable_pt)
		{
			if(adypt_multi_get_spp(m_gpus) == 0) update_config_args(); // the config is re-read when path tracing (re)starts
			m_viewer_type = kPTRadiance;
			if(adypt_multi_trace_spp(m_gpus, 1) != ADYPT_OK) printf("[PT]ERR: %s\n", adypt_multi_last_error(m_gpus));
		}
		else
		{
			if(m_viewer_type == kPTRadiance) m_viewer_type = kDiffuse;
			if(adypt_multi_trace_primary(m_gpus, m_viewer_type) != ADYPT_OK) printf("[PT]ERR: %s\n", adypt_multi_last_error(m_gpus));
		}
	}

	int GetSPP() const { return adypt_multi_get_spp(m_gpus); }

	// what DrawScreen puts on screen, for a caller-owned W x H RGBA8 texture / window (every device fills in its own tiles)
	bool ReadScreen(std::vector<uint8_t> *rgba8) const
	{
		rgba8->resize((size_t)m_width * m_height * 4);
		if(adypt_multi_read_display(m_gpus, rgba8->data()) != ADYPT_OK) return false; // every device converts and writes its own tiles
		return true;
	}

	void SaveResult(const char *filename, bool save_as_fp16)
	{
		std::vector<float> pixels((size_t)m_width * m_height * 3);
		if(adypt_multi_read_radiance(m_gpus, pixels.data()) != ADYPT_OK) { printf("[PT]ERR: %s\n", adypt_multi_last_error(m_gpus)); return; }
		const char *err = nullptr;
		if(SaveEXR(pixels.data(), m_width, m_height, 3, save_as_fp16, filename, &err) < 0)
		{
			printf("[PT]ERR: %s\n", err);
			FreeEXRErrorMessage(err);
		}
		else printf("[PT]INFO: Saved image to %s\n", filename);
	}
};

#endif // ADYPT_HIPPATHTRACER_HPP
