// TEST INFRASTRUCTURE — the drop-in, end to end: the REFERENCE's own CPU half (InstanceConfig, Scene + tinyobj, SBVHBuilder,
// WideBVHBuilder, WideBVH — compiled from /root/reference by oracle/Makefile, nothing copied) hands its arrays to
// libadypt_hip.so through integration/HipPathTracer.hpp, exactly where Instance::Initialize hands them to OglScene /
// OglPathTracer (src/Instance.cpp:10-42).  tests/test_gpu_dropin.py runs the binary on the GPU box and compares its EXR
// with the one the product's own loader + builder + CLI writes for the same .config: the bytes must be identical.
//
//   adypt_dropin scene.config out.exr SPP [fp16]
//
// The shift-image seed is fixed (ADYPT_BINDING_FIXED_SEED) where the binding normally draws from std::random_device.
#define STB_IMAGE_IMPLEMENTATION
#define ADYPT_BINDING_FIXED_SEED 12345u
#include "Tracer/HipPathTracer.hpp"

#include "BVH/SBVH.hpp"
#include "BVH/SBVHBuilder.hpp"
#include "BVH/WideBVHBuilder.hpp"
#include <glm/gtc/matrix_transform.hpp>

#include <cstdlib>

int main(int argc, char **argv)
{
	if(argc < 4) { fprintf(stderr, "usage: %s scene.config out.exr SPP [fp16 [lookahead_frames]]\n", argv[0]); return 2; }
	InstanceConfig config;
	if(!config.LoadFromFile(argv[1])) { fprintf(stderr, "invalid config %s\n", argv[1]); return 1; }
	Scene scene;
	if(!scene.LoadFromFile(config.m_obj_filename.c_str())) { fprintf(stderr, "cannot load %s\n", config.m_obj_filename.c_str()); return 1; }
	WideBVH wbvh;
	{
		// always rebuilt here (the product's CLI may have left its own .bvh next to the scene: the point is the reference's builder)
		SBVH sbvh;
		SBVHBuilder{config.m_bvh_cfg, &sbvh, scene}.Run();
		WideBVHBuilder{config.m_bvh_cfg, &wbvh, sbvh}.Run();
	}
	HipPathTracer tracer;
	// headless: look-ahead on (a pass of up to 32 frames per traced call), as a render-to-EXR loop would use it; 1 = what a window would use (one frame
	// per pass, the next one started ahead on a second stream); 0 = strictly one frame per call
	const int lookahead_frames = argc > 5 ? atoi(argv[5]) : 32;
	if(!tracer.Initialize(&config.m_pt_cfg, scene, wbvh, config.m_width, config.m_height, std::vector<int>(1, 0), lookahead_frames)) return 1;
	// Camera::GetView / GetProjection (src/Tracer/Camera.cpp:13-23; Camera.cpp itself needs GLFW and ImGui for Control())
	const InstanceConfig::Cam &cam = config.m_cam_cfg;
	glm::mat4 view = glm::rotate(glm::identity<glm::mat4>(), glm::radians(-cam.m_pitch), glm::vec3(1.0f, 0.0f, 0.0f));
	view = glm::rotate(view, glm::radians(-cam.m_yaw), glm::vec3(0.0f, 1.0f, 0.0f));
	const glm::mat4 projection = glm::tweakedInfinitePerspective(glm::radians(cam.m_fov), config.m_width / (float)config.m_height, 0.01f);
	tracer.SetCamera(projection, view, cam.m_position);
	const int spp = atoi(argv[3]);
	for(int i = 0; i < spp; ++i) tracer.Trace(true); // one OglPathTracer::Trace(true) per displayed frame, as Instance::Update does
	printf("[DROPIN]spp %d\n", tracer.GetSPP());
	tracer.SaveResult(argv[2], argc > 4 && atoi(argv[4]) != 0);
	return 0;
}
