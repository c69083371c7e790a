// Developer tool (CPU, no GPU): mutation fuzzer for the OBJ / MTL loader (+ BVH build), the .bvh cache reader and the .config parser under
// AddressSanitizer + UBSan.  Build like tools/fuzz_images.cpp; run: /tmp/fuzz_loaders $PWD/tests/golden 6000
// Round 2 findings, fixed: a face index beyond the vertices defined so far was read unchecked (the reference has the same hole:
// src/Util/Scene.cpp:60-75 indexes attrib.vertices with whatever tinyobj stored); memcpy(nullptr, ..., 0) for an empty node array.
#include "../include/adypt_host.h"
#include "../include/adypt_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
static std::vector<unsigned char> rd(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); std::vector<unsigned char> b(1 << 22); b.resize(fread(b.data(), 1, b.size(), f)); fclose(f); return b; }
static void wr(const std::string &p, const std::vector<unsigned char> &b) { FILE *f = fopen(p.c_str(), "wb"); fwrite(b.data(), 1, b.size(), f); fclose(f); }
int main(int argc, char **argv)
{
	const std::string g = argv[1];
	const int iters = atoi(argv[2]);
	std::mt19937 rng(2);
	auto mutate = [&](std::vector<unsigned char> m, bool text) {
		int mode = rng() % 3;
		if(mode == 0) m.resize(rng() % m.size() + 1);
		else for(int j = 0, n = rng() % 6 + 1; j < n; ++j) { size_t p = rng() % m.size(); m[p] = text ? (unsigned char)("0123456789-+.e/ \n#fvtn{}[]:,\"a"[rng() % 31]) : (unsigned char)rng(); }
		return m;
	};
	long ok = 0, bad = 0;
	{	// OBJ + MTL
		std::vector<unsigned char> obj = rd(g + "/tiny2.obj"), mtl = rd(g + "/tiny2.mtl");
		for(int k = 0; k < iters; ++k)
		{
			wr("/tmp/adypt_fuzz_s.obj", k % 2 ? mutate(obj, true) : obj);
			std::vector<unsigned char> m2 = k % 2 ? mtl : mutate(mtl, true);
			wr("/tmp/adypt_fuzz_tiny2.mtl", m2);
			adypt_scene *s = nullptr;
			if(adypt_scene_load("/tmp/adypt_fuzz_s.obj", &s) == 0)
			{
				++ok;
				if(k % 8 == 0)
				{
					adypt_bvh_params p; p.max_spatial_depth = 48; p.triangle_sah = 0.3f; p.node_sah = 1.0f;
					adypt_bvh *b = nullptr; adypt_build_info bi;
					if(adypt_bvh_build(s, &p, &b, &bi) == 0) adypt_bvh_free(b);
				}
				adypt_scene_free(s);
			}
			else ++bad;
		}
	}
	printf("obj: ok %ld rejected %ld\n", ok, bad); ok = bad = 0;
	{	// .bvh
		std::vector<unsigned char> bvh = rd(g + "/tiny1.bvh");
		adypt_bvh_params p; p.max_spatial_depth = 48; p.triangle_sah = 0.3f; p.node_sah = 1.0f;
		for(int k = 0; k < iters; ++k)
		{
			wr("/tmp/adypt_fuzz_x.bvh", mutate(bvh, false));
			adypt_bvh *b = nullptr;
			if(adypt_bvh_load("/tmp/adypt_fuzz_x.bvh", &p, &b) == 0) { ++ok; adypt_bvh_free(b); } else ++bad;
		}
	}
	printf("bvh: ok %ld rejected %ld\n", ok, bad); ok = bad = 0;
	{	// .config
		adypt_config c; adypt_config_default(&c);
		std::vector<char> txt(1 << 16);
		size_t n = adypt_config_json(&c, txt.data(), txt.size());
		std::vector<unsigned char> base(txt.begin(), txt.begin() + n);
		for(int k = 0; k < iters * 4; ++k)
		{
			std::vector<unsigned char> m = mutate(base, true);
			m.push_back(0);
			adypt_config c2;
			if(adypt_config_parse((const char *)m.data(), &c2) == 0) ++ok; else ++bad;
		}
	}
	printf("config: ok %ld rejected %ld\n", ok, bad);
	return 0;
}
