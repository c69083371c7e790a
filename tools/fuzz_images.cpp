// Developer tool (CPU, no GPU): mutation fuzzer for the texture decoders under AddressSanitizer + UBSan.
//   cd adypt_amd/csrc && g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off -pthread -DADYPT_BUILD \
//        -I../../include -o /tmp/fuzz_images ../../tools/fuzz_images.cpp host/*.cpp -lz && /tmp/fuzz_images $PWD/../../tests/golden/images 400
// Round 2: 23 600 mutated files (truncations, byte flips, injected 0xff markers) of the 59 fixtures: no finding after the IDCT went to 64-bit.
#include "../include/adypt_host.h"
#include "../include/adypt_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <random>
#include <dirent.h>
int main(int argc, char **argv)
{
	std::string dir = argv[1];
	std::vector<std::string> files;
	DIR *d = opendir(dir.c_str());
	while(dirent *e = readdir(d)) { std::string n = e->d_name; if(n.size() > 4 && (n.rfind(".jpg") == n.size() - 4 || n.rfind(".png") == n.size() - 4 || n.rfind(".tga") == n.size() - 4 || n.rfind(".bmp") == n.size() - 4)) files.push_back(n); }
	closedir(d);
	std::mt19937 rng(1);
	long ok = 0, bad = 0;
	const int iters = atoi(argv[2]);
	for(const std::string &f : files)
	{
		FILE *fp = fopen((dir + "/" + f).c_str(), "rb");
		std::vector<unsigned char> b(1 << 20);
		b.resize(fread(b.data(), 1, b.size(), fp));
		fclose(fp);
		std::string ext = f.substr(f.size() - 4);
		for(int k = 0; k < iters; ++k)
		{
			std::vector<unsigned char> m = b;
			int mode = rng() % 4;
			if(mode == 0) m.resize(rng() % m.size() + 1);
			else for(int j = 0, n = rng() % 8 + 1; j < n; ++j) m[rng() % m.size()] = (unsigned char)rng();
			if(mode == 3) for(int j = 0; j < 3; ++j) { size_t p = rng() % m.size(); m[p] = 0xff; }
			std::string tmp = std::string("/tmp/adypt_fuzz_in") + ext;
			fp = fopen(tmp.c_str(), "wb"); fwrite(m.data(), 1, m.size(), fp); fclose(fp);
			uint8_t *rgb = nullptr; int32_t w = 0, h = 0;
			if(adypt_load_image_rgb8(tmp.c_str(), &rgb, &w, &h) == 0) { ++ok; volatile unsigned s = 0; for(size_t i = 0; i < (size_t)w * h * 3; ++i) s += rgb[i]; adypt_free(rgb); }
			else ++bad;
		}
	}
	printf("files %zu decoded %ld rejected %ld\n", files.size(), ok, bad);
	return 0;
}
