// Texture file decoding to tightly packed RGB8, rows top to bottom — what stbi_load(filename, &w, &h, &c, 3)
// hands to glTextureSubImage2D in the reference (src/Tracer/OglScene.cpp:26-34).  Formats: binary PPM/PGM,
// PNG (8/16-bit, grey / RGB / palette / alpha, non-interlaced; inflate via zlib), BMP (24/32-bit uncompressed),
// TGA (true-colour / grey, raw or RLE), JPEG (baseline and progressive, jpeg_decoder.cpp: stb_image's arithmetic).  A file that
// cannot be decoded is reported (adypt_scene_warnings) and the material falls back to "texture missing" exactly like a failed
// stbi_load: dtex = -1, Kd = 0.
#include "common.hpp"

#include <cstdio>
#include <zlib.h>

namespace adypt {
namespace {

bool read_all(const std::string &path, std::vector<uint8_t> *buf)
{
	FILE *f = fopen(path.c_str(), "rb");
	if(!f) return false;
	fseek(f, 0, SEEK_END);
	long n = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf->resize((size_t)(n > 0 ? n : 0));
	bool ok = n <= 0 || fread(buf->data(), 1, (size_t)n, f) == (size_t)n;
	fclose(f);
	return ok;
}

bool decode_pnm(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	size_t pos = 2;
	auto next_int = [&](int *v) {
		for(;;)
		{
			while(pos < b.size() && (b[pos] == ' ' || b[pos] == '\n' || b[pos] == '\r' || b[pos] == '\t')) ++pos;
			if(pos < b.size() && b[pos] == '#') { while(pos < b.size() && b[pos] != '\n') ++pos; continue; }
			break;
		}
		if(pos >= b.size() || b[pos] < '0' || b[pos] > '9') return false;
		int x = 0;
		while(pos < b.size() && b[pos] >= '0' && b[pos] <= '9') x = x * 10 + (b[pos++] - '0');
		*v = x;
		return true;
	};
	int w, h, maxv;
	if(!next_int(&w) || !next_int(&h) || !next_int(&maxv)) { *err = "bad PNM header"; return false; }
	++pos; // single whitespace after maxval
	const int comps = b[1] == '6' ? 3 : 1;
	const int bps = maxv > 255 ? 2 : 1;
	if(w <= 0 || h <= 0 || pos + (size_t)w * h * comps * bps > b.size()) { *err = "truncated PNM"; return false; }
	out->w = w; out->h = h;
	out->rgb.resize((size_t)w * h * 3);
	for(size_t i = 0; i < (size_t)w * h; ++i)
		for(int c = 0; c < 3; ++c)
		{
			size_t src = pos + (i * comps + (comps == 3 ? c : 0)) * bps;
			out->rgb[i * 3 + c] = b[src]; // 16-bit: high byte first -> top 8 bits
		}
	return true;
}

bool decode_bmp(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	auto u32 = [&](size_t o) { return (uint32_t)b[o] | (uint32_t)b[o + 1] << 8 | (uint32_t)b[o + 2] << 16 | (uint32_t)b[o + 3] << 24; };
	auto u16 = [&](size_t o) { return (uint32_t)b[o] | (uint32_t)b[o + 1] << 8; };
	if(b.size() < 54) { *err = "short BMP"; return false; }
	uint32_t off = u32(10);
	int32_t w = (int32_t)u32(18), h = (int32_t)u32(22);
	uint32_t bpp = u16(28), comp = u32(30);
	if((bpp != 24 && bpp != 32) || (comp != 0 && comp != 3)) { *err = "unsupported BMP variant"; return false; }
	bool flip = h > 0;
	if(h < 0) h = -h;
	size_t stride = (((size_t)w * bpp / 8) + 3) & ~(size_t)3;
	if(w <= 0 || off + stride * (size_t)h > b.size()) { *err = "truncated BMP"; return false; }
	out->w = w; out->h = h;
	out->rgb.resize((size_t)w * h * 3);
	for(int y = 0; y < h; ++y)
	{
		const uint8_t *row = b.data() + off + stride * (size_t)(flip ? h - 1 - y : y);
		for(int x = 0; x < w; ++x)
		{
			const uint8_t *p = row + (size_t)x * (bpp / 8);
			uint8_t *o = &out->rgb[((size_t)y * w + x) * 3];
			o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
		}
	}
	return true;
}

bool decode_tga(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	if(b.size() < 18) { *err = "short TGA"; return false; }
	int idlen = b[0], cmap = b[1], type = b[2];
	int w = b[12] | b[13] << 8, h = b[14] | b[15] << 8, bpp = b[16], desc = b[17];
	if(cmap != 0 || !(type == 2 || type == 3 || type == 10 || type == 11) || !(bpp == 8 || bpp == 24 || bpp == 32))
	{ *err = "unsupported TGA variant"; return false; }
	size_t pos = 18 + (size_t)idlen;
	int bytes = bpp / 8;
	std::vector<uint8_t> px((size_t)w * h * bytes);
	if(type == 2 || type == 3)
	{
		if(pos + px.size() > b.size()) { *err = "truncated TGA"; return false; }
		memcpy(px.data(), b.data() + pos, px.size());
	}
	else
	{
		size_t o = 0;
		while(o < px.size())
		{
			if(pos >= b.size()) { *err = "truncated TGA"; return false; }
			int hdr = b[pos++], cnt = (hdr & 0x7f) + 1;
			if(hdr & 0x80)
			{
				if(pos + bytes > b.size()) { *err = "truncated TGA"; return false; }
				for(int i = 0; i < cnt && o < px.size(); ++i, o += bytes) memcpy(&px[o], &b[pos], bytes);
				pos += bytes;
			}
			else
			{
				size_t nb = (size_t)cnt * bytes;
				if(pos + nb > b.size() || o + nb > px.size()) { *err = "truncated TGA"; return false; }
				memcpy(&px[o], &b[pos], nb);
				pos += nb; o += nb;
			}
		}
	}
	bool top_origin = (desc & 0x20) != 0;
	out->w = w; out->h = h;
	out->rgb.resize((size_t)w * h * 3);
	for(int y = 0; y < h; ++y)
		for(int x = 0; x < w; ++x)
		{
			const uint8_t *p = &px[((size_t)(top_origin ? y : h - 1 - y) * w + x) * bytes];
			uint8_t *o = &out->rgb[((size_t)y * w + x) * 3];
			if(bytes == 1) o[0] = o[1] = o[2] = p[0];
			else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
		}
	return true;
}

bool decode_png(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	auto be32 = [&](size_t o) { return (uint32_t)b[o] << 24 | (uint32_t)b[o + 1] << 16 | (uint32_t)b[o + 2] << 8 | (uint32_t)b[o + 3]; };
	size_t pos = 8;
	uint32_t w = 0, h = 0;
	int depth = 0, ctype = 0, interlace = 0;
	std::vector<uint8_t> idat, plte;
	while(pos + 8 <= b.size())
	{
		uint32_t len = be32(pos);
		if(pos + 12 + (size_t)len > b.size()) break;
		const uint8_t *tag = &b[pos + 4], *data = &b[pos + 8];
		if(memcmp(tag, "IHDR", 4) == 0 && len >= 13)
		{
			w = be32(pos + 8); h = be32(pos + 12);
			depth = data[8]; ctype = data[9]; interlace = data[12];
		}
		else if(memcmp(tag, "PLTE", 4) == 0) plte.assign(data, data + len);
		else if(memcmp(tag, "IDAT", 4) == 0) idat.insert(idat.end(), data, data + len);
		else if(memcmp(tag, "IEND", 4) == 0) break;
		pos += 12 + (size_t)len;
	}
	if(w == 0 || h == 0 || interlace != 0 || !(depth == 8 || depth == 16 || (ctype == 3 && depth <= 8)))
	{ *err = "unsupported PNG variant (interlaced or sub-byte depth)"; return false; }
	int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
	if(ctype == 3 && depth != 8) { *err = "unsupported PNG palette depth"; return false; }
	size_t bpp = (size_t)ch * depth / 8, stride = (size_t)w * bpp;
	std::vector<uint8_t> raw((stride + 1) * h);
	uLongf rawlen = (uLongf)raw.size();
	if(uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size())
	{ *err = "PNG inflate failed"; return false; }
	std::vector<uint8_t> img(stride * h);
	for(uint32_t y = 0; y < h; ++y)
	{
		const uint8_t *src = &raw[(stride + 1) * y];
		uint8_t ft = src[0];
		++src;
		uint8_t *dst = &img[stride * y];
		const uint8_t *up = y ? &img[stride * (y - 1)] : nullptr;
		for(size_t i = 0; i < stride; ++i)
		{
			int a = i >= bpp ? dst[i - bpp] : 0, bb = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, pr = 0;
			switch(ft)
			{
				case 0: pr = 0; break;
				case 1: pr = a; break;
				case 2: pr = bb; break;
				case 3: pr = (a + bb) >> 1; break;
				case 4: { int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
				default: *err = "bad PNG filter"; return false;
			}
			dst[i] = (uint8_t)(src[i] + pr);
		}
	}
	out->w = (int)w; out->h = (int)h;
	out->rgb.resize((size_t)w * h * 3);
	const size_t bs = depth / 8;
	for(size_t i = 0; i < (size_t)w * h; ++i)
	{
		const uint8_t *p = &img[i * bpp];
		uint8_t *o = &out->rgb[i * 3];
		if(ctype == 3)
		{
			size_t k = (size_t)p[0] * 3;
			if(k + 2 < plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; }
			else o[0] = o[1] = o[2] = 0;
		}
		else if(ch <= 2) o[0] = o[1] = o[2] = p[0];
		else { o[0] = p[0]; o[1] = p[bs]; o[2] = p[2 * bs]; }
	}
	return true;
}

}  // namespace

bool decode_image_rgb8(const std::string &path, TextureImage *out, std::string *err)
{
	std::vector<uint8_t> b;
	if(!read_all(path, &b) || b.size() < 8) { *err = "cannot read file"; return false; }
	if(b[0] == 'P' && (b[1] == '6' || b[1] == '5')) return decode_pnm(b, out, err);
	if(memcmp(b.data(), "\x89PNG\r\n\x1a\n", 8) == 0) return decode_png(b, out, err);
	if(b[0] == 'B' && b[1] == 'M') return decode_bmp(b, out, err);
	if(b[0] == 0xff && b[1] == 0xd8) return decode_jpeg(b, out, err);
	size_t dot = path.find_last_of('.');
	std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
	for(char &c : ext) c = (char)tolower(c);
	if(ext == "tga") return decode_tga(b, out, err);
	*err = "unknown image format";
	return false;
}

}  // namespace adypt
