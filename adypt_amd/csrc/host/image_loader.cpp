// Texture file decoding to tightly packed RGB8, rows top to bottom — what stbi_load(filename, &w, &h, &c, 3)
// hands to glTextureSubImage2D in the reference (src/Tracer/OglScene.cpp:26-34).  Formats: binary PPM/PGM,
// PNG (every colour type and bit depth, Adam7 interlace; inflate via zlib), BMP (24 / 32-bit and 4 / 8-bit palettised, uncompressed),
// TGA (true-colour 15-32 bits / grey / colour-mapped, raw or RLE), JPEG (baseline and progressive, jpeg_decoder.cpp: stb_image's arithmetic).  A file that
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

// BMP as stb_image reads it (dep/stb_image.h:4960-5230), uncompressed: 24 / 32 bits (BGR[A]) and 4 / 8-bit palettised (palette of
// BGR triples for the 12-byte core header, BGRX quads otherwise; 1-bit files are rejected, as stb_image rejects them); rows bottom-up unless the height is negative.
// 16-bit and bit-field BMPs are reported as unsupported.
bool decode_bmp(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	auto u32 = [&](size_t o) { return (uint32_t)b[o] | (uint32_t)b[o + 1] << 8 | (uint32_t)b[o + 2] << 16 | (uint32_t)b[o + 3] << 24; };
	auto u16 = [&](size_t o) { return (uint32_t)b[o] | (uint32_t)b[o + 1] << 8; };
	if(b.size() < 30) { *err = "short BMP"; return false; }
	const uint32_t off = u32(10), hsz = u32(14);
	int32_t w, h;
	uint32_t bpp, comp = 0;
	if(hsz == 12) { w = (int32_t)u16(18); h = (int32_t)u16(20); bpp = u16(24); }
	else
	{
		if(b.size() < 54 || !(hsz == 40 || hsz == 56 || hsz == 108 || hsz == 124)) { *err = "unsupported BMP header"; return false; }
		w = (int32_t)u32(18); h = (int32_t)u32(22); bpp = u16(28); comp = u32(30);
	}
	if(bpp == 1) { *err = "monochrome BMP (stb_image rejects it too)"; return false; }
	const bool paletted = bpp == 4 || bpp == 8;
	if(!(paletted || bpp == 24 || bpp == 32) || !(comp == 0 || (comp == 3 && bpp == 32))) { *err = "unsupported BMP variant"; return false; }
	const bool flip = h > 0;
	if(h < 0) h = -h;
	if(w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > ((uint64_t)1 << 28)) { *err = "bad BMP size"; return false; } // also keeps stride * h below SIZE_MAX
	std::vector<uint8_t> palette;
	if(paletted)
	{
		const size_t eb = hsz == 12 ? 3 : 4, first = 14 + (size_t)hsz;
		if(off < first || off > b.size()) { *err = "bad BMP palette"; return false; }
		const size_t n = std::min<size_t>((off - first) / eb, 256);
		if(n == 0) { *err = "bad BMP palette"; return false; }
		palette.assign(256 * 3, 0);
		for(size_t i = 0; i < n; ++i) { palette[i * 3] = b[first + i * eb + 2]; palette[i * 3 + 1] = b[first + i * eb + 1]; palette[i * 3 + 2] = b[first + i * eb]; }
	}
	const size_t stride = ((((size_t)w * bpp + 7) / 8) + 3) & ~(size_t)3;
	if((size_t)off + stride * (size_t)h > b.size()) { *err = "truncated BMP"; return false; }
	out->w = w; out->h = h;
	out->rgb.resize((size_t)w * h * 3);
	for(int y = 0; y < h; ++y)
	{
		const uint8_t *row = b.data() + off + stride * (size_t)(flip ? h - 1 - y : y);
		for(int x = 0; x < w; ++x)
		{
			uint8_t *o = &out->rgb[((size_t)y * w + x) * 3];
			if(paletted)
			{
				const size_t bit = (size_t)x * bpp;
				const unsigned idx = (row[bit >> 3] >> (8 - bpp - (bit & 7))) & ((1u << bpp) - 1u);
				memcpy(o, &palette[(size_t)idx * 3], 3);
			}
			else
			{
				const uint8_t *p = row + (size_t)x * (bpp / 8);
				o[0] = p[2]; o[1] = p[1]; o[2] = p[0];
			}
		}
	}
	return true;
}

// TGA as stb_image reads it (dep/stb_image.h:5240-5510): true-colour 24 / 32 bits (BGR[A]) and 15 / 16 bits (5-5-5, each field
// scaled (v * 255) / 31), grey 8 bits and 16 bits (grey + alpha), colour-mapped with 8- or 16-bit indices (an index past the palette
// reads entry 0), each raw or run-length encoded; rows bottom-up unless descriptor bit 5 is set; alpha dropped.
bool decode_tga(const std::vector<uint8_t> &b, TextureImage *out, std::string *err)
{
	if(b.size() < 18) { *err = "short TGA"; return false; }
	const int idlen = b[0], cmap = b[1], type = b[2];
	const int pal_start = b[3] | b[4] << 8, pal_len = b[5] | b[6] << 8, pal_bits = b[7];
	const int w = b[12] | b[13] << 8, h = b[14] | b[15] << 8, bpp = b[16], desc = b[17];
	const bool rle = type >= 8, indexed = (type & 7) == 1, grey = (type & 7) == 3;
	if(!((type & 7) >= 1 && (type & 7) <= 3) || type > 11 || w <= 0 || h <= 0) { *err = "unsupported TGA type"; return false; }
	if(indexed ? !(cmap == 1 && (bpp == 8 || bpp == 16) && (pal_bits == 15 || pal_bits == 16 || pal_bits == 24 || pal_bits == 32))
			   : !(cmap == 0 && (grey ? (bpp == 8 || bpp == 16) : (bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32))))
	{ *err = "unsupported TGA variant"; return false; }
	size_t pos = 18 + (size_t)idlen;
	// one decoded colour = RGB from a little-endian source pixel of `bits` bits
	auto to_rgb = [](const uint8_t *p, int bits, bool is_grey, uint8_t *o) {
		if(is_grey) { o[0] = o[1] = o[2] = p[0]; }
		else if(bits == 15 || bits == 16)
		{
			const unsigned px = p[0] | p[1] << 8;
			o[0] = (uint8_t)((((px >> 10) & 31) * 255) / 31); o[1] = (uint8_t)((((px >> 5) & 31) * 255) / 31); o[2] = (uint8_t)(((px & 31) * 255) / 31);
		}
		else { o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
	};
	std::vector<uint8_t> palette; // RGB triples
	if(indexed)
	{
		const size_t eb = (size_t)(pal_bits + 7) / 8;
		pos += (size_t)pal_start * eb; // stb skips `palette start` entries' worth of bytes before the table
		if(pal_len == 0 || pos + (size_t)pal_len * eb > b.size()) { *err = "truncated TGA palette"; return false; }
		palette.resize((size_t)pal_len * 3);
		for(int i = 0; i < pal_len; ++i) to_rgb(&b[pos + (size_t)i * eb], pal_bits, false, &palette[(size_t)i * 3]);
		pos += (size_t)pal_len * eb;
	}
	const int bytes = (bpp + 7) / 8;
	if((uint64_t)w * (uint64_t)h > ((uint64_t)1 << 28)) { *err = "TGA image too large"; return false; } // before anything is allocated
	std::vector<uint8_t> px((size_t)w * h * bytes);
	if(!rle)
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
			const int hdr = b[pos++], cnt = (hdr & 0x7f) + 1;
			if(hdr & 0x80)
			{
				if(pos + bytes > b.size()) { *err = "truncated TGA"; return false; }
				for(int i = 0; i < cnt && o < px.size(); ++i, o += bytes) memcpy(&px[o], &b[pos], bytes);
				pos += bytes;
			}
			else
			{
				const size_t nb = std::min((size_t)cnt * bytes, px.size() - o);
				if(pos + nb > b.size()) { *err = "truncated TGA"; return false; }
				memcpy(&px[o], &b[pos], nb);
				pos += nb; o += nb;
			}
		}
	}
	const bool top_origin = (desc & 0x20) != 0;
	out->w = w; out->h = h;
	out->rgb.resize((size_t)w * h * 3);
	for(int y = 0; y < h; ++y)
		for(int x = 0; x < w; ++x)
		{
			const uint8_t *p = &px[((size_t)(top_origin ? y : h - 1 - y) * w + x) * bytes];
			uint8_t *o = &out->rgb[((size_t)y * w + x) * 3];
			if(indexed)
			{
				size_t idx = bytes == 1 ? p[0] : (size_t)(p[0] | p[1] << 8);
				if(idx >= (size_t)pal_len) idx = 0;
				memcpy(o, &palette[idx * 3], 3);
			}
			else to_rgb(p, bpp, grey, o);
		}
	return true;
}

// PNG as stb_image reduces it to 8-bit RGB (dep/stb_image.h:4260-4560, 4680-4830): every colour type and bit depth, Adam7 interlace;
// grey samples of 1 / 2 / 4 bits are scaled to 0..255 (x 255 / 85 / 17), palette indices are not; 16-bit samples keep their high byte;
// alpha (channel or tRNS) is dropped, not blended.
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
	const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
						  (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
						  ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
	if(w == 0 || h == 0 || interlace > 1 || !depth_ok || (uint64_t)w * h > ((uint64_t)1 << 28)) { *err = "bad or unsupported PNG header"; return false; }
	const int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4;
	const size_t bits_px = (size_t)ch * depth, bpp = std::max<size_t>(1, bits_px / 8); // filter distance in bytes
	// pass geometry: one pass, or the seven of Adam7
	static const int xo[7] = {0, 4, 0, 2, 0, 1, 0}, yo[7] = {0, 0, 4, 0, 2, 0, 1}, xs[7] = {8, 8, 4, 4, 2, 2, 1}, ys[7] = {8, 8, 8, 4, 4, 2, 2};
	const int n_pass = interlace ? 7 : 1;
	size_t raw_size = 0;
	for(int p = 0; p < n_pass; ++p)
	{
		const size_t pw = interlace ? (w - xo[p] + xs[p] - 1) / xs[p] : w, ph = interlace ? (h - yo[p] + ys[p] - 1) / ys[p] : h;
		if(pw && ph) raw_size += ((pw * bits_px + 7) / 8 + 1) * ph;
	}
	std::vector<uint8_t> raw(raw_size);
	uLongf rawlen = (uLongf)raw.size();
	if(uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) { *err = "PNG inflate failed"; return false; }
	std::vector<uint8_t> samples((size_t)w * h * ch); // 8 bits per channel
	const uint8_t grey_scale = (ctype == 0 && depth < 8) ? (depth == 1 ? 0xff : depth == 2 ? 0x55 : 0x11) : 1;
	size_t rp = 0;
	std::vector<uint8_t> cur, prev;
	for(int p = 0; p < n_pass; ++p)
	{
		const size_t pw = interlace ? (w - xo[p] + xs[p] - 1) / xs[p] : w, ph = interlace ? (h - yo[p] + ys[p] - 1) / ys[p] : h;
		if(!pw || !ph) continue;
		const size_t stride = (pw * bits_px + 7) / 8;
		cur.assign(stride, 0); prev.assign(stride, 0);
		for(size_t y = 0; y < ph; ++y)
		{
			const uint8_t ft = raw[rp++];
			const uint8_t *src = &raw[rp];
			rp += stride;
			for(size_t i = 0; i < stride; ++i)
			{
				const int a = i >= bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
				int pr = 0;
				switch(ft)
				{
					case 0: pr = 0; break;
					case 1: pr = a; break;
					case 2: pr = bb; break;
					case 3: pr = (a + bb) >> 1; break;
					case 4: { const int q = a + bb - c, pa = abs(q - a), pb = abs(q - bb), pc = abs(q - c); pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
					default: *err = "bad PNG filter"; return false;
				}
				cur[i] = (uint8_t)(src[i] + pr);
			}
			const size_t oy = interlace ? (size_t)yo[p] + y * ys[p] : y;
			for(size_t x = 0; x < pw; ++x)
			{
				const size_t ox = interlace ? (size_t)xo[p] + x * xs[p] : x;
				uint8_t *o = &samples[(oy * w + ox) * ch];
				if(depth == 8) for(int k = 0; k < ch; ++k) o[k] = cur[x * ch + k];
				else if(depth == 16) for(int k = 0; k < ch; ++k) o[k] = cur[(x * ch + k) * 2]; // high byte
				else
				{
					const size_t bit = x * depth;
					o[0] = (uint8_t)(((cur[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1)) * grey_scale);
				}
			}
			cur.swap(prev);
		}
	}
	out->w = (int)w; out->h = (int)h;
	out->rgb.resize((size_t)w * h * 3);
	for(size_t i = 0; i < (size_t)w * h; ++i)
	{
		const uint8_t *p = &samples[i * ch];
		uint8_t *o = &out->rgb[i * 3];
		if(ctype == 3)
		{
			const size_t k = (size_t)p[0] * 3;
			if(k + 2 < plte.size()) { o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; }
			else o[0] = o[1] = o[2] = 0;
		}
		else if(ch <= 2) o[0] = o[1] = o[2] = p[0];
		else { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
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
