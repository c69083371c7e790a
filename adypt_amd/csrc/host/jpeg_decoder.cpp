// JPEG texture decoding to tightly packed RGB8 with exactly the pixels stbi_load(filename, &w, &h, &c, 3) hands to
// glTextureSubImage2D in the reference (src/Tracer/OglScene.cpp:26-34; dep/stb_image.h, the `stbi__jpeg` decoder, lines
// 1650-3700).  JPEG decoders differ in their inverse DCT, chroma up-sampling filter and colour matrix, and diffuse textures feed
// the radiance directly (shaders/pathtracer.glsl:88-96), so "some JPEG decoder" would not do: this one restates stb_image's
// arithmetic — the integer IDCT with its two rounding points (stb_image.h:2110-2206), on-the-fly dequantisation into 16-bit
// coefficients (:1905-1955), the centred 3:1 / 9:3:3:1 chroma filters and their edge rules (:3124-3187), the 20-bit fixed-point
// YCbCr matrix with the masked green term (:3317-3343), the row stepping of load_jpeg_image (:3560-3600) — and is pinned to the
// reference's own stb_image, compiled from its source by oracle/Makefile, on fixtures covering baseline / progressive, 4:4:4 /
// 4:2:2 / 4:2:0 / 4:4:0 / 4:1:1 sampling, grey, CMYK and restart intervals (tests/golden/images, tests/test_host_golden.py).
//
// What is this file's own: the bit reader, a canonical-code Huffman decoder without acceleration tables, the marker walk.  Bit
// streams that stb_image rejects are rejected here too, but not necessarily with the same partial output: for corrupt files the
// only contract is "an error, no crash".
#include "common.hpp"

#include <cstring>

namespace adypt {
namespace {

const uint8_t kZigZag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
							 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
	// canonical code: codes of length L are consecutive integers starting at first_code[L], their symbols start at first_sym[L]
	int count[17] = {0}, first_code[17] = {0}, first_sym[17] = {0};
	uint8_t symbols[256] = {0};
	bool defined = false;
	bool build(const int counts[16])
	{
		int code = 0, k = 0;
		for(int len = 1; len <= 16; ++len)
		{
			count[len] = counts[len - 1];
			first_code[len] = code; first_sym[len] = k;
			code += count[len]; k += count[len];
			if(code > (1 << len)) return false; // more codes of this length than exist (stb: "bad code lengths")
			code <<= 1;
		}
		defined = k <= 256;
		return defined;
	}
};

struct Component {
	int id = 0, h = 1, v = 1, tq = 0, hd = 0, ha = 0;
	int dc_pred = 0;
	int x = 0, y = 0;    // effective pixels of this component
	int w2 = 0, h2 = 0;  // allocated (whole MCUs)
	std::vector<uint8_t> data;   // decoded samples, w2 x h2
	std::vector<int16_t> coeff;  // progressive: 64 coefficients per block, (w2 / 8) x (h2 / 8) blocks
};

struct Decoder {
	const uint8_t *p = nullptr, *end = nullptr;
	std::string err;
	// entropy-coded segment reader
	uint32_t bitbuf = 0; int bitcnt = 0;
	int marker = -1; // marker met inside the entropy-coded data (the reader then feeds zero bits, as stb's does)
	bool fail(const char *m) { if(err.empty()) err = m; return false; }

	int get8() { return p < end ? *p++ : 0; }
	int get16() { int a = get8(); return a << 8 | get8(); }
	bool eof() const { return p >= end; }

	void fill()
	{
		while(bitcnt <= 24)
		{
			int b = 0;
			if(marker < 0)
			{
				b = get8();
				if(b == 0xff)
				{
					int c = get8();
					while(c == 0xff) c = get8(); // fill bytes
					if(c != 0) { marker = c; b = 0; }
				}
			}
			bitbuf |= (uint32_t)b << (24 - bitcnt);
			bitcnt += 8;
		}
	}
	int bits(int n) // n in 0..16, MSB first
	{
		if(n == 0) return 0;
		if(bitcnt < n) fill();
		const uint32_t v = bitbuf >> (32 - n);
		bitbuf <<= n; bitcnt -= n;
		return (int)v;
	}
	int bit() { return bits(1); }
	// JPEG RECEIVE + EXTEND (ITU T.81 F.2.2.1)
	int receive_extend(int n)
	{
		if(n == 0) return 0;
		const int v = bits(n);
		return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
	}
	int decode(const Huffman &h)
	{
		if(!h.defined) { fail("JPEG: scan uses an undefined Huffman table"); return -1; }
		int code = 0;
		for(int len = 1; len <= 16; ++len)
		{
			code = code << 1 | bit();
			const int idx = code - h.first_code[len];
			if(idx >= 0 && idx < h.count[len]) return h.symbols[h.first_sym[len] + idx];
		}
		fail("JPEG: bad Huffman code");
		return -1;
	}
	void reset_entropy() { bitbuf = 0; bitcnt = 0; marker = -1; }

	// frame / scan state
	int width = 0, height = 0, ncomp = 0;
	bool progressive = false, jfif = false;
	int adobe_transform = -1, rgb_ids = 0;
	int hmax = 1, vmax = 1, mcu_x = 0, mcu_y = 0;
	int restart_interval = 0;
	Component comp[4];
	uint16_t dequant[4][64] = {{0}};
	Huffman huff_dc[4], huff_ac[4];
	int scan_n = 0, order[4] = {0, 0, 0, 0};
	int spec_start = 0, spec_end = 63, succ_high = 0, succ_low = 0, eob_run = 0;
};

inline uint8_t clamp8(int x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }

// stb_image.h:2110-2206 (its scalar, SSE2 and NEON kernels are bit-identical by construction): jidctint's "islow" butterflies
// with 12-bit constants; the column pass keeps 2 extra bits ((x + 512) >> 10), the row pass removes 17 with the +128 level shift
// folded into the rounding constant; columns whose AC terms are all zero short-cut to dc << 2.
#define ADYPT_F2F(x) ((long long)((x) * 4096 + 0.5))
#define ADYPT_IDCT_1D(s0, s1, s2, s3, s4, s5, s6, s7)                                                  \
	long long t0, t1, t2, t3, p1, p2, p3, p4, p5, x0, x1, x2, x3; /* 64-bit: a corrupt stream cannot overflow them */ \
	p2 = s2; p3 = s6;                                                                                  \
	p1 = (p2 + p3) * ADYPT_F2F(0.5411961f);                                                            \
	t2 = p1 + p3 * ADYPT_F2F(-1.847759065f);                                                           \
	t3 = p1 + p2 * ADYPT_F2F(0.765366865f);                                                            \
	p2 = s0; p3 = s4;                                                                                  \
	t0 = (p2 + p3) * 4096; t1 = (p2 - p3) * 4096;                                                      \
	x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;                                            \
	t0 = s7; t1 = s5; t2 = s3; t3 = s1;                                                                \
	p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;                                            \
	p5 = (p3 + p4) * ADYPT_F2F(1.175875602f);                                                          \
	t0 = t0 * ADYPT_F2F(0.298631336f); t1 = t1 * ADYPT_F2F(2.053119869f);                              \
	t2 = t2 * ADYPT_F2F(3.072711026f); t3 = t3 * ADYPT_F2F(1.501321110f);                              \
	p1 = p5 + p1 * ADYPT_F2F(-0.899976223f); p2 = p5 + p2 * ADYPT_F2F(-2.562915447f);                  \
	p3 = p3 * ADYPT_F2F(-1.961570560f); p4 = p4 * ADYPT_F2F(-0.390180644f);                            \
	t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;

inline uint8_t clamp8(long long x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }

void idct_block(uint8_t *out, int stride, const int16_t d[64])
{
	long long val[64];
	for(int i = 0; i < 8; ++i)
	{
		const int16_t *c = d + i;
		long long *v = val + i;
		if(c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0)
		{
			const long long dc = c[0] * 4;
			v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dc;
		}
		else
		{
			ADYPT_IDCT_1D(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56])
			x0 += 512; x1 += 512; x2 += 512; x3 += 512;
			v[0] = (x0 + t3) >> 10; v[56] = (x0 - t3) >> 10;
			v[8] = (x1 + t2) >> 10; v[48] = (x1 - t2) >> 10;
			v[16] = (x2 + t1) >> 10; v[40] = (x2 - t1) >> 10;
			v[24] = (x3 + t0) >> 10; v[32] = (x3 - t0) >> 10;
		}
	}
	for(int i = 0; i < 8; ++i)
	{
		const long long *v = val + i * 8;
		uint8_t *o = out + (size_t)i * stride;
		ADYPT_IDCT_1D(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7])
		x0 += 65536 + (128 << 17); x1 += 65536 + (128 << 17); x2 += 65536 + (128 << 17); x3 += 65536 + (128 << 17);
		o[0] = clamp8((x0 + t3) >> 17); o[7] = clamp8((x0 - t3) >> 17);
		o[1] = clamp8((x1 + t2) >> 17); o[6] = clamp8((x1 - t2) >> 17);
		o[2] = clamp8((x2 + t1) >> 17); o[5] = clamp8((x2 - t1) >> 17);
		o[3] = clamp8((x3 + t0) >> 17); o[4] = clamp8((x3 - t0) >> 17);
	}
}

// one baseline block: DC difference + run-length AC, de-zigzagged and dequantised into 16-bit on the fly (stb_image.h:1905-1955)
bool decode_block_baseline(Decoder &z, int16_t data[64], Component &c)
{
	memset(data, 0, 64 * sizeof(int16_t));
	const uint16_t *dq = z.dequant[c.tq];
	const int t = z.decode(z.huff_dc[c.hd]);
	if(t < 0 || t > 15) return z.fail("JPEG: bad DC code");
	// (unsigned arithmetic: a corrupt stream may push the prediction out of int range — wraps instead of overflowing; identical
	// to stb's int arithmetic for every stream whose values fit)
	c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)z.receive_extend(t));
	data[0] = (int16_t)((uint32_t)c.dc_pred * (uint32_t)dq[0]);
	for(int k = 1; k < 64;)
	{
		const int rs = z.decode(z.huff_ac[c.ha]);
		if(rs < 0) return false;
		const int s = rs & 15, r = rs >> 4;
		if(s == 0)
		{
			if(rs != 0xf0) break; // end of block
			k += 16;
		}
		else
		{
			k += r;
			if(k > 63) return z.fail("JPEG: AC run past the block");
			const int zig = kZigZag[k++];
			data[zig] = (int16_t)(z.receive_extend(s) * dq[zig]);
		}
	}
	return true;
}

// progressive scans accumulate into the coefficient array (stb_image.h:1957-2105)
bool decode_block_prog_dc(Decoder &z, int16_t data[64], Component &c)
{
	if(z.spec_end != 0) return z.fail("JPEG: progressive scan mixes DC and AC");
	if(z.succ_high == 0)
	{
		memset(data, 0, 64 * sizeof(int16_t));
		const int t = z.decode(z.huff_dc[c.hd]);
		if(t < 0 || t > 15) return z.fail("JPEG: bad DC code");
		c.dc_pred = (int)((uint32_t)c.dc_pred + (uint32_t)z.receive_extend(t));
		data[0] = (int16_t)((uint32_t)c.dc_pred * (1u << z.succ_low));
	}
	else if(z.bit()) data[0] = (int16_t)(data[0] + (int16_t)(1 << z.succ_low));
	return true;
}

bool decode_block_prog_ac(Decoder &z, int16_t data[64], Component &c)
{
	if(z.spec_start == 0) return z.fail("JPEG: progressive scan mixes DC and AC");
	const Huffman &hac = z.huff_ac[c.ha];
	if(z.succ_high == 0)
	{
		const int shift = z.succ_low;
		if(z.eob_run) { --z.eob_run; return true; }
		int k = z.spec_start;
		do
		{
			const int rs = z.decode(hac);
			if(rs < 0) return false;
			const int s = rs & 15, r = rs >> 4;
			if(s == 0)
			{
				if(r < 15)
				{
					z.eob_run = (1 << r);
					if(r) z.eob_run += z.bits(r);
					--z.eob_run;
					break;
				}
				k += 16;
			}
			else
			{
				k += r;
				if(k > 63) return z.fail("JPEG: AC run past the block");
				const int zig = kZigZag[k++];
				data[zig] = (int16_t)(z.receive_extend(s) * (1 << shift));
			}
		} while(k <= z.spec_end);
	}
	else
	{
		const int16_t bit = (int16_t)(1 << z.succ_low);
		auto refine = [&](int16_t *p) {
			if(z.bit() && (*p & bit) == 0) *p = (int16_t)(*p > 0 ? *p + bit : *p - bit);
		};
		if(z.eob_run)
		{
			--z.eob_run;
			for(int k = z.spec_start; k <= z.spec_end; ++k)
			{
				int16_t *p = &data[kZigZag[k]];
				if(*p != 0) refine(p);
			}
		}
		else
		{
			int k = z.spec_start;
			do
			{
				const int rs = z.decode(hac);
				if(rs < 0) return false;
				int s = rs & 15, r = rs >> 4;
				if(s == 0)
				{
					if(r < 15)
					{
						z.eob_run = (1 << r) - 1;
						if(r) z.eob_run += z.bits(r);
						r = 64; // force end of block
					}
					// r == 15: a run of 16 zeros = skip 15 and place a zero
				}
				else
				{
					if(s != 1) return z.fail("JPEG: bad refinement code");
					s = z.bit() ? bit : -bit;
				}
				while(k <= z.spec_end)
				{
					int16_t *p = &data[kZigZag[k++]];
					if(*p != 0) refine(p);
					else
					{
						if(r == 0) { *p = (int16_t)s; break; }
						--r;
					}
				}
			} while(k <= z.spec_end);
		}
	}
	return true;
}

bool process_marker(Decoder &z, int m)
{
	if(m == 0xdd) // DRI
	{
		if(z.get16() != 4) return z.fail("JPEG: bad DRI length");
		z.restart_interval = z.get16();
		return true;
	}
	if(m == 0xdb) // DQT
	{
		int L = z.get16() - 2;
		while(L > 0)
		{
			const int q = z.get8(), prec = q >> 4, t = q & 15;
			if(prec > 1 || t > 3) return z.fail("JPEG: bad DQT");
			for(int i = 0; i < 64; ++i) z.dequant[t][kZigZag[i]] = (uint16_t)(prec ? z.get16() : z.get8());
			L -= prec ? 129 : 65;
		}
		return L == 0 || z.fail("JPEG: bad DQT length");
	}
	if(m == 0xc4) // DHT
	{
		int L = z.get16() - 2;
		while(L > 0)
		{
			const int q = z.get8(), tc = q >> 4, th = q & 15;
			if(tc > 1 || th > 3) return z.fail("JPEG: bad DHT header");
			int counts[16], n = 0;
			for(int i = 0; i < 16; ++i) { counts[i] = z.get8(); n += counts[i]; }
			if(n > 256) return z.fail("JPEG: bad DHT counts");
			Huffman &h = tc == 0 ? z.huff_dc[th] : z.huff_ac[th];
			if(!h.build(counts)) return z.fail("JPEG: bad code lengths");
			for(int i = 0; i < n; ++i) h.symbols[i] = (uint8_t)z.get8();
			L -= 17 + n;
		}
		return L == 0 || z.fail("JPEG: bad DHT length");
	}
	if((m >= 0xe0 && m <= 0xef) || m == 0xfe) // APPn / COM
	{
		int L = z.get16();
		if(L < 2) return z.fail("JPEG: bad APP / COM length");
		L -= 2;
		if(m == 0xe0 && L >= 5)
		{
			static const uint8_t tag[5] = {'J', 'F', 'I', 'F', 0};
			bool ok = true;
			for(int i = 0; i < 5; ++i) ok &= z.get8() == tag[i];
			L -= 5;
			if(ok) z.jfif = true;
		}
		else if(m == 0xee && L >= 12)
		{
			static const uint8_t tag[6] = {'A', 'd', 'o', 'b', 'e', 0};
			bool ok = true;
			for(int i = 0; i < 6; ++i) ok &= z.get8() == tag[i];
			L -= 6;
			if(ok)
			{
				z.get8(); z.get16(); z.get16();   // version, flags0, flags1
				z.adobe_transform = z.get8();
				L -= 6;
			}
		}
		if(z.end - z.p < L) return z.fail("JPEG: truncated segment");
		z.p += L;
		return true;
	}
	return z.fail(m < 0 ? "JPEG: expected a marker" : "JPEG: unknown marker");
}

bool process_frame_header(Decoder &z)
{
	const int Lf = z.get16();
	if(Lf < 11) return z.fail("JPEG: bad SOF length");
	if(z.get8() != 8) return z.fail("JPEG: only 8-bit samples are supported (as in stb_image)");
	z.height = z.get16(); z.width = z.get16();
	if(z.height == 0 || z.width == 0) return z.fail("JPEG: zero image dimension");
	z.ncomp = z.get8();
	if(z.ncomp != 1 && z.ncomp != 3 && z.ncomp != 4) return z.fail("JPEG: bad component count");
	if(Lf != 8 + 3 * z.ncomp) return z.fail("JPEG: bad SOF length");
	if((int64_t)z.width * z.height > ((int64_t)1 << 28)) return z.fail("JPEG: image too large");
	z.rgb_ids = 0;
	for(int i = 0; i < z.ncomp; ++i)
	{
		Component &c = z.comp[i];
		c.id = z.get8();
		if(z.ncomp == 3 && c.id == "RGB"[i]) ++z.rgb_ids;
		const int q = z.get8();
		c.h = q >> 4; c.v = q & 15; c.tq = z.get8();
		if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return z.fail("JPEG: bad sampling factors");
		z.hmax = std::max(z.hmax, c.h); z.vmax = std::max(z.vmax, c.v);
	}
	// a sampling factor that does not divide the maximum (e.g. 2x1, 3x1, 1x1) has no integer up-sampling step: the row filters
	// below would read W / (hmax / h) samples from a row that holds fewer (current stb_image rejects these as "bad H" / "bad V";
	// the copy vendored by the reference, dep/stb_image.h:2976-2977, does not and reads past its buffer)
	for(int i = 0; i < z.ncomp; ++i)
		if(z.hmax % z.comp[i].h != 0 || z.vmax % z.comp[i].v != 0) return z.fail("JPEG: sampling factors do not divide the maximum");
	z.mcu_x = (z.width + z.hmax * 8 - 1) / (z.hmax * 8);
	z.mcu_y = (z.height + z.vmax * 8 - 1) / (z.vmax * 8);
	for(int i = 0; i < z.ncomp; ++i)
	{
		Component &c = z.comp[i];
		c.x = (z.width * c.h + z.hmax - 1) / z.hmax;
		c.y = (z.height * c.v + z.vmax - 1) / z.vmax;
		c.w2 = z.mcu_x * c.h * 8; c.h2 = z.mcu_y * c.v * 8;
		c.data.assign((size_t)c.w2 * c.h2, 0);
		if(z.progressive) c.coeff.assign((size_t)c.w2 * c.h2, 0);
	}
	return true;
}

bool process_scan_header(Decoder &z)
{
	const int Ls = z.get16();
	z.scan_n = z.get8();
	if(z.scan_n < 1 || z.scan_n > 4 || z.scan_n > z.ncomp || Ls != 6 + 2 * z.scan_n) return z.fail("JPEG: bad SOS");
	for(int i = 0; i < z.scan_n; ++i)
	{
		const int id = z.get8(), q = z.get8();
		int which = 0;
		while(which < z.ncomp && z.comp[which].id != id) ++which;
		if(which == z.ncomp) return z.fail("JPEG: SOS names an unknown component");
		z.comp[which].hd = q >> 4; z.comp[which].ha = q & 15;
		if(z.comp[which].hd > 3 || z.comp[which].ha > 3) return z.fail("JPEG: bad Huffman table index");
		z.order[i] = which;
	}
	z.spec_start = z.get8(); z.spec_end = z.get8();
	const int aa = z.get8();
	z.succ_high = aa >> 4; z.succ_low = aa & 15;
	if(z.progressive)
	{
		if(z.spec_start > 63 || z.spec_end > 63 || z.spec_start > z.spec_end || z.succ_high > 13 || z.succ_low > 13) return z.fail("JPEG: bad SOS");
	}
	else
	{
		if(z.spec_start != 0 || z.succ_high != 0 || z.succ_low != 0) return z.fail("JPEG: bad SOS");
		z.spec_end = 63;
	}
	return true;
}

// one scan's entropy-coded data (stb_image.h:2637-2758): interleaved scans walk MCUs, single-component scans walk that
// component's own blocks; the restart interval counts MCUs / blocks and, at a RSTn, resets predictions, bit buffer and EOB run
bool parse_entropy_coded_data(Decoder &z)
{
	auto reset = [&]() {
		z.reset_entropy();
		for(Component &c : z.comp) c.dc_pred = 0;
		z.eob_run = 0;
	};
	reset();
	int todo = z.restart_interval ? z.restart_interval : 0x7fffffff;
	int16_t block[64];
	// returns false when the data simply stops at a non-restart marker (stb: "bail, so we get corrupt data rather than no data")
	auto counted = [&]() -> bool {
		if(--todo > 0) return true;
		if(z.bitcnt < 24) z.fill();
		if(!(z.marker >= 0xd0 && z.marker <= 0xd7)) return false;
		reset();
		todo = z.restart_interval ? z.restart_interval : 0x7fffffff;
		return true;
	};
	auto do_block = [&](Component &c, int bx, int by) -> bool {
		if(!z.progressive)
		{
			if(!decode_block_baseline(z, block, c)) return false;
			idct_block(c.data.data() + (size_t)c.w2 * by * 8 + (size_t)bx * 8, c.w2, block);
			return true;
		}
		int16_t *data = c.coeff.data() + 64 * ((size_t)bx + (size_t)by * (c.w2 / 8));
		return z.spec_start == 0 ? decode_block_prog_dc(z, data, c) : decode_block_prog_ac(z, data, c);
	};
	if(z.scan_n == 1)
	{
		Component &c = z.comp[z.order[0]];
		const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
		for(int j = 0; j < h; ++j)
			for(int i = 0; i < w; ++i)
			{
				if(!do_block(c, i, j)) return false;
				if(!counted()) return true;
			}
		return true;
	}
	if(z.progressive && z.spec_start != 0) return z.fail("JPEG: interleaved progressive AC scan");
	for(int j = 0; j < z.mcu_y; ++j)
		for(int i = 0; i < z.mcu_x; ++i)
		{
			for(int k = 0; k < z.scan_n; ++k)
			{
				Component &c = z.comp[z.order[k]];
				for(int y = 0; y < c.v; ++y)
					for(int x = 0; x < c.h; ++x)
						if(!do_block(c, i * c.h + x, j * c.v + y)) return false;
			}
			if(!counted()) return true;
		}
	return true;
}

int next_marker(Decoder &z)
{
	if(z.marker >= 0) { const int m = z.marker; z.marker = -1; return m; }
	int x = z.get8();
	if(x != 0xff) return -1;
	while(x == 0xff) x = z.get8();
	return x;
}

bool decode_to_components(Decoder &z)
{
	if(next_marker(z) != 0xd8) return z.fail("JPEG: no SOI");
	int m = next_marker(z);
	while(!(m == 0xc0 || m == 0xc1 || m == 0xc2))
	{
		if(m == 0xc3 || (m >= 0xc5 && m <= 0xcf && m != 0xc8 && m != 0xcc)) return z.fail("JPEG: lossless / arithmetic / hierarchical frames are not supported (as in stb_image)");
		if(!process_marker(z, m)) return false;
		m = next_marker(z);
		while(m < 0)
		{
			if(z.eof()) return z.fail("JPEG: no SOF");
			m = next_marker(z);
		}
	}
	z.progressive = m == 0xc2;
	if(!process_frame_header(z)) return false;
	m = next_marker(z);
	while(m != 0xd9)
	{
		if(m == 0xda)
		{
			if(!process_scan_header(z) || !parse_entropy_coded_data(z)) return false;
			if(z.marker < 0)
			{
				// zero padding after the scan data: look for the next 0xff
				while(!z.eof())
				{
					if(z.get8() == 255) { z.marker = z.get8(); break; }
				}
			}
		}
		else if(m == 0xdc) { z.get16(); z.get16(); } // DNL
		else if(!process_marker(z, m)) return false;
		m = next_marker(z);
		if(m < 0 && z.eof()) return z.fail("JPEG: no EOI");
	}
	if(z.progressive)
		for(int n = 0; n < z.ncomp; ++n)
		{
			Component &c = z.comp[n];
			const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
			for(int j = 0; j < h; ++j)
				for(int i = 0; i < w; ++i)
				{
					int16_t *data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * (c.w2 / 8));
					for(int k = 0; k < 64; ++k) data[k] = (int16_t)(data[k] * z.dequant[c.tq][k]);
					idct_block(c.data.data() + (size_t)c.w2 * j * 8 + (size_t)i * 8, c.w2, data);
				}
		}
	return true;
}

// chroma up-sampling, one output row at a time (stb_image.h:3110-3187, 3305-3316): centred 3:1 filters, replicated at the ends
const uint8_t *resample_row(int hs, int vs, uint8_t *out, const uint8_t *near_, const uint8_t *far_, int w)
{
	if(hs == 1 && vs == 1) return near_;
	if(hs == 1 && vs == 2)
	{
		for(int i = 0; i < w; ++i) out[i] = (uint8_t)((3 * near_[i] + far_[i] + 2) >> 2);
		return out;
	}
	if(hs == 2 && vs == 1)
	{
		const uint8_t *in = near_;
		if(w == 1) { out[0] = out[1] = in[0]; return out; }
		out[0] = in[0];
		out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
		int i = 1;
		for(; i < w - 1; ++i)
		{
			const int n = 3 * in[i] + 2;
			out[i * 2 + 0] = (uint8_t)((n + in[i - 1]) >> 2);
			out[i * 2 + 1] = (uint8_t)((n + in[i + 1]) >> 2);
		}
		out[i * 2 + 0] = (uint8_t)((in[w - 2] * 3 + in[w - 1] + 2) >> 2);
		out[i * 2 + 1] = in[w - 1];
		return out;
	}
	if(hs == 2 && vs == 2)
	{
		if(w == 1) { out[0] = out[1] = (uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return out; }
		int t1 = 3 * near_[0] + far_[0];
		out[0] = (uint8_t)((t1 + 2) >> 2);
		for(int i = 1; i < w; ++i)
		{
			const int t0 = t1;
			t1 = 3 * near_[i] + far_[i];
			out[i * 2 - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
			out[i * 2] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
		}
		out[w * 2 - 1] = (uint8_t)((t1 + 2) >> 2);
		return out;
	}
	for(int i = 0; i < w; ++i) // any other ratio: nearest neighbour, horizontally
		for(int j = 0; j < hs; ++j) out[i * hs + j] = near_[i];
	return out;
}

inline uint8_t blinn_8x8(uint8_t x, uint8_t y) { const unsigned t = (unsigned)x * y + 128; return (uint8_t)((t + (t >> 8)) >> 8); }

// stb_image.h:3317-3343: 20-bit fixed point; the blue-difference term of green is truncated to 16 bits before it is added
void ycbcr_to_rgb_row(uint8_t *out, const uint8_t *y, const uint8_t *pcb, const uint8_t *pcr, int count)
{
	auto f2f = [](float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; };
	const int kr = f2f(1.40200f), kg1 = f2f(0.71414f), kg2 = f2f(0.34414f), kb = f2f(1.77200f);
	for(int i = 0; i < count; ++i, out += 3)
	{
		const int yf = (y[i] << 20) + (1 << 19), cr = pcr[i] - 128, cb = pcb[i] - 128;
		int r = yf + cr * kr;
		int g = yf + (cr * -kg1) + (int)((unsigned)(cb * -kg2) & 0xffff0000u);
		int b = yf + cb * kb;
		r >>= 20; g >>= 20; b >>= 20;
		out[0] = clamp8(r); out[1] = clamp8(g); out[2] = clamp8(b);
	}
}

}  // namespace

bool decode_jpeg(const std::vector<uint8_t> &bytes, TextureImage *out, std::string *err)
{
	Decoder z;
	z.p = bytes.data(); z.end = bytes.data() + bytes.size();
	if(!decode_to_components(z)) { *err = z.err.empty() ? "JPEG: corrupt" : z.err; return false; }
	const int W = z.width, H = z.height, n = z.ncomp;
	const bool is_rgb = n == 3 && (z.rgb_ids == 3 || (z.adobe_transform == 0 && !z.jfif));
	out->w = W; out->h = H;
	out->rgb.assign((size_t)W * H * 3, 0);
	// row stepping of load_jpeg_image (stb_image.h:3560-3600): each component keeps a (line0, line1) pair and a phase
	struct Res { int hs, vs, ystep, ypos, w_lores; const uint8_t *line0, *line1; std::vector<uint8_t> buf; } res[4];
	for(int k = 0; k < n; ++k)
	{
		Res &r = res[k];
		r.hs = z.hmax / z.comp[k].h; r.vs = z.vmax / z.comp[k].v;
		r.ystep = r.vs >> 1; r.ypos = 0;
		r.w_lores = std::min((W + r.hs - 1) / r.hs, z.comp[k].w2); // never more samples than a decoded row holds
		r.line0 = r.line1 = z.comp[k].data.data();
		r.buf.assign((size_t)W + 3 + 8, 0);
	}
	const uint8_t *co[4] = {nullptr, nullptr, nullptr, nullptr};
	for(int j = 0; j < H; ++j)
	{
		uint8_t *o = out->rgb.data() + (size_t)j * W * 3;
		for(int k = 0; k < n; ++k)
		{
			Res &r = res[k];
			const bool y_bot = r.ystep >= (r.vs >> 1);
			co[k] = resample_row(r.hs, r.vs, r.buf.data(), y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores);
			if(++r.ystep >= r.vs)
			{
				r.ystep = 0;
				r.line0 = r.line1;
				if(++r.ypos < z.comp[k].y) r.line1 += z.comp[k].w2;
			}
		}
		if(n == 3)
		{
			if(is_rgb) for(int i = 0; i < W; ++i) { o[i * 3] = co[0][i]; o[i * 3 + 1] = co[1][i]; o[i * 3 + 2] = co[2][i]; }
			else ycbcr_to_rgb_row(o, co[0], co[1], co[2], W);
		}
		else if(n == 4)
		{
			if(z.adobe_transform == 0) // CMYK
				for(int i = 0; i < W; ++i)
				{
					const uint8_t m = co[3][i];
					o[i * 3] = blinn_8x8(co[0][i], m); o[i * 3 + 1] = blinn_8x8(co[1][i], m); o[i * 3 + 2] = blinn_8x8(co[2][i], m);
				}
			else
			{
				ycbcr_to_rgb_row(o, co[0], co[1], co[2], W);
				if(z.adobe_transform == 2) // YCCK
					for(int i = 0; i < W; ++i)
					{
						const uint8_t m = co[3][i];
						o[i * 3] = blinn_8x8((uint8_t)(255 - o[i * 3]), m); o[i * 3 + 1] = blinn_8x8((uint8_t)(255 - o[i * 3 + 1]), m); o[i * 3 + 2] = blinn_8x8((uint8_t)(255 - o[i * 3 + 2]), m);
					}
			}
		}
		else for(int i = 0; i < W; ++i) o[i * 3] = o[i * 3 + 1] = o[i * 3 + 2] = co[0][i];
	}
	return true;
}

}  // namespace adypt
