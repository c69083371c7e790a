// Canonical arithmetic of the tracer (device side).
//
// GLSL leaves the rounding of normalize / dot / 1/x / pow / sin / cos and fma contraction to the driver, so the
// build pins one arithmetic and implements it twice, independently: here for gfx950 and in oracle/oracle.cpp for
// the CPU checker.  The rules (DESIGN.md "Canonical arithmetic"):
//   * binary32 IEEE-754 round-to-nearest-even + - * / sqrt in the written order (hipcc: -ffp-contract=off,
//     correctly rounded fp32 divide/sqrt are the HIP default);
//   * fused multiply-add only where fmaf()/fma() is written;
//   * dot3(a,b) = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x)); normalize(v) = v * (1/sqrt(dot3(v,v)));
//   * sin/cos/pow = fixed binary64 series below, rounded once to binary32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// a four-component register "defined" by an empty asm — no instruction — for values that are only read where a conditional load has written them
#define ADYPT_DEF4(v) asm volatile("" : "=v"((v).x), "=v"((v).y), "=v"((v).z), "=v"((v).w))

namespace adypt {

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ F3 fma3(F3 a, float s, F3 c) { return f3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }
__device__ __forceinline__ F3 fma3(F3 a, F3 b, F3 c) { return f3(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)); }
// two binary32 values in adjacent registers: +, * and pk_fma map to v_pk_add/mul/fma_f32 — each half is the same IEEE
// operation as the scalar instruction, so packing never changes a result
typedef float V2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ V2 v2s(float s) { V2 r; r.x = s; r.y = s; return r; }
__device__ __forceinline__ V2 pk_fma(V2 a, V2 b, V2 c) { return __builtin_elementwise_fma(a, b, c); }
// fma(a, s, c) per half with s = the HIGH half of the register pair `b` (a loaded float4's .y or .w): v_pk_fma_f32 reads it in place through op_sel.
// Written out because the compiler forms this broadcast with a v_mov_b32 into a fresh pair (3 per Woop test).  Same IEEE fma per half as pk_fma.
__device__ __forceinline__ V2 pk_fma_hi(V2 a, V2 b, V2 c)
{
	V2 r;
	asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
	return r;
}
// IEEE maxNum / minNum of two values that are never signalling NaNs (slab distances, tmin, the hit distance): the instruction itself.  Written out
// because the compiler, not knowing that, first quiets a loop-carried or kernel-argument operand with a v_max_f32 x, x of its own (two per slab test).
__device__ __forceinline__ float max_num(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min_num(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// byte J of x, moved to bits 23 .. 30: the biased exponent byte of a CWBVH8 node as a float's exponent field (= 2^(e - 127)), one SDWA shift
template <int J> __device__ __forceinline__ float exp_byte(uint32_t x)
{
	uint32_t r;
	if(J == 0) asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(x));
	else if(J == 1) asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(x));
	else asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(x));
	return __uint_as_float(r);
}
// (x byte J) << (s byte J) for two words of four packed bytes, one SDWA instruction (the shift uses the low 5 bits of the selected byte)
template <int J> __device__ __forceinline__ uint32_t shl_bytes(uint32_t s, uint32_t x)
{
	uint32_t r;
	if(J == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(s), "v"(x));
	else if(J == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(s), "v"(x));
	else if(J == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(s), "v"(x));
	else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(s), "v"(x));
	return r;
}
__device__ __forceinline__ float dot3(F3 a, F3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b)
{
	return f3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
// 1 / x, correctly rounded (= the IEEE quotient, bit for bit): v_rcp_f32 and one Newton step for |x| in [2^-126, 2^126) — exhaustively verified
// against the division sequence over all 2^32 operands (tools/microbench/rcp_exact.hip: the only mismatches have biased exponent 0, 253, 254, 255) —
// and the division sequence itself for the others (and NaN)
__device__ __forceinline__ float rcp_ieee(float x)
{
	const float ax = __builtin_fabsf(x);
	if(__builtin_expect(ax >= 0x1p-126f && ax < 0x1p126f, 1))
	{
		const float r = __builtin_amdgcn_rcpf(x);
		const float e = __builtin_fmaf(-x, r, 1.0f);
		return __builtin_fmaf(e, r, r);
	}
	return 1.0f / x;
}
__device__ __forceinline__ F3 normalize3(F3 a) { float inv = rcp_ieee(sqrtf(dot3(a, a))); return a * inv; }
__device__ __forceinline__ F3 reflect3(F3 i, F3 n) { float k = 2.0f * dot3(n, i); return fma3(n, -k, i); }
// (the helpers below that rewrite EXEC or use SDWA encodings — or_if_le, exp_byte, shl_bytes, traverse_trip.inc's acceptance steps — are GFX9-family wave64 code: v_cmpx writes
// VCC and EXEC, a lane mask is 64 bits, SDWA exists.  The Makefile's ARCH / HIPFLAGS overrides cannot turn this file into something else silently.)
#if defined(__HIP_DEVICE_COMPILE__) && !(defined(__gfx950__) || defined(__gfx942__) || defined(__gfx90a__))
#error "canon_math.hpp: the EXEC-writing and SDWA helpers are written for GFX9-family (CDNA) wave64 targets"
#endif
// acc |= bits in the lanes where a <= b: the compare writes the lane mask straight into EXEC, the OR runs under it, EXEC comes back — 4.3 + 2.5 issue cycles where
// v_cmp + v_cndmask + (half a) v_or3 take 10.8; eight of them per node visit
__device__ __forceinline__ void or_if_le(uint32_t &acc, float a, float b, uint32_t bits)
{
	unsigned long long ex;
	asm volatile("s_mov_b64 %1, exec\n\tv_cmpx_le_f32_e32 vcc, %2, %3\n\tv_or_b32_e32 %0, %0, %4\n\ts_mov_b64 exec, %1" : "+v"(acc), "=&s"(ex) : "v"(a), "v"(b), "v"(bits) : "vcc"); // (EXEC is back before the statement ends; as a clobber the compiler rejects it: a reserved register)
}
// GLSL min/max NaN rule
__device__ __forceinline__ float gl_min(float x, float y) { return y < x ? y : x; }
__device__ __forceinline__ float gl_max(float x, float y) { return x < y ? y : x; }

// One Horner step p * x + c of the fp64 series with the coefficient in a scalar register pair: written in C the compiler
// materialises every coefficient with two v_mov_b32 and uses the two-address v_fmac_f64 — ~90 VALU slots per path in the
// VALU-bound k_shade for nothing.  v_fma_f64 is the IEEE fused multiply-add either way, so results do not change.
__device__ __forceinline__ double horner(double p, double x, double c)
{
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c));
	return r;
}

__device__ inline void canon_sincos(float xf, float *s_out, float *c_out)
{
	const double TWO_OVER_PI = 0.63661977236758134308;
	const double PIO2_HI = 1.57079632679489655800e+00;
	const double PIO2_LO = 6.12323399573676603587e-17;
	double x = (double)xf;
	double kd = rint(x * TWO_OVER_PI);
	double y = fma(-kd, PIO2_HI, x);
	y = fma(-kd, PIO2_LO, y);
	double y2 = y * y;
	double ps = -1.0 / 121645100408832000.0;
	ps = horner(ps, y2, 1.0 / 355687428096000.0);
	ps = horner(ps, y2, -1.0 / 1307674368000.0);
	ps = horner(ps, y2, 1.0 / 6227020800.0);
	ps = horner(ps, y2, -1.0 / 39916800.0);
	ps = horner(ps, y2, 1.0 / 362880.0);
	ps = horner(ps, y2, -1.0 / 5040.0);
	ps = horner(ps, y2, 1.0 / 120.0);
	ps = horner(ps, y2, -1.0 / 6.0);
	double sn = fma(y * y2, ps, y);
	double pc = 1.0 / 2432902008176640000.0;
	pc = horner(pc, y2, -1.0 / 6402373705728000.0);
	pc = horner(pc, y2, 1.0 / 20922789888000.0);
	pc = horner(pc, y2, -1.0 / 87178291200.0);
	pc = horner(pc, y2, 1.0 / 479001600.0);
	pc = horner(pc, y2, -1.0 / 3628800.0);
	pc = horner(pc, y2, 1.0 / 40320.0);
	pc = horner(pc, y2, -1.0 / 720.0);
	pc = horner(pc, y2, 1.0 / 24.0);
	pc = horner(pc, y2, -0.5);
	double cs = fma(y2, pc, 1.0);
	long long k = (long long)kd;
	double s, c;
	switch(k & 3)
	{
		case 0: s = sn; c = cs; break;
		case 1: s = cs; c = -sn; break;
		case 2: s = -sn; c = -cs; break;
		default: s = -cs; c = sn; break;
	}
	*s_out = (float)s;
	*c_out = (float)c;
}

// c / 255 for an 8-bit c (RG8 / RGB8 UNORM decoding): the correctly rounded binary32 quotient without the ~10-instruction IEEE
// division sequence and without a table.  q = c * rn(1/255) is off by at most one ulp; the remainder c - 255 q is exact in an fma,
// and one correction step lands on rn(c / 255) for every c in 0..255 (checked with exact rational arithmetic over all 256 inputs,
// tests/test_oracle_golden.py::test_unorm8_decode_formula; the oracle divides).  Round 3: the 256-entry table this replaces cost a
// dependent vector-memory instruction per channel — 14 per textured path — on the pipeline k_shade is bound by.
__device__ __forceinline__ float unorm8_to_float(uint32_t c)
{
	const float x = (float)(c & 0xffu), r = 0x1.010102p-8f; // rn(1 / 255)
	const float q = x * r;
	return fmaf(fmaf(-q, 255.0f, x), r, q);
}

__device__ inline float canon_pow(float xf, float yf)
{
	if(yf == 0.0f) return 1.0f;
	// x^1 = x: the series below returns exactly x for every positive finite x (its double result is within 1e-15 of x,
	// far inside the binary32 rounding interval; checked over the whole range by tests/test_oracle_golden.py), so the
	// diffuse lobe's pow(1 - r.y, 1 / (0 + 1)) (pathtracer.glsl:58) costs nothing
	if(yf == 1.0f && xf > 0.0f && xf < __uint_as_float(0x7f800000u)) return xf;
	if(xf != xf || yf != yf) return xf + yf;
	if(xf < 0.0f) return __uint_as_float(0x7fc00000u);
	if(xf == 0.0f) return yf > 0.0f ? 0.0f : __uint_as_float(0x7f800000u);
	if(xf == __uint_as_float(0x7f800000u)) return yf > 0.0f ? xf : 0.0f;
	double x = (double)xf;
	unsigned long long bits = (unsigned long long)__double_as_longlong(x);
	int e = (int)((bits >> 52) & 0x7ff) - 1023;
	bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
	double m = __longlong_as_double((long long)bits);
	if(m > 1.41421356237309514547) { m *= 0.5; e += 1; }
	double s = (m - 1.0) / (m + 1.0);
	double s2 = s * s;
	double p = 1.0 / 23.0;
	p = horner(p, s2, 1.0 / 21.0);
	p = horner(p, s2, 1.0 / 19.0);
	p = horner(p, s2, 1.0 / 17.0);
	p = horner(p, s2, 1.0 / 15.0);
	p = horner(p, s2, 1.0 / 13.0);
	p = horner(p, s2, 1.0 / 11.0);
	p = horner(p, s2, 1.0 / 9.0);
	p = horner(p, s2, 1.0 / 7.0);
	p = horner(p, s2, 1.0 / 5.0);
	p = horner(p, s2, 1.0 / 3.0);
	double ln_m = 2.0 * fma(s * s2, p, s);
	const double LOG2E = 1.44269504088896338700;
	double log2x = fma(ln_m, LOG2E, (double)e);
	double t = (double)yf * log2x;
	if(t >= 129.0) return __uint_as_float(0x7f800000u);
	if(t <= -151.0) return 0.0f;
	double n = rint(t);
	double f = t - n;
	const double LN2 = 0.69314718055994528623;
	double z = f * LN2;
	double q = 1.0 / 6227020800.0;
	q = horner(q, z, 1.0 / 479001600.0);
	q = horner(q, z, 1.0 / 39916800.0);
	q = horner(q, z, 1.0 / 3628800.0);
	q = horner(q, z, 1.0 / 362880.0);
	q = horner(q, z, 1.0 / 40320.0);
	q = horner(q, z, 1.0 / 5040.0);
	q = horner(q, z, 1.0 / 720.0);
	q = horner(q, z, 1.0 / 120.0);
	q = horner(q, z, 1.0 / 24.0);
	q = horner(q, z, 1.0 / 6.0);
	q = horner(q, z, 0.5);
	q = horner(q, z, 1.0);
	q = horner(q, z, 1.0);
	unsigned long long sb = (unsigned long long)((long long)n + 1023) << 52;
	double scale = __longlong_as_double((long long)sb);
	return (float)(q * scale);
}

}  // namespace adypt
