// Adypt `.config` scene interface: strict JSON reader and pretty writer for the InstanceConfig schema
// (reference: src/InstanceConfig.hpp:12-48, src/InstanceConfig.cpp:10-192; the reference delegates to rapidjson).
//
// Reader rules reproduced (SURVEY.md §5 "Config / flags"): every key is required; integer fields must be JSON
// integers in [0, 2^32) with no fraction/exponent (rapidjson IsUint); float fields must be written with a fraction
// or exponent (rapidjson IsFloat: `"clamp": 4` is rejected) and fit binary32; "sun" / "position" have exactly 3
// entries; the root must be an object followed by nothing but whitespace.
// Writer rules reproduced: rapidjson PrettyWriter defaults (4-space indent, one array element per line, no
// trailing newline), doubles printed with the shortest round-trip digits laid out by rapidjson's Prettify rules
// ("12.0", "0.0001", "1e-7", "1.5e21").
#include "common.hpp"
#include "../../../include/adypt_hip.h"
#include "../../../include/adypt_host.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

namespace adypt {

static thread_local std::string g_host_error;
void set_host_error(const std::string &msg) { g_host_error = msg; }

namespace {

struct JValue {
	enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
	bool b = false;
	bool is_integer = false;   // no '.', 'e', 'E' in the literal
	bool negative = false;
	uint64_t magnitude = 0;    // valid when is_integer && !int_overflow
	bool int_overflow = false;
	double d = 0.0;
	std::string s;
	std::vector<JValue> arr;
	std::vector<std::pair<std::string, JValue>> obj;

	const JValue *find(const char *key) const
	{
		for(const auto &kv : obj) if(kv.first == key) return &kv.second;
		return nullptr;
	}
	bool is_uint() const { return kind == Number && is_integer && !negative && !int_overflow && magnitude <= 0xffffffffull; }
	bool is_float() const
	{
		if(kind != Number) return false;
		if(is_integer && !int_overflow) return false; // carries an integer type flag, not the double flag
		return d >= -3.4028234e38 && d <= 3.4028234e38;
	}
};

class Parser {
public:
	explicit Parser(const char *text) : p_(text) {}
	bool parse_document(JValue *out, std::string *err)
	{
		skip_ws();
		if(!parse_value(out, err)) return false;
		skip_ws();
		if(*p_ != '\0') { *err = "trailing characters after the JSON root"; return false; }
		return true;
	}

private:
	const char *p_;
	void skip_ws() { while(*p_ == ' ' || *p_ == '\n' || *p_ == '\r' || *p_ == '\t') ++p_; }

	bool parse_value(JValue *v, std::string *err)
	{
		switch(*p_)
		{
			case 'n': return literal("null", v, JValue::Null, false, err);
			case 't': return literal("true", v, JValue::Bool, true, err);
			case 'f': return literal("false", v, JValue::Bool, false, err);
			case '"': v->kind = JValue::String; return parse_string(&v->s, err);
			case '[': return parse_array(v, err);
			case '{': return parse_object(v, err);
			default: return parse_number(v, err);
		}
	}
	bool literal(const char *word, JValue *v, JValue::Kind k, bool b, std::string *err)
	{
		size_t n = strlen(word);
		if(strncmp(p_, word, n) != 0) { *err = "invalid value"; return false; }
		p_ += n; v->kind = k; v->b = b;
		return true;
	}
	bool parse_number(JValue *v, std::string *err)
	{
		const char *start = p_;
		bool neg = false;
		if(*p_ == '-') { neg = true; ++p_; }
		if(*p_ == '0') ++p_;
		else if(*p_ >= '1' && *p_ <= '9') while(*p_ >= '0' && *p_ <= '9') ++p_;
		else { *err = "invalid value"; return false; }
		bool integer = true;
		if(*p_ == '.')
		{
			integer = false; ++p_;
			if(!(*p_ >= '0' && *p_ <= '9')) { *err = "missing fraction digits"; return false; }
			while(*p_ >= '0' && *p_ <= '9') ++p_;
		}
		if(*p_ == 'e' || *p_ == 'E')
		{
			integer = false; ++p_;
			if(*p_ == '+' || *p_ == '-') ++p_;
			if(!(*p_ >= '0' && *p_ <= '9')) { *err = "missing exponent digits"; return false; }
			while(*p_ >= '0' && *p_ <= '9') ++p_;
		}
		v->kind = JValue::Number; v->is_integer = integer; v->negative = neg;
		std::string lit(start, p_);
		v->d = strtod(lit.c_str(), nullptr);
		if(integer)
		{
			uint64_t m = 0; bool ovf = false;
			for(const char *c = start + (neg ? 1 : 0); c < p_; ++c)
			{
				uint64_t dgt = (uint64_t)(*c - '0');
				if(m > (UINT64_MAX - dgt) / 10) { ovf = true; break; }
				m = m * 10 + dgt;
			}
			if(neg && m > (uint64_t)INT64_MAX + 1) ovf = true;
			v->magnitude = m; v->int_overflow = ovf;
		}
		return true;
	}
	static void put_utf8(std::string *s, unsigned cp)
	{
		if(cp < 0x80) s->push_back((char)cp);
		else if(cp < 0x800) { s->push_back((char)(0xc0 | cp >> 6)); s->push_back((char)(0x80 | (cp & 0x3f))); }
		else if(cp < 0x10000) { s->push_back((char)(0xe0 | cp >> 12)); s->push_back((char)(0x80 | (cp >> 6 & 0x3f))); s->push_back((char)(0x80 | (cp & 0x3f))); }
		else { s->push_back((char)(0xf0 | cp >> 18)); s->push_back((char)(0x80 | (cp >> 12 & 0x3f))); s->push_back((char)(0x80 | (cp >> 6 & 0x3f))); s->push_back((char)(0x80 | (cp & 0x3f))); }
	}
	bool hex4(unsigned *out)
	{
		unsigned v = 0;
		for(int i = 0; i < 4; ++i)
		{
			char c = *p_++;
			v <<= 4;
			if(c >= '0' && c <= '9') v |= (unsigned)(c - '0');
			else if(c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
			else if(c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
			else return false;
		}
		*out = v;
		return true;
	}
	bool parse_string(std::string *s, std::string *err)
	{
		++p_;
		s->clear();
		for(;;)
		{
			unsigned char c = (unsigned char)*p_;
			if(c == '"') { ++p_; return true; }
			if(c == '\0' || c < 0x20) { *err = "invalid string"; return false; }
			if(c == '\\')
			{
				++p_;
				char e = *p_++;
				switch(e)
				{
					case '"': s->push_back('"'); break;
					case '\\': s->push_back('\\'); break;
					case '/': s->push_back('/'); break;
					case 'b': s->push_back('\b'); break;
					case 'f': s->push_back('\f'); break;
					case 'n': s->push_back('\n'); break;
					case 'r': s->push_back('\r'); break;
					case 't': s->push_back('\t'); break;
					case 'u': {
						unsigned cp;
						if(!hex4(&cp)) { *err = "invalid \\u escape"; return false; }
						if(cp >= 0xd800 && cp <= 0xdbff)
						{
							unsigned lo;
							if(p_[0] != '\\' || p_[1] != 'u') { *err = "invalid surrogate"; return false; }
							p_ += 2;
							if(!hex4(&lo) || lo < 0xdc00 || lo > 0xdfff) { *err = "invalid surrogate"; return false; }
							cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00);
						}
						put_utf8(s, cp);
						break;
					}
					default: *err = "invalid escape"; return false;
				}
				continue;
			}
			s->push_back((char)c);
			++p_;
		}
	}
	bool parse_array(JValue *v, std::string *err)
	{
		++p_;
		v->kind = JValue::Array;
		skip_ws();
		if(*p_ == ']') { ++p_; return true; }
		for(;;)
		{
			v->arr.emplace_back();
			skip_ws();
			if(!parse_value(&v->arr.back(), err)) return false;
			skip_ws();
			if(*p_ == ',') { ++p_; continue; }
			if(*p_ == ']') { ++p_; return true; }
			*err = "missing comma or ']' in array";
			return false;
		}
	}
	bool parse_object(JValue *v, std::string *err)
	{
		++p_;
		v->kind = JValue::Object;
		skip_ws();
		if(*p_ == '}') { ++p_; return true; }
		for(;;)
		{
			skip_ws();
			if(*p_ != '"') { *err = "missing object member name"; return false; }
			std::string key;
			if(!parse_string(&key, err)) return false;
			skip_ws();
			if(*p_ != ':') { *err = "missing ':' after member name"; return false; }
			++p_;
			skip_ws();
			v->obj.emplace_back(key, JValue());
			if(!parse_value(&v->obj.back().second, err)) return false;
			skip_ws();
			if(*p_ == ',') { ++p_; continue; }
			if(*p_ == '}') { ++p_; return true; }
			*err = "missing comma or '}' in object";
			return false;
		}
	}
};

// ---- typed extraction ---------------------------------------------------------------------------------------
struct Reader {
	std::string err;
	const JValue *member(const JValue &o, const char *key, const char *where)
	{
		const JValue *m = o.find(key);
		if(!m) err = std::string("missing \"") + key + "\" in " + where;
		return m;
	}
	bool get_uint(const JValue &o, const char *key, const char *where, int32_t *out)
	{
		const JValue *m = member(o, key, where);
		if(!m) return false;
		if(!m->is_uint()) { err = std::string("undefined Uint \"") + key + "\" in " + where; return false; }
		*out = (int32_t)(uint32_t)m->magnitude;
		return true;
	}
	bool float_of(const JValue &m, const std::string &what, float *out)
	{
		if(!m.is_float()) { err = "undefined Float " + what; return false; }
		*out = (float)m.d;
		return true;
	}
	bool get_float(const JValue &o, const char *key, const char *where, float *out)
	{
		const JValue *m = member(o, key, where);
		return m && float_of(*m, std::string("\"") + key + "\" in " + where, out);
	}
	bool get_string(const JValue &o, const char *key, const char *where, char *out, size_t cap)
	{
		const JValue *m = member(o, key, where);
		if(!m) return false;
		if(m->kind != JValue::String) { err = std::string("undefined String \"") + key + "\" in " + where; return false; }
		if(m->s.size() + 1 > cap) { err = std::string("string too long: ") + key; return false; }
		memcpy(out, m->s.c_str(), m->s.size() + 1);
		return true;
	}
	const JValue *get_object(const JValue &o, const char *key, const char *where)
	{
		const JValue *m = member(o, key, where);
		if(m && m->kind != JValue::Object) { err = std::string("undefined Object \"") + key + "\" in " + where; return nullptr; }
		return m;
	}
	bool get_float3(const JValue &o, const char *key, const char *where, float out[3])
	{
		const JValue *m = member(o, key, where);
		if(!m) return false;
		if(m->kind != JValue::Array) { err = std::string("undefined Array \"") + key + "\" in " + where; return false; }
		if(m->arr.size() != 3) { err = std::string("size of \"") + key + "\" array is not 3"; return false; }
		for(int i = 0; i < 3; ++i)
			if(!float_of(m->arr[(size_t)i], std::string(key) + "[" + std::to_string(i) + "]", &out[i])) return false;
		return true;
	}
};

// ---- writer ---------------------------------------------------------------------------------------------------
void write_string(std::string *o, const char *s)
{
	static const char hex[] = "0123456789ABCDEF";
	o->push_back('"');
	for(const unsigned char *p = (const unsigned char *)s; *p; ++p)
	{
		unsigned char c = *p;
		switch(c)
		{
			case '"': *o += "\\\""; break;
			case '\\': *o += "\\\\"; break;
			case '\b': *o += "\\b"; break;
			case '\f': *o += "\\f"; break;
			case '\n': *o += "\\n"; break;
			case '\r': *o += "\\r"; break;
			case '\t': *o += "\\t"; break;
			default:
				if(c < 0x20) { *o += "\\u00"; o->push_back(hex[c >> 4]); o->push_back(hex[c & 15]); }
				else o->push_back((char)c);
		}
	}
	o->push_back('"');
}

void write_exponent(std::string *o, int k)
{
	if(k < 0) { o->push_back('-'); k = -k; }
	*o += std::to_string(k);
}

// ---- Grisu2 (Loitsch, PLDI 2010) — the double -> shortest-ish decimal conversion behind rapidjson's Writer::Double.
// Reproduced (not just "shortest round trip") because Grisu2 occasionally emits a different final digit than the
// closest shortest representation (e.g. float 0.3 -> 0.30000001192092898), and GetJson() text is pinned by golden files.
struct DiyFp {
	uint64_t f; int e;
	DiyFp() : f(0), e(0) {}
	DiyFp(uint64_t f_, int e_) : f(f_), e(e_) {}
	explicit DiyFp(double d)
	{
		uint64_t u; memcpy(&u, &d, 8);
		const int biased = (int)((u >> 52) & 0x7ff);
		const uint64_t sig = u & 0x000fffffffffffffull;
		if(biased != 0) { f = sig + 0x0010000000000000ull; e = biased - 0x433; }
		else { f = sig; e = 1 - 0x433; }
	}
	DiyFp operator-(const DiyFp &r) const { return DiyFp(f - r.f, e); }
	DiyFp operator*(const DiyFp &r) const
	{
		const unsigned __int128 p = (unsigned __int128)f * r.f;
		uint64_t h = (uint64_t)(p >> 64);
		if((uint64_t)p & (1ull << 63)) ++h; // round half up
		return DiyFp(h, e + r.e + 64);
	}
	DiyFp normalize() const { int s = __builtin_clzll(f); return DiyFp(f << s, e - s); }
	DiyFp normalize_boundary() const
	{
		DiyFp r = *this;
		while(!(r.f & (0x0010000000000000ull << 1))) { r.f <<= 1; r.e--; }
		r.f <<= (64 - 52 - 2); r.e -= (64 - 52 - 2);
		return r;
	}
	void boundaries(DiyFp *minus, DiyFp *plus) const
	{
		DiyFp pl = DiyFp((f << 1) + 1, e - 1).normalize_boundary();
		DiyFp mi = (f == 0x0010000000000000ull) ? DiyFp((f << 2) - 1, e - 2) : DiyFp((f << 1) - 1, e - 1);
		mi.f <<= mi.e - pl.e;
		mi.e = pl.e;
		*plus = pl; *minus = mi;
	}
};

const struct { uint64_t f; int e; } kPow10Cache[87] = {
#include "grisu_pow10.inc"
};

DiyFp cached_power(int e, int *K)
{
	const double dk = (-61 - e) * 0.30102999566398114 + 347;
	int k = (int)dk;
	if(dk - k > 0.0) ++k;
	const unsigned index = (unsigned)((k >> 3) + 1);
	*K = -(-348 + (int)(index << 3));
	return DiyFp(kPow10Cache[index].f, kPow10Cache[index].e);
}

void grisu_round(char *buf, int len, uint64_t delta, uint64_t rest, uint64_t ten_kappa, uint64_t wp_w)
{
	while(rest < wp_w && delta - rest >= ten_kappa && (rest + ten_kappa < wp_w || wp_w - rest > rest + ten_kappa - wp_w))
	{
		buf[len - 1]--;
		rest += ten_kappa;
	}
}

int decimal_digits32(uint32_t n)
{
	int digits = 1;
	for(uint32_t lim = 10; digits < 9 && n >= lim; lim *= 10) ++digits;
	return digits;
}

void digit_gen(const DiyFp &W, const DiyFp &Mp, uint64_t delta, char *buf, int *len, int *K)
{
	static const uint32_t kPow10[] = {1, 10, 100, 1000, 10000, 100000, 1000000, 10000000, 100000000, 1000000000};
	const DiyFp one((uint64_t)1 << -Mp.e, Mp.e);
	const DiyFp wp_w = Mp - W;
	uint32_t p1 = (uint32_t)(Mp.f >> -one.e);
	uint64_t p2 = Mp.f & (one.f - 1);
	int kappa = decimal_digits32(p1);
	*len = 0;
	while(kappa > 0)
	{
		const uint32_t div = kPow10[kappa - 1];
		const uint32_t d = p1 / div;
		p1 %= div;
		if(d || *len) buf[(*len)++] = (char)('0' + d);
		--kappa;
		const uint64_t tmp = ((uint64_t)p1 << -one.e) + p2;
		if(tmp <= delta)
		{
			*K += kappa;
			grisu_round(buf, *len, delta, tmp, (uint64_t)kPow10[kappa] << -one.e, wp_w.f);
			return;
		}
	}
	for(;;)
	{
		p2 *= 10;
		delta *= 10;
		const char d = (char)(p2 >> -one.e);
		if(d || *len) buf[(*len)++] = (char)('0' + d);
		p2 &= one.f - 1;
		--kappa;
		if(p2 < delta)
		{
			*K += kappa;
			const int index = -kappa;
			grisu_round(buf, *len, delta, p2, one.f, wp_w.f * (index < 9 ? kPow10[index] : 0));
			return;
		}
	}
}

void grisu2(double value, char *buf, int *len, int *K)
{
	const DiyFp v(value);
	DiyFp w_m, w_p;
	v.boundaries(&w_m, &w_p);
	const DiyFp c_mk = cached_power(w_p.e, K);
	const DiyFp W = v.normalize() * c_mk;
	DiyFp Wp = w_p * c_mk, Wm = w_m * c_mk;
	Wm.f++;
	Wp.f--;
	digit_gen(W, Wp, Wp.f - Wm.f, buf, len, K);
}

void write_double(std::string *o, double d)
{
	if(d == 0.0) { *o += std::signbit(d) ? "-0.0" : "0.0"; return; }
	if(d < 0) { o->push_back('-'); d = -d; }
	char buf[32];
	int length = 0, k = 0; // value = digits * 10^k
	grisu2(d, buf, &length, &k);
	const std::string digits(buf, (size_t)length);
	const int kk = length + k; // 10^(kk-1) <= value < 10^kk
	// rapidjson Prettify with the default maxDecimalPlaces (324)
	if(0 <= k && kk <= 21) { *o += digits; o->append((size_t)k, '0'); *o += ".0"; }
	else if(0 < kk && kk <= 21) { o->append(digits, 0, (size_t)kk); o->push_back('.'); o->append(digits, (size_t)kk, std::string::npos); }
	else if(-6 < kk && kk <= 0) { *o += "0."; o->append((size_t)(-kk), '0'); *o += digits; }
	else if(kk < -324) *o += "0.0";
	else if(length == 1) { *o += digits; o->push_back('e'); write_exponent(o, kk - 1); }
	else { o->push_back(digits[0]); o->push_back('.'); o->append(digits, 1, std::string::npos); o->push_back('e'); write_exponent(o, kk - 1); }
}

struct Pretty {
	std::string out;
	int level = 0;
	std::vector<int> counts; // values written at each open level
	std::vector<bool> is_obj;
	void indent() { out.append((size_t)level * 4, ' '); }
	void prefix(bool is_key_or_array_elem_value)
	{
		(void)is_key_or_array_elem_value;
		if(counts.empty()) return;
		int &n = counts.back();
		if(is_obj.back())
		{
			if(n % 2 == 0) { if(n > 0) out.push_back(','); out.push_back('\n'); indent(); }
			else out += ": ";
		}
		else { if(n > 0) out.push_back(','); out.push_back('\n'); indent(); }
		++n;
	}
	void key(const char *k) { prefix(true); write_string(&out, k); }
	void val_int(int v) { prefix(false); out += std::to_string(v); }
	void val_double(double v) { prefix(false); write_double(&out, v); }
	void val_string(const char *s) { prefix(false); write_string(&out, s); }
	void open(bool object) { prefix(false); out.push_back(object ? '{' : '['); counts.push_back(0); is_obj.push_back(object); ++level; }
	void close()
	{
		bool object = is_obj.back();
		bool empty = counts.back() == 0;
		counts.pop_back(); is_obj.pop_back(); --level;
		if(!empty) { out.push_back('\n'); indent(); }
		out.push_back(object ? '}' : ']');
	}
};

}  // namespace
}  // namespace adypt

using namespace adypt;

extern "C" {

const char *adypt_host_last_error(void) { return g_host_error.c_str(); }

void adypt_config_default(adypt_config *c)
{
	memset(c, 0, sizeof(*c));
	c->width = 1280; c->height = 720;
	c->bvh.max_spatial_depth = 48; c->bvh.triangle_sah = 0.3f; c->bvh.node_sah = 1.0f;
	c->invocation_size = 8; c->stack_size = 12; c->max_bounce = 5; c->subpixel = 8; c->tmp_lifetime = 16;
	c->ray_tmin = 0.0001f; c->clamp = 4.0f;
	c->speed = 1.0f; c->mouse_sensitive = 0.3f; c->fov = 45.0f;
}

int adypt_config_parse(const char *text, adypt_config *c)
{
	if(!text || !c) { set_host_error("adypt_config_parse: null argument"); return ADYPT_E_INVALID; }
	JValue root;
	std::string err;
	if(!Parser(text).parse_document(&root, &err)) { set_host_error("[PARSER]ERR: Failed to parse json: " + err); return ADYPT_E_PARSE; }
	if(root.kind != JValue::Object) { set_host_error("[PARSER]ERR: Failed to parse json: root is not an object"); return ADYPT_E_PARSE; }
	adypt_config tmp = *c;
	Reader r;
	bool ok = r.get_uint(root, "width", "document", &tmp.width) && r.get_uint(root, "height", "document", &tmp.height);
	const JValue *o;
	ok = ok && (o = r.get_object(root, "scene", "document")) && r.get_string(*o, "filename", "scene", tmp.obj_filename, sizeof(tmp.obj_filename));
	ok = ok && (o = r.get_object(root, "pathTracer", "document"))
		 && r.get_uint(*o, "invocationSize", "pathTracer", &tmp.invocation_size) && r.get_uint(*o, "stackSize", "pathTracer", &tmp.stack_size)
		 && r.get_uint(*o, "maxBounce", "pathTracer", &tmp.max_bounce) && r.get_uint(*o, "subpixel", "pathTracer", &tmp.subpixel)
		 && r.get_uint(*o, "tmpLifetime", "pathTracer", &tmp.tmp_lifetime) && r.get_float(*o, "rayTMin", "pathTracer", &tmp.ray_tmin)
		 && r.get_float(*o, "clamp", "pathTracer", &tmp.clamp) && r.get_float3(*o, "sun", "pathTracer", tmp.sun);
	ok = ok && (o = r.get_object(root, "bvh", "document")) && r.get_string(*o, "filename", "bvh", tmp.bvh_filename, sizeof(tmp.bvh_filename))
		 && r.get_uint(*o, "maxSpatialDepth", "bvh", &tmp.bvh.max_spatial_depth) && r.get_float(*o, "triangleSAH", "bvh", &tmp.bvh.triangle_sah)
		 && r.get_float(*o, "nodeSAH", "bvh", &tmp.bvh.node_sah);
	ok = ok && (o = r.get_object(root, "camera", "document")) && r.get_float(*o, "speed", "camera", &tmp.speed)
		 && r.get_float(*o, "mouseSensitive", "camera", &tmp.mouse_sensitive) && r.get_float(*o, "fov", "camera", &tmp.fov)
		 && r.get_float(*o, "yaw", "camera", &tmp.yaw) && r.get_float(*o, "pitch", "camera", &tmp.pitch)
		 && r.get_float3(*o, "position", "camera", tmp.position);
	if(!ok) { set_host_error("[PARSER]ERR: " + r.err); return ADYPT_E_PARSE; }
	*c = tmp;
	return ADYPT_OK;
}

int adypt_config_load(const char *path, adypt_config *c)
{
	if(!path || !c) { set_host_error("adypt_config_load: null argument"); return ADYPT_E_INVALID; }
	std::ifstream in(path);
	if(!in.is_open()) { set_host_error(std::string("cannot open ") + path); return ADYPT_E_IO; }
	std::stringstream ss;
	ss << in.rdbuf();
	return adypt_config_parse(ss.str().c_str(), c);
}

size_t adypt_config_json(const adypt_config *c, char *buf, size_t cap)
{
	Pretty w;
	w.open(true);
	w.key("width"); w.val_int(c->width);
	w.key("height"); w.val_int(c->height);
	w.key("scene"); w.open(true);
	w.key("filename"); w.val_string(c->obj_filename);
	w.close();
	w.key("pathTracer"); w.open(true);
	w.key("invocationSize"); w.val_int(c->invocation_size);
	w.key("stackSize"); w.val_int(c->stack_size);
	w.key("maxBounce"); w.val_int(c->max_bounce);
	w.key("subpixel"); w.val_int(c->subpixel);
	w.key("tmpLifetime"); w.val_int(c->tmp_lifetime);
	w.key("rayTMin"); w.val_double(c->ray_tmin);
	w.key("clamp"); w.val_double(c->clamp);
	w.key("sun"); w.open(false);
	for(int i = 0; i < 3; ++i) w.val_double(c->sun[i]);
	w.close();
	w.close();
	w.key("bvh"); w.open(true);
	w.key("filename"); w.val_string(c->bvh_filename);
	w.key("maxSpatialDepth"); w.val_int(c->bvh.max_spatial_depth);
	w.key("triangleSAH"); w.val_double(c->bvh.triangle_sah);
	w.key("nodeSAH"); w.val_double(c->bvh.node_sah);
	w.close();
	w.key("camera"); w.open(true);
	w.key("speed"); w.val_double(c->speed);
	w.key("mouseSensitive"); w.val_double(c->mouse_sensitive);
	w.key("fov"); w.val_double(c->fov);
	w.key("yaw"); w.val_double(c->yaw);
	w.key("pitch"); w.val_double(c->pitch);
	w.key("position"); w.open(false);
	for(int i = 0; i < 3; ++i) w.val_double(c->position[i]);
	w.close();
	w.close();
	w.close();
	size_t need = w.out.size() + 1;
	if(buf && cap)
	{
		size_t n = need <= cap ? need - 1 : cap - 1;
		memcpy(buf, w.out.data(), n);
		buf[n] = '\0';
	}
	return need;
}

int adypt_config_save(const char *path, const adypt_config *c)
{
	std::vector<char> buf(adypt_config_json(c, nullptr, 0));
	adypt_config_json(c, buf.data(), buf.size());
	std::ofstream out(path);
	if(!out.is_open()) { set_host_error(std::string("cannot write ") + path); return ADYPT_E_IO; }
	out << buf.data();
	return ADYPT_OK;
}

}  // extern "C"
