// Wavefront OBJ/MTL -> flat Triangle[] (100 B) + GPUMaterial[] (64 B) + decoded RGB8 textures.
//
// Stands in for Scene::LoadFromFile (src/Util/Scene.cpp:9-136, which delegates parsing to the vendored
// tinyobjloader 1.2.0) and for OglScene::init_materials / load_texture (src/Tracer/OglScene.cpp:12-91).
// The triangle order, vertex values and material ids must equal the reference's for the BVH arrays to match, so
// the parts of tinyobj's behaviour that influence them are reproduced:
//   * number parsing recipe (digit accumulation in double, fraction digits scaled by a 10^-k table, optional
//     exponent applied as ldexp(m*5^e, e)) — dep/tiny_obj_loader.h:525-638; tokens it rejects (e.g. ".5") read as 0
//   * index fix-up (1-based, negative = relative), i, i/j, i//k, i/j/k
//   * faces keep file order across g/o/usemtl; polygons with more than 3 corners are ear-clipped with tinyobj's
//     projection-axis / area-sign / point-in-triangle rules (dep/tiny_obj_loader.h:1043-1236)
//   * material defaults illum 0, Ns 1, Ni 1, d 1, colours 0; first definition of a name wins in the name map
// Scene-side rules from the reference (Scene.cpp:49-125): v of vt is flipped (1 - v); missing normal/uv stay 0;
// a flat normal is generated when the *last* corner has no normal; matid may be -1.
#include "common.hpp"
#include "../../../include/adypt_hip.h"
#include "../../../include/adypt_host.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>

namespace adypt {
namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }
inline bool is_newline(char c) { return c == '\r' || c == '\n' || c == '\0'; }

bool parse_number(const char *s, const char *end, double *result)
{
	if(s >= end) return false;
	double mantissa = 0.0;
	int exponent = 0;
	char sign = '+', exp_sign = '+';
	const char *c = s;
	int read = 0;
	if(*c == '+' || *c == '-') { sign = *c; ++c; }
	else if(!is_digit(*c)) return false;
	bool more = c != end;
	while(more && is_digit(*c)) { mantissa *= 10; mantissa += (int)(*c - '0'); ++c; ++read; more = c != end; }
	if(read == 0) return false;
	if(more)
	{
		bool has_exp = false;
		if(*c == '.')
		{
			++c; read = 1; more = c != end;
			static const double lut[] = {1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001};
			while(more && is_digit(*c))
			{
				mantissa += (int)(*c - '0') * (read < 8 ? lut[read] : std::pow(10.0, -read));
				++read; ++c; more = c != end;
			}
			has_exp = more;
		}
		else if(*c == 'e' || *c == 'E') has_exp = true;
		if(has_exp && (*c == 'e' || *c == 'E'))
		{
			++c; more = c != end;
			if(more && (*c == '+' || *c == '-')) { exp_sign = *c; ++c; }
			else if(!is_digit(*c)) return false;
			read = 0; more = c != end;
			while(more && is_digit(*c)) { exponent *= 10; exponent += (int)(*c - '0'); ++c; ++read; more = c != end; }
			exponent *= (exp_sign == '+' ? 1 : -1);
			if(read == 0) return false;
		}
	}
	*result = (sign == '+' ? 1 : -1) * (exponent ? std::ldexp(mantissa * std::pow(5.0, exponent), exponent) : mantissa);
	return true;
}

float parse_real(const char **tok, double def = 0.0)
{
	*tok += strspn(*tok, " \t");
	const char *end = *tok + strcspn(*tok, " \t\r");
	double v = def;
	parse_number(*tok, end, &v);
	*tok = end;
	return (float)v;
}
int parse_int(const char **tok)
{
	*tok += strspn(*tok, " \t");
	int i = atoi(*tok);
	*tok += strcspn(*tok, " \t\r");
	return i;
}

struct Corner { int v = -1, vn = -1, vt = -1; };

bool fix_index(int idx, int n, int *out)
{
	if(idx > 0) { *out = idx - 1; return true; }
	if(idx == 0) return false;
	*out = n + idx;
	return true;
}

bool parse_corner(const char **tok, int nv, int nvn, int nvt, Corner *out)
{
	Corner c;
	if(!fix_index(atoi(*tok), nv, &c.v)) return false;
	*tok += strcspn(*tok, "/ \t\r");
	if((*tok)[0] != '/') { *out = c; return true; }
	++*tok;
	if((*tok)[0] == '/')
	{
		++*tok;
		if(!fix_index(atoi(*tok), nvn, &c.vn)) return false;
		*tok += strcspn(*tok, "/ \t\r");
		*out = c;
		return true;
	}
	if(!fix_index(atoi(*tok), nvt, &c.vt)) return false;
	*tok += strcspn(*tok, "/ \t\r");
	if((*tok)[0] != '/') { *out = c; return true; }
	++*tok;
	if(!fix_index(atoi(*tok), nvn, &c.vn)) return false;
	*tok += strcspn(*tok, "/ \t\r");
	*out = c;
	return true;
}

struct ObjMaterial {
	std::string name, diffuse_tex;
	float diffuse[3] = {0, 0, 0}, specular[3] = {0, 0, 0}, emission[3] = {0, 0, 0};
	float shininess = 1.0f, ior = 1.0f, dissolve = 1.0f;
	int illum = 0;
};

std::string texture_name_from(const char *tok)
{
	// options (-blendu on, -o u v w, ...) precede the file name; the name is the remainder of the line
	static const struct { const char *opt; int nargs; } opts[] = {
		{"-blendu", 1}, {"-blendv", 1}, {"-clamp", 1}, {"-boost", 1}, {"-bm", 1}, {"-o", 3}, {"-s", 3}, {"-t", 3},
		{"-type", 1}, {"-imfchan", 1}, {"-mm", 2}, {"-texres", 1}};
	for(;;)
	{
		tok += strspn(tok, " \t");
		bool matched = false;
		for(const auto &o : opts)
		{
			size_t l = strlen(o.opt);
			if(strncmp(tok, o.opt, l) == 0 && is_space(tok[l]))
			{
				tok += l;
				for(int k = 0; k < o.nargs; ++k) { tok += strspn(tok, " \t"); tok += strcspn(tok, " \t\r"); }
				matched = true;
				break;
			}
		}
		if(!matched) break;
	}
	std::string s(tok);
	size_t e = s.find_last_not_of(" \t\r\n");
	return e == std::string::npos ? std::string() : s.substr(0, e + 1);
}

void load_mtl(std::istream &in, std::map<std::string, int> *name_map, std::vector<ObjMaterial> *mats)
{
	ObjMaterial cur;
	bool has_d = false;
	std::string line;
	while(std::getline(in, line))
	{
		if(!line.empty()) line = line.substr(0, line.find_last_not_of(" \t") + 1);
		while(!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
		if(line.empty()) continue;
		const char *t = line.c_str();
		t += strspn(t, " \t");
		if(t[0] == '\0' || t[0] == '#') continue;
		if(strncmp(t, "newmtl", 6) == 0 && is_space(t[6]))
		{
			if(!cur.name.empty()) { name_map->insert({cur.name, (int)mats->size()}); mats->push_back(cur); }
			cur = ObjMaterial();
			has_d = false;
			cur.name = t + 7;
			continue;
		}
		auto real3 = [&](float *dst) { for(int k = 0; k < 3; ++k) dst[k] = parse_real(&t); };
		if(t[0] == 'K' && t[1] == 'd' && is_space(t[2])) { t += 2; real3(cur.diffuse); continue; }
		if(t[0] == 'K' && t[1] == 's' && is_space(t[2])) { t += 2; real3(cur.specular); continue; }
		if(t[0] == 'K' && t[1] == 'e' && is_space(t[2])) { t += 2; real3(cur.emission); continue; }
		if(t[0] == 'N' && t[1] == 'i' && is_space(t[2])) { t += 2; cur.ior = parse_real(&t); continue; }
		if(t[0] == 'N' && t[1] == 's' && is_space(t[2])) { t += 2; cur.shininess = parse_real(&t); continue; }
		if(strncmp(t, "illum", 5) == 0 && is_space(t[5])) { t += 6; cur.illum = parse_int(&t); continue; }
		if(t[0] == 'd' && is_space(t[1])) { t += 1; cur.dissolve = parse_real(&t); has_d = true; continue; }
		if(t[0] == 'T' && t[1] == 'r' && is_space(t[2])) { t += 2; if(!has_d) cur.dissolve = 1.0f - parse_real(&t); continue; }
		if(strncmp(t, "map_Kd", 6) == 0 && is_space(t[6])) { cur.diffuse_tex = texture_name_from(t + 7); continue; }
	}
	name_map->insert({cur.name, (int)mats->size()});
	mats->push_back(cur);
}

// tinyobj's even-odd point-in-polygon test on 3 vertices
int pnpoly3(const float *vx, const float *vy, float tx, float ty)
{
	int c = 0;
	for(int i = 0, j = 2; i < 3; j = i++)
		if(((vy[i] > ty) != (vy[j] > ty)) && (tx < (vx[j] - vx[i]) * (ty - vy[i]) / (vy[j] - vy[i]) + vx[i])) c = !c;
	return c;
}

void triangulate(const std::vector<Corner> &face, const std::vector<float> &v, std::vector<Corner> *out)
{
	size_t np = face.size();
	if(np < 3) return;
	if(np == 3) { out->insert(out->end(), face.begin(), face.end()); return; }
	size_t axes[2] = {1, 2};
	for(size_t k = 0; k < np; ++k)
	{
		size_t a = (size_t)face[k % np].v, b = (size_t)face[(k + 1) % np].v, c = (size_t)face[(k + 2) % np].v;
		if(3 * a + 2 >= v.size() || 3 * b + 2 >= v.size() || 3 * c + 2 >= v.size()) continue;
		float e0x = v[b * 3] - v[a * 3], e0y = v[b * 3 + 1] - v[a * 3 + 1], e0z = v[b * 3 + 2] - v[a * 3 + 2];
		float e1x = v[c * 3] - v[b * 3], e1y = v[c * 3 + 1] - v[b * 3 + 1], e1z = v[c * 3 + 2] - v[b * 3 + 2];
		float cx = std::fabs(e0y * e1z - e0z * e1y), cy = std::fabs(e0z * e1x - e0x * e1z), cz = std::fabs(e0x * e1y - e0y * e1x);
		const float eps = FLT_EPSILON;
		if(cx > eps || cy > eps || cz > eps)
		{
			if(!(cx > cy && cx > cz)) { axes[0] = 0; if(cz > cx && cz > cy) axes[1] = 1; }
			break;
		}
	}
	float area = 0;
	for(size_t k = 0; k < np; ++k)
	{
		size_t a = (size_t)face[k].v, b = (size_t)face[(k + 1) % np].v;
		if(a * 3 + axes[0] >= v.size() || a * 3 + axes[1] >= v.size() || b * 3 + axes[0] >= v.size() || b * 3 + axes[1] >= v.size()) continue;
		area += (v[a * 3 + axes[0]] * v[b * 3 + axes[1]] - v[a * 3 + axes[1]] * v[b * 3 + axes[0]]) * 0.5f;
	}
	int rounds = 10;
	std::vector<Corner> rem = face;
	size_t guess = 0;
	Corner ind[3];
	float vx[3], vy[3];
	while(rem.size() > 3 && rounds > 0)
	{
		np = rem.size();
		if(guess >= np) { rounds -= 1; guess -= np; }
		for(size_t k = 0; k < 3; ++k)
		{
			ind[k] = rem[(guess + k) % np];
			size_t vi = (size_t)ind[k].v;
			if(vi * 3 + axes[0] >= v.size() || vi * 3 + axes[1] >= v.size()) { vx[k] = 0; vy[k] = 0; }
			else { vx[k] = v[vi * 3 + axes[0]]; vy[k] = v[vi * 3 + axes[1]]; }
		}
		float e0x = vx[1] - vx[0], e0y = vy[1] - vy[0], e1x = vx[2] - vx[1], e1y = vy[2] - vy[1];
		float cross = e0x * e1y - e0y * e1x;
		if(cross * area < 0.0f) { guess += 1; continue; }
		bool overlap = false;
		for(size_t other = 3; other < np; ++other)
		{
			size_t idx = (guess + other) % np;
			if(idx >= rem.size()) continue;
			size_t ovi = (size_t)rem[idx].v;
			if(ovi * 3 + axes[0] >= v.size() || ovi * 3 + axes[1] >= v.size()) continue;
			if(pnpoly3(vx, vy, v[ovi * 3 + axes[0]], v[ovi * 3 + axes[1]])) { overlap = true; break; }
		}
		if(overlap) { guess += 1; continue; }
		out->push_back(ind[0]); out->push_back(ind[1]); out->push_back(ind[2]);
		size_t removed = (guess + 1) % np;
		while(removed + 1 < np) { rem[removed] = rem[removed + 1]; removed += 1; }
		rem.pop_back();
	}
	if(rem.size() == 3) { out->push_back(rem[0]); out->push_back(rem[1]); out->push_back(rem[2]); }
}

}  // namespace

struct SceneData {
	std::vector<TriRec> tris;
	std::vector<MatRec> mats;
	std::vector<TextureImage> textures;
	std::vector<adypt_texture> tex_desc;
	Box box;
	std::string base_dir;
	std::string warnings; // one line per texture that could not be decoded (the material then renders as the reference's failed stbi_load: black)
};

static bool load_obj(const char *path, SceneData *sc, std::string *err)
{
	size_t len = strlen(path);
	if(len == 0) { *err = "empty scene filename"; return false; }
	{
		const char *s = path + len;
		while(s > path && *(s - 1) != '/' && *(s - 1) != '\\') --s;
		sc->base_dir.assign(path, s);
	}
	std::ifstream in(path);
	if(!in) { *err = std::string("cannot open ") + path; return false; }

	std::vector<float> v, vn, vt;
	std::vector<ObjMaterial> mats;
	std::map<std::string, int> mat_map;
	int material = -1;
	std::vector<Corner> face, tri_corners;
	std::string line;
	while(std::getline(in, line))
	{
		while(!line.empty() && (line.back() == '\n' || line.back() == '\r')) line.pop_back();
		if(line.empty()) continue;
		const char *t = line.c_str();
		t += strspn(t, " \t");
		if(t[0] == '\0' || t[0] == '#') continue;
		if(t[0] == 'v' && is_space(t[1]))
		{
			t += 2;
			for(int k = 0; k < 3; ++k) v.push_back(parse_real(&t));
			continue;
		}
		if(t[0] == 'v' && t[1] == 'n' && is_space(t[2]))
		{
			t += 3;
			for(int k = 0; k < 3; ++k) vn.push_back(parse_real(&t));
			continue;
		}
		if(t[0] == 'v' && t[1] == 't' && is_space(t[2]))
		{
			t += 3;
			for(int k = 0; k < 2; ++k) vt.push_back(parse_real(&t));
			continue;
		}
		if(t[0] == 'f' && is_space(t[1]))
		{
			t += 2;
			t += strspn(t, " \t");
			face.clear();
			while(!is_newline(t[0]))
			{
				Corner c;
				if(!parse_corner(&t, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2), &c))
				{
					*err = "failed to parse `f' line (zero face index)";
					return false;
				}
				face.push_back(c);
				t += strspn(t, " \t\r");
			}
			tri_corners.clear();
			triangulate(face, v, &tri_corners);
			for(size_t k = 0; k + 2 < tri_corners.size(); k += 3)
			{
				sc->tris.emplace_back();
				TriRec &tr = sc->tris.back();
				memset(&tr, 0, sizeof(tr));
				tr.matid = material;
				for(int c = 0; c < 3; ++c)
				{
					const Corner &cn = tri_corners[k + (size_t)c];
					// tinyobj stores whatever index the file gives and the reference's Scene.cpp then reads attrib.vertices[3 * idx]
					// unchecked (src/Util/Scene.cpp:60-75): an index outside the vertices defined so far is undefined behaviour there,
					// an error here
					if(cn.v < 0 || 3 * (size_t)cn.v + 2 >= v.size() || cn.vn < -1 || (cn.vn >= 0 && 3 * (size_t)cn.vn + 2 >= vn.size()) ||
					   cn.vt < -1 || (cn.vt >= 0 && 2 * (size_t)cn.vt + 1 >= vt.size()))
					{
						*err = "`f' line references a vertex / normal / texture coordinate that is not defined";
						return false;
					}
					tr.p[c] = {v[3 * (size_t)cn.v], v[3 * (size_t)cn.v + 1], v[3 * (size_t)cn.v + 2]};
					if(cn.vn != -1) tr.n[c] = {vn[3 * (size_t)cn.vn], vn[3 * (size_t)cn.vn + 1], vn[3 * (size_t)cn.vn + 2]};
					if(cn.vt != -1) { tr.tc[c][0] = vt[2 * (size_t)cn.vt]; tr.tc[c][1] = 1.0f - vt[2 * (size_t)cn.vt + 1]; }
				}
				if(tri_corners[k + 2].vn == -1)
				{
					// glm::normalize(glm::cross(p1 - p0, p2 - p0)), plain (un-fused) arithmetic
					Vec3 a = tr.p[1] - tr.p[0], b = tr.p[2] - tr.p[0];
					Vec3 c = {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
					float inv = 1.0f / std::sqrt(c.x * c.x + c.y * c.y + c.z * c.z);
					tr.n[0] = tr.n[1] = tr.n[2] = c * inv;
				}
				sc->box.grow(tr.bounds());
			}
			continue;
		}
		if(strncmp(t, "usemtl", 6) == 0 && is_space(t[6]))
		{
			std::string name(t + 7);
			auto it = mat_map.find(name);
			material = it == mat_map.end() ? -1 : it->second;
			continue;
		}
		if(strncmp(t, "mtllib", 6) == 0 && is_space(t[6]))
		{
			std::stringstream ss(std::string(t + 7));
			std::string fn;
			while(std::getline(ss, fn, ' '))
			{
				std::ifstream mi((sc->base_dir + fn).c_str());
				if(!mi) continue;
				load_mtl(mi, &mat_map, &mats);
				break;
			}
			continue;
		}
	}

	// OglScene::init_materials (src/Tracer/OglScene.cpp:51-82)
	std::map<std::string, int> tex_ids;
	for(const ObjMaterial &m : mats)
	{
		MatRec g;
		memset(&g, 0, sizeof(g));
		g.illum = 1;
		if(!m.diffuse_tex.empty())
		{
			std::string full = sc->base_dir + m.diffuse_tex;
			auto it = tex_ids.find(full);
			if(it != tex_ids.end()) g.dtex = it->second;
			else
			{
				TextureImage img;
				std::string ierr;
				if(decode_image_rgb8(full, &img, &ierr))
				{
					g.dtex = (int32_t)sc->textures.size();
					tex_ids[full] = g.dtex;
					sc->textures.push_back(std::move(img));
				}
				else
				{
					// the reference prints and carries on with m_dtex = -1 and Kd = 0 (src/Tracer/OglScene.cpp:12-43,62-66); so does this
					// loader, but the caller can ask what was lost (adypt_scene_warnings): formats stb_image reads and this decoder
					// does not (interlaced PNG, GIF, PSD, ...) would otherwise render differently without a trace
					fprintf(stderr, "[adypt] unable to load texture %s (%s)\n", full.c_str(), ierr.c_str());
					sc->warnings += "texture " + full + ": " + ierr + " — material '" + m.name + "' renders with Kd = 0\n";
					g.dtex = -1;
				}
			}
		}
		else { g.dtex = -1; g.dr = m.diffuse[0]; g.dg = m.diffuse[1]; g.db = m.diffuse[2]; }
		g.er = m.emission[0]; g.eg = m.emission[1]; g.eb = m.emission[2];
		g.sr = m.specular[0]; g.sg = m.specular[1]; g.sb = m.specular[2];
		g.illum = m.illum; g.shininess = m.shininess; g.dissolve = m.dissolve; g.ior = m.ior;
		sc->mats.push_back(g);
	}
	for(const TextureImage &t : sc->textures) sc->tex_desc.push_back(adypt_texture{t.w, t.h, t.rgb.data()});
	return true;
}

}  // namespace adypt

using namespace adypt;

struct adypt_scene { SceneData d; };

extern "C" {

int adypt_scene_load(const char *obj_path, adypt_scene **out)
{
	if(!obj_path || !out) { set_host_error("adypt_scene_load: null argument"); return ADYPT_E_INVALID; }
	adypt_scene *s = new adypt_scene();
	std::string err;
	if(!load_obj(obj_path, &s->d, &err)) { delete s; set_host_error(err); return ADYPT_E_IO; }
	*out = s;
	return ADYPT_OK;
}

int adypt_scene_from_arrays(const void *tris, int64_t n_tris, const void *mats, int64_t n_mats, adypt_scene **out)
{
	if(!tris || n_tris < 0 || n_mats < 0 || !out) { set_host_error("adypt_scene_from_arrays: bad argument"); return ADYPT_E_INVALID; }
	adypt_scene *s = new adypt_scene();
	s->d.tris.assign((const TriRec *)tris, (const TriRec *)tris + n_tris);
	if(mats) s->d.mats.assign((const MatRec *)mats, (const MatRec *)mats + n_mats);
	for(const TriRec &t : s->d.tris) s->d.box.grow(t.bounds());
	*out = s;
	return ADYPT_OK;
}

void adypt_scene_free(adypt_scene *s) { delete s; }
int64_t adypt_scene_triangles(const adypt_scene *s, const void **tris) { if(tris) *tris = s->d.tris.data(); return (int64_t)s->d.tris.size(); }
int64_t adypt_scene_materials(const adypt_scene *s, const void **mats) { if(mats) *mats = s->d.mats.data(); return (int64_t)s->d.mats.size(); }
int32_t adypt_scene_textures(const adypt_scene *s, const void **tex) { if(tex) *tex = s->d.tex_desc.data(); return (int32_t)s->d.tex_desc.size(); }
const char *adypt_scene_warnings(const adypt_scene *s) { return s ? s->d.warnings.c_str() : ""; }
void adypt_scene_aabb(const adypt_scene *s, float lo[3], float hi[3])
{
	for(int k = 0; k < 3; ++k) { lo[k] = s->d.box.lo[k]; hi[k] = s->d.box.hi[k]; }
}

}  // extern "C"
