// Binary spatial-split BVH (Stich et al.) with exactly one triangle reference per leaf — the producer of the
// tree that wide_builder.cpp collapses into CWBVH8.
//
// This is a from-scratch restatement of the reference's builder decisions (src/BVH/SBVHBuilder.hpp:95-306,
// src/BVH/SBVHBuilder.cpp:8-71) so that the flattened node / index arrays are bit-identical to the ones Adypt's
// own CPU builder hands to its tracer (pinned by tests/golden/*.bvh, generated from the compiled reference).
// Structure differs: an explicit task stack instead of recursion (deep trees cannot overflow the call stack) and
// flat scratch arrays; every float expression keeps the reference's evaluation order (-ffp-contract=off).
//
// Contract reproduced from the reference:
//   * references live on one stack; a node owns the last `n` entries; the right child is built first and
//     lands at parent+1, the left child's index is stored (SBVH.hpp:28-29, SBVHBuilder.cpp:38-41)
//   * object split: sort by (centroid[axis], triangle id) for each axis, sweep, cost
//       nodeSAH*2*A + triSAH*(i*A_left + (n-i)*A_right), first strictly smaller wins (hpp:95-134)
//   * spatial split only if depth <= maxSpatialDepth and overlap area >= 1e-5 * scene area (cpp:20-26),
//     32 bins per axis with exact triangle clipping (hpp:136-228), unsplit-left/unsplit-right/duplicate
//     decision per straddling reference (hpp:230-306)
#include "builders.hpp"

#include <algorithm>
#include <chrono>

namespace adypt {
namespace {

constexpr int kBins = 32;

struct Ref { Box box; int32_t tri; };
struct Spec { Box box; int32_t n; };

struct ObjSplit { Box left, right; int dim = 0, left_n = 0; float sah = FLT_MAX; };
struct SpatSplit { int dim = 0; float pos = 0.0f, sah = FLT_MAX; };
struct Bin { Box box; int in = 0, out = 0; };

template <int D> bool ref_less(const Ref &l, const Ref &r)
{
	float lc = l.box.center()[D], rc = r.box.center()[D];
	return lc < rc || (lc == rc && l.tri < r.tri);
}

class Builder {
public:
	Builder(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg, std::vector<BinNode> *out)
		: tris_(tris), n_tris_(n_tris), scene_box_(scene_box), cfg_(cfg), nodes_(*out) {}

	int64_t run()
	{
		nodes_.clear();
		nodes_.reserve((size_t)n_tris_ * 2);
		refs_.reserve((size_t)n_tris_ * 2);
		refs_.resize((size_t)n_tris_);
		for(int64_t i = 0; i < n_tris_; ++i) { refs_[(size_t)i].tri = (int32_t)i; refs_[(size_t)i].box = tris_[i].bounds(); }
		min_overlap_ = scene_box_.area() * 1e-5f;

		struct Task { Spec spec; int depth; int32_t patch_parent; };
		std::vector<Task> todo;
		todo.push_back({{scene_box_, (int32_t)n_tris_}, 0, -1});
		int64_t leaves = 0;
		while(!todo.empty())
		{
			Task t = todo.back();
			todo.pop_back();
			int32_t node = (int32_t)nodes_.size();
			nodes_.emplace_back();
			nodes_[(size_t)node].box = t.spec.box;
			nodes_[(size_t)node].tri = 0;
			if(t.patch_parent >= 0) nodes_[(size_t)t.patch_parent].left = node;
			if(t.spec.n == 1)
			{
				nodes_[(size_t)node].left = -1;
				nodes_[(size_t)node].tri = refs_.back().tri;
				refs_.pop_back();
				++leaves;
				continue;
			}
			Spec left, right;
			split(t.spec, t.depth, &left, &right);
			// left is pushed first so that the whole right subtree is emitted before it (right child = node + 1)
			todo.push_back({left, t.depth + 1, node});
			todo.push_back({right, t.depth + 1, -1});
		}
		nodes_.shrink_to_fit();
		return leaves;
	}

private:
	const TriRec *tris_;
	int64_t n_tris_;
	Box scene_box_;
	adypt_bvh_params cfg_;
	std::vector<BinNode> &nodes_;
	std::vector<Ref> refs_;
	std::vector<Box> right_boxes_;
	Bin bins_[kBins];
	float min_overlap_ = 0.0f;

	float tri_cost(int count) const { return cfg_.triangle_sah * count; }
	float node_cost(int count) const { return cfg_.node_sah * count; }
	size_t first_ref(const Spec &s) const { return refs_.size() - (size_t)s.n; }

	void sort_refs(const Spec &s, int dim)
	{
		Ref *b = refs_.data() + first_ref(s), *e = refs_.data() + refs_.size();
		if(dim == 0) std::sort(b, e, ref_less<0>);
		else if(dim == 1) std::sort(b, e, ref_less<1>);
		else std::sort(b, e, ref_less<2>);
	}

	void object_split_axis(const Spec &s, int dim, float node_sah, ObjSplit *os)
	{
		sort_refs(s, dim);
		const Ref *r = refs_.data() + first_ref(s);
		const int n = s.n;
		right_boxes_.resize((size_t)n);
		right_boxes_[(size_t)n - 1] = r[n - 1].box;
		for(int i = n - 2; i >= 1; --i) right_boxes_[(size_t)i] = Box::join(r[i].box, right_boxes_[(size_t)i + 1]);
		Box left = r[0].box;
		for(int i = 1; i <= n - 1; ++i)
		{
			float sah = node_sah + tri_cost(i) * left.area() + tri_cost(n - i) * right_boxes_[(size_t)i].area();
			if(sah < os->sah)
			{
				os->dim = dim; os->left_n = i; os->left = left; os->right = right_boxes_[(size_t)i]; os->sah = sah;
			}
			left.grow(r[i].box);
		}
	}

	// clip one reference at plane x[dim] = pos into a left and a right piece (exact triangle clipping)
	void clip_ref(const Ref &ref, int dim, float pos, Ref *l, Ref *r) const
	{
		l->box = r->box = Box();
		l->tri = r->tri = ref.tri;
		const TriRec &t = tris_[ref.tri];
		for(int i = 0; i < 3; ++i)
		{
			const Vec3 &v0 = t.p[i], &v1 = t.p[(i + 1) % 3];
			float p0 = v0[dim], p1 = v1[dim];
			if(p0 <= pos) l->box.grow(v0);
			if(p0 >= pos) r->box.grow(v0);
			if((p0 < pos && pos < p1) || (p1 < pos && pos < p0))
			{
				float a = fmin_glm(fmax_glm((pos - p0) / (p1 - p0), 0.0f), 1.0f);
				Vec3 d = v1 - v0;
				Vec3 x = {v0.x + a * d.x, v0.y + a * d.y, v0.z + a * d.z};
				l->box.grow(x);
				r->box.grow(x);
			}
		}
		l->box.hi[dim] = pos;
		l->box.clip(ref.box);
		r->box.lo[dim] = pos;
		r->box.clip(ref.box);
	}

	static int clampi(int x, int lo, int hi) { int m = x < lo ? lo : x; return hi < m ? hi : m; }

	void spatial_split_axis(const Spec &s, int dim, float node_sah, SpatSplit *ss)
	{
		for(Bin &b : bins_) b = Bin();
		float bin_w = s.box.extent()[dim] / kBins, inv_w = 1.0f / bin_w;
		float base = s.box.lo[dim];
		const Ref *r = refs_.data() + first_ref(s);
		Ref cur, lp, rp;
		for(int i = 0; i < s.n; ++i)
		{
			// float -> int conversion as the x86 truncating convert (NaN / out of range give INT_MIN -> bin 0)
			int bin = clampi(trunc_to_int((r[i].box.lo[dim] - base) * inv_w), 0, kBins - 1);
			int last = clampi(trunc_to_int((r[i].box.hi[dim] - base) * inv_w), 0, kBins - 1);
			bins_[bin].in++;
			cur = r[i];
			for(; bin < last; ++bin)
			{
				clip_ref(cur, dim, (bin + 1) * bin_w + base, &lp, &rp);
				bins_[bin].box.grow(lp.box);
				cur = rp;
			}
			bins_[last].box.grow(cur.box);
			bins_[last].out++;
		}
		right_boxes_.resize(kBins);
		right_boxes_[kBins - 1] = bins_[kBins - 1].box;
		for(int i = kBins - 2; i >= 1; --i) right_boxes_[(size_t)i] = Box::join(bins_[i].box, right_boxes_[(size_t)i + 1]);
		Box left = bins_[0].box;
		int ln = 0, rn = s.n;
		for(int i = 1; i < kBins; ++i)
		{
			ln += bins_[i - 1].in;
			rn -= bins_[i - 1].out;
			float sah = node_sah + tri_cost(ln) * left.area() + tri_cost(rn) * right_boxes_[(size_t)i].area();
			if(sah < ss->sah) { ss->sah = sah; ss->dim = dim; ss->pos = base + i * bin_w; }
			left.grow(bins_[i].box);
		}
	}

	static int trunc_to_int(float f)
	{
		if(!(f > -2147483904.0f && f < 2147483648.0f)) return INT32_MIN; // cvttss2si "integer indefinite"
		return (int)f;
	}

	void do_spatial_split(const Spec &s, const SpatSplit &ss, Spec *left, Spec *right)
	{
		left->box = right->box = Box();
		const size_t base = first_ref(s);
		int lb = 0, le = 0, rb = s.n, re = s.n;
		for(int i = lb; i < rb; ++i)
		{
			if(refs_[base + i].box.hi[ss.dim] <= ss.pos)
			{
				left->box.grow(refs_[base + i].box);
				std::swap(refs_[base + i], refs_[base + (le++)]);
			}
			else if(refs_[base + i].box.lo[ss.dim] >= ss.pos)
			{
				right->box.grow(refs_[base + i].box);
				std::swap(refs_[base + i], refs_[base + (--rb)]);
				--i;
			}
		}
		Ref lp, rp;
		while(le < rb)
		{
			clip_ref(refs_[base + le], ss.dim, ss.pos, &lp, &rp);
			Box lub = left->box, ldb = left->box, rub = right->box, rdb = right->box;
			lub.grow(refs_[base + le].box);
			rub.grow(refs_[base + le].box);
			ldb.grow(lp.box);
			rdb.grow(rp.box);
			float lac = tri_cost(le - lb), rac = tri_cost(re - rb), lbc = tri_cost(1 + le - lb), rbc = tri_cost(1 + re - rb);
			float unsplit_l = lub.area() * lbc + right->box.area() * rac;
			float unsplit_r = left->box.area() * lac + rub.area() * rbc;
			float dup = ldb.area() * lbc + rdb.area() * rbc;
			if(unsplit_l < unsplit_r && unsplit_l < dup) { left->box = lub; ++le; }
			else if(unsplit_r < dup)
			{
				right->box = rub;
				std::swap(refs_[base + le], refs_[base + (--rb)]);
			}
			else
			{
				refs_.emplace_back();
				left->box = ldb;
				right->box = rdb;
				refs_[base + (le++)] = lp;
				refs_[base + (re++)] = rp;
			}
		}
		left->n = le - lb;
		right->n = re - rb;
	}

	void split(const Spec &s, int depth, Spec *left, Spec *right)
	{
		float node_sah = s.box.area() * node_cost(2);
		ObjSplit os;
		object_split_axis(s, 0, node_sah, &os);
		object_split_axis(s, 1, node_sah, &os);
		object_split_axis(s, 2, node_sah, &os);
		SpatSplit ss;
		if(depth <= cfg_.max_spatial_depth)
		{
			Box overlap = os.left;
			overlap.clip(os.right);
			if(overlap.area() >= min_overlap_)
			{
				spatial_split_axis(s, 0, node_sah, &ss);
				spatial_split_axis(s, 1, node_sah, &ss);
				spatial_split_axis(s, 2, node_sah, &ss);
			}
		}
		left->n = right->n = 0;
		if(ss.sah < os.sah) do_spatial_split(s, ss, left, right);
		if(left->n == 0 || right->n == 0)
		{
			sort_refs(s, os.dim);
			left->n = os.left_n; left->box = os.left;
			right->n = s.n - os.left_n; right->box = os.right;
		}
	}
};

}  // namespace

int64_t build_sbvh(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg,
				   std::vector<BinNode> *nodes, double *ms)
{
	auto t0 = std::chrono::steady_clock::now();
	int64_t leaves = 0;
	if(n_tris > 0) leaves = Builder(tris, n_tris, scene_box, cfg, nodes).run();
	if(ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return leaves;
}

}  // namespace adypt
