// Binary spatial-split BVH (Stich et al.) with exactly one triangle reference per leaf — the producer of the
// tree that wide_builder.cpp collapses into CWBVH8.
//
// north_star reuses the reference's CPU builder unchanged; this file restates it so that the product needs none of the
// reference's sources, and the flattened node / index arrays stay bit-identical to the ones Adypt's own builder hands to
// its tracer (pinned by tests/golden/*.bvh, generated from the compiled reference).  Provenance, said plainly: the split
// search and the reference splitting (object_split_axis, clip_ref, spatial_split_axis, do_spatial_split below) were
// written with src/BVH/SBVHBuilder.hpp:95-306 open beside them and follow it statement by statement — bit-identical
// output forces the same decision sequence and the same float evaluation order (-ffp-contract=off), so that third of the
// file is a restatement, not a re-design.  What is this repo's own: the explicit task stack instead of recursion (deep
// trees cannot overflow the call stack), the flat scratch arrays, and the parallel build with its stitching (below) and
// exact_sort.hpp.
//
// Parallel build.  What a subtree looks like depends only on (its box, its depth, the ORDER of its
// references on the stack): the reference builder never touches stack entries below the node it works on.  So the
// top of the tree is cut into tasks — a task owns a private copy of its references in stack order, performs either
// ONE split (large nodes: its two children become tasks) or the whole sequential build of its subtree (small nodes)
// — the tasks run on a thread pool, and the per-task node arrays are stitched together afterwards in the reference's
// emission order (node, right subtree, left subtree) with the `left` indices rebased.  Every split sees exactly the
// bytes it would have seen in the sequential build (same std::sort calls on the same sequences), so the node array is
// bit-identical for any thread count (pinned by tests/test_host_golden.py against the compiled reference).
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
#include "exact_sort.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

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

inline std::vector<Box> &sweep_scratch()
{
	static thread_local std::vector<Box> scratch;
	return scratch;
}

class Builder {
public:
	Builder(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg, std::vector<BinNode> *out)
		: tris_(tris), n_tris_(n_tris), scene_box_(scene_box), cfg_(cfg), nodes_(*out) {}

	// the references of the whole scene, in triangle order (SBVHBuilder.cpp:52-61)
	void init_scene_refs()
	{
		refs_.reserve((size_t)n_tris_ * 2);
		refs_.resize((size_t)n_tris_);
		for(int64_t i = 0; i < n_tris_; ++i) { refs_[(size_t)i].tri = (int32_t)i; refs_[(size_t)i].box = tris_[i].bounds(); }
	}
	std::vector<Ref> &refs() { return refs_; }
	void set_sort_threads(int threads, int64_t min_task) { sort_threads_ = threads; sort_min_task_ = min_task; }

	int64_t run()
	{
		init_scene_refs();
		return run_subtree({scene_box_, (int32_t)n_tris_}, 0);
	}

	// sequential build of the subtree whose references are the last `root.n` entries of refs(); node indices are
	// local to `out` (root = 0)
	int64_t run_subtree(const Spec &root, int root_depth)
	{
		nodes_.clear();
		nodes_.reserve((size_t)root.n * 2);
		min_overlap_ = scene_box_.area() * 1e-5f;

		struct Task { Spec spec; int depth; int32_t patch_parent; };
		std::vector<Task> todo;
		todo.push_back({root, root_depth, -1});
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

	// one split of the node that owns all of refs(): afterwards refs() = [left->n references | right->n references]
	void split_once(const Spec &s, int depth, Spec *left, Spec *right)
	{
		min_overlap_ = scene_box_.area() * 1e-5f;
		split(s, depth, left, right);
	}

private:
	const TriRec *tris_;
	int64_t n_tris_;
	Box scene_box_;
	adypt_bvh_params cfg_;
	std::vector<BinNode> &nodes_;
	std::vector<Ref> refs_;
	std::vector<Box> &right_boxes_ = sweep_scratch(); // per thread, kept across tasks: fresh pages are expensive to touch
	Bin bins_[kBins];
	float min_overlap_ = 0.0f;
	int sort_threads_ = 1;
	int64_t sort_min_task_ = 1 << 15;

	float tri_cost(int count) const { return cfg_.triangle_sah * count; }
	float node_cost(int count) const { return cfg_.node_sah * count; }
	size_t first_ref(const Spec &s) const { return refs_.size() - (size_t)s.n; }

	void sort_refs(const Spec &s, int dim)
	{
		Ref *b = refs_.data() + first_ref(s), *e = refs_.data() + refs_.size();
		bool (*less)(const Ref &, const Ref &) = dim == 0 ? ref_less<0> : dim == 1 ? ref_less<1> : ref_less<2>;
		// large nodes of the task-parallel build: the same permutation as std::sort, computed on several threads
		if(sort_threads_ > 1 && e - b >= 2 * sort_min_task_) { ExactSort<Ref, decltype(less)>(less, sort_threads_, sort_min_task_).sort(b, e); return; }
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
#ifdef ADYPT_BUILD_TIMING
		auto T0 = std::chrono::steady_clock::now();
		auto lap = [&](const char *what) { if(s.n > 500000) { auto T1 = std::chrono::steady_clock::now(); fprintf(stderr, "[split n=%d d=%d thr=%d] %s %.2f s\n", s.n, depth, sort_threads_, what, std::chrono::duration<double>(T1 - T0).count()); T0 = T1; } };
#else
		auto lap = [](const char *) {};
#endif
		float node_sah = s.box.area() * node_cost(2);
		ObjSplit os;
		object_split_axis(s, 0, node_sah, &os);
		object_split_axis(s, 1, node_sah, &os);
		object_split_axis(s, 2, node_sah, &os);
		lap("object split (3 sorts + sweeps)");
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
		lap("spatial bins");
		left->n = right->n = 0;
		if(ss.sah < os.sah) do_spatial_split(s, ss, left, right);
		lap("do_spatial_split");
		if(left->n == 0 || right->n == 0)
		{
			sort_refs(s, os.dim);
			lap("final sort");
			left->n = os.left_n; left->box = os.left;
			right->n = s.n - os.left_n; right->box = os.right;
		}
	}
};

// ---------------------------------------------------------------------------------------------------------------
// task-parallel driver (see the header comment)
// ---------------------------------------------------------------------------------------------------------------
constexpr int32_t kSequentialRefs = 8192; // nodes with at most this many references are built whole by one task
                                          // ($ADYPT_BUILD_GRAIN overrides: the tests cut the tiny reference fixtures too)
int32_t sequential_refs()
{
	if(const char *ev = getenv("ADYPT_BUILD_GRAIN")) { int v = atoi(ev); if(v >= 1) return v; }
	return kSequentialRefs;
}
int64_t sort_min_task()  // smallest range the parallel sort hands to another thread ($ADYPT_BUILD_SORT_GRAIN: tests)
{
	if(const char *ev = getenv("ADYPT_BUILD_SORT_GRAIN")) { int v = atoi(ev); if(v >= 1) return v; }
	return 1 << 15;
}

struct SubTask {
	Spec spec;
	int depth = 0;
	std::vector<Ref> refs;        // input: the node's references in stack order (released once consumed)
	std::vector<BinNode> nodes;   // result of a whole-subtree task (local indices)
	SubTask *right = nullptr, *left = nullptr; // result of a one-split task
	int64_t leaves = 0;
};

class ParallelBuild {
public:
	ParallelBuild(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg, int n_threads)
		: tris_(tris), n_tris_(n_tris), scene_box_(scene_box), cfg_(cfg), n_threads_(n_threads), grain_(sequential_refs()), sort_min_task_(sort_min_task()) {}

	int64_t run(std::vector<BinNode> *out)
	{
		SubTask *root = new_task();
		root->spec = {scene_box_, (int32_t)n_tris_};
		{
			std::vector<BinNode> dummy;
			Builder b(tris_, n_tris_, scene_box_, cfg_, &dummy);
			b.init_scene_refs();
			root->refs = std::move(b.refs());
		}
		push(root);
		std::vector<std::thread> pool;
		for(int i = 1; i < n_threads_; ++i) pool.emplace_back([this] { work(); });
		work();
		for(std::thread &t : pool) t.join();
		return stitch(root, out);
	}

private:
	const TriRec *tris_;
	int64_t n_tris_;
	Box scene_box_;
	adypt_bvh_params cfg_;
	int n_threads_;
	int32_t grain_;
	int64_t sort_min_task_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<SubTask *> ready_;
	std::vector<std::unique_ptr<SubTask>> all_;
	int64_t unfinished_ = 0;

	// Reference buffers are recycled between tasks: a finished task's vector (capacity intact) serves a later right
	// child.  First-touching fresh pages costs more than the copy itself on the virtualised hosts this runs on.
	std::vector<std::vector<Ref>> buffers_;
	std::vector<Ref> acquire(size_t n)
	{
		std::vector<Ref> v;
		{
			std::lock_guard<std::mutex> g(mu_);
			size_t best = buffers_.size();
			for(size_t i = 0; i < buffers_.size(); ++i)
				if(buffers_[i].capacity() >= n && (best == buffers_.size() || buffers_[i].capacity() < buffers_[best].capacity())) best = i;
			if(best != buffers_.size()) { v = std::move(buffers_[best]); buffers_[best] = std::move(buffers_.back()); buffers_.pop_back(); }
		}
		v.clear();
		return v;
	}
	void release(std::vector<Ref> &&v)
	{
		if(v.capacity() < 1024) return;
		v.clear();
		std::lock_guard<std::mutex> g(mu_);
		buffers_.push_back(std::move(v));
	}

	SubTask *new_task()
	{
		std::lock_guard<std::mutex> g(mu_);
		all_.emplace_back(new SubTask());
		return all_.back().get();
	}
	void push(SubTask *t)
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			ready_.push_back(t);
			++unfinished_;
		}
		cv_.notify_one();
	}

	void work()
	{
		for(;;)
		{
			SubTask *t;
			{
				std::unique_lock<std::mutex> g(mu_);
				cv_.wait(g, [this] { return !ready_.empty() || unfinished_ == 0; });
				if(ready_.empty()) return;
				// largest pending nodes first: they spawn the work the other threads are waiting for
				t = ready_.front();
				ready_.pop_front();
			}
			execute(t);
			bool all_done;
			{
				std::lock_guard<std::mutex> g(mu_);
				all_done = --unfinished_ == 0;
			}
			if(all_done) cv_.notify_all();
		}
	}

	void execute(SubTask *t)
	{
		Builder b(tris_, n_tris_, scene_box_, cfg_, &t->nodes);
		b.refs() = std::move(t->refs);
		if(t->spec.n <= grain_)
		{
			t->leaves = b.run_subtree(t->spec, t->depth);
			release(std::move(b.refs()));
			return;
		}
		// the few large nodes at the top of the tree are the critical path: their sorts get a share of the threads
		// proportional to the node's share of the scene (the nodes of one level together own about all references)
		b.set_sort_threads((int)std::max<int64_t>(1, std::min<int64_t>(n_threads_, ((int64_t)n_threads_ * t->spec.n + n_tris_ / 2) / n_tris_)), sort_min_task_);
		Spec ls, rs;
		b.split_once(t->spec, t->depth, &ls, &rs);
		std::vector<Ref> &r = b.refs(); // = [left | right] (spatial splits may have added references)
		SubTask *right = new_task(), *left = new_task();
		right->spec = rs; right->depth = t->depth + 1;
		right->refs = acquire((size_t)rs.n + (size_t)rs.n / 4);
		right->refs.reserve((size_t)rs.n + (size_t)rs.n / 4);
		right->refs.assign(r.end() - rs.n, r.end());
		left->spec = ls; left->depth = t->depth + 1;
		// the left child owns what is on top of the stack once the right subtree is done; it keeps the parent's buffer
		if(r.size() != (size_t)ls.n + (size_t)rs.n) r.erase(r.begin(), r.end() - rs.n - ls.n);
		r.resize((size_t)ls.n);
		left->refs = std::move(r);
		t->right = right; t->left = left;
		push(right);
		push(left);
	}

	// emission order of the reference: node, its whole right subtree (right child = node + 1), then the left subtree
	int64_t stitch(SubTask *root, std::vector<BinNode> *out)
	{
		size_t total = 0;
		int64_t leaves = 0;
		for(const auto &t : all_) { total += t->right ? 1 : t->nodes.size(); leaves += t->leaves; }
		out->clear();
		out->reserve(total);
		struct Item { SubTask *t; int32_t patch_parent; };
		std::vector<Item> todo;
		todo.push_back({root, -1});
		while(!todo.empty())
		{
			Item it = todo.back();
			todo.pop_back();
			const int32_t at = (int32_t)out->size();
			if(it.patch_parent >= 0) (*out)[(size_t)it.patch_parent].left = at;
			if(it.t->right)
			{
				BinNode n;
				n.box = it.t->spec.box; n.tri = 0; n.left = -1;
				out->push_back(n);
				todo.push_back({it.t->left, at});
				todo.push_back({it.t->right, -1});
			}
			else
			{
				for(BinNode n : it.t->nodes)
				{
					if(n.left != -1) n.left += at;
					out->push_back(n);
				}
				std::vector<BinNode>().swap(it.t->nodes);
			}
		}
		return leaves;
	}
};

}  // namespace

int64_t build_sbvh(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg,
				   std::vector<BinNode> *nodes, double *ms, int n_threads)
{
	auto t0 = std::chrono::steady_clock::now();
	int64_t leaves = 0;
	if(n_tris > 0)
	{
		if(n_threads <= 1 || n_tris <= sequential_refs()) leaves = Builder(tris, n_tris, scene_box, cfg, nodes).run();
		else leaves = ParallelBuild(tris, n_tris, scene_box, cfg, n_threads).run(nodes);
		std::vector<Box>().swap(sweep_scratch()); // the calling thread's scratch (the workers' died with them)
	}
	if(ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	return leaves;
}

}  // namespace adypt
