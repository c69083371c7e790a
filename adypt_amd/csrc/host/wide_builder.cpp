// Collapse of the binary SBVH into the 8-wide compressed BVH ("CWBVH8", Ylitie et al. 2017) with 80-byte nodes.
//
// Restates the decisions of the reference's collapse (src/BVH/WideBVHBuilder.cpp:24-273) so that node and index
// arrays come out bit-identical (pinned by tests/golden/*.bvh):
//   * bottom-up SAH dynamic program over "i in 1..7 roots of a forest cut from this subtree" (cpp:24-91):
//       leaf cost  A*triSAH*n (n <= 3), internal cost A*nodeSAH*8 + min_k(L[k]+R[8-k]), distribute cost
//       min_k(L[k]+R[i-k]) vs. dp[i-1]; strict '<' everywhere (first minimum wins)
//   * top-down emission: children of a wide node = the DP's cut (cpp:93-108); child -> slot assignment by a
//     min-cost assignment on the signed centroid offsets (cpp:199-208, 120-163); per-node power-of-two cell size
//     with 8-bit quantised child boxes (cpp:172-192, 224-237); meta byte / imask encoding (cpp:238-265)
//   * leaves' triangle references are emitted right-subtree-first (cpp:110-118)
// Not reproduced: the out-of-bounds write `order[p[i]-1]` for p[i]==0 (cpp:160-161) — it only scribbles on the
// reference's own stack; unused slots are simply skipped here.
//
// Parallel collapse.  Both phases are subtree-local, and the binary tree is laid out in pre-order
// (a subtree is one contiguous index range), so they split over threads without changing a byte:
//   * the DP of a subtree only reads its own range: disjoint subtrees are swept concurrently, the few nodes above them
//     afterwards;
//   * emission is depth first, so a wide subtree occupies one contiguous run of the node array and of the index
//     array; the run lengths are known from the DP (wide nodes below a binary node: `wide_below_`, references: the
//     leaf count), so every subtree can be emitted by its own task at its final position.
#include "builders.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace adypt {
namespace {

enum CutType : uint8_t { kInternal = 0, kLeaf = 1, kDistribute = 2 };
struct Cut { float sah; uint8_t type; uint8_t split[2]; uint8_t pad; }; // 8 B: 56 B of DP state per binary node
struct CutRow { Cut c[7]; Cut &operator[](int i) { return c[i - 1]; } const Cut &operator[](int i) const { return c[i - 1]; } };

class Collapser {
public:
	Collapser(const std::vector<BinNode> &bin, const adypt_bvh_params &cfg, std::vector<NodeRec> *nodes, std::vector<int32_t> *idx, int n_threads)
		: bin_(bin), cfg_(cfg), nodes_(*nodes), idx_(*idx), n_threads_(std::max(1, n_threads))
	{
		if(const char *ev = getenv("ADYPT_BUILD_GRAIN")) { int v = atoi(ev); if(v >= 1) grain_ = v; } // tests: cut the tiny fixtures too
	}

	void run(int64_t leaf_count)
	{
		nodes_.clear();
		idx_.clear();
		if(bin_.empty()) return;
#ifdef ADYPT_BUILD_TIMING
		auto T0 = std::chrono::steady_clock::now();
		auto lap = [&](const char *what) { auto T1 = std::chrono::steady_clock::now(); fprintf(stderr, "[wide] %s %.2f s\n", what, std::chrono::duration<double>(T1 - T0).count()); T0 = T1; };
#else
		auto lap = [](const char *) {};
#endif
		// DP state, left uninitialised: every entry is written by eval_cost before anything reads it, and the threads
		// of the parallel sweep are then the first to touch their own ranges
		cost_.reset(new CutRow[bin_.size()]);
		tri_count_.reset(new int32_t[bin_.size()]);
		wide_below_.reset(new uint32_t[bin_.size()]);
		lap("alloc");
		const bool parallel = n_threads_ > 1 && (int64_t)bin_.size() > 4 * grain_ && !is_leaf(0);
		// ---- phase 1: bottom-up DP.  Children always have larger indices than their parent (pre-order emission), so a
		// reverse sweep over a subtree's index range is a post-order walk of it.
		if(!parallel)
			for(int64_t i = (int64_t)bin_.size() - 1; i >= 0; --i) eval_cost((int)i);
		else
		{
			struct Range { int32_t begin, end; };
			std::vector<Range> subtrees;   // disjoint subtrees swept concurrently
			std::vector<int32_t> top;      // the nodes above them, in increasing index order
			std::vector<Range> todo{{0, (int32_t)bin_.size()}};
			while(!todo.empty())
			{
				Range r = todo.back();
				todo.pop_back();
				if(r.end - r.begin <= grain_ || is_leaf(r.begin)) { subtrees.push_back(r); continue; }
				top.push_back(r.begin);
				todo.push_back({r.begin + 1, left(r.begin)});  // right subtree = [node + 1, left child)
				todo.push_back({left(r.begin), r.end});        // left subtree = [left child, end of the parent's range)
			}
			std::sort(subtrees.begin(), subtrees.end(), [](const Range &a, const Range &b) { return a.end - a.begin > b.end - b.begin; });
			std::atomic<size_t> next{0};
			auto sweep = [&] {
				for(size_t k; (k = next.fetch_add(1)) < subtrees.size();)
					for(int32_t i = subtrees[k].end - 1; i >= subtrees[k].begin; --i) eval_cost(i);
			};
			std::vector<std::thread> pool;
			for(int t = 1; t < n_threads_; ++t) pool.emplace_back(sweep);
			sweep();
			for(std::thread &t : pool) t.join();
			std::sort(top.begin(), top.end());
			for(size_t k = top.size(); k-- > 0;) eval_cost(top[k]);
		}
		lap("dp");
		// ---- phase 2: top-down emission
		if(!parallel)
		{
			nodes_.emplace_back(NodeRec{});
			idx_.reserve((size_t)leaf_count);
			Cursor cur{&nodes_, &idx_, 1, 0, true}; // node 0 is the root
			emit_subtree(0, 0, &cur);
			nodes_.shrink_to_fit();
			return;
		}
		nodes_.assign((size_t)std::max(1u, wide_below_[0]), NodeRec{}); // wide_below_[0] counts the root itself
		idx_.assign((size_t)tri_count_[0], 0);
		push({0, 0, 1, 0});
		std::vector<std::thread> pool;
		for(int t = 1; t < n_threads_; ++t) pool.emplace_back([this] { work(); });
		work();
		for(std::thread &t : pool) t.join();
		lap("emit");
	}

private:
	const std::vector<BinNode> &bin_;
	adypt_bvh_params cfg_;
	std::vector<NodeRec> &nodes_;
	std::vector<int32_t> &idx_;
	std::unique_ptr<CutRow[]> cost_;
	std::unique_ptr<int32_t[]> tri_count_;
	std::unique_ptr<uint32_t[]> wide_below_;  // wide nodes emitted for binary node n as a direct child: itself + everything below (0 for a leaf cut)
	int n_threads_;
	int32_t grain_ = 1 << 15;           // binary nodes per DP sweep task / references per emission task

	// where the next wide node / triangle reference of a subtree goes.  Sequential build: the arrays grow; parallel
	// build: they are pre-sized and every task writes its own run.
	struct Cursor { std::vector<NodeRec> *nodes; std::vector<int32_t> *idx; uint32_t node, tri; bool growing; };
	struct EmitTask { int w, s; uint32_t node_base, tri_base; };
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<EmitTask> ready_;
	int64_t unfinished_ = 0;

	void push(EmitTask t)
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
			EmitTask t;
			{
				std::unique_lock<std::mutex> g(mu_);
				cv_.wait(g, [this] { return !ready_.empty() || unfinished_ == 0; });
				if(ready_.empty()) return;
				t = ready_.front();
				ready_.pop_front();
			}
			Cursor cur{&nodes_, &idx_, t.node_base, t.tri_base, false};
			if(tri_count_[(size_t)t.s] <= grain_) emit_subtree(t.w, t.s, &cur);
			else
			{
				// one wide node here, its internal children as tasks at their final positions
				std::vector<std::pair<int, int>> kids;
				emit(t.w, t.s, &kids, &cur);
				uint32_t node_at = cur.node, tri_at = cur.tri;
				for(size_t k = kids.size(); k-- > 0;) // `kids` is in reversed gather order (it is a stack for the sequential walk)
				{
					push({kids[k].first, kids[k].second, node_at, tri_at});
					node_at += wide_below_[(size_t)kids[k].second] - 1;
					tri_at += (uint32_t)tri_count_[(size_t)kids[k].second];
				}
			}
			bool done;
			{
				std::lock_guard<std::mutex> g(mu_);
				done = --unfinished_ == 0;
			}
			if(done) cv_.notify_all();
		}
	}

	// depth-first emission of the wide subtree rooted at wide node w = binary node s
	void emit_subtree(int w, int s, Cursor *cur)
	{
		std::vector<std::pair<int, int>> todo; // (wide node, binary node)
		todo.emplace_back(w, s);
		while(!todo.empty())
		{
			auto [tw, ts] = todo.back();
			todo.pop_back();
			emit(tw, ts, &todo, cur);
		}
	}

	// wide nodes emitted for the forest the DP cuts out of binary node n with a budget of i roots
	uint32_t forest_wide(int n, int i) const
	{
		const Cut &c = cost_[(size_t)n][i];
		if(c.type != kDistribute) return wide_below_[(size_t)n];
		return forest_wide(left(n), c.split[0]) + forest_wide(right(n), c.split[1]);
	}

	bool is_leaf(int i) const { return bin_[(size_t)i].left == -1; }
	int left(int i) const { return bin_[(size_t)i].left; }
	static int right(int i) { return i + 1; }
	float tri_cost(int n) const { return cfg_.triangle_sah * n; }
	float node_cost(int n) const { return cfg_.node_sah * n; }

	void eval_cost(int n)
	{
		float area = bin_[(size_t)n].box.area();
		CutRow &dp = cost_[(size_t)n];
		if(is_leaf(n))
		{
			for(int i = 1; i <= 7; ++i) { dp[i].sah = tri_cost(1) * area; dp[i].type = kLeaf; dp[i].split[0] = dp[i].split[1] = 0; }
			tri_count_[(size_t)n] = 1;
			wide_below_[(size_t)n] = 0;
			return;
		}
		const int l = left(n), r = right(n);
		const CutRow &L = cost_[(size_t)l], &R = cost_[(size_t)r];
		const int tc = tri_count_[(size_t)r] + tri_count_[(size_t)l];
		tri_count_[(size_t)n] = tc;
		{
			float c_leaf = tc <= 3 ? area * tri_cost(tc) : FLT_MAX;
			float c_int = FLT_MAX;
			float node_sah = area * node_cost(8);
			dp[1].split[0] = dp[1].split[1] = 0;
			for(int k = 1; k < 8; ++k)
			{
				float v = node_sah + L[k].sah + R[8 - k].sah;
				if(v < c_int) { c_int = v; dp[1].split[0] = (uint8_t)k; dp[1].split[1] = (uint8_t)(8 - k); }
			}
			if(c_leaf < c_int) { dp[1].sah = c_leaf; dp[1].type = kLeaf; }
			else { dp[1].sah = c_int; dp[1].type = kInternal; }
		}
		for(int i = 2; i <= 7; ++i)
		{
			float c_dist = FLT_MAX;
			dp[i].split[0] = dp[i].split[1] = 0;
			for(int k = 1; k < i; ++k)
			{
				float v = L[k].sah + R[i - k].sah;
				if(v < c_dist) { c_dist = v; dp[i].split[0] = (uint8_t)k; dp[i].split[1] = (uint8_t)(i - k); }
			}
			if(c_dist < dp[i - 1].sah) { dp[i].sah = c_dist; dp[i].type = kDistribute; }
			else dp[i] = dp[i - 1];
		}
		wide_below_[(size_t)n] = dp[1].type == kInternal ? 1u + forest_wide(l, dp[1].split[0]) + forest_wide(r, dp[1].split[1]) : 0u;
	}

	// the binary nodes that become the children of a wide node rooted at (n, i)
	void gather_children(int n, int i, int *count, int out[8]) const
	{
		const int child[2] = {left(n), right(n)};
		const int share[2] = {cost_[(size_t)n][i].split[0], cost_[(size_t)n][i].split[1]};
		for(int c = 0; c < 2; ++c)
		{
			if(cost_[(size_t)child[c]][share[c]].type == kDistribute) gather_children(child[c], share[c], count, out);
			else out[(*count)++] = child[c];
		}
	}

	int append_leaf_refs(int n, Cursor *cur)
	{
		// right subtree first; iterative to stay safe on deep chains (a leaf cut holds at most 3 references)
		int cnt = 0;
		int st[8], sp = 0;
		st[sp++] = n;
		while(sp)
		{
			int c = st[--sp];
			if(is_leaf(c))
			{
				if(cur->growing) idx_.push_back(bin_[(size_t)c].tri);
				else idx_[cur->tri] = bin_[(size_t)c].tri;
				++cur->tri;
				++cnt;
			}
			else { st[sp++] = left(c); st[sp++] = right(c); }
		}
		return cnt;
	}

	// min-cost assignment of `n` rows (children) to 8 columns (slots), potentials method; slot_of[row] = column
	static void assign_slots(const float cost[8][8], int n, int slot_of[8])
	{
		const float INF = 1e12f;
		int match[9], way[9];     // match[col] = row matched to col (1-based, 0 = none)
		float u[9], v[9], minv[9];
		bool used[9];
		std::fill(u, u + 9, 0.0f); std::fill(v, v + 9, 0.0f);
		std::fill(way, way + 9, 0); std::fill(match, match + 9, 0);
		for(int row = 1; row <= n; ++row)
		{
			match[0] = row;
			int j0 = 0;
			std::fill(minv, minv + 9, INF);
			std::fill(used, used + 9, false);
			do
			{
				used[j0] = true;
				int i0 = match[j0], j1 = 0;
				float delta = INF;
				for(int j = 1; j <= 8; ++j)
					if(!used[j])
					{
						float cur = cost[i0 - 1][j - 1] - u[i0] - v[j];
						if(cur < minv[j]) { minv[j] = cur; way[j] = j0; }
						if(minv[j] < delta) { delta = minv[j]; j1 = j; }
					}
				for(int j = 0; j <= 8; ++j)
					if(used[j]) { u[match[j]] += delta; v[j] -= delta; }
					else minv[j] -= delta;
				j0 = j1;
			} while(match[j0] != 0);
			do
			{
				int j1 = way[j0];
				match[j0] = match[j1];
				j0 = j1;
			} while(j0);
		}
		for(int j = 1; j <= 8; ++j)
			if(match[j] != 0) slot_of[match[j] - 1] = j - 1;
	}

	static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
	static uint32_t to_u32(float f) { return (uint32_t)(int64_t)f; } // x86-64 float -> unsigned conversion path

	void emit(int w, int s, std::vector<std::pair<int, int>> *todo, Cursor *cur_pos)
	{
		int child[8], n_child = 0;
		// a one-triangle scene has a leaf as its binary root (the reference's collapse dereferences its missing children and
		// crashes): the root wide node then gets that leaf as its only child
		if(is_leaf(s)) child[n_child++] = s;
		else gather_children(s, 1, &n_child, child);
		const Box &box = bin_[(size_t)s].box;
		Vec3 cell;
		{
			NodeRec &cur = nodes_[(size_t)w];
			cur.px = box.lo.x; cur.py = box.lo.y; cur.pz = box.lo.z;
			const float kBase = float(1.0 / double((1 << 8) - 1));
			cell = (box.hi - box.lo) * kBase;
			int ex = cell.x == 0 ? -128 : (int)std::ceil(std::log2(cell.x));
			int ey = cell.y == 0 ? -128 : (int)std::ceil(std::log2(cell.y));
			int ez = cell.z == 0 ? -128 : (int)std::ceil(std::log2(cell.z));
			cell.x = exp2f((float)ex); cell.y = exp2f((float)ey); cell.z = exp2f((float)ez);
			cur.ex = (uint8_t)(f2u(cell.x) >> 23); cur.ey = (uint8_t)(f2u(cell.y) >> 23); cur.ez = (uint8_t)(f2u(cell.z) >> 23);
		}
		int slot_of[8];
		{
			float m[8][8];
			const Vec3 pc = box.center();
			for(int i = 0; i < n_child; ++i)
				for(int j = 0; j < 8; ++j)
				{
					Vec3 d = bin_[(size_t)child[i]].box.center() - pc;
					m[i][j] = ((j & 1) ? -d.x : d.x) + ((j & 2) ? -d.y : d.y) + ((j & 4) ? -d.z : d.z);
				}
			assign_slots(m, n_child, slot_of);
		}
		int in_slot[8];
		std::fill(in_slot, in_slot + 8, -1);
		for(int i = 0; i < n_child; ++i) in_slot[slot_of[i]] = child[i];

		const uint32_t child_base = cur_pos->node, tri_base = cur_pos->tri;
		nodes_[(size_t)w].imask = 0;
		nodes_[(size_t)w].child_base = child_base;
		nodes_[(size_t)w].tri_base = tri_base;
		for(int i = 0; i < 8; ++i)
		{
			int c = in_slot[i];
			if(c < 0) { nodes_[(size_t)w].meta[i] = 0; continue; }
			const Box &cb = bin_[(size_t)c].box;
			Vec3 ql = cb.lo - box.lo, qh = cb.hi - box.lo;
			uint32_t lo[3] = {to_u32(std::floor(ql.x / cell.x)), to_u32(std::floor(ql.y / cell.y)), to_u32(std::floor(ql.z / cell.z))};
			uint32_t hi[3] = {to_u32(std::ceil(qh.x / cell.x)), to_u32(std::ceil(qh.y / cell.y)), to_u32(std::ceil(qh.z / cell.z))};
			for(int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], 255u); hi[a] = std::min(hi[a], 255u); }
			NodeRec &cur = nodes_[(size_t)w];
			cur.qlox[i] = (uint8_t)lo[0]; cur.qloy[i] = (uint8_t)lo[1]; cur.qloz[i] = (uint8_t)lo[2];
			cur.qhix[i] = (uint8_t)hi[0]; cur.qhiy[i] = (uint8_t)hi[1]; cur.qhiz[i] = (uint8_t)hi[2];
			const int32_t type = cost_[(size_t)c][1].type;
			if(type == kLeaf)
			{
				uint32_t off = cur_pos->tri - tri_base;
				int cnt = append_leaf_refs(c, cur_pos);
				uint8_t head = cnt == 1 ? 0x20 : cnt == 2 ? 0x60 : 0xe0;
				nodes_[(size_t)w].meta[i] = (uint8_t)(head | off);
			}
			else if(type == kInternal)
			{
				uint32_t widx = cur_pos->node - child_base;
				if(cur_pos->growing) nodes_.emplace_back(NodeRec{});
				++cur_pos->node;
				NodeRec &cur2 = nodes_[(size_t)w];
				cur2.meta[i] = (uint8_t)(cur2.meta[i] | (1u << 5) | (widx + 24u));
				cur2.imask = (uint8_t)(cur2.imask | (1u << widx));
			}
		}
		// descend into internal children in gather order (depth first): push reversed so they pop in order
		for(int i = n_child - 1; i >= 0; --i)
			if(cost_[(size_t)child[i]][1].type == kInternal)
				todo->emplace_back((int)(child_base + (nodes_[(size_t)w].meta[slot_of[i]] & 0x1fu) - 24u), child[i]);
	}
};

}  // namespace

void build_wide_bvh(const std::vector<BinNode> &bin, int64_t leaf_count, const adypt_bvh_params &cfg,
					std::vector<NodeRec> *nodes, std::vector<int32_t> *tri_indices, double *ms, int n_threads)
{
	auto t0 = std::chrono::steady_clock::now();
	Collapser(bin, cfg, nodes, tri_indices, n_threads).run(leaf_count);
	if(ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace adypt
