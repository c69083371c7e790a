// A multi-threaded sort that returns, element for element, the permutation `std::sort` of this toolchain's C++
// library (libstdc++) returns for the same input and comparator.
//
// Why it exists: the reference's SBVH builder orders triangle references with std::sort by (centroid[axis], triangle
// id) (src/BVH/SBVHBuilder.hpp:95-134).  std::sort is not stable, and references that tie are common — the two halves
// of a spatially split wall or floor triangle share the triangle id and, on the axis the triangle is flat in, the
// centroid — so WHICH permutation comes back is part of what makes the node array bit-identical to the reference's.
// That permutation is a deterministic function of the input sequence: libstdc++'s sort is the published introsort
// (Musser 1997) — quicksort on the median of (first+1, middle, last-1) moved to the front, unguarded Hoare partition,
// the right part handled first and the left part iterated, heap sort (partial_sort over the whole range) once
// 2*floor(log2 n) partitions deep, ranges of <= 16 elements left to one final stable insertion pass.  This file
// restates that algorithm; the only change is that the right part of a partition may run on another thread, and the
// insertion pass is applied per finished <= 16 range (equivalent: the ranges are mutually ordered, so the final pass
// never moves an element out of its range).  `adypt_host_selftest_sort` (host_api.cpp) checks it against std::sort on
// random, heavily tied, sorted, reversed and adversarial ("quicksort killer", reaches the heap-sort fallback) inputs.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace adypt {

template <class T, class Less> class ExactSort {
public:
	// min_task: ranges smaller than this are never handed to another thread
	ExactSort(Less less, int threads, int64_t min_task = 1 << 15) : less_(less), threads_(threads < 1 ? 1 : threads), min_task_(min_task < 32 ? 32 : min_task) {}

	void sort(T *first, T *last)
	{
		if(last - first < 2) return;
		int depth = 0;
		for(uint64_t n = (uint64_t)(last - first); n > 1; n >>= 1) ++depth; // floor(log2 n)
		if(threads_ == 1 || last - first < 2 * min_task_) { loop(first, last, 2 * depth, false); return; }
		push({first, last, 2 * depth});
		std::vector<std::thread> pool;
		for(int i = 1; i < threads_; ++i) pool.emplace_back([this] { work(); });
		work();
		for(std::thread &t : pool) t.join();
	}

private:
	struct Range { T *first, *last; int depth_limit; };
	static constexpr int kThreshold = 16;
	Less less_;
	int threads_;
	int64_t min_task_;
	std::mutex mu_;
	std::condition_variable cv_;
	std::deque<Range> ready_;
	int64_t unfinished_ = 0;

	void push(Range r)
	{
		{
			std::lock_guard<std::mutex> g(mu_);
			ready_.push_back(r);
			++unfinished_;
		}
		cv_.notify_one();
	}
	void work()
	{
		for(;;)
		{
			Range r;
			{
				std::unique_lock<std::mutex> g(mu_);
				cv_.wait(g, [this] { return !ready_.empty() || unfinished_ == 0; });
				if(ready_.empty()) return;
				r = ready_.front();
				ready_.pop_front();
			}
			loop(r.first, r.last, r.depth_limit, true);
			bool done;
			{
				std::lock_guard<std::mutex> g(mu_);
				done = --unfinished_ == 0;
			}
			if(done) cv_.notify_all();
		}
	}

	// median of *a, *b, *c swapped into *result
	void median_to_first(T *result, T *a, T *b, T *c)
	{
		if(less_(*a, *b))
		{
			if(less_(*b, *c)) std::iter_swap(result, b);
			else if(less_(*a, *c)) std::iter_swap(result, c);
			else std::iter_swap(result, a);
		}
		else if(less_(*a, *c)) std::iter_swap(result, a);
		else if(less_(*b, *c)) std::iter_swap(result, c);
		else std::iter_swap(result, b);
	}
	T *partition(T *first, T *last, T *pivot)
	{
		for(;;)
		{
			while(less_(*first, *pivot)) ++first;
			--last;
			while(less_(*pivot, *last)) --last;
			if(!(first < last)) return first;
			std::iter_swap(first, last);
			++first;
		}
	}
	// stable insertion sort of one finished range
	void insertion(T *first, T *last)
	{
		if(first == last) return;
		for(T *i = first + 1; i != last; ++i)
		{
			T val = std::move(*i);
			T *j = i;
			while(j != first && less_(val, *(j - 1))) { *j = std::move(*(j - 1)); --j; }
			*j = std::move(val);
		}
	}

	void loop(T *first, T *last, int depth_limit, bool may_spawn)
	{
		while(last - first > kThreshold)
		{
			if(depth_limit == 0)
			{
				std::partial_sort(first, last, last, less_); // the library's heap sort of the whole range (sorted: the final pass leaves it alone)
				return;
			}
			--depth_limit;
			T *mid = first + (last - first) / 2;
			median_to_first(first, first + 1, mid, last - 1);
			T *cut = partition(first + 1, last, first);
			if(may_spawn && last - cut >= min_task_ && cut - first >= min_task_) push({cut, last, depth_limit});
			else loop(cut, last, depth_limit, may_spawn);
			last = cut;
		}
		insertion(first, last);
	}
};

}  // namespace adypt
