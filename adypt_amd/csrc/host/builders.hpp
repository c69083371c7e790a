// Internal interface of the CPU BVH producers (binary SBVH -> 8-wide compressed BVH).
#pragma once
#include "common.hpp"
#include "../../../include/adypt_host.h"

namespace adypt {

// returns the number of leaves (= triangle references incl. spatial-split duplicates); the node array does not
// depend on n_threads
int64_t build_sbvh(const TriRec *tris, int64_t n_tris, const Box &scene_box, const adypt_bvh_params &cfg,
				   std::vector<BinNode> *nodes, double *ms, int n_threads);

void build_wide_bvh(const std::vector<BinNode> &bin, int64_t leaf_count, const adypt_bvh_params &cfg,
					std::vector<NodeRec> *nodes, std::vector<int32_t> *tri_indices, double *ms, int n_threads);

}  // namespace adypt
