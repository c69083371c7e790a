// Every knob of the device side in one place.  read_tunables() (tracer.hip) is the ONLY function of csrc/device that looks at the
// environment: adypt_create / adypt_create_multi / the first RCCL use call it once and keep the clamped result.
//
//   variable                      field                  range          default   what
//   ADYPT_FRAMES_IN_FLIGHT        frames_in_flight       1..128         auto      frames per wavefront pass (adypt_set_frames_in_flight overrides)
//   ADYPT_PIPELINE                pipeline               1..4           1         sub-batch chains of the launch-per-bounce pipeline (adypt_set_pipeline)
//   ADYPT_FUSED_BOUNCES           fused_bounces          0/1            1         bounces 1.. of a batch in one k_path launch (adypt_set_fused_bounces)
//   ADYPT_FIRST_FUSED             first_fused            0/1            1         camera ray + bounce 0 in k_shade_first
//   ADYPT_SINGLE_FUSED            single_fused           0/1            1         a single frame runs as a batch of one through k_path
//   ADYPT_SINGLE_OVERLAP          single_overlap         0/1            1         single frames in a row: frame k + 1 is enqueued on a second stream under the end of frame k's k_path
//   ADYPT_GEN_DEAL                gen_deal               0/1            1         new paths dealt to the 8 queue segments in 256-path chunks
//   ADYPT_SHADE_BIN               shade_bin              0/1            0         k_shade bins by material class (measured slower)
//   ADYPT_REFILL_MIN              refill_min             1..64          16        idle lanes at which a wave refills (secondary rays)
//   ADYPT_REFILL_MIN_PRIMARY      refill_min_primary     1..64          64        ... camera rays (a whole 8x8 tile per wave)
//   ADYPT_BITE / _PRIMARY         bite, bite_primary     1..4096        see traverse.hpp   rays a wave takes from its workgroup's reservation
//   ADYPT_CHUNK                   chunk                  16..4096       see traverse.hpp   rays a workgroup reserves per device atomic
//   ADYPT_ENDGAME                 endgame                0..1024        see traverse.hpp   size of the end-of-launch ray pool
//   ADYPT_SHADE_MIN               shade_min              1..64          64        deposited hits a k_path wave waits for before shading
//   ADYPT_RARE_MIN                rare_min               0..64          48        deferred hits (glossy lobe / dielectric) a k_path shading round waits for; 0 = nothing is deferred
//   ADYPT_DEFER_MAX               defer_max              0..64          24        ... and a round defers them only when it holds at most this many
//   ADYPT_LDS_STACK_DEPTH         lds_stack_depth        1..kLdsStackMax  auto    LDS part of k_trace's stack (tests: forces the HBM spill path)
//   ADYPT_TRACE_BLOCKS_PER_CU     trace_blocks_per_cu    1..16          auto      k_trace workgroups per CU
//   ADYPT_PATH_BLOCKS_PER_CU      path_blocks_per_cu     1..8           6         k_path workgroups per CU
//   ADYPT_PATH_LDS_DEPTH          path_lds_depth         1..kLdsStackMax  auto    LDS part of k_path's stack
//   ADYPT_PATH_VERBOSE            path_verbose           0/1            0         print k_path's launch geometry
//   ADYPT_REF_TRIANGLES_MAX_MB    ref_triangles_max_mb   0..2^20        auto      per-reference triangle copy for k_path only below this size (0 = never)
//   ADYPT_RCCL_LIB                rccl_lib               path           ""        librccl to dlopen first (adypt_amd/_native.py: the one bundled with torch)
//   ADYPT_GATHER_TIMEOUT          gather_timeout_s       0..86400       120       watchdog of the multi-GPU gather (0 = none): the process exits non-zero
//
// Test hooks — honoured ONLY after adypt_enable_test_hooks(ADYPT_TEST_HOOKS_MAGIC) in this process (tests/, bench.py --rehearsal); otherwise the
// shipped library behaves as if they were unset, whatever the environment says:
//   ADYPT_MULTI_SHARED_DEVICE=1     several tile shards on ONE device, peer -> root by device copies instead of ncclSend / ncclRecv
//   ADYPT_COMM_TRANSPORT=host       the RCCL call table served by a shared-memory transport (host_transport.hpp)
//   ADYPT_HOST_TRANSPORT_TIMEOUT    seconds that transport waits for a peer
//   ADYPT_AUDIT_SELFTEST=1          the slot-claim audit plants a double claim before every check
//   ADYPT_GATHER_STALL_TEST=1       the gather stalls forever after arming its watchdog (tests the watchdog)
#pragma once
#include <string>

namespace adypt {

struct Tunables {
	int frames_in_flight = 0;       // 0 = automatic
	int pipeline = 1;
	int fused_bounces = 1, first_fused = 1, single_fused = 1, single_overlap = 1, gen_deal = 1, shade_bin = 0;
	int refill_min = 0, refill_min_primary = 0, bite = 0, bite_primary = 0, chunk = 0, endgame = -1, shade_min = 0, rare_min = -1, defer_max = -1; // 0 (endgame, rare_min, defer_max: -1) = the built-in default
	int lds_stack_depth = 0, trace_blocks_per_cu = 0, path_blocks_per_cu = 0, path_lds_depth = 0, path_verbose = 0;
	long ref_triangles_max_mb = -1; // -1 = automatic
	std::string rccl_lib;
	double gather_timeout_s = 120.0;
	// test hooks (all off unless adypt_enable_test_hooks was called)
	bool multi_shared_device = false, comm_transport_host = false, audit_selftest = false, gather_stall_test = false;
	double host_transport_timeout_s = 0.0; // 0 = the transport's default
};

Tunables read_tunables();
bool test_hooks_enabled();

}  // namespace adypt
