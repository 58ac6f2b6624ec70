// hm_knobs.h - the library's tuning and test knobs (common.cpp); no HIP types: the host parser includes it too.
#ifndef HM_KNOBS_H
#define HM_KNOBS_H
#ifdef __cplusplus
extern "C" {
#endif
// Tuning and test knobs.  NOTHING in the library reads them from the environment (r05: a stray HM_CHAIN_RING in a service's
// environment must not change every decode's kernel), and (r06) the library that ships has no exported way to set them:
// hm_knob_set is hidden.  The setter the tests and measurement scripts use, hm_debug_set, lives in test_hooks.cpp, which is
// linked only into libheif_mi355x_test.so - the same objects as libheif_mi355x.so plus that one file (csrc/Makefile);
// tests/knobs.py maps the HM_* variables of the measurement scripts onto it.  Read through hm_knob().  Atomic words: worker
// threads read them while a test sets them.
//   chain_spin_limit (0)   polls without news before a chain wave gives up a wait (0: the default bound)
//   chain_test_stall (0)   fault injection: the first band of every picture never announces its progress
//   batch_fail_width (0)   hm_batch_execute refuses batches holding a picture of that width (failure-isolation tests)
//   chain_pairs (-1)       cut of a picture in k_chain: 0 a wave per picture, 1 / 2 / 3 a wave per pair of rows / row / chain
//   chain_share (0)        >= 2: waves per picture that take its pairs of rows in turn through HBM
//   chain_ring (-1)        W >= 2: the ring of W waves per picture in one workgroup; 0: never
//   chain_alt (1)          0: the one-chain waves of a ring keep their kind of chain
//   chain_np (0)           pictures per workgroup of the wave-per-picture cut
//   chain_debug (0)        print the launcher's choice to stderr
//   chain_split (1)        0: the partial last round of a wave per picture stays in the one launch; 1 / 2: a launch of its own beside
//                          the full rounds (queued first / second)
//   resid_segs (0)         runs of CTUs a row of k_residual is cut into
//   recon_waves (0)        waves per workgroup of k_recon
//   quad_class (-1)        record order of the parser: 1 split chains for every class that has them, 0 for none
//   tail_fused (1)         0: the separate filter / colour kernels also where the fused ones apply
//   grid_slab_rows (-1)    hm_decode_item on a grid: tile rows per slab of the decode that runs under the host's entropy decode (0: one batch behind it, as before r06; -1: about a round of the parsing threads)
//   chain_early (1)        0: the few-pictures cuts of k_chain start a CTU when the CTU above-RIGHT is done (the rule before r06) instead of the CTU above
//   tail_hdr16 (1)         0: 9..11-bit 4:2:0 pictures -> RGB24 / RGBA32 (the HDR class) through k_tailf instead of k_tail420's 16-bit instantiation
//   stream_interleaved (0) 1: records in decode order (format of the rare-syntax classes) for every picture
enum hm_knob_id { HM_KNOB_CHAIN_SPIN_LIMIT, HM_KNOB_CHAIN_TEST_STALL, HM_KNOB_BATCH_FAIL_WIDTH, HM_KNOB_CHAIN_PAIRS, HM_KNOB_CHAIN_SHARE, HM_KNOB_CHAIN_RING,
                  HM_KNOB_CHAIN_ALT, HM_KNOB_CHAIN_NP, HM_KNOB_CHAIN_DEBUG, HM_KNOB_RESID_SEGS, HM_KNOB_RECON_WAVES, HM_KNOB_QUAD_CLASS, HM_KNOB_TAIL_FUSED,
                  HM_KNOB_STREAM_INTERLEAVED, HM_KNOB_CHAIN_SPLIT, HM_KNOB_TAIL_HDR16, HM_KNOB_CHAIN_EARLY, HM_KNOB_GRID_SLAB_ROWS, HM_KNOB_COUNT };
int hm_knob(int id);
int hm_knob_set(const char* name, int value); // 0, or -1 for an unknown name (hidden: reached through test_hooks.cpp's hm_debug_set only)
#ifdef __cplusplus
}
#endif
#endif
