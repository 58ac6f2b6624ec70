// hm_device.h — device-side job descriptors shared by the host batch code and the HIP kernels.
// Internal (not part of the C ABI).
#ifndef HM_DEVICE_H
#define HM_DEVICE_H

#include <stdint.h>

#include "hm_stream.h"

// One coded picture (HEIF tile / single image) of a batch.  All pointers are device pointers.
struct hm_dev_pic {
  const uint8_t* blob;      // command stream (hm_pic at offset 0)
  uint8_t* plane[3];        // working planes: reconstruction, deblocked in place
  int32_t pitch[3];         // bytes
  uint16_t* meta;           // per 4x4 luma block: bit0 vertical transform edge on its left, bit1 horizontal edge on
                            // top (deblock.cc:31-62), bit2 PCM / bit3 transquant-bypass coding unit, bits 8-15 QpY (int8)
                            // - one store from k_recon, one load in k_deblock
  int32_t w4, h4;           // size of the 4x4-block maps
  int16_t* resid;           // pictures with split chains: residual samples, k_residual -> k_chain (recon_common.h: ResidGeom)
  uint32_t* mops;           // pictures with split chains: micro-ops, k_residual -> k_chain: 4 dwords per record (recon_common.h:
                            // make_micro_op)
  int16_t* res4;            // pictures with split chains: residual of the 4x4 blocks, k_residual -> k_chain: 16 samples per
                            // record, indexed like the records (entries of other records are never written / looked at)
  uint8_t* hand;            // pictures with split chains: hand-over lines of k_chain's wave-per-row-pair mode - per pair of CTB
                            // rows (monochrome: four rows) the bottom sample line of its last row: ctb_w * ctb luma samples, then
                            // Cb, then Cr (ctb_w * ctb / 2 each)
  // final output of the in-loop filters (SAO stage): written straight into the destination
  // image = fused tile paste (context.cc:2457-2535 of the reference)
  uint8_t* dst[3];          // destination plane origin (tile origin already applied)
  int32_t dst_pitch[3];     // bytes
  int32_t copy_w[3], copy_h[3]; // samples to write per plane (tile cropped at the canvas edge)
  int32_t src_x[3], src_y[3];   // first sample to copy (conformance-window offset of the coded picture)
  int32_t rescale;          // 1: limited->full range rescale quirk of the grid paste (context.cc:2504-2528)
  int32_t width, height;    // luma size (copy of the header)
  int32_t chroma_format;    // 1 or 2
  int32_t bit_depth;
  int32_t log2_ctb;
  int32_t ctb_w, ctb_h;
  int32_t flags;            // hm_pic.flags
  int32_t cb_qp_offset, cr_qp_offset; // pps offsets (chroma deblocking QpC)
  int32_t pcm_loop_filter_disabled;   // sps.pcm_loop_filter_disable_flag
  int32_t n_slices;                   // hm_pic.n_slices
  // sections of the command stream, resolved on the host so that the filter kernels need no dependent load of the header
  const hm_slice* slices;
  const hm_ctb* ctbs;
};

#endif
