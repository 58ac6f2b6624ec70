/* hm_stream.h — the GPU command stream of one coded picture (one HEIF tile / image).
 *
 * Produced on the host by the entropy decoder (hm_hevc_parse, CABAC stays on the CPU as in
 * the reference: third-party/libde265/libde265/slice.cc:2886-5600), consumed by the HIP
 * reconstruction / deblock / SAO kernels and — in tests — by the oracle's scalar executors.
 * It carries exactly the information libde265 keeps in tctx->coeffList/coeffPos/nCoeff
 * (decctx.h:88-92), its per-CB/TU metadata arrays (image.h:190-215,414-420) and sao_info
 * (slice.h:457-465), flattened into plain little-endian POD arrays.
 *
 * Layout of one picture blob (all offsets in bytes from the blob start, each 16-byte aligned):
 *     hm_pic            header
 *     hm_slice[n_slices]
 *     hm_ctb[n_ctbs]    raster order
 *     hm_tu[n_tus]      see "record order" below (hm_tu6[n_tus] in pictures with HM_PIC_SPLIT_CHAINS)
 *     hm_coeff[n_coeffs]
 *
 * Record order.  Intra prediction chains the blocks of one colour plane; luma and chroma never read each other (pictures
 * with cross-component prediction are rare-syntax pictures in decode order), so a picture holds two independent block
 * chains per CTB row.
 *   HM_PIC_SPLIT_CHAINS set (every picture without rare syntax): for CTB row 0, 1, ...: the luma records of the row's
 *     CTBs in raster order, each CTB's in decode order, then the chroma records (Cb and Cr, decode order) of the row's
 *     CTBs likewise.  hm_ctb.tu_first / tu_count delimit the CTB's luma records, tu_first_c / tu_count_c its chroma
 *     records; within a row both lists are contiguous from CTB to CTB, so a kernel walks each with a running index.
 *     The records are the compact 6-byte hm_tu6, and the levels (hm_coeff) lie in the order of the records, so a
 *     record's first level is the running sum of the counts before it: hm_ctb.coeff_first / coeff_first_c give that
 *     sum at the CTB's first luma / chroma record.  Format HSM5 (r04): the records no longer carry the five neighbour-
 *     availability answers of a block - they are a pure function of the block's rectangle, the picture size and four
 *     bits per CTB (hm_ctb.nb_avail), which the residual pre-pass evaluates with a lane per record (hm_avail.h) - nor a
 *     second QP (the deblocking QpY of a luma block is its dequantisation QP minus QpBdOffset; chroma records need
 *     none): 0.40 bytes of command stream per luma sample on the benchmark tiles (HSM4: 0.48, HSM3: 0.78) - the
 *     stream crosses PCIe, and it is what bounds the device-inclusive clock.
 *   HM_PIC_SPLIT_CHAINS clear (pictures with HM_PIC_RARE_SYNTAX): all records of a CTB in decode order in
 *     [tu_first, tu_first + tu_count), CTBs in raster order; tu_count_c = 0.
 */
#ifndef HM_STREAM_H
#define HM_STREAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HM_STREAM_MAGIC 0x354d5348u /* "HSM5" */

/* hm_pic.flags */
#define HM_PIC_STRONG_INTRA_SMOOTHING 0x0001u /* sps.strong_intra_smoothing_enable_flag        */
#define HM_PIC_SAO_ENABLED            0x0002u /* sps.sample_adaptive_offset_enabled_flag       */
#define HM_PIC_DEBLOCK_ANY            0x0004u /* at least one slice has deblocking enabled     */
#define HM_PIC_SAO_ANY                0x0008u /* at least one slice has SAO luma or chroma on  */
#define HM_PIC_HAS_VUI_COLOUR         0x0010u /* video_signal_type present in the VUI          */
#define HM_PIC_SIGN_HIDING            0x0020u /* informational                                 */
#define HM_PIC_TILES                  0x0040u /* pps.tiles_enabled_flag                        */
#define HM_PIC_LF_ACROSS_TILES        0x0080u /* pps.loop_filter_across_tiles_enabled_flag     */
#define HM_PIC_SCALING_LIST           0x0100u /* sps.scaling_list_enable_flag: off_scaling valid */
#define HM_PIC_PCMF                   0x0200u /* (pcm_enabled && pcm_loop_filter_disabled) || transquant_bypass_enabled:
                                                 the reference's deblocking takes its "pcmf" branches (deblock.cc:723)  */
#define HM_PIC_LOSSLESS_CUS           0x0400u /* at least one PCM or cu_transquant_bypass coding unit            */
#define HM_PIC_444                    0x0800u /* chroma_format_idc 3 (chroma_format says the same; the flag puts such
                                                 pictures into the rare-syntax classes)                        */
#define HM_PIC_SPLIT_CHAINS           0x1000u /* luma and chroma records in separate lists (see "record order")   */
/* range-extension tools that change the reconstruction (sps.cc:1375-1390, pps.cc:68 of the reference) */
#define HM_PIC_TS_ROTATION            0x2000u /* transform_skip_rotation_enabled_flag: the levels of a 4x4 transform-skip /
                                                 bypass block are rotated by 180 degrees (transform.cc:427-447, 575)   */
#define HM_PIC_IMPLICIT_RDPCM         0x4000u /* implicit_rdpcm_enabled_flag: a transform-skip / bypass block predicted
                                                 with mode 10 (26) accumulates its residual along rows (columns)
                                                 (slice.cc:3774-3779); bypass blocks: no boundary filter of modes 10 / 26
                                                 (intrapred.cc:323-326)                                             */
#define HM_PIC_NO_INTRA_SMOOTHING     0x8000u /* intra_smoothing_disabled_flag (intrapred.cc:307)                       */
#define HM_PIC_CROSS_COMPONENT        0x10000u /* cross_component_prediction_enabled_flag (4:4:4): hm_tu.qpy of a chroma
                                                 record holds ResScaleVal (0, +-1, +-2, +-4, +-8); its residual gets
                                                 (ResScaleVal * luma residual) >> 3 added (transform.cc:251-267), also
                                                 when the record has no levels of its own (slice.cc:3797-3805)        */
#define HM_PIC_LARGE_TSKIP            0x20000u /* log2_max_transform_skip_block_size > 2: transform-skip blocks of 8x8 and
                                                 larger may occur (the common kernel's 8x8 path has no skip branch)  */
#define HM_PIC_RARE_SYNTAX            (HM_PIC_SCALING_LIST | HM_PIC_PCMF | HM_PIC_LOSSLESS_CUS | HM_PIC_444 | HM_PIC_TS_ROTATION | \
                                       HM_PIC_IMPLICIT_RDPCM | HM_PIC_NO_INTRA_SMOOTHING | HM_PIC_CROSS_COMPONENT | HM_PIC_LARGE_TSKIP) /* pictures that
                                                 need the kernel variant of the reconstruction with the rare paths   */

/* ScalingFactor tables of a picture with scaling lists (transform.cc:509-533): one byte per coefficient position
 * x + nT * y.  Matrices of intra blocks only: 4x4 cIdx 0..2 at 0, 8x8 at 48, 16x16 at 240, 32x32 (luma) at 1008. */
#define HM_SCALING_BYTES 2048
#define HM_SCALING_OFFSET(log2_size, cidx) \
  ((log2_size) == 2 ? 16 * (cidx) : ((log2_size) == 3 ? 48 + 64 * (cidx) : ((log2_size) == 4 ? 240 + 256 * (cidx) : 1008)))

typedef struct hm_pic {
  uint32_t magic;
  uint32_t total_bytes;        /* size of the whole blob                                     */
  uint16_t width, height;      /* luma samples (pic_width/height_in_luma_samples)            */
  uint16_t crop_left, crop_right, crop_top, crop_bottom; /* conformance window, luma samples */
  uint8_t  chroma_format;      /* 0 mono, 1 4:2:0, 2 4:2:2, 3 4:4:4                          */
  uint8_t  bit_depth_y, bit_depth_c;
  uint8_t  log2_ctb;           /* 4..6                                                       */
  uint8_t  log2_min_tb;        /* 2..5                                                       */
  uint8_t  log2_min_cb;
  uint8_t  log2_sao_offset_scale_y, log2_sao_offset_scale_c; /* already applied to offsets   */
  uint16_t ctb_w, ctb_h;       /* picture size in CTBs                                       */
  int8_t   pps_cb_qp_offset, pps_cr_qp_offset; /* chroma deblocking QpC (deblock.cc:1695)   */
  uint8_t  pcm_loop_filter_disabled;
  uint8_t  reserved0;
  uint32_t flags;              /* HM_PIC_*                                                   */
  /* VUI colour description (vui.cc:93-97 defaults: 2,2,2, full_range 0)                     */
  uint8_t  colour_primaries, transfer_characteristics, matrix_coeffs, full_range;
  uint32_t n_slices, n_ctbs, n_tus, n_coeffs;
  uint32_t off_slices, off_ctbs, off_tus, off_coeffs;
  uint32_t off_scaling;        /* HM_SCALING_BYTES of scaling factors when HM_PIC_SCALING_LIST  */
  uint32_t concealed_ctbs;     /* CTBs the data did not define (damaged slice data parsed with HM_PARSE_CONCEAL): written as
                                  plain intra CTUs without a residual; 0 for every intact picture                           */
  uint32_t first_concealed_ctb;/* raster address + 1 of the first of them (0: none)                                         */
} hm_pic;

/* one entry per slice (not slice segment) */
typedef struct hm_slice {
  uint32_t slice_addr;          /* SliceAddrRS                                               */
  int8_t   beta_offset_div2, tc_offset_div2;
  uint8_t  deblocking_disabled; /* slice_deblocking_filter_disabled_flag                     */
  uint8_t  sao_luma, sao_chroma;
  uint8_t  lf_across_slices;    /* slice_loop_filter_across_slices_enabled_flag              */
  int8_t   slice_qp;
  uint8_t  reserved;
} hm_slice;

/* hm_ctb.flags */
#define HM_CTB_DEBLOCK_LEFT  0x01u /* filter this CTB's left picture-internal edge (deblock.cc:160-196) */
#define HM_CTB_DEBLOCK_TOP   0x02u /* ... top edge                                                       */
#define HM_CTB_CODED         0x04u /* CTB was present in the bitstream                                   */
#define HM_CTB_DEBLOCK_OFF   0x08u /* slice_deblocking_filter_disabled_flag of the CTB's slice           */
#define HM_CTB_SAO_LUMA      0x10u /* slice_sao_luma_flag of the CTB's slice                            */
#define HM_CTB_SAO_CHROMA    0x20u /* slice_sao_chroma_flag of the CTB's slice                          */
#define HM_CTB_LOSSLESS      0x40u /* the CTB holds a PCM or cu_transquant_bypass coding unit (image.h:190 of the reference:
                                      has_pcm_or_cu_transquant_bypass, which sends SAO down its per-sample path)      */

/* hm_ctb.nb_avail */
#define HM_CTB_NB_NW 0x01u
#define HM_CTB_NB_N  0x02u
#define HM_CTB_NB_NE 0x04u
#define HM_CTB_NB_W  0x08u

/* SAO parameters of one colour component of one CTB (slice.h:457-465, offsets pre-scaled
 * by log2_sao_offset_scale as slice.cc:2996-3007 does) */
typedef struct hm_sao {
  uint8_t type;          /* 0 off, 1 band, 2 edge (SaoTypeIdx)                                */
  uint8_t eo_class;      /* SaoEoClass 0..3                                                   */
  uint8_t band_position; /* sao_band_position 0..31                                           */
  int8_t  offset[4];
  uint8_t reserved;
} hm_sao;

typedef struct hm_ctb {
  uint32_t tu_first;     /* index of the first (luma) hm_tu of this CTB; within a CTB row tu_first of CTB i+1
                            == tu_first + tu_count of CTB i (see "record order")                      */
  uint16_t tu_count;
  uint16_t slice_idx;    /* index into hm_slice[]                                             */
  uint8_t  flags;        /* HM_CTB_*                                                          */
  uint8_t  sao_nb_mask;  /* luma: bit k set: neighbour CTB k usable by SAO edge offset; k = 0..7 =
                            NW,N,NE,W,E,SW,S,SE (sao.cc:336-424 slice/tile tests)             */
  uint8_t  sao_nb_mask_c;/* the same for the chroma planes.  It differs from the luma mask because the reference
                            looks up "the slice of this CTB" with the CTB's *chroma* sample position
                            (sao.cc:291: get_SliceHeader(xC, yC), xC = xCtb * nSW), i.e. for 4:2:0 it takes
                            the slice address of CTB (x/2, y/2) - quirk Q13, reproduced                   */
  uint8_t  sao_ring_c;   /* chroma: 1 = the samples of the CTB's outer ring (first / last row / column: the
                            only ones the reference tests, sao.cc:366) may use a neighbour sample inside
                            their own CTB; 0 when the mis-addressed slice of Q13 forbids it (luma: always 1) */
  hm_sao   sao[3];
  uint32_t tu_first_c;   /* HM_PIC_SPLIT_CHAINS: the CTB's chroma records (contiguous along the CTB row)  */
  uint16_t tu_count_c;
  uint8_t  nb_avail;     /* HM_CTB_NB_*: which neighbouring CTBs intra prediction may read (inside the picture, earlier in
                            tile scan, same slice, same tile: intrapred.h:536-618 of the reference); the CTBs to the right
                            and below are always later                                                             */
  uint8_t  reserved;
  uint32_t coeff_first;  /* HM_PIC_SPLIT_CHAINS: index of the first level of the CTB's first luma record ...     */
  uint32_t coeff_first_c;/* ... and of its first chroma record (levels lie in record order)                       */
} hm_ctb; /* 52 bytes = HM_CTB_DWORDS dwords */
#define HM_CTB_DWORDS 13

/* hm_tu.info */
#define HM_TU_LOG2_MASK 0x07u  /* log2 block size 2..5 (component samples)                    */
#define HM_TU_CIDX_SHIFT 3     /* bits 3-4: colour component                                  */
#define HM_TU_CBF     0x20u    /* residual present                                            */
#define HM_TU_TSKIP   0x40u    /* transform_skip_flag                                         */
#define HM_TU_AVAIL_TL 0x80u   /* top-left neighbour sample available                         */

/* hm_tu.pred_mode: IntraPredMode in bits 0-5, and */
#define HM_TU_MODE_MASK   0x3Fu
#define HM_TU_MODE_BYPASS 0x40u /* cu_transquant_bypass_flag: the levels are the residual (transform.cc:431-449)     */
#define HM_TU_MODE_PCM    0x80u /* pcm_flag: no prediction; the n_coeff = nT*nT "levels" are the samples, already
                                   shifted to the bit depth, in raster order (slice.cc:4462-4504)                  */

/* One reconstruction step: predict block, then add its residual.  x,y are relative to the CTB
 * origin in samples of the component (chroma: chroma samples).  avail_* count available
 * neighbour samples (intrapred.h:620-667: picture bounds, slice, tile and z-order already
 * applied, clamped to the picture). */
typedef struct hm_tu {
  uint8_t  x, y;
  uint8_t  info;
  uint8_t  pred_mode;    /* IntraPredMode 0..34 (chroma: final mode, 4:2:2 remap applied) | HM_TU_MODE_* */
  uint8_t  qp;           /* qP of (8.6.1) incl. QpBdOffset: the dequantisation QP             */
  int8_t   qpy;          /* luma: QpY of the coding unit (deblocking); chroma in HM_PIC_CROSS_COMPONENT pictures: ResScaleVal */
  uint16_t n_coeff;      /* number of hm_coeff pairs                                          */
  uint32_t coeff_first;  /* index into hm_coeff[]                                             */
  uint8_t  avail_left, avail_bottom_left, avail_top, avail_top_right;
} hm_tu; /* 16 bytes */

/* The same step in pictures with HM_PIC_SPLIT_CHAINS (no rare syntax: no PCM / bypass flags, levels in record order):
 *   pos        x >> 2 | (y >> 2) << 4           (block positions are multiples of 4 samples of their plane)
 *   info       as in hm_tu, without HM_TU_AVAIL_TL; bit 7 instead: HM_TU6_NEXT_TO_LAST, the record's CTB is the last but
 *              one of its row
 *   pred_mode  as in hm_tu, without flags
 *   qp         the dequantisation qP as in hm_tu; for a luma record also QpY + QpBdOffsetY of its coding unit (the
 *              deblocking filter's QpY = qp - 6 * (bit_depth_y - 8)), also when the record has no residual
 *   count      n_coeff (bits 0-10) | hm_ctb.nb_avail of the record's CTB << 11 | HM_TU6_LAST_COLUMN if that CTB is the
 *              last of its row (what a lane that holds a record needs to know of its CTB, so that it need not find it)
 * Neighbour availability is not stored: hm_avail.h derives it from the record's position, the four neighbour bits of
 * its CTB and the picture size, exactly as the parser derives the hm_tu fields. */
#define HM_TU6_COUNT_MASK  0x07FFu
#define HM_TU6_NB_SHIFT    11
#define HM_TU6_LAST_COLUMN 0x8000u
#define HM_TU6_NEXT_TO_LAST 0x80u /* in info (the only columns in which the picture's right edge can cut an above-right run) */
typedef struct hm_tu6 {
  uint8_t  pos;
  uint8_t  info;
  uint8_t  pred_mode;
  uint8_t  qp;
  uint16_t count;
} hm_tu6; /* 6 bytes */

typedef struct hm_coeff {
  uint16_t pos;          /* x + y * nT (coeffPos, slice.cc:3694-3696)                         */
  int16_t  value;        /* TransCoeffLevel before scaling                                    */
} hm_coeff;

#ifdef __cplusplus
}
#endif
#endif /* HM_STREAM_H */
