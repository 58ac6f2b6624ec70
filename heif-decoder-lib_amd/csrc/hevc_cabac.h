// hevc_cabac.h — CABAC arithmetic decoding engine and context set (ITU-T H.265 §9.3), host side.
// The reference keeps entropy decoding on the CPU (libde265 cabac.cc / contextmodel.cc); so do we.
// Tables are the normative ones of the standard (Table 9-46, 9-47; initValues of Tables 9-5..9-37,
// initType 0 = I slices, the only type on the still-image path).
#ifndef HM_HEVC_CABAC_H
#define HM_HEVC_CABAC_H

#include <cstdint>
#include <cstring>

#include "hevc_types.h"

namespace hm {

// ---- context indices ---------------------------------------------------------------------
enum Ctx : int {
  CTX_SAO_MERGE = 0,
  CTX_SAO_TYPE = 1,
  CTX_SPLIT_CU = 2,        // 3
  CTX_TQ_BYPASS = 5,
  CTX_PART_MODE = 6,
  CTX_PREV_INTRA = 7,
  CTX_CHROMA_PRED = 8,
  CTX_SPLIT_TF = 9,        // 3
  CTX_CBF_LUMA = 12,       // 2
  CTX_CBF_CHROMA = 14,     // 5
  CTX_CU_QP_DELTA = 19,    // 2
  CTX_TSKIP = 21,          // 2 (luma, chroma)
  CTX_LAST_X = 23,         // 18
  CTX_LAST_Y = 41,         // 18
  CTX_CSBF = 59,           // 4
  CTX_SIG = 63,            // 44
  CTX_GT1 = 107,           // 24
  CTX_GT2 = 131,           // 6
  // range extensions (contextmodel.cc:350-353 of the reference: all initialised from the value 154)
  CTX_CHROMA_QP_OFFSET_FLAG = 137,
  CTX_CHROMA_QP_OFFSET_IDX = 138,
  CTX_RES_SCALE_ABS = 139, // 8: 4 * (cIdx - 1) + binIdx
  CTX_RES_SCALE_SIGN = 147, // 2
  CTX_COUNT = 149
};

// (16-bit states on purpose: a store through a character type may alias anything, the arithmetic decoder's range / value /
//  bit count included - the compiler then writes them to memory and reads them back around every bin; a 16-bit store
//  cannot alias their 32-bit type, so they stay in registers across the bins of an inlined loop)
typedef uint16_t ctx_state;
struct ContextSet {
  ctx_state state[CTX_COUNT]; // (pStateIdx << 1) | valMps
};

namespace cabac_tables {
static const uint8_t kInit[CTX_COUNT] = {
    153,                                   // sao_merge
    200,                                   // sao_type_idx
    139, 141, 157,                         // split_cu_flag
    154,                                   // cu_transquant_bypass_flag
    184,                                   // part_mode
    184,                                   // prev_intra_luma_pred_flag
    63,                                    // intra_chroma_pred_mode
    153, 138, 138,                         // split_transform_flag
    111, 141,                              // cbf_luma
    94, 138, 182, 154, 154,                // cbf_cb / cbf_cr
    154, 154,                              // cu_qp_delta_abs
    139, 139,                              // transform_skip_flag
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, // last x
    110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63, // last y
    91, 171, 134, 141,                     // coded_sub_block_flag
    111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107,
    125, 141, 179, 153, 125, 140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111, 141, 111, // sig
    140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197, // gt1
    138, 153, 136, 167, 152, 152,          // gt2
    154, 154,                              // cu_chroma_qp_offset_flag / idx
    154, 154, 154, 154, 154, 154, 154, 154, // log2_res_scale_abs_plus1
    154, 154,                              // res_scale_sign_flag
};

static const uint8_t kRangeTabLps[64][4] = {
    {128, 176, 208, 240}, {128, 167, 197, 227}, {128, 158, 187, 216}, {123, 150, 178, 205}, {116, 142, 169, 195},
    {111, 135, 160, 185}, {105, 128, 152, 175}, {100, 122, 144, 166}, {95, 116, 137, 158},  {90, 110, 130, 150},
    {85, 104, 123, 142},  {81, 99, 117, 135},   {77, 94, 111, 128},   {73, 89, 105, 122},   {69, 85, 100, 116},
    {66, 80, 95, 110},    {62, 76, 90, 104},    {59, 72, 86, 99},     {56, 69, 81, 94},     {53, 65, 77, 89},
    {51, 62, 73, 85},     {48, 59, 69, 80},     {46, 56, 66, 76},     {43, 53, 63, 72},     {41, 50, 59, 69},
    {39, 48, 56, 65},     {37, 45, 54, 62},     {35, 43, 51, 59},     {33, 41, 48, 56},     {32, 39, 46, 53},
    {30, 37, 43, 50},     {29, 35, 41, 48},     {27, 33, 39, 45},     {26, 31, 37, 43},     {24, 30, 35, 41},
    {23, 28, 33, 39},     {22, 27, 32, 37},     {21, 26, 30, 35},     {20, 24, 29, 33},     {19, 23, 27, 31},
    {18, 22, 26, 30},     {17, 21, 25, 28},     {16, 20, 23, 27},     {15, 19, 22, 25},     {14, 18, 21, 24},
    {14, 17, 20, 23},     {13, 16, 19, 22},     {12, 15, 18, 21},     {12, 14, 17, 20},     {11, 14, 16, 19},
    {11, 13, 15, 18},     {10, 12, 15, 17},     {10, 12, 14, 16},     {9, 11, 13, 15},      {9, 11, 12, 14},
    {8, 10, 12, 14},      {8, 9, 11, 13},       {7, 9, 11, 12},       {7, 9, 10, 12},       {7, 8, 10, 11},
    {6, 8, 9, 11},        {6, 7, 9, 10},        {6, 7, 8, 9},         {2, 2, 2, 2}};

static constexpr uint8_t kTransIdxLps[64] = {0,  0,  1,  2,  2,  4,  4,  5,  6,  7,  8,  9,  9,  11, 11, 12,
                                         13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                         24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33,
                                         33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63};

// state after a bin: [0] most probable symbol decoded, [1] least probable (Table 9-47 transIdxMps / transIdxLps; valMps
// flips when an LPS arrives in state 0), indexed by (pStateIdx << 1) | valMps
struct NextState {
  uint8_t v[2][128];
  constexpr NextState() : v()
  {
    for (int c = 0; c < 128; c++) {
      const int st = c >> 1, mps = c & 1;
      v[0][c] = (uint8_t)(((st < 62 ? st + 1 : st) << 1) | mps);
      v[1][c] = (uint8_t)((kTransIdxLps[st] << 1) | (st == 0 ? !mps : mps));
    }
  }
};
static constexpr NextState kNextState{};
} // namespace cabac_tables

// §9.3.2.2: initialisation of context variables
inline void init_contexts(ContextSet& cs, int slice_qp_y)
{
  int qp = slice_qp_y < 0 ? 0 : (slice_qp_y > 51 ? 51 : slice_qp_y);
  for (int i = 0; i < CTX_COUNT; i++) {
    int iv = cabac_tables::kInit[i];
    int m = (iv >> 4) * 5 - 45;
    int n = ((iv & 15) << 3) - 16;
    int pre = ((m * qp) >> 4) + n;
    pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
    int mps = pre <= 63 ? 0 : 1;
    int st = mps ? pre - 64 : 63 - pre;
    cs.state[i] = (ctx_state)((st << 1) | mps);
  }
}

// ---- arithmetic decoder (§9.3.4.3) -----------------------------------------------------------
// The 9-bit offset (ivlOffset) sits in bits 62..54 of a 64-bit window (bit 63 is room for the doubling of a bypass
// bin), `bits_` look-ahead bits follow it; the window
// is refilled 32 bits at a time when the look-ahead runs out (about once per 32 stream bits: a well-predicted branch -
// a byte-at-a-time refill is taken on every fourth bin or so and mispredicts).  The read position the standard defines
// (9 bits at start-up, one more per renormalisation shift) is P = 8 * pos_ - bits_; the last bit read before a
// terminating bin of value 1 is the stop bit, so the next syntax (the next sub-stream, PCM samples) starts at byte
// ceil(P / 8).
class CabacDecoder {
 public:
  void init(const uint8_t* p, const uint8_t* end)
  {
    base_ = p;
    len_ = (size_t)(end - p);
    pos_ = 0;
    range_ = 510;
    value_ = 0;
    bits_ = -9;
    refill();
  }
  // 9.3.2.5: the first nine bits are the offset, and 510 / 511 are not allowed - with an offset >= range every bin would
  // decode from a state the arithmetic never reaches (multi-bin bypass reads leave their value range, the excess doubles with
  // every renormalisation and wraps at the register width: nothing a second implementation reproduces).  Callers refuse.
  bool bad_start() const { return (value_ >> kOffsetShift) >= 510u; }
  const uint8_t* position() const { return base_ + ((ptrdiff_t)pos_ - (bits_ >> 3)); }
  // a decision has depended on bits behind the end of the data: the read position P (see above; it only grows) lies past
  // the last bit.  No valid slice gets there - the last bit a terminating bin of value 1 reads is the stop bit -, and what a
  // damaged one decodes to from there on is not defined by the data: this decoder reads zeros, the reference (its 16-bit
  // refill, cabac.cc:276-296) whatever follows the NAL in its buffer.
  bool overrun() const { return (ptrdiff_t)(8 * pos_) - bits_ > (ptrdiff_t)(8 * len_); }

  // One context-coded bin (9.3.4.3.2).  The MPS / LPS decision of a well-compressed stream is as good as random, so it
  // is taken with masks instead of a branch (a mispredicted branch per bin costs more than the arithmetic of both
  // paths); the state transition is one table look-up on (LPS?, state); renormalisation shifts by the leading zeros.
  __attribute__((always_inline)) inline int decode_bin(ctx_state& ctx)
  {
    const uint32_t c = ctx;
    const uint32_t lps_range = cabac_tables::kRangeTabLps[c >> 1][(range_ >> 6) & 3];
    uint32_t r = range_ - lps_range;
    const uint64_t scaled = (uint64_t)r << kOffsetShift;
    const uint32_t lps = value_ >= scaled ? 1u : 0u;
    const uint32_t mask = 0u - lps;
    value_ -= scaled & (uint64_t)(int64_t)(int32_t)mask;
    r += (lps_range - r) & mask;
    ctx = cabac_tables::kNextState.v[lps][c];
    const int shift = __builtin_clz(r) - 23; // r in [6, 510]: 0 for r >= 256
    range_ = r << shift;
    value_ <<= shift;
    bits_ -= shift;
    if (__builtin_expect(bits_ < 0, 0)) refill();
    return (int)((c & 1u) ^ lps);
  }

  __attribute__((always_inline)) inline int decode_bypass()
  {
    value_ <<= 1;
    if (__builtin_expect(--bits_ < 0, 0)) refill();
    const uint64_t scaled = (uint64_t)range_ << kOffsetShift;
    const uint64_t keep = value_ < scaled ? 0 : ~(uint64_t)0; // all ones when value >= scaled (bin 1)
    value_ -= scaled & keep;
    return (int)((uint32_t)keep & 1u);
  }

  // n bypass bins at once, first bin in the most significant bit (n <= 16).  Bypass bins leave the range alone: bin after
  // bin is the long division of the offset, extended by the next n stream bits, by the range - done here as one 32-bit
  // division (25-bit dividend, 9-bit divisor) instead of n dependent compare-subtract steps.
  __attribute__((always_inline)) inline uint32_t decode_bypass_bits(int n)
  {
    if (n <= 0) return 0;
    if (bits_ < n) refill();
    const int s = kOffsetShift - n;
    const uint32_t q = (uint32_t)(value_ >> s) / range_;
    value_ = (value_ - ((uint64_t)(q * range_) << s)) << n;
    bits_ -= n;
    return q;
  }

  inline int decode_terminate()
  {
    range_ -= 2;
    const uint64_t scaled = (uint64_t)range_ << kOffsetShift;
    if (value_ >= scaled) return 1;
    if (range_ < 256u) {
      range_ <<= 1;
      value_ <<= 1;
      if (--bits_ < 0) refill();
    }
    return 0;
  }

 private:
  static constexpr int kOffsetShift = 54;
  // 32 more bits below the valid ones (bits_ <= 22 here); bytes behind the end of the data read as zero
  __attribute__((always_inline)) inline void refill()
  {
    uint32_t w;
    if (__builtin_expect(pos_ + 4 <= len_, 1)) {
      std::memcpy(&w, base_ + pos_, 4);
      w = __builtin_bswap32(w);
    }
    else w = tail_word(base_, pos_, len_);
    pos_ += 4;
    value_ |= (uint64_t)w << (kOffsetShift - 32 - bits_);
    bits_ += 32;
  }
  __attribute__((noinline)) static uint32_t tail_word(const uint8_t* base, size_t pos, size_t len)
  {
    uint32_t w = 0;
    for (size_t k = 0; k < 4; k++) w = (w << 8) | (pos + k < len ? base[pos + k] : 0u);
    return w;
  }
  const uint8_t* base_ = nullptr;
  size_t len_ = 0, pos_ = 0;
  uint64_t value_ = 0;
  uint32_t range_ = 510;
  int bits_ = 0;
};

} // namespace hm
#endif
