// hevc_parse.cpp — host front end: NAL de-framing, parameter sets, CABAC slice-data decoding of one
// coded HEVC intra picture into the GPU command stream (include/hm_stream.h).
//
// Replaces, on the host, what the reference does in libheif/plugins/decoder_libde265.cc:269-303
// (NAL de-framing of the `[u32 BE length][NAL]...` byte string the plugin receives through
// push_data) and in libde265's decctx.cc:1209-1290 / slice.cc (parameter sets, slice header,
// slice data).  Reconstruction itself is NOT done here - that is the GPU's job.
#include <cstdlib>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <memory>

#include "hevc_syntax.h"
#include "hm_knobs.h"
#include "hm_internal.h"

namespace hm {
// slice segments whose sub-streams were entropy-decoded side by side since the library was loaded (tests): [0] WPP rows, [1] tiles
static std::atomic<long> g_parallel_segments[2];

// entropy-coder adaptor handed to SliceWalker: plain CABAC decoding, kinds ignored
class DecoderEC {
 public:
  DecoderEC(const uint8_t* begin, const uint8_t* end) : cur_(begin), end_(end) {}
  inline int bin(int ctx, int, int) { return dec_.decode_bin(cs_.state[ctx]); }
  inline int bypass(int, int) { return dec_.decode_bypass(); }
  // n bypass bins (index idx0, idx0 + step, ...; the first one most significant), n <= 16 per division
  inline uint32_t bypass_bits(int, int, int, int n)
  {
    uint32_t v = 0;
    for (; n > 16; n -= 16) v = (v << 16) | dec_.decode_bypass_bits(16);
    return (v << n) | dec_.decode_bypass_bits(n);
  }
  inline int terminate(int) { int b = dec_.decode_terminate(); check(); return b; }
  ContextSet& contexts() { return cs_; }
  const uint8_t* position() const { return dec_.position(); } // after a terminating bin: the next byte-aligned position
  inline int pcm_flag() { return terminate(0); }
  // PCM samples start at the byte the arithmetic decoder's read pointer stands on after the terminating bin, and the
  // decoder restarts on the next byte boundary behind them (slice.cc:4506-4536, cabac.cc:658-677 of the reference)
  void pcm_begin() { raw_ = dec_.position(); raw_bit_ = 0; }
  uint32_t pcm_bits(int n)
  {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) {
      if (raw_ >= end_) throw ParseError(HM_ERR_BITSTREAM, "PCM samples run past the end of the slice data");
      v = (v << 1) | ((*raw_ >> (7 - raw_bit_)) & 1u);
      if (++raw_bit_ == 8) { raw_bit_ = 0; raw_++; }
    }
    return v;
  }
  void pcm_end()
  {
    if (raw_bit_) { raw_bit_ = 0; raw_++; }
    if (end_ - raw_ < 2) throw ParseError(HM_ERR_BITSTREAM, "slice data ends inside a PCM coding unit");
    dec_.init(raw_, end_);
    if (dec_.bad_start()) throw ParseError(HM_ERR_BITSTREAM, "arithmetic decoder restarts with an offset of 510 or 511 (9.3.2.5)");
  }
  void start_substream()
  {
    if (started_) cur_ = dec_.position();
    if (end_ - cur_ < 2) throw ParseError(HM_ERR_BITSTREAM, "slice data truncated");
    dec_.init(cur_, end_);
    if (dec_.bad_start()) throw ParseError(HM_ERR_BITSTREAM, "arithmetic decoder starts with an offset of 510 or 511 (9.3.2.5)");
    started_ = true;
  }
  // (checked once per CTB, behind its terminating bin: the read position only grows)
  void check() const
  {
    if (dec_.overrun()) throw ParseError(HM_ERR_BITSTREAM, "CABAC read past the end of the slice data");
  }

 private:
  CabacDecoder dec_;
  ContextSet cs_;
  const uint8_t* cur_;
  const uint8_t* end_;
  const uint8_t* raw_ = nullptr; // PCM sample reader
  int raw_bit_ = 0;
  bool started_ = false;
};

namespace {

// One per host thread, reused for picture after picture (thread_local in hm_hevc_parse): a 512x512 tile needs about
// 1 MB of tables and vectors, and allocating / freeing that per tile from a crew of 100+ threads costs more than the
// entropy decode itself (page faults, heap trimming, TLB shoot-downs: measured 1.4 -> 6 ms per tile at 64 threads).
struct Decoder {
  SPS sps[16];
  PPS pps[64];
  PictureState pic;
  std::vector<uint8_t> rbsp; // unescaped NAL, reused
  std::vector<uint32_t> removed; // positions of its emulation prevention bytes in the escaped NAL
  int record_order = 0;      // hm_parse_options.record_order
  // Concealment (HM_PARSE_CONCEAL; r05).  The reference keeps a picture whose slice data is damaged: libde265 notes the error, marks
  // the slice as processed (decctx.cc:876-995) and hands the picture out (decoder_libde265.cc:311-336) - with the CTBs it did not
  // decode holding whatever its image memory held.  What the data DEFINES is the CTBs in front of the error; this parser decodes
  // exactly those and writes every other CTB - the rest of the damaged slice segment, the segments that depended on it, CTBs no
  // segment covers - as plain intra CTUs without a residual (hevc_syntax.h: ConcealEC): a valid command stream, a picture, and
  // hm_pic.concealed_ctbs says how much of it is made up.  Without the option such streams are refused (HM_ERR_BITSTREAM) as before.
  bool conceal = false;
  int concealed = 0;                       // CTBs written by conceal_range and still standing (ctb_concealed)
  std::vector<uint8_t> ctb_concealed;      // per CTB (raster address): written by conceal_range, not taken over by a later segment
  // Something in the picture's slice data has been seen to be wrong (a caught error, a gap, segments that overlap).  Only then may
  // slice NALs behind the picture's last CTB still belong to it (a damaged segment ran on to the end; the segments behind it take
  // their CTBs over): behind an INTACT picture they are another picture's and are ignored, as without concealment.
  bool damage_seen = false;
  bool later_picture = false;              // a first_slice_segment_in_pic behind the finished picture: nothing after it is ours
  int last_slice_idx = -1;                 // of the last slice segment that was parsed (concealed CTBs join its slice)
  bool want_split = false;   // the current picture's records go out as split chains (unless it turns out to use rare syntax)
  int threads = 1;           // > 1: slice segments with WPP entry points are parsed row-parallel (parse_rows_parallel)
  void start_stream()
  {
    for (SPS& s : sps) s.valid = false;
    for (PPS& p : pps) p.valid = false;
    pic_started = have_prev_sh = picture_done = false;
    cur_sps = nullptr; cur_pps = nullptr;
    next_ts = 0;
    concealed = 0; last_slice_idx = -1;
    ctb_concealed.clear();
    damage_seen = later_picture = false;
  }
  bool pic_started = false;
  const SPS* cur_sps = nullptr;
  const PPS* cur_pps = nullptr;
  SliceHeader prev_sh;
  bool have_prev_sh = false;
  int next_ts = 0;
  bool picture_done = false;

  void handle_nal(const uint8_t* p, size_t n)
  {
    if (n < 2) return;
    const int nal_type = (p[0] >> 1) & 0x3F;
    const int layer = ((p[0] & 1) << 5) | (p[1] >> 3);
    if (layer != 0) return; // only the base layer
    if (nal_type == 33 || nal_type == 34 || nal_type <= 21) {
      unescape_nal(p, n, rbsp, threads > 1 ? &removed : nullptr);
      if (rbsp.size() < 2) return;
      BitReader br(rbsp.data() + 2, rbsp.size() - 2);
      if (nal_type == 33) {
        SPS s;
        parse_sps(br, s);
        if (pic_started && cur_sps == &sps[s.sps_id]) throw ParseError(HM_ERR_BITSTREAM, "SPS replaced inside a picture");
        // the scan tables of a PPS are derived from the SPS that was active when the PPS was parsed (pps.cc:585-800
        // of the reference does the same and re-derives on activation): a new SPS with this id makes them stale
        for (PPS& q : pps)
          if (q.valid && q.sps_id == s.sps_id) q.valid = false;
        sps[s.sps_id] = s;
      }
      else if (nal_type == 34) {
        PPS q;
        parse_pps(br, q, sps);
        if (pic_started && cur_pps == &pps[q.pps_id]) throw ParseError(HM_ERR_BITSTREAM, "PPS replaced inside a picture");
        pps[q.pps_id] = q;
      }
      else if (nal_type <= 9 || (nal_type >= 16 && nal_type <= 21)) {
        // a still-image item holds one picture; ignore anything after it (concealing, and damage was seen: a damaged segment may have
        // run on to the picture's last CTB - the segments behind it still take their CTBs over, up to the next picture's first one)
        if (picture_done && !(conceal && damage_seen && !later_picture)) return;
        if (!conceal) slice_nal(br, nal_type, rbsp);
        else {
          // a slice segment whose HEADER is damaged is dropped like the reference drops it (decctx.cc:639-712: the NAL's decoding
          // error ends that NAL only); its CTBs are concealed when the next segment - or the picture's end - shows the gap
          try { slice_nal(br, nal_type, rbsp); }
          catch (const ParseError& e) {
            if (e.status != HM_ERR_BITSTREAM || !pic_started) throw;
            pic.dep_ok = false;
            damage_seen = true;
          }
        }
      }
    }
    // VPS (32), AUD, SEI, EOS ...: nothing the reconstruction needs
  }

  void slice_nal(BitReader& br, int nal_type, const std::vector<uint8_t>& rbsp)
  {
    SliceHeader sh;
    parse_slice_header(br, nal_type, sps, pps, have_prev_sh ? &prev_sh : nullptr, sh);
    if (picture_done && sh.first_slice_segment_in_pic) { later_picture = true; return; } // (another picture behind the item's one: ignored, with all its segments)
    const PPS& p = pps[sh.pps_id];
    const SPS& s = sps[p.sps_id];
    // every table the slice walker indexes with CTB / minimum-TB addresses must have the active SPS's size
    if (p.CtbAddrRStoTS.size() != (size_t)s.ctb_w * s.ctb_h || p.CtbAddrTStoRS.size() != p.CtbAddrRStoTS.size() ||
        p.TileIdRS.size() != p.CtbAddrRStoTS.size())
      throw ParseError(HM_ERR_BITSTREAM, "PPS scan tables do not match the active SPS");
    if (sh.first_slice_segment_in_pic) {
      if (pic_started) throw ParseError(HM_ERR_UNSUPPORTED, "more than one coded picture in the item");
      check_supported(s, p);
      cur_sps = &s;
      cur_pps = &p;
      {
        const bool interleaved = hm_knob(HM_KNOB_STREAM_INTERLEAVED) == 1;
        const int force_class = hm_knob(HM_KNOB_QUAD_CLASS); // (A/B measurements: 1 split chains for all classes, 0 for none)
        want_split = !interleaved && (record_order == HM_RECORDS_SPLIT || (record_order == HM_RECORDS_AUTO && (force_class >= 0 ? force_class != 0 : quad_class(s))));
      }
      pic.reset(s, p, want_split);
      pic_started = true;
      next_ts = 0;
    }
    else if (!pic_started) throw ParseError(HM_ERR_BITSTREAM, "slice segment without a first_slice_segment_in_pic");
    if (&p != cur_pps) throw ParseError(HM_ERR_UNSUPPORTED, "PPS changes inside a picture");

    if (sh.dependent) sh.SliceAddrRS = prev_sh.SliceAddrRS;
    int slice_idx;
    if (!sh.dependent) {
      hm_slice hs;
      std::memset(&hs, 0, sizeof(hs));
      hs.slice_addr = (uint32_t)sh.SliceAddrRS;
      hs.beta_offset_div2 = (int8_t)sh.beta_offset_div2;
      hs.tc_offset_div2 = (int8_t)sh.tc_offset_div2;
      hs.deblocking_disabled = sh.deblocking_disabled;
      hs.sao_luma = sh.sao_luma;
      hs.sao_chroma = sh.sao_chroma;
      hs.lf_across_slices = sh.lf_across_slices;
      hs.slice_qp = (int8_t)sh.SliceQPY;
      pic.slices.push_back(hs);
    }
    if (pic.slices.empty()) throw ParseError(HM_ERR_BITSTREAM, "dependent slice segment first in picture");
    slice_idx = (int)pic.slices.size() - 1;

    const int start_ts = p.CtbAddrRStoTS[sh.slice_segment_address];
    if (start_ts != next_ts) {
      // (concealing: the CTBs between the last decoded one and this segment's first - the rest of a damaged segment, a lost one -
      //  join the slice in front of them)
      if (conceal && start_ts > next_ts && have_prev_sh && last_slice_idx >= 0) conceal_range(next_ts, start_ts, prev_sh, last_slice_idx);
      else if (conceal && start_ts < next_ts && !sh.dependent) take_back(start_ts); // (overlapping segments: no conforming stream has them)
      else throw ParseError(HM_ERR_BITSTREAM, "slice segments out of order or CTBs missing");
    }
    // slice data starts right after the header in the unescaped payload (+2 for the NAL header)
    const uint8_t* begin = rbsp.data() + 2 + sh.data_byte_offset;
    const uint8_t* end = rbsp.data() + rbsp.size();
    // (StatCoeff and CuQpOffsetCb / Cr run on from one CTB row into the next in the reference: such segments stay serial)
    if (threads > 1 && p.entropy_coding_sync && !p.tiles_enabled && !sh.dependent && sh.num_entry_points > 0 &&
        (sh.slice_segment_address % s.ctb_w) == 0 && !s.persistent_rice && !sh.cu_chroma_qp_offset_enabled)
      next_ts = parse_rows_parallel(sh, slice_idx, start_ts, begin, end);
    else if (threads > 1 && p.tiles_enabled && !p.entropy_coding_sync && !sh.dependent && sh.num_entry_points > 0 && p.num_tile_rows > 1 &&
             !s.persistent_rice && !sh.cu_chroma_qp_offset_enabled)
      next_ts = parse_tiles_parallel(sh, slice_idx, start_ts, begin, end);
    else {
      DecoderEC ec(begin, end);
      SliceWalker<DecoderEC> walker(ec, pic, sh, slice_idx);
      if (!conceal) next_ts = walker.decode_slice_segment(start_ts);
      else {
        try { next_ts = walker.decode_slice_segment(start_ts); }
        catch (const ParseError& e) {
          if (e.status != HM_ERR_BITSTREAM) throw;
          // the segment ends in front of the CTU it failed in; that CTU is taken back, the tables a dependent segment would inherit
          // are gone.  How far the segment would have reached shows with the next one (the gap above) or at the picture's end.
          next_ts = walker.current_ts();
          walker.discard_current_ctu();
          pic.dep_ok = false;
          damage_seen = true;
          pic.uses_pcm |= walker.uses_pcm();
          pic.uses_tq_bypass |= walker.uses_tq_bypass();
        }
      }
    }
    prev_sh = sh;
    have_prev_sh = true;
    last_slice_idx = slice_idx;
    if (next_ts == s.ctb_w * s.ctb_h) picture_done = true;
  }

  // Concealing: a slice segment that starts INSIDE what the segment in front of it has written - a damaged segment runs on behind
  // its real end until the arithmetic decoder notices - takes those CTBs over, as in the reference, where every segment decodes into
  // its own CTBs whatever stood there (decctx.cc: decode_slice_unit_sequential).  The CTBs [ts0, next_ts), the last ones written, are
  // taken back in reverse order: their records and levels are the tails of their lists.
  void take_back(int ts0)
  {
    const PPS& p = *cur_pps;
    const SPS& s = *cur_sps;
    for (int ts = next_ts - 1; ts >= ts0; ts--) {
      const int rs = p.CtbAddrTStoRS[ts];
      hm_ctb& c = pic.ctbs[rs];
      if (!(c.flags & HM_CTB_CODED)) continue;
      if (pic.direct) {
        PictureState::RowChains& R = pic.rows[(size_t)(rs / s.ctb_w)];
        R.tu[0].resize(c.tu_first); R.tu[1].resize(c.tu_first_c);
        R.lv[0].resize(c.coeff_first); R.lv[1].resize(c.coeff_first_c);
        c.tu_count = c.tu_count_c = 0;
      }
      else {
        pic.ctb_tus[rs].clear();
        pic.coeffs.resize(pic.ctb_coeff_mark[rs]);
      }
      c.flags &= (uint8_t)~HM_CTB_CODED;
      __atomic_store_n(&pic.ctb_slice_addr[rs], -1, __ATOMIC_RELAXED);
      if (!ctb_concealed.empty() && ctb_concealed[(size_t)rs]) { ctb_concealed[(size_t)rs] = 0; concealed--; } // (no longer made up: the segment that takes it over defines it)
    }
    next_ts = ts0;
    picture_done = false;
    pic.dep_ok = false;
    damage_seen = true;
  }

  // CTBs [from_ts, to_ts) in tile scan as concealed CTUs of slice `slice_idx` (header sh)
  void conceal_range(int from_ts, int to_ts, const SliceHeader& sh, int slice_idx)
  {
    const PPS& p = *cur_pps;
    ConcealEC cec;
    SliceWalker<ConcealEC> w(cec, pic, sh, slice_idx);
    if (ctb_concealed.size() != pic.ctbs.size()) ctb_concealed.assign(pic.ctbs.size(), 0);
    for (int ts = from_ts; ts < to_ts; ts++) {
      w.decode_ctu(ts);
      ctb_concealed[(size_t)p.CtbAddrTStoRS[ts]] = 1;
      concealed++;
    }
    next_ts = to_ts;
    damage_seen = true;
  }

  // Wavefront-parallel parse of one slice segment whose header carries an entry point per CTB row (WPP, no tiles): the
  // rows are independent sub-streams except for the context tables handed down after the second CTB of a row
  // (9.3.1 / slice.cc:5004-5083 of the reference) and the neighbour data of the row above, so row r may run two CTBs
  // behind row r - 1 - the schedule of the reference's own WPP threads (decctx.cc:1004-1116).  Every row gets its own
  // arithmetic decoder, walker, QP predictor state and level list; `threads` workers take the rows in order.
  // The serial parser never looks at the entry point offsets (a sub-stream starts where the previous one's arithmetic
  // decoder stopped); this path needs them and therefore checks them: a row that does not end exactly where the next
  // one is said to begin makes the whole stream fall back to the serial parse (Inconsistent), as does any error - the
  // serial parse then reports it in decoding order.
  struct Inconsistent {};
  int parse_rows_parallel(const SliceHeader& sh, int slice_idx, int start_ts, const uint8_t* begin, const uint8_t* end)
  {
    const SPS& s = *cur_sps;
    const int W = s.ctb_w, row0 = sh.slice_segment_address / W, n_rows = sh.num_entry_points + 1;
    // sub-stream starts in the unescaped payload: the offsets count the bytes of the escaped NAL (7.4.7.1)
    const size_t data_unesc = (size_t)(begin - rbsp.data());
    auto escaped_of = [&](size_t u) { size_t k = 0; while (k < removed.size() && removed[k] <= u + k) k++; return u + k; };
    auto unescaped_of = [&](size_t e) { size_t k = 0; while (k < removed.size() && removed[k] < e) k++; return e - k; };
    std::vector<const uint8_t*> start((size_t)n_rows + 1);
    size_t e = escaped_of(data_unesc);
    start[0] = begin;
    for (int k = 1; k < n_rows; k++) {
      e += sh.entry_point_offset[k - 1];
      const size_t u = unescaped_of(e);
      if (u > rbsp.size() || rbsp.data() + u < start[k - 1]) throw Inconsistent();
      start[k] = rbsp.data() + u;
    }
    start[n_rows] = end;

    struct alignas(128) Row { // (a row's private state and the counters its neighbour spins on: separate cache lines)
      std::vector<hm_coeff> coeffs;
      PictureState::QpState qs;
      int ended_slice = 0;
      bool uses_pcm = false, uses_tq = false;
      alignas(128) std::atomic<int> done{0};      // CTBs finished
      alignas(128) std::atomic<int> ctx_ready{0}; // wpp_ctx[row] stored
    };
    std::vector<Row> rows((size_t)n_rows);
    std::atomic<int> next_row{0};
    std::atomic<bool> failed{false};
    std::exception_ptr first_error;
    std::mutex error_lock;
    const int N = W * s.ctb_h;

    auto parse_row = [&](int k) {
      const int row = row0 + k;
      Row& R = rows[(size_t)k];
      DecoderEC ec(start[k], start[k + 1]);
      SliceWalker<DecoderEC> walker(ec, pic, sh, slice_idx);
      R.qs.last_qpy_prev_qg = R.qs.current_qpy = sh.SliceQPY; // (the first quantisation group of a WPP row predicts from the slice QP)
      walker.use_private_state(&R.qs, &R.coeffs);
      auto wait_for = [&](std::atomic<int>& v, int need) {
        for (int spins = 0; v.load(std::memory_order_acquire) < need; spins++) {
          if (failed.load(std::memory_order_relaxed)) throw Inconsistent();
          if (spins > 64) std::this_thread::yield();
        }
      };
      if (k == 0) init_contexts(ec.contexts(), sh.SliceQPY);
      else if (W < 2) init_contexts(ec.contexts(), sh.SliceQPY); // (import_wpp_row: a picture one CTB wide)
      else {
        wait_for(rows[(size_t)k - 1].ctx_ready, 1);
        ec.contexts() = pic.wpp_ctx[(size_t)row - 1];
        pic.wpp_ok[(size_t)row - 1] = 0; // taken (import_wpp_row)
      }
      ec.start_substream();
      for (int x = 0; x < W; x++) {
        if (k > 0) wait_for(rows[(size_t)k - 1].done, std::min(x + 2, W));
        const int ts = row * W + x; // (no tiles: tile scan == raster scan)
        walker.decode_ctu(ts);
        if (x == 1 && row < s.ctb_h - 1) {
          pic.wpp_ctx[(size_t)row] = ec.contexts();
          pic.wpp_ok[(size_t)row] = 1; // (read by the row below behind ctx_ready, or by a later slice segment)
          R.ctx_ready.store(1, std::memory_order_release);
        }
        const int end_of_slice = ec.terminate(ts + 1 == N ? 1 : -1);
        R.done.store(x + 1, std::memory_order_release);
        if (end_of_slice) {
          if (k != n_rows - 1) throw Inconsistent(); // the slice ends before its last entry point
          if (cur_pps->dependent_slice_segments_enabled) { pic.dep_ctx = ec.contexts(); pic.dep_ok = true; }
          R.ended_slice = ts + 1;
          break;
        }
        if (ts + 1 >= N) throw ParseError(HM_ERR_BITSTREAM, "missing end_of_slice_segment_flag");
        if (x == W - 1) {
          if (k == n_rows - 1) throw Inconsistent(); // the slice goes on behind its last entry point
          if (!ec.terminate(2)) throw ParseError(HM_ERR_BITSTREAM, "end_of_subset_one_bit not set");
          if (ec.position() != start[k + 1]) throw Inconsistent();
        }
      }
      if (W == 1 && row < s.ctb_h - 1) R.ctx_ready.store(1, std::memory_order_release);
      R.uses_pcm = walker.uses_pcm(); R.uses_tq = walker.uses_tq_bypass();
    };
    auto worker = [&]() {
      for (;;) {
        const int k = next_row.fetch_add(1);
        if (k >= n_rows || failed.load()) return;
        try { parse_row(k); }
        catch (...) {
          std::lock_guard<std::mutex> g(error_lock);
          if (!first_error) first_error = std::current_exception();
          failed.store(true);
          return;
        }
      }
    };
    const int n_workers = std::min(threads, n_rows);
    std::vector<std::thread> crew;
    try {
      for (int i = 1; i < n_workers; i++) crew.emplace_back(worker);
    }
    catch (...) { // no more threads to be had: the rows that were started finish, the serial parse takes over
      failed.store(true);
      for (std::thread& t : crew) t.join();
      throw Inconsistent();
    }
    worker();
    for (std::thread& t : crew) t.join();
    if (first_error) throw Inconsistent(); // (whatever it was: the serial parse finds it in decoding order)
    if (!rows.back().ended_slice) throw Inconsistent();
    // the rows' level lists behind the picture's, the records rebased
    for (int k = 0; k < n_rows; k++) {
      const uint32_t base = (uint32_t)pic.coeffs.size();
      Row& R = rows[(size_t)k];
      pic.coeffs.insert(pic.coeffs.end(), R.coeffs.begin(), R.coeffs.end());
      if (base)
        for (int x = 0; x < W; x++) {
          const size_t rs = (size_t)(row0 + k) * W + x;
          for (hm_tu& t : pic.ctb_tus[rs]) t.coeff_first += base;
          pic.ctb_coeff_mark[rs] += base; // (take_back cuts the picture's list there)
        }
      pic.uses_pcm |= R.uses_pcm;
      pic.uses_tq_bypass |= R.uses_tq;
    }
    pic.qs = rows.back().qs;
    g_parallel_segments[0]++;
    return rows.back().ended_slice;
  }

  // HEVC tiles in parallel: a slice segment whose header carries an entry point per tile (no WPP).  Tiles are
  // independent sub-streams - fresh context tables, no prediction and no QP history across their borders (the first
  // quantisation group of a tile predicts from the slice QP) -; what they share is where their records go: the chains of
  // a CTB row (hevc_syntax.h: PictureState::rows) take the records of every tile the row crosses, from left to right.
  // So the unit of work is a ROW of tiles: its tiles one after the other on one thread - exactly the serial order inside
  // the CTB rows it covers -, the rows of tiles side by side on up to `threads` threads.  Same command stream byte for
  // byte; anything that does not fit (entry points that do not match the sub-streams, a slice that ends early or goes
  // on behind its last entry point, any error) falls back to the serial parse, which reports errors in decoding order.
  int parse_tiles_parallel(const SliceHeader& sh, int slice_idx, int start_ts, const uint8_t* begin, const uint8_t* end)
  {
    const SPS& s = *cur_sps;
    const PPS& p = *cur_pps;
    const int W = s.ctb_w, N = W * s.ctb_h, n_sub = sh.num_entry_points + 1;
    if (start_ts > 0 && p.TileId[start_ts] == p.TileId[start_ts - 1]) throw Inconsistent(); // (the segment starts inside a tile)
    // tile-scan ranges of the sub-streams = tiles
    std::vector<int> ts0((size_t)n_sub + 1);
    {
      int ts = start_ts;
      for (int k = 0; k < n_sub; k++) {
        if (ts >= N) throw Inconsistent(); // more entry points than tiles left
        ts0[(size_t)k] = ts;
        const int tid = p.TileId[ts];
        while (ts < N && p.TileId[ts] == tid) ts++;
      }
      ts0[(size_t)n_sub] = ts;
    }
    // sub-stream starts in the unescaped payload (the offsets count bytes of the escaped NAL, 7.4.7.1)
    const size_t data_unesc = (size_t)(begin - rbsp.data());
    auto escaped_of = [&](size_t u) { size_t k = 0; while (k < removed.size() && removed[k] <= u + k) k++; return u + k; };
    auto unescaped_of = [&](size_t e) { size_t k = 0; while (k < removed.size() && removed[k] < e) k++; return e - k; };
    std::vector<const uint8_t*> start((size_t)n_sub + 1);
    size_t e = escaped_of(data_unesc);
    start[0] = begin;
    for (int k = 1; k < n_sub; k++) {
      e += sh.entry_point_offset[k - 1];
      const size_t u = unescaped_of(e);
      if (u > rbsp.size() || rbsp.data() + u < start[(size_t)k - 1]) throw Inconsistent();
      start[(size_t)k] = rbsp.data() + u;
    }
    start[(size_t)n_sub] = end;
    // the rows of tiles: consecutive sub-streams whose tiles have the same tile row
    struct alignas(128) Group {
      int first = 0, count = 0; // sub-streams
      std::vector<hm_coeff> coeffs;
      PictureState::QpState qs;
      int ended_slice = 0;
      bool uses_pcm = false, uses_tq = false;
    };
    std::vector<Group> groups;
    for (int k = 0; k < n_sub; k++) {
      const int trow = p.TileId[ts0[(size_t)k]] / p.num_tile_cols;
      if (groups.empty() || p.TileId[ts0[(size_t)groups.back().first]] / p.num_tile_cols != trow) { groups.emplace_back(); groups.back().first = k; }
      groups.back().count++;
    }
    const int n_groups = (int)groups.size();
    if (n_groups < 2) throw Inconsistent(); // (nothing to run side by side: the serial parse is the cheaper one)
    std::atomic<int> next_group{0};
    std::atomic<bool> failed{false};

    auto parse_group = [&](int gi) {
      Group& G = groups[(size_t)gi];
      for (int k = G.first; k < G.first + G.count; k++) {
        DecoderEC ec(start[(size_t)k], start[(size_t)k + 1]);
        SliceWalker<DecoderEC> walker(ec, pic, sh, slice_idx);
        G.qs.last_qpy_prev_qg = G.qs.current_qpy = sh.SliceQPY; // (the first quantisation group of a tile predicts from the slice QP)
        G.qs.cur_qg_x = G.qs.cur_qg_y = -1;
        walker.use_private_state(&G.qs, &G.coeffs);
        init_contexts(ec.contexts(), sh.SliceQPY); // every tile starts with fresh tables
        ec.start_substream();
        for (int ts = ts0[(size_t)k]; ts < ts0[(size_t)k + 1]; ts++) {
          if (failed.load(std::memory_order_relaxed)) throw Inconsistent();
          walker.decode_ctu(ts);
          const int end_of_slice = ec.terminate(ts + 1 == N ? 1 : -1);
          if (end_of_slice) {
            if (k != n_sub - 1 || ts + 1 != ts0[(size_t)k + 1]) throw Inconsistent(); // the slice ends before its last entry point / inside a tile
            if (cur_pps->dependent_slice_segments_enabled) { pic.dep_ctx = ec.contexts(); pic.dep_ok = true; }
            G.ended_slice = ts + 1;
            break;
          }
          if (ts + 1 >= N) throw ParseError(HM_ERR_BITSTREAM, "missing end_of_slice_segment_flag");
          if (ts + 1 == ts0[(size_t)k + 1]) { // the tile is done, the slice is not
            if (k == n_sub - 1) throw Inconsistent(); // the slice goes on behind its last entry point
            if (!ec.terminate(2)) throw ParseError(HM_ERR_BITSTREAM, "end_of_subset_one_bit not set");
            if (ec.position() != start[(size_t)k + 1]) throw Inconsistent();
          }
        }
        G.uses_pcm |= walker.uses_pcm(); G.uses_tq |= walker.uses_tq_bypass();
      }
    };
    auto worker = [&]() {
      for (;;) {
        const int gi = next_group.fetch_add(1);
        if (gi >= n_groups || failed.load()) return;
        try { parse_group(gi); }
        catch (...) { failed.store(true); return; } // (whatever it was: the serial parse finds it in decoding order)
      }
    };
    const int n_workers = std::min(threads, n_groups);
    std::vector<std::thread> crew;
    try {
      for (int i = 1; i < n_workers; i++) crew.emplace_back(worker);
    }
    catch (...) { // no more threads to be had: what was started finishes, the serial parse takes over
      failed.store(true);
      for (std::thread& t : crew) t.join();
      throw Inconsistent();
    }
    worker();
    for (std::thread& t : crew) t.join();
    if (failed.load() || !groups.back().ended_slice) throw Inconsistent();
    // the groups' level lists behind the picture's, the records rebased (split chains keep their levels per CTB row: nothing to do)
    for (int gi = 0; gi < n_groups; gi++) {
      Group& G = groups[(size_t)gi];
      const uint32_t base = (uint32_t)pic.coeffs.size();
      pic.coeffs.insert(pic.coeffs.end(), G.coeffs.begin(), G.coeffs.end());
      if (base)
        for (int ts = ts0[(size_t)G.first]; ts < ts0[(size_t)(G.first + G.count)]; ts++) {
          const size_t rs = (size_t)p.CtbAddrTStoRS[ts];
          for (hm_tu& t : pic.ctb_tus[rs]) t.coeff_first += base;
          pic.ctb_coeff_mark[rs] += base;
        }
      pic.uses_pcm |= G.uses_pcm;
      pic.uses_tq_bypass |= G.uses_tq;
    }
    pic.qs = groups.back().qs;
    g_parallel_segments[1]++;
    return groups.back().ended_slice;
  }

  static void check_supported(const SPS& s, const PPS& p)
  {
    if (s.unsupported_extension) throw ParseError(HM_ERR_UNSUPPORTED, "multilayer / 3D / screen-content extension");
    if (s.separate_colour_plane) throw ParseError(HM_ERR_UNSUPPORTED, "separate colour planes");
    if (s.chroma_format_idc != 0 && s.bit_depth_y != s.bit_depth_c) throw ParseError(HM_ERR_UNSUPPORTED, "different luma / chroma bit depth");
    if (s.bit_depth_y > 12) throw ParseError(HM_ERR_UNSUPPORTED, "bit depth above 12");
    // (the reference only warns about this combination and then predicts from mis-sized blocks, pps.cc:68-72)
    if (p.cross_component_prediction && s.chroma_format_idc != 3) throw ParseError(HM_ERR_UNSUPPORTED, "cross-component prediction outside 4:4:4");
    if (s.width > 16384 || s.height > 16384) throw ParseError(HM_ERR_UNSUPPORTED, "picture larger than 16384x16384");
  }

  // ---- finalisation: flatten into one blob (malloc'ed: handed to the caller) ---------------------
  uint8_t* finish(size_t* out_size)
  {
    if (!pic_started) throw ParseError(HM_ERR_BITSTREAM, "no coded picture in the data");
    const SPS& s = *cur_sps;
    const PPS& p = *cur_pps;
    const int N = s.ctb_w * s.ctb_h;
    if (!picture_done && conceal && have_prev_sh && last_slice_idx >= 0 && next_ts < N) { // (the picture's last CTBs: lost with a damaged segment)
      conceal_range(next_ts, N, prev_sh, last_slice_idx);
      picture_done = true;
    }
    if (!picture_done) throw ParseError(HM_ERR_BITSTREAM, "picture incomplete: missing slice segments");
    int first_concealed = -1; // the first CTB in decoding order that is still a concealed one: its raster address
    if (concealed > 0)
      for (int ts = 0; ts < N && first_concealed < 0; ts++)
        if (ctb_concealed[(size_t)p.CtbAddrTStoRS[ts]]) first_concealed = p.CtbAddrTStoRS[ts];

    // Record order (hm_stream.h): pictures without rare syntax get their luma and chroma records in separate lists,
    // row by row (the four-rows-per-wave kernel walks the two chains independently); the knob stream_interleaved = 1 (hm_debug_set;
    // A/B measurements of the one-row-per-wave kernel) keeps the decode order for every picture.
    const bool force_interleaved = hm_knob(HM_KNOB_STREAM_INTERLEAVED) == 1;
    const bool rare = s.scaling_list_enabled || (s.pcm_enabled && s.pcm_loop_filter_disabled) || p.transquant_bypass_enabled ||
                      pic.uses_pcm || pic.uses_tq_bypass || s.chroma_format_idc == 3 || s.transform_skip_rotation || s.implicit_rdpcm ||
                      s.intra_smoothing_disabled || p.cross_component_prediction ||
                      (p.transform_skip_enabled && p.log2_max_transform_skip_size > 2); // == HM_PIC_RARE_SYNTAX of the flags below
    const bool split = !rare && !force_interleaved && want_split;
    const bool direct = pic.direct; // the chains were written in their final form while parsing (hevc_syntax.h: PictureState::rows)
    if (direct && !split) throw ParseError(HM_ERR_INTERNAL, "direct chains of a picture with rare syntax");
    size_t n_tus = 0, n_levels = pic.coeffs.size();
    for (int i = 0; i < N; i++) {
      if (!(pic.ctbs[i].flags & HM_CTB_CODED)) throw ParseError(HM_ERR_BITSTREAM, "CTB not coded");
      if (pic.ctb_tus[i].size() > 65535) throw ParseError(HM_ERR_INTERNAL, "too many TUs in a CTB");
      n_tus += pic.ctb_tus[i].size();
    }
    if (direct) { // row-relative indices -> picture-wide: row 0 luma, row 0 chroma, row 1 luma, ...
      uint32_t tu_at = 0, lv_at = 0;
      for (int cy = 0; cy < s.ctb_h; cy++) {
        const PictureState::RowChains& R = pic.rows[(size_t)cy];
        const uint32_t tl = tu_at, ll = lv_at;
        tu_at += (uint32_t)R.tu[0].size(); lv_at += (uint32_t)R.lv[0].size();
        const uint32_t tc = tu_at, lc = lv_at;
        tu_at += (uint32_t)R.tu[1].size(); lv_at += (uint32_t)R.lv[1].size();
        for (int cx = 0; cx < s.ctb_w; cx++) {
          hm_ctb& c = pic.ctbs[(size_t)cx + (size_t)cy * s.ctb_w];
          c.tu_first += tl; c.coeff_first += ll; c.tu_first_c += tc; c.coeff_first_c += lc;
        }
      }
      n_tus = tu_at; n_levels = lv_at;
    }
    else {
      if (damage_seen) // (CTBs were taken back and written again: every record's levels must lie inside the list that goes out)
        for (int i = 0; i < N; i++)
          for (const hm_tu& t : pic.ctb_tus[i])
            if ((size_t)t.coeff_first + t.n_coeff > n_levels) throw ParseError(HM_ERR_INTERNAL, "levels of a record outside the picture's list");
      auto is_luma = [](const hm_tu& t) { return ((t.info >> HM_TU_CIDX_SHIFT) & 3) == 0; };
      size_t at = 0;
      for (int cy = 0; cy < s.ctb_h; cy++) {
        for (int pass = 0; pass < (split ? 2 : 1); pass++)
          for (int cx = 0; cx < s.ctb_w; cx++) {
            const int i = cx + cy * s.ctb_w;
            size_t cnt = pic.ctb_tus[i].size();
            if (split) {
              cnt = 0;
              for (const hm_tu& t : pic.ctb_tus[i]) cnt += is_luma(t) == (pass == 0);
            }
            if (pass == 0) { pic.ctbs[i].tu_first = (uint32_t)at; pic.ctbs[i].tu_count = (uint16_t)cnt; pic.ctbs[i].tu_first_c = 0; pic.ctbs[i].tu_count_c = 0; }
            else { pic.ctbs[i].tu_first_c = (uint32_t)at; pic.ctbs[i].tu_count_c = (uint16_t)cnt; }
            at += cnt;
          }
      }
    }
    // SAO neighbour masks (sao.cc:323-424 of the reference).  The reference's fast path (all neighbours inside the
    // picture usable, slice flags ignored) is taken per CTB: pps_loop_filter_across_slices && !tiles && the CTB holds no
    // PCM / transquant-bypass unit (sao.cc:323).  Otherwise every sample of the CTB's outer ring is tested against the
    // CTB that holds its neighbour sample - which may be the CTB itself - with "the slice of the current CTB" looked up
    // at the CTB's position in samples of the *component* (sao.cc:291), i.e. at the wrong CTB for sub-sampled chroma
    // (quirk Q13): chroma gets its own mask and a ring flag.
    static const int dx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
    const int csw = (s.chroma_format_idc == 1 || s.chroma_format_idc == 2) ? 1 : 0, csh = s.chroma_format_idc == 1 ? 1 : 0;
    for (int cy = 0; cy < s.ctb_h; cy++)
      for (int cx = 0; cx < s.ctb_w; cx++) {
        const int c = cx + cy * s.ctb_w;
        const bool fast = p.lf_across_slices && !p.tiles_enabled && !(pic.ctbs[c].flags & HM_CTB_LOSSLESS);
        const int so = pic.ctb_slice_addr[c];
        const bool lf_own = pic.slices[pic.ctbs[c].slice_idx].lf_across_slices != 0;
        // may a sample of CTB c use a neighbour sample lying in CTB nb, given the slice address the reference compares with
        auto usable = [&](int nb, int sq) {
          if (fast) return true;
          const int sa = pic.ctb_slice_addr[nb];
          if (sa < sq && !lf_own) return false;
          if (sa > sq && !pic.slices[pic.ctbs[nb].slice_idx].lf_across_slices) return false;
          if (!p.lf_across_tiles && p.TileIdRS[nb] != p.TileIdRS[c]) return false;
          return true;
        };
        const int sq_c = pic.ctb_slice_addr[(cx >> csw) + (cy >> csh) * s.ctb_w]; // sao.cc:291 with chroma coordinates
        uint8_t mask = 0, mask_c = 0;
        for (int k = 0; k < 8; k++) {
          const int nx = cx + dx[k], ny = cy + dy[k];
          if (nx < 0 || ny < 0 || nx >= s.ctb_w || ny >= s.ctb_h) continue;
          const int nb = nx + ny * s.ctb_w;
          if (usable(nb, so)) mask |= (uint8_t)(1u << k);
          if (usable(nb, sq_c)) mask_c |= (uint8_t)(1u << k);
        }
        pic.ctbs[c].sao_nb_mask = mask;
        pic.ctbs[c].sao_nb_mask_c = mask_c;
        pic.ctbs[c].sao_ring_c = usable(c, sq_c) ? 1 : 0;
      }

    auto align16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t off_slices = align16(sizeof(hm_pic));
    const size_t off_ctbs = align16(off_slices + pic.slices.size() * sizeof(hm_slice));
    const size_t off_tus = align16(off_ctbs + (size_t)N * sizeof(hm_ctb));
    const size_t tu_bytes = split ? sizeof(hm_tu6) : sizeof(hm_tu);
    const size_t off_coeffs = align16(off_tus + n_tus * tu_bytes);
    const size_t off_scaling = align16(off_coeffs + n_levels * sizeof(hm_coeff));
    const size_t total = off_scaling + (s.scaling_list_enabled ? (size_t)HM_SCALING_BYTES : 0);
    if (total > 0xFFFFFFFFu) throw ParseError(HM_ERR_INTERNAL, "command stream too large");
    struct Mem { // (the section gaps - at most 15 bytes each - are zeroed, the sections are copied straight in)
      uint8_t* p;
      uint8_t* data() const { return p; }
      ~Mem() { std::free(p); } // (an exception on the way: the blob goes back)
    } blob{(uint8_t*)std::malloc(total)};
    if (!blob.p) throw std::bad_alloc();
    auto zero_gap = [&](size_t from, size_t to) { if (to > from) std::memset(blob.p + from, 0, to - from); };
    zero_gap(sizeof(hm_pic), off_slices);
    zero_gap(off_slices + pic.slices.size() * sizeof(hm_slice), off_ctbs);
    zero_gap(off_ctbs + (size_t)N * sizeof(hm_ctb), off_tus);
    zero_gap(off_tus + n_tus * tu_bytes, off_coeffs);
    zero_gap(off_coeffs + n_levels * sizeof(hm_coeff), total);
    hm_pic h;
    std::memset(&h, 0, sizeof(h));
    h.magic = HM_STREAM_MAGIC;
    h.total_bytes = (uint32_t)total;
    h.width = (uint16_t)s.width;
    h.height = (uint16_t)s.height;
    h.crop_left = (uint16_t)s.conf_left; h.crop_right = (uint16_t)s.conf_right;
    h.crop_top = (uint16_t)s.conf_top; h.crop_bottom = (uint16_t)s.conf_bottom;
    h.chroma_format = (uint8_t)s.chroma_format_idc;
    h.bit_depth_y = (uint8_t)s.bit_depth_y;
    h.bit_depth_c = (uint8_t)s.bit_depth_c;
    h.log2_ctb = (uint8_t)s.log2_ctb;
    h.log2_min_tb = (uint8_t)s.log2_min_tb;
    h.log2_min_cb = (uint8_t)s.log2_min_cb;
    h.log2_sao_offset_scale_y = (uint8_t)p.log2_sao_offset_scale_luma;
    h.log2_sao_offset_scale_c = (uint8_t)p.log2_sao_offset_scale_chroma;
    h.ctb_w = (uint16_t)s.ctb_w;
    h.ctb_h = (uint16_t)s.ctb_h;
    h.pps_cb_qp_offset = (int8_t)p.cb_qp_offset;
    h.pps_cr_qp_offset = (int8_t)p.cr_qp_offset;
    h.pcm_loop_filter_disabled = s.pcm_loop_filter_disabled;
    h.concealed_ctbs = (uint32_t)concealed;
    h.first_concealed_ctb = first_concealed < 0 ? 0u : (uint32_t)first_concealed + 1u;
    if (concealed < 0 || (concealed > 0) != (first_concealed >= 0)) throw ParseError(HM_ERR_INTERNAL, "concealment bookkeeping");
    uint32_t flags = 0;
    if (s.strong_intra_smoothing) flags |= HM_PIC_STRONG_INTRA_SMOOTHING;
    if (s.sao_enabled) flags |= HM_PIC_SAO_ENABLED;
    for (const hm_slice& sl : pic.slices) {
      if (!sl.deblocking_disabled) flags |= HM_PIC_DEBLOCK_ANY;
      if (s.sao_enabled && (sl.sao_luma || sl.sao_chroma)) flags |= HM_PIC_SAO_ANY;
    }
    if (s.vui_colour_present) flags |= HM_PIC_HAS_VUI_COLOUR;
    if (p.sign_data_hiding) flags |= HM_PIC_SIGN_HIDING;
    if (p.tiles_enabled) flags |= HM_PIC_TILES;
    if (p.lf_across_tiles) flags |= HM_PIC_LF_ACROSS_TILES;
    if (s.scaling_list_enabled) flags |= HM_PIC_SCALING_LIST;
    if ((s.pcm_enabled && s.pcm_loop_filter_disabled) || p.transquant_bypass_enabled) flags |= HM_PIC_PCMF;
    if (s.chroma_format_idc == 3) flags |= HM_PIC_444;
    if (pic.uses_pcm || pic.uses_tq_bypass) flags |= HM_PIC_LOSSLESS_CUS;
    if (s.transform_skip_rotation) flags |= HM_PIC_TS_ROTATION;
    if (s.implicit_rdpcm) flags |= HM_PIC_IMPLICIT_RDPCM;
    if (s.intra_smoothing_disabled) flags |= HM_PIC_NO_INTRA_SMOOTHING;
    if (p.cross_component_prediction) flags |= HM_PIC_CROSS_COMPONENT;
    if (p.transform_skip_enabled && p.log2_max_transform_skip_size > 2) flags |= HM_PIC_LARGE_TSKIP;
    if (split) flags |= HM_PIC_SPLIT_CHAINS;
    if (rare != ((flags & HM_PIC_RARE_SYNTAX) != 0)) throw ParseError(HM_ERR_INTERNAL, "rare-syntax classification");
    h.flags = flags;
    h.colour_primaries = (uint8_t)s.colour_primaries;
    h.transfer_characteristics = (uint8_t)s.transfer_characteristics;
    h.matrix_coeffs = (uint8_t)s.matrix_coeffs;
    h.full_range = (uint8_t)s.video_full_range;
    h.n_slices = (uint32_t)pic.slices.size();
    h.n_ctbs = (uint32_t)N;
    h.n_tus = (uint32_t)n_tus;
    h.n_coeffs = (uint32_t)n_levels;
    h.off_slices = (uint32_t)off_slices;
    h.off_ctbs = (uint32_t)off_ctbs;
    h.off_tus = (uint32_t)off_tus;
    h.off_coeffs = (uint32_t)off_coeffs;
    if (s.scaling_list_enabled) { // the PPS lists when it carries its own, else the SPS lists / defaults (pps.cc:473-482)
      static_assert(ScalingFactors::kBytes == HM_SCALING_BYTES, "scaling table layout");
      h.off_scaling = (uint32_t)off_scaling;
      std::memcpy(blob.data() + off_scaling, (p.scaling_list_present ? p.scaling : s.scaling).f, HM_SCALING_BYTES);
    }
    std::memcpy(blob.data(), &h, sizeof(h));
    std::memcpy(blob.data() + off_slices, pic.slices.data(), pic.slices.size() * sizeof(hm_slice));
    if (!split) {
      hm_tu* const tp = reinterpret_cast<hm_tu*>(blob.data() + off_tus);
      for (int i = 0; i < N; i++)
        if (!pic.ctb_tus[i].empty()) std::memcpy(tp + pic.ctbs[i].tu_first, pic.ctb_tus[i].data(), pic.ctb_tus[i].size() * sizeof(hm_tu));
      std::memcpy(blob.data() + off_ctbs, pic.ctbs.data(), (size_t)N * sizeof(hm_ctb));
      if (!pic.coeffs.empty()) std::memcpy(blob.data() + off_coeffs, pic.coeffs.data(), pic.coeffs.size() * sizeof(hm_coeff));
    }
    else if (direct) {
      uint8_t* tp = blob.data() + off_tus;
      uint8_t* cp = blob.data() + off_coeffs;
      for (int cy = 0; cy < s.ctb_h; cy++)
        for (int k = 0; k < 2; k++) {
          const PictureState::RowChains& R = pic.rows[(size_t)cy];
          if (!R.tu[k].empty()) { std::memcpy(tp, R.tu[k].data(), R.tu[k].size() * sizeof(hm_tu6)); tp += R.tu[k].size() * sizeof(hm_tu6); }
          if (!R.lv[k].empty()) { std::memcpy(cp, R.lv[k].data(), R.lv[k].size() * sizeof(hm_coeff)); cp += R.lv[k].size() * sizeof(hm_coeff); }
        }
      std::memcpy(blob.data() + off_ctbs, pic.ctbs.data(), (size_t)N * sizeof(hm_ctb));
    }
    else {
      // compact records (hm_stream.h: hm_tu6) in chain order, the levels gathered into the order of the records
      hm_tu6* const tp = reinterpret_cast<hm_tu6*>(blob.data() + off_tus);
      hm_coeff* const cp = reinterpret_cast<hm_coeff*>(blob.data() + off_coeffs);
      uint32_t level_at = 0;
      const int qp_bd_offset_y = 6 * (s.bit_depth_y - 8);
      auto put = [&](hm_tu6* d, const hm_tu& t, uint32_t ctb_bits, uint32_t info_bits) {
        if ((t.x | t.y) & 3) throw ParseError(HM_ERR_INTERNAL, "block geometry not a multiple of 4");
        if (t.n_coeff > HM_TU6_COUNT_MASK || (t.pred_mode & ~HM_TU_MODE_MASK)) throw ParseError(HM_ERR_INTERNAL, "record does not fit the compact form");
        d->pos = (uint8_t)((t.x >> 2) | ((t.y >> 2) << 4));
        d->info = (uint8_t)((t.info & ~HM_TU_AVAIL_TL) | info_bits); d->pred_mode = t.pred_mode;
        // (a luma record's QP is QpY + QpBdOffsetY of its coding unit - what it was dequantised with if it has a residual)
        d->qp = ((t.info >> HM_TU_CIDX_SHIFT) & 3) == 0 ? (uint8_t)(t.qpy + qp_bd_offset_y) : t.qp;
        d->count = (uint16_t)(t.n_coeff | ctb_bits);
        if (t.n_coeff) std::memcpy(cp + level_at, pic.coeffs.data() + t.coeff_first, t.n_coeff * sizeof(hm_coeff));
        level_at += t.n_coeff;
      };
      // record order = CTB row by CTB row: the row's luma records, then its chroma records (the order of tu_first / tu_first_c)
      for (int cy = 0; cy < s.ctb_h; cy++)
        for (int pass = 0; pass < 2; pass++)
          for (int cx = 0; cx < s.ctb_w; cx++) {
            const int i = cx + cy * s.ctb_w;
            hm_tu6* d = tp + (pass == 0 ? pic.ctbs[i].tu_first : pic.ctbs[i].tu_first_c);
            (pass == 0 ? pic.ctbs[i].coeff_first : pic.ctbs[i].coeff_first_c) = level_at;
            const uint32_t ctb_bits = ((uint32_t)pic.ctbs[i].nb_avail << HM_TU6_NB_SHIFT) | (cx + 1 == s.ctb_w ? HM_TU6_LAST_COLUMN : 0u);
            for (const hm_tu& t : pic.ctb_tus[i])
              if ((((t.info >> HM_TU_CIDX_SHIFT) & 3) == 0) == (pass == 0)) put(d++, t, ctb_bits, cx + 2 == s.ctb_w ? HM_TU6_NEXT_TO_LAST : 0u);
          }
      if (level_at != pic.coeffs.size()) throw ParseError(HM_ERR_INTERNAL, "levels lost while reordering");
      std::memcpy(blob.data() + off_ctbs, pic.ctbs.data(), (size_t)N * sizeof(hm_ctb));
    }
    *out_size = total;
    uint8_t* const done = blob.p;
    blob.p = nullptr;
    return done;
  }
};

} // namespace
} // namespace hm

static_assert(sizeof(hm_tu) == 16, "hm_tu layout");
static_assert(sizeof(hm_tu6) == 6, "hm_tu6 layout");
static_assert(sizeof(hm_ctb) == 4 * HM_CTB_DWORDS, "hm_ctb layout");
static_assert(sizeof(hm_coeff) == 4, "hm_coeff layout");
static_assert(sizeof(hm_slice) == 12, "hm_slice layout");
static_assert(sizeof(hm_sao) == 8, "hm_sao layout");
static_assert(sizeof(hm_pic) % 4 == 0, "hm_pic layout");

static int hm_hevc_parse_run(const uint8_t* data, size_t size, int annexb, int threads, int record_order_and_flags, uint8_t** out_blob, size_t* out_size)
{
  const int record_order = record_order_and_flags & 0xFF;
  const bool conceal = (record_order_and_flags & HM_PARSE_CONCEAL) != 0;
#if defined(__BMI2__) || defined(__LZCNT__)
  // this translation unit is built with BMI / BMI2 / LZCNT (Makefile: HOST_ISA): a host without them gets an error, not SIGILL
  if (!hm_host_has_bmi2_lzcnt()) return hm_fail(HM_ERR_UNSUPPORTED, "this build of the entropy decoder needs BMI2 and LZCNT (rebuild with HOST_ISA=)");
#endif
  try {
    static thread_local std::unique_ptr<hm::Decoder> workspace;
    if (!workspace) workspace = std::make_unique<hm::Decoder>();
    hm::Decoder* dec = workspace.get();
    dec->threads = threads;
    dec->record_order = record_order;
    dec->conceal = conceal;
    dec->start_stream();
    if (annexb) {
      // split at 00 00 01 start codes
      size_t i = 0;
      auto find_sc = [&](size_t from) -> size_t {
        for (size_t k = from; k + 3 <= size; k++)
          if (data[k] == 0 && data[k + 1] == 0 && data[k + 2] == 1) return k;
        return size;
      };
      i = find_sc(0);
      while (i < size) {
        const size_t start = i + 3;
        size_t next = find_sc(start);
        size_t end = next;
        while (end > start && data[end - 1] == 0) end--; // trailing_zero_8bits
        if (end > start) dec->handle_nal(data + start, end - start);
        i = next;
      }
    }
    else {
      // [u32 big-endian length][NAL] records (decoder_libde265.cc:269-303)
      size_t p = 0;
      while (p < size) {
        if (p + 4 > size) return hm_fail_detail(HM_ERR_BITSTREAM, HM_DETAIL_END_OF_DATA, "truncated NAL length field");
        const uint32_t n = ((uint32_t)data[p] << 24) | ((uint32_t)data[p + 1] << 16) | ((uint32_t)data[p + 2] << 8) | data[p + 3];
        p += 4;
        if (n > size - p) return hm_fail_detail(HM_ERR_BITSTREAM, HM_DETAIL_END_OF_DATA, "NAL length exceeds the data");
        dec->handle_nal(data + p, n);
        p += n;
      }
    }
    *out_blob = dec->finish(out_size);
    return HM_OK;
  }
  catch (const hm::ParseError& e) {
    return hm_fail(e.status, "%s", e.what());
  }
  catch (const std::bad_alloc&) {
    return hm_fail(HM_ERR_NOMEM, "out of memory");
  }
  catch (const std::exception& e) {
    return hm_fail(HM_ERR_INTERNAL, "%s", e.what());
  }
}

extern "C" {

int hm_hevc_parse(const uint8_t* data, size_t size, int annexb, uint8_t** out_blob, size_t* out_size)
{
  return hm_hevc_parse_mt(data, size, annexb, 1, out_blob, out_size);
}

int hm_hevc_parse_mt(const uint8_t* data, size_t size, int annexb, int threads, uint8_t** out_blob, size_t* out_size)
{
  hm_parse_options o;
  o.annexb = annexb; o.threads = threads; o.record_order = HM_RECORDS_AUTO;
  return hm_hevc_parse_opts(data, size, &o, out_blob, out_size);
}

// The same with up to `threads` host threads for one picture: slice segments coded with wavefront parallel processing
// (an entry point per CTB row) are entropy-decoded row-parallel, two CTBs apart (the reference: decctx.cc:1004-1116).
// Same command stream byte for byte; streams whose entry points do not match their sub-streams, and streams with
// errors, are parsed again serially.
int hm_hevc_parse_opts(const uint8_t* data, size_t size, const hm_parse_options* opts, uint8_t** out_blob, size_t* out_size)
{
  if (!data || !opts || !out_blob || !out_size) return hm_fail(HM_ERR_INVALID_ARG, "null argument");
  if ((opts->record_order & ~HM_PARSE_CONCEAL) < HM_RECORDS_AUTO || (opts->record_order & ~HM_PARSE_CONCEAL) > HM_RECORDS_DECODE_ORDER)
    return hm_fail(HM_ERR_INVALID_ARG, "record_order %d", opts->record_order);
  const int annexb = opts->annexb, threads = opts->threads, order = opts->record_order;
  *out_blob = nullptr;
  *out_size = 0;
  if (threads > 1) {
    try {
      if (hm_hevc_parse_run(data, size, annexb, threads, order, out_blob, out_size) == HM_OK) return HM_OK;
    }
    catch (const hm::Decoder::Inconsistent&) {
    }
    // entry points that do not match the sub-streams, or an error: the serial parse decides (and words the error)
    if (*out_blob) { std::free(*out_blob); *out_blob = nullptr; }
    *out_size = 0;
  }
  return hm_hevc_parse_run(data, size, annexb, 1, order, out_blob, out_size);
}

} // extern "C"

extern "C" {

// (test hook, hm_internal.h) slice segments parsed with their WPP rows (which = 0) / their rows of tiles (1) side by side
long hm_parse_parallel_segments(int which) { return which == 0 || which == 1 ? hm::g_parallel_segments[which].load() : -1; }

// (test hook, hm_internal.h) the arithmetic decoder alone: runs a script of bins over `data` with all contexts
// initialised for slice QP `qp` - op >= 0: a context-coded bin with that context index, -1: a bypass bin, -2: a
// terminating bin, -(n + 2), n = 1..32: n bypass bins in one read - and writes the value of every op to out[]; returns the
// byte position the standard's read pointer corresponds to after the last op (see CabacDecoder), -1 for a bad script
long hm_test_cabac_script(const uint8_t* data, size_t size, int qp, const int32_t* ops, int n_ops, uint32_t* out)
{
  if (!data || !ops || !out || size < 2) return -1;
  hm::ContextSet cs;
  hm::init_contexts(cs, qp);
  hm::CabacDecoder d;
  d.init(data, data + size);
  for (int i = 0; i < n_ops; i++) {
    const int op = ops[i];
    if (op >= hm::CTX_COUNT || op < -34) return -1;
    if (op >= 0) out[i] = (uint32_t)d.decode_bin(cs.state[op]);
    else if (op == -1) out[i] = (uint32_t)d.decode_bypass();
    else if (op == -2) out[i] = (uint32_t)d.decode_terminate();
    else {
      int n = -op - 2;
      uint32_t v = 0;
      for (; n > 16; n -= 16) v = (v << 16) | d.decode_bypass_bits(16);
      out[i] = (v << n) | d.decode_bypass_bits(n);
    }
  }
  return (long)(d.position() - data);
}

void hm_free(void* p) { std::free(p); }

} // extern "C"
