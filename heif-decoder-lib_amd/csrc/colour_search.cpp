// colour_search.cpp — which chain of colour operations the reference would run, decided the way the reference decides it.
//
// convert_colorspace() (libheif/color-conversion/colorconversion.cc:487-596) does not look a chain up: it SEARCHES one,
// cheapest path first, over "colour states" (colourspace, chroma format, alpha, sample depth and, for YCbCr, the nclx),
// with every registered operation as an edge generator (construct_pipeline, colorconversion.cc:266-420).  Several
// requests have more than one cheapest chain whose arithmetic differs (float op before or after the depth change, with
// or without an alpha plane in the picture), and which one wins depends on the order of the operation pool and on how
// the search keeps its frontier.  Hard-coding the outcomes case by case stops being right the moment a corner is
// missed, so this file restates the search itself: the same states, the same edges in the same order, the same
// frontier discipline - and hands the resulting op sequence to colour_host.cpp as a small execution plan for the
// fused kernels.  oracle/pipeline_search.py is the independent restatement the tests hold this one against.
//
// Operations are data here (a table of predicates and state edits), not classes: OpKind + one switch.
#include <cstring>
#include <vector>

#include "hm_colour_plan.h"

namespace {

enum Space : int8_t { SP_YCBCR = 0, SP_RGB = 1, SP_MONO = 2 };
// chroma / interleaved layouts, numbered like enum heif_chroma
enum Layout : int8_t { L_MONO = 0, L_420 = 1, L_422 = 2, L_444 = 3, L_RGB = 10, L_RGBA = 11, L_RRGGBB_BE = 12, L_RRGGBBAA_BE = 13, L_RRGGBB_LE = 14, L_RRGGBBAA_LE = 15 };

struct Profile { int matrix = 6, primaries = 1, transfer = 13; bool full = true; }; // a fresh color_profile_nclx: sRGB (nclx.cc:316-323)
struct CState {
  Space space = SP_YCBCR;
  int layout = L_420;
  bool alpha = false;
  int bits = 8;
  Profile p;
};

// ColorState::operator== (colorconversion.cc:146-166): the profile only counts for YCbCr states
bool equal(const CState& a, const CState& b)
{
  if (a.space != b.space || a.layout != b.layout || a.alpha != b.alpha || a.bits != b.bits) return false;
  if (a.space != SP_YCBCR) return true;
  return a.p.full == b.p.full && a.p.matrix == b.p.matrix && a.p.primaries == b.p.primaries;
}

constexpr int COST_TRIVIAL = 1, COST_OPTIMIZED = 6, COST_PLAIN = 11; // colorconversion.h:95-101

struct Options { int down = 2 /* average */, up = 2 /* bilinear */; bool only_preferred = false; }; // heif.cc:1080-1082

struct Edge { CState to; int cost; };

bool planar_yuv_layout(int l) { return l == L_MONO || l == L_420 || l == L_422 || l == L_444; }
bool matrix_without_ycbcr(int m) { return m == 0 || m == 8 || m == 11 || m == 14; }
bool wide_interleaved(int l) { return l == L_RRGGBB_BE || l == L_RRGGBB_LE || l == L_RRGGBBAA_BE || l == L_RRGGBBAA_LE; }

CState rgb_state(int layout, bool alpha, int bits)
{
  CState s;
  s.space = SP_RGB; s.layout = layout; s.alpha = alpha; s.bits = bits;
  return s;
}

// The edges one operation offers from `in` towards `target`.  The order of the operations is the order of the
// reference's pool (colorconversion.cc:218-255, built without libyuv / libsharpyuv like the oracle build): it decides
// ties.  Each case cites the state_after_conversion it restates.
void edges_of(int op, const CState& in, const CState& target, const Options& o, std::vector<Edge>& out)
{
  const bool nn_refused = in.layout != L_444 && o.up != 1 && o.only_preferred; // "this Op only implements nearest-neighbor"
  const bool nn_down_refused = target.layout != L_444 && o.down != 1 && o.only_preferred;
  switch (op) {
    case HM_OP_RGB_TO_RGB24_32: // rgb2rgb.cc:29-63
      if (in.space == SP_RGB && in.layout == L_444 && in.bits == 8) {
        out.push_back({rgb_state(L_RGBA, true, 8), COST_PLAIN});
        out.push_back({rgb_state(L_RGB, false, 8), COST_PLAIN});
      }
      break;
    case HM_OP_RGB24_32_TO_RGB: // rgb2rgb.cc:519-546
      if (in.space == SP_RGB && (in.layout == L_RGB || in.layout == L_RGBA) && in.bits == 8)
        out.push_back({rgb_state(L_444, target.alpha, in.bits), COST_PLAIN});
      break;
    case HM_OP_YCBCR_TO_RGB_16:
    case HM_OP_YCBCR_TO_RGB_8: { // yuv2rgb.cc:30-76
      const bool wide = op == HM_OP_YCBCR_TO_RGB_16;
      if (nn_refused) break;
      if (in.space != SP_YCBCR || !(in.layout == L_444 || in.layout == L_422 || in.layout == L_420)) break;
      if (in.p.matrix == 11 || in.p.matrix == 14) break;
      if ((in.bits != 8) != wide) break;
      out.push_back({rgb_state(L_444, in.alpha, in.bits), COST_PLAIN});
      break;
    }
    case HM_OP_YCBCR420_TO_RGB24: // yuv2rgb.cc:261-303
      if (nn_refused || in.space != SP_YCBCR || in.layout != L_420 || in.bits != 8 || in.alpha) break;
      if (matrix_without_ycbcr(in.p.matrix) || !in.p.full) break;
      out.push_back({rgb_state(L_RGB, false, 8), COST_PLAIN});
      break;
    case HM_OP_YCBCR420_TO_RGB32: // yuv2rgb.cc:370-413
      if (nn_refused || in.space != SP_YCBCR || in.layout != L_420 || in.bits != 8) break;
      if (matrix_without_ycbcr(in.p.matrix) || !in.p.full) break;
      out.push_back({rgb_state(L_RGBA, true, 8), COST_PLAIN});
      break;
    case HM_OP_YCBCR420_TO_RRGGBBAA: // yuv2rgb.cc:499-547
      if (nn_refused || in.space != SP_YCBCR || in.layout != L_420 || in.bits == 8) break;
      if (matrix_without_ycbcr(in.p.matrix)) break;
      out.push_back({rgb_state(in.alpha ? L_RRGGBBAA_LE : L_RRGGBB_LE, in.alpha, in.bits), COST_PLAIN});
      out.push_back({rgb_state(in.alpha ? L_RRGGBBAA_BE : L_RRGGBB_BE, in.alpha, in.bits), COST_PLAIN});
      break;
    case HM_OP_RGB_HDR_TO_RRGGBBAA_BE: // rgb2rgb.cc:147-186
    case HM_OP_RGB_TO_RRGGBBAA_BE:     // rgb2rgb.cc:276-315
      if (in.space != SP_RGB || in.layout != L_444) break;
      if ((op == HM_OP_RGB_HDR_TO_RRGGBBAA_BE) != (in.bits != 8)) break;
      if (!in.alpha) out.push_back({rgb_state(L_RRGGBB_BE, false, in.bits), COST_PLAIN});
      out.push_back({rgb_state(L_RRGGBBAA_BE, true, in.bits), COST_PLAIN});
      break;
    case HM_OP_MONO_TO_YCBCR420: // monochrome.cc:26-49
      if (in.space == SP_MONO && in.layout == L_MONO) {
        CState s; // (the op sets no profile: a fresh one)
        s.space = SP_YCBCR; s.layout = L_420; s.alpha = in.alpha; s.bits = in.bits;
        out.push_back({s, COST_OPTIMIZED});
      }
      break;
    case HM_OP_MONO_TO_RGB24_32: // monochrome.cc:160-198
      if (in.space != SP_MONO || in.layout != L_MONO || in.bits != 8) break;
      if (!in.alpha) out.push_back({rgb_state(L_RGB, false, 8), COST_PLAIN});
      out.push_back({rgb_state(L_RGBA, true, 8), COST_PLAIN});
      break;
    case HM_OP_SWAP_ENDIANNESS: // rgb2rgb.cc:614-673
      if (in.space != SP_RGB || !wide_interleaved(in.layout)) break;
      switch (in.layout) {
        case L_RRGGBB_LE: out.push_back({rgb_state(L_RRGGBB_BE, false, in.bits), COST_PLAIN}); break;
        case L_RRGGBB_BE: out.push_back({rgb_state(L_RRGGBB_LE, false, in.bits), COST_PLAIN}); break;
        case L_RRGGBBAA_LE: out.push_back({rgb_state(L_RRGGBBAA_BE, true, in.bits), COST_PLAIN}); break;
        default: out.push_back({rgb_state(L_RRGGBBAA_LE, true, in.bits), COST_PLAIN}); break;
      }
      break;
    case HM_OP_RRGGBBAA_BE_TO_RGB_HDR: // rgb2rgb.cc:405-433
      if (in.space == SP_RGB && (in.layout == L_RRGGBB_BE || in.layout == L_RRGGBBAA_BE) && in.bits != 8)
        out.push_back({rgb_state(L_444, target.alpha, in.bits), COST_PLAIN});
      break;
    case HM_OP_RGB24_32_TO_YCBCR: { // rgb2yuv.cc:473-518
      if (nn_down_refused) break;
      if (in.space != SP_RGB || !(in.layout == L_RGB || in.layout == L_RGBA)) break;
      if (!(target.layout == L_420 || target.layout == L_422 || target.layout == L_444)) break;
      if (matrix_without_ycbcr(target.p.matrix)) break;
      CState s; s.space = SP_YCBCR; s.layout = target.layout; s.alpha = target.alpha; s.bits = 8; s.p = target.p;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_RGB_TO_YCBCR_8:
    case HM_OP_RGB_TO_YCBCR_16: { // rgb2yuv.cc:32-85
      const bool wide = op == HM_OP_RGB_TO_YCBCR_16;
      if ((in.bits != 8) != wide) break;
      if (in.space != SP_RGB || in.layout != L_444) break;
      if (target.p.matrix == 8 || target.p.matrix == 11 || target.p.matrix == 14) break;
      CState s; s.space = SP_YCBCR; s.alpha = in.alpha; s.bits = in.bits; s.p = target.p;
      s.layout = (target.layout != L_444 && (o.down == 1 || !o.only_preferred)) ? target.layout : (int)L_444;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_RRGGBBXX_HDR_TO_YCBCR420: { // rgb2yuv.cc:281-331
      if (nn_down_refused) break;
      if (in.space != SP_RGB || !wide_interleaved(in.layout) || in.bits == 8) break;
      if (matrix_without_ycbcr(target.p.matrix) || !target.p.full || target.layout != L_420) break;
      CState s; s.space = SP_YCBCR; s.layout = L_420; s.alpha = in.alpha; s.bits = in.bits; s.p = target.p;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_RGB24_32_TO_YCBCR444_GBR: { // rgb2yuv.cc:776-809
      if (in.space != SP_RGB || !(in.layout == L_RGB || in.layout == L_RGBA)) break;
      if (target.p.matrix != 0 || !target.p.full) break;
      CState s; s.space = SP_YCBCR; s.layout = L_444; s.alpha = target.alpha; s.bits = 8; s.p = target.p;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_DROP_ALPHA_PLANE: // alpha.cc:25-52
      if (planar_yuv_layout(in.layout) && in.alpha && !target.alpha) {
        CState s = in; s.alpha = false;
        out.push_back({s, COST_TRIVIAL});
      }
      break;
    case HM_OP_TO_HDR_PLANES: // hdr_sdr.cc:26-50
      if (planar_yuv_layout(in.layout) && in.bits == 8) {
        CState s = in; s.bits = target.bits;
        out.push_back({s, COST_PLAIN});
      }
      break;
    case HM_OP_TO_SDR_PLANES: // hdr_sdr.cc:108-136
      if (planar_yuv_layout(in.layout) && in.bits != 8 && target.bits == 8) {
        CState s = in; s.bits = 8;
        out.push_back({s, COST_PLAIN});
      }
      break;
    case HM_OP_BILINEAR_420_8: case HM_OP_BILINEAR_420_16: case HM_OP_BILINEAR_422_8: case HM_OP_BILINEAR_422_16: { // chroma_sampling.cc:443-486, 720-763
      const bool wide = op == HM_OP_BILINEAR_420_16 || op == HM_OP_BILINEAR_422_16;
      const int from = (op == HM_OP_BILINEAR_420_8 || op == HM_OP_BILINEAR_420_16) ? L_420 : L_422;
      if (in.space != SP_YCBCR || in.layout != from || o.up != 2) break;
      if ((in.bits != 8) != wide || in.p.matrix == 0) break;
      CState s = in; s.layout = L_444;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_AVERAGE_420_8: case HM_OP_AVERAGE_420_16: case HM_OP_AVERAGE_422_8: case HM_OP_AVERAGE_422_16: { // chroma_sampling.cc:27-74, 245-292
      const bool wide = op == HM_OP_AVERAGE_420_16 || op == HM_OP_AVERAGE_422_16;
      const int to = (op == HM_OP_AVERAGE_420_8 || op == HM_OP_AVERAGE_420_16) ? L_420 : L_422;
      if (in.space != SP_YCBCR || in.layout != L_444 || o.down != 2) break;
      if ((in.bits != 8) != wide || in.p.matrix == 0 || target.layout != to) break;
      CState s = in; s.layout = to;
      out.push_back({s, COST_PLAIN});
      break;
    }
    case HM_OP_SHARP_YUV: // rgb2yuv_sharp.cc:56-122: returns nothing without libsharpyuv
      break;
    case HM_OP_RGBA_TO_RGB_8: case HM_OP_RGBA_TO_RGB_16: // rgb2rgb.cc:733-758 (both instantiations test the same)
      if (in.space == SP_RGB && in.layout == L_RGBA && in.alpha) out.push_back({rgb_state(L_RGB, false, in.bits), COST_TRIVIAL});
      break;
    default: break;
  }
}

struct Node { int prev; int op; CState state; int cost; };

} // namespace

// The op sequence of the cheapest chain from the image's state to the target, found as the reference finds it
// (colorconversion.cc:266-420): take the FIRST cheapest frontier node, remove it by overwriting it with the last one,
// expand it with every operation in pool order; a state already expanded is ignored, a state already on the frontier is
// replaced only by a strictly cheaper way to reach it.  Returns the number of ops (0: nothing to do), -1: no chain.
int hm_colour_search(const hm_colour_request* rq, int ops_out[HM_COLOUR_MAX_OPS])
{
  // input state (colorconversion.cc:520-532): the image's nclx or a fresh profile, undefined values replaced (nclx.cc:346-359)
  CState in;
  in.space = rq->chroma == HM_CHROMA_MONO ? SP_MONO : SP_YCBCR;
  in.layout = rq->chroma;
  in.alpha = rq->has_alpha != 0;
  in.bits = rq->bit_depth;
  if (rq->has_nclx) { in.p.matrix = rq->matrix; in.p.primaries = rq->primaries; in.p.transfer = rq->transfer; in.p.full = rq->full_range != 0; }
  if (in.p.matrix == 2) in.p.matrix = 6;
  if (in.p.primaries == 2) in.p.primaries = 1;
  if (in.p.transfer == 2) in.p.transfer = 13;
  // target state (colorconversion.cc:534-585)
  CState tg = in;
  tg.space = SP_RGB;
  tg.layout = rq->out_format;
  const bool interleaved = rq->out_format >= L_RGB;
  tg.alpha = interleaved ? (rq->out_format == L_RGBA || rq->out_format == L_RRGGBBAA_BE || rq->out_format == L_RRGGBBAA_LE) : in.alpha;
  if (rq->output_bits) tg.bits = rq->output_bits;
  if (rq->out_format == L_RGB || rq->out_format == L_RGBA) tg.bits = 8;
  if (wide_interleaved(rq->out_format) && tg.bits <= 8) tg.bits = 10;
  Options o;
  if (rq->forced_bilinear) o.only_preferred = true; // heif_chroma_upsampling_bilinear is the default preference

  if (equal(in, tg)) return 0;
  std::vector<Node> done, frontier;
  frontier.push_back({-1, -1, in, 0});
  std::vector<Edge> edges;
  while (!frontier.empty()) {
    size_t best = 0;
    for (size_t i = 1; i < frontier.size(); i++)
      if (frontier[i].cost < frontier[best].cost) best = i;
    done.push_back(frontier[best]);
    frontier[best] = frontier.back();
    frontier.pop_back();
    const int cur = (int)done.size() - 1;
    if (equal(done[cur].state, tg)) {
      int n = 0;
      for (int i = cur; i > 0; i = done[i].prev) n++;
      if (n > HM_COLOUR_MAX_OPS) return -1;
      int k = n;
      for (int i = cur; i > 0; i = done[i].prev) ops_out[--k] = done[i].op;
      return n;
    }
    for (int op = 0; op < HM_OP_COUNT; op++) {
      edges.clear();
      edges_of(op, done[cur].state, tg, o, edges);
      for (const Edge& e : edges) {
        const int cost = e.cost + done[cur].cost;
        bool known = false;
        for (const Node& d : done)
          if (equal(d.state, e.to)) { known = true; break; }
        if (known) continue;
        for (Node& f : frontier)
          if (equal(f.state, e.to)) {
            known = true;
            if (f.cost > cost) f = {cur, op, e.to, cost};
            break;
          }
        if (!known) frontier.push_back({cur, op, e.to, cost});
      }
    }
  }
  return -1;
}

// The chain as work for the fused kernels: [depth change of the YCbCr planes] [bilinear chroma upsampling] core op
// [depth change of the RGB planes], interleave implied by the target.  Chains with other shapes are not offered.
int hm_colour_make_plan(const hm_colour_request* rq, hm_colour_plan* plan)
{
  std::memset(plan, 0, sizeof(*plan));
  int ops[HM_COLOUR_MAX_OPS];
  const int n = hm_colour_search(rq, ops);
  if (n < 0) return HM_PLAN_NO_CHAIN;
  plan->n_ops = n;
  int bits = rq->bit_depth;
  const int target_bits = (rq->out_format == L_RGB || rq->out_format == L_RGBA) ? 8 : (rq->bit_depth > 8 ? rq->bit_depth : 10);
  for (int i = 0; i < n; i++) {
    plan->ops[i] = ops[i];
    switch (ops[i]) {
      case HM_OP_DROP_ALPHA_PLANE: break; // (the planes converted here never include it)
      case HM_OP_MONO_TO_YCBCR420: // monochrome.cc:26-155: the chain continues as for a 4:2:0 image with neutral chroma
        if (plan->core || plan->pre || plan->bilinear || plan->mono_expand) return HM_PLAN_UNSUPPORTED;
        plan->mono_expand = 1;
        break;
      case HM_OP_TO_HDR_PLANES:
      case HM_OP_TO_SDR_PLANES: {
        const int kind = ops[i] == HM_OP_TO_HDR_PLANES ? HM_DEPTH_TO_HDR : HM_DEPTH_TO_SDR;
        const int nb = ops[i] == HM_OP_TO_HDR_PLANES ? target_bits : 8;
        if (!plan->core) { if (plan->pre || plan->bilinear) return HM_PLAN_UNSUPPORTED; plan->pre = kind; plan->pre_bits = nb; }
        else { if (plan->post) return HM_PLAN_UNSUPPORTED; plan->post = kind; plan->post_bits = nb; }
        bits = nb;
        break;
      }
      case HM_OP_BILINEAR_420_8: case HM_OP_BILINEAR_420_16: case HM_OP_BILINEAR_422_8: case HM_OP_BILINEAR_422_16:
        if (plan->core || plan->bilinear) return HM_PLAN_UNSUPPORTED;
        plan->bilinear = 1;
        break;
      case HM_OP_YCBCR_TO_RGB_8: case HM_OP_YCBCR_TO_RGB_16: case HM_OP_YCBCR420_TO_RRGGBBAA:
        if (plan->core) return HM_PLAN_UNSUPPORTED;
        plan->core = HM_CORE_FLOAT; plan->core_bits = bits; plan->core_step = i;
        break;
      case HM_OP_YCBCR420_TO_RGB24: case HM_OP_YCBCR420_TO_RGB32:
        if (plan->core) return HM_PLAN_UNSUPPORTED;
        plan->core = HM_CORE_INT420; plan->core_bits = bits; plan->core_step = i;
        break;
      case HM_OP_MONO_TO_RGB24_32:
        if (plan->core || plan->pre == HM_DEPTH_TO_HDR) return HM_PLAN_UNSUPPORTED;
        plan->core = HM_CORE_MONO; plan->core_bits = bits; plan->core_step = i;
        break;
      case HM_OP_RGB_TO_RGB24_32: case HM_OP_RGB_HDR_TO_RRGGBBAA_BE: case HM_OP_RGB_TO_RRGGBBAA_BE: case HM_OP_SWAP_ENDIANNESS:
        if (!plan->core) return HM_PLAN_UNSUPPORTED; // the interleave that ends the chain
        break;
      default: return HM_PLAN_UNSUPPORTED;
    }
  }
  if (!plan->core) return HM_PLAN_UNSUPPORTED;
  if (plan->core == HM_CORE_MONO && (plan->post || plan->bilinear)) return HM_PLAN_UNSUPPORTED;
  if (plan->core == HM_CORE_INT420 && (plan->post || plan->bilinear)) return HM_PLAN_UNSUPPORTED;
  return HM_PLAN_OK;
}
