"""pipeline_search.py - TEST INFRASTRUCTURE (oracle): restatement of the reference's colour-conversion pipeline search.

Which chain of operations `convert_colorspace()` runs for an (input state, target) pair decides the arithmetic of the
result, and the reference finds it with a Dijkstra search over ColorStates whose tie-breaking depends on the order of
the operation pool and on the swap-with-last removal from its border list (libheif/color-conversion/
colorconversion.cc:266-420).  The product hard-codes the chains it offers (colour_host.cpp: hm_colour_pipeline); this
module restates the search itself - every operation's `state_after_conversion` (file:line cited per op) in pool order
(colorconversion.cc:218-255, libyuv and libsharpyuv absent as in the oracle build) - so that tests can pin
the product's choice for every input / target combination, including the equal-cost ties.

Parity status: unpinned against a reference run (libheif cannot be built here: generated heif_version.h, cmake); pinned
indirectly through the reference fingerprints of BASELINE.md that depend on the chain (tests/test_golden_heic.py), and
by tests/test_pipeline_search.py's known chains.  Only tests/ may import this file.
"""
from collections import namedtuple

# heif_colorspace / heif_chroma (libheif/api/libheif/heif.h)
CS_YCBCR, CS_RGB, CS_MONO = 0, 1, 2
C_MONO, C_420, C_422, C_444 = 0, 1, 2, 3
C_RGB, C_RGBA, C_RRGGBB_BE, C_RRGGBBAA_BE, C_RRGGBB_LE, C_RRGGBBAA_LE = 10, 11, 12, 13, 14, 15

COST_TRIVIAL, COST_HARDWARE, COST_OPT, COST_UNOPT, COST_SLOW = 1, 2, 6, 11, 16  # colorconversion.h:95-101

UP_NEAREST, UP_BILINEAR = 1, 2        # heif_chroma_upsampling_algorithm
DOWN_NEAREST, DOWN_AVERAGE, DOWN_SHARP = 1, 2, 3

Nclx = namedtuple("Nclx", "matrix primaries transfer full_range")
SRGB = Nclx(6, 1, 13, True)           # color_profile_nclx() default (nclx.cc:316-323)
State = namedtuple("State", "colorspace chroma has_alpha bpp nclx")
Options = namedtuple("Options", "down up only_preferred")
DEFAULT_OPTIONS = Options(DOWN_AVERAGE, UP_BILINEAR, False)  # heif.cc:1080-1082


def same(a, b):
    """ColorState::operator== (colorconversion.cc:146-166)"""
    if (a.colorspace, a.chroma, a.has_alpha, a.bpp) != (b.colorspace, b.chroma, b.has_alpha, b.bpp):
        return False
    if a.colorspace == CS_YCBCR:
        return (a.nclx.full_range, a.nclx.matrix, a.nclx.primaries) == (b.nclx.full_range, b.nclx.matrix, b.nclx.primaries)
    return True


def _nn_only_refuses(inp, opt):  # "this Op only implements the nearest-neighbor algorithm"
    return inp.chroma != C_444 and opt.up != UP_NEAREST and opt.only_preferred


def op_rgb_to_rgb24_32(i, t, o):  # rgb2rgb.cc:29-63
    if i.colorspace != CS_RGB or i.chroma != C_444 or i.bpp != 8:
        return []
    return [(State(CS_RGB, C_RGBA, True, 8, SRGB), COST_UNOPT), (State(CS_RGB, C_RGB, False, 8, SRGB), COST_UNOPT)]


def op_rgb24_32_to_rgb(i, t, o):  # rgb2rgb.cc:519-546
    if i.colorspace != CS_RGB or i.chroma not in (C_RGB, C_RGBA) or i.bpp != 8:
        return []
    return [(State(CS_RGB, C_444, t.has_alpha, i.bpp, SRGB), COST_UNOPT)]


def _ycbcr_to_rgb(hdr):  # yuv2rgb.cc:30-76
    def f(i, t, o):
        if _nn_only_refuses(i, o):
            return []
        if i.colorspace != CS_YCBCR or i.chroma not in (C_444, C_422, C_420):
            return []
        if i.nclx.matrix in (11, 14):
            return []
        if (i.bpp != 8) != hdr:
            return []
        return [(State(CS_RGB, C_444, i.has_alpha, i.bpp, SRGB), COST_UNOPT)]
    return f


def op_ycbcr420_to_rgb24(i, t, o):  # yuv2rgb.cc:261-303
    if _nn_only_refuses(i, o):
        return []
    if i.colorspace != CS_YCBCR or i.chroma != C_420 or i.bpp != 8 or i.has_alpha:
        return []
    if i.nclx.matrix in (0, 8, 11, 14) or not i.nclx.full_range:
        return []
    return [(State(CS_RGB, C_RGB, False, 8, SRGB), COST_UNOPT)]


def op_ycbcr420_to_rgb32(i, t, o):  # yuv2rgb.cc:370-413
    if _nn_only_refuses(i, o):
        return []
    if i.colorspace != CS_YCBCR or i.chroma != C_420 or i.bpp != 8:
        return []
    if i.nclx.matrix in (0, 8, 11, 14) or not i.nclx.full_range:
        return []
    return [(State(CS_RGB, C_RGBA, True, 8, SRGB), COST_UNOPT)]


def op_ycbcr420_to_rrggbbaa(i, t, o):  # yuv2rgb.cc:499-547
    if _nn_only_refuses(i, o):
        return []
    if i.colorspace != CS_YCBCR or i.chroma != C_420 or i.bpp == 8:
        return []
    if i.nclx.matrix in (0, 8, 11, 14):
        return []
    return [(State(CS_RGB, C_RRGGBBAA_LE if i.has_alpha else C_RRGGBB_LE, i.has_alpha, i.bpp, SRGB), COST_UNOPT),
            (State(CS_RGB, C_RRGGBBAA_BE if i.has_alpha else C_RRGGBB_BE, i.has_alpha, i.bpp, SRGB), COST_UNOPT)]


def op_rgb_hdr_to_rrggbbaa_be(i, t, o):  # rgb2rgb.cc:147-186
    if i.colorspace != CS_RGB or i.chroma != C_444 or i.bpp == 8:
        return []
    out = []
    if not i.has_alpha:
        out.append((State(CS_RGB, C_RRGGBB_BE, False, i.bpp, SRGB), COST_UNOPT))
    out.append((State(CS_RGB, C_RRGGBBAA_BE, True, i.bpp, SRGB), COST_UNOPT))
    return out


def op_rgb_to_rrggbbaa_be(i, t, o):  # rgb2rgb.cc:276-315
    if i.colorspace != CS_RGB or i.chroma != C_444 or i.bpp != 8:
        return []
    out = []
    if not i.has_alpha:
        out.append((State(CS_RGB, C_RRGGBB_BE, False, i.bpp, SRGB), COST_UNOPT))
    out.append((State(CS_RGB, C_RRGGBBAA_BE, True, i.bpp, SRGB), COST_UNOPT))
    return out


def op_mono_to_ycbcr420(i, t, o):  # monochrome.cc:26-49
    if i.colorspace != CS_MONO or i.chroma != C_MONO:
        return []
    return [(State(CS_YCBCR, C_420, i.has_alpha, i.bpp, SRGB), COST_OPT)]


def op_mono_to_rgb24_32(i, t, o):  # monochrome.cc:160-198
    if i.colorspace != CS_MONO or i.chroma != C_MONO or i.bpp != 8:
        return []
    out = []
    if not i.has_alpha:
        out.append((State(CS_RGB, C_RGB, False, 8, SRGB), COST_UNOPT))
    out.append((State(CS_RGB, C_RGBA, True, 8, SRGB), COST_UNOPT))
    return out


def op_rrggbbaa_swap_endianness(i, t, o):  # rgb2rgb.cc:614-673
    if i.colorspace != CS_RGB or i.chroma not in (C_RRGGBB_LE, C_RRGGBB_BE, C_RRGGBBAA_LE, C_RRGGBBAA_BE):
        return []
    swap = {C_RRGGBB_LE: C_RRGGBB_BE, C_RRGGBB_BE: C_RRGGBB_LE, C_RRGGBBAA_LE: C_RRGGBBAA_BE, C_RRGGBBAA_BE: C_RRGGBBAA_LE}
    return [(State(CS_RGB, swap[i.chroma], i.chroma in (C_RRGGBBAA_LE, C_RRGGBBAA_BE), i.bpp, SRGB), COST_UNOPT)]


def op_rrggbbaa_be_to_rgb_hdr(i, t, o):  # rgb2rgb.cc:405-433
    if i.colorspace != CS_RGB or i.chroma not in (C_RRGGBB_BE, C_RRGGBBAA_BE) or i.bpp == 8:
        return []
    return [(State(CS_RGB, C_444, t.has_alpha, i.bpp, SRGB), COST_UNOPT)]


def op_rgb24_32_to_ycbcr(i, t, o):  # rgb2yuv.cc:473-518
    if t.chroma != C_444 and o.down != DOWN_NEAREST and o.only_preferred:
        return []
    if i.colorspace != CS_RGB or i.chroma not in (C_RGB, C_RGBA):
        return []
    if t.chroma not in (C_420, C_422, C_444):
        return []
    if t.nclx.matrix in (0, 8, 11, 14):
        return []
    return [(State(CS_YCBCR, t.chroma, t.has_alpha, 8, t.nclx), COST_UNOPT)]


def _rgb_to_ycbcr(hdr):  # rgb2yuv.cc:32-85
    def f(i, t, o):
        if (i.bpp != 8) != hdr:
            return []
        if i.colorspace != CS_RGB or i.chroma != C_444:
            return []
        if t.nclx.matrix in (8, 11, 14):
            return []
        if t.chroma != C_444 and (o.down == DOWN_NEAREST or not o.only_preferred):
            return [(State(CS_YCBCR, t.chroma, i.has_alpha, i.bpp, t.nclx), COST_UNOPT)]
        return [(State(CS_YCBCR, C_444, i.has_alpha, i.bpp, t.nclx), COST_UNOPT)]
    return f


def op_rrggbbxx_hdr_to_ycbcr420(i, t, o):  # rgb2yuv.cc:281-331
    if t.chroma != C_444 and o.down != DOWN_NEAREST and o.only_preferred:
        return []
    if i.colorspace != CS_RGB or i.chroma not in (C_RRGGBB_BE, C_RRGGBB_LE, C_RRGGBBAA_BE, C_RRGGBBAA_LE) or i.bpp == 8:
        return []
    if t.nclx.matrix in (0, 8, 11, 14) or not t.nclx.full_range:
        return []
    if t.chroma != C_420:
        return []
    return [(State(CS_YCBCR, C_420, i.has_alpha, i.bpp, t.nclx), COST_UNOPT)]


def op_rgb24_32_to_ycbcr444_gbr(i, t, o):  # rgb2yuv.cc:776-809
    if i.colorspace != CS_RGB or i.chroma not in (C_RGB, C_RGBA):
        return []
    if t.nclx.matrix != 0 or not t.nclx.full_range:
        return []
    return [(State(CS_YCBCR, C_444, t.has_alpha, 8, t.nclx), COST_UNOPT)]


def op_drop_alpha_plane(i, t, o):  # alpha.cc:25-52
    if i.chroma not in (C_MONO, C_420, C_422, C_444) or not i.has_alpha or t.has_alpha:
        return []
    return [(i._replace(has_alpha=False), COST_TRIVIAL)]


def op_to_hdr_planes(i, t, o):  # hdr_sdr.cc:26-50
    if i.chroma not in (C_MONO, C_420, C_422, C_444) or i.bpp != 8:
        return []
    return [(i._replace(bpp=t.bpp), COST_UNOPT)]


def op_to_sdr_planes(i, t, o):  # hdr_sdr.cc:108-136
    if i.chroma not in (C_MONO, C_420, C_422, C_444) or i.bpp == 8:
        return []
    if t.bpp != 8:
        return []
    return [(i._replace(bpp=8), COST_UNOPT)]


def _bilinear_up(src_chroma, hdr):  # chroma_sampling.cc:443-486, 720-763
    def f(i, t, o):
        if i.colorspace != CS_YCBCR or i.chroma != src_chroma:
            return []
        if o.up != UP_BILINEAR:
            return []
        if (i.bpp != 8) != hdr:
            return []
        if i.nclx.matrix == 0:
            return []
        return [(State(CS_YCBCR, C_444, i.has_alpha, i.bpp, i.nclx), COST_UNOPT)]
    return f


def _average_down(dst_chroma, hdr):  # chroma_sampling.cc:27-74, 245-292
    def f(i, t, o):
        if i.colorspace != CS_YCBCR or i.chroma != C_444:
            return []
        if o.down != DOWN_AVERAGE:
            return []
        if (i.bpp != 8) != hdr:
            return []
        if i.nclx.matrix == 0:
            return []
        if t.chroma != dst_chroma:
            return []
        return [(State(CS_YCBCR, dst_chroma, i.has_alpha, i.bpp, i.nclx), COST_UNOPT)]
    return f


def op_sharp(i, t, o):  # rgb2yuv_sharp.cc:56-122 without HAVE_LIBSHARPYUV
    return []


def _rgba_general_to_rgb(i, t, o):  # rgb2rgb.cc:733-758 (both Pixel instantiations test the same thing)
    if i.colorspace != CS_RGB or i.chroma != C_RGBA or not i.has_alpha:
        return []
    return [(State(CS_RGB, C_RGB, False, i.bpp, SRGB), COST_TRIVIAL)]


# the operation pool in the order of ColorConversionPipeline::init_ops (colorconversion.cc:218-255; HAVE_YUV unset)
OPS = [
    ("Op_RGB_to_RGB24_32", op_rgb_to_rgb24_32),
    ("Op_RGB24_32_to_RGB", op_rgb24_32_to_rgb),
    ("Op_YCbCr_to_RGB<uint16_t>", _ycbcr_to_rgb(True)),
    ("Op_YCbCr_to_RGB<uint8_t>", _ycbcr_to_rgb(False)),
    ("Op_YCbCr420_to_RGB24", op_ycbcr420_to_rgb24),
    ("Op_YCbCr420_to_RGB32", op_ycbcr420_to_rgb32),
    ("Op_YCbCr420_to_RRGGBBaa", op_ycbcr420_to_rrggbbaa),
    ("Op_RGB_HDR_to_RRGGBBaa_BE", op_rgb_hdr_to_rrggbbaa_be),
    ("Op_RGB_to_RRGGBBaa_BE", op_rgb_to_rrggbbaa_be),
    ("Op_mono_to_YCbCr420", op_mono_to_ycbcr420),
    ("Op_mono_to_RGB24_32", op_mono_to_rgb24_32),
    ("Op_RRGGBBaa_swap_endianness", op_rrggbbaa_swap_endianness),
    ("Op_RRGGBBaa_BE_to_RGB_HDR", op_rrggbbaa_be_to_rgb_hdr),
    ("Op_RGB24_32_to_YCbCr", op_rgb24_32_to_ycbcr),
    ("Op_RGB_to_YCbCr<uint8_t>", _rgb_to_ycbcr(False)),
    ("Op_RGB_to_YCbCr<uint16_t>", _rgb_to_ycbcr(True)),
    ("Op_RRGGBBxx_HDR_to_YCbCr420", op_rrggbbxx_hdr_to_ycbcr420),
    ("Op_RGB24_32_to_YCbCr444_GBR", op_rgb24_32_to_ycbcr444_gbr),
    ("Op_drop_alpha_plane", op_drop_alpha_plane),
    ("Op_to_hdr_planes", op_to_hdr_planes),
    ("Op_to_sdr_planes", op_to_sdr_planes),
    ("Op_YCbCr420_bilinear_to_YCbCr444<uint8_t>", _bilinear_up(C_420, False)),
    ("Op_YCbCr420_bilinear_to_YCbCr444<uint16_t>", _bilinear_up(C_420, True)),
    ("Op_YCbCr422_bilinear_to_YCbCr444<uint8_t>", _bilinear_up(C_422, False)),
    ("Op_YCbCr422_bilinear_to_YCbCr444<uint16_t>", _bilinear_up(C_422, True)),
    ("Op_YCbCr444_to_YCbCr420_average<uint8_t>", _average_down(C_420, False)),
    ("Op_YCbCr444_to_YCbCr420_average<uint16_t>", _average_down(C_420, True)),
    ("Op_YCbCr444_to_YCbCr422_average<uint8_t>", _average_down(C_422, False)),
    ("Op_YCbCr444_to_YCbCr422_average<uint16_t>", _average_down(C_422, True)),
    ("Op_Any_RGB_to_YCbCr_420_Sharp", op_sharp),
    ("Op_RGBA_GENERAL_to_RGB_GENTRAL<uint8_t>", _rgba_general_to_rgb),
    ("Op_RGBA_GENERAL_to_RGB_GENTRAL<uint16_t>", _rgba_general_to_rgb),
]


def construct_pipeline(input_state, target_state, options=DEFAULT_OPTIONS):
    """ColorConversionPipeline::construct_pipeline (colorconversion.cc:266-420): the list of (operation name, state after
    it), [] when input == target, None when there is no chain.  Literal restatement of the search incl. its tie-breaks:
    first minimum of the border list, removal by overwriting with the last entry, a border node is replaced only by a
    strictly cheaper path."""
    if same(input_state, target_state):
        return []
    processed = []  # (prev index, op name, state, cost)
    border = [(-1, None, input_state, 0)]
    while border:
        min_idx = min(range(len(border)), key=lambda k: (border[k][3], k))
        processed.append(border[min_idx])
        border[min_idx] = border[-1]
        border.pop()
        cur = processed[-1]
        if same(cur[2], target_state):
            steps = []
            idx = len(processed) - 1
            while idx > 0:
                steps.append((processed[idx][1], processed[idx][2]))
                idx = processed[idx][0]
            return steps[::-1]
        for name, fn in OPS:
            for out_state, cost in fn(cur[2], target_state, options):
                new_cost = cost + cur[3]
                if any(same(p[2], out_state) for p in processed):
                    continue
                for k, bnode in enumerate(border):
                    if same(bnode[2], out_state):
                        if bnode[3] > new_cost:
                            border[k] = (len(processed) - 1, name, out_state, new_cost)
                        break
                else:
                    border.append((len(processed) - 1, name, out_state, new_cost))
    return None


def conversion_states(colorspace, chroma, has_alpha, bpp, nclx, target_colorspace, target_chroma, output_bpp=0):
    """input and target ColorState as convert_colorspace() builds them (colorconversion.cc:520-590); nclx None = the image
    carries no profile (a fresh ColorState: sRGB defaults)"""
    n = nclx if nclx is not None else SRGB
    # replace_undefined_values_with_sRGB_defaults (nclx.cc:346-359)
    n = Nclx(6 if n.matrix == 2 else n.matrix, 1 if n.primaries == 2 else n.primaries, 13 if n.transfer == 2 else n.transfer, n.full_range)
    inp = State(colorspace, chroma, has_alpha, bpp, n)
    interleaved = target_chroma >= C_RGB
    with_alpha = target_chroma in (C_RGBA, C_RRGGBBAA_BE, C_RRGGBBAA_LE)
    out_alpha = with_alpha if interleaved else has_alpha
    out_bpp = output_bpp if output_bpp else bpp
    if target_chroma in (C_RGB, C_RGBA):
        out_bpp = 8
    if target_chroma in (C_RRGGBB_LE, C_RRGGBB_BE, C_RRGGBBAA_LE, C_RRGGBBAA_BE) and out_bpp <= 8:
        out_bpp = 10
    return inp, State(target_colorspace, target_chroma, out_alpha, out_bpp, n)


def chain(colorspace, chroma, has_alpha, bpp, nclx, target_colorspace, target_chroma, output_bpp=0, options=DEFAULT_OPTIONS):
    inp, tgt = conversion_states(colorspace, chroma, has_alpha, bpp, nclx, target_colorspace, target_chroma, output_bpp)
    steps = construct_pipeline(inp, tgt, options)
    return None if steps is None else [name for name, _ in steps]
