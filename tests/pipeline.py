"""CPU restatement of the image-level flow (decode_image_user / decode_full_grid_image /
decode_and_paste_tile_image / convert_colorspace of the reference) out of oracle pieces:
product host parsing (C ABI) + oracle executors + oracle paste + oracle colour ops.
Test infrastructure only."""
import ctypes as C

import numpy as np

import hevcutil
import orc

SURVEY_FNV_INIT = 1469598103934665603  # the survey's harness used this (truncated) FNV-1a basis for BASELINE.md §2


class ImageInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "width height bit_depth chroma is_grid grid_rows grid_cols tile_width tile_height has_transforms has_alpha coded_width coded_height has_nclx".split()]


class DecodeParams(C.Structure):
    _fields_ = [("out_format", C.c_int32), ("host_threads", C.c_int32), ("ignore_transformations", C.c_int32),
                ("chroma_upsampling", C.c_int32), ("stream", C.c_void_p), ("ext_dst", C.c_void_p),
                ("ext_dst_len", C.c_uint32), ("ext_dst_stride", C.c_uint32), ("strict_decoding", C.c_int32),
                ("convert_hdr_to_8bit", C.c_int32)]


class Decoded(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "width height bit_depth chroma out_format has_nclx primaries transfer matrix full_range used_ext_dst".split()] + \
               [("plane", C.POINTER(C.c_uint8) * 3), ("stride", C.c_int32 * 3), ("plane_width", C.c_int32 * 3), ("plane_height", C.c_int32 * 3),
                ("has_alpha", C.c_int32), ("alpha", C.POINTER(C.c_uint8)), ("alpha_stride", C.c_int32), ("warnings", C.c_int32)]


def bind(hm):
    hm.hm_last_error.restype = C.c_char_p
    hm.hm_file_open.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p)]
    hm.hm_file_close.argtypes = [C.c_void_p]
    hm.hm_file_primary_item.argtypes = [C.c_void_p]
    hm.hm_file_primary_item.restype = C.c_uint32
    hm.hm_file_image_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(ImageInfo)]
    hm.hm_file_item_hevc_data.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t)]
    hm.hm_decode_item.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(DecodeParams), C.POINTER(Decoded)]
    hm.hm_decoded_free.argtypes = [C.POINTER(Decoded)]
    hm.hm_free.argtypes = [C.c_void_p]
    return hm


class HeifFile:
    def __init__(self, hm, data):
        self.hm = bind(hm)
        self.h = C.c_void_p()
        rc = hm.hm_file_open(data, len(data), C.byref(self.h))
        if rc:
            raise RuntimeError(f"hm_file_open: {rc}: {hm.hm_last_error().decode()}")

    def alpha_item(self, iid):
        self.hm.hm_file_alpha_item.restype = C.c_uint32
        self.hm.hm_file_alpha_item.argtypes = [C.c_void_p, C.c_uint32]
        return self.hm.hm_file_alpha_item(self.h, iid)

    def primary(self):
        return self.hm.hm_file_primary_item(self.h)

    def info(self, iid):
        i = ImageInfo()
        rc = self.hm.hm_file_image_info(self.h, iid, C.byref(i))
        if rc:
            raise RuntimeError(f"hm_file_image_info: {rc}: {self.hm.hm_last_error().decode()}")
        return i

    def hevc_data(self, iid):
        p = C.POINTER(C.c_uint8)()
        n = C.c_size_t()
        rc = self.hm.hm_file_item_hevc_data(self.h, iid, C.byref(p), C.byref(n))
        if rc:
            raise RuntimeError(f"hm_file_item_hevc_data: {rc}: {self.hm.hm_last_error().decode()}")
        out = C.string_at(p, n.value)
        self.hm.hm_free(p)
        return out

    def decode(self, iid, out_format, threads=1, upsampling=0, copy=True, ignore_transformations=0, strict=0):
        """GPU path through the C ABI; returns (array rows x stride, Decoded meta); copy=False only times the call
        (the pinned result is released without being copied into numpy arrays)."""
        prm = DecodeParams(out_format, threads, ignore_transformations, upsampling, None, None, 0, 0, strict, 0)
        d = Decoded()
        rc = self.hm.hm_decode_item(self.h, iid, C.byref(prm), C.byref(d))
        if rc:
            raise RuntimeError(f"hm_decode_item: {rc}: {self.hm.hm_last_error().decode()}")
        planes = []
        n = (1 if out_format else 3) if copy else 0
        for c in range(n):
            if not d.plane[c]:  # monochrome image: Y only
                continue
            rows = d.plane_height[c]
            a = np.ctypeslib.as_array(d.plane[c], shape=(rows, d.stride[c])).copy()
            planes.append(a)
        meta = {k: getattr(d, k) for k in "width height bit_depth chroma out_format has_nclx primaries transfer matrix full_range".split()}
        meta["stride"] = [d.stride[c] for c in range(3)]
        meta["plane_size"] = [(d.plane_width[c], d.plane_height[c]) for c in range(3)]
        meta["has_alpha"] = d.has_alpha
        meta["warnings"] = d.warnings
        if copy and d.alpha:
            rows = max(64, (d.height + 1) & ~1)
            meta["alpha"] = np.ctypeslib.as_array(d.alpha, shape=(rows, d.alpha_stride)).copy()
        self.hm.hm_decoded_free(C.byref(d))
        return planes, meta

    def close(self):
        if self.h:
            self.hm.hm_file_close(self.h)
            self.h = C.c_void_p()


class PipelineConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in "host_threads max_in_flight out_format chroma_upsampling ignore_transformations strict_decoding device cpu_first cpu_count".split()]


class PipelineResult(C.Structure):
    _fields_ = [("tag", C.c_uint64), ("status", C.c_int32), ("image", Decoded), ("handle", C.c_void_p)]


class Pipeline:
    """hm_pipeline_*: many HEIF files in flight (host entropy decode || H2D || kernels || D2H)."""

    def __init__(self, hm, out_format, host_threads=4, max_in_flight=8, device=-1, cpus=None):
        self.hm = bind(hm)
        hm.hm_pipeline_create.argtypes = [C.POINTER(PipelineConfig), C.POINTER(C.c_void_p)]
        hm.hm_pipeline_destroy.argtypes = [C.c_void_p]
        hm.hm_pipeline_submit.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_uint64]
        hm.hm_pipeline_pending.argtypes = [C.c_void_p]
        hm.hm_pipeline_next.argtypes = [C.c_void_p, C.POINTER(PipelineResult)]
        hm.hm_pipeline_release.argtypes = [C.c_void_p, C.POINTER(PipelineResult)]
        cpu_first, cpu_count = cpus if cpus else (0, 0)  # the crew's CPUs [first, first + count)
        cfg = PipelineConfig(host_threads, max_in_flight, out_format, 0, 0, 0, device, cpu_first, cpu_count)
        self.h = C.c_void_p()
        rc = hm.hm_pipeline_create(C.byref(cfg), C.byref(self.h))
        if rc:
            raise RuntimeError(f"hm_pipeline_create: {rc}: {hm.hm_last_error().decode()}")

    def submit(self, data, tag, item=0):
        """True if queued, False if the pipeline is full; raises on a malformed file"""
        rc = self.hm.hm_pipeline_submit(self.h, data, len(data), item, tag)
        if rc < 0:
            raise RuntimeError(f"hm_pipeline_submit: {rc}: {self.hm.hm_last_error().decode()}")
        return rc == 0

    def pending(self):
        return self.hm.hm_pipeline_pending(self.h)

    def next(self, copy=True):
        """(tag, status, array or None, meta) of the oldest pending image"""
        r = PipelineResult()
        rc = self.hm.hm_pipeline_next(self.h, C.byref(r))
        if rc:
            raise RuntimeError(f"hm_pipeline_next: {rc}: {self.hm.hm_last_error().decode()}")
        arr, meta = None, None
        if r.status == 0:
            d = r.image
            meta = dict(width=d.width, height=d.height, stride=d.stride[0], bit_depth=d.bit_depth)
            if copy:
                arr = np.ctypeslib.as_array(d.plane[0], shape=(d.plane_height[0], d.stride[0])).copy()
        else:
            meta = dict(error=self.hm.hm_last_error().decode())
        tag, status = r.tag, r.status
        self.hm.hm_pipeline_release(self.h, C.byref(r))
        return tag, status, arr, meta

    def close(self):
        if self.h:
            self.hm.hm_pipeline_destroy(self.h)
            self.h = C.c_void_p()


def cpu_decode(hm, tiles, tile_w, tile_h, canvas_w, canvas_h, cols, is_grid, out_fmt, tile_colr=None, decoder="oracle", bilinear=False, transforms=None,
               has_alpha=False, tile_transforms=None):
    """tiles: list of [len][NAL] strings.  Returns (rgb array, stride) following the reference flow."""
    o = orc.load()
    first = None
    canv = None
    for i, data in enumerate(tiles):
        if decoder == "ref":
            planes, info = orc.ref_decode(data, 0)
        else:
            blob = hevcutil.parse_concealing(hm, data)[0] if decoder == "oracle_concealing" else hevcutil.parse(hm, data)
            planes, info = orc.oracle_decode(blob, 3)
            cf = info["chroma"]
            tcw, tch = (tile_w if cf == 3 else (tile_w + 1) // 2), ((tile_h + 1) // 2 if cf == 1 else tile_h)
            planes = [planes[0][:tile_h, :tile_w], planes[1][:tch, :tcw], planes[2][:tch, :tcw]]
        bd, cf = info["bit_depth"], info["chroma"]
        bps = 2 if bd > 8 else 1
        nclx = (1, info["full_range"], info["matrix"], info["primaries"])
        if tile_colr is not None:
            nclx = (1, tile_colr[3], tile_colr[2], tile_colr[0])
        if first is None:
            first = dict(bd=bd, cf=cf, nclx=nclx)
            cw = canvas_w if cf == 3 else (canvas_w + 1) // 2
            ch = (canvas_h + 1) // 2 if cf == 1 else canvas_h
            canv = [orc.alloc_plane(canvas_w, canvas_h, bps), orc.alloc_plane(cw, ch, bps), orc.alloc_plane(cw, ch, bps)]
        x0, y0 = (i % cols) * tile_w, (i // cols) * tile_h
        if tile_transforms and i in tile_transforms:
            # the tile item's own irot / imir / clap: applied to the tile image before the paste (context.cc:1957-2020, 2407-2415)
            tcw0 = tile_w if cf == 3 else (tile_w + 1) // 2
            tch0 = (tile_h + 1) // 2 if cf == 1 else tile_h
            bufs = []
            for c, (w_, h_) in enumerate(((tile_w, tile_h), (tcw0, tch0), (tcw0, tch0))):
                buf, st = orc.alloc_plane(w_, h_, bps)
                src = np.ascontiguousarray(planes[c][:h_, :w_].astype(np.uint8 if bps == 1 else np.uint16))
                buf[:h_, :w_ * bps] = src.view(np.uint8).reshape(h_, w_ * bps)
                bufs.append((buf, st))
            tp, tdims, _, _ = orc.transform_planes(bufs, [(tile_w, tile_h), (tcw0, tch0), (tcw0, tch0)], tile_w, tile_h, bd, tile_transforms[i])
            planes = []
            for (buf, st), (w_, h_) in zip(tp, tdims):
                rows = np.ascontiguousarray(buf[:h_, :w_ * bps])
                planes.append(rows.view(np.uint8 if bps == 1 else np.uint16).reshape(h_, w_))
        for c in range(3):
            p = planes[c]
            raw = np.ascontiguousarray(p.astype(np.uint8)) if bps == 1 else np.ascontiguousarray(p.astype(np.uint16))
            has, full, mat = (nclx[0], nclx[1], nclx[2]) if is_grid else (0, 1, 1)
            rc = o.orc_paste_tile_plane(orc.ptr(raw), raw.shape[1] * bps, raw.shape[1], raw.shape[0], orc.ptr(canv[c][0]), canv[c][1],
                                        canvas_w, canvas_h, x0, y0, c, cf, bd, has, full, mat)
            assert rc == 0
    bd, cf = first["bd"], first["cf"]
    if transforms:  # irot / imir / clap on the decoded planes, before the colour conversion (context.cc:1957-2020)
        cw0 = canvas_w if cf == 3 else (canvas_w + 1) // 2
        ch0 = (canvas_h + 1) // 2 if cf == 1 else canvas_h
        canv, _, canvas_w, canvas_h = orc.transform_planes(canv, [(canvas_w, canvas_h), (cw0, ch0), (cw0, ch0)], canvas_w, canvas_h, bd, transforms)
    has_nclx = 0 if is_grid else 1
    _, full, mat, prim = first["nclx"]
    # the colour conversion: op by op along the chain the reference's pipeline search picks for this image and target
    # (oracle/pipeline_search.py; incl. the rule that every op after the first sees the intermediate state's profile)
    out, os_, _ = orc.convert_by_search(canv, canvas_w, canvas_h, bd, cf, (has_nclx, mat, prim, full), out_fmt, has_alpha=has_alpha,
                                        forced_bilinear=bool(bilinear))
    return out, os_, canv


def attach_alpha(hm, rgba, stride, w, h, alpha_hevc, aw, ah):
    """CPU flow of context.cc:2029-2078 for an RGBA result: decode the alpha auxiliary image, take its Y plane, scale it
    nearest-neighbour to the image size if needed, store it in byte 3 of every pixel."""
    o = orc.load()
    _, _, canv = cpu_decode(hm, [alpha_hevc], aw, ah, aw, ah, 1, False, 10)
    a, a_stride = canv[0]
    if (aw, ah) != (w, h):
        scaled, s_stride = orc.alloc_plane(w, h, 1)
        o.orc_scale_nn_plane(orc.ptr(a), a_stride, aw, ah, 1, orc.ptr(scaled), s_stride, w, h)
        a, a_stride = scaled, s_stride
    o.orc_set_alpha_rgba(orc.ptr(rgba), stride, w, h, orc.ptr(a), a_stride)
    return a, a_stride


def survey_fnv(buf, stride, row_bytes, rows):
    return f"{orc.load().orc_fnv1a64_rows(orc.ptr(buf), stride, row_bytes, rows, SURVEY_FNV_INIT):016x}"
