"""Minimal HEIF writer (test tool): wraps coded HEVC pictures ([u32 BE len][NAL] strings) into a
single-image or 'grid' .heic (ftyp / meta{hdlr,pitm,iloc,iinf,iref,iprp{ipco,ipma}} / mdat)."""
import struct


def _box(t, payload):
    return struct.pack(">I4s", 8 + len(payload), t) + payload


def _full(t, version, flags, payload):
    return _box(t, bytes([version]) + flags.to_bytes(3, "big") + payload)


def split_nals(lp):
    out, p = [], 0
    while p + 4 <= len(lp):
        n = int.from_bytes(lp[p:p + 4], "big")
        out.append(lp[p + 4:p + 4 + n])
        p += 4 + n
    return out


def _hvcc(param_nals, chroma_format, bit_depth):
    body = bytes([1, 1]) + b"\x60\x00\x00\x00" + b"\x90\x00\x00\x00\x00\x00" + bytes([183])
    body += b"\xf0\x00" + b"\xfc" + bytes([0xFC | chroma_format, 0xF8 | (bit_depth - 8), 0xF8 | (bit_depth - 8)])
    body += b"\x00\x00" + bytes([0x0F])  # avgFrameRate, 1 temporal layer, lengthSizeMinusOne = 3
    groups = {}
    for n in param_nals:
        groups.setdefault((n[0] >> 1) & 0x3F, []).append(n)
    body += bytes([len(groups)])
    for t in sorted(groups):
        body += bytes([0x80 | t]) + struct.pack(">H", len(groups[t]))
        for n in groups[t]:
            body += struct.pack(">H", len(n)) + n
    return _box(b"hvcC", body)


def _colr(nclx):
    prim, trc, mat, full = nclx
    return _box(b"colr", b"nclx" + struct.pack(">HHHB", prim, trc, mat, 0x80 if full else 0))


def _transform_box(t):
    """('irot', quarter_turns_ccw) | ('imir', axis) | ('clap', (wn, wd, hn, hd, hon, hod, von, vod))"""
    kind, v = t
    if kind == "irot":
        return _box(b"irot", bytes([v & 3]))
    if kind == "imir":
        return _box(b"imir", bytes([v & 1]))
    if kind == "clap":
        wn, wd, hn, hd, hon, hod, von, vod = v
        return _box(b"clap", struct.pack(">IIIIiIiI", wn, wd, hn, hd, hon, hod, von, vod))
    raise ValueError(kind)


def write_heic(pictures, size, grid=None, chroma_format=1, bit_depth=8, colr=None, sizes=None, transforms=None, aux=None, icc=None,
               tile_transforms=None):
    """pictures: list of [len][NAL] strings (each with VPS/SPS/PPS first); size: (w,h) of one picture
    (sizes: optional per-picture override of the declared ispe).
    grid: None for a single image, or (rows, cols, out_w, out_h).  colr: optional per-tile nclx tuple.
    icc: optional (b"prof" | b"rICC", profile bytes) attached to every coded image item as a second 'colr' box.
    tile_transforms: optional {picture index: [transforms]} - transformative properties of individual grid tile items."""
    items = []
    props = []
    assoc = {}
    index_of = {}

    def prop(box):  # identical properties are stored once and shared (as real writers do)
        if box not in index_of:
            props.append(box)
            index_of[box] = len(props)
        return index_of[box]

    for k, lp in enumerate(pictures):
        nals = split_nals(lp)
        params = [n for n in nals if ((n[0] >> 1) & 0x3F) in (32, 33, 34)]
        vcl = [n for n in nals if ((n[0] >> 1) & 0x3F) not in (32, 33, 34)]
        payload = b"".join(struct.pack(">I", len(n)) + n for n in vcl)
        a = [0x8000 | prop(_hvcc(params, chroma_format, bit_depth))]  # essential
        a.append(prop(_full(b"ispe", 0, 0, struct.pack(">II", *(sizes[k] if sizes else size)))))
        if colr is not None:
            a.append(prop(_colr(colr)))
        if icc is not None:
            a.append(prop(_box(b"colr", icc[0] + icc[1])))
        if tile_transforms and k in tile_transforms:
            a += [0x8000 | prop(_transform_box(t)) for t in tile_transforms[k]]
        items.append((k + 1, b"hvc1", payload))
        assoc[k + 1] = a
    primary = 1
    iref_boxes = []
    if grid is None and transforms:  # transformative properties of the single image, in order
        assoc[1] += [0x8000 | prop(_transform_box(t)) for t in transforms]
    if grid is not None:
        rows, cols, ow, oh = grid
        gid = len(pictures) + 1
        items.append((gid, b"grid", bytes([0, 0, rows - 1, cols - 1]) + struct.pack(">HH", ow, oh)))
        assoc[gid] = [prop(_full(b"ispe", 0, 0, struct.pack(">II", ow, oh)))]
        assoc[gid] += [0x8000 | prop(_transform_box(t)) for t in (transforms or [])]
        primary = gid
        iref_boxes.append(_box(b"dimg", struct.pack(">HH", gid, len(pictures)) +
                               b"".join(struct.pack(">H", k + 1) for k in range(len(pictures)))))
    wide = len(props) > 127  # ipma flags & 1: 15-bit property indices
    # auxiliary images (e.g. alpha): (coded picture, (w, h), aux type URN[, chroma_format_idc, bit depth, target item id]);
    # by default they belong to the primary item
    for entry in (aux or []):
        lp, asize, urn = entry[:3]
        aux_cf = entry[3] if len(entry) > 3 else chroma_format  # chroma_format_idc of the auxiliary picture (0 = monochrome)
        aid = len(items) + 1
        nals = split_nals(lp)
        params = [n for n in nals if ((n[0] >> 1) & 0x3F) in (32, 33, 34)]
        vcl = [n for n in nals if ((n[0] >> 1) & 0x3F) not in (32, 33, 34)]
        items.append((aid, b"hvc1", b"".join(struct.pack(">I", len(n)) + n for n in vcl)))
        aux_bd = entry[4] if len(entry) > 4 else bit_depth
        assoc[aid] = [0x8000 | prop(_hvcc(params, aux_cf, aux_bd)), prop(_full(b"ispe", 0, 0, struct.pack(">II", *asize))),
                      0x8000 | prop(_full(b"auxC", 0, 0, urn.encode() + b"\0"))]
        target = entry[5] if len(entry) > 5 else primary  # item the auxiliary image belongs to (6th element: e.g. a grid tile's item id = picture index + 1)
        iref_boxes.append(_box(b"auxl", struct.pack(">HHH", aid, 1, target)))
    iref = _full(b"iref", 0, 0, b"".join(iref_boxes)) if iref_boxes else b""
    ipma = struct.pack(">I", len(assoc))
    for iid in sorted(assoc):
        ipma += struct.pack(">HB", iid, len(assoc[iid]))
        for v in assoc[iid]:
            ipma += struct.pack(">H", v) if wide else bytes([(0x80 if v & 0x8000 else 0) | (v & 0x7F)])
    iprp = _box(b"iprp", _box(b"ipco", b"".join(props)) + _full(b"ipma", 0, 1 if wide else 0, ipma))
    hdlr = _full(b"hdlr", 0, 0, struct.pack(">I4s", 0, b"pict") + b"\0" * 13)
    pitm = _full(b"pitm", 0, 0, struct.pack(">H", primary))
    iinf = struct.pack(">H", len(items))
    for iid, typ, _ in items:
        iinf += _full(b"infe", 2, 0, struct.pack(">HH4s", iid, 0, typ) + b"\0")
    iinf = _full(b"iinf", 0, 0, iinf)
    ftyp = _box(b"ftyp", b"heic" + struct.pack(">I", 0) + b"mif1heic")

    def meta_with(offsets):
        iloc = bytes([0x44, 0x00]) + struct.pack(">H", len(items))
        for (iid, _, payload), off in zip(items, offsets):
            iloc += struct.pack(">HHH", iid, 0, 1) + struct.pack(">II", off, len(payload))
        return _full(b"meta", 0, 0, hdlr + pitm + _full(b"iloc", 0, 0, iloc) + iinf + iref + iprp)

    meta_len = len(meta_with([0] * len(items)))
    off = len(ftyp) + meta_len + 8
    offsets = []
    for _, _, payload in items:
        offsets.append(off)
        off += len(payload)
    return ftyp + meta_with(offsets) + _box(b"mdat", b"".join(p for _, _, p in items))
