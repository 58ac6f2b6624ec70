"""BASELINE.json configurations 2, 3, 4 and 5 at full size through the C ABI (hm_file_open -> hm_decode_item: box parse,
host entropy decode, GPU reconstruction / filters / paste / colour, D2H; config 3: hm_batch over many images),
bit-exact against the CPU flow (oracle restatement / reference decoder).
Config 3 over 8 GPUs is this batch sharded by image: bench.py --gpus N + tests/test_shard_gloo.py."""
import numpy as np
import pytest

import heifwriter
import orc
import pipeline
import synthutil
from corpus import TILE

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [dict(vui=1, full_range=1, matrix=6), dict(vui=0)], ids=["vui_full_range", "no_vui_limited_paste"])
@pytest.mark.parametrize("fmt", [10, 11])
def test_config2_12mp_grid(hm, variant, fmt):
    """4032x3024 output, 8x6 grid of 512x512 tiles (seeds 1200000 + i), right 64 / bottom 48 px cropped by the paste;
    the no-VUI variant takes the limited->full range paste rescale (context.cc:2504-2528)."""
    tiles = [synthutil.picture(1200000 + i, **TILE, **variant) for i in range(48)]
    data = heifwriter.write_heic(tiles, (512, 512), grid=(6, 8, 4032, 3024))
    f = pipeline.HeifFile(hm, data)
    info = f.info(f.primary())
    assert (info.is_grid, info.grid_rows, info.grid_cols, info.width, info.height) == (1, 6, 8, 4032, 3024)
    planes, meta = f.decode(f.primary(), fmt, threads=8)
    f.close()
    exp, stride, _ = pipeline.cpu_decode(hm, tiles, 512, 512, 4032, 3024, 8, True, fmt)
    bpp = 3 if fmt == 10 else 4
    assert meta["stride"][0] == stride and (meta["width"], meta["height"]) == (4032, 3024)
    np.testing.assert_array_equal(planes[0][:3024, :4032 * bpp], exp[:3024, :4032 * bpp])
    if orc.have_ref():  # ... and against the real libde265's tiles (the product's parser is not in this leg)
        exp2, _, _ = pipeline.cpu_decode(hm, tiles, 512, 512, 4032, 3024, 8, True, fmt, decoder="ref")
        np.testing.assert_array_equal(planes[0][:3024, :4032 * bpp], exp2[:3024, :4032 * bpp])


@pytest.mark.parametrize("nclx", [dict(full_range=0, matrix=9, primaries=9), dict(full_range=1, matrix=1, primaries=1)],
                         ids=["bt2020_limited", "bt709_full"])
@pytest.mark.parametrize("fmt", [14, 12], ids=["RRGGBB_LE", "RRGGBB_BE"])
def test_config4_10bit_422_single_image(hm, nclx, fmt):
    """2048x1536 single (non-grid) 10-bit 4:2:2 image -> interleaved 16-bit RGB (values stay 10-bit)."""
    pic = synthutil.picture(4220010, width=2048, height=1536, chroma_format=2, bit_depth=10, log2_ctb=5, qp=30, vui=1, **nclx)
    data = heifwriter.write_heic([pic], (2048, 1536), chroma_format=2, bit_depth=10)
    f = pipeline.HeifFile(hm, data)
    planes, meta = f.decode(f.primary(), fmt, threads=1)
    native, nmeta = f.decode(f.primary(), 0)
    f.close()
    assert (meta["bit_depth"], meta["chroma"]) == (10, 2)
    exp, stride, canv = pipeline.cpu_decode(hm, [pic], 2048, 1536, 2048, 1536, 1, False, fmt)
    assert meta["stride"][0] == stride
    np.testing.assert_array_equal(planes[0][:1536, :2048 * 6], exp[:1536, :2048 * 6])
    if orc.have_ref():  # the same image with the real libde265 as the decoder
        exp2, _, _ = pipeline.cpu_decode(hm, [pic], 2048, 1536, 2048, 1536, 1, False, fmt, decoder="ref")
        np.testing.assert_array_equal(planes[0][:1536, :2048 * 6], exp2[:1536, :2048 * 6])
    # native planar output = the decoder plugin's planes (decoder_libde265.cc:88-157)
    for c, (w, h) in enumerate(((2048, 1536), (1024, 1536), (1024, 1536))):
        np.testing.assert_array_equal(native[c][:h, :w * 2], canv[c][0][:h, :w * 2])


def test_config5_16384_grid(hm):
    """16384x16384 output, 32x32 grid of 512x512 tiles.  The 1024 tiles are drawn from six distinct coded pictures
    (seeds 5000000 + k), so every tile-sized region of the output must equal the single-tile result of its picture:
    a size-independent check of grid geometry, batching (1024 pictures in one launch) and paste at full scale."""
    pool = [synthutil.picture(5000000 + k, **TILE, vui=1, full_range=1, matrix=6) for k in range(6)]
    pick = [(t * 7 + t // 32) % 6 for t in range(1024)]
    data = heifwriter.write_heic([pool[k] for k in pick], (512, 512), grid=(32, 32, 16384, 16384))
    f = pipeline.HeifFile(hm, data)
    planes, meta = f.decode(f.primary(), 10, threads=16)
    f.close()
    assert (meta["width"], meta["height"]) == (16384, 16384)
    rgb = planes[0]
    expect = [pipeline.cpu_decode(hm, [p], 512, 512, 512, 512, 1, True, 10)[0][:512, :512 * 3] for p in pool]
    if orc.have_ref():  # the six pictures by the real libde265: the same tiles
        for k, p in enumerate(pool):
            assert np.array_equal(pipeline.cpu_decode(hm, [p], 512, 512, 512, 512, 1, True, 10, decoder="ref")[0][:512, :512 * 3], expect[k]), f"picture {k}: oracle != libde265"
    for t in range(1024):
        r, c = divmod(t, 32)
        got = rgb[r * 512:(r + 1) * 512, c * 1536:(c + 1) * 1536]
        assert np.array_equal(got, expect[pick[t]]), f"tile {t} (row {r}, col {c}) differs"


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0], "all"], ids=["two_slabs", "three_slabs", "every_gpu_of_the_box"])
@pytest.mark.parametrize("ext", [False, True], ids=["pinned_plane", "ext_dst"])
def test_one_grid_over_several_devices(hm, devices, ext):
    """hm_decode_item_devices: ONE grid cut into slabs of tile rows, one per listed device, each decoded and converted on
    its device (own batch, own stream, own host thread) and copied straight into its rows of the destination - here the
    one GPU of the box listed several times: the slabs run side by side on it.  BASELINE config 5's shape (16384 x 16384,
    32 x 32 tiles; 24 distinct tiles as in the config-5 test) must come out bit for bit as the one-device decode; a second,
    cropped grid (5 tile rows for 2 / 3 devices: uneven slabs, a last slab cut by the canvas) likewise, also into ext_dst."""
    import ctypes as C
    if devices == "all":  # (r05) the case that scales itself to the box: a slab per GPU that is there
        import torch
        n = min(torch.cuda.device_count(), 8)
        if n < 2:
            pytest.skip("one GPU visible: the slabs of the other cases share it")
        devices = list(range(n))
    pool = [synthutil.picture(5000000 + i, **TILE, vui=1, full_range=1, matrix=6) for i in range(24)]
    for (rows, cols, w, h) in ((32, 32, 16384, 16384), (5, 3, 1500, 2300)):
        if ext and rows == 32:
            continue  # (the large grid once is enough)
        pick = [(7 * t + 3 * (t // cols)) % 24 for t in range(rows * cols)]
        data = heifwriter.write_heic([pool[k] for k in pick], (512, 512), grid=(rows, cols, w, h))
        f = pipeline.HeifFile(hm, data)
        fmt = 11 if ext else 10
        one, meta1 = f.decode(f.primary(), fmt, threads=16)
        bpp = 4 if ext else 3
        dev = (C.c_int32 * len(devices))(*devices)
        d = pipeline.Decoded()
        buf = None
        if ext:
            stride = w * 4 + 64
            buf = np.zeros((h, stride), dtype=np.uint8)
            prm = pipeline.DecodeParams(fmt, 16, 0, 0, None, buf.ctypes.data_as(C.c_void_p), buf.size, stride, 0, 0)
        else:
            prm = pipeline.DecodeParams(fmt, 16, 0, 0, None, None, 0, 0, 0, 0)
        hm.hm_decode_item_devices.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p]
        rc = hm.hm_decode_item_devices(f.h, f.primary(), C.byref(prm), dev, len(devices), C.byref(d))
        assert rc == 0, hm.hm_last_error()
        assert (d.width, d.height, d.out_format, d.has_nclx, d.primaries, d.transfer, d.matrix, d.full_range) == (w, h, fmt, 1, 1, 13, 6, 1)
        if ext:
            assert d.used_ext_dst == 1 and not d.plane[0]
            got = buf
        else:
            assert d.stride[0] == meta1["stride"][0]
            got = np.ctypeslib.as_array(d.plane[0], shape=(h, d.stride[0]))
        assert np.array_equal(got[:h, :w * bpp], one[0][:h, :w * bpp]), f"{rows}x{cols} grid over {len(devices)} slabs differs from the one-device decode"
        hm.hm_decoded_free(C.byref(d))
        f.close()


@pytest.mark.parametrize("case", ["8bit_rgb24", "8bit_rgba_ext_dst", "10bit_rrggbb"])
def test_a_grid_decoded_slab_by_slab_equals_the_one_batch_decode(hm_hooks, case):
    """r06: hm_decode_item takes a grid of more tiles than parsing threads in slabs of tile rows, each queued on a stream of its own as
    soon as its tiles are through the entropy decode (hm_image.cpp: decode_grid_cut, pipelined) - the pixels must be those of the one
    batch behind the whole entropy decode (knob grid_slab_rows = 0, the path before r06), whatever the slab height: the default (up to
    eight slabs), two rows, a height that does not divide the grid's rows.  5 x 3 tiles of 512 x 512, the canvas cropped inside the last
    tile row and column; the same metadata."""
    import ctypes as C
    L = hm_hooks
    bd = 10 if case.startswith("10bit") else 8
    fmt = {"8bit_rgb24": 10, "8bit_rgba_ext_dst": 11, "10bit_rrggbb": 14}[case]  # HM_OUT_RGB / RGBA / RRGGBB_LE
    bpp = {10: 3, 11: 4, 14: 6}[fmt]
    rows, cols, w, h = 5, 3, 1500, 2300
    pool = [synthutil.picture(5100000 + i, **dict(TILE, bit_depth=bd), vui=1, full_range=1, matrix=6) for i in range(8)]
    data = heifwriter.write_heic([pool[(5 * t + 3 * (t // cols)) % 8] for t in range(rows * cols)], (512, 512), grid=(rows, cols, w, h), bit_depth=bd)
    f = pipeline.HeifFile(L, data)
    ext = case.endswith("ext_dst")

    def decode(slab_rows):
        assert L.hm_debug_set(b"grid_slab_rows", slab_rows) == 0
        if not ext:
            img, meta = f.decode(f.primary(), fmt, threads=4)
            return img[0][:h, :w * bpp].copy(), {k: v for k, v in meta.items() if k != "stride"}
        stride = w * bpp + 64
        buf = np.zeros((h, stride), dtype=np.uint8)
        prm = pipeline.DecodeParams(fmt, 4, 0, 0, None, buf.ctypes.data_as(C.c_void_p), buf.size, stride, 0, 0)
        d = pipeline.Decoded()
        assert L.hm_decode_item(f.h, f.primary(), C.byref(prm), C.byref(d)) == 0, L.hm_last_error()
        meta = (d.width, d.height, d.out_format, d.used_ext_dst, d.bit_depth, d.has_nclx, d.primaries, d.transfer, d.matrix, d.full_range, d.warnings)
        L.hm_decoded_free(C.byref(d))
        return buf[:, :w * bpp].copy(), meta
    try:
        one, meta1 = decode(0)
        assert one.any()
        for slab_rows in (-1, 2, 3, 4):
            got, meta = decode(slab_rows)
            assert meta == meta1, (slab_rows, meta, meta1)
            assert np.array_equal(got, one), f"{case}: slabs of {slab_rows} tile rows differ from the one-batch decode"
    finally:
        L.hm_debug_set(b"grid_slab_rows", -1)
        f.close()


def test_a_broken_tile_fails_the_slabbed_decode_like_the_one_batch_decode(hm_hooks):
    """strict decoding, a grid of 4 x 3 tiles whose tile 7 (third tile row) has its slice data cut short: hm_decode_item must fail with the SAME status and
    message whether the grid goes to the device in slabs under the entropy decode (the slabs in front of the broken one are already queued then: they are
    drained before the call returns) or as one batch behind it; without strict decoding both give the same concealed image."""
    import ctypes as C
    import hevcutil
    L = hm_hooks
    small = dict(TILE, width=256, height=256)
    tiles = [synthutil.picture(6900000 + i, **small, vui=1, full_range=1, matrix=6) for i in range(12)]
    nals = hevcutil.split_nals(tiles[7])
    tiles[7] = hevcutil.join_nals(nals[:-1] + [nals[-1][:len(nals[-1]) // 2]])
    f = pipeline.HeifFile(L, heifwriter.write_heic(tiles, (256, 256), grid=(4, 3, 768, 1024)))
    got = {}
    try:
        for slab_rows in (0, -1, 2):
            assert L.hm_debug_set(b"grid_slab_rows", slab_rows) == 0
            prm = pipeline.DecodeParams(10, 4, 0, 0, None, None, 0, 0, 1, 0)  # strict
            d = pipeline.Decoded()
            rc = L.hm_decode_item(f.h, f.primary(), C.byref(prm), C.byref(d))
            got[slab_rows] = (rc, L.hm_last_error().decode())
            assert rc != 0
            img, meta = f.decode(f.primary(), 10, threads=4)  # concealing
            got[("img", slab_rows)] = (img[0][:1024, :768 * 3].copy(), meta["warnings"])
    finally:
        L.hm_debug_set(b"grid_slab_rows", -1)
        f.close()
    assert got[0] == got[-1] == got[2], got
    assert "tile 7" in got[0][1]
    for slab_rows in (-1, 2):
        assert got[("img", slab_rows)][1] == got[("img", 0)][1] and (got[("img", 0)][1] & 8)
        assert np.array_equal(got[("img", slab_rows)][0], got[("img", 0)][0])


def test_grids_decoded_by_several_threads_at_once(hm):
    """r06: hm_decode_item on a grid runs a queueing thread of its own beside the parsing crew (slabs under the entropy decode) - three caller
    threads decode three different grids at the same time, four times each: every image equals the one decoded alone (the crew takes one
    fan-out at a time, the slabs' streams come from the shared per-device cache)."""
    import threading
    small = dict(TILE, width=256, height=256)
    files, want = [], []
    for k in range(3):
        tiles = [synthutil.picture(6800000 + 10 * k + i, **small, vui=1, full_range=1, matrix=6) for i in range(12)]
        f = pipeline.HeifFile(hm, heifwriter.write_heic(tiles, (256, 256), grid=(4, 3, 760, 1000)))
        files.append(f)
        want.append(f.decode(f.primary(), 10, threads=4)[0][0][:1000, :760 * 3].copy())
    errors = []

    def work(k):
        try:
            for _ in range(4):
                got = files[k].decode(files[k].primary(), 10, threads=4)[0][0][:1000, :760 * 3]
                if not np.array_equal(got, want[k]):
                    errors.append(f"grid {k} differs")
        except Exception as e:  # noqa: BLE001
            errors.append(f"grid {k}: {e}")
    threads = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a decode hangs"
    assert not errors, errors
    for f in files:
        f.close()


def test_items_that_do_not_cut_fall_back_to_the_first_device(hm):
    """a single image, and a grid asked for planar output: hm_decode_item_devices decodes them on devices[0] like hm_decode_item"""
    import ctypes as C
    pic = synthutil.picture(4242, **TILE, vui=1, full_range=1, matrix=6)
    f = pipeline.HeifFile(hm, heifwriter.write_heic([pic], (512, 512)))
    one, _ = f.decode(f.primary(), 10, threads=2)
    dev = (C.c_int32 * 2)(0, 0)
    d = pipeline.Decoded()
    prm = pipeline.DecodeParams(10, 2, 0, 0, None, None, 0, 0, 0, 0)
    hm.hm_decode_item_devices.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p]
    assert hm.hm_decode_item_devices(f.h, f.primary(), C.byref(prm), dev, 2, C.byref(d)) == 0
    got = np.ctypeslib.as_array(d.plane[0], shape=(512, d.stride[0]))
    assert np.array_equal(got[:, :1536], one[0][:512, :1536])
    hm.hm_decoded_free(C.byref(d))
    bad = (C.c_int32 * 2)(0, 99)
    assert hm.hm_decode_item_devices(f.h, f.primary(), C.byref(prm), bad, 2, C.byref(d)) != 0  # a device that does not exist
    f.close()


@pytest.mark.parametrize("case", ["tiles_do_not_cover", "tile_row_below_canvas", "tiles_of_different_sizes"])
def test_grids_with_bad_geometry_are_not_cut(hm, case):
    """the whole-grid checks of the one-device path (context.cc:2299-2359: equal tile sizes, the canvas covered, every tile's
    origin inside it) happen before hm_decode_item_devices cuts a grid into slabs - a slab only sees its own reduced canvas
    (r04 advice: such a grid returned HM_OK with rows of the destination never written).  The call reports what
    hm_decode_item reports."""
    import ctypes as C
    small = dict(TILE, width=128, height=128)
    pics = [synthutil.picture(6600000 + i, **small, vui=1, full_range=1, matrix=6) for i in range(6)]
    if case == "tiles_do_not_cover":      # 3 rows of 128 declared for a canvas of 600 rows: 128 < 600 / 3
        data = heifwriter.write_heic(pics, (128, 128), grid=(3, 2, 256, 600))
    elif case == "tile_row_below_canvas":  # the third tile row starts at y = 256 = the canvas height
        data = heifwriter.write_heic(pics, (128, 128), grid=(3, 2, 256, 256))
    else:
        data = heifwriter.write_heic(pics, (128, 128), grid=(3, 2, 256, 384), sizes=[(128, 128)] * 5 + [(128, 120)])
    f = pipeline.HeifFile(hm, data)
    prm = pipeline.DecodeParams(10, 2, 0, 0, None, None, 0, 0, 0, 0)
    d1, d2 = pipeline.Decoded(), pipeline.Decoded()
    rc1 = hm.hm_decode_item(f.h, f.primary(), C.byref(prm), C.byref(d1))
    msg1 = hm.hm_last_error()
    dev = (C.c_int32 * 2)(0, 0)
    hm.hm_decode_item_devices.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_void_p]
    rc2 = hm.hm_decode_item_devices(f.h, f.primary(), C.byref(prm), dev, 2, C.byref(d2))
    msg2 = hm.hm_last_error()
    assert rc1 != 0 and rc2 == rc1 and msg2 == msg1, (case, rc1, msg1, rc2, msg2)
    f.close()


def test_a_grid_with_a_damaged_tile_comes_back_like_the_reference_gives_it_back(hm):
    """r05 (VERDICT r04 "missing" 2): one tile of a 3 x 2 grid has its slice data cut short.  The reference returns the image - the
    damaged tile decoded up to the damage (decctx.cc:876-995, decoder_libde265.cc:311-336).  hm_decode_item does the same unless
    strict decoding is asked: the image comes back with HM_WARN_CONCEALED, every other tile bit for bit as in the intact grid, the
    damaged tile equal to the oracle's picture of the concealing parse; with strict decoding the call fails as before r05."""
    import hevcutil
    small = dict(TILE, width=256, height=256)
    tiles = [synthutil.picture(6700000 + i, **small, vui=1, full_range=1, matrix=6) for i in range(6)]
    cut = list(tiles)
    # ([length][NAL] framing stays whole: only the last NAL - the slice - loses its tail)
    nals = hevcutil.split_nals(tiles[4])
    cut[4] = hevcutil.join_nals(nals[:-1] + [nals[-1][:len(nals[-1]) * 2 // 3]])
    f_ok = pipeline.HeifFile(hm, heifwriter.write_heic(tiles, (256, 256), grid=(2, 3, 768, 512)))
    f_cut = pipeline.HeifFile(hm, heifwriter.write_heic(cut, (256, 256), grid=(2, 3, 768, 512)))
    good, m0 = f_ok.decode(f_ok.primary(), 10, threads=4)
    got, m1 = f_cut.decode(f_cut.primary(), 10, threads=4)
    assert m0["warnings"] == 0 and (m1["warnings"] & 8), (m0["warnings"], m1["warnings"])  # HM_WARN_CONCEALED
    for t in range(6):
        r, c = divmod(t, 3)
        a, b = got[0][r * 256:(r + 1) * 256, c * 768:(c + 1) * 768], good[0][r * 256:(r + 1) * 256, c * 768:(c + 1) * 768]
        assert np.array_equal(a, b) == (t != 4), f"tile {t}"
    exp, _, _ = pipeline.cpu_decode(hm, [cut[4]], 256, 256, 256, 256, 1, True, 10, decoder="oracle_concealing")
    assert np.array_equal(got[0][256:512, 768:1536], exp[:256, :768])
    with pytest.raises(RuntimeError, match="tile 4"):
        f_cut.decode(f_cut.primary(), 10, threads=4, strict=1)
    f_ok.close()
    f_cut.close()


def test_config3_batch_of_12mp_grids(pkg, hm):
    """BASELINE config 3 on one rank: 32 DIFFERENT 12 MP grids (image j: tiles 1200000 + 48 j + i) in ONE hm_batch, one
    batched colour conversion - EVERY image compared with the CPU flow (the real reference decoder oracle/_ref for the
    tiles where it is present, else the oracle's executors; paste + colour by the oracle)."""
    import ctypes as C
    import bench
    import orc
    import torch
    n_images = 32
    dev = torch.device("cuda:0")
    gb = bench.GridBatch(pkg, dev, 8, 6, 512, 4032, 3024)
    kept = []
    made = bench.make_streams(pkg.capi, (1200000 + k for k in range(n_images * 48)))
    for j in range(n_images):
        tiles = [next(made) for _ in range(48)]
        gb.add_image([b for _, b in tiles])
        kept.append(tiles)
    st = torch.cuda.current_stream().cuda_stream
    gb.finish(st)
    gb.step(st)
    torch.cuda.synchronize()
    use_ref = orc.have_ref()
    for j in range(n_images):
        got = gb.images[j]["rgb"].cpu().numpy()
        exp = bench.cpu_grid_image([d for d, _ in kept[j]], [b for _, b in kept[j]], 8, 6, 512, 4032, 3024, (gb.ys, gb.cs, gb.os), use_ref)
        assert np.array_equal(got[:3024, :4032 * 3], exp[:3024, :4032 * 3]), f"image {j} of the batch differs"
    # image sharding over ranks: the ranks' images are exactly these (bench.py --gpus N decodes images [rank B, rank B + B))
    sh = pkg.shard
    assert [list(sh.image_shard(1024, r, 8))[0] for r in range(8)] == [128 * r for r in range(8)]


@pytest.mark.parametrize("shape", [(8, 6, 4032, 3024), (3, 2, 1500, 1000), (2, 2, 1001, 999), (2, 2, 1024, 1024), (1, 1, 512, 512)],
                         ids=["12mp_crop64x48", "crop_not_16_aligned", "odd_canvas", "no_crop", "single_tile"])
@pytest.mark.parametrize("stages", [3, 1, 2, 0], ids=["deblock+sao", "deblock", "sao", "none"])
def test_fused_tail_equals_separate_kernels(pkg, hm, shape, stages):
    """hm_batch_set_colour: the fused kernel (deblocking + SAO + paste + colour, filters.hip k_tail420) against the four
    separate kernels on the same batch - 3 images of different tiles, every filter-stage combination, canvases cropped
    at 16-aligned and unaligned widths; the 12 MP shape is also checked against the CPU flow."""
    import bench
    import orc
    import torch
    cols, rows, w, h = shape
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n_images = 3
    made = list(bench.make_streams(pkg.capi, (7700000 + 31 * k for k in range(n_images * cols * rows))))
    out = []
    for group in (0, -1):  # 0: fused where possible, -1: never
        gb = bench.GridBatch(pkg, dev, cols, rows, 512, w, h)
        for j in range(n_images):
            gb.add_image([b for _, b in made[j * cols * rows:(j + 1) * cols * rows]])
        gb.finish(st, group)
        gb.batch.execute(stages, st)
        torch.cuda.synchronize()
        assert gb.batch.tail_fused() == (group == 0)
        out.append([im["rgb"].cpu().numpy()[:h, :w * 3].copy() for im in gb.images])
        gb.batch.close()
    for j in range(n_images):
        assert np.array_equal(out[0][j], out[1][j]), f"image {j}: fused tail differs from the separate kernels"
    if shape[2] == 4032 and stages == 3:
        tiles = made[:cols * rows]
        exp = bench.cpu_grid_image([d for d, _ in tiles], [b for _, b in tiles], cols, rows, 512, w, h, (gb.ys, gb.cs, gb.os), orc.have_ref())
        assert np.array_equal(out[0][0], exp[:h, :w * 3])


@pytest.mark.parametrize("shape", [(8, 6, 4032, 3024), (3, 2, 1500, 1000)], ids=["12mp_crop64x48", "crop_not_16_aligned"])
def test_fused_tail_with_limited_range_paste(pkg, hm, shape):
    """tiles without a VUI (the decoder plugin then reports limited range, matrix 2: decoder_libde265.cc:339-362) are rescaled
    to full range while they are pasted (context.cc:2504-2528) - the class of the reference's own examples/example.heic.  r05: part
    of the fused kernel (integer form of the float expression, filters.hip pk_rescale): fused, equal to the separate kernels,
    and equal to the CPU flow (oracle / real libde265 tiles, the oracle's float paste, integer matrix)."""
    import bench
    import torch
    cols, rows, w, h = shape
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n_images = 2
    made = list(bench.make_streams(pkg.capi, (7600000 + 29 * k for k in range(n_images * cols * rows)), vui=0))
    out = []
    for group in (0, -1):  # 0: fused where possible, -1: never
        gb = bench.GridBatch(pkg, dev, cols, rows, 512, w, h, nclx=(1, 0, 2))
        for j in range(n_images):
            gb.add_image([b for _, b in made[j * cols * rows:(j + 1) * cols * rows]])
        gb.finish(st, group)
        gb.batch.execute(3, st)
        torch.cuda.synchronize()
        gb.batch.check()
        assert gb.batch.tail_fused() == (group == 0)
        out.append([im["rgb"].cpu().numpy()[:h, :w * 3].copy() for im in gb.images])
        strides = (gb.ys, gb.cs, gb.os)
        gb.batch.close()
    for j in range(n_images):
        assert np.array_equal(out[0][j], out[1][j]), f"image {j}: fused tail differs from the separate kernels"
        tiles = made[j * cols * rows:(j + 1) * cols * rows]
        for use_ref in [False] + ([True] if orc.have_ref() else []):
            exp = bench.cpu_grid_image([d for d, _ in tiles], [b for _, b in tiles], cols, rows, 512, w, h, strides, use_ref)
            assert np.array_equal(out[0][j], exp[:h, :w * 3]), f"image {j}: differs from the CPU flow (libde265: {use_ref})"
    # the rescale really happened: a full-range paste of the same tiles gives other pixels
    gb = bench.GridBatch(pkg, dev, cols, rows, 512, w, h, nclx=(1, 1, 2))
    gb.add_image([b for _, b in made[:cols * rows]])
    gb.finish(st, 0)
    gb.batch.execute(3, st)
    torch.cuda.synchronize()
    assert not np.array_equal(gb.images[0]["rgb"].cpu().numpy()[:h, :w * 3], out[0][0])
    gb.batch.close()


@pytest.mark.parametrize("slicing", [dict(slices=40), dict(slices=60, dependent=400, slice_lf_random=1, deblock_override=1, slice_sao_random=1, slice_qp_random=1),
                                     dict(slices=60, pps_lf_across_slices_off=1, slice_lf_random=1)],
                         ids=["slices", "slice_headers", "filters_stop_at_slices"])
def test_fused_tail_with_several_slices(pkg, hm, slicing):
    """pictures of several slices (own deblocking / SAO switches and offsets per slice) through the fused kernel: equal to the
    separate kernels; fused whenever no CTB needs the per-sample SAO ring test (always when the filters cross slice borders).
    r06: ... and both equal to the CPU flow - the oracle's filters and, where it is built, the real libde265's tiles, then the oracle's paste
    and integer matrix -, so that the fused tail and the separate kernels cannot drift TOGETHER."""
    import bench
    import torch
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    cols, rows, w, h = 3, 2, 1500, 1000
    made = list(bench.make_streams(pkg.capi, (7900000 + 13 * k for k in range(2 * cols * rows)), **slicing))
    out, fused = [], []
    for group in (0, -1):
        gb = bench.GridBatch(pkg, dev, cols, rows, 512, w, h)
        for j in range(2):
            gb.add_image([b for _, b in made[j * cols * rows:(j + 1) * cols * rows]])
        gb.finish(st, group)
        gb.batch.execute(3, st)
        torch.cuda.synchronize()
        fused.append(gb.batch.tail_fused())
        out.append([im["rgb"].cpu().numpy()[:h, :w * 3].copy() for im in gb.images])
        strides = (gb.ys, gb.cs, gb.os)
        gb.batch.close()
    assert not fused[1]
    if not slicing.get("pps_lf_across_slices_off"):
        assert fused[0], "pictures whose filters cross slice borders take the fused tail"
    for j in range(2):
        assert np.array_equal(out[0][j], out[1][j]), f"image {j}: fused tail differs from the separate kernels"
        tiles = made[j * cols * rows:(j + 1) * cols * rows]
        for use_ref in [False] + ([True] if orc.have_ref() else []):
            exp = bench.cpu_grid_image([d for d, _ in tiles], [b for _, b in tiles], cols, rows, 512, w, h, strides, use_ref)
            assert np.array_equal(out[0][j], exp[:h, :w * 3]), f"image {j}: differs from the CPU flow (libde265: {use_ref})"


FLOAT_TAIL_CLASSES = {
    # name: (bit depth, chroma format, full range, matrix, output format, bytes per pixel)
    "config4_422_10_rrggbb_le": (10, 2, 0, 9, "HM_OUT_RRGGBB_LE", 6),
    "420_10_rrggbb_be": (10, 1, 1, 1, "HM_OUT_RRGGBB_BE", 6),
    "420_10_rgb24": (10, 1, 0, 6, "HM_OUT_RGB", 3),
    "422_12_rrggbb_le": (12, 2, 1, 6, "HM_OUT_RRGGBB_LE", 6),
    "422_8_rgb24": (8, 2, 1, 6, "HM_OUT_RGB", 3),
    "420_8_limited_rgba": (8, 1, 0, 1, "HM_OUT_RGBA", 4),
    # r05: deep full-range 4:2:0 -> RGB24 / RGBA32 is Op_to_sdr_planes + the INTEGER 4:2:0 operation (the class of 10-bit HDR
    # photographs): mode 4 of the same fused kernel
    "420_10_full_rgb24": (10, 1, 1, 6, "HM_OUT_RGB", 3),
    "420_12_full_rgba": (12, 1, 1, 1, "HM_OUT_RGBA", 4),
    "420_10_full_bt2020_rgb24": (10, 1, 1, 9, "HM_OUT_RGB", 3),
    # grids as decode_full_grid_image builds them: the tiles carry their nclx, the canvas none - limited-range tiles are rescaled
    # while they are pasted (context.cc:2504-2528; 16-bit storage byte by byte: quirk Q1), part of the fused kernel since r05
    "grid_422_8_limited_rgb24": (8, 2, 0, 1, "HM_OUT_RGB", 3, True),
    "grid_420_10_limited_rrggbb_le": (10, 1, 0, 9, "HM_OUT_RRGGBB_LE", 6, True),
    "grid_422_10_full_rrggbb_be": (10, 2, 1, 9, "HM_OUT_RRGGBB_BE", 6, True),
    "grid_420_10_full_rgb24": (10, 1, 1, 9, "HM_OUT_RGB", 3, True),
    "grid_420_10_limited_rgb24": (10, 1, 0, 9, "HM_OUT_RGB", 3, True),
    # r06: the HDR class runs on k_tail420's 16-bit instantiation (9..11 bits; 12-bit pictures keep k_tailf) - its CTB-16 variant (SAO
    # parameters per lane), CTBs of 64, RGBA32, the other depths (8th element: log2 of the CTB size)
    "hdr_420_10_full_rgba": (10, 1, 1, 9, "HM_OUT_RGBA", 4, False, 5),
    "hdr_420_10_ctb16_rgb24": (10, 1, 1, 9, "HM_OUT_RGB", 3, False, 4),
    "hdr_420_10_ctb16_grid_rgba": (10, 1, 1, 1, "HM_OUT_RGBA", 4, True, 4),
    "hdr_420_10_ctb64_rgb24": (10, 1, 1, 6, "HM_OUT_RGB", 3, False, 6),
    "hdr_420_9_full_rgb24": (9, 1, 1, 9, "HM_OUT_RGB", 3, False, 5),
    "hdr_420_11_ctb16_rgba": (11, 1, 1, 9, "HM_OUT_RGBA", 4, False, 4),
    "hdr_420_11_ctb64_grid_rgb24": (11, 1, 1, 1, "HM_OUT_RGB", 3, True, 6),
}


@pytest.mark.parametrize("name", list(FLOAT_TAIL_CLASSES))
@pytest.mark.parametrize("stages", [3, 0], ids=["deblock+sao", "none"])
def test_fused_float_tail_equals_separate_kernels(pkg, hm, name, stages):
    """hm_batch_set_colour on the classes whose colour chain is the reference's float operation (10 / 12 bit, 4:2:2, limited
    range; BASELINE config 4 is the first): the fused kernel k_tailf (deblocking + SAO + paste + float matrix + repack) against
    k_deblock + k_sao_paste + k_ycbcr_float on the same batch - two images of 3 x 2 tiles of 256 x 192, the canvas cropped."""
    import ctypes as C
    import torch
    capi, L = pkg.capi, pkg.lib()
    bd, cf, full, matrix, fmt, obpp = FLOAT_TAIL_CLASSES[name][:6]
    grid = len(FLOAT_TAIL_CLASSES[name]) > 6 and FLOAT_TAIL_CLASSES[name][6]
    log2_ctb = FLOAT_TAIL_CLASSES[name][7] if len(FLOAT_TAIL_CLASSES[name]) > 7 else 5
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    cols, rows, tw, th, w, h = 3, 2, 256, 192, 700, 330
    bps = 2 if bd > 8 else 1
    datas = [synthutil.picture(8100000 + 7 * k, width=tw, height=th, chroma_format=cf, bit_depth=bd, log2_ctb=log2_ctb, qp=28, vui=1,
                               full_range=full, matrix=matrix, primaries=1, slices=(30 if k % 3 == 0 else 0)) for k in range(2 * cols * rows)]
    blobs = [capi.parse_hevc(d) for d in datas]
    ys, cs, os_ = L.hm_plane_stride(w, bps), L.hm_plane_stride((w + 1) // 2, bps), L.hm_plane_stride(w, obpp)
    ch = (h + 1) // 2 if cf == 1 else h
    out = []
    for group in (0, -1):  # 0: fused where possible, -1: never
        batch = capi.Batch()
        ims = []
        for j in range(2):
            im = dict(y=torch.zeros((max(64, h), ys), dtype=torch.uint8, device=dev), cb=torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev),
                      cr=torch.zeros((max(64, ch), cs), dtype=torch.uint8, device=dev), rgb=torch.zeros((max(64, h), os_), dtype=torch.uint8, device=dev))
            for t in range(cols * rows):
                d = capi.TileDest()
                d.plane[0], d.plane[1], d.plane[2] = im["y"].data_ptr(), im["cb"].data_ptr(), im["cr"].data_ptr()
                d.pitch[0], d.pitch[1], d.pitch[2] = ys, cs, cs
                d.canvas_width, d.canvas_height = w, h
                d.x0, d.y0 = (t % cols) * tw, (t // cols) * th
                # (tile_has_nclx: 0 = the canvas carries the tiles' profile, nothing is rescaled; the grid classes: the tiles' own
                #  profile - a limited-range tile is rescaled while it is pasted, context.cc:2504-2509)
                if grid:
                    d.tile_has_nclx, d.tile_full_range, d.tile_matrix = 1, full, matrix
                batch.add(blobs[j * cols * rows + t], d)
            ims.append(im)
        batch.upload(st)
        PtrArr = C.c_void_p * 2
        ptrs = [PtrArr(*[im[k].data_ptr() for im in ims]) for k in ("y", "cb", "cr", "rgb")]
        desc = capi.ColourDesc(w, h, bd, cf, 0 if grid else 1, matrix, 1, full, getattr(capi, fmt), ys, cs, cs, os_)
        batch.set_colour(desc, 2, *ptrs, group)
        batch.execute(stages, st)
        torch.cuda.synchronize()
        batch.check()
        assert batch.tail_fused() == (group == 0), name
        out.append([im["rgb"].cpu().numpy()[:h, :w * obpp].copy() for im in ims])
        batch.close()
    for j in range(2):
        assert np.array_equal(out[0][j], out[1][j]), f"{name}, image {j}: fused float tail differs from the separate kernels"
    assert out[0][0].any()
    if stages == 3:
        # ... and against the CPU flow: the tiles decoded by the oracle's executors and by the real libde265, pasted by the oracle
        # (no rescale: the canvas carries the tiles' own profile), converted along the chain the reference's search picks
        for j in range(2):
            tiles = datas[j * cols * rows:(j + 1) * cols * rows]
            for decoder in ["oracle"] + (["ref"] if orc.have_ref() else []):
                exp, _, _ = pipeline.cpu_decode(hm, tiles, tw, th, w, h, cols, grid, getattr(capi, fmt), decoder=decoder)
                assert np.array_equal(out[0][j], exp[:h, :w * obpp]), f"{name}, image {j}: fused float tail differs from the CPU flow ({decoder})"


@pytest.mark.parametrize("groups", [2, 3, 8])
def test_grouped_streams_equal_single_stream(pkg, hm, groups):
    """hm_batch_set_concurrency: the images of a step as `groups` groups on streams of their own - the same pixels as the
    single-stream step, execute after execute (the streams are joined on the caller's), also with fewer images than groups"""
    import bench
    import torch
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    n_images = 5
    made = list(bench.make_streams(pkg.capi, (7800000 + 17 * k for k in range(n_images * 4))))
    gb = bench.GridBatch(pkg, dev, 2, 2, 512, 1000, 900)
    for j in range(n_images):
        gb.add_image([b for _, b in made[j * 4:(j + 1) * 4]])
    gb.finish(st, 0)
    gb.batch.execute(3, st)
    torch.cuda.synchronize()
    assert gb.batch.tail_fused()
    want = [im["rgb"].clone() for im in gb.images]
    gb.batch.set_concurrency(groups)
    for _ in range(3):
        for im in gb.images:
            im["rgb"].zero_()
        gb.batch.execute(3, st)
        torch.cuda.synchronize()
        for j in range(n_images):
            assert torch.equal(gb.images[j]["rgb"], want[j]), f"image {j}"
    gb.batch.set_concurrency(0)
    gb.batch.execute(3, st)
    torch.cuda.synchronize()
    assert all(torch.equal(gb.images[j]["rgb"], want[j]) for j in range(n_images))
    gb.batch.close()
