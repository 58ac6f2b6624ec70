"""ctypes view of the decoder plugin (struct heif_decoder_plugin, heif_plugin.h:53-112 of the reference, fork ABI
new_decoder(void**, int)) exported by libheif_mi355x_api.so, and the call sequence HeifContext::decode_image_planar
drives it with (context.cc:1787-1835): new_decoder -> set_strict_decoding -> push_data -> decode_image -> free_decoder.
Used by the facade tests and by bench.py's plugin_path leg."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Err(C.Structure):
    _fields_ = [("code", C.c_int), ("subcode", C.c_int), ("message", C.c_char_p)]


class Plugin(C.Structure):
    _fields_ = [("plugin_api_version", C.c_int), ("get_plugin_name", C.c_void_p), ("init_plugin", C.c_void_p),
                ("deinit_plugin", C.c_void_p), ("does_support_format", C.CFUNCTYPE(C.c_int, C.c_int)),
                ("new_decoder", C.CFUNCTYPE(Err, C.POINTER(C.c_void_p), C.c_int)), ("free_decoder", C.CFUNCTYPE(None, C.c_void_p)),
                ("push_data", C.CFUNCTYPE(Err, C.c_void_p, C.c_char_p, C.c_size_t)),
                ("decode_image", C.CFUNCTYPE(Err, C.c_void_p, C.POINTER(C.c_void_p))),
                ("set_strict_decoding", C.CFUNCTYPE(None, C.c_void_p, C.c_int)), ("id_name", C.c_char_p)]


def load_api(pkg):
    pkg.lib()  # loads torch's HIP runtime + the core library first
    a = C.CDLL(os.path.join(ROOT, "heif-decoder-lib_amd", "libheif_mi355x_api.so"))
    a.hm_get_decoder_plugin.restype = C.POINTER(Plugin)
    a.heif_image_release.argtypes = [C.c_void_p]
    a.heif_image_get_plane_readonly.restype = C.POINTER(C.c_uint8)
    a.heif_image_get_plane_readonly.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    a.heif_image_get_width.argtypes = [C.c_void_p, C.c_int]
    a.heif_image_get_height.argtypes = [C.c_void_p, C.c_int]
    return a


def decode_tile(plugin, data, nthreads=0):
    """one decoder instance for one coded picture ([u32 length][NAL] records): returns the heif_image* (caller releases)"""
    dec = C.c_void_p()
    e = plugin.new_decoder(C.byref(dec), nthreads)
    if e.code:
        raise RuntimeError(f"new_decoder: {e.message}")
    try:
        plugin.set_strict_decoding(dec, 0)
        e = plugin.push_data(dec, data, len(data))
        if e.code:
            raise RuntimeError(f"push_data: {e.message}")
        img = C.c_void_p()
        e = plugin.decode_image(dec, C.byref(img))
        if e.code:
            raise RuntimeError(f"decode_image: {e.message}")
        return img
    finally:
        plugin.free_decoder(dec)


_driver = None


def drive_grid(plugin_ptr, tiles, max_threads):
    """the tiles of one grid through the plugin from C++ threads, the reference's way (tests/synth/plugin_driver.cpp:
    context.cc:2361-2401's window of max_threads async tasks); returns the heif_image* of every tile (caller releases)"""
    global _driver
    if _driver is None:
        _driver = C.CDLL(os.path.join(ROOT, "tests", "synth", "libhm_plugin_driver.so"))
        _driver.hm_test_drive_grid.restype = C.c_int
        _driver.hm_test_drive_grid.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    n = len(tiles)
    data = (C.c_char_p * n)(*tiles)
    size = (C.c_size_t * n)(*[len(t) for t in tiles])
    out = (C.c_void_p * n)()
    rc = _driver.hm_test_drive_grid(C.cast(plugin_ptr, C.c_void_p), data, size, n, max_threads, out)
    if rc:
        raise RuntimeError(f"plugin driver: heif_error code {rc}")
    return [C.c_void_p(v) for v in out]
