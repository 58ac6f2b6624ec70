"""The caching allocator (csrc/devpool.cpp) keeps one pool per GPU: a block freed by work on device 0 must never be
handed to an allocation made with device 1 current (pipelines on several GPUs in one process).  Built here with host
stand-ins for the HIP calls (HM_POOL_HOST_STUB), so the logic runs without a GPU."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "heif-decoder-lib_amd", "csrc")


@pytest.fixture(scope="module")
def pool(tmp_path_factory):
    out = tmp_path_factory.mktemp("devpool") / "libdevpool_stub.so"
    stub = tmp_path_factory.mktemp("devpool_src") / "stub.cpp"
    stub.write_text('#include "hm_internal.h"\nextern "C" int hm_check_hip(hipError_t e, const char*) { return e == hipSuccess ? 0 : -5; }\n'
                    'extern "C" int hm_fail(int s, const char*, ...) { return s; }\n')
    cmd = ["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-DHM_POOL_HOST_STUB", "-D__HIP_PLATFORM_AMD__", f"-I{CSRC}", f"-I{ROOT}/include", "-I/opt/rocm/include",
           os.path.join(CSRC, "devpool.cpp"), str(stub), "-o", str(out)]
    subprocess.run(cmd, check=True)
    L = C.CDLL(str(out))
    L.hm_pool_device_alloc.restype = C.c_void_p
    L.hm_pool_device_alloc.argtypes = [C.c_size_t]
    L.hm_pool_device_free.argtypes = [C.c_void_p]
    L.hm_pool_device_cached.restype = C.c_size_t
    L.hm_pool_device_cached.argtypes = [C.c_int]
    return L


def test_blocks_stay_with_their_device(pool):
    pool.hm_pool_stub_set_device(0)
    a = pool.hm_pool_device_alloc(1 << 20)
    assert a
    pool.hm_pool_stub_set_device(1)  # another device is current when the block is released ...
    pool.hm_pool_device_free(a)
    assert pool.hm_pool_device_cached(0) == 1 << 20 and pool.hm_pool_device_cached(1) == 0  # ... it still goes home
    b = pool.hm_pool_device_alloc(1 << 20)  # device 1 must get a block of its own
    assert b and b != a
    pool.hm_pool_stub_set_device(0)
    c = pool.hm_pool_device_alloc(1 << 20)  # device 0 gets its cached block back
    assert c == a and pool.hm_pool_device_cached(0) == 0
    pool.hm_pool_device_free(b)
    pool.hm_pool_device_free(c)
    assert pool.hm_pool_device_cached(0) == 1 << 20 and pool.hm_pool_device_cached(1) == 1 << 20


def test_buckets_recycle(pool):
    pool.hm_pool_stub_set_device(2)
    a = pool.hm_pool_device_alloc(100_000)  # bucket: 128 KiB subdivided in eighths -> 106496
    pool.hm_pool_device_free(a)
    assert pool.hm_pool_device_cached(2) == 106496
    b = pool.hm_pool_device_alloc(105_000)  # same bucket: recycled
    assert b == a
    pool.hm_pool_device_free(b)


def test_device_slab_plan_of_a_grid(hm):
    """hm_plan_device_slabs: the cut hm_decode_item_devices makes - contiguous slabs of tile rows, one per listed device,
    sizes at most one row apart, devices beyond the row count get nothing; the same cut as shard.row_slabs (N > 1 harness)."""
    import __graft_entry__ as g
    sh = g.load_package().shard
    hm.hm_plan_device_slabs.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    for rows in (0, 1, 2, 5, 6, 32, 33):
        for n in (1, 2, 3, 8):
            first = (C.c_int32 * n)()
            count = (C.c_int32 * n)()
            assert hm.hm_plan_device_slabs(rows, n, first, count) == 0
            got = [(first[d], count[d]) for d in range(n)]
            assert got == sh.row_slabs(rows, n)
            assert sum(c for _, c in got) == rows and max(c for _, c in got) - min(c for _, c in got) <= 1
            covered = [r for f, c in got for r in range(f, f + c)]
            assert covered == list(range(rows))
    assert hm.hm_plan_device_slabs(4, 0, None, None) != 0
