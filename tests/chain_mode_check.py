"""Helper of test_chain_modes_gpu.py (run as a script, a process per set of tuning / fault-injection knobs, which the
script takes from its environment and sets through the library's test hook): reconstruct a few corpus pictures on the GPU and compare with the oracle.
Prints OK, or the first difference; exit status 0 / 1; 3 = hm_batch_check reported a wave that gave up."""
import sys

import numpy as np

import __graft_entry__ as g
import corpus
import gpudecode
import orc


def main():
    import ctypes
    import os
    pkg = g.load_package(test_knobs=True)
    # the cuts and the fault-injection knobs are library test hooks (hm_debug_set), not environment variables of the product: the
    # test hands them to this script through the environment and load_package(test_knobs=True) sets them (tests/knobs.py)
    hm = pkg.lib()
    names = sys.argv[1:] or ["tile512_a", "ctb64_wpp", "hi422_10", "mono8", "ragged"]
    copies = int(os.environ.get("HM_CHECK_COPIES", "3"))  # (hundreds: the cuts the launcher chooses for mid-size batches)
    for name in names:
        if name == "wide16k":  # the widest picture class: CTB 64, 16-bit storage, 4:2:2, 16384 columns, two CTU rows
            import synthutil
            blobs = [pkg.capi.parse_hevc(synthutil.picture(515151, width=16384, height=128, log2_ctb=6, bit_depth=10, chroma_format=2, qp=32, density=30))] * copies
        elif name == "big422":  # BASELINE config 4's picture: 2048x1536 10-bit 4:2:2, 48 rows of 64 CTUs - a long wavefront
            import synthutil
            blobs = [pkg.capi.parse_hevc(synthutil.picture(4220010, width=2048, height=1536, chroma_format=2, bit_depth=10, log2_ctb=5, qp=30, vui=1,
                                                           full_range=0, matrix=9, primaries=9))] * copies
        elif name == "mono10_wide":  # 16-bit monochrome, CTBs of 32, 512 columns: luma of four rows per wave = 15 KB of LDS
            import synthutil
            blobs = [pkg.capi.parse_hevc(synthutil.picture(4001032, width=512, height=192, chroma_format=0, bit_depth=10, log2_ctb=5, qp=30))] * copies
        elif name == "mixed":  # pictures of one class and different sizes in one launch (the cut follows the tallest; short ones leave waves idle)
            blobs = [pkg.capi.parse_hevc(corpus.stream(n)) for n in ("tile512_a", "ragged", "dense_lowqp", "no_deblock", "tile512_b", "ragged")] * 2
        else:
            blobs = [pkg.capi.parse_hevc(corpus.stream(name))] * copies
        try:
            got = gpudecode.decode_pictures(pkg, blobs, 3)
        except RuntimeError as e:
            print("CHECK FAILED:", e)
            return 3
        expected = {}
        for i, pic in enumerate(got):
            if id(blobs[i]) not in expected:
                expected[id(blobs[i])] = orc.oracle_decode(blobs[i], 3, crop=True)[0]
            exp = expected[id(blobs[i])]
            for c in range(len(exp)):
                if not np.array_equal(pic[c], exp[c]):
                    print(f"{name}: picture {i} plane {c} differs")
                    return 1
    print("OK")
    return 0


if __name__ == "__main__":
    sys.exit(main())
