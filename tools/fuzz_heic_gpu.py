"""mutated HEIC files through the whole image path on the GPU (hm_file_open + hm_decode_item): every call must return -
an error or an image - and the process must stay healthy (a known-good decode is repeated after every 25 files)"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import heifwriter, pipeline, synthutil
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
HERE = os.path.join(ROOT, "tests", "data")
seeds = [open(os.path.join(HERE, n), "rb").read() for n in ("colors-no-alpha.heic", "colors-with-alpha.heic", "example.heic")]
tiles = [synthutil.picture(8100 + t, width=64, height=64) for t in range(6)]
seeds.append(heifwriter.write_heic(tiles, (64, 64), grid=(2, 3, 180, 120)))
seeds.append(heifwriter.write_heic([synthutil.picture(8200, width=128, height=72, bit_depth=10, chroma_format=2)], (128, 72), chroma_format=2, bit_depth=10))
good = seeds[3]
f = pipeline.HeifFile(hm, good); want, _ = f.decode(f.primary(), 10, threads=4); f.close()
ok = err = 0
for it in range(600):
    b = bytearray(seeds[it % len(seeds)])
    for _ in range(rng.randrange(1, 6)):
        i = rng.randrange(len(b))
        m = rng.randrange(4)
        if m == 0: b[i] ^= 1 << rng.randrange(8)
        elif m == 1: b[i] = rng.randrange(256)
        elif m == 2: b[i:i + 4] = rng.randrange(1 << 32).to_bytes(4, "big")
        else: del b[i:i + rng.randrange(1, 9)]
    try:
        f = pipeline.HeifFile(hm, bytes(b))
    except RuntimeError:
        err += 1
        continue
    try:
        f.decode(f.primary(), rng.choice((10, 11, 0, 14)), threads=rng.choice((1, 4)), copy=False)
        ok += 1
    except RuntimeError:
        err += 1
    f.close()
    if it % 25 == 0:
        g = pipeline.HeifFile(hm, good); got, _ = g.decode(g.primary(), 10, threads=4); g.close()
        assert np.array_equal(got[0][:120, :180 * 3], want[0][:120, :180 * 3]), "a good file decodes differently after a damaged one"  # (row padding is not part of the image)
print("decoded", ok, "refused", err)
