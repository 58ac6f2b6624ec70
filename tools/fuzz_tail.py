"""differential fuzz of the fused tail: corrupted-but-parsable 512x512 tiles (8-bit 4:2:0) as 2x2 grids, k_tail420 against
the separate kernels (k_deblock, k_sao_paste, k_ycbcr420_int) on the same batch"""
import importlib, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, corpus, hevcutil, synthutil
pkg = importlib.import_module("heif-decoder-lib_amd")
hm = pkg.lib()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
blobs = []
tries = 0
while len(blobs) < 64 and tries < 20000:
    tries += 1
    data = synthutil.picture(1200000 + tries % 8, vui=1, full_range=1, matrix=6, **corpus.TILE)
    b = bytearray(data)
    for _ in range(rng.randrange(1, 5)):
        b[rng.randrange(len(b) // 4, len(b))] ^= 1 << rng.randrange(8)
    try:
        blobs.append(hevcutil.parse(hm, bytes(b)))
    except RuntimeError:
        pass
print(len(blobs), "corrupted tiles parse after", tries, "tries", flush=True)
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
out = []
n_img = len(blobs) // 4
for group in (0, -1):
    gb = bench.GridBatch(pkg, dev, 2, 2, 512, 1000, 1010)
    for j in range(n_img):
        gb.add_image(blobs[4 * j:4 * j + 4])
    gb.finish(st, group)
    gb.batch.execute(3, st)
    torch.cuda.synchronize()
    assert gb.batch.tail_fused() == (group == 0)
    out.append([im["rgb"].cpu().numpy()[:1010, :3000].copy() for im in gb.images])
    gb.batch.close()
bad = sum(not np.array_equal(a, b) for a, b in zip(*out))
print("images that differ between the fused tail and the separate kernels:", bad, "of", n_img)
