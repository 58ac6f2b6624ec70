"""Generate the committed test fixtures.  Runs only in the build container (needs /root/reference
and the reference decoder build oracle/_ref):

  tests/data/*.hevc         coded pictures as [u32 BE length][NAL] records (what a libheif decoder
                            plugin receives through push_data) - DATA files taken from the
                            reference's own test material, re-framed, never source code
  tests/golden/decode.json  FNV-1a-64 fingerprints of the planes the REAL reference decoder
                            (libde265, oracle/_ref) produces for each fixture, per stage
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402

REF = "/root/reference"


def annexb_to_lp(data):
    out = bytearray()
    i = 0
    n = len(data)
    starts = []
    while i + 3 <= n:
        if data[i] == 0 and data[i + 1] == 0 and data[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    for k, s in enumerate(starts):
        e = (starts[k + 1] - 3) if k + 1 < len(starts) else n
        while e > s and data[e - 1] == 0:
            e -= 1
        nal = data[s:e]
        out += len(nal).to_bytes(4, "big") + nal
    return bytes(out)


def fingerprint(planes):
    h = 0
    for p in planes:
        a = p if p.max() > 255 else p.astype("uint8")
        buf = a.tobytes()
        h = orc.load().orc_fnv1a64(buf, len(buf), h)
    return f"{h:016x}"


def main():
    os.makedirs(os.path.join(ROOT, "tests", "data"), exist_ok=True)
    golden = {}
    srcs = {
        "basketball_1080p_qp32": f"{REF}/third-party/libde265/testfile/BasketballDrive_1920x1080_32.265",
        "basketball_1080p_qp25": f"{REF}/third-party/libde265/testfile/BasketballDrive_1920x1080_25.265",
        "basketball_1080p_qp1": f"{REF}/third-party/libde265/testfile/BasketballDrive_1920x1080_1.265",
    }
    for name, path in srcs.items():
        lp = annexb_to_lp(open(path, "rb").read())
        open(os.path.join(ROOT, "tests", "data", name + ".hevc"), "wb").write(lp)
        entry = {"source": os.path.relpath(path, REF), "bytes": len(lp)}
        for stage, flags in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO), ("deblock", orc.REF_F_NO_SAO), ("full", 0)):
            planes, info = orc.ref_decode(lp, flags)
            entry[stage] = fingerprint(planes)
            entry["width"], entry["height"] = int(planes[0].shape[1]), int(planes[0].shape[0])
            entry["info"] = info
        golden[name] = entry
        print(name, entry)
    json.dump(golden, open(os.path.join(ROOT, "tests", "golden", "decode.json"), "w"), indent=1, sort_keys=True)

    # synthetic corpus: blessed by the reference decoder (SIMD build = the oracle configuration, and
    # its scalar build must agree - the two differ for 8-bit SAO on 8-sample-wide chroma CTBs, which
    # the corpus therefore avoids)
    import corpus
    synth = {}
    for name in sorted(corpus.CASES):
        data = corpus.stream(name)
        entry = {"bytes": len(data), "stream_fnv": f"{orc.load().orc_fnv1a64(data, len(data), 0):016x}"}
        for stage, flags in (("recon", orc.REF_F_NO_DEBLOCK | orc.REF_F_NO_SAO), ("deblock", orc.REF_F_NO_SAO), ("full", 0)):
            planes, info = orc.ref_decode(data, flags)
            if name not in corpus.SIMD_BUILD_ONLY:
                scalar, _ = orc.ref_decode(data, flags | orc.REF_F_SCALAR)
                assert all((a == b).all() for a, b in zip(planes, scalar)), f"{name}: reference SIMD != scalar"
            entry[stage] = fingerprint(planes)
        entry["width"], entry["height"] = int(planes[0].shape[1]), int(planes[0].shape[0])
        entry["info"] = info
        synth[name] = entry
        print(name, entry["bytes"], entry["full"])
    json.dump(synth, open(os.path.join(ROOT, "tests", "golden", "synth.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
