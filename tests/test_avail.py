"""hm_avail.h (neighbour availability of a block from its position + the CTB's four neighbour bits: what k_residual runs
per record since format HSM5) against a plain restatement with z-scan codes, exhaustively over CTB sizes, chroma formats,
block positions / sizes, neighbour bits and picture-edge distances.  Compiled here as host code."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include "hm_avail.h"
static unsigned z(int x4, int y4) { unsigned r = 0; for (int b = 0; b < 4; b++) r |= ((x4 >> b) & 1u) << (2 * b) | ((y4 >> b) & 1u) << (2 * b + 1); return r; }
int main()
{
  long n = 0;
  for (int l2c = 4; l2c <= 6; l2c++)
    for (int fmt = 0; fmt < 3; fmt++) { // luma, chroma 4:2:0, chroma 4:2:2
      const int lw = fmt ? 1 : 0, lh = fmt == 1 ? 1 : 0, cs = 1 << l2c;
      for (int l2 = 2; l2 <= 5; l2++) {
        const int nT = 1 << l2, wL = nT << lw, hL = nT << lh;
        if (wL > cs || hL > cs) continue;
        for (int yi = 0; yi + hL <= cs; yi += hL)
          for (int xi = 0; xi + wL <= cs; xi += wL)
            for (unsigned nb = 0; nb < 16; nb++)
              for (int rx = -4; rx <= nT + 4; rx += 4)
                for (int ry = -4; ry <= nT + 4; ry += 4) {
                  const hm_avail a = hm_derive_avail(xi, yi, wL, hL, nT, rx, ry, l2c, nb);
                  // restatement: a neighbour unit inside the CTB is available iff its z-scan code is smaller
                  const unsigned zc = z(xi >> 2, yi >> 2);
                  const unsigned left = xi ? 1u : (nb >> 3) & 1u, top = yi ? 1u : (nb >> 1) & 1u;
                  const unsigned tl = xi && yi ? 1u : (xi ? top : (yi ? left : nb & 1u));
                  unsigned bl, tr;
                  if (yi + hL >= cs) bl = 0; else if (xi == 0) bl = left; else bl = z((xi - 1) >> 2, (yi + hL) >> 2) < zc;
                  if (yi == 0) tr = xi + wL >= cs ? (nb >> 2) & 1u : (nb >> 1) & 1u; else if (xi + wL >= cs) tr = 0; else tr = z((xi + wL) >> 2, (yi - 1) >> 2) < zc;
                  const int n_bl = (bl && left && ry > 0) ? (nT < ry ? nT : ry) : 0, n_tr = (tr && rx > 0) ? (nT < rx ? nT : rx) : 0;
                  if (a.left != left || a.top != top || a.tl != tl || a.n_bl != n_bl || a.n_tr != n_tr) {
                    std::printf("MISMATCH ctb %d fmt %d nT %d at (%d,%d) nb %u rooms %d %d: %u %u %u %d %d vs %u %u %u %d %d\n", cs, fmt, nT, xi, yi, nb, rx, ry,
                                a.left, a.top, a.tl, a.n_bl, a.n_tr, left, top, tl, n_bl, n_tr);
                    return 1;
                  }
                  n++;
                }
      }
    }
  std::printf("OK %ld\n", n);
  return 0;
}
"""


def test_closed_form_equals_z_scan_codes(tmp_path):
    src = tmp_path / "avail_check.cpp"
    src.write_text(SRC)
    exe = tmp_path / "avail_check"
    subprocess.run(["g++", "-std=c++17", "-O1", f"-I{ROOT}/heif-decoder-lib_amd/csrc", f"-I{ROOT}/include", str(src), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
