"""tri_blocked (csrc/ba_internal.h): the super-block enumeration of the lower triangle that the bulk trailing update walks.
It must visit every tile (ii >= jj) of an m x m grid exactly once for every m and every super-block size, and start with
(0,0), (1,0), (1,1) -- the tiles the hoisted diagonal kernels wait for.  The function is host + device code: compiled
here with g++ from the header's own text (CPU only)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "bundleadjustment.jl_amd", "csrc", "ba_internal.h")

MAIN = r"""
#include <cstdio>
#include <set>
#include <utility>
int main() {
  for (int sb : {2, 3, 4, 8, 12, 16})
    for (int m = 1; m <= 260; m++) {
      std::set<std::pair<int, int>> seen;
      const int n = m * (m + 1) / 2;
      for (int t = 0; t < n; t++) {
        int i = -1, j = -1;
        tri_blocked(t, m, &i, &j, sb);
        if (j > i || i >= m || j < 0) { std::printf("bad sb=%d m=%d t=%d -> %d %d\n", sb, m, t, i, j); return 1; }
        seen.insert({i, j});
      }
      if ((int)seen.size() != n) { std::printf("not a bijection sb=%d m=%d\n", sb, m); return 1; }
      if (m >= 2) {
        int i, j;
        tri_blocked(0, m, &i, &j, sb); if (i != 0 || j != 0) { std::printf("t=0 wrong\n"); return 1; }
        tri_blocked(1, m, &i, &j, sb); if (i != 1 || j != 0) { std::printf("t=1 wrong\n"); return 1; }
        tri_blocked(2, m, &i, &j, sb); if (i != 1 || j != 1) { std::printf("t=2 wrong\n"); return 1; }
      }
    }
  std::printf("ok\n");
  return 0;
}
"""


def test_tri_blocked_is_a_bijection(tmp_path):
    text = open(HDR).read()
    m = re.search(r"constexpr int TSB = \d+;\n__host__ __device__ inline void tri_blocked\(.*?\n}\n", text, re.S)
    assert m, "tri_blocked not found in ba_internal.h"
    src = tmp_path / "tri.cpp"
    src.write_text("#define __host__\n#define __device__\n" + m.group(0) + MAIN)
    exe = tmp_path / "tri"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
