import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reid_gan_amd import ops
from reid_gan_amd._lib import lib
import numpy as np
rows = []
orig = ops.WgradJobs.add
def add(self, g, p, kh, kw, sh, sw, ph, pw, slab_ptr, splitk):
    n0 = sum(len(v[0]) for v in self.groups.values())
    orig(self, g, p, kh, kw, sh, sw, ph, pw, slab_ptr, splitk)
    for key, (recs, counts) in self.groups.items():
        pass
    # find the job just added
    tile, nb = np.zeros(2, dtype=np.int32), np.zeros(1, dtype=np.int32)
    rec = np.zeros(lib.ried_conv2d_wgrad_job_bytes(), dtype=np.uint8)
    lib.ried_conv2d_wgrad_make_job(rec.ctypes.data, g.ptr, g.ld, g.C, g.N, g.H, g.W, p.ptr, p.ld, p.C, p.H, p.W,
                                   kh, kw, sh, sw, ph, pw, slab_ptr, splitk, tile.ctypes.data, nb.ctypes.data)
    rows.append((tuple(tile), int(nb[0]), splitk, g.C, p.C, kh, kw, g.N * g.H * g.W, p.npix, 2.0 * p.npix * p.C * g.C * kh * kw / 1e9))
ops.WgradJobs.add = add
import bench
from reid_gan_amd.engine import Generator, State
g, d, vgg = bench.synth_states()
G = Generator(State(g, "cuda"), 128, 64, 64)
rows.sort()
tot = {}
for r in rows:
    tot.setdefault(r[0], [0, 0, 0.0]); tot[r[0]][0] += 1; tot[r[0]][1] += r[1]; tot[r[0]][2] += r[-1]
print(tot)
for r in sorted(rows, key=lambda r: -r[-1])[:40]:
    print(r)
