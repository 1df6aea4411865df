#!/usr/bin/env python3
"""Development aid: at 2048 x 2048, are two runs of the resident CG bit-identical, and do the RECT and the general body agree?"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("srmeetsps-cuda_amd")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 5
iters = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [100]
sc = pkg.synth.make_scene(size, size, 4, n_img, seed=seed, mask_kind="full")
dh = pkg.DataHandler.from_scene(sc)
for max_iter in iters:
  print(f"--- {size}^2, {n_img} images, seed {seed}, cg_max_iter {max_iter}")
  out = {}
  for name, opts in (("rect_a", dict(cg_resident_rect=1)), ("rect_b", dict(cg_resident_rect=1)), ("gen_a", dict(cg_resident_rect=0)), ("gen_b", dict(cg_resident_rect=0)),
                     ("rect_2sync", dict(cg_resident_rect=1, cg_one_sync=0)), ("gen_2sync", dict(cg_resident_rect=0, cg_one_sync=0))):
      ctx = pkg.Context(device_id=0)
      ctx.set_option("cg_resident_tile", 512); ctx.set_option("cg_max_iter", max_iter)
      for k, v in opts.items():
          ctx.set_option(k, v)
      ctx.setup(dh)
      ctx.lighting(); ctx.albedo()
      e = ctx.depth()
      out[name] = (e, ctx.get("z"))
      ctx.close()
  def cmp(a, b):
      d = np.abs(out[a][1] - out[b][1])
      bad = np.flatnonzero(d > 0)
      where = ""
      if bad.size:
          j, i = np.divmod(bad, size)
          where = f" rows {i.min()}..{i.max()} cols {j.min()}..{j.max()}"
      print(f"{a} vs {b}: energies {out[a][0]} {out[b][0]}, differing pixels {bad.size}, max {d.max():.3e}{where}")
  cmp("rect_a", "rect_b"); cmp("gen_a", "gen_b"); cmp("rect_a", "gen_a"); cmp("rect_2sync", "gen_2sync")
