#!/usr/bin/env python3
"""Dump the geometry of a synthetic scene (positions, indices) for scripts/sim/bvh_sim (CPU model of the tree traversal stage)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
S = importlib.import_module("mitsuba-im_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "atrium"
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/%s" % name
sc = getattr(S, name)(64, 36, 1)
np.asarray(sc.pos, np.float32).tofile(out + ".pos"); np.asarray(sc.idx, np.uint32).tofile(out + ".idx")
np.asarray(sc.cam_to_world, np.float32).tofile(out + ".cam")
print(name, len(sc.pos), "verts", len(sc.idx), "tris", "xfov", sc.xfov)
