#!/usr/bin/env python3
"""Host-side scene build time and a hash of the resulting BVH2 / wide BVH / triangle order: HIPR_BVH_THREADS=1 and =N must print the same hash.
usage: HIPR_BVH_TIMING=1 HIPR_BVH_THREADS=N python tools/bvh_build_probe.py <atrium triangle target>"""
import sys, time, hashlib, ctypes as C
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from bifrost3d_amd.host import Scene
n = int(sys.argv[1])
Scene("cornell")     # loads the library outside the timing
t = time.time(); s = Scene("atrium", param0=n, param1=1); dt = time.time() - t
d = s.desc
h = hashlib.sha256()
h.update(C.string_at(d.nodes, d.node_count * 64))
h.update(C.string_at(d.wide_nodes, d.wide_node_count * 64))
h.update(C.string_at(d.triangles, d.triangle_count * 48))
h.update(C.string_at(d.wide8_slots, d.wide8_slot_count * 64))
print(d.triangle_count, d.node_count, d.wide_node_count, d.wide8_slot_count, d.wide8_height, f"{dt:.2f}s", h.hexdigest()[:16])
