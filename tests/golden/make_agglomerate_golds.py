"""Writes tests/golden/reference_agglomerate_ids.json: the literal one-rank arrays of the reference's
tests/test_agglomerate.cc (simple_agglomerate_2d / _3d: 8 x 8 and 8 x 8 x 8 cells, block agglomerates of 2 x 3 (x 4) cells).
Data only (numbers); run where /root/reference exists:  python tests/golden/make_agglomerate_golds.py"""
import json
import os
import re

src = open("/root/reference/tests/test_agglomerate.cc").read()


def grab(case):
    i = src.index("BOOST_AUTO_TEST_CASE(%s)" % case)
    j = src.index("world_size == 1", i)
    k = src.index("ref_agglomerates =", j)
    e = src.index("};", k)
    return [int(v) for v in re.findall(r"\d+", src[k:e])]


out = {"source": "tests/test_agglomerate.cc:69-230 (world_size == 1), agglomeration nx 2 ny 3 nz 4 (:49-51), 3 global refinements",
       "agglomerate_2d": grab("simple_agglomerate_2d"), "agglomerate_3d": grab("simple_agglomerate_3d")}
assert len(out["agglomerate_2d"]) == 64 and len(out["agglomerate_3d"]) == 512
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_agglomerate_ids.json"), "w") as f:
    json.dump(out, f)
