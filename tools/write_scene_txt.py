"""Writes the default world (the 487 spheres of the compiled reference) as the scene.txt the C++ examples read.
usage: python tools/write_scene_txt.py scene.txt"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import scene as S
with open(sys.argv[1], "w") as f:
    for o in S.rtiow_records():
        f.write(" ".join(repr(float(x)) for x in o) + "\n")
