"""A/B helper (round 5): config3 share with and without the big one-workgroup voxel form.  python tools/r05_share_ab.py <tag>"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import bench_worlds

which = tuple(sys.argv[2].split(",")) if len(sys.argv) > 2 else ("config3_share",)
d = bench_worlds.measure_shares(which=which)
print(json.dumps(d))
