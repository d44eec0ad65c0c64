"""The culls' margins, checked on every query.

Three culls of the HIP path are exact only because padded bounds and distance slacks cover the reach of the binary32
triangle test's false accepts (utils.cu:49-85) and of the binary32 sphere pre-test: the culled world-list scan
(hitable_list.cu:7-25), the grouped sphere scan (sphere.cu:11-44) and the mesh search whose finds the replay of
bvh.cuh:123-158 / bvh.cu:6-30 then filters.  `librtmi_check1.so` -- the same sources built with -DRTMI_CHECK_MARGINS
-DRTMI_CHECK_EVERY=1 by __graft_entry__.build() -- answers EVERY query a second time without any of them (meshes: by the
reference's own walk of its own tree) and counts the disagreements.  It is a diagnostic build, so the campaign
(tests/margin_campaign.py) runs in a process of its own with RTMI_LIB_PATH pointing at it; the oracle is not involved.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK_LIB = os.path.join(ROOT, "ray-tracing-cuda_amd", "lib", "librtmi_check1.so")


@pytest.mark.gpu
def test_every_query_agrees_with_the_uncullled_answer():
    assert os.path.exists(CHECK_LIB), "librtmi_check1.so missing: run __graft_entry__.build() (make -C csrc check1)"
    env = dict(os.environ, RTMI_LIB_PATH=CHECK_LIB)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "margin_campaign.py")], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode in (0, 1), r.stderr[-4000:]
    out = json.loads(r.stdout[r.stdout.index("{"):])
    # every group of worlds re-did rays, and none of them disagreed
    want = {"cornell_box_256x256x64_d50", "spheres_256x256x16_d8", "birthday_128x128x16_d10", "quilt_100_0", "quilt_300_5",
            "quilt_100_from_1000", "quilt_100_from_10000", "spheres_200_from_1000", "spheres_200_from_10000",
            "bunny_128x128x8_d10", "far_views_without_slivers", "far_views_with_slivers", "far_views_known_sliver_cases",
            "grazing_views", "needles", "far_sphere_clouds", "needle_lists"}
    assert want <= set(out), sorted(want - set(out))
    for tag, v in out.items():
        assert v["re_done"] > 0, (tag, v)
        assert v["disagreements"] == 0, (tag, v)
    worlds = sum(v.get("worlds", 1) for v in out.values())
    assert worlds >= 200, worlds
    # every ray of the plain scenes was re-done (RTMI_CHECK_EVERY=1)
    for tag in ("cornell_box_256x256x64_d50", "bunny_128x128x8_d10"):
        assert out[tag]["re_done"] == out[tag]["rays"], (tag, out[tag])
