#!/usr/bin/env python3
"""Long margin-check campaigns (GPU box): tests/margin_campaign.py with the round-3 world counts.
usage: RTMI_LIB_PATH=ray-tracing-cuda_amd/lib/librtmi_check1.so tools/gpu_check_margins.py [scenes|meshes|lists]"""
import os
import runpy
import sys
for k, v in (("FAR", "3000"), ("GRAZE", "1500"), ("NEEDLES", "1500"), ("SPHERES", "1500"), ("NEEDLE_LISTS", "1500")):
    os.environ.setdefault("RTMI_CHECK_" + k, v)
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "margin_campaign.py"),
               run_name="__main__")
