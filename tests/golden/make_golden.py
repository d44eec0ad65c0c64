#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle (oracle/oracle.cc).

The reference (CUDA-only) cannot run in this image and ships no fixtures, so these
vectors are produced by the oracle — itself pinned by the known-answer tests in
tests/test_oracle_known_answers.py and the rocRAND cross-check — and serve as
(1) a cross-machine determinism check of the oracle and (2) fixed expected outputs for
the HIP path.  Each file holds the inputs' description and the expected outputs:
rgb float32 (H,W,3), per-pixel ray counts uint32 (H,W), final RNG states uint32 (H*W,6).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import common  # noqa: E402

# name, H, W, spp, depth, post, extra kwargs
CASES = [
    ("cornell_box", 24, 32, 4, 10, True, {}),
    ("cornell_box", 16, 16, 2, 50, False, {}),
    ("spheres", 24, 32, 2, 8, True, {}),
    ("bunny", 24, 32, 2, 10, True, {"k_min": 64}),
    ("birthday", 24, 32, 4, 10, True, {}),
    ("mixed", 20, 28, 4, 10, True, {}),
    ("furnace", 16, 16, 4, 10, True, {}),
    ("sky_only", 16, 24, 2, 10, True, {}),
]


def case_file(name, h, w, spp, depth, post):
    return os.path.join(HERE, "%s_%dx%d_s%d_d%d_%s.npz" % (name, h, w, spp, depth, "post" if post else "raw"))


def main():
    for name, h, w, spp, depth, post, kw in CASES:
        rgb, rays, states, total, _ = common.oracle_render(name, h, w, spp, depth, post=post, **kw)
        np.savez_compressed(case_file(name, h, w, spp, depth, post), rgb=rgb, rays=rays, states=states,
                            total=np.uint64(total), seed=np.uint64(common.scene_seed(name)))
        print("%-12s %dx%d s%d d%d %s -> %d rays" % (name, h, w, spp, depth, "post" if post else "raw", total))


if __name__ == "__main__":
    main()
