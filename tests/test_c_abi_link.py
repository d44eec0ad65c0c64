"""include/rtmi.h is a C header: compile a C99 program against it with gcc, link librtmi.so and
run the host-only entry points (no GPU involved).  Also guards against reference source text
having been copied into the repository."""
import difflib
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ray-tracing-cuda_amd", "lib")

C_PROG = r'''
#include <stdio.h>
#include <string.h>
#include "rtmi.h"
static void twice(const float in[3], float out[3], void *user) { (void)user; out[0] = 2 * in[0]; out[1] = 2 * in[1]; out[2] = 2 * in[2]; }
int main(void) {
  if (rtmi_version() != RTMI_VERSION) return 1;
  rtmi_scene *s = rtmi_scene_create();
  float white[3] = {1, 1, 1}, c[3] = {0, 0, -1}, up[3] = {0, 1, 0}, eye[3] = {0, 0, 1}, len[3] = {1, 2, 3};
  int m = rtmi_lambertian(s, white);
  if (m < 0 || rtmi_add_sphere(s, c, 0.5, m) != RTMI_OK || rtmi_add_sky(s) != RTMI_OK) return 2;
  if (rtmi_add_parallelepiped_lengths(s, len, m, twice, NULL) != RTMI_OK) return 3;
  if (rtmi_add_sphere(s, c, 0.5, 99) != RTMI_ERR_INVALID || !strlen(rtmi_last_error())) return 4;
  if (rtmi_camera_pinhole(s, eye, c, up, 1.0, 1.5) != RTMI_OK) return 5;
  int64_t st[8];
  if (rtmi_scene_stats(s, st) != RTMI_OK || st[0] != 3 || st[1] != 1 || st[2] != 6) return 6;
  rtmi_frame f = {20, 30, 4, 10, 1, 0, 1};
  if (rtmi_frame_work_items(&f) != 3 * 4 * 64 || rtmi_frame_pixel_of(&f, 0) != 0) return 7;
  if (rtmi_get_workload(1, 3, 100) != 33 || rtmi_get_workload(0, 3, 100) != 34) return 8;
  uint32_t st6[RTMI_STATE_WORDS];
  if (rtmi_rng_host_state(1024, 5, st6) != RTMI_OK) return 9;
  float x = rtmi_rng_host_random_float(0.f, 1.f, st6);
  if (!(x > 0.f && x <= 1.f)) return 10;
  rtmi_scene_destroy(s);
  printf("c-abi ok, devices=%d\n", rtmi_device_count());
  return 0;
}
'''


def test_header_is_c99_and_library_links_from_c(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text(C_PROG)
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-L", LIB, "-lrtmi", "-Wl,-rpath," + LIB, "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "c-abi ok" in r.stdout


def _shingles(text, k=6):
    import re
    toks = re.findall(r"[A-Za-z_][A-Za-z_0-9]*|\d+|\S", text)
    return {tuple(toks[i:i + k]) for i in range(max(0, len(toks) - k + 1))}


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_no_reference_source_text_in_the_repository():
    """Nothing in the repo may be a (renamed) copy of a reference source file: the share of a
    file's own 6-token shingles that also occur in one reference file must stay small.  (The
    oracle restates the reference statement by statement, by design; it is one large file and no
    reference file accounts for more than a few percent of it.)"""
    ref = {}
    for dp, _, fs in os.walk("/root/reference"):
        if ".git" in dp:
            continue
        for f in fs:
            if f.endswith((".cu", ".cuh", ".h", ".cc")):
                sh = _shingles(open(os.path.join(dp, f), errors="ignore").read())
                if len(sh) > 30:
                    ref[os.path.join(dp, f)] = sh
    worst = (0.0, None, None)
    for dp, dn, fs in os.walk(ROOT):
        dn[:] = [d for d in dn if d not in (".git", "gpurun_out", "build", "lib", "_build", "__pycache__", ".pytest_cache")]
        for f in fs:
            if not f.endswith((".cuh", ".h", ".hip", ".cc", ".cpp", ".hpp", ".cu", ".py")):
                continue
            mine = _shingles(open(os.path.join(dp, f), errors="ignore").read())
            if len(mine) <= 30:
                continue
            for rp, rs in ref.items():
                common_share = len(mine & rs) / len(mine)
                if common_share > worst[0]:
                    worst = (common_share, os.path.join(dp, f), rp)
    assert worst[0] < 0.5, worst
