"""rtmi_gather / rtmi_reduce_sum against a stub RCCL, on the CPU (the exchange has never run on two GPUs here: the
pool's boxes have one).  librtmi.so looks RCCL up at run time -- in the process first, so that the caller's copy (the
one its communicator came from, e.g. the copy PyTorch bundles) is the one used.  Each scenario runs in a process of
its own (the lookup happens once per process) that loads ONLY the stub and librtmi.so through ctypes.
What the reference does at this point: MPI_Reduce of host buffers, utils.cu:115-130."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "ray-tracing-cuda_amd", "lib", "librtmi.so")
STUB_SRC = os.path.join(ROOT, "tests", "stubs", "stub_rccl.c")


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("stub") / "libstub_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", out, STUB_SRC], check=True)
    return out


PRELUDE = """
import ctypes as C, os, sys
stub = C.CDLL(sys.argv[1], mode=C.RTLD_GLOBAL)   # "RCCL is already in the process"
L = C.CDLL(sys.argv[2])
L.rtmi_last_error.restype = C.c_char_p
class Frame(C.Structure):
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("post_process", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32)]
def frame(rank, world):
    return Frame(32, 48, 4, 10, 1, rank, world)
L.rtmi_gather.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
L.rtmi_reduce_sum.argtypes = [C.c_void_p, C.POINTER(Frame), C.c_void_p, C.c_int, C.c_void_p]
L.rtmi_frame_work_items.argtypes = [C.POINTER(Frame)]
L.rtmi_frame_work_items.restype = C.c_int64
stub.stub_last_count.restype = C.c_size_t
COMM, BUF = 0x1000, 0x100000   # never dereferenced: the stub only records
def no_second_rccl():
    maps = open("/proc/self/maps").read()
    assert "librccl" not in maps, "a real RCCL was loaded next to the one in the process"
"""


def run(stub, body, env=None):
    code = textwrap.dedent(PRELUDE) + textwrap.dedent(body)
    e = dict(os.environ)
    e.pop("LD_PRELOAD", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", code, stub, LIB], capture_output=True, text=True, env=e, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_frame_struct_matches_the_header():
    """The ctypes Frame of the prelude is rtmi_frame of include/rtmi.h, field for field."""
    import re
    h = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    body = h[h.index("typedef struct rtmi_frame"):h.index("} rtmi_frame;")]
    fields = re.findall(r"int32_t\s+(\w+);", body)
    assert fields == ["height", "width", "spp", "max_depth", "post_process", "rank", "world_size"], fields


def test_a_peer_sends_to_the_root_through_the_rccl_already_in_the_process(stub):
    run(stub, """
        f = frame(1, 4)
        n = L.rtmi_frame_work_items(C.byref(f)) * 3
        assert L.rtmi_gather(COMM, C.byref(f), BUF, None, 0, None) == 0, L.rtmi_last_error()
        assert [stub.stub_calls(i) for i in range(5)] == [1, 1, 1, 0, 0]
        assert stub.stub_last_peer() == 0 and stub.stub_last_count() == n and stub.stub_open_groups() == 0
        no_second_rccl()
    """)


def test_the_root_receives_from_every_other_rank(stub):
    run(stub, """
        f = frame(2, 4)   # root = rank 2; its own tiles already sit in their slot of d_all_tiles: no device copy
        n = L.rtmi_frame_work_items(C.byref(f)) * 3
        assert L.rtmi_gather(COMM, C.byref(f), BUF + 2 * n * 4, BUF, 2, None) == 0, L.rtmi_last_error()
        assert [stub.stub_calls(i) for i in range(5)] == [1, 1, 0, 3, 0] and stub.stub_open_groups() == 0
        no_second_rccl()
    """)


def test_a_failing_recv_still_closes_the_group(stub):
    run(stub, """
        f = frame(0, 4)
        n = L.rtmi_frame_work_items(C.byref(f)) * 3
        rc = L.rtmi_gather(COMM, C.byref(f), BUF, BUF, 0, None)
        assert rc != 0 and b"ncclRecv" in L.rtmi_last_error() and b"stub internal error" in L.rtmi_last_error()
        assert stub.stub_calls(0) == 1 and stub.stub_calls(1) == 1 and stub.stub_open_groups() == 0
        assert stub.stub_calls(3) == 2   # the second recv failed, the third was not issued
    """, env={"STUB_RCCL_FAIL_AT": "2"})


def test_a_failing_send_still_closes_the_group(stub):
    run(stub, """
        f = frame(3, 4)
        rc = L.rtmi_gather(COMM, C.byref(f), BUF, None, 0, None)
        assert rc != 0 and b"ncclSend" in L.rtmi_last_error()
        assert stub.stub_calls(1) == 1 and stub.stub_open_groups() == 0
    """, env={"STUB_RCCL_FAIL_AT": "1"})


def test_gather_argument_errors(stub):
    run(stub, """
        f = frame(1, 4)
        assert L.rtmi_gather(None, C.byref(f), BUF, None, 0, None) != 0 and b"communicator" in L.rtmi_last_error()
        assert L.rtmi_gather(COMM, C.byref(f), BUF, None, 4, None) != 0   # root outside the frame's ranks
        f0 = frame(0, 4)
        assert L.rtmi_gather(COMM, C.byref(f0), BUF, None, 0, None) != 0 and b"d_all_tiles" in L.rtmi_last_error()
        assert sum(stub.stub_calls(i) for i in range(5)) == 0
    """)


def test_reduce_sum_checks_its_root_against_the_communicator(stub):
    run(stub, """
        f = frame(0, 1)   # the reference's sample split: every rank renders the whole frame
        n = L.rtmi_frame_work_items(C.byref(f)) * 3
        assert L.rtmi_reduce_sum(COMM, C.byref(f), BUF, 3, None) == 0, L.rtmi_last_error()
        assert stub.stub_calls(4) == 1 and stub.stub_last_peer() == 3 and stub.stub_last_count() == n
        assert L.rtmi_reduce_sum(COMM, C.byref(f), BUF, 4, None) != 0 and b"not a rank" in L.rtmi_last_error()   # the stub's communicator has 4 ranks
        assert stub.stub_calls(4) == 1
        # no communicator = a single rank: a no-op for root 0, an error for any other root
        assert L.rtmi_reduce_sum(None, C.byref(f), BUF, 0, None) == 0
        assert L.rtmi_reduce_sum(None, C.byref(f), BUF, 1, None) != 0 and b"single rank" in L.rtmi_last_error()
        no_second_rccl()
    """)
