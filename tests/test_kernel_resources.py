"""No kernel of libbsmi.so written in this repository may carry a scratch (private) segment.

Round 4: two forward passes overlapping on one GPU gave corrupted predictions, and the one launch of a pass whose output was wrong
was the head kernel -- the only one with a scratch segment (a private array indexed at run time, 272 bytes per lane).  Kept in
registers it is clean under any overlap.  The conv kernels also count their outstanding loads with
s_waitcnt vmcnt(N), which spill traffic would break.  This test reads the code objects' metadata; it needs no GPU."""
import os, sys
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from kernel_resources import kernels  # noqa: E402

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bootstrapper_amd", "libbsmi.so")


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(LIB):
        pytest.fail("bootstrapper_amd/libbsmi.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    return kernels(LIB)


def test_code_objects_are_read(table):
    names = [k["name"] for k in table]
    assert len(table) > 500
    for needle in ("head_kernel", "conv_igemm_kernel", "wino4_in_kernel", "blosc_plane_kernel", "wino_pack_kernel"):
        assert any(needle in n for n in names), needle


def test_own_kernels_have_no_scratch_segment(table):
    own = [k for k in table if "rocprim" not in k["name"] and "hipcub" not in k["name"]]
    assert len(own) > 200
    bad = [(k["name"], k["scratch"]) for k in own if k["scratch"] or k["dynamic_stack"]]
    assert not bad, f"kernels with a scratch segment: {bad[:5]}"


def test_scratch_is_confined_to_library_kernels(table):
    """what is left are rocPRIM's sort / reduce-by-key kernels of the segmentation stage (48-80 bytes per lane)"""
    rest = [k for k in table if k["scratch"]]
    assert all("rocprim" in k["name"] for k in rest), [k["name"][:80] for k in rest if "rocprim" not in k["name"]]
    assert max([k["scratch"] for k in rest], default=0) <= 128
