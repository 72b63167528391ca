"""What runs on the GPU box is the tree's own library, and the ISA it was linked from passed the ring lint."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.gpu
def test_the_library_under_test_is_current_and_linted(gpu_available):
    """The hot loop's correctness rests on epik_amd/csrc/lint_ring_asm.py having seen the ISA of THIS build (the
    streaming ring's loads are invisible to hipcc): the Makefile links only linted objects and records it; a library
    built some other way, or older than the kernel sources beside it, fails here instead of passing 900 tests for the
    wrong binary."""
    assert gpu_available
    from epik_amd import capi, provenance
    capi.load()
    s = provenance.check_library(strict_sources=True)
    assert s["lint_covers_this_build"] and s["library_is_current"] and s["hipcc"]
    # the library mapped into this process is the one the record sits beside
    with open("/proc/self/maps") as fh:
        assert "epik_amd/libepik_amd.so" in fh.read()
