"""GPU (-m gpu): nova_hip_trace_rays_noalloc — the one entry point that makes no allocating call (its queue
heads and stack-spill area are a static device array) — against nova_hip_trace_rays, through the C ABI.
Bar: every hit record bit-identical (closest-hit rays; for any-hit rays the occlusion answer, as everywhere:
which blocker is found first depends on the order waves take rays in).  Two calls in a row, the second with
a smaller input, because every call shares the array and it keeps the last call's contents."""
import numpy as np
import pytest

from nova_amd import capi, scene as S
from tests.conftest import random_rays

pytestmark = pytest.mark.gpu


def _same(got, ref, df):
    anyhit = (df[:, 3].view(np.uint32) & 1) == 1
    assert (got[~anyhit] == ref[~anyhit]).all()
    assert ((got[anyhit, 3].view(np.int32) >= 0) == (ref[anyhit, 3].view(np.int32) >= 0)).all()


def test_no_alloc_entry_point_equals_trace_rays_twice_in_a_row():
    capi.init(0)
    sc = S.dragon_standin()  # 871 k triangles: the tree needs more stack than the 16 LDS entries
    tb, bb = capi.Buffer(sc.triangles), capi.Buffer(sc.bvh)
    lib = capi.load()
    n1, n2 = 300_000, 4_097
    pt, df = random_rays(n1, seed=77, anyhit_every=5, tmax_every=3)
    ro, rd = capi.Buffer(pt), capi.Buffer(df)
    h_ref, h_new = capi.Buffer(nbytes=16 * n1), capi.Buffer(nbytes=16 * n1)
    # nothing cached for these buffers yet: the entry point refuses rather than pack (packing allocates)
    assert lib.nova_hip_trace_rays_noalloc(tb.ptr, bb.ptr, ro.ptr, rd.ptr, h_new.ptr, n1) == capi.E_NOT_CACHED
    capi.trace_rays(tb, bb, ro, rd, h_ref, n1)
    ref1 = h_ref.read(np.uint32).reshape(n1, 4).copy()
    assert (ref1[:, 3].view(np.int32) >= 0).mean() > 0.2
    h_new.write(np.full((n1, 4), 0xDEADBEEF, np.uint32))
    calls = capi.get_option(capi.OPT_ALLOC_CALLS)
    assert calls > 0  # the counter works: the calls above allocated
    capi.trace_rays_noalloc(tb, bb, ro, rd, h_new, n1)
    assert capi.get_option(capi.OPT_ALLOC_CALLS) == calls  # no hipMalloc / hipFree / event / stream made by the call
    _same(h_new.read(np.uint32).reshape(n1, 4), ref1, df)
    # second call, smaller input (other rays): stale queue heads or spill entries of the first would show
    pt2, df2 = random_rays(n2, seed=78, anyhit_every=4, tmax_every=2)
    ro2, rd2 = capi.Buffer(pt2), capi.Buffer(df2)
    capi.trace_rays(tb, bb, ro2, rd2, h_ref, n2)
    ref2 = h_ref.read(np.uint32).reshape(n1, 4)[:n2].copy()
    h_new.write(np.full((n1, 4), 0xDEADBEEF, np.uint32))
    calls = capi.get_option(capi.OPT_ALLOC_CALLS)
    capi.trace_rays_noalloc(tb, bb, ro2, rd2, h_new, n2)
    assert capi.get_option(capi.OPT_ALLOC_CALLS) == calls
    got2 = h_new.read(np.uint32).reshape(n1, 4)
    _same(got2[:n2], ref2, df2)
    assert (got2[n2:] == 0xDEADBEEF).all()  # nothing written past the batch
    # the validation kernels share the array too (none of them may grow the spill buffer)
    for kernel in (6, 11, 12):
        capi.set_option(capi.OPT_TRAVERSE_KERNEL, kernel)
        h_new.write(np.full((n1, 4), 0xDEADBEEF, np.uint32))
        calls = capi.get_option(capi.OPT_ALLOC_CALLS)
        capi.trace_rays_noalloc(tb, bb, ro2, rd2, h_new, n2)
        assert capi.get_option(capi.OPT_ALLOC_CALLS) == calls
        _same(h_new.read(np.uint32).reshape(n1, 4)[:n2], ref2, df2)
    capi.set_option(capi.OPT_TRAVERSE_KERNEL, 0)
    # argument errors come back as codes
    assert lib.nova_hip_trace_rays_noalloc(tb.ptr, bb.ptr, ro.ptr, rd.ptr, h_new.ptr, 1 << 31) == capi.E_TOO_LARGE
    assert lib.nova_hip_trace_rays_noalloc(tb.ptr, bb.ptr, 0x1000, rd.ptr, h_new.ptr, 16) == capi.E_UNKNOWN_BUFFER
    # rewriting a scene buffer drops the cached layout: refused again until an allocating call has seen it
    bb.write(sc.bvh)
    assert lib.nova_hip_trace_rays_noalloc(tb.ptr, bb.ptr, ro.ptr, rd.ptr, h_new.ptr, n1) == capi.E_NOT_CACHED
