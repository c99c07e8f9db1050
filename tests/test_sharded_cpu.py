"""Host logic of the stream-sharded batch that needs no GPU: how a batch is cut into slices."""
import pytest


@pytest.mark.parametrize("e,shards", [(131072, 2), (65536, 4), (3000, 2), (2500, 3), (100, 2), (1, 8), (64, 2), (65, 2),
                                      (1048576, 7)])
def test_slices_cover_the_batch_once_and_start_aligned(e, shards):
    from multigrid_amd.sharded import ALIGN_ENVS, slice_bounds
    b = slice_bounds(e, shards)
    assert 1 <= len(b) <= shards
    assert b[0][0] == 0 and b[-1][1] == e
    for (lo, hi), nxt in zip(b, b[1:] + [(e, e)]):
        assert lo < hi and hi == nxt[0] and lo % ALIGN_ENVS == 0
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes[:-1] or sizes) == 0 and sizes[-1] <= sizes[0]


def test_bad_slice_requests():
    from multigrid_amd.sharded import slice_bounds
    for e, s in ((0, 2), (10, 0), (-5, 1)):
        with pytest.raises(ValueError):
            slice_bounds(e, s)
