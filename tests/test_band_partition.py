"""Host-side index arithmetic of the partitioned band solver (tsba_bandp.h / tsba_bandcr.h) through the library's debug hooks: no GPU
needed.  The partition table every workgroup derives on the device, and the block pool of the cyclic-reduction separator solver."""
import ctypes as C
import os
import pytest

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "textslam_amd", "libtsba.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    L = C.CDLL(LIB)
    L.tsba_debug_bandp_part.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]; L.tsba_debug_bandp_part.restype = None
    L.tsba_debug_cr_blk_index.argtypes = [C.c_int, C.c_int, C.c_int]; L.tsba_debug_cr_blk_index.restype = C.c_longlong
    L.tsba_debug_cr_pool_blocks.argtypes = [C.c_int]; L.tsba_debug_cr_pool_blocks.restype = C.c_longlong
    return L


def _part(lib, nb, B, Pmax, p):
    o = (C.c_int*5)(); lib.tsba_debug_bandp_part(nb, B, Pmax, p, o); return list(o)


@pytest.mark.parametrize("nb,B,Pmax", [(4998, 10, 64), (998, 12, 11), (298, 8, 8), (100, 10, 21), (25, 10, 4), (7, 1, 3), (598, 13, 12)])
def test_partition_covers_the_band(lib, nb, B, Pmax):
    """Interiors [a, b) and separators of B blocks tile [0, nb); every interior holds >= 2 B + 2 blocks unless there is only one."""
    P = _part(lib, nb, B, Pmax, 0)[0]
    assert 1 <= P <= Pmax
    pos = 0
    for p in range(P):
        Pp, a, b, hl, hr = _part(lib, nb, B, Pmax, p)
        assert Pp == P and a == pos and b > a
        assert hl == (p > 0) and hr == (p < P - 1)
        if P > 1:
            assert b - a >= 2*B + 2
        pos = b + (B if hr else 0)
    assert pos == nb
    if P < Pmax:                                     # P shrank: one more interior would have been too short
        assert (nb - P*B)//(P + 1) < 2*B + 2


@pytest.mark.parametrize("mmax", [1, 2, 3, 4, 5, 7, 8, 20, 31, 63, 95])
def test_cyclic_reduction_block_pool(lib, mmax):
    """Every block the solver touches -- diagonal blocks, the original couplings (i + 1, i) and the coupling (i + h, i - h) each pivot
    i = (2 k + 1) h leaves behind -- has its own slot inside the pool."""
    used = {}
    def take(br, bc):
        idx = lib.tsba_debug_cr_blk_index(mmax, br, bc)
        assert 0 <= idx < lib.tsba_debug_cr_pool_blocks(mmax)
        assert used.setdefault(idx, (br, bc)) == (br, bc), (idx, used[idx], (br, bc))
    for i in range(mmax):
        take(i, i)
    for i in range(mmax - 1):
        take(i + 1, i)
    h = 1
    while h < mmax:
        for i in range(h, mmax, 2*h):                # pivots of this level
            a, c = i - h, i + h
            take(i, a)                               # read in place (stride-h coupling of this level)
            if c < mmax:
                take(c, i)
                take(c, a)                           # the new stride-2h coupling
        h *= 2
