"""Host-side index arithmetic of the partitioned band solver (tsba_bandp.h / tsba_bandcr.h) through the library's debug hooks: no GPU
needed.  The partition table every workgroup derives on the device, and the block pool of the cyclic-reduction separator solver."""
import ctypes as C
import numpy as np
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
    L.tsba_debug_sv_lmax.argtypes = [C.c_int, C.c_int, C.c_int]; L.tsba_debug_sv_lmax.restype = C.c_int
    return L


def _part(lib, nb, B, Pmax, p):
    o = (C.c_int*5)(); lib.tsba_debug_bandp_part(nb, B, Pmax, p, o); return list(o)


@pytest.mark.parametrize("nb,B,Pmax", [(4998, 10, 64), (998, 12, 11), (298, 8, 8), (100, 10, 21), (25, 10, 4), (7, 1, 3), (598, 13, 12)])
def test_partition_covers_the_band(lib, nb, B, Pmax):
    """Interiors [a, b) and separators of B blocks tile [0, nb); every interior holds >= 2 B + 2 blocks unless there is only one."""
    P = _part(lib, nb, B, Pmax, 0)[0]
    assert 1 <= P <= Pmax
    pos = 0
    sizes = []
    for p in range(P):
        Pp, a, b, hl, hr = _part(lib, nb, B, Pmax, p)
        assert Pp == P and a == pos and b > a
        sizes.append(b - a)
        assert hl == (p > 0) and hr == (p < P - 1)
        if P > 1:
            assert b - a >= 2*B + 2
        pos = b + (B if hr else 0)
    assert pos == nb
    assert max(sizes) - min(sizes) <= 1              # balanced: the launch lasts as long as its longest interior
    if P < Pmax:                                     # P shrank: one more interior would have been too short
        assert (nb - P*B)//(P + 1) < 2*B + 2


def test_solve_phase_bound_on_an_interiors_length(lib):
    """The single-vector solve phase (csrc/tsba_bandsv.h) keeps one value per row of an interior in LDS, sized on the host from the number of keyframes
    and the number of interiors asked for -- while the device partitions the FREE poses, of which there may be fewer, into possibly fewer interiors:
    no interior of any such partition is longer than the bound."""
    rng = np.random.default_rng(3)
    for _ in range(400):
        B = int(rng.integers(1, 14)); Pmax = int(rng.integers(2, 130)); n_kf = int(rng.integers(2*B + 4, 6000))
        lmax = lib.tsba_debug_sv_lmax(n_kf, B, Pmax)
        for nf in {n_kf, max(1, n_kf - 1), max(1, n_kf//2), max(1, int(rng.integers(1, n_kf + 1))), min(n_kf, 3*B + 3)}:
            P = _part(lib, nf, B, Pmax, 0)[0]
            longest = max(_part(lib, nf, B, Pmax, p)[2] - _part(lib, nf, B, Pmax, p)[1] for p in range(P))
            assert longest <= lmax, (n_kf, nf, B, Pmax, P, longest, lmax)


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


def _plan_band(P, o, reorder):
    import ctypes as C
    from textslam_amd.optimizer import load_library
    from textslam_amd import abi
    L = load_library()
    L.tsba_debug_plan_band.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    s = P.struct(); bw = C.c_int32(0); order = np.zeros(P.n_kf, np.int32)
    assert L.tsba_debug_plan_band(C.byref(s), C.byref(o), 0, reorder, C.byref(bw), order.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return bw.value, order


def _envelope(P, order):
    """Half bandwidth (pose blocks, Cholesky fill closed) of the reduced camera matrix under a row order, straight from the observations."""
    n = P.n_kf
    pos = np.empty(n, np.int64); pos[order] = np.arange(n)
    kf, host = P.sobs_kf[0].astype(np.int64), P.pt_host[P.sobs_pt[0]].astype(np.int64)
    m = (host >= 0) & (host != kf)
    lo = np.full(P.n_pt, n, np.int64); hi = np.full(P.n_pt, -1, np.int64)
    for a in (pos[kf[m]], pos[host[m]]):
        np.minimum.at(lo, P.sobs_pt[0][m], a); np.maximum.at(hi, P.sobs_pt[0][m], a)
    reach = np.arange(n)
    ok = hi >= 0
    np.maximum.at(reach, lo[ok], hi[ok])
    run, bw = -1, 0
    for k in range(n):
        r = max(reach[k], run) if run >= k else reach[k]
        run = max(run, r); bw = max(bw, r - k)
    return bw


def test_keyframe_reordering_of_a_loop_closure_map():
    """Host side of the loop-closure handling (no GPU): on a ring map the keyframe-order envelope spans the matrix, the reverse
    Cuthill-McKee order of the plan brings it to about twice the local band; the reported bound is the true envelope under that order;
    a banded map keeps the keyframe order."""
    from textslam_amd import synth, abi
    o = abi.options_global()
    P = synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True)
    bw_id, order_id = _plan_band(P, o, 0)
    bw_rcm, order = _plan_band(P, o, 1)
    assert np.array_equal(order_id, np.arange(600)) and bw_id >= 590
    assert sorted(order.tolist()) == list(range(600)) and not np.array_equal(order, np.arange(600))
    assert bw_rcm <= 3*8 and bw_rcm == _envelope(P, order) and bw_id == _envelope(P, np.arange(600))
    Q = synth.config_global(n_kf=300, n_pt=6000, band=8)
    bw_b, order_b = _plan_band(Q, o, 1)
    assert np.array_equal(order_b, np.arange(300)) and bw_b <= 9 and bw_b == _envelope(Q, np.arange(300))
    # every rank of a sharded solve derives the same order and band (the graph comes from all observations)
    for r in range(3):
        oo = abi.options_global(); oo.lm_shard, oo.lm_nshard = r, 3
        bw_r, order_r = _plan_band(P, oo, 1)
        assert bw_r == bw_rcm and np.array_equal(order_r, order)


@pytest.mark.parametrize("nb,B,Pmax", [(5008, 10, 128), (608, 9, 16), (130, 9, 4), (1510, 8, 8), (2411, 11, 64)])
def test_ring_partition_starts_and_ends_with_a_separator(lib, nb, B, Pmax):
    """Ring maps (nb counts the B ghost blocks behind the last pose): [sep 0][interior 0][sep 1] ... [interior P-1][ghost of sep 0], P a
    power of two, balanced interiors of at least 2 B + 2 blocks, every interior with both neighbours."""
    lib.tsba_debug_bandp_part_ring.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]; lib.tsba_debug_bandp_part_ring.restype = None
    def part(p):
        o = (C.c_int*5)(); lib.tsba_debug_bandp_part_ring(nb, B, Pmax, p, o); return list(o)
    P = part(0)[0]
    assert 2 <= P <= Pmax and P & (P - 1) == 0
    pos, sizes = B, []
    for p in range(P):
        Pp, a, b, hl, hr = part(p)
        assert Pp == P and a == pos and b > a and hl == 1 and hr == 1
        sizes.append(b - a); pos = b + B
    assert pos == nb and max(sizes) - min(sizes) <= 1 and (min(sizes) >= 2*B + 2 or P == 2)


def test_ring_plan_of_a_loop_closure_map():
    """Host side of the ghost-row path (no GPU): a map whose last keyframes see the landmarks of the first is recognised as a ring and keeps
    the band of the open chain; an open chain, a ring with a closure too wide for the separators, and a map with long-range
    observations all over are not."""
    from textslam_amd import synth, abi
    from textslam_amd.optimizer import load_library
    L = load_library()
    L.tsba_debug_plan_ring.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int, C.POINTER(C.c_int32)]
    o = abi.options_global()
    def ring(P, mx):
        s = P.struct(); bw = C.c_int32(0)
        r = L.tsba_debug_plan_ring(C.byref(s), C.byref(o), 0, mx, C.byref(bw)); assert r >= 0
        return r, bw.value
    P = synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True)
    r, bw = ring(P, 13)
    assert r == 1 and 8 <= bw <= 11                          # the open chain's band (+ fill), not the 23 of the reordered rows (upper bits: first keyframe of the loop = 0)
    bw_rcm, _ = _plan_band(P, o, 1)
    assert bw_rcm >= 2*bw - 4
    assert ring(P, 6)[0] == 0                                 # separators of at most 6 pose blocks cannot hold a band of 8
    assert ring(P, 0)[0] == 0                                 # not asked for (multi-GPU, several pyramid levels)
    assert ring(synth.config_global(n_kf=600, n_pt=12000, band=8), 13)[0] == 0                  # open chain
    r, bw = ring(synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True, loop_at=200), 13) # a tail before the loop: keyframe order kept, the loop starts at 200
    assert r & 15 == 1 and r >> 4 == 200 and 8 <= bw <= 11
    assert ring(synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True, loop_at=30), 13)[0] == 0     # a tail too short for an interior
    assert ring(synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True, loop_at=480), 13)[0] == 0    # a loop too short for four interiors
    assert ring(synth.config_global(n_kf=600, n_pt=12000, band=8, far_frac=0.02), 13)[0] == 0   # long-range observations: not a ring


def test_plan_does_not_depend_on_the_number_of_host_threads():
    """The slot-pair lists of the Schur complement (2 M entries at 5000 keyframes) are placed by several host threads on large maps: the
    lists must be the ones a single thread builds (block-major, landmark-major within a block)."""
    from textslam_amd import synth, abi
    from textslam_amd.optimizer import load_library
    L = load_library()
    L.tsba_debug_plan_checksum.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int]; L.tsba_debug_plan_checksum.restype = C.c_ulonglong
    L.tsba_debug_plan_knob.argtypes = [C.c_int, C.c_int]; L.tsba_debug_plan_knob.restype = None
    og3 = abi.options_global(); og3.lm_shard, og3.lm_nshard = 1, 3
    sums = set()
    for P, o, ring in ((synth.config_global(n_kf=700, n_pt=20000, band=9), abi.options_global(), 0), (synth.tiny(seed=3, n_kf=8, n_pt=300, n_text=6), abi.options_local(), 0),
                       (synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True, loop_at=200), abi.options_global(), 13),      # ring plan (ghost rows)
                       (synth.config_global(n_kf=600, n_pt=12000, band=8, loop=True), abi.options_global(), 0),                    # reordered plan
                       (synth.config_global(n_kf=700, n_pt=20000, band=9), og3, 0)):                                               # one shard of three
        s = P.struct()
        L.tsba_debug_plan_knob(2, ring)                          # (ring_max_blocks of the hook's plan)
        try:
            ref = L.tsba_debug_plan_checksum(C.byref(s), C.byref(o), 0, 1)     # (a checksum over EVERY list of the plan)
            assert ref != 0
            for t in (2, 3, 7, 16):
                assert L.tsba_debug_plan_checksum(C.byref(s), C.byref(o), 0, t) == ref, t
        finally:
            L.tsba_debug_plan_knob(2, 0)
        sums.add(ref)
    assert len(sums) == 5
    # a context keeps its plan objects (lists and work lists) from call to call: a plan built into an object that held ANOTHER plan -- larger or
    # smaller, built by more or fewer threads -- is the plan a fresh object gets
    L.tsba_debug_plan_checksum_recycled.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int] * 2
    L.tsba_debug_plan_checksum_recycled.restype = C.c_ulonglong
    big, small, win = synth.config_global(n_kf=700, n_pt=20000, band=9), synth.config_global(n_kf=60, n_pt=3000, band=6), synth.tiny(seed=3, n_kf=8, n_pt=300, n_text=6)
    og, ol = abi.options_global(), abi.options_local()
    probs = [(big, og, 0), (small, og, 0), (win, ol, 0), (win, ol, 2)]
    for (Pw, ow, lw) in probs:
        for (Pn, on, ln) in probs:
            sw, sn = Pw.struct(), Pn.struct()
            fresh = L.tsba_debug_plan_checksum(C.byref(sn), C.byref(on), ln, 1)
            for tw, tn in ((16, 1), (1, 5), (3, 16)):
                assert L.tsba_debug_plan_checksum_recycled(C.byref(sw), C.byref(ow), lw, tw, C.byref(sn), C.byref(on), ln, tn) == fresh


def test_single_frame_plan_is_the_generic_plan():
    """tsba_pose_optim's problems (one frame, every landmark frozen in a host outside it) get their plan from build_plan_single_frame -- the lists written
    down directly on the calling thread -- instead of build_plan's generic passes on plan threads: every list must be the one the generic builder makes
    (checksum over every list of the plan, per level; with / without text planes, without scene observations, into a recycled plan object)."""
    from textslam_amd import synth, abi
    from textslam_amd.optimizer import load_library
    L = load_library()
    L.tsba_debug_plan_checksum.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int]; L.tsba_debug_plan_checksum.restype = C.c_ulonglong
    L.tsba_debug_plan_checksum_single_frame.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int]; L.tsba_debug_plan_checksum_single_frame.restype = C.c_ulonglong
    o = abi.options_pose(); o_nt = abi.options_pose(); o_nt.use_text = 0
    cases = [(synth.config_c3(), o), (synth.config_c3(seed=5), o_nt),
             (synth.make_problem(1, 40, 3, 11, feats=(8, 6, 4), frozen_frac=1.0, n_out=2, max_targets=1, text_targets=1), o),
             (synth.make_problem(1, 200, 0, 12, frozen_frac=1.0, max_targets=1), o)]
    seen = set()
    for P, oo in cases:
        s = P.struct()
        for lvl in range(P.n_levels):
            ref = L.tsba_debug_plan_checksum(C.byref(s), C.byref(oo), lvl, 1)
            assert ref != 0
            for recycled in (0, 1):
                assert L.tsba_debug_plan_checksum_single_frame(C.byref(s), C.byref(oo), lvl, recycled) == ref, (lvl, recycled)
            seen.add(ref)
    assert len(seen) >= 9
    # not such a problem: a window, a frame with a landmark hosted in it
    W = synth.tiny(); sw = W.struct()
    assert L.tsba_debug_plan_checksum_single_frame(C.byref(sw), C.byref(abi.options_local()), 0, 0) == 0


@pytest.mark.parametrize("nf,row0,B,Gmax,Ptmax", [(598, 199, 8, 16, 8), (4998, 1499, 10, 128, 55), (1498, 299, 7, 64, 16), (898, 449, 8, 32, 32), (300, 60, 9, 8, 8), (1498, 999, 7, 16, 40)])
def test_ring_partition_with_a_tail(lib, nf, row0, B, Gmax, Ptmax):
    """[tail: interior, sep, ..., interior][S][loop: interior, sep, ..., interior][ghost of S]: the interiors and separators tile the rows, the
    loop has a power of two of interiors, the tail fewer than 128, and the separator labels run upwards from the tail's first separator
    through S = RING_OFF = 128 to the ghost."""
    lib.tsba_debug_bandp_part_ring2.argtypes = [C.c_int]*6 + [C.POINTER(C.c_int)]; lib.tsba_debug_bandp_part_ring2.restype = None
    def part(p):
        o = (C.c_int*8)(); lib.tsba_debug_bandp_part_ring2(nf, row0, B, Gmax + Ptmax, Gmax, p, o); return list(o)
    P, _, _, _, _, G, Pt, _ = part(0)
    assert P == G + Pt and G & (G - 1) == 0 and 2 <= G <= Gmax and 1 <= Pt <= min(127, Ptmax)
    pos = 0
    for p in range(P):
        Pp, a, b, hl, hr, Gp, Ptp, lbl = part(p)
        assert (Pp, Gp, Ptp) == (P, G, Pt) and a == pos and b > a and hr == 1 and hl == (p > 0) and lbl == 128 - Pt + p
        assert b - a >= 2*B + 2 or (p < Pt and Pt == 1) or G == 2
        pos = b + B                                              # the separator on its right
        if p == Pt - 1: assert b == row0                         # the tail ends where S begins
    assert pos == nf + B                                         # the ghost of S behind the last pose


def test_plan_threads_leave_the_callers_cpu_affinity_alone():
    """Pinning of the plan threads is opt-in (tsba_options.host_plan_pin; debug knob 3) and applies to the WORKER threads of a multi-threaded
    build only (tsba_plan.h: PlanPool): the calling thread's affinity mask is never touched, and the plan is the same either way."""
    import os
    from textslam_amd import synth, abi
    from textslam_amd.optimizer import load_library
    L = load_library()
    L.tsba_debug_plan_checksum.argtypes = [C.POINTER(abi.TsbaProblem), C.POINTER(abi.TsbaOptions), C.c_int, C.c_int]; L.tsba_debug_plan_checksum.restype = C.c_ulonglong
    L.tsba_debug_plan_knob.argtypes = [C.c_int, C.c_int]; L.tsba_debug_plan_knob.restype = None
    P = synth.config_global(n_kf=300, n_pt=8000, band=8); o = abi.options_global(); s = P.struct()
    assert o.host_plan_pin == 0                                   # the default leaves scheduling alone
    before = os.sched_getaffinity(0)
    sums = []
    try:
        for pin in (1, 0, 1):
            L.tsba_debug_plan_knob(3, pin)
            sums.append(L.tsba_debug_plan_checksum(C.byref(s), C.byref(o), 0, 6))
            assert os.sched_getaffinity(0) == before
        L.tsba_debug_plan_knob(3, 0)
        o.host_plan_pin = 1                                       # the per-call option
        sums.append(L.tsba_debug_plan_checksum(C.byref(s), C.byref(o), 0, 6))
        assert os.sched_getaffinity(0) == before
    finally:
        L.tsba_debug_plan_knob(3, 0)
    assert sums[0] == sums[1] == sums[2] == sums[3] != 0
