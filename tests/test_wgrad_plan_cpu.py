"""CPU: the launch plan of the 16-bit weight gradient (scade_mlp_wgrad_lp_plan, host code of mlp_bwd_lp.hip).
The kernel maps workgroup w to the stages [stage(bound[w]), stage(bound[w+1])) of every entry its range touches,
stage(x) = clamp(floor((x - cum[e]) / weight), 0, stages); this emulates that mapping and checks that every stage of
every (network, job) entry is done exactly once, that segment k of an entry lands on partial row k < rows, and that
the workgroups carry balanced work."""
import ctypes

import pytest


def _plan(P0, P1, s8):
    from scade_amd import _lib
    lib = _lib.load()
    P = (ctypes.c_int * 2)(P0, P1)
    info = (ctypes.c_int * 7)()
    bound = (ctypes.c_int * 257)(); cum = (ctypes.c_int * 33)()
    first = (ctypes.c_int * 32)(); nseg = (ctypes.c_int * 32)(); weight = (ctypes.c_int * 16)()
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    rc = lib.scade_mlp_wgrad_lp_plan(vp(P), int(s8), vp(info), vp(bound), vp(cum), vp(first), vp(nseg), vp(weight))
    assert rc == 0, lib.scade_last_error()
    return list(info), list(bound), list(cum), list(first), list(nseg), list(weight)


@pytest.mark.parametrize("s8", [0, 1])
@pytest.mark.parametrize("P0,P1", [(196608, 65536), (65536, 196608), (786432, 262144), (262144, 0), (100000, 1),
                                   (131072, 131072), (99999, 50001), (3_000_001, 1_234_567)])
def test_balanced_plan_partitions_every_entry_once(P0, P1, s8):
    info, bound, cum, first, nseg, weight = _plan(P0, P1, s8)
    nwg, nj, pt, rows, chunk = info[0], info[1], info[2], info[3], info[4]
    assert chunk == 0 and 1 <= nwg <= 256 and nj == 13 and pt == 32
    ne = 2 * nj
    W = cum[ne]
    assert bound[0] == 0 and bound[nwg] == W and all(bound[w] <= bound[w + 1] for w in range(nwg))
    nst = [(P0 + pt - 1) // pt, (P1 + pt - 1) // pt]
    covered = [0] * ne          # next stage every entry expects
    rows_written = [set() for _ in range(ne)]
    work = []
    for w in range(nwg):
        b0, b1 = bound[w], bound[w + 1]
        mine = 0
        for e in range(ne):
            ce, ce1 = cum[e], cum[e + 1]
            if ce1 == ce or ce >= b1 or ce1 <= b0:
                continue
            wj, n = weight[e % nj], nst[e // nj]
            x0, x1 = b0 - ce, b1 - ce
            s0 = 0 if x0 <= 0 else min(n, x0 // wj)
            s1 = min(n, x1 // wj)
            assert s0 == covered[e], (w, e, s0, covered[e])     # contiguous, no stage twice or skipped
            covered[e] = max(s1, s0)
            row = w - first[e]
            assert 0 <= row < nseg[e] <= rows and row not in rows_written[e]
            rows_written[e].add(row)
            mine += (max(s1, s0) - s0) * wj
        work.append(mine)
    for e in range(ne):
        n = nst[e // nj]
        assert covered[e] == n, (e, covered[e], n)
        assert len(rows_written[e]) == nseg[e] or cum[e + 1] == cum[e]     # every counted row is written
    busy = [x for x in work if x > 0]
    # balanced: no workgroup carries more than the mean + the largest per-segment allowance
    assert max(busy) <= sum(busy) / len(busy) * 1.25 + 2 * max(weight), (max(busy), sum(busy) / len(busy))


@pytest.mark.parametrize("P0,P1", [(24576, 8192), (8192, 24576), (4096, 0), (600, 200), (70000, 29999)])
def test_small_launches_use_a_one_round_grid(P0, P1):
    info, *_ = _plan(P0, P1, 0)
    nwg, nj, pt, rows, chunk, gx0, gx1 = info
    assert chunk > 0 and chunk % pt == 0
    assert gx0 == (P0 + chunk - 1) // chunk and gx1 == (P1 + chunk - 1) // chunk and max(gx0, gx1) <= rows
    assert nwg == (gx0 + gx1) * nj
    if P0 + P1 >= 512 * 19:
        assert nwg <= 256 + nj            # one round of workgroups (the two networks round up separately)


def test_plan_rejects_bad_sizes():
    from scade_amd import _lib
    lib = _lib.load()
    P = (ctypes.c_int * 2)(0, 5)
    z = (ctypes.c_int * 300)()
    vp = lambda a: ctypes.cast(a, ctypes.c_void_p)
    assert lib.scade_mlp_wgrad_lp_plan(vp(P), 0, vp(z), vp(z), vp(z), vp(z), vp(z), vp(z)) != 0
    assert lib.scade_mlp_wgrad_lp_plan(None, 0, vp(z), vp(z), vp(z), vp(z), vp(z), vp(z)) != 0
