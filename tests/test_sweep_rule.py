"""CPU test of the strain sweep's PASS RULE, independent of any GPU: the static tables the kernel walks
(clothhip_selftest_windows: window table, dependency masks, reach) are checked for the properties the exactness argument needs,
and a numpy model of the kernel's walk over those very tables -- pre-pass, windows from the first flagged spring on, per pass
"finish every spring none of whose predecessors in the window is over-stretched", end of the walk pushed out by the reach of
every correcting window -- is compared, bit for bit in fp64, with the reference's sequential loop (cloth.pyx:258-296) on
stretched states derived from the golden trajectories."""
import ctypes as C
import os

import numpy as np
import pytest

from test_host_logic import golden_cfg, lib  # noqa: F401  (fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C11 = 1.1


def window_table(lib, n_side):
    L = lib.load()
    p = lib.params_from_cfg({"cloth": {"num_width_points": n_side, "num_height_points": n_side, "width": 1, "height": 1,
                                       "density": 200.0, "ks": 1e4, "damping": 2.0, "thickness": 0.02, "plane_friction": 1.0,
                                       "tear_thresh": 2.0},
                             "frames_per_sec": 30, "simulation_steps": 30, "env": {"grip_radius": 0.003}})
    nw, ns, rs = C.c_int32(), C.c_int32(), C.c_int32()
    lib.check(L.clothhip_selftest_windows(C.byref(p), C.byref(nw), C.byref(ns), C.byref(rs), None, None, None, 0))
    spring_at = np.empty(ns.value, dtype=np.int32)
    ent = np.empty(ns.value, dtype=np.uint32)
    dep = np.empty(ns.value, dtype=np.uint64)
    with pytest.raises(ValueError):
        lib.check(L.clothhip_selftest_windows(C.byref(p), None, None, None, lib.i32p(spring_at), None, None, 5))
    lib.check(L.clothhip_selftest_windows(C.byref(p), None, None, None, lib.i32p(spring_at),
                                          ent.ctypes.data_as(C.POINTER(C.c_uint32)), dep.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          ns.value))
    S = 6 * n_side * n_side                                  # upper bound; the topology call gives the list itself
    a = np.empty(S, dtype=np.int32); b = np.empty(S, dtype=np.int32); t = np.empty(S, dtype=np.uint8)
    lib.check(L.clothhip_spring_topology(C.byref(p), lib.i32p(a), lib.i32p(b), lib.u8p(t)))
    n_springs = int(spring_at.max()) + 1
    return dict(nW=nw.value, n_slots=ns.value, rshift=rs.value, spring_at=spring_at, ent=ent, dep=dep, A=a[:n_springs].astype(np.int64),
                B=b[:n_springs].astype(np.int64), S=n_springs)


@pytest.mark.parametrize("n_side", [25, 50])
def test_window_table_invariants(lib, n_side):
    """What the exactness argument of the sweep rests on: (1) every spring sits in exactly one slot and the entry names its two
    particles; (2) slot order is a linear extension of the reference's list order restricted to springs that share a particle
    (independent springs commute, dependent ones keep their order); (3) dep[slot] holds only EARLIER lanes of the same window,
    contains every earlier lane that shares a particle with the slot's spring, and is transitively closed; (4) no spring incident
    to a particle of window w's springs sits behind window w + reach."""
    W = window_table(lib, n_side)
    A, B, S, sa = W["A"], W["B"], W["S"], W["spring_at"]
    used = sa[sa >= 0]
    assert len(used) == S and np.array_equal(np.sort(used), np.arange(S))
    slot_of = np.empty(S, dtype=np.int64)
    slot_of[used] = np.nonzero(sa >= 0)[0]
    assert W["nW"] == int(slot_of.max()) // 64 + 1 and W["n_slots"] >= (W["nW"] + 4) * 64
    for s in range(S):                                        # (1)
        e = int(W["ent"][slot_of[s]])
        assert (e & 0xFFF) == A[s] and ((e >> 12) & 0xFFF) == B[s]
    last = {}                                                 # (2): per particle, slots must increase along the list
    for s in range(S):
        for p_ in (int(A[s]), int(B[s])):
            assert last.get(p_, -1) < slot_of[s], (s, p_)
            last[p_] = int(slot_of[s])
    last_win = np.zeros(n_side * n_side, dtype=np.int64)
    for s in range(S):
        w = slot_of[s] >> 6
        last_win[A[s]] = max(last_win[A[s]], w); last_win[B[s]] = max(last_win[B[s]], w)
    for w in range(W["nW"]):                                  # (3), (4)
        touch = {}
        for l in range(64):
            s = int(sa[w * 64 + l])
            d = int(W["dep"][w * 64 + l])
            if s < 0:
                assert d == 0
                continue
            assert d >> l == 0, "only earlier lanes"
            direct = touch.get(int(A[s]), 0) | touch.get(int(B[s]), 0)
            assert d & direct == direct, "every earlier lane that shares a particle"
            m, closed = d, direct
            while m:
                j = m.bit_length() - 1
                m &= ~(1 << j)
                assert int(W["dep"][w * 64 + j]) & ~d == 0, "transitively closed"
            # and nothing beyond the closure of the direct predecessors
            frontier = direct
            while frontier:
                j = frontier.bit_length() - 1
                frontier &= ~(1 << j)
                new = int(W["dep"][w * 64 + j]) & ~closed
                closed |= new; frontier |= new
            assert closed == d
            touch[int(A[s])] = touch.get(int(A[s]), 0) | (1 << l); touch[int(B[s])] = touch.get(int(B[s]), 0) | (1 << l)
            reach = (int(W["ent"][w * 64 + l]) >> 28) << W["rshift"]
            assert max(last_win[A[s]], last_win[B[s]]) <= w + reach
        assert W["dep"][(W["nW"]) * 64:].sum() == 0            # the padding windows are empty


def _sequential(pos0, pin, rest, A, B, tear_thresh):
    """cloth.pyx:258-296 in list order."""
    pos = pos0.copy()
    tear = False
    for s in range(len(A)):
        a, b = A[s], B[s]
        if pin[a] and pin[b]:
            continue
        d = pos[a] - pos[b]
        ln = np.sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2])
        if ln > rest[s] * tear_thresh:
            tear = True
        if ln > rest[s] * C11:
            u = d / ln
            e = ln - rest[s] * C11
            if pin[a]:
                pos[b] = pos[b] + u * e
            elif pin[b]:
                pos[a] = pos[a] - u * e
            else:
                pos[a] = pos[a] - u * (e * 0.5)
                pos[b] = pos[b] + u * (e * 0.5)
    return pos, tear


def _window_walk(pos0, pin, rest, W, tear_thresh):
    """The kernel's walk (csrc/cloth_kernels.hpp::strain_sweep), one numpy evaluation per pass."""
    A, B, sa = W["A"], W["B"], W["spring_at"]
    pos = pos0.copy()
    tear = False
    # pre-pass: first / last over-stretched spring in table order at the start state
    valid = np.nonzero(sa >= 0)[0]
    ss = sa[valid].astype(np.int64)
    d = pos[A[ss]] - pos[B[ss]]
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    both = (pin[A[ss]] != 0) & (pin[B[ss]] != 0)
    fl = valid[(~both) & ((ln > rest[ss] * C11) | (ln > rest[ss] * tear_thresh))]
    passes = windows = 0
    if len(fl) == 0:
        return pos, tear, passes, windows
    w, w_end = int(fl.min()) >> 6, int(fl.max()) >> 6
    while w <= w_end:
        windows += 1
        lanes = np.nonzero(sa[w * 64:(w + 1) * 64] >= 0)[0]
        sw = sa[w * 64 + lanes].astype(np.int64)
        dep = [int(x) for x in W["dep"][w * 64 + lanes]]
        a, b = A[sw], B[sw]
        bothw = (pin[a] != 0) & (pin[b] != 0)
        pend = np.ones(len(lanes), dtype=bool)
        while True:
            passes += 1
            d = pos[a] - pos[b]
            ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
            trig = (ln > rest[sw] * C11) & ~bothw & pend
            tb = 0
            for l in lanes[trig]:
                tb |= 1 << int(l)
            bad = np.array([(dep[i] & tb) != 0 for i in range(len(lanes))], dtype=bool) & pend
            fin = pend & ~bad                                     # finished by this pass: tested for tear, corrected if over-stretched
            if ((ln > rest[sw] * tear_thresh) & fin & ~bothw).any():
                tear = True
            if tb == 0:
                break
            w_end = max(w_end, w + ((int(W["ent"][w * 64]) >> 28) << W["rshift"]))
            for i in np.nonzero(trig & ~bad)[0]:
                u = d[i] / ln[i]
                e = ln[i] - rest[sw[i]] * C11
                if pin[a[i]]:
                    pos[b[i]] = pos[b[i]] + u * e
                elif pin[b[i]]:
                    pos[a[i]] = pos[a[i]] - u * e
                else:
                    pos[a[i]] = pos[a[i]] - u * (e * 0.5)
                    pos[b[i]] = pos[b[i]] + u * (e * 0.5)
            pend = bad
            if not pend.any():
                break
        w += 1
    return pos, tear, passes, windows


def _window_walk_mw(pos0, pin, rest, W, tear_thresh, NW):
    """The kernel's walk by all NW waves of a cloth (csrc/cloth_kernels.hpp::strain_sweep_mw): per ROUND the first pass of the next
    NW windows is evaluated against ONE state; the windows before the first flagged one are finished as evaluated, the flagged one
    is run to completion by the pass rule, everything behind it is evaluated again in the next round."""
    A, B, sa = W["A"], W["B"], W["spring_at"]
    pos = pos0.copy()
    tear = False
    valid = np.nonzero(sa >= 0)[0]
    ss = sa[valid].astype(np.int64)
    d = pos[A[ss]] - pos[B[ss]]
    ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    both = (pin[A[ss]] != 0) & (pin[B[ss]] != 0)
    fl = valid[(~both) & ((ln > rest[ss] * C11) | (ln > rest[ss] * tear_thresh))]
    rounds = barriers = 0
    if len(fl) == 0:
        return pos, tear, rounds, barriers
    wb, w_end = int(fl.min()) >> 6, int(fl.max()) >> 6
    w_last = W["n_slots"] // 64 - 1

    def first_pass(w, state):
        lanes = np.nonzero(sa[w * 64:(w + 1) * 64] >= 0)[0]
        sw = sa[w * 64 + lanes].astype(np.int64)
        a, b = A[sw], B[sw]
        bothw = (pin[a] != 0) & (pin[b] != 0)
        d = state[a] - state[b]
        ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        return lanes, sw, a, b, bothw, d, ln

    while wb <= w_end:
        rounds += 1; barriers += 1
        snap = pos.copy()                                         # every wave evaluates against the state at the round's head
        hot, tears = [], []
        for rel in range(NW):
            w = min(wb + rel, w_last)
            lanes, sw, a, b, bothw, d, ln = first_pass(w, snap)
            hot.append(wb + rel <= w_end and bool(((ln > rest[sw] * C11) & ~bothw).any()))
            tears.append(bool(((ln > rest[sw] * tear_thresh) & ~bothw).any()))
        if not any(hot):
            for rel in range(NW):
                if wb + rel <= w_end and tears[rel]:
                    tear = True
            wb += NW
            continue
        f = hot.index(True)
        for rel in range(f):                                      # quiet at the very state the sequential sweep shows them
            if tears[rel]:
                tear = True
        w = wb + f
        barriers += 1
        lanes, sw, a, b, bothw, d, ln = first_pass(w, pos)
        dep = [int(x) for x in W["dep"][w * 64 + lanes]]
        pend = np.ones(len(lanes), dtype=bool)
        w_end = max(w_end, w + ((int(W["ent"][w * 64]) >> 28) << W["rshift"]))
        while True:
            trig = (ln > rest[sw] * C11) & ~bothw & pend
            tb = 0
            for l in lanes[trig]:
                tb |= 1 << int(l)
            bad = np.array([(dep[i] & tb) != 0 for i in range(len(lanes))], dtype=bool) & pend
            fin = pend & ~bad
            if ((ln > rest[sw] * tear_thresh) & fin & ~bothw).any():
                tear = True
            if tb == 0:
                break
            for i in np.nonzero(trig & ~bad)[0]:
                u = d[i] / ln[i]
                e = ln[i] - rest[sw[i]] * C11
                if pin[a[i]]:
                    pos[b[i]] = pos[b[i]] + u * e
                elif pin[b[i]]:
                    pos[a[i]] = pos[a[i]] - u * e
                else:
                    pos[a[i]] = pos[a[i]] - u * (e * 0.5)
                    pos[b[i]] = pos[b[i]] + u * (e * 0.5)
            pend = bad
            if not pend.any():
                break
            d = pos[a] - pos[b]
            ln = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        wb += f + 1
    return pos, tear, rounds, barriers


def test_pass_rule_equals_sequential_sweep(lib):
    """Stretched states (golden checkpoints of the lift-and-pull and fold trajectories blown up about their centroid, pulled at
    their pins, jittered; incl. a torn one and springs with both ends pinned): the window walk over the library's tables gives
    the sequential loop's positions and tear flag bit for bit, and needs fewer passes than one per corrected level."""
    from oracle.pyoracle import load_golden
    W = window_table(lib, 25)
    rng = np.random.RandomState(7)
    cases = 0
    stats = []
    for name, cps in (("g_traj_lift_pull_25.npz", (4, 9, 12)), ("g_traj_fold_25.npz", (3, 6)), ("g_traj_tear_25.npz", (2,))):
        g = load_golden(name)
        assert np.array_equal(g["spring_a"], W["A"]) and np.array_equal(g["spring_b"], W["B"])
        rest = np.asarray(g["rest"], dtype=np.float64)
        tt = float(g["cfg"]["tear_thresh"])
        for cp in cps:
            if cp >= len(g["cp_pos"]):
                continue
            base = np.asarray(g["cp_pos"][cp], dtype=np.float64)
            pin = np.asarray(g["cp_pinned"][cp]).astype(np.uint8).copy()
            for variant in range(4):
                pos = base.copy()
                if variant == 0:                                  # uniform blow-up: nearly every spring over its limit
                    pos = pos.mean(axis=0) + (pos - pos.mean(axis=0)) * 1.13
                elif variant == 1:                                # a local stretch: the pinned points (or a corner) dragged away
                    idx = np.nonzero(pin)[0] if pin.any() else np.array([0, 1, 25])
                    pos[idx] += np.array([0.03, 0.02, 0.05])
                elif variant == 2:                                # jitter: scattered over-stretched springs, short chains
                    pos += rng.normal(scale=0.004, size=pos.shape)
                else:                                             # neighbouring pinned pairs (springs with both ends pinned) + a far drag
                    pin[[100, 101, 126, 300, 325]] = 1
                    pos[[100, 126, 300]] += np.array([0.0, 0.06, 0.1])
                    pos[350:360] += np.array([0.2, 0.0, 0.3])
                p_seq, t_seq = _sequential(pos, pin, rest, W["A"], W["B"], tt)
                p_win, t_win, passes, windows = _window_walk(pos, pin, rest, W, tt)
                assert np.array_equal(p_seq, p_win), (name, cp, variant, float(np.abs(p_seq - p_win).max()))
                assert t_seq == t_win, (name, cp, variant)
                for nw in (4, 8, 16):                              # the same walk by all waves of a cloth, a window each per round
                    p_mw, t_mw, rounds, barriers = _window_walk_mw(pos, pin, rest, W, tt, nw)
                    assert np.array_equal(p_seq, p_mw), (name, cp, variant, nw, float(np.abs(p_seq - p_mw).max()))
                    assert t_seq == t_mw, (name, cp, variant, nw)
                    assert rounds <= max(windows, 1)
                moved = int((p_seq != pos).any(axis=1).sum())
                stats.append((moved, passes, windows))
                cases += 1
    assert cases >= 20
    assert max(m for m, _, _ in stats) > 300 and sum(1 for _, _, wd in stats if 0 < wd < 20) >= 1     # dense and local cases both occur
