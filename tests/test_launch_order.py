"""Host logic of the launch-order calibration (csrc/launch_order.cpp plan_order_from_measurement / set_tile_order; CPU only).
The plan is scheduling -- it can never change a pixel -- but it must always be a PERMUTATION of the tile rows, and on
a measured profile like BASELINE config C3's (tools/timeline.py) it must move the bottom of the hot range in front of
its middle."""
import numpy as np
import pytest


def records(starts_us, longest_us):
    rec = np.zeros((len(starts_us), 2), dtype=np.uint64)
    rec[:, 0] = (np.asarray(starts_us) * 100).astype(np.uint64) + 123456789  # (an arbitrary clock origin)
    rec[:, 1] = (np.asarray(longest_us) * 100).astype(np.uint64)
    return rec


def c3_like_profile(tile_rows=135, rot=43, end=103):
    """Starts under the rotation and longest waves per tile row, shaped like the measured C3 launch: horizon rows with
    very long waves first, a long middle of 40 us waves, a bottom band of 60 us waves, sky rows of 1 us."""
    q = np.full(tile_rows, 1.0)
    for t in range(rot, end):
        i = t - rot
        q[t] = 120.0 if i < 7 else (75.0 if i < 14 else (40.0 if t < end - 16 else 60.0))
    start = np.zeros(tile_rows)
    clock = 0.0
    for i in range(tile_rows):
        t = (rot + i) % tile_rows
        start[t] = clock
        clock += 1.4 if rot <= t < end else 0.25   # marching rows take longer to hand out
    return records(start, q), q, start


def model_makespan(perm, q, dispatch):
    """max over rows of (start under the order + longest wave), the plan's own model."""
    clock, worst = 0.0, 0.0
    for t in perm:
        worst = max(worst, clock + q[t])
        clock += dispatch[t]
    return worst


def test_c3_like_profile_moves_the_bottom_band_forward(hmrm):
    rec, q, start = c3_like_profile()
    pieces, perm = hmrm.plan_order(rec, 43)
    assert sorted(perm.tolist()) == list(range(135))            # a permutation of the tile rows
    assert len(pieces) >= 2 and perm[0] == 43                   # the horizon rows still start first
    pos = {int(t): i for i, t in enumerate(perm)}
    assert pos[95] < pos[70]                                    # bottom band (60 us waves) before the middle (40 us)
    dispatch = np.where((np.arange(135) >= 43) & (np.arange(135) < 103), 1.4, 0.25)
    rotation = [(43 + i) % 135 for i in range(135)]
    assert model_makespan(perm, q, dispatch) < 0.97 * model_makespan(rotation, q, dispatch)
    # the pieces are disjoint, contiguous together, and inside the frame
    rows = [t for b, c in pieces for t in range(b, b + c)]
    assert len(set(rows)) == len(rows) and min(rows) == 43 and max(rows) - min(rows) + 1 == len(rows)


def test_flat_and_monotone_profiles_keep_the_rotation(hmrm):
    n, rot = 64, 10
    start = np.array([((t - rot) % n) * 1.0 for t in range(n)])
    for q in (np.full(n, 30.0), np.array([max(1.0, 80.0 - 1.2 * ((t - rot) % n)) for t in range(n)])):
        pieces, perm = hmrm.plan_order(records(start, q), rot)
        assert pieces == [] and perm.tolist() == [(rot + i) % n for i in range(n)]


@pytest.mark.parametrize("seed", range(40))
def test_random_profiles_always_give_a_permutation(hmrm, seed):
    rng = np.random.RandomState(seed)
    n = int(rng.randint(12, 300))
    rot = int(rng.randint(0, n))
    q = rng.uniform(0.5, 100.0, size=n) * (rng.uniform(size=n) < rng.uniform(0.2, 1.0))
    q = np.maximum(q, 0.3)
    d = rng.uniform(0.1, 2.0, size=n)
    start = np.zeros(n)
    clock = 0.0
    for i in range(n):
        t = (rot + i) % n
        start[t] = clock + (rng.uniform(-0.3, 0.3) if seed % 3 == 0 else 0.0)  # (noisy starts too)
        clock += d[t]
    start -= start.min()
    pieces, perm = hmrm.plan_order(records(start, q), rot)
    assert sorted(perm.tolist()) == list(range(n)), (pieces, rot, n)
    assert len(pieces) <= 3
    if pieces:
        rows = [t for b, c in pieces for t in range(b, b + c)]
        assert len(set(rows)) == len(rows) and all(0 <= t < n for t in rows)
        assert max(rows) - min(rows) + 1 == len(rows) and min(rows) >= rot   # one contiguous range, not wrapping


def test_bad_arguments_and_missing_records(hmrm):
    rec, _, _ = c3_like_profile()
    with pytest.raises(hmrm.HmrmError):
        hmrm.plan_order(rec, 135)
    rec[50, 0] = 0                       # a tile row that never reported: no plan, the rotation stays
    pieces, perm = hmrm.plan_order(rec, 43)
    assert pieces == [] and perm.tolist() == [(43 + i) % 135 for i in range(135)]
    pieces, perm = hmrm.plan_order(rec[:8], 3)   # too few rows to bother
    assert pieces == [] and sorted(perm.tolist()) == list(range(8))


# ---- the calibration's state machine (csrc/launch_order.hpp OrderCalibration / KernelChoice; hmrm_debug_calibrate) ----
def _launch_records(n_launches, scale):
    """Per launch the C3-like profile with every wave `scale[i]` times as long (and the dispatch as slow): a launch whose
    measured makespan is scale[i] times the base one."""
    rec0, q, start = c3_like_profile()
    out = np.zeros((n_launches,) + rec0.shape, dtype=np.uint64)
    for i in range(n_launches):
        out[i] = records(start * scale[i], q * scale[i])
    return out


def test_calibration_sequence_and_settling(hmrm):
    rot = 43
    # launch 0 is never measured; then two samples per trial: rotation, the model's plan, the generic split, the other kernel
    r = hmrm.calibrate(_launch_records(12, [1.0] * 12), rot)
    assert r["n_trials"] == 4 and r["measured"] == [0, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0]
    assert r["trial"][:9] == [0, 0, 0, 1, 1, 2, 2, 3, 3] and r["group"][:9] == [0] * 7 + [1, 1]
    # nothing beat the rotation by 1 % (all makespans equal): the rotation stays, the production kernel stays
    assert r["settled_at"] == 8 and r["best"] == 0 and not r["scene_use_group"] and r["trial"][9:] == [0, 0, 0] and r["group"][9:] == [0, 0, 0]
    # trial 1 faster by 0.5 %: not enough; by 2 %: it stays; the shorter of a trial's two samples counts
    r = hmrm.calibrate(_launch_records(12, [1, 1, 1, 0.995, 0.995, 1, 1, 1, 1, 1, 1, 1]), rot)
    assert r["best"] == 0
    r = hmrm.calibrate(_launch_records(12, [1, 1, 1, 1.3, 0.98, 1, 1, 1, 1, 1, 1, 1]), rot)
    assert r["best"] == 1 and r["trial"][9:] == [1, 1, 1] and r["group"][9:] == [0, 0, 0]
    # two orders beat the rotation: the shorter one
    r = hmrm.calibrate(_launch_records(12, [1, 1, 1, 0.98, 0.98, 0.96, 0.97, 1, 1, 1, 1, 1]), rot)
    assert r["best"] == 2
    # the other kernel must beat the BEST order by 3 %
    r = hmrm.calibrate(_launch_records(12, [1, 1, 1, 0.98, 0.98, 1, 1, 0.96, 0.96, 1, 1, 1]), rot)
    assert r["best"] == 1 and not r["scene_use_group"]
    r = hmrm.calibrate(_launch_records(12, [1, 1, 1, 0.98, 0.98, 1, 1, 0.94, 0.99, 1, 1, 1]), rot)
    assert r["best"] == 3 and r["scene_use_group"] and r["group"][9:] == [1, 1, 1] and r["trial"][9:] == [3, 3, 3]


def test_calibration_probe_only_once_per_scene_and_only_when_allowed(hmrm):
    rot = 43
    for kw in ({"scene_already_probed": True}, {"may_probe": False}):
        r = hmrm.calibrate(_launch_records(10, [1.0] * 10), rot, **kw)
        assert r["n_trials"] == 3 and r["settled_at"] == 6 and sum(r["group"]) == 0 and sum(r["measured"]) == 6


def test_calibration_waits_while_it_cannot_measure(hmrm):
    rot = 43
    can = [1, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1]
    r = hmrm.calibrate(_launch_records(14, [1.0] * 14), rot, can_measure=can)
    # launches that may not be measured render with the rotation (or the order in force) and the machine does not advance
    assert r["measured"] == [0, 0, 0, 1, 0, 1, 1, 1, 1, 1, 1, 1, 0, 0] and r["trial"][:12] == [0, 0, 0, 0, 0, 0, 1, 1, 2, 2, 3, 3]
    assert r["settled_at"] == 11
    # a profile the model finds nothing in (every wave equally long: no hot range to reorder) still times the rotation and the probe
    flat = np.zeros((8, 40, 2), dtype=np.uint64)
    flat[:, :, 0] = 1000 + np.arange(40)[None, :] * 10
    flat[:, :, 1] = 500
    r = hmrm.calibrate(flat, 0)
    assert r["n_trials"] in (2, 3) and r["settled_at"] >= 4 and r["best"] in (0, r["n_trials"] - 1)


def test_which_kernel_a_frame_is_launched_with(hmrm):
    """launch_order.hpp pick_fast_kernel: the other kernel is the record kernel where the frame can run it (nearest sampling),
    else the plain groups; a verdict holds for the frames that would run what the probe measured; HMRM_KERNEL overrides."""
    PLAIN, LEAPS, RECORDS = 0, 1, 2
    pick = hmrm.pick_kernel
    # nothing asks for the other kernel: production, whatever the frame could run
    assert pick(records_ok=True)[0] == LEAPS and pick(records_ok=False)[0] == LEAPS
    # a probe's own launch (no verdict yet): the alternative of THIS frame, remembered
    assert pick(use_other=True, records_ok=True) == (RECORDS, True)
    assert pick(use_other=True, records_ok=False, verdict_with_records=True) == (PLAIN, False)
    # a verdict obtained with the records: nearest frames run them, bilinear / float-heights frames keep the production kernel
    assert pick(use_other=True, records_ok=True, verdict=True, verdict_with_records=True) == (RECORDS, True)
    assert pick(use_other=True, records_ok=False, verdict=True, verdict_with_records=True) == (LEAPS, True)
    # ... and one obtained with the plain groups (a bilinear camera was probed) the other way round
    assert pick(use_other=True, records_ok=False, verdict=True, verdict_with_records=False) == (PLAIN, False)
    assert pick(use_other=True, records_ok=True, verdict=True, verdict_with_records=False) == (LEAPS, False)
    # HMRM_KERNEL=group / rec
    for verdict in (False, True):
        assert pick(forced=1, use_other=verdict, records_ok=True, verdict=verdict)[0] == PLAIN
        assert pick(forced=3, records_ok=True, verdict=verdict)[0] == RECORDS
        assert pick(forced=3, records_ok=False, verdict=verdict)[0] == PLAIN
