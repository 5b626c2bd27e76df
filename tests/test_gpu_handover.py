"""The free-running kernel's hand-over of a sequence between teams - across XCDs too - and its exits.

* A team in `local` mode skips the L2 write-back at its barriers (seq_kernel.h team_sync); a sequence whose previous scan ran
  on ANOTHER XCD is read through another L2, so the hand-over (sched_release / sched_pick) must publish everything at agent
  scope.  The tests force such migrations (an XCD with more sequences than teams beside XCDs that run out of work), count them
  (ptl_batch_sched_counters) and compare EVERY sequence bit for bit with its run alone - and with the same batch under
  PTL_TEAM_SYNC=agent (no XCD-local shortcut at all).
* No exit of the persistent kernel is silent: a workgroup that dies between two scans - the leader or a teammate -
  is reported by ptl_batch_wait, the sequence it held is named, the others finish."""
import os
import time

import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, synth

pytestmark = pytest.mark.gpu


def _load(b, seqs, n, n_imu):
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])


def _single(sq, n, n_imu, team):
    r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=team, gn_lanes_per_point=8, gn_threads=512)
    for k in range(n):
        r.upload_scan(k, sq.scan(k))
    r.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    r.run()
    out = r.results()
    r.close()
    return out


def _same(a, b):
    return (np.array_equal(a["kiss_poses"], b["kiss_poses"]) and np.array_equal(a["res_poses"], b["res_poses"])
            and np.array_equal(a["res_t"], b["res_t"]) and a["stats"] == b["stats"])


def test_sequences_that_migrate_between_xcds_equal_their_runs_alone():
    """17 sequences on a grid of 32 workgroups in teams of 2: XCD 0 hosts three sequences on its two teams, the other XCDs two
    on two - they run out of work while XCD 0 still has a sequence waiting, and take it.  13 sweeps in launches of 4.  Every
    sequence (not a sample) equals its run alone with two workgroups; migrations did happen; the agent-scope build of the same
    barriers (PTL_TEAM_SYNC=agent) gives the same bits."""
    S, n, team = 17, 13, 2
    seqs = [synth.make_sequence(seed=1800 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    kw = dict(use_imu_prediction=True, with_ekf=True, team_workgroups=team, scans_per_launch=4, gn_workgroups=32)
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, **kw)
    _load(b, seqs, n, n_imu)
    b.run()
    assert b.status() == 0
    outs = [b.results(s) for s in range(S)]
    cnt = [b.sched_counters(s) for s in range(S)]
    stolen, moved = sum(c["stolen"] for c in cnt), sum(c["cross_xcd_handovers"] for c in cnt)
    assert stolen > 0 and moved > 0, cnt
    assert all(0 <= c["last_xcc"] < 8 for c in cnt)
    # a second run of the same handle: same bits (the schedule differs from run to run, the results must not)
    b.run()
    for s in range(S):
        assert _same(b.results(s), outs[s]), s
    b.close()
    for s in range(S):
        assert _same(outs[s], _single(seqs[s], n, n_imu, team)), (s, cnt[s])
    old = os.environ.get("PTL_TEAM_SYNC")
    os.environ["PTL_TEAM_SYNC"] = "agent"
    try:
        b2 = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, **kw)
    finally:
        if old is None:
            del os.environ["PTL_TEAM_SYNC"]
        else:
            os.environ["PTL_TEAM_SYNC"] = old
    _load(b2, seqs, n, n_imu)
    b2.run()
    for s in range(S):
        assert _same(b2.results(s), outs[s]), s
    b2.close()
    # round 6: a team helps another XCD as soon as that XCD's least-advanced free sequence is a scan behind its own (PTL_SCHED_MARGIN,
    # default 1).  The policy of rounds 3-5 (-1: only when the own XCD has nothing left) must give the same bits.
    old = os.environ.get("PTL_SCHED_MARGIN")
    os.environ["PTL_SCHED_MARGIN"] = "-1"
    try:
        b3 = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, **kw)
    finally:
        if old is None:
            del os.environ["PTL_SCHED_MARGIN"]
        else:
            os.environ["PTL_SCHED_MARGIN"] = old
    _load(b3, seqs, n, n_imu)
    b3.run()
    assert b3.status() == 0
    for s in range(S):
        assert _same(b3.results(s), outs[s]), s
    stolen3 = sum(b3.sched_counters(s)["stolen"] for s in range(S))
    assert stolen3 > 0, (stolen3, stolen)  # (XCD 0's third sequence is still taken by the others once they have run out of work)
    b3.close()


def test_default_geometry_counts_its_migrations_and_stays_bit_exact():
    """the bench's geometry in small: 40 sequences on the full grid in teams of 2 (16 teams per XCD, five sequences each: every team
    busy, sequences change teams all the time inside their XCD and at the end across XCDs); ten of them against their runs alone"""
    S, n, team = 40, 8, 2
    seqs = [synth.make_sequence(seed=1900 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=team, scans_per_launch=3)
    _load(b, seqs, n, n_imu)
    b.run()
    assert b.status() == 0
    for s in range(0, S, 4):
        assert _same(b.results(s), _single(seqs[s], n, n_imu, team)), (s, b.sched_counters(s))


@pytest.mark.parametrize("who", ["teammate", "leader"])
def test_a_workgroup_that_dies_between_two_scans_is_reported(who):
    """four sequences on a grid of 32 workgroups (one team of 2 per XCD).  A workgroup of XCD 1's team returns right before the
    job barrier of its second job (test hook).  teammate: the leader holds sequence 1 and waits - its poll budget runs out, the
    sequence gets the time-out flag and goes off the schedule.  leader: the sequence stays marked busy with scans left - the
    check after the launch flags it.  Either way ptl_batch_wait names sequence 1, nothing hangs, the status word says why the
    teams left, and the other sequences ran all their scans."""
    S, n = 4, 4
    seqs = [synth.make_sequence(seed=1950 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=2, gn_workgroups=32)
    _load(b, seqs, n, n_imu)
    b.run(1)
    assert b.status() == 0
    b.debug_stall_block(1 + 8 * (1 if who == "teammate" else 0), 1)  # block x + 8 j: XCD 1, workgroup j of its team
    t0 = time.perf_counter()
    b.enqueue(n - 1)
    with pytest.raises(RuntimeError, match=r"sequence 1.*0x10"):
        b.wait()
    assert time.perf_counter() - t0 < 180.0
    st = b.status()
    assert st & 2, st                      # a job barrier expired
    if who == "leader":
        assert st & 16, st                 # ... and the sequence the dead leader held never reached the last scan
    for s in (0, 2, 3):
        stats = b.results(s)["stats"]
        assert len(stats) == n and all(q["iterations"] > 0 for q in stats[1:]), s
    # the handle is usable again after a reset
    b.debug_stall_block(-1)
    b.run()
    assert b.status() == 0
    assert _same(b.results(1), _single(seqs[1], n, n_imu, 2))
