"""CPU: the bookkeeping of stage 2 of the two-stage Hessenberg-triangular reduction (csrc/ht_twostage.hip,
`ht_two_stage_device`) restated: wavefront tau holds the sweeps j with position t = tau - LAG j on the diagonal; the
reflectors of a group of GS sweeps live in slot (group mod nslot) until the group's last wavefront, when they go to
Q and Z.  For every n: each step is visited exactly once, jlo / jhi bracket exactly the live sweeps, and a slot is
never claimed by a new group before the group that held it has been closed (`ht2_nslot`)."""
import pytest

R2, GS, LAG, MAXSLOT, EXTRASLOT = 64, 64, 2, 256, 40


def tstride(n):
    return (n - 3) // R2 + 1


def need_slots(n):
    """what the chase itself needs (`ht2_need_slots`); the store holds EXTRASLOT more so that the chase does not wait for the
    side stream that applies a closed group to Q and Z (`ht2_nslot`)"""
    return (tstride(n) + LAG * GS - LAG) // (LAG * GS) + 2


def nslot(n):
    return min(MAXSLOT, need_slots(n) + EXTRASLOT)


def last_wave(n, g):
    jl = min(g * GS + GS - 1, n - 3)
    return LAG * jl + (n - 3 - jl) // R2


@pytest.mark.parametrize("n", [3, 4, 65, 66, 67, 130, 200, 1500, 1601, 4163, 12000, 20011])
def test_wavefronts_cover_every_step_once_and_slots_are_free_when_claimed(n):
    r = R2
    ngroups = (n - 2 + GS - 1) // GS
    opened = closed = 0
    steps = 0
    max_count = 0
    tau = 0
    while True:
        jhi = min(tau // LAG, n - 3)
        num = tau * r - (n - 3)
        jlo = 0 if num <= 0 else (num + (LAG * r - 1) - 1) // (LAG * r - 1)
        if jlo > jhi:
            if tau // LAG >= n - 3:
                break
            tau += 1
            continue
        # exactly the sweeps with a position on the diagonal
        live = [j for j in range(max(0, jlo - 2), min(n - 3, jhi + 2) + 1)
                if tau - LAG * j >= 0 and j + 1 + (tau - LAG * j) * r <= n - 2]
        assert live == list(range(jlo, jhi + 1)), (n, tau, jlo, jhi, live)
        steps += len(live)
        max_count = max(max_count, len(live))
        while opened <= jhi // GS:
            assert opened - closed < need_slots(n) <= nslot(n), (n, tau, opened, closed)      # its slot's previous group is closed
            opened += 1
        while closed < ngroups and last_wave(n, closed) <= tau:
            assert closed < opened
            closed += 1
        tau += 1
    assert closed == ngroups or all(last_wave(n, g) >= tau for g in range(closed, ngroups))
    # every (sweep, position) with p = j + 1 + t r <= n - 2
    assert steps == sum((n - 3 - j) // r + 1 for j in range(n - 2))
    assert max_count <= n // (LAG * R2 - 1) + 4                              # the workspace's maxcount
