"""The keyed stream's draws beyond a segment's Philox block (one xoshiro128 engine step per rejection
attempt, oracle/rt_philox.h): the points random_in_unit_sphere (vec3.h:130-136) accepts from them are
uniform in the ball, like the ones accepted from the block's own words."""
import numpy as np


def test_continuation_draws_fill_the_ball_uniformly(oracle):
    n = 60000
    pts, used = np.zeros((n, 3)), np.zeros(n, int)
    for k in range(n):
        pts[k], used[k] = oracle.sample("in_unit_sphere", 2024, k, 3, 1 + k % 5)
    cont = used > 3
    assert abs(cont.mean() - (1 - np.pi / 6)) < 4 * np.sqrt(0.25 / n)              # rejection rate of the first attempt
    assert (used % 3 == 0).all() and used.max() >= 18
    for sel in (~cont, cont):
        q = pts[sel]
        m = len(q)
        r2 = (q ** 2).sum(1)
        assert (r2 < 1).all()
        assert np.abs(q.mean(0)).max() < 4.5 * np.sqrt(0.2 / m)                     # var of a coordinate is 1/5
        assert np.abs((q ** 2).mean(0) - 0.2).max() < 4.5 * np.sqrt(0.0229 / m)      # var of x^2 is 3/35 - 1/25
        octant = (q[:, 0] > 0) * 4 + (q[:, 1] > 0) * 2 + (q[:, 2] > 0)
        cnt = np.bincount(octant, minlength=8)
        assert ((cnt - m / 8) ** 2 / (m / 8)).sum() < 26.0                         # chi-square, 7 dof, p ~ 5e-4
        u = np.sort(r2 ** 1.5)                                                     # r^3 is uniform
        assert np.abs(u - (np.arange(m) + 0.5) / m).max() * np.sqrt(m) < 2.0       # Kolmogorov-Smirnov
        assert np.abs(np.corrcoef(q.T)[np.triu_indices(3, 1)]).max() < 4.5 / np.sqrt(m)


def test_stream_is_a_pure_function_of_its_key(oracle):
    a = [oracle.draw(5, 17, 3, 2, d) for d in range(12)]
    assert a == [oracle.draw(5, 17, 3, 2, d) for d in range(12)]
    assert a != [oracle.draw(5, 17, 3, 3, d) for d in range(12)]
    # segment 0 serves four words of its block directly, a scatter segment three
    blk = oracle.philox((17, 3, 0, 0), (5, 0))
    assert [oracle.draw(5, 17, 3, 0, d) for d in range(4)] == [w >> 8 for w in blk]
    blk = oracle.philox((17, 3, 2, 0), (5, 0))
    assert a[:3] == [w >> 8 for w in blk[:3]] and a[3] != blk[3] >> 8
