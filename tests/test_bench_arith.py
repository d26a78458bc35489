"""bench.py's rank / shard arithmetic and `value` formula, without a GPU."""
import bench


def test_strong_scaling_row_ranges_partition_the_total():
    for total in (10_000_000, 10, 7, 1):
        for world in (1, 2, 3, 4, 8):
            spans = [bench.shard_rows(total, r, world, strong=True) for r in range(world)]
            assert spans[0][0] == 0
            for (lo, n), (lo2, _) in zip(spans, spans[1:]):
                assert lo + n == lo2                                  # contiguous, in rank order
            assert spans[-1][0] + spans[-1][1] == total               # covers every row exactly once
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1


def test_weak_scaling_keeps_pairs_per_gpu():
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert bench.shard_rows(10_000_000, r, world, strong=False) == (r * 10_000_000, 10_000_000)


def test_value_is_whole_job_pairs_per_second():
    # 3 losses x 10 M pairs x 20 steps in 9.16 ms at N = 1 -> 65.5 G pairs/s (the r01 driver record)
    v1 = bench.job_value(10_000_000, 1, False, 20, 20 * 0.4583e-3)
    assert abs(v1 - 65459.3) < 1.0
    # weak scaling: 8 ranks, same step time -> 8x; strong scaling: the total stays 10 M pairs
    assert abs(bench.job_value(10_000_000, 8, False, 20, 20 * 0.4583e-3) - 8 * v1) < 1e-6 * v1
    assert abs(bench.job_value(10_000_000, 8, True, 20, 20 * 0.4583e-3) - v1) < 1e-6 * v1
