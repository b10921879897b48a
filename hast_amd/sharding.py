"""Read sharding across ranks: step s of rank r owns reads [(s*world + r)*R, +R).
Counters are additive integers, so shards merge with one all_reduce(sum) whatever the shard order."""


def shard_first_read(step: int, world: int, rank: int, batch_reads: int) -> int:
    return (step * world + rank) * batch_reads


def job_reads(steps: int, world: int, batch_reads: int) -> int:
    return steps * world * batch_reads
