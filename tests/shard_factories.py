"""Engine factories for the shard workers of ``shard_front.ShardLeader`` in the CPU tier (named through ISCC_HIP_SHARD_ENGINE_FACTORY)."""


def oracle(local_rank):
    """Every rank runs the oracle-backed stand-in engine of the CPU tests: (local engine, ops factory, device)."""
    from oracle_engine import OracleEngine
    from test_sharded_gloo import OracleShardOps

    return OracleEngine(), OracleShardOps, None
