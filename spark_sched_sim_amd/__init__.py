"""MI355X-native batched Spark-scheduling simulator (drop-in for the reference's
`SparkSchedSimEnv.reset()/step()` hot path). See DESIGN.md."""
from .vec_env import BatchedObs, VecSparkSchedSimEnv  # noqa: F401

__all__ = ["VecSparkSchedSimEnv", "BatchedObs"]
