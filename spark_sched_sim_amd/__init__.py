"""MI355X-native batched Spark-scheduling simulator (drop-in for the reference's
`SparkSchedSimEnv.reset()/step()` hot path). See DESIGN.md."""
from . import metrics  # noqa: F401
from .env import SparkSchedSimEnv  # noqa: F401
from .schedulers import RandomScheduler, RoundRobinScheduler, Scheduler, make_scheduler  # noqa: F401
from .vec_env import BatchedObs, VecSparkSchedSimEnv  # noqa: F401

__all__ = ["VecSparkSchedSimEnv", "BatchedObs", "SparkSchedSimEnv", "Scheduler", "RoundRobinScheduler",
           "RandomScheduler", "make_scheduler", "metrics"]
