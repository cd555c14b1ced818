from .parallel import ParallelPlan, SeqParallel  # noqa: F401
