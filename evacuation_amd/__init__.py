"""evacuation_amd -- MI355X-native step path of the cinemere/evacuation environment.

Public surface = the reference's ``src.env`` exports (src/env/__init__.py:3-21):
``setup_env, EvacuationEnv, EnvConfig, EnvWrappersConfig, Status`` plus the batched form
``BatchedEvacuationEnv``, its SyncVectorEnv-shaped host face ``HostVectorEnv``, the sharded form ``ShardedEvacuationEnv`` (across GPUs) and ``SplitBatchEnv`` (across streams of one GPU).  Importing the package does
not touch the GPU; constructing an env loads libevac.so and fails loudly without it."""
from .config import EnvConfig, EnvWrappersConfig
from .statuses import Status

__all__ = ["EnvConfig", "EnvWrappersConfig", "Status", "setup_env", "EvacuationEnv", "BatchedEvacuationEnv",
           "ShardedEvacuationEnv", "SplitBatchEnv", "NormalizedVectorEnv", "HostVectorEnv", "RandomAgent", "KernelOptions", "kernel_options"]


def __getattr__(name):   # lazy: keeps `import evacuation_amd` light and torch-free for config users
    if name in ("setup_env", "EvacuationEnv"):
        from . import env as _env
        return getattr(_env, name)
    if name == "BatchedEvacuationEnv":
        from .vector_env import BatchedEvacuationEnv
        return BatchedEvacuationEnv
    if name == "ShardedEvacuationEnv":
        from .distributed import ShardedEvacuationEnv
        return ShardedEvacuationEnv
    if name == "SplitBatchEnv":
        from .split_env import SplitBatchEnv
        return SplitBatchEnv
    if name == "NormalizedVectorEnv":
        from .wrappers import NormalizedVectorEnv
        return NormalizedVectorEnv
    if name == "HostVectorEnv":
        from .host_env import HostVectorEnv
        return HostVectorEnv
    if name in ("KernelOptions", "kernel_options"):
        from . import options as _options
        return getattr(_options, name)
    if name == "RandomAgent":
        from .agents import RandomAgent
        return RandomAgent
    raise AttributeError(name)
