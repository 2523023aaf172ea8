"""MI355X-native SAC/TD3 update engine behind the reference's `Agent` surface.

Import name: ``sac_td3_cudagraphs_pytorch_amd`` (the directory carries the repository's hyphenated
name; ``/sac_td3_cudagraphs_pytorch_amd.py`` at the repo root aliases it).

  Engine        thin object wrapper over the C ABI of libsactd3_hip.so (include/sactd3.h)
  Agent         drop-in for the reference's agents/agent.py:Agent (update_qnets / update_actor /
                update_targ_nets / predict / rb / counters)
  ReplayBuffer  drop-in for TensorDictReplayBuffer(LazyTensorStorage(...)) (extend / sample / len)
"""
from ._lib import EngineError, build_library, library_path, load_library  # noqa: F401
from .engine import Config, Engine  # noqa: F401
from .agent import Agent, BatchHandle, ReplayBuffer, StaleBatchError  # noqa: F401
from . import launcher, loop  # noqa: F401
