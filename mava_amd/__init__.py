"""mava_amd - MI355X-native PPO hot path behind Mava's LearnerFn contract.

Only the rollout -> GAE -> minibatch-PPO loop of mava/systems/ppo is built here;
all device arithmetic runs in hand-written gfx950 kernels (mava_amd/csrc, libmavahip.so).
"""
__version__ = "0.1.0"
