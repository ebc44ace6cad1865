"""MI355X-native batched radar-jamming environment + QMix/MP-DQN trainer (hot path only).

Directory name follows the build contract (``ma-cjd-cooperative-jamming-decision-making-via-marl_amd``);
it is not a valid Python identifier, so import it through the root-level alias module ``macjd_amd``
(``import macjd_amd``), which loads this directory as the package ``macjd_amd``.

Sub-packages mirror the reference's module layout for the hot path so that the reference's
``main.py`` can switch by changing its imports only:

    macjd_amd.simulation.environment   ElectromagneticEnvironment, BatchedElectromagneticEnvironment
    macjd_amd.core.networks            RNNAgent, QMixer
    macjd_amd.core.mac                 BasicMAC
    macjd_amd.core.qmix                QMixLearner
    macjd_amd.utils.action_selectors   EpsilonGreedyActionSelector
    macjd_amd.utils.replay_buffer      EpisodeReplayBuffer
    macjd_amd.runners.episode_runner   EpisodeRunner, BatchedEpisodeRunner
"""
__version__ = "0.1.0"

# One of the GPU's four hardware queues is set aside for HIP-graph launches (hipgraph.py); this has to
# happen before the HIP runtime initialises, i.e. before the first torch.cuda call of the process.
from . import hipgraph as _hipgraph  # noqa: E402

_hipgraph.reserve_launch_queue()
