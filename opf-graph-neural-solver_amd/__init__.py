"""MI355X-native GNS K-step hot path behind the reference's ``GNS`` module API.

    from opf_graph_neural_solver_amd import GNS, get_BLG      # instead of ``from main import GNS``

``GNS(latent_dim, hidden_dim, K, gamma, multiple_phi)`` keeps the constructor, ``state_dict`` keys and
``forward(buses, lines, generators, B, L, G) -> (v, theta, total_loss, last_loss)`` of
LeonOrou/OPF-Graph-Neural-Solver ``GNS/main.py:107-202``; the K-step loop runs in hand-written HIP kernels
(``csrc/``) through the C-ABI of ``include/gns_hip.h``.  There is no CPU fallback: without the built
library or without a ROCm device the forward raises.
"""
import os as _os

# HIP graphs (training.GraphedStep): ROCm 7 replays an instantiated graph from pre-built AQL packets by default.  On that path a replay of
# the captured training step that follows a host-side wait on the stream (hipStreamSynchronize / hipDeviceSynchronize: any
# ``torch.cuda.synchronize()``) was measured to compute a wrong gradient - forward loss right, parameters NaN a few steps later - while
# the same graph replayed node by node is bit-exact (tools/gpu_graph_replay_stream.py, profiles/r03/graph_replay_packet_capture.txt).
# The switch is read when the HIP runtime initialises, so it is set here, before this package makes its first GPU call; a caller's own
# setting wins, and GraphedStep verifies its first replays against eager steps whatever the setting.
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

from .gns import GNS, LearningBlock, get_BLG, GNSError
from . import synth
from . import dist
from . import prepare, metrics, training
from .prepare import prepare_grids
from ._lib import load_library, library_path, set_option, get_option

__all__ = ['GNS', 'LearningBlock', 'get_BLG', 'GNSError', 'synth', 'dist', 'prepare', 'metrics', 'training', 'prepare_grids', 'load_library', 'library_path', 'set_option', 'get_option']
