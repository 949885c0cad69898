"""MI355X-native GNS K-step hot path behind the reference's ``GNS`` module API.

    from opf_graph_neural_solver_amd import GNS, get_BLG      # instead of ``from main import GNS``

``GNS(latent_dim, hidden_dim, K, gamma, multiple_phi)`` keeps the constructor, ``state_dict`` keys and
``forward(buses, lines, generators, B, L, G) -> (v, theta, total_loss, last_loss)`` of
LeonOrou/OPF-Graph-Neural-Solver ``GNS/main.py:107-202``; the K-step loop runs in hand-written HIP kernels
(``csrc/``) through the C-ABI of ``include/gns_hip.h``.  There is no CPU fallback: without the built
library or without a ROCm device the forward raises.
"""
from .gns import GNS, LearningBlock, get_BLG, GNSError
from . import synth
from . import dist
from . import prepare, metrics, training
from .prepare import prepare_grids
from ._lib import load_library, library_path, set_option, get_option

__all__ = ['GNS', 'LearningBlock', 'get_BLG', 'GNSError', 'synth', 'dist', 'prepare', 'metrics', 'training', 'prepare_grids', 'load_library', 'library_path', 'set_option', 'get_option']
