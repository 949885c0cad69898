"""Import shim: the package directory is named ``opf-graph-neural-solver_amd`` (not a legal Python
identifier), so it is loaded from its path and registered under ``opf_graph_neural_solver_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'opf-graph-neural-solver_amd')
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
