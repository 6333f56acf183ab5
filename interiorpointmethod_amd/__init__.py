"""interiorpointmethod_amd -- MI355X-native Newton/KKT hot path of a Mehrotra interior-point
LP solver (drop-in for the loop of payakorn/InteriorPointMethod)."""
from ._lib import IpmError, IpmLibraryError, LIB_PATH, load as load_library  # noqa: F401
from .general_form import add_bound_into_matrix, get_Abc, new_interior_sparse  # noqa: F401
from .solver import (IpmSolver, LockstepBatch, direction_corrected, direction_corrected_sparse, direction_predicted,  # noqa: F401
                     direction_predicted_sparse,
                     interior, interior_sparse, last_info, lockstep_eligible, solve, solve_linear, solve_lockstep, solve_with_info)

__all__ = ["IpmSolver", "solve", "solve_with_info", "interior", "interior_sparse",
           "direction_predicted_sparse", "direction_corrected_sparse", "direction_predicted", "direction_corrected", "solve_linear", "last_info",
           "solve_lockstep", "LockstepBatch", "lockstep_eligible", "new_interior_sparse", "get_Abc", "add_bound_into_matrix",
           "IpmError", "IpmLibraryError", "load_library", "LIB_PATH"]
