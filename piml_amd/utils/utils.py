"""Hot-path members of the reference's utils.utils (src/utils/utils.py)."""
from .. import ops


def calc_acceleration(relative_data, equation_version='v0', dataset='gc1560', eps=1e-6):
    """Pair acceleration label on gathered neighbours (utils.py:31-100); same arguments."""
    return ops.calc_acceleration(relative_data, equation_version, dataset, eps)
