"""Operator surface of tf_euler.python.euler_ops for the hot path."""
from .base import *          # noqa: F401,F403
from .type_ops import *      # noqa: F401,F403
from .sample_ops import *    # noqa: F401,F403
from .neighbor_ops import *  # noqa: F401,F403
from .walk_ops import *      # noqa: F401,F403
from .mp_ops import *        # noqa: F401,F403
from .feature_ops import *   # noqa: F401,F403
