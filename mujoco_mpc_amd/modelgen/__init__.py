from .builder import ModelBuilder, mass_matrix, kinematics  # noqa: F401
from .tasks import REGISTRY, cartpole, particle, quadruped  # noqa: F401
