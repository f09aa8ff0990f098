from .builder import ModelBuilder, mass_matrix, kinematics  # noqa: F401
from .tasks import REGISTRY, cartpole, humanoid_track, particle, quadruped  # noqa: F401
