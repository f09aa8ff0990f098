from .builder import ModelBuilder, mass_matrix, kinematics  # noqa: F401
from .tasks import REGISTRY, humanoid_interact, swimmer, quadrotor, linkage, welded, filter_arm, servo_arm, particle_task, acrobot, ball_chain, cartpole, cylinder_pile, humanoid_stand, humanoid_track, humanoid_walk, particle, quadruped, quadruped_hill, shadow_hand, terrain_balls, walker  # noqa: F401
