// declaration-only stand-in (see mujoco.h in this directory): <mujoco/mjmodel.h> is part of the same public header set
#include "mujoco.h"
