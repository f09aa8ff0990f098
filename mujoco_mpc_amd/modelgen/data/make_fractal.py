"""Derives modelgen/data/quadruped_hill_terrain.npz from the reference's height-field image (run where /root/reference exists):
mjpc/tasks/quadruped/assets/fractal_noise.png (100 x 100 RGB) -> grey = mean of the channels, normalised to [0, 1], rows flipped so
that row 0 is y = -radius_y (MuJoCo reads image rows top to bottom as +y to -y); plus the stage goals of task_hill.xml:82-101."""
import os
import re

import numpy as np
from PIL import Image

REF = "/root/reference/mjpc/tasks/quadruped"
img = np.asarray(Image.open(os.path.join(REF, "assets", "fractal_noise.png")).convert("RGB"), float)
grey = img.mean(axis=2)
grey = (grey - grey.min()) / (grey.max() - grey.min())
data = grey[::-1].copy()
keys = []
for line in open(os.path.join(REF, "task_hill.xml")):
    m = re.search(r'<key .*mpos="([^"]+)" mquat="([^"]+)"', line)
    if m:
        keys.append([float(x) for x in m.group(1).split()] + [float(x) for x in m.group(2).split()])
home = re.search(r'<key name="home" qpos="([^"]+)"', open(os.path.join(REF, "task_hill.xml")).read()).group(1)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "quadruped_hill_terrain.npz"),
                    data=data, size=np.array([5.0, 5.0, 1.0, 2.0]), stages=np.array(keys), home=np.array([float(x) for x in home.split()]))
print(data.shape, len(keys), "stages; height at the start", data[50, 50])
