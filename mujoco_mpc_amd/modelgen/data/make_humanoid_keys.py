"""Builds humanoid_motion_keys.npz from the reference's key-frame DATA files (mjpc/tasks/humanoid/tracking/keyframes/*.xml, CMU
mocap retargeted to the humanoid): the mocap-body positions of every key of the ten motions, in the order of the motion table
of tracking.cc:43-54 (Jump, Kick Spin, Spin Kick, Cartwheel (1), Crouch Flip, Cartwheel (2), Monkey Flip, Dance, Run, Walk), plus
the qpos / qvel of each motion's first key (the state Transition resets to on a motion switch, tracking.cc:231-238).
Run in the build container only (it reads /root/reference); the .npz is what ships.  Data, not code, is taken from the reference."""
import os
import sys
import xml.etree.ElementTree as ET

import numpy as np

REF = "/root/reference/mjpc/tasks/humanoid/tracking/keyframes"
FILES = ["CMU-CMU-02-02_04", "CMU-CMU-87-87_01", "CMU-CMU-88-88_06", "CMU-CMU-88-88_07", "CMU-CMU-88-88_08", "CMU-CMU-88-88_09",
         "CMU-CMU-90-90_19", "CMU-CMU-103-103_08", "CMU-CMU-108-108_13", "CMU-CMU-137-137_40"]
LENGTHS = [121, 154, 115, 78, 145, 188, 260, 279, 39, 510]          # kMotionLengths, tracking.cc:43-54

mpos, q0, v0, names = [], [], [], []
for f, n in zip(FILES, LENGTHS):
    keys = ET.parse(os.path.join(REF, f + "_poses.xml")).getroot().find("keyframe").findall("key")
    assert len(keys) == n, (f, len(keys), n)
    for k in keys:
        mpos.append(np.array(k.get("mpos").split(), float))
    q0.append(np.array(keys[0].get("qpos").split(), float))
    v0.append(np.array(keys[0].get("qvel").split(), float) if keys[0].get("qvel") else np.zeros(27))
    names.append(keys[0].get("name").rsplit("_", 1)[0])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "humanoid_motion_keys.npz")
np.savez_compressed(out, mpos=np.array(mpos), qpos0=np.array(q0), qvel0=np.array(v0), lengths=np.array(LENGTHS), names=np.array(names))
print(out, np.array(mpos).shape, names, os.path.getsize(out))
old = os.path.join(os.path.dirname(out), "humanoid_jump_keys.npz")
if os.path.exists(old):
    d = np.load(old)
    print("jump identical to the round-1 file:", np.array_equal(d["mpos"], np.array(mpos)[:121]), np.array_equal(d["qpos0"], q0[0]), np.array_equal(d["qvel0"], v0[0]))
