#!/usr/bin/env python
"""How much of the schedule history's gain survives when the rays CHANGE from launch to launch (rodent_hip_schedule_history: chunks traced
longest first by the previous launch's per-chunk cost): 1 Mi primary rays of the atrium from a camera that turns by a fixed angle (and moves
along its view direction) every frame, history off / on; every frame's hits are compared with the history-off run of the same frame. usage:
python scripts/history_experiment.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from rodent_amd import abi, raygen, scenes

path = scenes.scene_bvh("atrium")
bvh = abi.DeviceBvh.load(path, 2, 0)
eye, d, up, fov = scenes.CAMERAS["atrium"]
eye, d = np.asarray(eye, np.float64), np.asarray(d, np.float64)
FRAMES = 12


def frame_rays(step_deg, k):
    a = np.radians(step_deg * k)
    rot = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])          # yaw about the up axis (y)
    dk = rot @ d
    return raygen.primary_rays(eye + dk / np.linalg.norm(dk) * (5.0 * step_deg * k), dk, up, fov, 1024, 1024, 0.0, 5000.0)


st = torch.cuda.current_stream()
print(f"{'camera step per frame':>24s} {'history off ms':>15s} {'history on ms':>14s} {'gain':>7s}  identical hits")
for step in (0.0, 0.1, 0.5, 1.0, 2.0, 5.0, 15.0):
    frames = [abi.to_device(frame_rays(step, k), 0) for k in range(FRAMES)]
    n = 1 << 20
    out = {}
    for mode in (0, 1):
        abi.lib().rodent_hip_schedule_history(mode)
        hits = [torch.zeros(n * 16, dtype=torch.uint8, device="cuda:0") for _ in range(FRAMES)]
        for _ in range(2):                                   # warm-up: the same sequence of frames
            for k in range(FRAMES):
                abi.traverse_async(bvh, frames[k], hits[k], n, False, 0, st)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(FRAMES)]
        for k in range(FRAMES):
            ev[k][0].record(st); abi.traverse_async(bvh, frames[k], hits[k], n, False, 0, st); ev[k][1].record(st)
        torch.cuda.synchronize()
        # frame 0 follows the LAST frame of the warm-up: skipped
        out[mode] = (float(np.median([s.elapsed_time(e) for s, e in ev[1:]])), hits)
    abi.lib().rodent_hip_schedule_history(0)
    same = all(torch.equal(a, b) for a, b in zip(out[0][1], out[1][1]))
    print(f"{step:21.1f} deg {out[0][0]:15.4f} {out[1][0]:14.4f} {out[0][0] / out[1][0]:6.3f}x  {same}")
