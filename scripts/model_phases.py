#!/usr/bin/env python
"""CPU model of phased traversal with compaction (k_bvh2_phase): from the oracle's per-ray step counts
(oracle.binding.ray_steps, saved as /tmp/steps_<set>.npy) computes, for caps K1, K2, ... on the wave iterations of each
phase, the total wave iterations (VALU issue proxy), the lane utilisation bound and the critical path in iterations.
usage: python scripts/model_phases.py /tmp/steps_primary.npy 48 64"""
import sys
import numpy as np

steps = np.load(sys.argv[1]).sum(1).astype(np.int64)
caps = [int(x) for x in sys.argv[2:]]


def run(remaining, caps):
    total_iters, crit, waves_per_phase, lane_steps = 0, 0, [], remaining.sum()
    for k, cap in enumerate(caps + [None]):
        n = len(remaining)
        if n == 0:
            break
        pad = (-n) % 64
        r = np.concatenate([remaining, np.zeros(pad, np.int64)]).reshape(-1, 64)
        wmax = r.max(1)
        it = wmax if cap is None else np.minimum(wmax, cap)
        total_iters += it.sum(); crit += it.max(); waves_per_phase.append((len(r), int(it.sum()), int(it.max())))
        if cap is None:
            break
        remaining = remaining[remaining > cap] - cap           # survivors, stream order kept
    return total_iters, crit, waves_per_phase, lane_steps


base = run(steps, [])
print(f"baseline: wave-iterations {base[0]}, critical path {base[1]} iterations, lane utilisation bound {base[3] / (base[0] * 64):.3f}")
import itertools
for caps_ in ([caps] if caps else [[k] for k in (16, 24, 32, 40, 48, 64, 80)] + [[a, b] for a in (24, 32, 40, 48) for b in (24, 32, 48,
    64)] + [[32, 32, 32], [24, 24, 24, 24], [32, 32, 32, 32], [40, 40, 40]]):
    t, c, w, ls = run(steps, caps_)
    print(f"caps {str(caps_):22s}: wave-iterations {t} ({t / base[0]:.3f} of baseline), lane util {ls / (t * 64):.3f}, critical path {c} "
        f"iterations, phases (waves, iterations, longest) {w}")
