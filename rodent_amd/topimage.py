"""Which BVH2 nodes the default traversal kernel keeps in LDS (k_bvh2_top_image in csrc/traversal.hip), restated on the host
for the benchmark's accounting and the analysis scripts: the first `capacity` inner nodes in breadth-first order, slots
handed out level by level in rounds of 64 slots -- first children of a round before its second children."""
import numpy as np

DEFAULT_CAPACITY = 255      # TOPN of the default mapping "top"


def image_nodes(nodes, capacity=DEFAULT_CAPACITY):
    """0-based indices into `nodes` (NODE2 structured array) of the nodes in the image, in slot order."""
    child = np.asarray(nodes["child"]).reshape(-1, 2)
    slots = [0]
    begin, end = 0, 1
    while begin < end:
        nxt = end
        for first in range(begin, end, 64):
            ids = slots[first:min(first + 64, end)]
            for k in (0, 1):
                for i in ids:
                    c = int(child[i][k])
                    if c > 0:
                        if nxt < capacity and len(slots) < capacity:
                            slots.append(c - 1)
                        nxt += 1
            nxt = min(nxt, capacity)
        begin, end = end, min(nxt, len(slots))
    return np.asarray(slots, np.int64)
