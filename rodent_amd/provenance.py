"""Which kernel sources a committed profile belongs to.

`bench.py` quotes counter values (VALU instructions per launch, HBM bytes per launch, per-kernel durations of the renderer)
from profiles committed under profiles/rNN_*.json.  Those numbers are only valid for the kernels they were measured on: the
scripts that write the profiles (scripts/pmc_digest.py, scripts/profile_digest.py, scripts/render_profile.py) store
`source_sha(kind)` in them and bench.py refuses every figure whose hash differs from the sources it is running on."""
from __future__ import annotations

import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
FILES = {
    "traversal": ["traversal.hip", "traversal_top.h", "traversal_wide.h", "traversal_device.h"],
    "render": ["render.hip", "shading.h", "traversal_device.h"],
}


def source_sha(kind: str) -> str:
    """sha256 (first 16 hex digits) over the sources of the product kernels of `kind` ("traversal" / "render")."""
    h = hashlib.sha256()
    for name in FILES[kind]:
        h.update(name.encode())
        h.update((CSRC / name).read_bytes())
    return h.hexdigest()[:16]


def stamp(kind: str) -> dict:
    return {"source_sha": source_sha(kind), "files": FILES[kind]}


def is_current(meta, kind: str) -> bool:
    return isinstance(meta, dict) and meta.get("source_sha") == source_sha(kind)
