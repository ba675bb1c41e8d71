"""Hash of the sources libmas_hip.so is built from (csrc/*.hpp, csrc/*.inc, csrc/*.hip, include/*.h).

tools/collect_profiles.sh stamps it into every profiles/*.json it writes; bench.py recomputes it and refuses to quote counter data
(HBM traffic, SQ utilisation) collected from other sources -- a stale profile shows up as `null` with the reason, not as a number."""
from __future__ import annotations

import hashlib
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files():
    csrc = os.path.join(_ROOT, "multi_agent_solver_amd", "csrc")
    inc = os.path.join(_ROOT, "include")
    return ([os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hpp", ".hip", ".inc"))] +
            [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")])


def source_hash() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
