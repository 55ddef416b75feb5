"""Regenerates tests/golden/oracle_digests.json from the CPU oracle (run from the repo root)."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.join(HERE, "..", "oracle"), os.path.join(HERE, "..", "video-super-resolution-library_amd")]
import oracle_py as O  # noqa: E402
import synth  # noqa: E402
from common import CASES, folder, dtype_for, oracle_y  # noqa: E402

got = {}
for cid, fold, (rn, rd), bits, passes, mode, asm, full in CASES:
    for nm, fr in (("natural", synth.natural_y(96, 64, bits, seed=4242)), ("random", synth.random_y(96, 64, bits, seed=99))):
        out = oracle_y(fr, (cid, fold, (rn, rd), bits, passes, mode, asm, full))
        got[f"{cid}/{nm}"] = hashlib.sha256(out.tobytes()).hexdigest()
json.dump(got, open(os.path.join(HERE, "golden", "oracle_digests.json"), "w"), indent=1, sort_keys=True)
print(len(got), "digests written")
