"""CPU: every archived experiment patch (scripts/exp_*.patch) is reproducible -- scripts/README.md names the commit it applies to and it
does apply there (checked against a scratch index of that commit: no checkout, nothing in the work tree is touched).  Skipped where the
repository's history is not available (the GPU box gets a snapshot without .git)."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table():
    rows = {}
    for m in re.finditer(r"^\| `(exp_[a-z0-9_]+\.patch)` \| `([0-9a-f]{7,40}|HEAD)` \|", open(os.path.join(ROOT, "scripts", "README.md")).read(), re.M):
        rows[m.group(1)] = m.group(2)
    return rows


def test_every_patch_has_a_base_commit_row():
    have = {os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "scripts", "exp_*.patch"))}
    assert have and have == set(_table()), sorted(have ^ set(_table()))


@pytest.mark.parametrize("patch", sorted(_table()))
def test_patch_applies_to_its_base_commit(patch):
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("no git history here")
    base = _table()[patch]
    if subprocess.run(["git", "-C", ROOT, "cat-file", "-e", f"{base}^{{commit}}"], capture_output=True).returncode:
        pytest.skip(f"commit {base} is not in this clone")
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, GIT_INDEX_FILE=os.path.join(tmp, "index"))
        subprocess.check_call(["git", "-C", ROOT, "read-tree", base], env=env)
        r = subprocess.run(["git", "-C", ROOT, "apply", "--cached", "--check", os.path.join("scripts", patch)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, f"{patch} does not apply to {base}: {r.stderr[:300]}"
