"""DESIGN.md is the library's current state in a form a maintainer can read: at most 300 lines of at most 140
characters, and every file of this repository that it, README.md or INTEGRATION.md name (profiles/, tools/, tests/test_*, include/ ...) exists.  The log of what was tried is
EXPERIMENTS.md (no limits there)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_is_short_and_narrow():
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().split("\n")
    assert len(lines) <= 300, len(lines)
    long = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 140]
    assert not long, long


def test_files_named_in_the_documents_exist():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`((?:profiles|tools|tests/test_|tests/golden|include|oracle|examples)/?[A-Za-z0-9_./+-]+)`", text):
            path = m.group(1).rstrip(".")
            if "*" in path or path.endswith("/") or "..." in path:
                continue
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
    assert not missing, missing
