"""DESIGN.md section 0 carries a table of the round's numbers that is WRITTEN by tools/design_numbers.py from the committed
artefacts under profiles/ (the text had drifted from its own files).  This test fails when the block in DESIGN.md
is not what the artefacts say -- re-run `python tools/design_numbers.py --write` after replacing an artefact."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_are_the_artefacts_numbers():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr + p.stdout


def test_every_cited_artefact_of_the_block_exists():
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    a, b = s.index("<!-- numbers:begin"), s.index("<!-- numbers:end -->")
    block = s[a:b]
    assert "(artefact missing)" not in block, "an artefact the round's numbers come from is not committed"
