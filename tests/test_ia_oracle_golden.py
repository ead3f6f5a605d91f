"""The CPU restatement of the reference's InitialAligner (oracle/ia_oracle.c) against fixtures made with the
compiled reference (oracle/gen_golden.py --ia)."""
import gzip
import json
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT


def ia_cases():
    with open(os.path.join(GOLDEN, "ia_cases.json")) as f:
        return json.load(f)["cases"]


def ia_file(name, kind) -> bytes:
    with gzip.open(os.path.join(GOLDEN, f"{name}.{kind}.gz"), "rb") as f:
        return f.read()


@pytest.mark.parametrize("case", ia_cases(), ids=[c["name"] for c in ia_cases()])
def test_ia_oracle_matches_reference_fixture(case, tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    t, r = tmp_path / "x_Template.fasta", tmp_path / "x_Seq.fasta"
    t.write_bytes(ia_file(case["name"], "template"))
    r.write_bytes(ia_file(case["name"], "reads"))
    msa, cls = tmp_path / "msa", tmp_path / "cls"
    p = subprocess.run([os.path.join(ROOT, "oracle", "ia_oracle"), str(t), str(r), str(msa), str(cls), str(case["cutoff"])])
    assert p.returncode == 0
    assert cls.read_text() == case["seqclass"]
    assert msa.read_bytes() == ia_file(case["name"], "msa")
