"""Pins the CPU restatement (oracle/pw_oracle.c) to the compiled reference: every fixture under
tests/golden/ was produced by oracle/_ref/pw_ref (see oracle/gen_golden.py); the restatement must
reproduce the output file byte for byte, the exit code, and every score line."""
import os

import pytest

from conftest import GOLDEN, golden_cases, golden_input, golden_output

CASES = golden_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_port_matches_reference(case, oracle, tmp_path):
    rc, out, lines = oracle.run_cli(golden_input(case["name"]), case["bandwidth"], str(tmp_path))
    assert rc == case["exit_code"]
    exp = golden_output(case["name"])
    assert (out is not None) == case["wrote_output"]
    assert out == exp
    got = [l for l in lines if l.startswith(("OverallScore", "Rows ", "bandwidth"))]
    assert got == case["stdout"]


def test_fullscale_rounds_fixture_is_a_chain():
    """tests/golden/tree_default_rounds.json (oracle/gen_fullscale.py): every round's record is either the sequential
    reference run's or a link whose input digest is the digest of the round before -- and where both exist they agree."""
    import json
    with open(os.path.join(GOLDEN, "tree_default_rounds.json")) as f:
        fx = json.load(f)
    assert fx["input_sha256"] == "74e4e7db407afe9d0db10af5bf0afe7a6051dcea856ebdc17c620cac15cdf01e" and fx["bandwidth"] == 1000
    rounds = fx["rounds"]
    assert len(rounds) >= 2
    links = {l["round"]: l for l in fx["chained"]["links"]}
    seq = fx["sequential"]["rounds"]
    prev_sha, prev_score = fx["input_sha256"], None
    for i, r in enumerate(rounds):
        assert r["round"] == i + 1
        if prev_score is not None:
            assert (r["score"] < prev_score) == r["improved"]
        if i < len(seq):
            assert seq[i]["score"] == r["score"] and seq[i].get("output_sha256") == r.get("output_sha256")
        if r["round"] in links and links[r["round"]]["input_sha256"] == prev_sha:
            assert links[r["round"]]["score"] == r["score"] and links[r["round"]].get("output_sha256") == r.get("output_sha256")
        else:
            assert i < len(seq), "a round that neither the sequential run nor an anchored link vouches for"
        if r["improved"]:
            assert r["rows"] == fx["input_rows"] and len(r["output_sha256"]) == 64
            prev_sha, prev_score = r["output_sha256"], r["score"]
        else:
            assert i == len(rounds) - 1 and fx["converged"]
