"""Pins the CPU restatement (oracle/pw_oracle.c) to the compiled reference: every fixture under
tests/golden/ was produced by oracle/_ref/pw_ref (see oracle/gen_golden.py); the restatement must
reproduce the output file byte for byte, the exit code, and every score line."""
import pytest

from conftest import golden_cases, golden_input, golden_output

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
