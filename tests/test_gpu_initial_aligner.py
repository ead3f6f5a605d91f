"""-m gpu: the HIP InitialAligner (include/pia.h) against the reference's fixtures (tests/golden/ia_*, made
with the compiled reference) and against the CPU restatement (oracle/ia_oracle.c) on seeded inputs: every placement and
every distance must be identical -- the bit-vector kernel has to reproduce the reference's tie order (diagonal, then
left only if strictly smaller, then up, IA:308-322) and its entry rule (IA:333-345)."""
import ctypes
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ia_oracle():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libiaoracle.so"))
    lib.iao_align.restype = ctypes.c_long
    lib.iao_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                              ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]

    def align(read, templ):
        a = (ctypes.c_int * max(len(read), 1))()
        e = ctypes.c_int()
        codes = ctypes.create_string_buffer(max(len(read) * len(templ), 1))
        d = lib.iao_align(read, len(read), templ, len(templ), a, ctypes.byref(e), codes)
        return list(a[:len(read)]), int(d)
    return align


def _mutate(rng, seq, rate):
    out = bytearray()
    for c in seq:
        u = rng.random()
        if u < rate / 3:
            continue
        if u < 2 * rate / 3:
            out.append(rng.choice(list(b"acgt")))
        elif u < rate:
            out.append(c)
            out.append(rng.choice(list(b"acgt")))
        else:
            out.append(c)
    return bytes(out)


def _check(templ, reads, ia_oracle, mem_budget=None):
    from repeatresolver_amd.initial_aligner import InitialAligner
    g = InitialAligner(templ)
    if mem_budget is not None:
        g.set_option("mem_budget", mem_budget)
    got, dist = g.align(reads)
    st = g.stats()
    g.close()
    assert st["cells"] == sum(len(r) for r in reads) * len(templ)
    for j, r in enumerate(reads):
        exp, d = ia_oracle(r, templ)
        assert int(dist[j]) == d, (j, len(r), len(templ))
        assert list(got[j]) == exp, (j, len(r), len(templ))


def ia_cases():
    with open(os.path.join(GOLDEN, "ia_cases.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", ia_cases(), ids=[c["name"] for c in ia_cases()])
def test_cli_reproduces_reference_files(case, tmp_path):
    from repeatresolver_amd.initial_aligner import run_files
    t, r = tmp_path / "x_Template.fasta", tmp_path / "x_Seq.fasta"
    for kind, path in (("template", t), ("reads", r)):
        with gzip.open(os.path.join(GOLDEN, f"{case['name']}.{kind}.gz"), "rb") as f:
            path.write_bytes(f.read())
    msa, cls = tmp_path / "msa", tmp_path / "cls"
    rc, lines = run_files(t, r, msa, cls, cutoff=case["cutoff"])
    assert rc == 0, lines
    assert lines[-2:] == ["", "Files written."] and any(l.startswith("template length ") for l in lines)
    assert cls.read_text() == case["seqclass"]
    with gzip.open(os.path.join(GOLDEN, f"{case['name']}.msa.gz"), "rb") as f:
        assert msa.read_bytes() == f.read()


def test_default_output_names_follow_the_template_path(tmp_path):
    """IA:676-700: <prefix>Template.fasta -> <prefix>MSA and <prefix>SeqClass, default cut-off 0.30"""
    case = ia_cases()[0]
    t, r = tmp_path / "Sim_Template.fasta", tmp_path / "Sim_Seq.fasta"
    for kind, path in (("template", t), ("reads", r)):
        with gzip.open(os.path.join(GOLDEN, f"{case['name']}.{kind}.gz"), "rb") as f:
            path.write_bytes(f.read())
    cli = os.path.join(ROOT, "repeatresolver_amd", "csrc", "InitialAligner")
    p = subprocess.run([cli, str(t), str(r), "-p", "4"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout
    assert (tmp_path / "Sim_MSA").exists() and (tmp_path / "Sim_SeqClass").exists()
    if case["cutoff"] == 0.30:
        with gzip.open(os.path.join(GOLDEN, f"{case['name']}.msa.gz"), "rb") as f:
            assert (tmp_path / "Sim_MSA").read_bytes() == f.read()


@pytest.mark.parametrize("L2", [1, 2, 31, 32, 33, 64, 2047, 2048, 2049, 4100])
def test_lane_and_word_boundaries(L2, ia_oracle):
    """template lengths around the 32-column words and the 64 x 32 columns one lane step covers; reads of 0, 1, 63, 64,
    65 ... bases (the 64 rows in flight per wave), small alphabets (many ties), unrelated reads (wide bands)"""
    import random
    rng = random.Random(1000 + L2)
    reads = []
    for alpha in (b"a", b"ac", b"acgt"):
        templ_a = bytes(rng.choice(list(alpha)) for _ in range(L2))
        rs = [b"", b"a", b"t", templ_a[:63], templ_a[-64:], templ_a[L2 // 3:L2 // 3 + 65], templ_a,
              _mutate(rng, templ_a[L2 // 4:L2 // 4 + 300], 0.2), _mutate(rng, templ_a, 0.1)[:700],
              bytes(rng.choice(list(b"acgt")) for _ in range(130)), templ_a[:40] + templ_a[-40:]]
        _check(templ_a, rs, ia_oracle)
        reads += rs
    assert len(reads) == 33


def test_reads_into_a_repeat_template(ia_oracle):
    """the shape of the real input: a template of some thousand bases with internal repeats, noisy reads of it (15 %
    errors: the stored band is a fraction of the matrix), reads reaching over both template ends, unrelated reads"""
    import random
    rng = random.Random(77)
    unit = bytes(rng.choice(list(b"acgt")) for _ in range(700))
    templ = bytes(rng.choice(list(b"acgt")) for _ in range(1500)) + unit + _mutate(rng, unit, 0.02) + unit[:350] + \
        bytes(rng.choice(list(b"acgt")) for _ in range(3000)) + _mutate(rng, unit, 0.01) + bytes(rng.choice(list(b"acgt")) for _ in range(1800))
    reads = []
    for k in range(36):
        a = rng.randrange(0, len(templ) - 500)
        reads.append(_mutate(rng, templ[a:a + rng.randrange(300, 2600)], 0.15))
    reads.append(bytes(rng.choice(list(b"acgt")) for _ in range(500)) + templ[:800])           # hangs over the left end
    reads.append(templ[-900:] + bytes(rng.choice(list(b"acgt")) for _ in range(400)))          # and over the right end
    reads.append(bytes(rng.choice(list(b"acgt")) for _ in range(1500)))                         # unrelated
    _check(templ, reads, ia_oracle)


def test_longest_template_and_limits(ia_oracle):
    """IA:214: Template[70000] -> 35 words per lane; IA:742: reads up to 40000 bases"""
    import random
    from repeatresolver_amd.initial_aligner import InitialAligner
    from repeatresolver_amd.realigner import PwrError
    rng = random.Random(5)
    templ = bytes(rng.choice(list(b"acgt")) for _ in range(70000))
    reads = [_mutate(rng, templ[69000:], 0.12), _mutate(rng, templ[30000:31500], 0.12), templ[:700], _mutate(rng, templ[2040:2060 + 900], 0.3)]
    _check(templ, reads, ia_oracle)
    with pytest.raises(PwrError) as e:
        InitialAligner(b"a" * 70001)
    assert e.value.code == -5
    g = InitialAligner(b"acgtacgt")
    with pytest.raises(PwrError) as e:
        g.align([b"a" * 40001])
    assert e.value.code == -5
    got, dist = g.align([])
    assert got == [] and len(dist) == 0
    g.close()


@pytest.mark.parametrize("budget", [None, 1 << 16, 1])
def test_batches_of_pass_two(ia_oracle, budget):
    """many reads in one call, launched longest-first, pass 2 in one batch / in batches of 64 KB of direction bits / one
    read per batch; results come back in the caller's order"""
    import random
    rng = random.Random(9)
    templ = bytes(rng.choice(list(b"acgt")) for _ in range(2500))
    reads = []
    for k in range(300):
        a = rng.randrange(0, 2300)
        reads.append(_mutate(rng, templ[a:a + rng.randrange(1, 400)], rng.choice([0.0, 0.05, 0.3])))
    _check(templ, reads, ia_oracle, budget)


def test_pipeline_dataset_to_realigned_msa(tmp_path):
    """The GPU steps chained as RepeatResolver.c chains them (InitialAligner -> PW_ReAligner -> MaxCorrelation): a simulated data set
    (DataSimulator's files), the reads cut to their repeat part, aligned into the template on the GPU, the MSA realigned
    on the GPU -- every file equal to the one the two CPU restatements produce from the same input."""
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.initial_aligner import run_files
    from repeatresolver_amd.realigner import run_file
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    cfg = dg.SimConfig(kind="Tree", copies=6, coverage=12, difference=0.01, repeat_len=2500, flank=1500, length_scale=0.25,
                       min_aligned=200, seed=3)
    prefix = str(tmp_path / "Sim")
    counts = dg.write_dataset(prefix, cfg)
    assert counts["cut_reads"] > 20
    rc, lines = run_files(prefix + "_Template.fasta", prefix + "Seq.fasta", prefix + "_gpu_MSA", prefix + "_gpu_SeqClass")
    assert rc == 0, lines
    p = subprocess.run([os.path.join(ROOT, "oracle", "ia_oracle"), prefix + "_Template.fasta", prefix + "Seq.fasta",
                        prefix + "_cpu_MSA", prefix + "_cpu_SeqClass"])
    assert p.returncode == 0
    assert open(prefix + "_gpu_SeqClass").read() == open(prefix + "_cpu_SeqClass").read()
    msa = open(prefix + "_gpu_MSA", "rb").read()
    assert msa == open(prefix + "_cpu_MSA", "rb").read() and msa.count(b"\n") > 20
    rc, lines = run_file(prefix + "_gpu_MSA", prefix + "_gpu_MSAreal", bandwidth=1000, max_rounds=3)
    assert rc == 0, lines
    p = subprocess.run([os.path.join(ROOT, "oracle", "pw_oracle"), prefix + "_cpu_MSA", "-o", prefix + "_cpu_MSAreal", "-b", "1000",
                        "-r", "3"], capture_output=True)
    assert p.returncode == 0
    assert open(prefix + "_gpu_MSAreal", "rb").read() == open(prefix + "_cpu_MSAreal", "rb").read()
    assert [l for l in lines if l.startswith("OverallScore")] == [l for l in p.stdout.decode("latin1").splitlines() if l.startswith("OverallScore")]
    # ... and the step behind: MaxCorrelation on the realigned MSA (floating point: 1e-6 on the printed values)
    import numpy as np
    from repeatresolver_amd.max_correlation import run_file as run_mc
    rc, lines = run_mc("Sim_gpu_MSAreal", mincov=20, cwd=str(tmp_path))
    assert rc == 0, lines
    p = subprocess.run([os.path.join(ROOT, "oracle", "mc_oracle"), prefix + "_cpu_MSAreal", prefix + "_cpu_MaxCorrs", "20"])
    assert p.returncode == 0
    got = np.array([float(v) for v in open(str(tmp_path / "MaxCorrsOf_Sim_gpu_MSAreal")).read().split()])
    exp = np.array([float(v) for v in open(prefix + "_cpu_MaxCorrs").read().split()])
    assert got.shape == exp.shape and (exp > 0).sum() > 50
    assert np.allclose(got, exp, rtol=0, atol=1.5e-6)


def test_reads_longer_than_the_template(ia_oracle):
    """reads several times the template's length (most bases stay unaligned, the distance exceeds the template, the stored
    band is wider than the wave's 64 lanes), next to short ones in the same call"""
    import random
    rng = random.Random(12)
    templ = bytes(rng.choice(list(b"acgt")) for _ in range(500))
    reads = [bytes(rng.choice(list(b"acgt")) for _ in range(3000)), templ * 3, _mutate(rng, templ, 0.1),
             bytes(rng.choice(list(b"ac")) for _ in range(1200)), templ[100:160]]
    _check(templ, reads, ia_oracle)
    templ2 = bytes(rng.choice(list(b"acgt")) for _ in range(2300))           # two words per lane
    _check(templ2, [bytes(rng.choice(list(b"acgt")) for _ in range(5000)), templ2 + templ2[:700]], ia_oracle)


def test_bytes_other_than_the_four_bases_are_refused():
    """The C ABI maps a base to its code with two bits of the character: anything but acgt / ACGT would alias one of them,
    so pia_create and pia_align refuse it (the reference's reader leaves nothing else, IA:190-209)."""
    import ctypes
    from repeatresolver_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.pia_create(ctypes.byref(h), b"acgtnacgt", 9, 0) == -4            # PWR_ERR_INPUT
    assert lib.pia_create(ctypes.byref(h), b"acgtAcGt", 8, 0) == 0
    reads = b"acg-acgt"
    off = (ctypes.c_longlong * 2)(0, len(reads))
    al = (ctypes.c_int * len(reads))()
    dist = (ctypes.c_int * 1)()
    assert lib.pia_align(h, 1, reads, off, al, dist) == -4
    reads = b"ACgtacgt"
    assert lib.pia_align(h, 1, reads, off, al, dist) == 0 and dist[0] == 0 and list(al) == list(range(8))
    lib.pia_destroy(h)
