"""GPU: the sigtk-amd CLI reproduces the real reference's stdout byte for byte
(goldens from tests/golden/make_golden.py, which ran the reference built from source)."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from sigtk_amd import api, blow5, build

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
SP1 = os.path.join(GOLDEN, "sp1_dna.blow5")


@pytest.fixture(scope="module")
def cli(gpu):
    assert os.path.exists(build.CLI), "sigtk-amd not built (run __graft_entry__.build())"
    return build.CLI


def out(cli, *args):
    p = subprocess.run([cli, *args], capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return p.stdout


def gold(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


@pytest.fixture(scope="module")
def synth_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("synth")
    files = {}
    for name, spec in list(MANIFEST["_synth_specs"].items()) + list(MANIFEST["_synth_long_specs"].items()):
        n, ln, seed, kind, exp, kit = spec
        reads, dig, off, rng = api.synth_reads_host(n, ln, seed, kind)
        recs = [blow5.Read("synth-%08d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
                for i in range(n)]
        path = str(d / (name + ".blow5"))
        blow5.write_blow5(path, recs, {"experiment_type": exp, "sequencing_kit": kit})
        files[name] = path
    return files


@pytest.mark.parametrize("fname,args", [
    ("sp1_dna.event_c.tsv", ["event", "-c", SP1]),
    ("sp1_dna.stat.tsv", ["stat", SP1]),
    ("sp1_dna.jnn.tsv", ["jnn", SP1]),
    ("sp1_dna.jnn_c.tsv", ["jnn", SP1, "-c"]),          # options may follow positionals (GNU getopt)
    ("sp1_dna.prefix.tsv", ["prefix", SP1]),
    ("sp1_dna.prefix_stat.tsv", ["prefix", "--print-stat", SP1]),
    ("sp1_dna.ent.tsv", ["ent", SP1]),
    ("sp1_dna.pa3.tsv", ["pa", SP1, "00011a60-dd92-4aad-be1d-59a33545ab1d",
                         "0448591b-036c-4cc7-a702-6c542ccc07de", "03880e3d-b79d-4bd8-aab4-15724f1331af"]),
])
def test_sp1_outputs(cli, fname, args):
    assert out(cli, *args) == gold(fname)


def test_sp1_event_long_form_hash(cli):
    assert hashlib.sha256(out(cli, "event", SP1)).hexdigest() == MANIFEST["sp1_dna.event.tsv.sha256"]
    # BASELINE config 1: `sigtk pa` on the whole bundled file, byte for byte what the reference prints
    assert hashlib.sha256(out(cli, "pa", SP1)).hexdigest() == MANIFEST["sp1_dna.pa.tsv.sha256"]


def test_reference_golden_event_dna_exp(cli):
    """scripts/test.sh:71 -- event on one read id; event_dna.exp has a stale header line."""
    got = out(cli, "event", SP1, "05d90f17-f4a6-4349-924c-3ffd3457a99d")
    assert got.split(b"\n", 1)[1] == gold("event_dna.exp").split(b"\n", 1)[1]


def test_reference_golden_prefix_dna_exp(cli):
    assert out(cli, "prefix", SP1) == gold("prefix_dna.exp")          # scripts/test.sh:55


def test_no_header_and_small_batches(cli):
    full = out(cli, "event", "-c", SP1)
    assert out(cli, "event", "-c", "-n", SP1) == full.split(b"\n", 1)[1]
    # force many batches: rows must still come out in file order
    assert out(cli, "event", "-c", "--batch-samples", "20000", SP1) == full
    assert out(cli, "stat", "--batch-samples", "5000", SP1) == gold("sp1_dna.stat.tsv")


def test_pipeline_options_do_not_change_the_output(cli, synth_files):
    """host svb-zd decode vs GPU decode, 1 vs many host threads, tiny batches (many jobs in flight)"""
    for tool in (["event", "-c"], ["event"], ["stat"], ["jnn"], ["prefix", "--print-stat"], ["pa"]):
        for f in (SP1, synth_files["synth_rna"] if "synth_rna" in synth_files else SP1):
            if tool == ["pa"] and f != SP1:
                continue
            base = out(cli, *tool, f)
            assert out(cli, *tool, "--host-decode", f) == base
            assert out(cli, *tool, "-t", "1", f) == base
            assert out(cli, *tool, "--threads", "7", "--batch-samples", "30000", f) == base


@pytest.mark.parametrize("rp,sp", [(0, 0), (1, 0), (0, 1)])
def test_other_compression_layouts(cli, tmp_path, sp1, rp, sp):
    """record compression none/zlib x signal compression none/svb-zd (sp1_dna.blow5 itself is zlib + svb-zd)"""
    path = str(tmp_path / "x.blow5")
    blow5.write_blow5(path, sp1.reads, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"}, rp, sp)
    assert out(cli, "event", "-c", path) == gold("sp1_dna.event_c.tsv")
    assert out(cli, "stat", path) == gold("sp1_dna.stat.tsv")


def test_corrupt_signal_blob_fails_like_a_read_error(cli, tmp_path, sp1):
    recs = [blow5.Read(r.read_id, 0, r.digitisation, r.offset, r.range, r.sampling_rate, r.raw) for r in sp1.reads[:3]]
    path = str(tmp_path / "bad.blow5")
    blow5.write_blow5(path, recs, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"}, 0, 1)
    data = bytearray(open(path, "rb").read())
    # flip the count word of the last record's blob: the GPU decoder must flag it and the CLI must fail
    blob = blow5.svb_zd_encode(recs[2].raw)
    at = bytes(data).rfind(blob[:64])
    assert at > 0
    data[at] ^= 0x55
    open(path, "wb").write(bytes(data))
    p = subprocess.run([cli, "stat", path], capture_output=True)
    assert p.returncode != 0


@pytest.mark.parametrize("bits,method", [(1, "round"), (3, "round"), (2, "floor"), (4, "fill-ones")])
def test_qts_output_reads_back_like_the_references(cli, tmp_path, bits, method):
    """`qts` (src/qts.c): what a reader sees in our output file equals what it sees in the file the reference wrote
    (digest of ids, scaling and quantised samples, from make_golden.py); everything but the signal -- auxiliary
    fields included -- is kept byte for byte, and the header block is the input's."""
    import struct
    outp = str(tmp_path / "q.blow5")
    p = subprocess.run([cli, "qts", SP1, "-o", outp, "-b", str(bits), "-m", method, "--batch-samples", "150000"],
                       capture_output=True)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert blow5.digest(outp) == MANIFEST["sp1_dna.qts_b%d_%s.sha256" % (bits, method)]
    src, dst = open(SP1, "rb").read(), open(outp, "rb").read()
    (hsize,) = struct.unpack_from("<I", src, 64)
    assert dst[: 68 + hsize] == src[: 68 + hsize] and dst[-5:] == b"5WOLB"
    for a, b in zip(blow5.raw_records(SP1), blow5.raw_records(outp)):
        (idl,) = struct.unpack_from("<H", a, 0)
        sig = 2 + idl + 36          # offset of len_raw_signal
        (la,) = struct.unpack_from("<Q", a, sig)
        (lb,) = struct.unpack_from("<Q", b, sig)
        assert a[:sig] == b[:sig]                                   # id, read group, scaling, sampling rate
        assert a[sig + 8 + la:] == b[sig + 8 + lb:] and len(a[sig + 8 + la:]) > 0   # auxiliary fields


def test_qts_other_layouts_and_errors(cli, tmp_path, sp1):
    recs = sp1.reads[:7]
    for rp, sp in ((0, 0), (1, 0), (0, 1)):
        inp, outp = str(tmp_path / "i.blow5"), str(tmp_path / "o.blow5")
        blow5.write_blow5(inp, recs, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"}, rp, sp)
        assert subprocess.run([cli, "qts", inp, "-o", outp, "-b", "2"], capture_output=True).returncode == 0
        got = blow5.read_blow5(outp)
        assert (got.record_press, got.signal_press) == (rp, sp)
        for g, r in zip(got.reads, recs):
            x = r.raw.astype(np.int64)
            e = np.where((x & 3) < 2, x & ~3, (x & ~3) + 4).astype(np.int16)
            assert g.read_id == r.read_id and np.array_equal(g.raw, e)
    assert subprocess.run([cli, "qts", SP1], capture_output=True).returncode != 0                      # no -o
    assert subprocess.run([cli, "qts", SP1, "-o", str(tmp_path / "x"), "-b", "9"], capture_output=True).returncode != 0
    assert subprocess.run([cli, "qts", SP1, "-o", str(tmp_path / "x"), "-m", "nearest"], capture_output=True).returncode != 0


def test_more_gpus_requested_than_present(cli):
    p = subprocess.run([cli, "stat", "--gpus", "64", SP1], capture_output=True)
    assert p.returncode == 0 and p.stdout == gold("sp1_dna.stat.tsv")


@pytest.mark.parametrize("name", list(MANIFEST["_synth_specs"]))
def test_synthetic_outputs(cli, synth_files, name):
    f = synth_files[name]
    assert out(cli, "event", "-c", f) == gold(name + ".event_c.tsv")
    assert out(cli, "stat", f) == gold(name + ".stat.tsv")
    assert out(cli, "jnn", f) == gold(name + ".jnn.tsv")
    assert out(cli, "prefix", "--print-stat", f) == gold(name + ".prefix_stat.tsv")
    assert out(cli, "ent", f) == gold(name + ".ent.tsv")
    assert out(cli, "ent", "--no-header", "--batch-samples", "50000", f) == gold(name + ".ent.tsv").split(b"\n", 1)[1]
    assert hashlib.sha256(out(cli, "event", f)).hexdigest() == MANIFEST[name + ".event.tsv.sha256"]


@pytest.mark.parametrize("name", list(MANIFEST["_synth_long_specs"]))
def test_synthetic_long_reads(cli, synth_files, name):
    """reads of 300 000 - 700 000 samples through the CLI (its batches put them on k_long_chains for stat / jnn / prefix
    and on event's segments); the goldens are the real reference's output (tests/golden/make_golden.py)"""
    f = synth_files[name]
    assert out(cli, "stat", f) == gold(name + ".stat.tsv")
    assert out(cli, "jnn", f) == gold(name + ".jnn.tsv")
    assert out(cli, "prefix", "--print-stat", f) == gold(name + ".prefix_stat.tsv")
    assert hashlib.sha256(out(cli, "event", "-c", f)).hexdigest() == MANIFEST[name + ".event_c.tsv.sha256"]



def test_records_inflated_on_the_gpu_and_on_the_host_give_the_same_bytes(cli, tmp_path, sp1):
    """round 5: zlib records with an svb-zd signal go to the GPU as they sit in the file (sgk_inflate); --host-inflate
    keeps the host threads' zlib.  Same output either way, on the reference's fixture (which carries auxiliary fields
    behind the signal: their fixed size is read off the header) and on long synthetic reads."""
    for tool in (["event", "-c"], ["stat"], ["jnn"], ["prefix", "--print-stat"], ["ent"], ["pa"]):
        a = out(cli, *tool, SP1)
        b = out(cli, *tool, "--host-inflate", SP1)
        assert a == b, tool
    from sigtk_amd import api
    reads, dig, off, rng = api.synth_reads_host(40, [100000, 250, 1000, 777, 70001, 300000] + [25000] * 34, 5, 0)
    recs = [blow5.Read("r-%04d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i]) for i in range(len(reads))]
    path = str(tmp_path / "z.blow5")
    blow5.write_blow5(path, recs, {"experiment_type": "genomic_dna", "sequencing_kit": "sqk-lsk109"})
    for tool in (["event", "-c"], ["stat"], ["prefix"]):
        assert out(cli, *tool, path) == out(cli, *tool, "--host-inflate", path), tool
        assert out(cli, *tool, "--batch-samples", "200000", path) == out(cli, *tool, "--host-inflate", path), tool


def test_a_record_that_does_not_inflate_fails_like_a_read_error(cli, tmp_path, sp1):
    """a flipped byte inside a record's zlib stream: the GPU inflate flags it (bad code / check value) and the CLI
    fails as the reference does on a slow5_get_next error"""
    data = bytearray(open(SP1, "rb").read())
    import struct
    (hsize,) = struct.unpack_from("<I", data, 64)
    pos = 68 + hsize
    (size,) = struct.unpack_from("<Q", data, pos)
    data[pos + 8 + size // 2] ^= 0x10          # the middle of the first record
    path = str(tmp_path / "bad.blow5")
    open(path, "wb").write(bytes(data))
    for extra in ([], ["--host-inflate"]):
        p = subprocess.run([cli, "stat", *extra, path], capture_output=True)
        assert p.returncode != 0 and b"slow5_get_next" in p.stderr, (extra, p.stderr[-300:])
