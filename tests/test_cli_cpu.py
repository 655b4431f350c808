"""CPU: the sigtk-amd host CLI -- argument surface, the C BLOW5 reader, and the no-GPU error."""
import os
import time
import subprocess

import numpy as np
import pytest

from sigtk_amd import api, blow5, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def cli():
    path = build.CLI
    if not os.path.exists(path):
        build.build_lib()
        path = build.build_cli()
    return path


def run(cli, *args):
    return subprocess.run([cli, *args], capture_output=True, text=True)


def fnv(raw):
    h = 1469598103934665603
    for v in raw.astype(np.uint16).tolist():
        h ^= v
        h = (h * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_version_and_usage(cli):
    p = run(cli, "--version")
    assert p.returncode == 0 and p.stdout == "sigtk 0.2.0\n"          # src/main.c:101-104
    p = run(cli, "event", "--version")
    assert p.returncode == 0 and p.stdout == "sigtk 0.2.0\n"          # src/cmain.c:54-56
    p = run(cli)
    assert p.returncode == 1 and p.stderr.startswith("Usage: sigtk <command> [options]")
    p = run(cli, "--help")
    assert p.returncode == 0 and p.stdout.startswith("Usage: sigtk <command> [options]")
    p = run(cli, "event")
    assert p.returncode == 1 and "Usage: sigtk event reads.blow5" in p.stderr
    p = run(cli, "stat", "-h")
    assert p.returncode == 0 and "Usage: sigtk stat reads.blow5" in p.stdout
    p = run(cli, "frobnicate")
    assert p.returncode == 1 and "Unrecognised command frobnicate" in p.stderr
    p = run(cli, "event", "/nonexistent.blow5")
    assert p.returncode == 1 and "cannot open /nonexistent.blow5" in p.stderr


def _check_dump(cli, path, reads):
    p = run(cli, "_dump", path)
    assert p.returncode == 0, p.stderr
    # the pipelined reader's split API (next_raw + parse_raw + svb decode) sees the same records
    q = run(cli, "_dump", "--split", path)
    assert q.returncode == 0 and q.stdout == p.stdout
    q = run(cli, "_dump", "--map", path)                 # ... and so does the zero-copy reader over the mapped file
    assert q.returncode == 0 and q.stdout == p.stdout
    rows = [ln.split("\t") for ln in p.stdout.strip().split("\n")[1:]]
    assert len(rows) == len(reads)
    for row, r in zip(rows, reads):
        assert row[0] == r.read_id and int(row[1]) == r.raw.size
        assert float(row[2]) == r.digitisation and float(row[3]) == r.offset and float(row[4]) == r.range
        assert int(row[5], 16) == fnv(r.raw)


def test_c_reader_matches_python_reader(cli, sp1):
    _check_dump(cli, os.path.join(GOLDEN, "sp1_dna.blow5"), sp1.reads)


@pytest.mark.parametrize("rp,sp", [(0, 0), (1, 1), (1, 0), (0, 1)])
def test_c_reader_compression_variants(cli, tmp_path, rp, sp):
    reads, dig, off, rng = api.synth_reads_host(5, [0, 1, 777, 4096, 30000], 5, 0)
    rs = np.random.RandomState(2)
    reads[3] = rs.randint(-32768, 32767, size=4096).astype(np.int16)  # 3-byte svb codes
    recs = [blow5.Read("r%d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
            for i in range(5)]
    path = str(tmp_path / "x.blow5")
    blow5.write_blow5(path, recs, {"experiment_type": "rna", "sequencing_kit": "sqk-rna004"}, rp, sp)
    _check_dump(cli, path, recs)


def test_no_gpu_is_a_loud_error(cli):
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    p = run(cli, "event", "-c", os.path.join(GOLDEN, "sp1_dna.blow5"))
    assert p.returncode == 1 and "no usable GPU" in p.stderr
    assert "DNA data detected" in p.stderr and "R9 data detected" in p.stderr


def test_exact_fast_formatter_matches_printf(cli):
    """fmt.h: "%f" of floats and "%ld" of integers, against snprintf on every 9973rd
    float bit pattern plus ties/carries (dyadic fractions around the 6th decimal)."""
    p = run(cli, "_fmtcheck", "9973")
    assert p.returncode == 0 and p.stdout.strip().endswith("0 mismatches"), p.stdout


def test_truncated_file_is_an_error(cli, tmp_path):
    data = open(os.path.join(GOLDEN, "sp1_dna.blow5"), "rb").read()
    path = str(tmp_path / "cut.blow5")
    open(path, "wb").write(data[: len(data) // 2])
    for extra in ([], ["--split"], ["--map"]):
        p = run(cli, "_dump", *extra, path)
        assert p.returncode == 1


def _crafted(tmp_path, name, len_field, signal_bytes, signal_press):
    """an uncompressed-record BLOW5 with one record whose len_raw_signal field is `len_field`"""
    import struct
    path = str(tmp_path / name)
    blow5.write_blow5(path, [blow5.Read("r0", 0, 8192.0, 3.0, 1402.882324, 4000.0,
                                        np.arange(16, dtype=np.int16))], {}, 0, signal_press)
    data = bytearray(open(path, "rb").read())
    hdr_len = struct.unpack_from("<I", data, 64)[0]
    rec0 = 68 + hdr_len
    rid = b"r0"
    body = struct.pack("<H", len(rid)) + rid + struct.pack("<Iddddq", 0, 8192.0, 3.0, 1402.882324, 4000.0, 0)
    body = body[:-8] + struct.pack("<Q", len_field) + signal_bytes
    out = bytes(data[:rec0]) + struct.pack("<Q", len(body)) + body + b"5WOLB"
    open(path, "wb").write(out)
    return path


@pytest.mark.parametrize("len_field", [1 << 63, 1 << 32, (1 << 63) + 8, 0x7fffffff])
def test_crafted_sample_count_is_rejected(cli, tmp_path, len_field):
    """len_raw_signal is untrusted -- 2^63 made `ln * 2` wrap and the reader walk off its buffer"""
    path = _crafted(tmp_path, "evil.blow5", len_field, b"\x01\x00" * 8, 0)
    for extra in ([], ["--split"], ["--map"]):
        p = subprocess.run([cli, "_dump", *extra, path], capture_output=True, timeout=60)
        assert p.returncode == 1, (extra, p.returncode, p.stderr[-200:])


def test_crafted_svb_count_is_rejected(cli, tmp_path):
    """a 5-byte svb-zd blob that claims 2^31 - 1 samples must be refused at parse time, not after the buffers
    for it were sized"""
    import struct
    blob = struct.pack("<I", 0x7fffffff) + b"\x00"
    path = _crafted(tmp_path, "evil_svb.blow5", len(blob), blob, 1)
    for extra in ([], ["--split"], ["--map"]):
        p = subprocess.run([cli, "_dump", *extra, path], capture_output=True, timeout=60)
        assert p.returncode == 1, (extra, p.returncode, p.stderr[-200:])


def test_reader_survives_corrupt_input(cli, tmp_path):
    """bit flips, truncations and overwritten stretches of a real file: both reader paths must report an error
    (or read what is still consistent), never crash.  (The same loop was run 600 times against an
    -fsanitize=address,undefined build of the host sources without a report.)"""
    import random
    data = open(os.path.join(GOLDEN, "sp1_dna.blow5"), "rb").read()
    rnd = random.Random(5)
    path = str(tmp_path / "fz.blow5")
    for it in range(45):
        d = bytearray(data)
        if it % 3 == 0:
            for _ in range(rnd.randint(1, 5)):
                d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
        elif it % 3 == 1:
            d = d[: rnd.randrange(70, len(d))]
        else:
            i = rnd.randrange(60, len(d) - 8)
            d[i:i + 8] = bytes(rnd.getrandbits(8) for _ in range(8))
        open(path, "wb").write(bytes(d))
        for extra in ([], ["--split"], ["--map"]):
            p = subprocess.run([cli, "_dump", *extra, path], capture_output=True, timeout=60)
            assert p.returncode in (0, 1), (it, extra, p.returncode, p.stderr[-300:])


def test_index_file_is_written_reused_and_checked(cli, tmp_path, sp1):
    """read-id mode keeps "<file>.idx" beside the BLOW5 like the reference (src/cmain.c:127-131,
    slow5lib/src/slow5_idx.c): created by the first run, loaded by the next, ignored when it does not fit the file"""
    import shutil
    path = str(tmp_path / "a.blow5")
    shutil.copy(os.path.join(GOLDEN, "sp1_dna.blow5"), path)
    r = sp1.reads[37]
    p = run(cli, "_dump", "--id", r.read_id, path)
    assert p.returncode == 0 and p.stdout.split("\t")[0] == r.read_id and int(p.stdout.split("\t")[5], 16) == fnv(r.raw)
    idx = open(path + ".idx", "rb").read()
    assert idx[:9] == b"SLOW5IDX\x01" and idx[9:12] == bytes([0, 2, 0]) and idx[-8:] == b"XDI5WOLS"
    assert idx[12:64] == bytes(52)
    # 100 entries in file order: u16 id length, id, u64 offset, u64 size; offsets increase by the sizes
    import struct
    pos, off_prev, ids = 64, None, []
    while pos < len(idx) - 8:
        (idl,) = struct.unpack_from("<H", idx, pos)
        rid = idx[pos + 2:pos + 2 + idl].decode()
        off, size = struct.unpack_from("<QQ", idx, pos + 2 + idl)
        assert off_prev is None or off == off_prev
        off_prev = off + size
        ids.append(rid)
        pos += 2 + idl + 16
    assert ids == [x.read_id for x in sp1.reads] and off_prev == os.path.getsize(path) - 5
    # second run: loads the index (make the data file unscannable past the first record to prove it)
    first = sp1.reads[0]
    p = run(cli, "_dump", "--id", sp1.reads[99].read_id, path)
    assert p.returncode == 0 and int(p.stdout.split("\t")[5], 16) == fnv(sp1.reads[99].raw)
    # an index of another file (offsets beyond the end) is not trusted: the file is scanned again, the index rewritten
    bad = bytearray(idx)
    struct.pack_into("<Q", bad, 64 + 2 + len(ids[0]), 1 << 40)
    open(path + ".idx", "wb").write(bytes(bad))
    p = run(cli, "_dump", "--id", first.read_id, path)
    assert p.returncode == 0 and int(p.stdout.split("\t")[5], 16) == fnv(first.raw)
    assert open(path + ".idx", "rb").read() == idx
    # unknown id
    assert run(cli, "_dump", "--id", "no-such-read", path).returncode == 1


def test_index_of_another_file_is_detected(cli, tmp_path, sp1):
    """an "<file>.idx" that passes the format checks but belongs to ANOTHER file of that name (same
    records in another order: every offset + size is inside the file) must not return the wrong read -- the record
    fetched through the index is compared with the entry (id, size); an index older than the BLOW5 is not used at
    all (slow5lib only warns, slow5_idx.c:43)."""
    import shutil
    import time
    a = str(tmp_path / "a.blow5")
    shutil.copy(os.path.join(GOLDEN, "sp1_dna.blow5"), a)
    want = {r.read_id: fnv(r.raw) for r in sp1.reads}
    assert run(cli, "_dump", "--id", sp1.reads[0].read_id, a).returncode == 0      # writes a.blow5.idx
    good = open(a + ".idx", "rb").read()
    # the same reads in reverse order under the same name, with the OLD index made newer than the file
    blow5.write_blow5(a, list(reversed(sp1.reads)), {k: v[0] for k, v in sp1.attrs.items()}, 1, 1)
    open(a + ".idx", "wb").write(good)
    future = time.time() + 100
    os.utime(a + ".idx", (future, future))
    for rid in (sp1.reads[3].read_id, sp1.reads[97].read_id):
        for extra in ([], ["--split"]):
            p = run(cli, "_dump", *extra, "--id", rid, a)
            assert p.returncode == 0, p.stderr
            assert p.stdout.split("\t")[0] == rid and int(p.stdout.split("\t")[5], 16) == want[rid]
        # the stale index was replaced by one of the new file
        assert open(a + ".idx", "rb").read() != good
        open(a + ".idx", "wb").write(good)
        os.utime(a + ".idx", (future, future))
    # an index older than the data file is used with a warning, as slow5lib does (after cp / rsync / tar
    # every invocation would scan the whole file otherwise) -- the check of every fetched record still catches this one
    past = time.time() - 1000
    os.utime(a + ".idx", (past, past))
    p = run(cli, "_dump", "--id", sp1.reads[50].read_id, a)
    assert p.returncode == 0 and int(p.stdout.split("\t")[5], 16) == want[sp1.reads[50].read_id]
    assert "older than its BLOW5" in p.stderr
    # ... and a GOOD older index is simply used: no rescan, no rewrite
    blow5.write_blow5(a, sp1.reads, {k: v[0] for k, v in sp1.attrs.items()}, 1, 1)
    os.remove(a + ".idx")
    assert run(cli, "_dump", "--id", sp1.reads[0].read_id, a).returncode == 0
    fresh = open(a + ".idx", "rb").read()
    os.utime(a + ".idx", (past, past))
    mt = os.stat(a + ".idx").st_mtime
    p = run(cli, "_dump", "--id", sp1.reads[7].read_id, a)
    assert p.returncode == 0 and int(p.stdout.split("\t")[5], 16) == want[sp1.reads[7].read_id]
    assert open(a + ".idx", "rb").read() == fresh and os.stat(a + ".idx").st_mtime == mt


def test_read_id_mode_of_the_cli_checks_the_id_of_compressed_records(cli, tmp_path, sp1):
    """the pipeline's read-id path (b5_get_raw) only compared the SIZE of a compressed record with its
    index entry; an index of another file that points at another record of the same compressed size returned the wrong
    read.  Two reads are given identical signals (identical compressed sizes up to the id, which has a fixed length)
    and the index entries of the two are swapped."""
    import copy
    import struct
    a = str(tmp_path / "b.blow5")
    for tweak in "0123456789abcdef":
        reads = [copy.copy(r) for r in sp1.reads[:6]]
        rid = reads[1].read_id
        if rid[-1] == tweak:
            continue
        reads[4] = copy.copy(reads[1]); reads[4].read_id = rid[:-1] + tweak    # same bytes but one character of the id
        reads[4].raw = reads[1].raw.copy()
        blow5.write_blow5(a, reads, {k: v[0] for k, v in sp1.attrs.items()}, 1, 1)
        if os.path.exists(a + ".idx"):
            os.remove(a + ".idx")
        assert run(cli, "_dump", "--id", reads[0].read_id, a).returncode == 0          # writes b.blow5.idx
        idx = bytearray(open(a + ".idx", "rb").read())
        # parse the entries: (u16 len, id, u64 offset, u64 size)
        pos, ent = 64, {}
        while pos < len(idx) - 8:
            n = struct.unpack_from("<H", idx, pos)[0]
            ent[bytes(idx[pos + 2:pos + 2 + n]).decode()] = pos + 2 + n
            pos += 2 + n + 16
        p1, p4 = ent[reads[1].read_id], ent[reads[4].read_id]
        e1, e4 = bytes(idx[p1:p1 + 16]), bytes(idx[p4:p4 + 16])
        if struct.unpack_from("<Q", e1, 8)[0] == struct.unpack_from("<Q", e4, 8)[0]:
            break
    else:
        pytest.skip("no pair of records deflated to the same size")
    idx[p1:p1 + 16], idx[p4:p4 + 16] = e4, e1
    open(a + ".idx", "wb").write(bytes(idx))
    future = time.time() + 100
    os.utime(a + ".idx", (future, future))
    # the pipeline's read-id path (b5_get_raw + b5_parse_raw: `_dump --split --id`), for both records
    for rd in (reads[4], reads[1]):
        p = run(cli, "_dump", "--split", "--id", rd.read_id, a)
        assert p.returncode == 0, p.stderr
        assert p.stdout.split("\t")[0] == rd.read_id and int(p.stdout.split("\t")[5], 16) == fnv(rd.raw)
        open(a + ".idx", "wb").write(bytes(idx))
        os.utime(a + ".idx", (future, future))


@pytest.fixture(scope="module")
def cli_asan():
    try:
        return build.build_cli_asan()
    except (subprocess.CalledProcessError, OSError) as e:  # no libasan in this toolchain
        pytest.skip("sanitizer build not available: %s" % e)


def test_hostile_inputs_under_asan_ubsan(cli_asan, tmp_path, sp1):
    """The host sources built with -fsanitize=address,undefined (`python -m sigtk_amd.build --asan`, the counterpart
    of the reference's `make asan=1`, Makefile:31-34): the reader's hostile-input cases -- crafted length fields,
    truncations, bit flips, a stale index -- must end with exit code 0 or 1 and without a sanitizer report."""
    import random
    import struct
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:exitcode=98")

    def run_a(*args):
        p = subprocess.run([cli_asan, *args], capture_output=True, timeout=120, env=env)
        assert p.returncode in (0, 1), (args, p.returncode, p.stderr[-600:])
        assert b"Sanitizer" not in p.stderr and b"runtime error" not in p.stderr, p.stderr[-600:]
        return p

    good = os.path.join(GOLDEN, "sp1_dna.blow5")
    for extra in ([], ["--split"], ["--map"]):
        assert run_a("_dump", *extra, good).returncode == 0
    for lf in (1 << 63, 1 << 32, (1 << 63) + 8, 0x7fffffff):
        path = _crafted(tmp_path, "evil.blow5", lf, b"\x01\x00" * 8, 0)
        for extra in ([], ["--split"], ["--map"]):
            assert run_a("_dump", *extra, path).returncode == 1
    blob = struct.pack("<I", 0x7fffffff) + b"\x00"
    path = _crafted(tmp_path, "evil_svb.blow5", len(blob), blob, 1)
    for extra in ([], ["--split"], ["--map"]):
        assert run_a("_dump", *extra, path).returncode == 1
    data = open(good, "rb").read()
    rnd = random.Random(11)
    path = str(tmp_path / "fz.blow5")
    for it in range(24):
        d = bytearray(data)
        if it % 3 == 0:
            for _ in range(rnd.randint(1, 5)):
                d[rnd.randrange(len(d))] ^= 1 << rnd.randrange(8)
        elif it % 3 == 1:
            d = d[: rnd.randrange(70, len(d))]
        else:
            i = rnd.randrange(60, len(d) - 8)
            d[i:i + 8] = bytes(rnd.getrandbits(8) for _ in range(8))
        open(path, "wb").write(bytes(d))
        for extra in ([], ["--split"], ["--map"]):
            run_a("_dump", *extra, path)
    # read-id mode through a corrupted index
    import shutil
    a = str(tmp_path / "a.blow5")
    shutil.copy(good, a)
    assert run_a("_dump", "--id", sp1.reads[5].read_id, a).returncode == 0
    idx = bytearray(open(a + ".idx", "rb").read())
    for it in range(12):
        b = bytearray(idx)
        for _ in range(3):
            b[rnd.randrange(64, len(b))] ^= 1 << rnd.randrange(8)
        open(a + ".idx", "wb").write(bytes(b))
        run_a("_dump", "--id", sp1.reads[rnd.randrange(100)].read_id, a)
    assert run_a("_fmtcheck", "99991").returncode == 0
