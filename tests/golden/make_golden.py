#!/usr/bin/env python3
"""Regenerate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference, built into oracle/_ref by
`make -C oracle ref`).  Inputs: the reference's bundled fixture test/sp1_dna.blow5 (copied
here as a data file together with the reference's own goldens event_dna.exp, prefix_dna.exp,
prefix_dna.exp2) and synthetic reads from the repo's deterministic generator (regenerable from
the seeds below, so only the reference's OUTPUTS are stored).

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from sigtk_amd import api, blow5  # noqa: E402
from oracle.oracle import REF_BIN  # noqa: E402

SYNTH = {  # name -> (n_reads, read_len, seed, kind, experiment_type, sequencing_kit)
    "synth_dna": (4, 100000, 101, 0, "genomic_dna", "sqk-lsk109"),
    "synth_rna": (4, 100000, 202, 1, "rna", "sqk-rna002"),
    "synth_rna004": (3, 60000, 303, 1, "rna", "sqk-rna004"),
    "synth_ragged": (12, [200, 201, 250, 333, 1000, 2000, 2001, 2048, 4096, 4097, 9999, 30000], 404, 0,
                     "genomic_dna", "sqk-lsk109"),
}
# reads the long-read path of stat / jnn / prefix takes (k_long_chains) and event's segments: `event -c` as a hash
SYNTH_LONG = {
    "synth_long": (3, [700000, 5000, 300000], 505, 0, "genomic_dna", "sqk-lsk109"),
    "synth_long_rna": (2, [400000, 30000], 606, 1, "rna", "sqk-rna002"),
}
PA_READS = ["00011a60-dd92-4aad-be1d-59a33545ab1d", "0448591b-036c-4cc7-a702-6c542ccc07de",
            "03880e3d-b79d-4bd8-aab4-15724f1331af"]


def ref(cwd, *args):
    p = subprocess.run([REF_BIN, *args], capture_output=True, cwd=cwd)
    if p.returncode != 0:
        raise RuntimeError("reference failed: %s\n%s" % (args, p.stderr.decode()[-2000:]))
    return p.stdout


def write_synth_blow5(path, spec):
    n, ln, seed, kind, exp, kit = spec
    reads, dig, off, rng = api.synth_reads_host(n, ln, seed, kind)
    recs = [blow5.Read("synth-%08d" % i, 0, float(dig[i]), float(off[i]), float(rng[i]), 4000.0, reads[i])
            for i in range(n)]
    blow5.write_blow5(path, recs, {"experiment_type": exp, "sequencing_kit": kit})


def main():
    if not os.path.exists(REF_BIN):
        raise SystemExit("build the reference first: make -C oracle ref")
    manifest = {}

    def save(name, data):
        with open(os.path.join(HERE, name), "wb") as fh:
            fh.write(data)
        manifest[name] = hashlib.sha256(data).hexdigest()

    with tempfile.TemporaryDirectory() as tmp:
        sp1 = os.path.join(tmp, "sp1_dna.blow5")
        with open(os.path.join(HERE, "sp1_dna.blow5"), "rb") as src, open(sp1, "wb") as dst:
            dst.write(src.read())
        save("sp1_dna.event_c.tsv", ref(tmp, "event", "-c", sp1))
        save("sp1_dna.stat.tsv", ref(tmp, "stat", sp1))
        save("sp1_dna.jnn.tsv", ref(tmp, "jnn", sp1))
        save("sp1_dna.jnn_c.tsv", ref(tmp, "jnn", "-c", sp1))
        save("sp1_dna.prefix.tsv", ref(tmp, "prefix", sp1))
        save("sp1_dna.prefix_stat.tsv", ref(tmp, "prefix", "--print-stat", sp1))
        save("sp1_dna.pa3.tsv", ref(tmp, "pa", sp1, *PA_READS))
        save("sp1_dna.ent.tsv", ref(tmp, "ent", sp1))
        # long-form event output of the whole file is ~6.5 MB: keep its hash only
        manifest["sp1_dna.event.tsv.sha256"] = hashlib.sha256(ref(tmp, "event", sp1)).hexdigest()
        # BASELINE config 1: `pa` over the whole fixture (4.4 MB of text): hash only
        manifest["sp1_dna.pa.tsv.sha256"] = hashlib.sha256(ref(tmp, "pa", sp1)).hexdigest()
        # qts writes a BLOW5: keep a digest of what a reader sees in the reference's output (ids, scaling, signal)
        for bits, method in ((1, "round"), (3, "round"), (2, "floor"), (4, "fill-ones")):
            outp = os.path.join(tmp, "q.blow5")
            ref(tmp, "qts", sp1, "-o", outp, "-b", str(bits), "-m", method)
            manifest["sp1_dna.qts_b%d_%s.sha256" % (bits, method)] = blow5.digest(outp)
        for name, spec in SYNTH.items():
            f = os.path.join(tmp, name + ".blow5")
            write_synth_blow5(f, spec)
            save(name + ".event_c.tsv", ref(tmp, "event", "-c", f))
            save(name + ".stat.tsv", ref(tmp, "stat", f))
            save(name + ".jnn.tsv", ref(tmp, "jnn", f))
            save(name + ".prefix_stat.tsv", ref(tmp, "prefix", "--print-stat", f))
            save(name + ".ent.tsv", ref(tmp, "ent", f))
            manifest[name + ".event.tsv.sha256"] = hashlib.sha256(ref(tmp, "event", f)).hexdigest()
        for name, spec in SYNTH_LONG.items():
            f = os.path.join(tmp, name + ".blow5")
            write_synth_blow5(f, spec)
            save(name + ".stat.tsv", ref(tmp, "stat", f))
            save(name + ".jnn.tsv", ref(tmp, "jnn", f))
            save(name + ".prefix_stat.tsv", ref(tmp, "prefix", "--print-stat", f))
            manifest[name + ".event_c.tsv.sha256"] = hashlib.sha256(ref(tmp, "event", "-c", f)).hexdigest()
    manifest["_synth_specs"] = {k: list(v) for k, v in SYNTH.items()}
    manifest["_synth_long_specs"] = {k: list(v) for k, v in SYNTH_LONG.items()}
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as fh:
        json.dump(manifest, fh, indent=1, sort_keys=True)
    print("wrote %d golden files" % (len(manifest) - 1))


if __name__ == "__main__":
    main()
