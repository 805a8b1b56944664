#!/usr/bin/env python3
"""Generate tests/golden/kat.json from the reference's in-file known-answer tests.

Run in the build container only (needs /root/reference); the JSON it writes is the
committed fixture.  It extracts *data* (inputs and expected outputs) from the
reference's `#[test]` bodies - no reference code is copied.

Sources (path:line in /root/reference):
  kmer/src/kmer_minimisers.rs:214-298  100-mer + 70 canonical 31-mers (only k=31 KAT)
  kmer/src/kmer.rs:114-176             (fwd,rev) pairs, N reset, rev_comp, pos_map k=4
  kmer/src/lib.rs:57-71                numeric <-> ACGT
  composition/src/oligo.rs:270-309     AAAANGAGA vectors
  composition/src/oligocgr.rs:200-220  29-mer k=4 vecsize 16
  tests/test_kmers.py, tests/test_utils.py (python surface)
"""
import json
import pathlib
import re

REF = pathlib.Path("/root/reference")
out = {}

# --- k=31 KAT -------------------------------------------------------------
src = (REF / "kmer/src/kmer_minimisers.rs").read_text()
m = re.search(r'KmerMinimiserGenerator::new\(b"([ACGT]+)",\s*31,\s*7\)', src)
seq = m.group(1)
pairs = re.findall(r'\("([ACGT]{31})",\s*"([ACGT]{7})"\)', src)
assert len(seq) == 100 and len(pairs) == 70, (len(seq), len(pairs))
out["k31"] = {
    "source": "kmer/src/kmer_minimisers.rs:216-288",
    "seq": seq,
    "k": 31,
    "canonical_kmers_in_order": [p[0] for p in pairs],
}

# --- small inline KATs (values transcribed from the asserts) ----------------
out["kmers"] = [
    {"source": "kmer/src/kmer.rs:114-128", "seq": "ACGT", "k": 2,
     "pairs": [[1, 11], [6, 6], [11, 1]]},
    {"source": "kmer/src/kmer.rs:131-145", "seq": "ACNGTT", "k": 2,
     "pairs": [[1, 11], [11, 1], [15, 0]]},
]
out["rev_comp"] = [
    {"source": "kmer/src/kmer.rs:148-153", "kmer": 0b00011011, "k": 4, "rc": 0b00011011},
    {"source": "kmer/src/kmer.rs:148-153", "kmer": 0b001101101011, "k": 6, "rc": 0b000101100011},
]
out["pos_map_k4"] = {
    "source": "kmer/src/kmer.rs:156-176",
    "count": 136, "nonzero_entries": 135,
    "entries": {"0": 0, "255": 0, "3": 3},
}
out["numeric"] = {
    "source": "kmer/src/lib.rs:57-71; tests/test_utils.py:4-16",
    "to_acgt": [[111, 5, "ACGTT"], [27, 5, "AACGT"]],
    "to_numeric": [["ACGTT", 111, 27]],
}
out["py_kmers"] = {
    "source": "tests/test_kmers.py:5-11",
    "seq": "ACGTCC", "k": 3, "fwd_acgt": ["ACG", "CGT", "GTC", "TCC"],
}
out["oligo_one"] = {
    "source": "composition/src/oligo.rs:270-309",
    "seq": "AAAANGAGA", "k": 4,
    "raw_len": 256, "raw_header_first": "AAAA", "raw_header_last": "TTTT",
    "canon_norm_v0": 0.5, "canon_unnorm_v0": 1.0, "canon_unnorm_sum": 2.0,
    "canon_header_0": "AAAA", "canon_header_135": "TTAA",
}
out["oligocgr_one"] = {
    "source": "composition/src/oligocgr.rs:200-220",
    "seq": "aaaatgatgaaatagagagactttattaa", "k": 4, "vecsize": 16,
    "first_point": [0.5, 0.5], "norm_first_freq_den": 26, "unnorm_first_freq": 1.0,
}
out["reader"] = {
    "source": "ktio/src/seq.rs:165-233",
    "n_records": 2, "total_length": 144,
    "fq_ids": ["Read_1", "Read_2"], "fa_ids": ["Record_1", "Record_2"],
    "seqs": [
        "GGGTGATGGCCGCTGCCGATGGCGTCAAATCCCACCAAGTTACCCTTAACAACTTAAGGGTTTTCAAATAGA",
        "GTTCAGGGATACGACGTTTGTATTTTAAGAATCTGAAGCAGAAGTCGATGATAATACGCGTCGTTTTATCAT",
    ],
}

path = pathlib.Path(__file__).parent / "kat.json"
path.write_text(json.dumps(out, indent=1) + "\n")
print("wrote", path)
