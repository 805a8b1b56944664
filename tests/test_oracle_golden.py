"""Pins the CPU oracle (oracle/kt_oracle.c) to every fixture / known-answer test the
reference holds for the hot path.  CPU only."""
import numpy as np


def _reads(oracle, golden, name="reads.fq"):
    recs = oracle.read_records(golden / name)
    return oracle.to_csr([s for _, s in recs]), recs


def test_nt4_table(oracle):
    # kmer/src/kmer.rs:6-15
    valid = {ord(c): v for c, v in zip("ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3])}
    valid.update({0: 0, 1: 1, 2: 2, 3: 3})
    for c in range(256):
        assert oracle.nt4(c) == valid.get(c, 4), c


def test_kmer_pairs_inline(oracle, kat):
    for case in kat["kmers"]:
        f, r, _ = oracle.kmers(case["seq"], case["k"])
        assert [[int(a), int(b)] for a, b in zip(f, r)] == case["pairs"], case["source"]


def test_rev_comp_inline(oracle, kat):
    for c in kat["rev_comp"]:
        assert oracle.rev_comp(c["kmer"], c["k"]) == c["rc"]


def test_pos_map_k4(oracle, kat):
    m, pk, cnt = oracle.pos_maps(4)
    e = kat["pos_map_k4"]
    assert cnt == e["count"] and len(pk) == e["count"]
    assert int((m > 0).sum()) == e["nonzero_entries"]
    assert int(m.max()) < cnt
    for idx, v in e["entries"].items():
        assert int(m[int(idx)]) == v


def test_kcount_formula(oracle):
    # comment at kmer/src/kmer.rs:55
    for k, want in [(3, 32), (4, 136), (5, 512), (6, 2080), (7, 8192)]:
        assert oracle.pos_maps(k)[2] == want


def test_numeric_roundtrip(oracle, kat):
    for v, k, s in kat["numeric"]["to_acgt"]:
        assert oracle.numeric_to_kmer(v, k) == s
    for s, f, r in kat["numeric"]["to_numeric"]:
        assert oracle.kmer_to_numeric(s) == (f, r)


def test_py_kmers(oracle, kat):
    c = kat["py_kmers"]
    f, _, _ = oracle.kmers(c["seq"], c["k"])
    assert [oracle.numeric_to_kmer(int(x), c["k"]) for x in f] == c["fwd_acgt"]


def test_k31_canonical_kat(oracle, kat):
    c = kat["k31"]
    f, r, _ = oracle.kmers(c["seq"], 31)
    got = [oracle.numeric_to_kmer(int(min(a, b)), 31) for a, b in zip(f, r)]
    assert got == c["canonical_kmers_in_order"]
    assert int(min(f[0], r[0])) == 0x0E6336CA6D8E889C


def test_oligo_one_inline(oracle, kat):
    c = kat["oligo_one"]
    raw = oracle.oligo_one(c["seq"], 4, count_min=False)
    assert len(raw) == c["raw_len"]
    h = oracle.header(4, False)
    assert h[0] == c["raw_header_first"] and h[-1] == c["raw_header_last"] and len(h) == len(raw)
    assert oracle.oligo_one(c["seq"], 4, True, True)[0] == c["canon_norm_v0"]
    un = oracle.oligo_one(c["seq"], 4, True, False)
    assert un[0] == c["canon_unnorm_v0"] and un.sum() == c["canon_unnorm_sum"]
    hc = oracle.header(4, True)
    assert hc[0] == c["canon_header_0"] and hc[135] == c["canon_header_135"] and len(hc) == 136


def test_oligo_files(oracle, golden):
    (bases, offsets), _ = _reads(oracle, golden)
    norm = oracle.oligo_batch(bases, offsets, 4, True, True)
    assert oracle.oligo_text(norm, True) == (golden / "expected_fa.kmers").read_bytes()
    cnt = oracle.oligo_batch(bases, offsets, 4, True, False)
    assert oracle.oligo_text(cnt, False) == (golden / "expected_fa_batch_unnorm.kmers").read_bytes()
    assert oracle.oligo_text(norm, True, header_line=oracle.header(4, True)) == \
        (golden / "expected_fa_header.kmers").read_bytes()
    # threaded worker-pull path gives identical rows (oligo.rs:327-368 repeats with 8 threads)
    for _ in range(4):
        assert np.array_equal(oracle.oligo_batch(bases, offsets, 4, True, True, threads=8), norm)


def test_oligo_python_surface(oracle, golden):
    # tests/test_oligo.py:8-25: round(x, 6) equality against expected_fa.kmers
    (bases, offsets), _ = _reads(oracle, golden)
    got = oracle.oligo_batch(bases, offsets, 4)
    truth = [list(map(float, ln.split())) for ln in (golden / "expected_fa.kmers").read_text().splitlines()]
    for g, t in zip(got, truth):
        assert [round(float(x), 6) for x in g] == t


def test_py_raw_total_quirk(oracle):
    # pybindings/src/oligo.rs:61 -> normalised raw vectors sum to 0.5
    v = oracle.oligo_one("ACGTACGTAC", 3, count_min=False, norm=True, total_step=2.0)
    assert abs(v.sum() - 0.5) < 1e-15
    v1 = oracle.oligo_one("ACGTACGTAC", 3, count_min=False, norm=True, total_step=1.0)
    assert abs(v1.sum() - 1.0) < 1e-15


def test_oligocgr(oracle, kat, golden):
    c = kat["oligocgr_one"]
    xy = oracle.cgr_coords(4, 16)
    assert list(xy[0]) == c["first_point"]
    assert oracle.oligo_one(c["seq"], 4, True, True)[0] == 1.0 / c["norm_first_freq_den"]
    assert oracle.oligo_one(c["seq"], 4, True, False)[0] == c["unnorm_first_freq"]
    (bases, offsets), _ = _reads(oracle, golden)
    cnt = oracle.oligo_batch(bases, offsets, 4, True, False)
    assert oracle.oligocgr_text(cnt, xy) == (golden / "expected_reads.k4.cgr").read_bytes()


def test_counter_k15_file(oracle, golden):
    (bases, offsets), _ = _reads(oracle, golden)
    keys, counts = oracle.count_reads(bases, offsets, 15)
    want = sorted((golden / "expected_counts.part_0_chunk_0").read_text().splitlines())
    assert oracle.counts_lines(keys, counts) == want
    # partitioned + threaded restatement agrees (n_parts only changes placement)
    k2, c2 = oracle.count_reads(bases, offsets, 15, n_parts=3, threads=4)
    assert np.array_equal(keys, k2) and np.array_equal(counts, c2)


def test_counter_merge_fixtures(oracle, golden):
    ctr = oracle.Counter(2)
    for f in sorted((golden / "computed_counts_test").iterdir()):
        rows = [ln.split("\t") for ln in f.read_text().splitlines() if ln.strip()]
        ctr.add_pairs([int(a) for a, _ in rows], [int(b) for _, b in rows])
    keys, counts = ctr.export()
    assert oracle.counts_lines(keys, counts) == \
        sorted((golden / "expected_counts_test.counts").read_text().splitlines())
    assert oracle.counts_lines(keys, counts, k=15, acgt=True) == \
        sorted((golden / "expected_counts_acgt_test.counts").read_text().splitlines())


def test_counter_merge_through_files(oracle, golden, tmp_path):
    """the on-disk restatement (kto_merge) over the reference's own chunk files, and a spill -> merge round trip"""
    import shutil
    for acgt, want in [(False, "expected_counts_test.counts"), (True, "expected_counts_acgt_test.counts")]:
        d = tmp_path / ("m%d" % acgt)
        shutil.copytree(golden / "computed_counts_test", d)
        n = oracle.merge_files(d, 2, 2, threads=2, acgt=acgt, k=15, delete=True)   # counter/src/lib.rs:279-311
        got = sorted((d / "kmers.counts").read_text().splitlines())
        assert got == sorted((golden / want).read_text().splitlines()) and n == len(got)
        assert [f.name for f in d.iterdir()] == ["kmers.counts"]                     # temp files removed (:208-210)
    bases, offsets = oracle.to_csr([r[1] for r in oracle.read_records(golden / "reads.fq")])
    d = tmp_path / "rt"
    d.mkdir()
    for chunk in range(2):                                                            # the same reads as two chunks
        c = oracle.Counter(3)
        c.add_reads(bases, offsets, 15, threads=2)
        c.spill(d, chunk, threads=2)
    # chunk 0 of a 1-partition spill of reads.fq is the reference's own fixture (counter/src/lib.rs:260-276)
    c1 = oracle.Counter(1)
    c1.add_reads(bases, offsets, 15)
    d1 = tmp_path / "one"
    d1.mkdir()
    c1.spill(d1, 0)
    assert sorted((d1 / "temp_kmers.part_0_chunk_0").read_text().splitlines()) == \
        sorted((golden / "expected_counts.part_0_chunk_0").read_text().splitlines())
    oracle.merge_files(d, 3, 2, threads=2)
    keys, counts = oracle.count_reads(bases, offsets, 15)
    assert sorted((d / "kmers.counts").read_text().splitlines()) == oracle.counts_lines(keys, counts * 2)


def test_reader_fixtures(oracle, kat, golden):
    e = kat["reader"]
    for name, ids in [("reads.fq", e["fq_ids"]), ("reads.fa", e["fa_ids"]), ("reads.fq.gz", e["fq_ids"])]:
        recs = oracle.read_records(golden / name)
        assert [r[0] for r in recs] == ids
        assert [r[1].decode() for r in recs] == e["seqs"]
        assert sum(len(r[1]) for r in recs) == e["total_length"]


def test_display_format(oracle):
    f = oracle.fmt_display
    assert f(16.0) == "16" and f(0.5) == "0.5" and f(1 / 26) == "0.038461538461538464"
    assert f(1e-7) == "0.0000001" and f(1e21) == "1000000000000000000000" and f(0.0) == "0"
    assert f(1.5e-5) == "0.000015"


def test_coverage_fixtures(oracle, golden):
    # coverage/src/lib.rs:196-242: k=4, bin_size 2, bin_count 3 on reads.fq
    (bases, offsets), _ = _reads(oracle, golden)
    ctr = oracle.Counter(1)
    ctr.add_reads(bases, offsets, 4)
    norm = ctr.cov_batch(bases, offsets, 4, 2, 3, True)
    assert oracle.oligo_text(norm, True) == (golden / "expected_counts.vectors").read_bytes()
    cnt = ctr.cov_batch(bases, offsets, 4, 2, 3, False)
    assert oracle.oligo_text(cnt, False) == (golden / "expected_counts_unnorm.vectors").read_bytes()


def test_whole_sequence_cgr(oracle, golden):
    # composition/src/cgr.rs:153-186 inline KAT (vecsize 1)
    want = [(0.25, 0.25), (0.625, 0.125), (0.8125, 0.5625), (0.40625, 0.28125), (0.703125, 0.140625),
            (0.8515625, 0.5703125), (0.42578125, 0.28515625), (0.212890625, 0.142578125),
            (0.1064453125, 0.0712890625), (0.55322265625, 0.03564453125), (0.276611328125, 0.017822265625),
            (0.6383056640625, 0.5089111328125), (0.31915283203125, 0.25445556640625),
            (0.659576416015625, 0.627227783203125), (0.3297882080078125, 0.3136138916015625),
            (0.6648941040039062, 0.6568069458007812), (0.3324470520019531, 0.3284034729003906),
            (0.16622352600097656, 0.6642017364501953), (0.5831117630004883, 0.33210086822509766),
            (0.7915558815002441, 0.16605043411254883), (0.8957779407501221, 0.08302521705627441),
            (0.44788897037506104, 0.04151260852813721), (0.7239444851875305, 0.020756304264068604)]
    got = oracle.cgr_points("atgatgaaatagagagactttat", 1)
    assert [tuple(map(float, p)) for p in got] == want
    # :189-199 file fixture: reads.fq, vecsize 1
    recs = oracle.read_records(golden / "reads.fq")
    assert oracle.cgr_text([oracle.cgr_points(s, 1) for _, s in recs]) == (golden / "expected_reads.cgr").read_bytes()
    import pytest
    with pytest.raises(ValueError):
        oracle.cgr_points("ACGNT", 1)
    with pytest.raises(ValueError):
        oracle.cgr_points(b"AC\x01T", 1)   # raw codes are not CGR letters (cgr.rs:19-30)


def test_minimisers(oracle, golden):
    # kmer/src/minimiser.rs:183-280 (w=31, m=7) and :283-305 (an N inside, w=8, m=5)
    seq = ("ATGCGATATCGTAGGCGTCGATGGAGAGCTAGATCGATCGATCTAAATCCCGATCGATTCCGAGCGCGATCAAAGCGCGATAGGCTAGCTAAAG"
           "CTAGCA")
    got = oracle.minimisers(seq, 31, 7)
    want = [("ACGATAT", "ATGCGATATCGTAGGCGTCGATGGAGAGCTAGATCG"), ("ACGCCTA", "TATCGTAGGCGTCGATGGAGAGCTAGATCGATCGAT"),
            ("AGAGCTA", "AGGCGTCGATGGAGAGCTAGATCGATCGATCTAAATCC"),
            ("AAATCCC", "ATGGAGAGCTAGATCGATCGATCTAAATCCCGATCGATTCCGAGCGCGATCAAAG"),
            ("AATCCCG", "AATCCCGATCGATTCCGAGCGCGATCAAAGC"), ("AATCGAT", "ATCCCGATCGATTCCGAGCGCGATCAAAGCG"),
            ("AAAGCGC", "TCCCGATCGATTCCGAGCGCGATCAAAGCGCGATAGGCTAGCTAAAGCTAGCA")]
    assert [(oracle.numeric_to_kmer(k, 7), seq[s:e]) for k, s, e in got] == want
    seq2 = "ATGCGATATCGNTAGGCGTCGATGGA"
    got2 = [(seq2[s:e], oracle.numeric_to_kmer(k, 5)) for k, s, e in oracle.minimisers(seq2, 8, 5)]
    assert got2 == [("ATGCGATA", "ATCGC"), ("TGCGATATCG", "ATATC"), ("TAGGCGTCGA", "ACGCC"), ("GCGTCGATGGA", "ATCGA")]
    # misc/src/minimisers.rs:166-187: file fixtures (tests sort the lines)
    # compared like the reference's own test: every line trimmed, then sorted (ktio/src/fops.rs:15-25); the
    # program itself ends each s2m line with "\t\n" (join of [.., "\n"], misc/src/minimisers.rs:131-141)
    def trimmed(lines):
        return sorted(ln.strip() for ln in lines)
    recs = [(i, s.decode()) for i, s in oracle.read_records(golden / "reads.fq")]
    s2m = oracle.seq_to_min_lines(recs, 31, 7)
    assert all(ln.endswith("\t\n") for ln in s2m)
    assert trimmed(s2m) == trimmed((golden / "expected_seq_minimisers").read_text().splitlines())
    assert trimmed(oracle.bin_sequences_lines(recs, 0, 10)) == \
        trimmed((golden / "expected_minimisers").read_text().splitlines())
