"""Row 8f-2: PAF records and CIGAR emission."""
import numpy as np
import pytest

from nanorepeat_amd import paf as P, synth


def test_paf_parser_matches_reference(golden_1d):
    for c in golden_1d["paf"]:
        p = P.PAF(c["line"].split("\t"))
        for k, v in c["fields"].items():
            assert getattr(p, k) == v, (c["line"], k)
    with pytest.raises(ValueError):
        P.PAF(["a"] * 5)


def test_paf_writer_round_trips_and_feeds_the_window_rescoring(oracle):
    rng = np.random.default_rng(3)
    t = synth.rand_seq(rng, 400)
    q = synth.apply_errors(rng, t[120:330], "ont")
    for strand, s in (("+", q), ("-", synth.revcomp(q))):
        r = oracle.align_cigar(q, t)
        line = P.format_paf_line("rd", len(s), r["qstart"], r["qend"], strand, "17", len(t), r["tstart"], r["tend"],
                                 r["score"], r["cigar"])
        p = P.PAF(line.split("\t"))
        assert (p.qstart, p.qend, p.tstart, p.tend, p.align_score, p.cigar, p.is_primary) == \
               (r["qstart"], r["qend"], r["tstart"], r["tend"], r["score"], r["cigar"], True)
        nm, block = P.cigar_counts(r["cigar"])
        assert p.n_match == nm and p.align_len == block and int(p.tname) == 17
        # the record carries what tk.target_region_alignment_stats_from_cigar needs (tk.py:435-500)
        assert oracle.cigar_region_score(p.cigar, p.tstart, p.tend, 150, 300)[0] > 0


@pytest.mark.gpu
def test_cigar_emission_matches_oracle_traceback(capi, oracle):
    rng = np.random.default_rng(21)
    seqs, pq, pt = [], [], []
    for i in range(14):
        L, R = synth.rand_seq(rng, int(rng.integers(30, 300))), synth.rand_seq(rng, int(rng.integers(30, 300)))
        u = ["TATTG", "CAG", "AT", "GGCCCC"][i % 4]; k = int(rng.integers(0, 40))
        t = L + u * k + R
        q = synth.apply_errors(rng, L[-60:] + u * int(rng.integers(0, 40)) + R[:60], "ont" if i % 2 else "hifi")
        if i == 5:
            q = q[:30] + "NNN" + q[33:]
        seqs += [q, t]; pq.append(2 * i); pt.append(2 * i + 1)
    seqs += ["", synth.rand_seq(rng, 50)]
    pq += [28, 29, 0]; pt += [1, 3, 29]
    for over in ({}, {"min_dp_score": 0}):
        g = capi.align_pairs_cigar(seqs, pq, pt, sc=capi.default_scoring(**over))
        for i, (a, b) in enumerate(zip(pq, pt)):
            o = oracle.align_cigar(seqs[a], seqs[b], sc=oracle.default_scoring(**over)) if seqs[a] and seqs[b] else dict(score=0)
            lo = max(1, over.get("min_dp_score", 80))
            if o["score"] < lo:
                assert g["score"][i] == -1 and g["cigar"][i] == ""
                continue
            got = (int(g["score"][i]), g["cigar"][i], int(g["tstart"][i]), int(g["tend"][i]), int(g["qstart"][i]), int(g["qend"][i]))
            want = (o["score"], o["cigar"], o["tstart"], o["tend"], o["qstart"], o["qend"])
            assert got == want, (i, got, want)
        # the extents agree with the payload kernel's
        p = capi.align_pairs(seqs, pq, pt, sc=capi.default_scoring(**over))
        for k in ("score", "tstart", "tend"):
            assert np.array_equal(p[k], g[k]), k
