#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the reference's own Python.

Run in the build container only (needs /root/reference):

    PYTHONPATH=/root/reference/src python3 -B tests/golden/make_golden.py [1d 2d wide upstream io phasing joint_round1]

The reference's drivers import pysam / pyminimap2 / Levenshtein, none of which exist
offline; empty placeholder modules are registered for them so the pure
Python around the aligner becomes callable.  Where a driver calls the aligner
(`pymm2.main(cmd)`), a stand-in is installed that either records the call (routing
fixtures) or answers with PAF text produced by this repo's CPU oracle (end-to-end
fixtures: they pin everything AROUND the aligner -- window rule, bank construction,
selectors, grid routing, CIGAR window rescoring, tie averaging -- not the aligner).

Fixtures hold inputs and expected outputs only; no reference source is copied.
"""
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

for _name in ("pysam", "pyminimap2", "Levenshtein"):
    sys.modules.setdefault(_name, types.ModuleType(_name))

import numpy as np  # noqa: E402
from NanoRepeat import tk, paf as ref_paf, repeat_region as ref_rr  # noqa: E402
from NanoRepeat import nanoRepeat_bam as ref_bam, nanoRepeat_joint as ref_joint  # noqa: E402
from NanoRepeat import split_alleles as ref_split  # noqa: E402
from oracle import oracle as O  # noqa: E402

tk.eprint = lambda *a, **k: None
ref_joint.tk.eprint = tk.eprint
RNG = random.Random(20260116)


def rand_seq(n, rng=RNG):
    return "".join(rng.choice("ACGT") for _ in range(n))


def mutate(s, sub, ins, dele, rng=RNG):
    out = []
    for c in s:
        x = rng.random()
        if x < dele:
            continue
        if x < dele + sub:
            out.append(rng.choice([b for b in "ACGT" if b != c]))
        else:
            out.append(c)
        if rng.random() < ins:
            out.append(rng.choice("ACGT"))
    return "".join(out)


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def fnum(x):
    return None if x is None else float(x)


# --------------------------------------------------------------------------- aligner stand-ins
class Recorder:
    """pymm2.main stand-in: records each call; optionally answers through `answer`."""

    def __init__(self, answer=None):
        self.calls = []
        self.answer = answer

    def __call__(self, cmd):
        toks = cmd.split()
        rec = {"cmd": cmd}
        files = [t for t in toks if os.path.exists(t)]
        rec["files"] = files
        self.calls.append(rec)
        if self.answer:
            return self.answer(cmd, toks, rec)
        return "", ""


def read_fasta(path):
    recs, name, seq = [], None, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith(">"):
            if name is not None:
                recs.append((name, "".join(seq)))
            name, seq = line[1:], []
        else:
            seq.append(line)
    if name is not None:
        recs.append((name, "".join(seq)))
    return recs


def read_fastx_names_seqs(path):
    lines = open(path).read().split("\n")
    out = []
    if lines and lines[0].startswith(">"):
        return read_fasta(path)
    for i in range(0, len(lines) - 3, 4):
        if lines[i].startswith("@"):
            out.append((lines[i][1:].split()[0], lines[i + 1]))
    return out


def paf_line(qname, qlen, strand, tname, tlen, r):
    nm = sum(int(x[:-1]) for x in __import__("re").findall(r"\d+=", r["cigar"]))
    alen = sum(int(x[:-1]) for x in __import__("re").findall(r"\d+[=XID]", r["cigar"]))
    if strand == "+":
        qs, qe = r["qstart"], r["qend"]
    else:
        qs, qe = qlen - r["qend"], qlen - r["qstart"]
    return "\t".join(map(str, [qname, qlen, qs, qe, strand, tname, tlen, r["tstart"], r["tend"],
                               nm, alen, 60, "tp:A:P", f"AS:i:{r['score']}", f"cg:Z:{r['cigar']}"]))


def oracle_aligner(min_score=80, both_strands=False):
    """Answers an aligner call with PAF text from the CPU oracle (optimal local DP)."""

    def answer(cmd, toks, rec):
        tfile, qfile = rec["files"][-2], rec["files"][-1]
        out = []
        for qname, qseq in read_fastx_names_seqs(qfile):
            for tname, tseq in read_fasta(tfile):
                best = None
                for strand, s in (("+", qseq), ("-", revcomp(qseq))) if both_strands else (("+", qseq),):
                    r = O.align_cigar(s, tseq)
                    if r["score"] >= min_score and (best is None or r["score"] > best[1]["score"]):
                        best = (strand, r)
                if best:
                    out.append(paf_line(qname, len(qseq), best[0], tname, len(tseq), best[1]))
        text = "\n".join(out) + ("\n" if out else "")
        if "-o" in toks:
            with open(toks[toks.index("-o") + 1], "a") as f:
                f.write(text)
            return "", ""
        return text, ""

    return answer


# --------------------------------------------------------------------------- 1D fixtures
def make_region(left, unit, right, tmp):
    rr = ref_rr.RepeatRegion()
    rr.left_anchor_seq, rr.right_anchor_seq = left, right
    rr.left_anchor_len, rr.right_anchor_len = len(left), len(right)
    rr.repeat_unit_seq = unit
    rr.chrom, rr.start_pos, rr.end_pos = "chrT", 1000, 1100
    rr.temp_out_dir = tmp
    rr.read_dict = dict()
    rr.read_core_seq_dict = dict()
    return rr


def add_read(rr, name, core, r2):
    rd = ref_rr.Read()
    rd.read_name = name
    rd.round2_repeat_size = r2
    rr.read_dict[name] = rd
    rr.read_core_seq_dict[name] = core


def gen_1d(tmp):
    fx = {}
    # -- window rule + bank content (nanoRepeat_bam.py:452-500) with a recording stand-in
    left, unit, right = rand_seq(40), "TATTG", rand_seq(40)
    cases = []
    for r2, fast in [(19.4, False), (19.9, False), (3.0, False), (0.0, False), (14.99, False),
                     (15.0, False), (299.99, False), (300.0, False), (319.9, False), (400.0, False),
                     (2999.0, False), (3000.0, False), (4000.0, False), (4000.0, True),
                     (19.4, True), (7.5, True), (None, False), (123.456, False), (1e-9, False)]:
        d = os.path.join(tmp, "w"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        rr = make_region(left, unit, right, d)
        add_read(rr, "r0", "ACGT" * 10, r2)
        rec = Recorder()
        ref_bam.pymm2.main = rec
        ref_bam.round3_align(4, fast, rr, "ont_q20")
        c = {"r2": r2, "fast_mode": fast, "n_calls": len(rec.calls)}
        if rec.calls:
            bank = read_fasta(rec.calls[0]["files"][0])
            ks = [int(n) for n, _ in bank]
            c.update(kmin=ks[0], kmax=ks[-1], n_templates=len(ks), contiguous=ks == list(range(ks[0], ks[-1] + 1)),
                     bank_ok=all(s == left + unit * k + right for (n, s), k in zip(bank, ks)),
                     cmd_flags=rec.calls[0]["cmd"].split(rec.calls[0]["files"][0])[0])
            rd = read_fasta(rec.calls[0]["files"][1])
            c["read_fasta"] = rd
        cases.append(c)
    fx["window_rule"] = {"left": left, "unit": unit, "right": right, "cases": cases}

    # -- selector on canned PAF records (nanoRepeat_bam.py:408-434)
    sel = []
    rng = random.Random(7)

    def run_selector(ll, rl, r2, recs):
        rr = make_region("A" * ll, "TATTG", "C" * rl, tmp)
        add_read(rr, "x", "ACGT", r2)
        lines = []
        for (k, AS, ts, te, tl) in recs:
            lines.append("\t".join(map(str, ["x", 500, 0, 500, "+", k, tl, ts, te, 400, 500, 60,
                                             "tp:A:P", f"AS:i:{AS}", "cg:Z:500="])))
        rr.read_dict["x"].round3_paf_text = "\n".join(lines) + "\n" if lines else ""
        ref_bam.round3_estimation_from_alignment(rr)
        return fnum(rr.read_dict["x"].round3_repeat_size)

    fixed = [
        (1000, 1000, 19.4, [(19, 560, 905, 1190, 2095), (20, 572, 905, 1195, 2100),
                            (21, 572, 905, 1200, 2105), (22, 580, 1001, 1205, 2110)]),
        (1000, 1000, 19.4, [(19, 560, 905, 1190, 2095), (20, 572, 905, 1195, 2100),
                            (21, 572, 905, 1200, 2105)]),
        (1000, 1000, 5.5, []),
        (1000, 1000, 5.5, [(5, 100, 999, 1026, 2025)]),
        (1000, 1000, 5.5, [(5, 100, 1000, 1026, 2025)]),
        (1000, 1000, 5.5, [(5, 100, 999, 1025, 2025)]),
        (1000, 1000, 5.5, [(5, 100, 999, 1026, 2025), (6, 100, 999, 1031, 2030), (7, 100, 999, 1036, 2035)]),
    ]
    for ll, rl, r2, recs in fixed:
        sel.append({"left_len": ll, "right_len": rl, "r2": r2, "records": recs,
                    "result": run_selector(ll, rl, r2, recs)})
    for _ in range(200):
        ll, rl = rng.choice([50, 1000]), rng.choice([50, 1000])
        r2 = round(rng.uniform(0, 60), 2)
        recs = []
        base = rng.randint(80, 900)
        for k in rng.sample(range(0, 60), rng.randint(0, 8)):
            AS = base - rng.choice([0, 0, 0, 2, 6, 10])
            tl = ll + 5 * k + rl
            ts = rng.choice([ll - 1, ll, ll + 1, max(0, ll - 40), 0])
            te = rng.choice([ll + 5 * k, ll + 5 * k + 1, tl, tl - 1, ll + 5 * k + 20])
            te = min(te, tl)
            recs.append((k, AS, ts, te, tl))
        sel.append({"left_len": ll, "right_len": rl, "r2": r2, "records": recs,
                    "result": run_selector(ll, rl, r2, recs)})
    fx["selector"] = sel

    # -- PAF record parsing (paf.py:32-79)
    pafs = []
    for line in [
        "rd1\t500\t10\t480\t+\t17\t2085\t905\t1380\t450\t480\t60\ttp:A:P\tcm:i:5\tAS:i:812\tcg:Z:470=",
        "rd2\t500\t10\t480\t-\t9-4\t2085\t905\t1380\t450\t480\t0\ttp:A:S\tAS:i:-3\tcg:Z:30=2I70=",
        "rd3\t123\t0\t123\t+\t0\t2000\t940\t1063\t123\t123\t60",
    ]:
        p = ref_paf.PAF(line.split("\t"))
        pafs.append({"line": line, "fields": {k: getattr(p, k) for k in
                     ("qname", "qlen", "qstart", "qend", "strand", "tname", "tlen", "tstart", "tend",
                      "n_match", "align_len", "mapq", "is_primary", "align_score", "cigar")}})
    fx["paf"] = pafs

    # -- end to end: reference round3_estimation with the oracle answering the aligner
    e2e = []
    for case_id, (unit, alleles, errs, nreads, flank, fast) in enumerate([
        ("TATTG", (8, 21), (0.02, 0.01, 0.02), 10, 60, False),
        ("CAG", (12, 30), (0.03, 0.02, 0.03), 10, 50, True),
        ("AT", (0, 9), (0.01, 0.01, 0.01), 6, 60, False),
        ("GGCCCC", (5, 14), (0.0, 0.0, 0.0), 4, 60, False),
    ]):
        left, right = rand_seq(150), rand_seq(150)
        d = os.path.join(tmp, f"e{case_id}"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        rr = make_region(left, unit, right, d)
        reads = []
        for i in range(nreads):
            kt = alleles[i % 2]
            core = mutate(left[-flank:] + unit * kt + right[:flank], *errs)
            r2 = max(0.0, kt + RNG.choice([-1.2, -0.4, 0.0, 0.3, 1.7]))
            if i == nreads - 1:
                r2 = None                      # skipped read (nanoRepeat_bam.py:460)
            if i == nreads - 2:
                core = rand_seq(90)            # junk: nothing reaches min score -> empty PAF
            add_read(rr, f"read{i}", core, r2)
            reads.append({"name": f"read{i}", "core": core, "r2": r2})
        ref_bam.pymm2.main = Recorder(oracle_aligner(80))
        ref_bam.round3_estimation("ont", fast, rr, 4)
        res = {n: fnum(rd.round3_repeat_size) for n, rd in rr.read_dict.items()}
        # text output (split_alleles.py:536-558)
        rr.out_prefix = os.path.join(d, "out"); rr.no_details = False
        ref_split.output_repeat_size_1d(rr)
        txt = open(rr.out_prefix + ".repeat_size.txt").read()
        e2e.append({"left": left, "unit": unit, "right": right, "fast_mode": fast, "reads": reads,
                    "round3": res, "repeat_size_txt": txt, "unique_id": rr.to_unique_id()})
    fx["e2e"] = e2e
    return fx


# --------------------------------------------------------------------------- 2D fixtures
def make_repeat(chrom, start, end, unit, max_size):
    return ref_joint.Repeat().init_from_string(f"{chrom}:{start}:{end}:{unit}:{max_size}")


def gen_2d(tmp):
    fx = {}
    rng = random.Random(11)
    # -- CIGAR window rescoring (tk.py:435-500), exact-match counter (tk.py:405-431)
    cases = [("50=1X10=2I20=3D30=", 100, 214, 120, 200)]
    ops = "=XID"
    for _ in range(300):
        n = rng.randint(1, 12)
        cig, last, tl = "", "", 0
        for _ in range(n):
            op = rng.choice([o for o in ops if o != last]); last = op
            l = rng.randint(1, 30)
            cig += f"{l}{op}"
            if op in "=XD":
                tl += l
        ts = rng.randint(0, 60)
        a = rng.randint(0, ts + tl + 10); b = a + rng.randint(0, tl + 20)
        cases.append((cig, ts, ts + tl, a, b))
    # CIGARs from the oracle's own traceback on noisy repeat reads
    for _ in range(60):
        L, R = rand_seq(60, rng), rand_seq(60, rng)
        u = rng.choice(["CAG", "TATTG", "AT", "CCG"]); k = rng.randint(0, 15)
        t = L + u * k + R
        q = mutate(L[-30:] + u * rng.randint(0, 15) + R[:30], 0.04, 0.03, 0.04, rng)
        r = O.align_cigar(q, t)
        if r["score"] > 0:
            cases.append((r["cigar"], r["tstart"], r["tend"], max(0, 50), min(len(t), 60 + len(u) * k + 10)))
    out = []
    for cig, ts, te, a, b in cases:
        e = tk.target_region_alignment_stats_from_cigar(cig, ts, te, a, b)
        out.append({"cigar": cig, "tstart": ts, "tend": te, "a": a, "b": b, "score": e.score,
                    "num_match": e.num_match, "num_mismatch": e.num_mismatch,
                    "num_ins": e.num_ins, "num_del": e.num_del})
    fx["cigar_stats"] = out
    fx["exact_match"] = [{"cigar": c, "tstart": ts, "ref_start": rs, "m": m,
                          "result": tk.calculate_repeat_size_from_exact_match(c, ts, rs, m)}
                         for c, ts, rs, m in [("40=1X30=2D9=", 90, 100, 3), ("100=", 0, 50, 3),
                                              ("10=2I30=1X9=", 95, 100, 5), ("7=", 100, 100, 3)]]

    # -- step size (nanoRepeat_joint.py:351-374)
    steps = []
    for _ in range(60):
        m = rng.randint(1, 60)
        rep = ref_joint.Repeat(); rep.repeat_unit_size = m
        d = {f"r{i}": (a, a + rng.randint(0, 80)) for i, a in enumerate(rng.sample(range(0, 200), rng.randint(1, 6)))}
        steps.append({"m": m, "ranges": list(d.values()), "step": int(ref_joint.choose_best_step_size(rep, d))})
    rep = ref_joint.Repeat(); rep.repeat_unit_size = 3
    steps.append({"m": 3, "ranges": [(0, 25), (2, 30)],
                  "step": int(ref_joint.choose_best_step_size(rep, {"a": (0, 25), "b": (2, 30)}))})
    fx["step_size"] = steps

    # -- anchors + template (nanoRepeat_joint.py:480-507)
    Lf, Rf = rand_seq(1200, rng), rand_seq(1100, rng)
    mid = "CAACAGCCGCCAC"
    chrom = Lf + "CAG" * 19 + mid + "CCG" * 7 + Rf
    r1 = make_repeat("chr4", len(Lf), len(Lf) + 57, "CAG", 200)
    r2 = make_repeat("chr4", len(Lf) + 57 + 13, len(Lf) + 57 + 13 + 21, "CCG", 20)
    la, ma, ra = ref_joint.extract_anchor_seq_for_two_repeats(chrom, r1, r2, 1000)
    tf = os.path.join(tmp, "t.fa")
    ref_joint.build_fasta_template_for_two_repeats(la, ma, ra, r1, r2, 3, 2, tf)
    fx["template"] = {"chrom_sha1": hashlib.sha1(chrom.encode()).hexdigest(),
                      "left_len": len(la), "mid": ma, "right_len": len(ra),
                      "left_tail": la[-20:], "right_head": ra[:20],
                      "template_k3_k2": open(tf).read().replace(la, "<L>").replace(ra, "<R>")}

    # -- selector on canned PAF (nanoRepeat_joint.py:427-478)
    pf = os.path.join(tmp, "c.paf")
    lines = [
        "x\t300\t0\t100\t+\t10-5\t2060\t950\t1050\t100\t100\t60\tAS:i:200\tcg:Z:100=",
        "x\t300\t0\t100\t+\t11-5\t2063\t950\t1050\t100\t100\t60\tAS:i:180\tcg:Z:100=",
        "x\t300\t0\t100\t+\t12-6\t2069\t950\t1050\t99\t100\t60\tAS:i:190\tcg:Z:60=1X39=",
        "y\t300\t0\t102\t-\t7-3\t2045\t960\t1060\t100\t102\t60\tAS:i:170\tcg:Z:30=2I70=",
        "y\t300\t0\t102\t-\t8-3\t2048\t960\t1063\t100\t105\t60\tAS:i:160\tcg:Z:30=3D70=",
    ]
    open(pf, "w").write("\n".join(lines) + "\n")
    rp1 = ref_joint.Repeat(); rp1.repeat_unit_size = 3
    rp2 = ref_joint.Repeat(); rp2.repeat_unit_size = 3
    est = ref_joint.estimate_two_repeats_from_paf(pf, 1000, 13, rp1, rp2)
    fx["selector"] = {"lines": lines, "left_len": 1000, "mid_len": 13, "m1": 3, "m2": 3,
                      "k1": {k: fnum(v) for k, v in est.repeat1_count_dict.items()},
                      "k2": {k: fnum(v) for k, v in est.repeat2_count_dict.items()}}

    # -- grid routing with a recording stand-in (nanoRepeat_joint.py:376-425, 275-349)
    def routing(ranges1, ranges2, maxs, r2sizes=None, step=None):
        d = os.path.join(tmp, "rt"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        init = ref_joint.Round1Estimation()
        init.repeat1_count_range_dict = dict(ranges1)
        init.repeat2_count_range_dict = dict(ranges2)
        fq = {n: f"@{n}\nACGT\n+\n!!!!\n" for n in ranges1}
        a = make_repeat("chr4", len(Lf), len(Lf) + 57, "CAG", maxs[0])
        b = make_repeat("chr4", len(Lf) + 70, len(Lf) + 91, "CCG", maxs[1])
        a.round1_min_size = min(v[0] for v in ranges1.values()); a.round1_max_size = min(maxs[0], max(v[1] for v in ranges1.values()))
        b.round1_min_size = min(v[0] for v in ranges2.values()); b.round1_max_size = min(maxs[1], max(v[1] for v in ranges2.values()))
        rec = Recorder(lambda cmd, toks, r: (r.update(reads=[n for n, _ in read_fastx_names_seqs(r["files"][-1])],
                                                     tname=read_fasta(r["files"][-2])[0][0]), ("", ""))[1])
        ref_joint.pymm2.main = rec
        if r2sizes is None:
            est = ref_joint.round2_estimation_of_repeat_size(init, fq, chrom, a, b, "ont", 4, d)
            return {"round": 2, "ranges1": ranges1, "ranges2": ranges2,
                    "round1_min1": a.round1_min_size, "round1_max1": a.round1_max_size,
                    "round1_min2": b.round1_min_size, "round1_max2": b.round1_max_size,
                    "step1": int(est.step_size1), "step2": int(est.step_size2),
                    "calls": [[c["tname"], c["reads"]] for c in rec.calls],
                    "cmd0": rec.calls[0]["cmd"].replace(d, "<D>") if rec.calls else None}
        r2e = ref_joint.RepeatSize()
        r2e.repeat1_count_dict = {k: v[0] for k, v in r2sizes.items()}
        r2e.repeat2_count_dict = {k: v[1] for k, v in r2sizes.items()}
        r2e.step_size1, r2e.step_size2 = step
        ref_joint.round3_estimation_of_repeat_size(init, r2e, fq, chrom, a, b, "ont", 4, d)
        return {"round": 3, "ranges1": ranges1, "ranges2": ranges2, "r2sizes": r2sizes, "step": list(step),
                "calls": [[c["tname"], c["reads"]] for c in rec.calls]}

    rts = [routing({"x": (5, 30), "y": (20, 70)}, {"x": (0, 15), "y": (2, 14)}, (210, 30))]
    rts.append(routing({"x": (5, 30), "y": (20, 70)}, {"x": (0, 15), "y": (2, 14)}, (210, 30),
                       r2sizes={"x": (17.0, 10.0), "y": (54.5, 7.0)}, step=(4, 2)))
    rts.append(routing({"a": (0, 25), "b": (2, 30), "c": (10, 12)}, {"a": (3, 9), "b": (0, 4), "c": (5, 6)}, (60, 40)))
    rts.append(routing({"a": (0, 25), "b": (2, 30), "c": (10, 12)}, {"a": (3, 9), "b": (0, 4), "c": (5, 6)}, (60, 40),
                       r2sizes={"a": (1.0, 3.5), "b": (29.0, 0.0), "c": (10.5, 5.0)}, step=(3, 2)))
    fx["routing"] = rts

    # -- end to end: round 2 (+3) with the oracle answering the aligner
    e2e = []
    for cid, (u1, u2, mid_s, alleles, errs, nreads, flank) in enumerate([
        ("CAG", "CCG", "CAACAGCCGCCAC", ((17, 10), (30, 7)), (0.02, 0.01, 0.02), 6, 70),
        ("TATTG", "AC", "GGT", ((6, 12), (11, 3)), (0.01, 0.01, 0.01), 4, 70),
    ]):
        d = os.path.join(tmp, f"j{cid}"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        Lx, Rx = rand_seq(160, rng), rand_seq(160, rng)
        kref1, kref2 = 5, 4
        chrom2 = Lx + u1 * kref1 + mid_s + u2 * kref2 + Rx
        s1 = len(Lx); e1 = s1 + len(u1) * kref1; s2 = e1 + len(mid_s); e2_ = s2 + len(u2) * kref2
        a = make_repeat("chrJ", s1, e1, u1, 40); b = make_repeat("chrJ", s2, e2_, u2, 20)
        a.max_size += 10; b.max_size += 10
        init = ref_joint.Round1Estimation()
        reads = []
        for i in range(nreads):
            k1t, k2t = alleles[i % 2]
            s = mutate(Lx[-flank:] + u1 * k1t + mid_s + u2 * k2t + Rx[:flank], *errs, rng)
            strand = 1
            if i % 3 == 2:
                s = revcomp(s); strand = -1
            name = f"jr{i}"
            init.repeat1_count_range_dict[name] = (max(0, k1t - 9), k1t + 4)
            init.repeat2_count_range_dict[name] = (max(0, k2t - 6), k2t + 3)
            reads.append({"name": name, "seq": s, "strand": strand,
                          "range1": list(init.repeat1_count_range_dict[name]),
                          "range2": list(init.repeat2_count_range_dict[name])})
        fqp = os.path.join(d, "in.fastq")
        with open(fqp, "w") as f:
            for r in reads:
                f.write(f"@{r['name']}\n{r['seq']}\n+\n{'!' * len(r['seq'])}\n")
        ref_joint.pymm2.main = Recorder(oracle_aligner(80, both_strands=True))
        # fine_tune_read_count computes the global ranges, runs round 2 and (maybe) round 3
        final = ref_joint.fine_tune_read_count(init, fqp, chrom2, a, b, "ont", 4, d)
        txt_prefix = os.path.join(d, "out")
        ref_split.tk.eprint = tk.eprint
        ref_split.output_repeat_size_2d("in.fastq", a.repeat_id, b.repeat_id, txt_prefix,
                                        final.repeat1_count_dict, final.repeat2_count_dict)
        e2e.append({"chrom": chrom2, "repeat1": f"chrJ:{s1}:{e1}:{u1}:40", "repeat2": f"chrJ:{s2}:{e2_}:{u2}:20",
                    "reads": reads,
                    "round1_min1": a.round1_min_size, "round1_max1": a.round1_max_size,
                    "round1_min2": b.round1_min_size, "round1_max2": b.round1_max_size,
                    "final_step": [int(final.step_size1), int(final.step_size2)],
                    "k1": {k: fnum(v) for k, v in final.repeat1_count_dict.items()},
                    "k2": {k: fnum(v) for k, v in final.repeat2_count_dict.items()},
                    "repeat_size_txt": open(txt_prefix + ".repeat_size.txt").read()})
    fx["e2e"] = e2e
    return fx


# --------------------------------------------------------------------------- the reference's defaults at full size
def score_only_aligner(min_score=80, both_strands=False):
    """Like oracle_aligner but without the traceback (O.align: score and extents only): the 1D selector reads AS, tstart,
    tend and tlen of a record, never its CIGAR (nanoRepeat_bam.py:423-428), so long templates stay cheap."""

    def answer(cmd, toks, rec):
        tfile, qfile = rec["files"][-2], rec["files"][-1]
        out = []
        for qname, qseq in read_fastx_names_seqs(qfile):
            for tname, tseq in read_fasta(tfile):
                s, ts, te = O.align(qseq, tseq)
                if s >= min_score:
                    out.append("\t".join(map(str, [qname, len(qseq), 0, len(qseq), "+", tname, len(tseq), ts, te,
                                                    te - ts, te - ts, 60, "tp:A:P", f"AS:i:{s}", "cg:Z:1="])))
        return "\n".join(out) + ("\n" if out else ""), ""

    return answer


def gen_wide(tmp):
    """What the small e2e fixtures leave out: the reference's DEFAULT 1000-bp anchors (nanoRepeat.py:122), a read whose
    round-2 size of >= 3000 gives the full K = 301 window (buffer capped at 150, nanoRepeat_bam.py:463-472), and joint
    runs that record whether round 3 ran at all (final_step is [1, 1] either way, nanoRepeat_joint.py:268,345-346)."""
    fx = {}
    rng = random.Random(20261005)
    e2e = []
    for case_id, (unit, ks, errs, flank) in enumerate([
        ("TATTG", (8, 21, 8, 21, 35, 21), (0.02, 0.01, 0.02), 100),        # K = 31 windows, cores of 240 - 380 bases
        ("AC", (3000,), (0.004, 0.003, 0.004), 100),                       # r2 = 3001.3: window [2851, 3151], K = 301
    ]):
        left, right = rand_seq(1000, rng), rand_seq(1000, rng)
        d = os.path.join(tmp, f"w{case_id}"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        rr = make_region(left, unit, right, d)
        reads = []
        for i, kt in enumerate(ks):
            core = mutate(left[-flank:] + unit * kt + right[:flank], *errs, rng)
            r2 = kt + rng.choice([-1.2, -0.4, 0.0, 0.3, 1.3])
            if kt >= 3000:
                r2 = kt + 1.3
            add_read(rr, f"read{i}", core, r2)
            reads.append({"name": f"read{i}", "core": core, "r2": r2})
        rec = Recorder(score_only_aligner(80))
        ref_bam.pymm2.main = rec
        ref_bam.round3_estimation("ont", False, rr, 4)
        banks = [len(read_fasta(c["files"][0])) for c in rec.calls] if False else None
        res = {n: fnum(rd.round3_repeat_size) for n, rd in rr.read_dict.items()}
        windows = {r["name"]: [max(0, int(r["r2"] - min(150, max(15, int(0.05 * r["r2"]))))),
                               int(r["r2"] + min(150, max(15, int(0.05 * r["r2"]))))] for r in reads}
        rr.out_prefix = os.path.join(d, "out"); rr.no_details = False
        ref_split.output_repeat_size_1d(rr)
        e2e.append({"left": left, "unit": unit, "right": right, "fast_mode": False, "reads": reads, "round3": res,
                    "n_records": {r["name"]: len([l for l in rr.read_dict[r["name"]].round3_paf_text.split("\n") if l]) for r in reads},
                    "windows": windows,
                    "repeat_size_txt": open(rr.out_prefix + ".repeat_size.txt").read(), "unique_id": rr.to_unique_id()})
    fx["e2e_1d"] = e2e

    # -- joint: both steps > 1 (round 3 runs), and an axis whose round-1 ranges are so narrow that its step is 1 (round 3 skipped)
    e2e2 = []
    for cid, (u1, u2, mid_s, alleles, errs, nreads, flank, w1, w2) in enumerate([
        ("CAG", "CCG", "CAACAGCCGCCAC", ((17, 10), (30, 7)), (0.02, 0.01, 0.02), 6, 70, (9, 4), (6, 3)),
        ("CAG", "CCG", "CAACAGCCGCCAC", ((17, 10), (30, 7)), (0.02, 0.01, 0.02), 6, 70, (9, 4), (1, 1)),
        ("TATTG", "AC", "GGT", ((6, 12), (11, 3)), (0.01, 0.01, 0.01), 4, 70, (1, 2), (6, 3)),
    ]):
        d = os.path.join(tmp, f"wj{cid}"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        Lx, Rx = rand_seq(160, rng), rand_seq(160, rng)
        kref1, kref2 = 5, 4
        chrom2 = Lx + u1 * kref1 + mid_s + u2 * kref2 + Rx
        s1 = len(Lx); e1 = s1 + len(u1) * kref1; s2 = e1 + len(mid_s); e2_ = s2 + len(u2) * kref2
        a = make_repeat("chrJ", s1, e1, u1, 40); b = make_repeat("chrJ", s2, e2_, u2, 20)
        a.max_size += 10; b.max_size += 10
        init = ref_joint.Round1Estimation()
        reads = []
        for i in range(nreads):
            k1t, k2t = alleles[i % 2]
            sq = mutate(Lx[-flank:] + u1 * k1t + mid_s + u2 * k2t + Rx[:flank], *errs, rng)
            strand = 1
            if i % 3 == 2:
                sq = revcomp(sq); strand = -1
            name = f"jr{i}"
            init.repeat1_count_range_dict[name] = (max(0, k1t - w1[0]), k1t + w1[1])
            init.repeat2_count_range_dict[name] = (max(0, k2t - w2[0]), k2t + w2[1])
            reads.append({"name": name, "seq": sq, "strand": strand,
                          "range1": list(init.repeat1_count_range_dict[name]),
                          "range2": list(init.repeat2_count_range_dict[name])})
        fqp = os.path.join(d, "in.fastq")
        with open(fqp, "w") as f:
            for r in reads:
                f.write(f"@{r['name']}\n{r['seq']}\n+\n{'!' * len(r['seq'])}\n")
        ref_joint.pymm2.main = Recorder(oracle_aligner(80, both_strands=True))
        seen = {"round2": None, "round3_ran": False}
        r2_fn, r3_fn = ref_joint.round2_estimation_of_repeat_size, ref_joint.round3_estimation_of_repeat_size

        def spy2(*args, **kw):
            est = r2_fn(*args, **kw)
            seen["round2"] = {"step": [int(est.step_size1), int(est.step_size2)],
                              "k1": {k: fnum(v) for k, v in est.repeat1_count_dict.items()},
                              "k2": {k: fnum(v) for k, v in est.repeat2_count_dict.items()}}
            return est

        def spy3(*args, **kw):
            seen["round3_ran"] = True
            return r3_fn(*args, **kw)

        ref_joint.round2_estimation_of_repeat_size, ref_joint.round3_estimation_of_repeat_size = spy2, spy3
        try:
            final = ref_joint.fine_tune_read_count(init, fqp, chrom2, a, b, "ont", 4, d)
        finally:
            ref_joint.round2_estimation_of_repeat_size, ref_joint.round3_estimation_of_repeat_size = r2_fn, r3_fn
        e2e2.append({"chrom": chrom2, "repeat1": f"chrJ:{s1}:{e1}:{u1}:40", "repeat2": f"chrJ:{s2}:{e2_}:{u2}:20",
                     "reads": reads, "round2": seen["round2"], "round3_ran": seen["round3_ran"],
                     "final_step": [int(final.step_size1), int(final.step_size2)],
                     "k1": {k: fnum(v) for k, v in final.repeat1_count_dict.items()},
                     "k2": {k: fnum(v) for k, v in final.repeat2_count_dict.items()}})
    fx["e2e_2d"] = e2e2
    return fx


# --------------------------------------------------------------------------- upstream rows (8f-1)
def anchor_aligner(min_score=80):
    """Stand-in for the anchors-vs-reads call (nanoRepeat_bam.py:281): reads are the PAF queries,
    the two anchors the targets.  One record per (read, anchor, strand) from the CPU oracle,
    computed with the anchor as DP query so that the extents are read coordinates."""

    def answer(cmd, toks, rec):
        tfile, qfile = rec["files"][-2], rec["files"][-1]
        anchors = read_fasta(tfile)
        out = []
        for qname, qseq in read_fastx_names_seqs(qfile):
            for tname, tseq in anchors:
                for strand, s in (("+", qseq), ("-", revcomp(qseq))):
                    sc, ts, te = O.align(tseq, s)
                    if sc < min_score:
                        continue
                    qs, qe = (ts, te) if strand == "+" else (len(qseq) - te, len(qseq) - ts)
                    out.append("\t".join(map(str, [qname, len(qseq), qs, qe, strand, tname, len(tseq), 0, len(tseq),
                                                   te - ts, te - ts, 60, "tp:A:P", f"AS:i:{sc}"])))
        return "\n".join(out) + ("\n" if out else ""), ""

    return answer


def read_fields(rd):
    keys = ("read_name", "full_read_len", "left_anchor_is_good", "right_anchor_is_good", "both_anchors_are_good",
            "core_seq_start_pos", "core_seq_end_pos", "mid_seq_start_pos", "mid_seq_end_pos",
            "dist_between_anchors", "strand", "left_buffer_len", "right_buffer_len",
            "round1_repeat_size", "round2_repeat_size", "round3_repeat_size")
    return {k: (fnum(getattr(rd, k)) if k.startswith("round") else getattr(rd, k)) for k in keys}


def gen_upstream(tmp):
    fx = {}
    rng = random.Random(23)
    # -- anchor logic on canned PAF (nanoRepeat_bam.py:165-258)
    def P(q, qlen, qs, qe, st, t, AS, alen=900, mapq=60):
        return "\t".join(map(str, [q, qlen, qs, qe, st, t, 1000, 0, 1000, alen, alen, mapq, "tp:A:P", f"AS:i:{AS}"]))
    cases = [
        [P("a", 5000, 1000, 2000, "+", "left_anchor", 1800), P("a", 5000, 2100, 3100, "+", "right_anchor", 1700)],
        [P("b", 5000, 1000, 2000, "-", "left_anchor", 1800), P("b", 5000, 2100, 3100, "-", "right_anchor", 1700)],
        [P("c", 5000, 1000, 2000, "+", "left_anchor", 1800), P("c", 5000, 2100, 3100, "-", "right_anchor", 1700)],
        [P("d", 5000, 1000, 2000, "+", "left_anchor", 1800), P("d", 5000, 1985, 2985, "+", "right_anchor", 1700)],
        [P("e", 5000, 1000, 2000, "+", "left_anchor", 1800), P("e", 5000, 1991, 2991, "+", "right_anchor", 1700)],
        [P("f", 5000, 1000, 2000, "+", "left_anchor", 1800)],
        [P("g", 5000, 1000, 2000, "+", "left_anchor", 1800), P("g", 5000, 3000, 3600, "+", "left_anchor", 1300),
         P("g", 5000, 2100, 3100, "+", "right_anchor", 1700)],
        [P("h", 5000, 1000, 2000, "+", "left_anchor", 1800), P("h", 5000, 3000, 3600, "+", "left_anchor", 1100),
         P("h", 5000, 2100, 3100, "+", "right_anchor", 1700)],
        [P("i", 5000, 1000, 2000, "+", "left_anchor", 1800, mapq=20), P("i", 5000, 3000, 3600, "+", "left_anchor", 100),
         P("i", 5000, 2100, 3100, "+", "right_anchor", 1700)],
        [P("j", 2150, 40, 1040, "+", "left_anchor", 1800), P("j", 2150, 1100, 2100, "+", "right_anchor", 1700)],
        [P("k", 5000, 1000, 2000, "+", "left_anchor", 1800, alen=9), P("k", 5000, 3000, 3600, "+", "left_anchor", 100),
         P("k", 5000, 2100, 3100, "+", "right_anchor", 1700)],
    ]
    out = []
    for lines in cases:
        rr = make_region("A" * 1000, "TATTG", "C" * 1000, tmp)
        ref_bam.find_anchor_locations_from_paf(rr, "\n".join(lines) + "\n")
        out.append({"lines": lines, "reads": {n: read_fields(rd) for n, rd in rr.read_dict.items()}})
    # several reads in one PAF text (grouping by consecutive qname)
    rr = make_region("A" * 1000, "TATTG", "C" * 1000, tmp)
    allines = [l for c in cases[:6] for l in c]
    ref_bam.find_anchor_locations_from_paf(rr, "\n".join(allines) + "\n")
    out.append({"lines": allines, "reads": {n: read_fields(rd) for n, rd in rr.read_dict.items()}})
    fx["anchor_logic"] = out

    # -- end to end: steps 1-3 of quantify1repeat_from_bam (:656-679) with the oracle as aligner
    e2e = []
    for cid, (unit, alleles, errs, nreads, anchor_len) in enumerate([
        ("TATTG", (7, 19), (0.02, 0.01, 0.02), 8, 300),
        ("CAG", (11, 33), (0.03, 0.02, 0.03), 6, 250),
    ]):
        d = os.path.join(tmp, f"u{cid}"); shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        left, right = rand_seq(anchor_len, rng), rand_seq(anchor_len, rng)
        far_l, far_r = rand_seq(400, rng), rand_seq(400, rng)
        rr = make_region(left, unit, right, d)
        rr.region_fq_file = os.path.join(d, "region.fastq")
        reads = []
        with open(rr.region_fq_file, "w") as f:
            for i in range(nreads):
                kt = alleles[i % 2]
                s = mutate(far_l[rng.randint(0, 300):] + left + unit * kt + right + far_r[:rng.randint(100, 400)], *errs, rng)
                if i == nreads - 1:
                    s = mutate(far_l + left + unit * kt, *errs, rng)        # no right anchor: read rejected
                if i % 3 == 1:
                    s = revcomp(s)
                f.write(f"@rd{i}\n{s}\n+\n{'I' * len(s)}\n")
                reads.append({"name": f"rd{i}", "seq": s})
        ref_bam.pymm2.main = Recorder(anchor_aligner(80))
        ref_bam.find_anchor_locations_in_reads("ont", rr, 4)
        ref_bam.make_core_seq_fastq(rr)
        ref_bam.pymm2.main = Recorder(oracle_aligner(80))
        ref_bam.round1_and_round2_estimation("ont", rr, 4)
        ref_bam.round3_estimation("ont", False, rr, 4)
        e2e.append({"left": left, "unit": unit, "right": right, "reads": reads,
                    "read_dict": {n: read_fields(rd) for n, rd in rr.read_dict.items()},
                    "core": dict(rr.read_core_seq_dict)})
    fx["e2e"] = e2e
    return fx


# --------------------------------------------------------------------------- ingestion (8f-3)
def gen_io(tmp):
    fx = {}
    bed_text = open("/root/reference/example_data/HTT_repeat_region.bed", "rb").read().decode()
    bed = os.path.join(tmp, "r.bed"); open(bed, "wb").write(bed_text.encode())
    regs = ref_rr.read_repeat_region_file(bed, False)
    fx["bed"] = {"text": bed_text, "regions": [[r.chrom, r.start_pos, r.end_pos, r.repeat_unit_seq, r.to_unique_id()] for r in regs]}
    rng = random.Random(5)
    fa = os.path.join(tmp, "ref.fa")
    seqs = {"chr4": rand_seq(3000, rng), "7": rand_seq(500, rng), "chrM extra words": rand_seq(90, rng).lower()}
    with open(fa, "w") as f:
        for n, s in seqs.items():
            f.write(f">{n}\n")
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
            f.write("\n")
    d = tk.fasta_file2dict(fa)
    fx["fasta"] = {"text": open(fa).read(), "names": list(d), "lens": [len(v) for v in d.values()],
                   "sha1": [hashlib.sha1(v.encode()).hexdigest() for v in d.values()]}
    cases = []
    for chrom, st, en, anchor in [("chr4", 1200, 1260, 1000), ("4", 1200, 1260, 1000), ("chr4", 300, 360, 1000),
                                  ("chr4", 2900, 2960, 1000), ("chr7", 100, 130, 50), ("7", 0, 30, 50),
                                  ("chr4", 1200, 1260, 5), ("chr4", 2990, 3001, 1000)]:
        rr = ref_rr.RepeatRegion(f"{chrom}\t{st}\t{en}\tCAG")
        rr.anchor_len = anchor
        ref_bam.extract_ref_sequence(d, rr)
        cases.append({"chrom": chrom, "start": st, "end": en, "anchor_len": anchor,
                      "left": rr.left_anchor_seq, "right": rr.right_anchor_seq, "mid": rr.mid_ref_seq,
                      "anchor_len_after": rr.anchor_len})
    fx["flanks"] = cases
    fx["one_chr"] = {n: hashlib.sha1(tk.read_one_chr_from_fasta_file(fa, n).encode()).hexdigest()
                     for n in ("chr4", "7", "chrM", "nope")}
    # reads of a region from an alignment file (nanoRepeat_bam.py:576-600) with a stand-in for pysam
    recs = [["r1", "ACGTACGTAC", [30] * 10, 900, 1400], ["r2", "GGGTTTAAAC", None, 1190, 1300],
            ["r1", "TTTT", [1, 2, 3, 4], 1250, 1254], ["r3", "", None, 1200, 1201], ["r4", None, None, 1200, 1300],
            ["r5", "CCCCC", [0, 41, 60, 93, 2], 1359, 1364], ["r6", "AAAAA", [9] * 5, 1360, 1365],
            ["r7", "TTTTT", [9] * 5, 100, 1141], ["r8", "GGGGG", [9] * 5, 100, 1140]]
    calls = []

    class FakeRead:
        def __init__(self, r):
            self.query_name, self.query_sequence, self.query_qualities, self.pos, self.end = r

    class FakeAlignmentFile:
        def __init__(self, path, mode, reference_filename=None):
            calls.append(["open", os.path.basename(path), mode, reference_filename])

        def fetch(self, chrom, start, end):
            calls.append(["fetch", chrom, start, end])
            return [FakeRead(r) for r in recs if r[3] < end and r[4] > start]

        def close(self):
            calls.append(["close"])

    ref_bam.pysam.AlignmentFile = FakeAlignmentFile
    bam_cases = []
    for chrom, st, en, flank in (("chr4", 1200, 1260, 100), ("chr4", 40, 60, 100), ("chr4", 1200, 1260, 0)):
        rr = ref_rr.RepeatRegion(f"{chrom}\t{st}\t{en}\tCAG")
        out_fq = os.path.join(tmp, "bam_out.fastq")
        del calls[:]
        ref_bam.extract_fastq_from_bam(types.SimpleNamespace(ref_fasta="ref.fa"), os.path.join(tmp, "in.bam"), rr, flank, out_fq)
        bam_cases.append({"region": [chrom, st, en], "flank": flank, "calls": list(calls), "fastq": open(out_fq).read()})
    fx["bam_extract"] = {"records": recs, "cases": bam_cases}
    fq = os.path.join(tmp, "x.fastq")
    fq_text = "@a first\nACGT\n+\nIIII\n@b\nGGNA\n+b\n!!!!\n@a again\nTT\n+\nII\n@trunc\nAC\n"
    open(fq, "w").write(fq_text)
    fx["fastq_dict"] = {"text": fq_text, "dict": ref_joint.fastq_file_to_dict(fq)}
    return fx


def _sizes(rng, spec, jitter):
    """spec: [(mean, n)], sizes as the selectors emit them (x.0 / x.5 / x.33..)."""
    out = []
    for mean, n in spec:
        for _ in range(n):
            v = mean + rng.gauss(0, jitter * (1 + mean / 50.0))
            out.append(max(0.0, round(v * 2) / 2.0))
    return out


def gen_phasing(tmp):
    """Step 4 (split_alleles.py:82-534, nanoRepeat_bam.py:502-574, nanoRepeat_joint.py:675-747) run by
    the reference itself with both of its random sources seeded; plots are switched off."""
    fx = {"cases_1d": [], "cases_2d": []}
    for mod in (ref_bam, ref_joint, ref_split):
        for fn in ("plot_repeat_counts_1d", "plot_repeat_counts_2d", "scatter_plot_with_contour_2d"):
            if hasattr(mod, fn):
                setattr(mod, fn, lambda *a, **k: None)
    rng = random.Random(77)
    specs_1d = [
        ("two_alleles", [(40, 50), (150, 45)], 1.0, dict(ploidy=2, error_rate=0.07, max_mutual_overlap=0.15, max_num_components=22, remove_noisy_reads=False), [400.0]),
        ("one_allele", [(30, 60)], 0.6, dict(ploidy=2, error_rate=0.07, max_mutual_overlap=0.15, max_num_components=22, remove_noisy_reads=False), []),
        ("noisy_third", [(20, 60), (75, 55), (130, 9)], 0.7, dict(ploidy=2, error_rate=0.07, max_mutual_overlap=0.15, max_num_components=22, remove_noisy_reads=True), []),
        ("close_alleles", [(40, 40), (46, 40)], 0.5, dict(ploidy=2, error_rate=0.03, max_mutual_overlap=0.15, max_num_components=6, remove_noisy_reads=False), []),
        ("haploid_cap", [(12, 30), (60, 30), (200, 30)], 0.8, dict(ploidy=1, error_rate=0.07, max_mutual_overlap=0.1, max_num_components=2, remove_noisy_reads=True), []),
        ("two_reads", [(10, 1), (90, 1)], 0.0, dict(ploidy=2, error_rate=0.07, max_mutual_overlap=0.15, max_num_components=22, remove_noisy_reads=False), []),
        ("one_read", [(10, 1)], 0.0, dict(ploidy=2, error_rate=0.07, max_mutual_overlap=0.15, max_num_components=22, remove_noisy_reads=False), []),
    ]
    for ci, (label, spec, jitter, par, extra) in enumerate(specs_1d):
        sizes = _sizes(rng, spec, jitter) + extra
        order = list(range(len(sizes))); rng.shuffle(order)
        rr = ref_rr.RepeatRegion()
        rr.chrom, rr.start_pos, rr.end_pos, rr.repeat_unit_seq = "chr4", 3074876, 3074933, "CAG"
        rr.no_details = False
        rr.out_prefix = os.path.join(tmp, f"ph1_{ci}")
        rr.region_fq_file = os.path.join(tmp, f"ph1_{ci}.fastq")
        reads = []
        with open(rr.region_fq_file, "w") as f:
            for j, oi in enumerate(order):
                name = f"r{j:03d}"
                rd = ref_rr.Read(); rd.read_name = name
                rd.round3_repeat_size = sizes[oi] if not (j % 17 == 5 and len(sizes) > 10) else None
                rr.read_dict[name] = rd
                reads.append([name, rd.round3_repeat_size])
                f.write(f"@{name} len=8\nACGTACGT\n+\nIIIIIIII\n")
        seed = 1000 + ci
        random.seed(seed); np.random.seed(seed)
        ref_split.output_repeat_size_1d(rr)
        ref_bam.split_allele_using_gmm_1d(rr, par["ploidy"], par["error_rate"], par["max_mutual_overlap"],
                                          par["max_num_components"], par["remove_noisy_reads"])
        rr.get_final_output()
        files = {}
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith(f"ph1_{ci}.") and fn != f"ph1_{ci}.fastq":
                files[fn.replace(f"ph1_{ci}", "PREFIX")] = open(os.path.join(tmp, fn)).read()
        fx["cases_1d"].append({"label": label, "reads": reads, "params": par, "seed": seed, "files": files,
                               "final_output": rr.final_output,
                               "num_alleles": len(rr.results.quantified_allele_list)})
    specs_2d = [
        ("htt_like", [((17, 10), 46), ((55, 7), 54)], dict(ploidy=2, error_rate=0.1, max_mutual_overlap=0.1, max_num_components=22, remove_noisy_reads=False)),
        ("single", [((30, 12), 40)], dict(ploidy=2, error_rate=0.1, max_mutual_overlap=0.1, max_num_components=22, remove_noisy_reads=False)),
        ("noisy_third", [((15, 5), 50), ((70, 20), 50), ((120, 40), 8)], dict(ploidy=2, error_rate=0.05, max_mutual_overlap=0.1, max_num_components=22, remove_noisy_reads=True)),
        ("same_axis1", [((40, 5), 40), ((40, 25), 40)], dict(ploidy=2, error_rate=0.05, max_mutual_overlap=0.1, max_num_components=5, remove_noisy_reads=False)),
        ("one_read", [((40, 5), 1)], dict(ploidy=1, error_rate=0.05, max_mutual_overlap=0.1, max_num_components=5, remove_noisy_reads=False)),
    ]
    r1 = make_repeat("chr4", 3074876, 3074933, "CAG", 200)
    r2 = make_repeat("chr4", 3074946, 3074966, "CCG", 20)
    for ci, (label, spec, par) in enumerate(specs_2d):
        rows = []
        for (m1, m2), n in spec:
            a = _sizes(rng, [(m1, n)], 0.8); b = _sizes(rng, [(m2, n)], 0.4)
            rows += list(zip(a, b))
        rng.shuffle(rows)
        fq = os.path.join(tmp, f"ph2_{ci}.fastq")
        joint = {}
        with open(fq, "w") as f:
            for j, (a, b) in enumerate(rows):
                name = f"q{j:03d}"
                joint[name] = (a, b)
                f.write(f"@{name}\nACGTACGTAC\n+\nIIIIIIIIII\n")
        prefix = os.path.join(tmp, f"ph2_{ci}")
        seed = 2000 + ci
        random.seed(seed); np.random.seed(seed)
        ref_joint.split_alleles_using_gmm_2d(par["ploidy"], par["error_rate"], par["max_mutual_overlap"],
                                             par["remove_noisy_reads"], par["max_num_components"], r1, r2,
                                             dict(joint), 0, fq, prefix)
        files = {}
        for fn in sorted(os.listdir(tmp)):
            if fn.startswith(f"ph2_{ci}.") and fn != f"ph2_{ci}.fastq":
                files[fn.replace(f"ph2_{ci}", "PREFIX")] = open(os.path.join(tmp, fn)).read().replace(fq, "IN.fastq")
        fx["cases_2d"].append({"label": label, "reads": [[n, a, b] for n, (a, b) in joint.items()], "params": par,
                               "seed": seed, "files": files})
    return fx


def gen_joint_round1(tmp):
    """Joint round 1 (nanoRepeat_joint.py:509-649) run by the reference with the oracle-backed
    aligner stand-in (best strand per template as the one primary record)."""
    fx = {"cases": []}
    rng = random.Random(31)
    for ci, (u1, u2, k1ref, k2ref, flank, alleles) in enumerate([
            ("CAG", "CCG", 19, 9, 300, [(17, 10), (55, 7)]),
            ("TATTG", "AC", 6, 12, 220, [(8, 30), (25, 12)])]):
        mid = "CAACAGCCGCCAC" if ci == 0 else ""
        left, right = rand_seq(flank + 50, rng), rand_seq(flank + 80, rng)
        chrom = left + u1 * k1ref + mid + u2 * k2ref + right
        s1 = len(left); e1 = s1 + len(u1) * k1ref; s2 = e1 + len(mid); e2 = s2 + len(u2) * k2ref
        r1 = make_repeat("chrJ", s1, e1, u1, 60); r2 = make_repeat("chrJ", s2, e2, u2, 40)
        r1.max_size += 10; r2.max_size += 10
        fq = os.path.join(tmp, f"j1_{ci}.fastq")
        reads = []
        with open(fq, "w") as f:
            for i in range(16):
                a, b = alleles[i % 2]
                seq = left[-(150 + 5 * i):] + u1 * a + mid + u2 * b + right[:140 + 7 * i]
                seq = mutate(seq, 0.025, 0.015, 0.03, rng) if i % 5 else seq
                if i == 7:
                    seq = seq[:len(seq) // 2]                  # no right anchor
                if i == 11:
                    seq = rand_seq(400, rng)                    # unrelated read
                if i == 13:
                    seq = left[-150:] + u1 * 70 + mid + u2 * b + right[:150]   # longer than the template's repeat
                if i % 3 == 1:
                    seq = revcomp(seq)
                name = f"jr{i:02d}"
                reads.append([name, seq])
                f.write(f"@{name}\n{seq}\n+\n{'I' * len(seq)}\n")
        out_dir = os.path.join(tmp, f"j1_{ci}.tmp"); os.makedirs(out_dir)
        ref_joint.pymm2.main = Recorder(oracle_aligner(both_strands=True))
        est = ref_joint.initial_estimate_repeat_size(chrom, fq, "ont", 1, r1, r2, flank, out_dir)
        fx["cases"].append({
            "chrom": chrom, "repeat1": f"chrJ:{s1}:{e1}:{u1}:60", "repeat2": f"chrJ:{s2}:{e2}:{u2}:40",
            "max_anchor_len": flank, "reads": reads,
            "round1_paf": open(os.path.join(out_dir, "round1.paf")).read(),
            "repeat1_count_range": {k: list(v) for k, v in est.repeat1_count_range_dict.items()},
            "repeat2_count_range": {k: list(v) for k, v in est.repeat2_count_range_dict.items()},
            "potential_repeat_region": {k: list(v) for k, v in est.potential_repeat_region_dict.items()},
            "bad_reads": sorted(est.bad_reads_set)})
    return fx


def main():
    which = set(sys.argv[1:]) or {"1d", "2d", "wide", "upstream", "io", "phasing", "joint_round1"}
    gens = (("1d", "ref_1d.json", gen_1d), ("2d", "ref_2d.json", gen_2d), ("wide", "ref_wide.json", gen_wide),
            ("upstream", "ref_upstream.json", gen_upstream),
            ("io", "ref_io.json", gen_io), ("phasing", "ref_phasing.json", gen_phasing),
            ("joint_round1", "ref_joint_round1.json", gen_joint_round1))
    tmp = tempfile.mkdtemp(prefix="nr_golden_")
    out = []
    try:
        for key, name, gen in gens:
            if key in which:
                out.append((name, gen(tmp)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for name, fx in out:
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fx, f, indent=1, sort_keys=True)
            f.write("\n")
        print("wrote", name, os.path.getsize(os.path.join(HERE, name)), "bytes")


if __name__ == "__main__":
    main()
